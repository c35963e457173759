/*
 * zamd_engine.hip — HIP kernels (gfx950) and host orchestration of the chunked
 * DEFLATE engine.  C ABI in include/zamd.h; per-lane logic in zamd_core.h.
 *
 * Pipeline for levels 4-9 (one batch of chunks per pass, up to four batches in
 * flight on four streams):
 *   k_chain   hash-chain build: INSERT_STRING for every position, head table in
 *             LDS, writes the distance to the previous same-hash string
 *   k_match   longest_match for every position ahead of the parse; window bytes
 *             and chain links of a 16 Ki-position tile (+32 KiB halo) staged in LDS
 *             (k_follow, zamd_follow_impl.h: the search that follows the parse, for
 *             deflateTune values the precomputed records cannot express)
 *   k_resolve / k_hop / k_expand
 *             deflate_slow's lazy evaluation over the match records: per-position
 *             hop words in parallel, the serial walk over hops, symbol stream and
 *             block table (lit_bufsize rule) expanded in parallel
 *             (k_parse: the serial replay, used for Z_RLE / Z_HUFFMAN_ONLY / level 0)
 *   k_plan    per block: _tr_tally histograms with LDS atomics, build_tree x3,
 *             stored/static/dynamic choice, code tables, header bits
 *   k_layout  per chunk bit layout, k_scan exclusive scan -> output offsets
 *   k_emit    compress_block: prefix sums over code lengths, LDS bit staging,
 *             OR into the zeroed output
 * Levels 1-3: k_fastw / k_fastr / k_fast (zamd_fast_impl.h, by chain budget) replace chain/match/parse.
 * Inflate: zamd_inflate_impl.h.
 */
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "zamd_core.h"
#include "../../include/zamd.h"

#ifndef ZK_TILE
#define ZK_TILE      16384u                   /* positions per k_match workgroup */
#endif
#define ZK_HALO      32768u
#define ZK_WIN_BYTES (ZK_HALO + ZK_TILE + 320u)
#define ZK_MATCH_THREADS 1024
#ifndef ZK_MATCH_C
#define ZK_MATCH_ASM 1                     /* k_match's candidate round written by hand; -DZK_MATCH_C builds the C++ form of the same round */
#endif
#ifndef ZK_MATCH_STEPS
#define ZK_MATCH_STEPS 8                     /* candidate steps between two looks at the number of stopped lanes */
#endif
#define ZK_PLAN_THREADS  64                   /* one wave per block: the tree builds are serial, what counts is blocks in flight */
#define ZK_EMIT_THREADS  256
#define ZK_STAGE_WORDS   400                  /* 31 + 256*48 + 31 bits */

enum { ZE_CHAIN_ORDER = 1, ZE_BLOCK_OVERFLOW = 2, ZE_OUT_OVERFLOW = 4, ZE_INFLATE_DATA = 8, ZE_TREE_DEPTH = 16 };

struct ZkJob {
    const uint8_t *in; uint64_t n;
    uint64_t chunk0; uint32_t nchunks; int finish; uint32_t head;
    uint64_t last_chunk;          /* absolute number of the call's last chunk: the one P.tail applies to */
    uint32_t prime_bits, prime_val;  /* deflatePrime bits in front of the call's first chunk */
    ZcParams P;
    uint16_t *pdist; uint64_t pdist_base, pdist_len;   /* chain links: of the hash chains; where k_match4 searches, of the four-byte chains once k_reach has read those */
    uint2 *rec; uint64_t rec_base;
    uint32_t *sym;
    uint32_t *hop; uint32_t *arr; uint32_t *nhop;
    ZcBlock *blk; ZcBlockTab *tab;
    uint32_t *nblk; uint32_t *nsym;
    uint64_t *chunk_bytes; uint64_t *sync_bit;
    uint64_t *chunk_off;          /* global table, indexed by absolute chunk number */
    uint64_t *running_total;      /* bytes written by earlier batches */
    uint32_t *out32; uint64_t out_cap;
    uint32_t *err;
    uint16_t *fhead;              /* levels 1-3: head table of every chunk of the batch, 65536 entries each */
    uint64_t *splits; uint32_t split_parts;   /* split records of every chunk (global table, split_parts - 1 per chunk), or null */
};

/* ------------------------------------------------------------------ k_chain
 * INSERT_STRING (deflate.c:186-189) for every position of one chunk (plus its
 * dictionary), head table in LDS.  The strings must be inserted in position
 * order per hash value, which is all the chains need: the 16 wavefronts of the
 * workgroup split the HASH VALUES between them (hash mod 16), so each wave owns
 * its slice of the head table.  Per round of 5120 positions (what fits beside the
 * table in 160 KiB of LDS): every thread hashes
 * five positions and files them under the wavefronts that own their hashes (a counting
 * sort by owner that keeps position order: rank inside the filing wavefront from
 * a returning LDS add, offsets from the 16 x 16 table of counts, which every
 * wavefront sums for itself), then every
 * wavefront inserts the positions filed under it, 64 to an instruction, all lanes
 * busy.  The LDS atomic max returns the previous (older) position with that hash —
 * lanes and successive steps hold increasing positions — which becomes the chain
 * link.  (LDS serves the lanes of one atomic instruction that hit one address in
 * lane order; a violation would show as a chain that runs forwards and raises
 * ZE_CHAIN_ORDER.) */
#define ZK_CHAIN_THREADS 1024
#ifndef ZK_CHAIN_NP
#define ZK_CHAIN_NP      5u                       /* positions per thread and round */
#endif
#define ZK_CHAIN_ROUND   (1024u * ZK_CHAIN_NP)
#define ZK_CHAIN_WSPAN   (64u * ZK_CHAIN_NP)      /* consecutive positions of a round that one wavefront takes, 64 at a time */
/* A workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global load and store of the
 * wave (s_waitcnt vmcnt(0)): with a prefetch in flight that exposes a trip to memory at every barrier.  The data handed
 * between waves here lives in LDS; global loads are consumed by the thread that issued them (the compiler waits for
 * those itself), global stores are read by later kernels only. */
#define ZK_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
/* inclusive sum over the 16 lanes of a row, one DPP add per step (lanes shifted in from outside the row read zero) */
__device__ inline uint32_t zk_row_scan16(uint32_t x)
{
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true);     /* row_shr:1 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, true);     /* row_shr:2 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xf, 0xf, true);     /* row_shr:4 */
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xf, 0xf, true);     /* row_shr:8 */
    return x;
}
/* inclusive sum over the 64 lanes of a wavefront with DPP row shifts and broadcasts (no LDS) */
__device__ inline uint32_t zk_wave_scan(uint32_t v)
{
    uint32_t s = v;
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true);    /* row_shr:1 */
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true);    /* row_shr:2 */
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, true);    /* row_shr:3 */
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xe, true);    /* row_shr:4, banks 1-3 */
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xf, 0xc, true);    /* row_shr:8, banks 2-3 */
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x142, 0xa, 0xf, true);    /* row_bcast:15 into rows 1, 3 */
    s += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x143, 0xc, 0xf, true);    /* row_bcast:31 into rows 2, 3 */
    return s;
}
/* inclusive maximum over the 64 lanes of a wavefront, the same way */
__device__ inline uint32_t zk_wave_scan_max(uint32_t v)
{
    uint32_t s = v;
    s = max(s, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
    s = max(s, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
    s = max(s, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x113, 0xf, 0xf, true));
    s = max(s, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x114, 0xf, 0xe, true));
    s = max(s, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x118, 0xf, 0xc, true));
    s = max(s, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x142, 0xa, 0xf, true));
    s = max(s, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)s, 0x143, 0xc, 0xf, true));
    return s;
}
/* mode 0: the hash chains of the reference (UPDATE_HASH with the stream's hash_bits).  The same build serves two chains
 * of this engine's own, each 15 bits wide and one pass, each taking the previous one's place in J.pdist once its reader
 * has run: mode 2, strings with the same zc_hash3e of their three bytes (k_first3 finds the nearest three-byte match
 * along them), and mode 1, the FOUR-BYTE chains (zc_hash4 of the first four bytes) that k_match walks. */
enum { ZK_CHAIN_REF = 0, ZK_CHAIN_FOUR = 1, ZK_CHAIN_THREE = 2 };
__global__ __launch_bounds__(ZK_CHAIN_THREADS) void k_chain(ZkJob J, int mode)
{
    const bool four = mode == ZK_CHAIN_FOUR;
    __shared__ uint32_t head[32768];
    __shared__ uint16_t s_hash[ZK_CHAIN_ROUND];       /* hash of every position of the round, 0xffff = not part of this pass */
    __shared__ uint16_t s_link[ZK_CHAIN_ROUND];
    __shared__ uint16_t s_sorted[ZK_CHAIN_ROUND];     /* filed positions ordered by (owner, position); bit 15: opens a run */
    /* [owner][wavefront]: how many positions the wavefront files under the owner.  Rows of 20 words: a wavefront's 64 counting
     * atomics go to one column, and with rows of 16 that column lies in two LDS banks (eight addresses each); rows of 20
     * spread it over eight banks and keep the rows aligned for 16-byte reads */
    __shared__ __attribute__((aligned(16))) uint32_t s_cnt[16][20];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = (uint32_t)__builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const uint64_t c = J.chunk0 + blockIdx.x;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, c, J.finish, J.head, J.P.wsize);
    const uint32_t E = g.dict_len + g.len;
    /* positions with three (four) bytes available in the whole input */
    uint64_t avail = J.n - g.dstart;
    uint32_t Ex = E;
    const uint32_t more = four ? 3u : 2u;
    if ((uint64_t)E + more > avail) Ex = avail >= more ? (uint32_t)(avail - more) : 0;
    const uint32_t out_from = blockIdx.x == 0 ? 0u : g.dict_len;
    const uint8_t *src = J.in + g.dstart;
    uint16_t *pd = J.pdist + (g.dstart - J.pdist_base);
    const uint32_t passes = (mode == ZK_CHAIN_REF && J.P.hash_bits > 15) ? (1u << (J.P.hash_bits - 15)) : 1u;
    /* the two aligned words that hold bytes a..a+2, fetched one round ahead, without a branch so that nothing waits for
     * them before the next round uses them.  A dictionary of any length in front of the first chunk leaves a-coordinate 0
     * at any address: `mis` is its offset from a dword boundary.  Positions past the last string read the last string's
     * words (and are not hashed); the second word is read only when the string reaches into it: an aligned word that
     * holds a byte of the input lies in the input's pages. */
    const uint32_t mis = (uint32_t)((uintptr_t)src & 3u);
    const uint8_t *const srca = src - mis;
    auto fetch = [&](uint32_t a, uint32_t &w0, uint32_t &w1) {
        const uint32_t x = (a < Ex ? a : Ex - 1u) + mis, al = x & ~3u;
        w0 = *(const uint32_t *)(srca + al);
        w1 = *(const uint32_t *)(srca + al + (four ? ((x & 3u) ? 4u : 0u) : ((x & 2u) << 1)));
    };
    for (uint32_t pass = 0; pass < passes; pass++) {
        for (uint32_t i = tid; i < 32768; i += ZK_CHAIN_THREADS) head[i] = 0;
        if (tid < 320) (&s_cnt[0][0])[tid] = 0;
        uint32_t w0[ZK_CHAIN_NP], w1[ZK_CHAIN_NP];
#pragma unroll
        for (uint32_t u = 0; u < ZK_CHAIN_NP; u++) { w0[u] = 0; w1[u] = 0; if (Ex) fetch(wv * ZK_CHAIN_WSPAN + u * 64u + lane, w0[u], w1[u]); }
        __syncthreads();
        for (uint32_t r0 = 0; r0 < Ex; r0 += ZK_CHAIN_ROUND) {
            /* 1. the hashes of this thread's positions: the wavefront takes a stretch of the round, 64 consecutive positions
             *    at a time.  A stretch of equal hashes inside one 64 (a run of one byte value, mostly) needs the table twice
             *    only: the string that opens it reads the previous occurrence, the one that closes it leaves its position
             *    behind; every string in between links to its neighbour, one back.  Only opening and closing strings are
             *    filed; `rank` numbers them among those their wavefront files under the same owner, in position order: the
             *    counting atomics of one 64 number their lanes in lane order, and a wavefront's LDS instructions run in
             *    program order. */
            uint32_t hv[ZK_CHAIN_NP], rank[ZK_CHAIN_NP], own[ZK_CHAIN_NP];
            bool filed[ZK_CHAIN_NP], opens[ZK_CHAIN_NP];
#pragma unroll
            for (uint32_t u = 0; u < ZK_CHAIN_NP; u++) {
                const uint32_t a = r0 + wv * ZK_CHAIN_WSPAN + u * 64u + lane;
                hv[u] = 0xffffu; rank[u] = 0;
                if (a < Ex) {
                    uint32_t b3 = __builtin_amdgcn_alignbyte(w1[u], w0[u], (a + mis) & 3u);
                    uint32_t h = four ? zc_hash4(b3) : mode == ZK_CHAIN_THREE ? zc_hash3e(b3) :
                                 zc_hash3(b3 & 0xffu, (b3 >> 8) & 0xffu, (b3 >> 16) & 0xffu, J.P.hash_shift, J.P.hash_mask);
                    if ((h >> 15) == pass) hv[u] = h & 32767u;
                }
                const uint32_t hp = __shfl_up(hv[u], 1), hn = __shfl_down(hv[u], 1);
                opens[u] = lane == 0 || hp != hv[u];
                const bool closes = lane == 63 || hn != hv[u];
                filed[u] = hv[u] != 0xffffu && (opens[u] || closes);
                s_hash[wv * ZK_CHAIN_WSPAN + u * 64u + lane] = (uint16_t)hv[u];
                s_link[wv * ZK_CHAIN_WSPAN + u * 64u + lane] = 1;
                /* the owner mixes all hash bits: the low four alone are the low nibble of the string's third byte */
                own[u] = (hv[u] ^ (hv[u] >> 5) ^ (hv[u] >> 10)) & 15u;
                if (filed[u]) rank[u] = atomicAdd(&s_cnt[own[u]][wv], 1u);
                fetch(a + ZK_CHAIN_ROUND, w0[u], w1[u]);
            }
            ZK_LDS_BARRIER();
            /* 2. counts -> offsets, by every wavefront for itself (lane d of every row of 16 takes owner d): where owner
             *    d's stretch of s_sorted starts, and where inside it the positions of this wavefront go */
            uint32_t seg0, seg1;
            {
                const uint32_t d = lane & 15u;
                uint32_t run = 0, before = 0;
#pragma unroll
                for (uint32_t q = 0; q < 16; q += 4) {
                    const uint4 n4 = *(const uint4 *)&s_cnt[d][q];
                    const uint32_t n[4] = {n4.x, n4.y, n4.z, n4.w};
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) { if (q + k == wv) before = run; run += n[k]; }
                }
                const uint32_t incl = zk_row_scan16(run);             /* scan of the owners' totals over each row of 16 lanes */
                const uint32_t start = incl - run;
                /* 3. file the positions: lane `own` of this lane's row holds that owner's numbers */
#pragma unroll
                for (uint32_t u = 0; u < ZK_CHAIN_NP; u++) {
                    const uint32_t base = __shfl(start + before, own[u], 16);
                    if (filed[u]) s_sorted[base + rank[u]] = (uint16_t)((wv * ZK_CHAIN_WSPAN + u * 64u + lane) | (opens[u] ? 0x8000u : 0u));
                }
                /* every row holds the same numbers */
                seg0 = (uint32_t)__builtin_amdgcn_readlane((int)start, (int)wv);
                seg1 = (uint32_t)__builtin_amdgcn_readlane((int)incl, (int)wv);
            }
            ZK_LDS_BARRIER();
            /* 4. every wavefront enters the strings whose hash it owns, in position order */
            /* two entries per lane at a time (the second 64 further on: the same lane's next one, so that entries of one hash
             * value still reach the table in position order — a wavefront's LDS instructions run in program order) */
            for (uint32_t j = seg0 + lane; j < seg1; j += 128u) {
                const bool two = j + 64u < seg1;
                const uint32_t e0 = s_sorted[j], e1 = two ? s_sorted[j + 64u] : 0u;
                const uint32_t q0 = e0 & 0x7fffu, q1 = e1 & 0x7fffu;
                const uint32_t h0 = s_hash[q0], h1 = s_hash[q1], a0 = r0 + q0, a1 = r0 + q1;
                const uint32_t old0 = atomicMax(&head[h0], a0 + 1);
                uint32_t old1 = 0;
                if (two) old1 = atomicMax(&head[h1], a1 + 1);
                if (old0 > a0 || (two && old1 > a1)) atomicOr(J.err, (uint32_t)ZE_CHAIN_ORDER);
                if (e0 & 0x8000u) {
                    const uint32_t d = old0 ? a0 + 1 - old0 : 0;
                    s_link[q0] = (uint16_t)(d > 32767u ? 0u : d);
                }
                if (two && (e1 & 0x8000u)) {
                    const uint32_t d = old1 ? a1 + 1 - old1 : 0;
                    s_link[q1] = (uint16_t)(d > 32767u ? 0u : d);
                }
            }
            if (tid < 320) (&s_cnt[0][0])[tid] = 0;        /* read for the last time in step 2 */
            ZK_LDS_BARRIER();
#pragma unroll
            for (uint32_t u = 0; u < ZK_CHAIN_NP; u++) {
                const uint32_t a = r0 + wv * ZK_CHAIN_WSPAN + u * 64u + lane;
                if (a < Ex && a >= out_from && hv[u] != 0xffffu) pd[a] = s_link[wv * ZK_CHAIN_WSPAN + u * 64u + lane];
            }
            /* s_link and s_hash of these positions are written again by this thread, and read by others a barrier later */
        }
        __syncthreads();
    }
    /* strings without three (four) bytes of input are never inserted */
    for (uint32_t a = Ex + tid; a < E; a += ZK_CHAIN_THREADS) if (a >= out_from) pd[a] = 0;
}

/* ------------------------------------------------------------------ k_match */
#define ZK_PD_OFF   ZK_WIN_BYTES                 /* byte offset of the link image inside the LDS block */
#define ZK_LDS_BYTES (ZK_WIN_BYTES + 2u * (ZK_HALO + ZK_TILE))
#define ZK_PD_NONE  0xffffu                      /* "no earlier string" in the LDS link image */

struct LdsWin {
    const uint32_t *w; uint32_t a0;
    __device__ uint32_t load4(uint32_t a) const
    {
        uint32_t off = a - a0, i = off >> 2;
        return __builtin_amdgcn_alignbyte(w[i + 1], w[i], off & 3u);
    }
};
struct LdsPd {
    const uint16_t *pd; uint32_t a0;
    __device__ uint32_t operator()(uint32_t a) const { uint32_t v = pd[a - a0]; return v == ZK_PD_NONE ? 0u : v; }
};

/* Input bytes of a-coordinates a0 .. a0 + nbytes - 1 of a chunk into LDS (zero outside the input). */
__device__ inline void zk_load_window(const ZkJob &J, const ZcChunkGeom &g, int64_t a0, uint32_t *s_win, uint32_t nbytes, uint32_t tid)
{
    if (a0 >= 0 && g.dstart + (uint64_t)a0 + nbytes <= J.n && nbytes / 16u >= 3u * ZK_MATCH_THREADS && nbytes / 16u < 4u * ZK_MATCH_THREADS) {
        /* the whole image lies inside the input (every tile but a chunk's first few and the input's last): the loads of a
         * thread are all in flight before the first is stored */
        const uint4 *src = (const uint4 *)(J.in + g.dstart + a0);
        const uint4 q0 = src[tid], q1 = src[tid + ZK_MATCH_THREADS], q2 = src[tid + 2u * ZK_MATCH_THREADS];
        uint4 q3 = make_uint4(0, 0, 0, 0);
        const bool more = tid + 3u * ZK_MATCH_THREADS < nbytes / 16u;
        if (more) q3 = src[tid + 3u * ZK_MATCH_THREADS];
        ((uint4 *)s_win)[tid] = q0; ((uint4 *)s_win)[tid + ZK_MATCH_THREADS] = q1; ((uint4 *)s_win)[tid + 2u * ZK_MATCH_THREADS] = q2;
        if (more) ((uint4 *)s_win)[tid + 3u * ZK_MATCH_THREADS] = q3;
        return;
    }
    for (uint32_t v = tid; v < nbytes / 16; v += ZK_MATCH_THREADS) {
        int64_t a = a0 + 16 * (int64_t)v;
        int64_t abs = (int64_t)g.dstart + a;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (a >= 0 && (uint64_t)abs + 16 <= J.n) q = *(const uint4 *)(J.in + abs);
        else if (a + 16 > 0 && abs < (int64_t)J.n) {
            uint32_t t[4] = {0, 0, 0, 0};
            for (int k = 0; k < 16; k++) {
                int64_t ak = a + k, absk = abs + k;
                if (ak >= 0 && absk < (int64_t)J.n) t[k >> 2] |= (uint32_t)J.in[absk] << (8 * (k & 3));
            }
            q = make_uint4(t[0], t[1], t[2], t[3]);
        }
        ((uint4 *)s_win)[v] = q;
    }
}
/* Chain links of a-coordinates a0 .. a0 + count - 1 into LDS; "none" (0) is stored as 0xffff so that adding it to a
 * distance always exceeds MAX_DIST. */
__device__ inline uint4 zk_links_none(uint4 q)
{
    uint32_t t[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if ((t[k] & 0xffffu) == 0) t[k] |= 0xffffu;
        if ((t[k] >> 16) == 0) t[k] |= 0xffff0000u;
    }
    return make_uint4(t[0], t[1], t[2], t[3]);
}
__device__ inline void zk_load_links(const ZkJob &J, const uint16_t *links, const ZcChunkGeom &g, int64_t a0, uint16_t *s_pd, uint32_t count, uint32_t tid)
{
    if (a0 >= 0 && g.dstart + (uint64_t)a0 - J.pdist_base + count <= J.pdist_len && count == 6u * 8u * ZK_MATCH_THREADS) {
        /* all links exist: six loads per thread in flight together */
        const uint4 *src = (const uint4 *)(links + (g.dstart + (uint64_t)a0 - J.pdist_base));
        uint4 q[6];
#pragma unroll
        for (uint32_t k = 0; k < 6; k++) q[k] = src[tid + k * ZK_MATCH_THREADS];
#pragma unroll
        for (uint32_t k = 0; k < 6; k++) ((uint4 *)s_pd)[tid + k * ZK_MATCH_THREADS] = zk_links_none(q[k]);
        return;
    }
    for (uint32_t v = tid; v < count / 8; v += ZK_MATCH_THREADS) {
        int64_t a = a0 + 8 * (int64_t)v;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (a >= 0) {
            uint64_t idx = g.dstart + (uint64_t)a - J.pdist_base;
            if (idx + 8 <= J.pdist_len) q = *(const uint4 *)(links + idx);
        }
        uint32_t t[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((t[k] & 0xffffu) == 0) t[k] |= 0xffffu;
            if ((t[k] >> 16) == 0) t[k] |= 0xffff0000u;
        }
        ((uint4 *)s_pd)[v] = make_uint4(t[0], t[1], t[2], t[3]);
    }
}

/* ------------------------------------------------------------------ k_first3, k_reach
 * What k_match needs to know about a position before it walks the four-byte chain (zamd_core.h, "the same search
 * along four-byte chains"): the nearest earlier string with the same three bytes, and how far back the chain budget
 * reaches — the distance to the position's max_chain-th (and max_chain/4-th) predecessor in the hash chain.
 *
 * k_first3: one workgroup per tile of 16 Ki positions, window bytes and the links of the exact three-byte chains
 * (k_chain in mode ZK_CHAIN_THREE) of tile + 32 KiB halo in LDS as in k_match.  The nearest member of that chain is the
 * match unless two different strings share a hash value; then the walk goes on, a step or two. */
__global__ __launch_bounds__(ZK_MATCH_THREADS) void k_first3(ZkJob J, uint32_t tiles_per_chunk)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_lds[ZK_LDS_BYTES / 4];
    uint32_t *s_win = s_lds;
    uint16_t *s_pd = (uint16_t *)(s_lds + ZK_PD_OFF / 4);
    const uint32_t tid = threadIdx.x;
    const uint32_t lc = blockIdx.x / tiles_per_chunk, tile = blockIdx.x % tiles_per_chunk;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    if ((uint64_t)tile * ZK_TILE >= g.len) return;
    const uint32_t E = g.dict_len + g.len;
    const uint32_t t0 = g.dict_len + tile * ZK_TILE;
    const uint32_t tlen = (E - t0) < ZK_TILE ? (E - t0) : ZK_TILE;
    const int64_t a0 = (int64_t)t0 - (int64_t)ZK_HALO;
    zk_load_window(J, g, a0, s_win, ZK_WIN_BYTES, tid);
    zk_load_links(J, J.pdist, g, a0, s_pd, ZK_HALO + ZK_TILE, tid);
    __syncthreads();
    LdsWin win; win.w = s_win; win.a0 = (uint32_t)a0;
    LdsPd pd; pd.pd = s_pd; pd.a0 = (uint32_t)a0;
    /* the word waits where the symbols will be: those are written once k_match is through */
    uint32_t *const f3w = J.sym + (uint64_t)lc * J.P.chunk_size + tile * ZK_TILE;
    /* zc_first3 for the thread's positions i = tid, tid + 1024, ...  The nearest member is the answer nearly
     * everywhere, so every position takes its first step in straight-line code, sixteen positions' reads in flight
     * together.  What is left — two strings that share a hash value — walks on one chain step per turn of a loop,
     * whatever position a lane is at: a rare string that shares its value with a frequent one walks past hundreds of
     * members, and a loop over positions would keep the whole wavefront waiting for it at every one of them. */
    const uint32_t max_dist = J.P.max_dist, la0 = (uint32_t)a0;
    uint32_t pend = 0;
    {
        uint32_t d[ZK_TILE / ZK_MATCH_THREADS], s3[ZK_TILE / ZK_MATCH_THREADS], c3[ZK_TILE / ZK_MATCH_THREADS];
#pragma unroll
        for (uint32_t j = 0; j < ZK_TILE / ZK_MATCH_THREADS; j++) {
            const uint32_t i = j * ZK_MATCH_THREADS + tid, pp = t0 + (i < tlen ? i : 0u);
            d[j] = s_pd[pp - la0];
            s3[j] = win.load4(pp);
        }
#pragma unroll
        for (uint32_t j = 0; j < ZK_TILE / ZK_MATCH_THREADS; j++) {
            const uint32_t i = j * ZK_MATCH_THREADS + tid, pp = t0 + (i < tlen ? i : 0u);
            const bool ok = d[j] <= max_dist && d[j] <= pp;                  /* "none" is 0xffff */
            c3[j] = win.load4(ok ? pp - d[j] : pp);
        }
#pragma unroll
        for (uint32_t j = 0; j < ZK_TILE / ZK_MATCH_THREADS; j++) {
            const uint32_t i = j * ZK_MATCH_THREADS + tid, pp = t0 + i;
            if (i >= tlen) continue;
            const bool ok = d[j] <= max_dist && d[j] <= pp && E - pp >= ZC_MIN_MATCH;
            if (!ok) f3w[i] = 0u;
            else if (((c3[j] ^ s3[j]) & 0x00ffffffu) == 0u) f3w[i] = d[j];
            else pend |= 1u << j;
        }
    }
    uint32_t i = 0, p = 0, cur = 0, scan3 = 0;
    bool fresh = true;
    while (pend) {
        if (fresh) {
            i = ((uint32_t)__ffs((int)pend) - 1u) * ZK_MATCH_THREADS + tid; p = t0 + i;
            cur = p - s_pd[p - la0];                     /* the member the first step found wanting */
            scan3 = win.load4(p) & 0x00ffffffu;
            fresh = false;
        }
        const uint32_t d = pd(cur);
        uint32_t res = 0;
        bool done = d == 0 || d > cur;
        if (!done) {
            cur -= d;
            if (p - cur > max_dist) done = true;
            else if ((win.load4(cur) & 0x00ffffffu) == scan3) { done = true; res = p - cur; }
        }
        if (done) { f3w[i] = res; pend &= pend - 1u; fresh = true; }
    }
}

/* k_reach: the k-th predecessor in the hash chain by pointer jumping.  One workgroup per 32 Ki positions; the image of
 * the hash-chain links of those and the 32 Ki in front of them is 65 536 16-bit entries, so that every 16-bit index
 * lies inside it and the gathers need no bounds.  Entry x holds the distance from x to its 2^r-th predecessor after
 * round r; round r + 1 adds the entry found that far back: A[x] += A[x - A[x]], for every entry of the image, in place
 * (all reads of a round are made before a barrier, all writes behind it), two entries to a register and the sums and
 * differences two at a time.  A distance of 0x8000 or more stands for "none within 32767" — the loader writes 0xffff
 * for the end of a chain, the sums saturate, so such an entry never comes back below 0x8000 — and entry 0 is none by
 * decree: x - A[x] saturates at 0, the index that a link out of the image's front becomes (nothing the tile can use
 * lies that far back).  A tile position collects the rounds that are bits of its budgets. */
typedef unsigned short zk_u16x2 __attribute__((ext_vector_type(2)));
#define ZK_REACH_TILE 32768u
#define ZK_REACH_NP   ((ZK_HALO + ZK_REACH_TILE) / 2u / ZK_MATCH_THREADS)       /* pairs of entries per thread: 32 */
__device__ inline zk_u16x2 zk_pair(uint32_t w) { zk_u16x2 v; v.x = (unsigned short)w; v.y = (unsigned short)(w >> 16); return v; }
__device__ inline uint32_t zk_word(zk_u16x2 v) { return (uint32_t)v.x | ((uint32_t)v.y << 16); }
__device__ inline zk_u16x2 zk_gather(const uint16_t *img, zk_u16x2 idx) { zk_u16x2 v; v.x = img[idx.x]; v.y = img[idx.y]; return v; }
__global__ __launch_bounds__(ZK_MATCH_THREADS) void k_reach(ZkJob J, uint32_t tiles_per_chunk)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_img[(ZK_HALO + ZK_REACH_TILE) / 2u];
    uint16_t *const img = (uint16_t *)s_img;
    const uint32_t tid = threadIdx.x;
    const uint32_t lc = blockIdx.x / tiles_per_chunk, tile = blockIdx.x % tiles_per_chunk;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    if ((uint64_t)tile * ZK_REACH_TILE >= g.len) return;
    const uint32_t E = g.dict_len + g.len;
    const uint32_t t0 = g.dict_len + tile * ZK_REACH_TILE;
    const uint32_t tlen = (E - t0) < ZK_REACH_TILE ? (E - t0) : ZK_REACH_TILE;
    const int64_t a0 = (int64_t)t0 - (int64_t)ZK_HALO;
    {
        const uint64_t i0 = g.dstart + (uint64_t)a0 - J.pdist_base;
        if (a0 >= 0 && i0 + (ZK_HALO + ZK_REACH_TILE) <= J.pdist_len) {
            /* all links exist: eight loads per thread in flight together */
            const uint4 *src = (const uint4 *)(J.pdist + i0);
            uint4 q[8];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) q[k] = src[tid + k * ZK_MATCH_THREADS];
#pragma unroll
            for (uint32_t k = 0; k < 8; k++) ((uint4 *)s_img)[tid + k * ZK_MATCH_THREADS] = zk_links_none(q[k]);
        } else zk_load_links(J, J.pdist, g, a0, img, ZK_HALO + ZK_REACH_TILE, tid);
    }
    __syncthreads();
    if (tid == 0) img[0] = ZK_PD_NONE;
    __syncthreads();
    const uint32_t max_chain = J.P.chain, qchain = max_chain >> 2;
    /* R: where the walk of `budget` steps stands, as an index into the image (0: it has left the picture) */
    zk_u16x2 RF[ZK_REACH_NP / 2], RQ[ZK_REACH_NP / 2];
#pragma unroll
    for (uint32_t j = 0; j < ZK_REACH_NP / 2; j++) {
        const uint32_t x = 2u * ((j + ZK_REACH_NP / 2) * ZK_MATCH_THREADS + tid);
        RF[j] = zk_pair(x | ((x + 1u) << 16)); RQ[j] = RF[j];
    }
    const uint32_t top = 31u - (uint32_t)__clz((int)max_chain);
    for (uint32_t r = 0; r <= top; r++) {
        if ((max_chain >> r) & 1u) {
#pragma unroll
            for (uint32_t j = 0; j < ZK_REACH_NP / 2; j++) RF[j] = __builtin_elementwise_sub_sat(RF[j], zk_gather(img, RF[j]));
        }
        if ((qchain >> r) & 1u) {
#pragma unroll
            for (uint32_t j = 0; j < ZK_REACH_NP / 2; j++) RQ[j] = __builtin_elementwise_sub_sat(RQ[j], zk_gather(img, RQ[j]));
        }
        if (r == top) break;
        uint32_t nv[ZK_REACH_NP];
#pragma unroll
        for (uint32_t j0 = 0; j0 < ZK_REACH_NP; j0 += 8) {
            uint32_t own[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) own[u] = s_img[(j0 + u) * ZK_MATCH_THREADS + tid];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) {
                const uint32_t x = 2u * ((j0 + u) * ZK_MATCH_THREADS + tid);
                const zk_u16x2 o = zk_pair(own[u]);
                const zk_u16x2 idx = __builtin_elementwise_sub_sat(zk_pair(x | ((x + 1u) << 16)), o);
                nv[j0 + u] = zk_word(__builtin_elementwise_add_sat(o, zk_gather(img, idx)));
            }
            asm volatile("" ::: "memory");
        }
        if (tid == 0) nv[0] |= 0xffffu;               /* entry 0 stays "none" */
        ZK_LDS_BARRIER();
#pragma unroll
        for (uint32_t j = 0; j < ZK_REACH_NP; j++) s_img[j * ZK_MATCH_THREADS + tid] = nv[j];
        ZK_LDS_BARRIER();
    }
    /* the word waits where the hop words will be: those are written once k_match is through */
    uint32_t *const limw = J.hop + (uint64_t)lc * J.P.chunk_size + tile * ZK_REACH_TILE;
    const uint16_t *const link = J.pdist + (g.dstart + t0 - J.pdist_base);      /* the position's own link, from memory again */
#pragma unroll
    for (uint32_t j = 0; j < ZK_REACH_NP / 2; j++) {
        const uint32_t x = 2u * ((j + ZK_REACH_NP / 2) * ZK_MATCH_THREADS + tid), i = x - ZK_HALO;
        if (i >= tlen) continue;
        const uint32_t rf[2] = {RF[j].x, RF[j].y}, rq[2] = {RQ[j].x, RQ[j].y};
        uint32_t w[2] = {0, 0};
#pragma unroll
        for (uint32_t e = 0; e < 2; e++) {
            if (i + e >= tlen) break;
            uint32_t df = rf[e] ? x + e - rf[e] : 0xffffu, dq = rq[e] ? x + e - rq[e] : 0xffffu;
            if (df > 32767u) df = 0xffffu;
            if (dq > 32767u) dq = 0xffffu;
            w[e] = zc_lim_word(t0 + i + e, link[i + e], df, dq, J.P.max_dist, g.corner);
        }
        if (i + 1u < tlen) *(uint2 *)(limw + i) = make_uint2(w[0], w[1]);
        else limw[i] = w[0];
    }
}

/* Service threshold: a wavefront leaves the candidate loop once this many of its lanes have stopped walking. */
#define ZK_MATCH_THRESH 28
__global__ __launch_bounds__(ZK_MATCH_THREADS) void k_match(ZkJob J, uint32_t tiles_per_chunk, int rle)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_lds[ZK_LDS_BYTES / 4];
    __shared__ uint32_t s_next;
    uint32_t *s_win = s_lds;
    uint16_t *s_pd = (uint16_t *)(s_lds + ZK_PD_OFF / 4);
    const uint32_t tid = threadIdx.x;
    const uint32_t lc = blockIdx.x / tiles_per_chunk, tile = blockIdx.x % tiles_per_chunk;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    if ((uint64_t)tile * ZK_TILE >= g.len) return;
    const uint32_t E = g.dict_len + g.len;
    const uint32_t t0 = g.dict_len + tile * ZK_TILE;
    const uint32_t tlen = (E - t0) < ZK_TILE ? (E - t0) : ZK_TILE;
    const int64_t a0 = (int64_t)t0 - (int64_t)ZK_HALO;      /* a-coordinate of LDS byte 0 */
    zk_load_window(J, g, a0, s_win, ZK_WIN_BYTES, tid);
    if (!rle) zk_load_links(J, J.pdist, g, a0, s_pd, ZK_HALO + ZK_TILE, tid);
    if (tid == 0) s_next = 0;
    __syncthreads();
    LdsWin win; win.w = s_win; win.a0 = (uint32_t)a0;
    LdsPd pd; pd.pd = s_pd; pd.a0 = (uint32_t)a0;
    uint2 *rec = J.rec + (g.dstart + t0 - J.rec_base);
    const uint32_t lane = tid & 63u;
    if (rle) {
        const uint32_t la0r = (uint32_t)a0;
        for (;;) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&s_next, 64u);
            base = __shfl(base, 0);
            if (base >= tlen) break;
            uint32_t i = base + lane;
            if (i < tlen) {
                /* zc_rle_pos, 16 bytes of the run per LDS round trip instead of 4 */
                const uint32_t pp = t0 + i, look = E - pp;
                uint32_t r = 0;
                if (look >= ZC_MIN_MATCH && pp != 0) {
                    const uint32_t cap = look < ZC_MAX_MATCH ? look : ZC_MAX_MATCH;
                    const uint32_t pat = (win.load4(pp - 1u) & 0xffu) * 0x01010101u;
                    uint32_t len = 0;
                    while (len < cap) {
                        const uint32_t o = pp + len - la0r, k = o >> 2;
                        const uint32_t w0 = s_win[k], w1 = s_win[k + 1], w2 = s_win[k + 2], w3 = s_win[k + 3], w4 = s_win[k + 4];
                        const uint32_t x0 = __builtin_amdgcn_alignbyte(w1, w0, o) ^ pat, x1 = __builtin_amdgcn_alignbyte(w2, w1, o) ^ pat;
                        const uint32_t x2 = __builtin_amdgcn_alignbyte(w3, w2, o) ^ pat, x3 = __builtin_amdgcn_alignbyte(w4, w3, o) ^ pat;
                        const uint32_t n = x0 ? zc_ctz32(x0) >> 3 : x1 ? 4u + (zc_ctz32(x1) >> 3) : x2 ? 8u + (zc_ctz32(x2) >> 3) : x3 ? 12u + (zc_ctz32(x3) >> 3) : 16u;
                        len += n;
                        if (n < 16u) break;
                    }
                    if (len > cap) len = cap;
                    r = len >= ZC_MIN_MATCH ? (len | (1u << 16)) : 0u;
                }
                rec[i] = make_uint2(r, r);
            }
        }
        return;
    }
    /* Persistent lanes.  A lane walks the chain of one position; a lane whose walk
     * is over draws the next unclaimed position of the tile once enough lanes of
     * the wavefront are idle, so chains of very different length do not leave the
     * wavefront waiting on its slowest lane.  The per-candidate step makes the same
     * decisions as zc_match_peek / zc_match_advance in zamd_core.h but is written
     * as straight-line integer arithmetic on sign masks, because the scalar unit a
     * CU's four SIMDs share is what nested predicates would saturate.  Candidates
     * that pass the 3-byte peek are parked and compared in full, together, at the
     * next service point. */
    const uint32_t nice_match = J.P.nice, max_chain = J.P.chain, qb = max_chain >> 2, max_dist = J.P.max_dist;
    const uint32_t la0 = (uint32_t)a0;
    uint32_t p = t0, cur = t0, best = 2, bdist = 0, qrec = 0, scan3 = 0, cap = 0, nice = 0;
    auto set_scan = [&](uint32_t b) { scan3 = win.load4(p + b - 2u) & 0x00ffffffu; };
    int32_t kk = 0;                           /* candidates examined minus max_chain: negative while budget remains */
    int32_t walking = 0, parked = 0;          /* sign masks: -1 / 0 */
    uint32_t wcur = 0, wend = 0;              /* this wave's ticket of unclaimed tile positions */
    bool exhausted = false;
    for (;;) {
        /* ---- full comparisons of parked candidates (zc_match_extend, then zc_match_advance) */
        if (parked < 0) {
            /* the first 16 bytes of both strings (most comparisons end there; the compiler reads them a pair of dwords at a
             * time and stops at the first difference, which costs short comparisons fewer LDS reads than all ten at once
             * and measured the same), then 32 bytes a round trip */
            uint32_t len, dnext;
            {
                const uint32_t oc = cur - la0, op = p - la0, ic = oc >> 2, ip = op >> 2;
                dnext = pd(cur);                /* the link, fetched in the same round trip */
                const uint32_t c0 = s_win[ic], c1 = s_win[ic + 1], c2 = s_win[ic + 2], c3 = s_win[ic + 3], c4 = s_win[ic + 4];
                const uint32_t q0 = s_win[ip], q1 = s_win[ip + 1], q2 = s_win[ip + 2], q3 = s_win[ip + 3], q4 = s_win[ip + 4];
                const uint32_t x0 = __builtin_amdgcn_alignbyte(c1, c0, oc) ^ __builtin_amdgcn_alignbyte(q1, q0, op);
                const uint32_t x1 = __builtin_amdgcn_alignbyte(c2, c1, oc) ^ __builtin_amdgcn_alignbyte(q2, q1, op);
                const uint32_t x2 = __builtin_amdgcn_alignbyte(c3, c2, oc) ^ __builtin_amdgcn_alignbyte(q3, q2, op);
                const uint32_t x3 = __builtin_amdgcn_alignbyte(c4, c3, oc) ^ __builtin_amdgcn_alignbyte(q4, q3, op);
                len = x0 ? zc_ctz32(x0) >> 3 : x1 ? 4u + (zc_ctz32(x1) >> 3) : x2 ? 8u + (zc_ctz32(x2) >> 3) : x3 ? 12u + (zc_ctz32(x3) >> 3) : 16u;
            }
            if (len == 16u) {
                /* long matches (runs, repeated records) 16 bytes a round trip as well: a wavefront stays in this loop as
                 * long as its longest comparison does */
                while (len < cap) {
                    const uint32_t oc = cur + len - la0, op = p + len - la0, ic = oc >> 2, ip = op >> 2;
                    uint32_t c[9], q[9], x[8];
#pragma unroll
                    for (int k = 0; k < 9; k++) { c[k] = s_win[ic + k]; q[k] = s_win[ip + k]; }
#pragma unroll
                    for (int k = 0; k < 8; k++) x[k] = __builtin_amdgcn_alignbyte(c[k + 1], c[k], oc) ^ __builtin_amdgcn_alignbyte(q[k + 1], q[k], op);
                    uint32_t n = 32u;
#pragma unroll
                    for (int k = 7; k >= 0; k--) n = x[k] ? 4u * (uint32_t)k + (zc_ctz32(x[k]) >> 3) : n;
                    len += n;
                    if (n < 32u) break;
                }
            }
            if (len > cap) len = cap;
            bool fin = false;
            /* the step that parked this walk already counted the candidate */
            const uint32_t k = (uint32_t)(kk + (int32_t)max_chain);
            if (len > best) {
                best = len; bdist = p - cur;
                if (k <= qb) qrec = best | (bdist << 16);   /* best of the first max_chain/4 candidates */
                if (len >= nice) fin = true;
                else set_scan(len);
            }
            if (!fin) {
                const uint32_t d = dnext;
                if (kk >= 0 || d == 0 || d >= cur) fin = true;
                else { cur -= d; fin = p - cur >= max_dist; }
            }
            parked = 0; walking = -1;
            if (fin) {
                uint32_t full = best >= 3u ? (best | (bdist << 16)) : 0u;
                uint32_t rq = qb == 0 ? 0u : (k > qb ? qrec : full);
                rec[p - t0] = make_uint2(full, rq);
                walking = 0;
            }
        }
        /* ---- refill idle lanes from the wave's ticket (64 positions per LDS atomic) */
        for (int pass = 0; pass < 2; pass++) {
            unsigned long long idle = __ballot(walking == 0);
            if (idle == 0ull) break;
            if (wcur == wend) {
                if (exhausted) break;
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&s_next, 64u);
                base = __shfl(base, 0);
                wcur = base; wend = base + 64u < tlen ? base + 64u : tlen;
                if (base + 64u >= tlen) exhausted = true;
                if (base >= tlen) { wcur = wend = 0; break; }
            }
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const uint32_t take = (uint32_t)__popcll(idle) < wend - wcur ? (uint32_t)__popcll(idle) : wend - wcur;
            if (walking == 0 && rank < take) {
                const uint32_t idx = wcur + rank;
                ZcMatchState st;
                if (zc_match_init(win, pd, t0 + idx, E, g.corner, nice_match, st, max_dist)) {
                    p = st.p; cur = st.cur; kk = -(int32_t)max_chain; best = 2; bdist = 0; qrec = 0;
                    scan3 = st.scan3; cap = st.cap; nice = st.nice;
                    walking = -1;
                } else rec[idx] = make_uint2(0u, 0u);
            }
            wcur += take;
        }
        if (__ballot(walking < 0) == 0ull) { if (exhausted && wcur == wend) break; continue; }
        /* ---- candidate steps until a quarter of the lanes need service */
        const uint32_t full_now = best >= 3u ? (best | (bdist << 16)) : 0u;   /* best is fixed while walking */
        const uint32_t pout = p - t0;
        const int32_t lim = (int32_t)(p < max_dist ? p : max_dist);      /* stop once the distance reaches this */
        /* the loop runs on ac = candidate + (best - 2) - la0, the LDS byte offset of the candidate's bytes best-2..,
         * so that neither the peek address nor the link address nor the distance test needs an addition of its own */
        const uint32_t peek_off = (best - 2u) - la0;
        const uint16_t *link = s_pd - (best - 2u);   /* link[ac] is the link of the candidate */
        uint32_t c = cur + peek_off;
        const int32_t plim = (int32_t)p - lim + (int32_t)peek_off;   /* the next candidate c' is in reach while plim - c' < 0 */
        /* a walk that has stopped is masked off for the rest of the round: c, its count and the last peek result stay as
         * the stopping step left them; lanes that are not walking get a count that says "budget spent" */
        int32_t wm = walking;
        int32_t kc = kk & walking;
        uint32_t xlast = 1u;
#ifdef ZK_MATCH_ASM
        /* The same eight-step round by hand: a lane that stops is taken out of EXEC by the comparison that stops it
         * (its candidate, count and last peek stay as that step left them) and EXEC is restored after the round, so a
         * step needs no mask register, no multiply and no scalar instruction: 10 vector + 2 LDS instructions.  The
         * reach test runs on r = plim - c, which the step's own sum carries to the next step. */
        {
            const uint32_t wbase = (uint32_t)(uintptr_t)s_win;
            uint32_t ca = c + wbase;
            const uint32_t l0 = (uint32_t)(uintptr_t)s_pd - 2u * (best - 2u) - 2u * wbase;
            int32_t r = plim - (int32_t)c;
            asm volatile(
            "s_mov_b64 s[90:91], exec\n\t"
            "v_cmp_gt_i32_e32 vcc, 0, %[wm]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_mov_b32 s92, 8\n\t"
            "1:\n\t"
            "v_and_b32 v50, -4, %[c]\n\t"
            "ds_read2_b32 v[52:53], v50 offset1:1\n\t"
            "v_lshl_add_u32 v51, %[c], 1, %[l0]\n\t"
            "ds_read_u16 v54, v51\n\t"
            "v_add_u32 %[kc], 1, %[kc]\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_alignbyte_b32 v52, v53, v52, %[c]\n\t"
            "v_bitop3_b32 %[x], v52, %[scan], %[mask] bitop3:0x6c\n\t"
            "v_sub_u32 v55, 0, %[x]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_add_u32 v57, %[r], v54\n\t"
            "v_bitop3_b32 v56, v57, %[kc], v55 bitop3:0x80\n\t"
            "v_cmpx_gt_i32_e32 vcc, 0, v56\n\t"
            "v_sub_u32 %[c], %[c], v54\n\t"
            "v_and_b32 v50, -4, %[c]\n\t"
            "ds_read2_b32 v[52:53], v50 offset1:1\n\t"
            "v_lshl_add_u32 v51, %[c], 1, %[l0]\n\t"
            "ds_read_u16 v54, v51\n\t"
            "v_add_u32 %[kc], 1, %[kc]\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_alignbyte_b32 v52, v53, v52, %[c]\n\t"
            "v_bitop3_b32 %[x], v52, %[scan], %[mask] bitop3:0x6c\n\t"
            "v_sub_u32 v55, 0, %[x]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_add_u32 %[r], v57, v54\n\t"
            "v_bitop3_b32 v56, %[r], %[kc], v55 bitop3:0x80\n\t"
            "v_cmpx_gt_i32_e32 vcc, 0, v56\n\t"
            "v_sub_u32 %[c], %[c], v54\n\t"
            "v_and_b32 v50, -4, %[c]\n\t"
            "ds_read2_b32 v[52:53], v50 offset1:1\n\t"
            "v_lshl_add_u32 v51, %[c], 1, %[l0]\n\t"
            "ds_read_u16 v54, v51\n\t"
            "v_add_u32 %[kc], 1, %[kc]\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_alignbyte_b32 v52, v53, v52, %[c]\n\t"
            "v_bitop3_b32 %[x], v52, %[scan], %[mask] bitop3:0x6c\n\t"
            "v_sub_u32 v55, 0, %[x]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_add_u32 v57, %[r], v54\n\t"
            "v_bitop3_b32 v56, v57, %[kc], v55 bitop3:0x80\n\t"
            "v_cmpx_gt_i32_e32 vcc, 0, v56\n\t"
            "v_sub_u32 %[c], %[c], v54\n\t"
            "v_and_b32 v50, -4, %[c]\n\t"
            "ds_read2_b32 v[52:53], v50 offset1:1\n\t"
            "v_lshl_add_u32 v51, %[c], 1, %[l0]\n\t"
            "ds_read_u16 v54, v51\n\t"
            "v_add_u32 %[kc], 1, %[kc]\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_alignbyte_b32 v52, v53, v52, %[c]\n\t"
            "v_bitop3_b32 %[x], v52, %[scan], %[mask] bitop3:0x6c\n\t"
            "v_sub_u32 v55, 0, %[x]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_add_u32 %[r], v57, v54\n\t"
            "v_bitop3_b32 v56, %[r], %[kc], v55 bitop3:0x80\n\t"
            "v_cmpx_gt_i32_e32 vcc, 0, v56\n\t"
            "v_sub_u32 %[c], %[c], v54\n\t"
            "v_and_b32 v50, -4, %[c]\n\t"
            "ds_read2_b32 v[52:53], v50 offset1:1\n\t"
            "v_lshl_add_u32 v51, %[c], 1, %[l0]\n\t"
            "ds_read_u16 v54, v51\n\t"
            "v_add_u32 %[kc], 1, %[kc]\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_alignbyte_b32 v52, v53, v52, %[c]\n\t"
            "v_bitop3_b32 %[x], v52, %[scan], %[mask] bitop3:0x6c\n\t"
            "v_sub_u32 v55, 0, %[x]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_add_u32 v57, %[r], v54\n\t"
            "v_bitop3_b32 v56, v57, %[kc], v55 bitop3:0x80\n\t"
            "v_cmpx_gt_i32_e32 vcc, 0, v56\n\t"
            "v_sub_u32 %[c], %[c], v54\n\t"
            "v_and_b32 v50, -4, %[c]\n\t"
            "ds_read2_b32 v[52:53], v50 offset1:1\n\t"
            "v_lshl_add_u32 v51, %[c], 1, %[l0]\n\t"
            "ds_read_u16 v54, v51\n\t"
            "v_add_u32 %[kc], 1, %[kc]\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_alignbyte_b32 v52, v53, v52, %[c]\n\t"
            "v_bitop3_b32 %[x], v52, %[scan], %[mask] bitop3:0x6c\n\t"
            "v_sub_u32 v55, 0, %[x]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_add_u32 %[r], v57, v54\n\t"
            "v_bitop3_b32 v56, %[r], %[kc], v55 bitop3:0x80\n\t"
            "v_cmpx_gt_i32_e32 vcc, 0, v56\n\t"
            "v_sub_u32 %[c], %[c], v54\n\t"
            "v_and_b32 v50, -4, %[c]\n\t"
            "ds_read2_b32 v[52:53], v50 offset1:1\n\t"
            "v_lshl_add_u32 v51, %[c], 1, %[l0]\n\t"
            "ds_read_u16 v54, v51\n\t"
            "v_add_u32 %[kc], 1, %[kc]\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_alignbyte_b32 v52, v53, v52, %[c]\n\t"
            "v_bitop3_b32 %[x], v52, %[scan], %[mask] bitop3:0x6c\n\t"
            "v_sub_u32 v55, 0, %[x]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_add_u32 v57, %[r], v54\n\t"
            "v_bitop3_b32 v56, v57, %[kc], v55 bitop3:0x80\n\t"
            "v_cmpx_gt_i32_e32 vcc, 0, v56\n\t"
            "v_sub_u32 %[c], %[c], v54\n\t"
            "v_and_b32 v50, -4, %[c]\n\t"
            "ds_read2_b32 v[52:53], v50 offset1:1\n\t"
            "v_lshl_add_u32 v51, %[c], 1, %[l0]\n\t"
            "ds_read_u16 v54, v51\n\t"
            "v_add_u32 %[kc], 1, %[kc]\n\t"
            "s_waitcnt lgkmcnt(1)\n\t"
            "v_alignbyte_b32 v52, v53, v52, %[c]\n\t"
            "v_bitop3_b32 %[x], v52, %[scan], %[mask] bitop3:0x6c\n\t"
            "v_sub_u32 v55, 0, %[x]\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "v_add_u32 %[r], v57, v54\n\t"
            "v_bitop3_b32 v56, %[r], %[kc], v55 bitop3:0x80\n\t"
            "v_cmpx_gt_i32_e32 vcc, 0, v56\n\t"
            "v_sub_u32 %[c], %[c], v54\n\t"
            "s_bcnt1_i32_b64 s93, exec\n\t"
            "s_cmp_le_u32 s93, %[keep]\n\t"
            "s_cbranch_scc1 2f\n\t"
            "s_add_i32 s92, s92, -1\n\t"
            "s_cmp_lg_u32 s92, 0\n\t"
            "s_cbranch_scc1 1b\n\t"
            "2:\n\t"
            "s_mov_b64 s[94:95], exec\n\t"
            "s_mov_b64 exec, s[90:91]\n\t"
            "v_mov_b32 %[wm], 0\n\t"
            "s_mov_b64 exec, s[94:95]\n\t"
            "v_mov_b32 %[wm], -1\n\t"
            "s_mov_b64 exec, s[90:91]\n\t"
            : [c] "+v"(ca), [kc] "+v"(kc), [x] "+v"(xlast), [r] "+v"(r), [wm] "+v"(wm)
            : [l0] "v"(l0), [scan] "v"(scan3), [mask] "s"(0x00ffffffu), [keep] "n"(64 - ZK_MATCH_THRESH)
            : "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "s90", "s91", "s92", "s93", "s94", "s95", "vcc", "scc");
            c = ca - wbase;
        }
#else
#pragma unroll 1
        for (int it = 0; it < 64 / ZK_MATCH_STEPS; it++) {
#pragma unroll
            for (int u = 0; u < ZK_MATCH_STEPS; u++) {
                if (wm >= 0) continue;                            /* stopped lanes are masked off, their LDS reads with them */
                const uint32_t wi = c >> 2;
                const uint32_t v = __builtin_amdgcn_alignbyte(s_win[wi + 1], s_win[wi], c);   /* the instruction uses bits 1:0 of the shift operand; one
                                                                                               * ds_read_b32 at the byte address itself is legal on gfx950 and
                                                                                               * measured 2.2 times slower for the whole kernel */
                const int32_t d = (int32_t)link[c];
                const uint32_t x = (v & 0x00ffffffu) ^ scan3;
                const int32_t nohit = -(int32_t)x;               /* negative unless the peek matched */
                xlast = x;
                kc -= wm;                                         /* count the candidate */
                const int32_t go = ((plim - (int32_t)c) + d) & kc & nohit;   /* negative iff the walk continues */
                wm = go >> 31;
                c += (uint32_t)__mul24(d, wm);                    /* one multiply-add: c -= d where the walk continues */
                asm("" : "+v"(c));                              /* keeps the compiler from deriving the next link address by a multiply of its own */
            }
            if (__popcll(__ballot(wm == 0)) >= ZK_MATCH_THRESH) break;
        }
#endif
        if (walking < 0) kk = kc;
        cur = c - peek_off;
        /* a walk that stopped in this round stands on the candidate that stopped it: it is parked if that
         * candidate passed the peek */
        parked = xlast == 0u ? (walking & ~wm) : 0;
        /* a walk that ended without a parked candidate leaves its counters frozen (a lane that is not
         * walking counts nothing): its record is written here, outside the hot loop */
        if ((walking & ~wm & ~parked) < 0) {
            uint32_t k = (uint32_t)(kk + (int32_t)max_chain);
            uint32_t rq = qb == 0 ? 0u : (k > qb ? qrec : full_now);
            rec[pout] = make_uint2(full_now, rq);
        }
        walking = wm | parked;                /* parked walks are still owned by the lane */
    }
}

/* ------------------------------------------------------------------ k_match4
 * The search of k_match along the four-byte chains, for large chain budgets (levels 8 and 9): see "the same search
 * along four-byte chains" in zamd_core.h.  Same tile, same persistent lanes, same records. */
__global__ __launch_bounds__(ZK_MATCH_THREADS) void k_match4(ZkJob J, uint32_t tiles_per_chunk)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_lds[ZK_LDS_BYTES / 4];
    __shared__ __attribute__((aligned(16))) uint2 s_tick[ZK_MATCH_THREADS];       /* per wavefront: the two words of the 64 positions of its ticket */
    __shared__ uint32_t s_next;
    uint32_t *s_win = s_lds;
    uint16_t *s_pd = (uint16_t *)(s_lds + ZK_PD_OFF / 4);
    const uint32_t tid = threadIdx.x;
    const uint32_t lc = blockIdx.x / tiles_per_chunk, tile = blockIdx.x % tiles_per_chunk;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    if ((uint64_t)tile * ZK_TILE >= g.len) return;
    const uint32_t E = g.dict_len + g.len;
    const uint32_t t0 = g.dict_len + tile * ZK_TILE;
    const uint32_t tlen = (E - t0) < ZK_TILE ? (E - t0) : ZK_TILE;
    const int64_t a0 = (int64_t)t0 - (int64_t)ZK_HALO;      /* a-coordinate of LDS byte 0 */
    zk_load_window(J, g, a0, s_win, ZK_WIN_BYTES, tid);
    zk_load_links(J, J.pdist, g, a0, s_pd, ZK_HALO + ZK_TILE, tid);
    if (tid == 0) s_next = 0;
    __syncthreads();
    LdsWin win; win.w = s_win; win.a0 = (uint32_t)a0;
    uint2 *rec = J.rec + (g.dstart + t0 - J.rec_base);
    const uint32_t lane = tid & 63u;
    /* Persistent lanes.  A lane walks the four-byte chain of one position (zc_match_pos4 in zamd_core.h says what the
     * walk computes); a lane whose walk is over draws the next unclaimed position of the tile once enough lanes of
     * the wavefront are idle, so chains of very different length do not leave the wavefront waiting on its slowest
     * lane.  The per-candidate step is written as straight-line integer arithmetic, because the scalar unit a CU's
     * four SIMDs share is what nested predicates would saturate.  Candidates that pass the 3-byte peek are parked and
     * compared in full, together, at the next service point.
     * Positions are handed out in tickets of 64 to a wavefront.  The two words k_reach and k_first3 left for a ticket's positions are
     * fetched when the ticket is drawn, one ticket ahead of its use, and wait in registers; when the wavefront starts
     * on the ticket they go to the wavefront's slice of s_tick, where the lane that takes a position finds them. */
    const uint32_t nice_match = J.P.nice, max_dist = J.P.max_dist;
    const uint32_t la0 = (uint32_t)a0;
    const uint32_t *const limw = J.hop + (uint64_t)lc * J.P.chunk_size + tile * ZK_TILE;
    const uint32_t *const f3w = J.sym + (uint64_t)lc * J.P.chunk_size + tile * ZK_TILE;
    uint2 *const tick = s_tick + (tid - lane);
    uint32_t p = t0, cur = t0, best = ZC_MIN_MATCH, full = 0, qrec = 0, scan3 = 0, cap = 0, nice = 0, wf = 0, wq = 0;
    auto set_scan = [&](uint32_t b) { scan3 = win.load4(p + b - 2u) & 0x00ffffffu; };
    int32_t walking = 0, parked = 0;          /* sign masks: -1 / 0 */
    uint32_t wcur = 0, wend = 0;              /* the ticket in use: unclaimed positions [wcur, wend) */
    uint32_t tbase = 0;                       /* first position of the ticket in use */
    uint32_t nbase;                           /* first position of the ticket drawn ahead (>= tlen: there is none) */
    uint2 nlim = make_uint2(0, 0);            /* this lane's share of its two words */
    auto draw = [&]() {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&s_next, 64u);
        nbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        const uint32_t at = nbase + lane < tlen ? nbase + lane : tlen - 1u;  /* no branch: nothing waits for the loads here */
        nlim.x = limw[at]; nlim.y = f3w[at];
    };
    draw();
    for (;;) {
        /* ---- full comparisons of parked candidates */
        if (parked < 0) {
            /* the first 16 bytes of both strings (most comparisons end there), then 32 bytes a round trip */
            uint32_t len, dnext;
            {
                const uint32_t oc = cur - la0, op = p - la0, ic = oc >> 2, ip = op >> 2;
                dnext = s_pd[oc];               /* the link, fetched in the same round trip; "none" is 0xffff */
                const uint32_t c0 = s_win[ic], c1 = s_win[ic + 1], c2 = s_win[ic + 2], c3 = s_win[ic + 3], c4 = s_win[ic + 4];
                const uint32_t q0 = s_win[ip], q1 = s_win[ip + 1], q2 = s_win[ip + 2], q3 = s_win[ip + 3], q4 = s_win[ip + 4];
                const uint32_t x0 = __builtin_amdgcn_alignbyte(c1, c0, oc) ^ __builtin_amdgcn_alignbyte(q1, q0, op);
                const uint32_t x1 = __builtin_amdgcn_alignbyte(c2, c1, oc) ^ __builtin_amdgcn_alignbyte(q2, q1, op);
                const uint32_t x2 = __builtin_amdgcn_alignbyte(c3, c2, oc) ^ __builtin_amdgcn_alignbyte(q3, q2, op);
                const uint32_t x3 = __builtin_amdgcn_alignbyte(c4, c3, oc) ^ __builtin_amdgcn_alignbyte(q4, q3, op);
                len = x0 ? zc_ctz32(x0) >> 3 : x1 ? 4u + (zc_ctz32(x1) >> 3) : x2 ? 8u + (zc_ctz32(x2) >> 3) : x3 ? 12u + (zc_ctz32(x3) >> 3) : 16u;
            }
            if (len == 16u) {
                /* long matches (runs, repeated records): a wavefront stays in this loop as long as its longest comparison does */
                while (len < cap) {
                    const uint32_t oc = cur + len - la0, op = p + len - la0, ic = oc >> 2, ip = op >> 2;
                    uint32_t c[9], q[9], x[8];
#pragma unroll
                    for (int k = 0; k < 9; k++) { c[k] = s_win[ic + k]; q[k] = s_win[ip + k]; }
#pragma unroll
                    for (int k = 0; k < 8; k++) x[k] = __builtin_amdgcn_alignbyte(c[k + 1], c[k], oc) ^ __builtin_amdgcn_alignbyte(q[k + 1], q[k], op);
                    uint32_t n = 32u;
#pragma unroll
                    for (int k = 7; k >= 0; k--) n = x[k] ? 4u * (uint32_t)k + (zc_ctz32(x[k]) >> 3) : n;
                    len += n;
                    if (n < 32u) break;
                }
            }
            if (len > cap) len = cap;
            bool fin = false;
            const uint32_t bd = p - cur;
            if (len > best) {
                best = len; full = len | (bd << 16);
                if (bd <= wq) qrec = full;      /* within the quarter budget as well */
                if (len >= nice) fin = true;    /* deflate.c:1276 */
                else set_scan(len);
            }
            if (!fin) {
                if (bd + dnext > wf) fin = true;          /* the next member is out of reach (budget, MAX_DIST, NIL) or there is none */
                else cur -= dnext;
            }
            parked = 0; walking = -1;
            if (fin) { rec[p - t0] = make_uint2(full, qrec); walking = 0; }
        }
        /* ---- refill idle lanes from the wavefront's ticket */
        for (int pass = 0; pass < 2; pass++) {
            unsigned long long idle = __ballot(walking == 0);
            if (idle == 0ull) break;
            if (wcur == wend) {
                if (nbase >= tlen) break;                 /* the tile's positions are all claimed */
                /* the ticket drawn ahead is the ticket in use from here on: its words go where the lanes find them */
                tbase = nbase; wcur = nbase; wend = nbase + 64u < tlen ? nbase + 64u : tlen;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                tick[lane] = nlim;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                draw();
            }
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            const uint32_t take = (uint32_t)__popcll(idle) < wend - wcur ? (uint32_t)__popcll(idle) : wend - wcur;
            if (walking == 0 && rank < take) {
                const uint32_t idx = wcur + rank, pp = t0 + idx, look = E - pp;
                const uint2 L = tick[idx - tbase];
                const uint32_t d3 = L.y & 0x7fffu;
                uint32_t f = 0, q = 0, d4 = 0, b0 = ZC_MIN_MATCH, nwf = L.x & 0x7fffu, nwq = L.x >> 16;
                bool walk = false;
                if (look >= ZC_MIN_MATCH && d3 != 0u) {
                    if (L.x & 0x8000u) {
                        /* the hash-chain head at exactly MAX_DIST is the only candidate there is: it is measured in full if its
                         * three bytes match (the first step's peek then parks it), and nothing lies within reach behind it */
                        if (d3 == max_dist) { walk = true; d4 = max_dist; b0 = ZC_MIN_MATCH - 1u; nwf = 0; nwq = 0xffffu; }
                    } else if (d3 <= nwf) {
                        f = ZC_MIN_MATCH | (d3 << 16); q = d3 <= nwq ? f : 0u;
                        d4 = s_pd[pp - la0];
                        walk = look > ZC_MIN_MATCH && d4 <= nwf;
                    }
                }
                if (walk) {
                    p = pp; cur = pp - d4; best = b0; full = f; qrec = q; wf = nwf; wq = nwq;
                    cap = look < ZC_MAX_MATCH ? look : ZC_MAX_MATCH; nice = nice_match < look ? nice_match : look;
                    set_scan(b0);
                    walking = -1;
                } else rec[idx] = make_uint2(f, q);
            }
            wcur += take;
        }
        if (__ballot(walking < 0) == 0ull) { if (wcur == wend && nbase >= tlen) break; continue; }
        /* ---- candidate steps until a quarter of the lanes need service */
        const uint32_t pout = p - t0;
        /* the loop runs on c = candidate + (best - 2) - la0, the LDS byte offset of the candidate's bytes best-2..,
         * so that neither the peek address nor the link address nor the reach test needs an addition of its own */
        const uint32_t peek_off = (best - 2u) - la0;
        uint32_t c = cur + peek_off;
        const int32_t plim = (int32_t)(p - wf - 1u + peek_off);      /* the next candidate c' is in reach while plim - c' < 0 */
        /* a walk that has stopped is masked off for the rest of the round: c and the last peek result stay as the
         * stopping step left them */
        int32_t wm = walking;
        uint32_t xlast = 1u;
#ifdef ZK_MATCH_ASM
        /* The round by hand: a lane that stops is taken out of EXEC by the comparison that stops it — the peek that
         * matched (the lane stays on that candidate), or the reach test on the next member — and EXEC is restored
         * after the round, so a step needs no mask register and no scalar instruction: 8 vector + 2 LDS instructions.
         * The reach test runs on r = plim - c, which the step's own sum carries to the next step.  (v50-v54 and
         * s90-s95 are named outright because ds_read2_b32 wants a register PAIR whose halves the next instruction
         * names one by one; the clobber list keeps the compiler off them.) */
        {
            const uint32_t wbase = (uint32_t)(uintptr_t)s_win;
            uint32_t ca = c + wbase;
            const uint32_t l0 = (uint32_t)(uintptr_t)s_pd - 2u * (best - 2u) - 2u * wbase;
            int32_t r = plim - (int32_t)c;
#define ZK_STEP \
            "v_and_b32 v50, -4, %[c]\n\t" \
            "ds_read2_b32 v[52:53], v50 offset1:1\n\t" \
            "v_lshl_add_u32 v51, %[c], 1, %[l0]\n\t" \
            "ds_read_u16 v54, v51\n\t" \
            "s_waitcnt lgkmcnt(1)\n\t" \
            "v_alignbyte_b32 v52, v53, v52, %[c]\n\t" \
            "v_bitop3_b32 %[x], v52, %[scan], %[mask] bitop3:0x6c\n\t" \
            "v_cmpx_ne_u32_e32 vcc, 0, %[x]\n\t" \
            "s_waitcnt lgkmcnt(0)\n\t" \
            "v_add_u32 %[r], %[r], v54\n\t" \
            "v_cmpx_gt_i32_e32 vcc, 0, %[r]\n\t" \
            "v_sub_u32 %[c], %[c], v54\n\t"
            asm volatile(
            "s_mov_b64 s[90:91], exec\n\t"
            "v_cmp_gt_i32_e32 vcc, 0, %[wm]\n\t"
            "s_and_b64 exec, exec, vcc\n\t"
            "s_mov_b32 s92, %[rounds]\n\t"
            "1:\n\t"
            ZK_STEP ZK_STEP ZK_STEP ZK_STEP
            "s_bcnt1_i32_b64 s93, exec\n\t"
            "s_cmp_le_u32 s93, %[keep]\n\t"
            "s_cbranch_scc1 2f\n\t"
            "s_add_i32 s92, s92, -1\n\t"
            "s_cmp_lg_u32 s92, 0\n\t"
            "s_cbranch_scc1 1b\n\t"
            "2:\n\t"
            "s_mov_b64 s[94:95], exec\n\t"
            "s_mov_b64 exec, s[90:91]\n\t"
            "v_mov_b32 %[wm], 0\n\t"
            "s_mov_b64 exec, s[94:95]\n\t"
            "v_mov_b32 %[wm], -1\n\t"
            "s_mov_b64 exec, s[90:91]\n\t"
            : [c] "+v"(ca), [x] "+v"(xlast), [r] "+v"(r), [wm] "+v"(wm)
            : [l0] "v"(l0), [scan] "v"(scan3), [mask] "s"(0x00ffffffu), [keep] "n"(64 - ZK_MATCH_THRESH), [rounds] "n"(64 / 4)
            : "v50", "v51", "v52", "v53", "v54", "s90", "s91", "s92", "s93", "s94", "s95", "vcc", "scc");
#undef ZK_STEP
            c = ca - wbase;
        }
#else
        {
            const uint16_t *link = s_pd - (best - 2u);   /* link[c] is the link of the candidate */
#pragma unroll 1
            for (int it = 0; it < 64 / ZK_MATCH_STEPS; it++) {
#pragma unroll
                for (int u = 0; u < ZK_MATCH_STEPS; u++) {
                    if (wm >= 0) continue;                            /* stopped lanes are masked off, their LDS reads with them */
                    const uint32_t wi = c >> 2;
                    const uint32_t v = __builtin_amdgcn_alignbyte(s_win[wi + 1], s_win[wi], c);
                    const int32_t d = (int32_t)link[c];
                    const uint32_t x = (v & 0x00ffffffu) ^ scan3;
                    xlast = x;
                    if (x == 0u || (plim - (int32_t)c) + d >= 0) wm = 0;      /* parked on this candidate, or the next one is out of reach */
                    else c -= (uint32_t)d;
                }
                if (__popcll(__ballot(wm == 0)) >= ZK_MATCH_THRESH) break;
            }
        }
#endif
        cur = c - peek_off;
        /* a walk that stopped in this round stands on the candidate that stopped it: it is parked if that
         * candidate passed the peek */
        parked = xlast == 0u ? (walking & ~wm) : 0;
        /* a walk that ended without a parked candidate: its record is written here, outside the hot loop */
        if ((walking & ~wm & ~parked) < 0) rec[pout] = make_uint2(full, qrec);
        walking = wm | parked;                /* parked walks are still owned by the lane */
    }
}

#include "zamd_follow_impl.h"

/* ------------------------------------------------------------------ k_parse
 * One wavefront per chunk.  The lazy-evaluation state machine is serial, so all
 * 64 lanes run it redundantly (every value is wave-uniform, branches are scalar)
 * and use their width only to stream the match records through a small LDS tile
 * with coalesced loads; lane 0 writes symbols and blocks. */
#define ZK_PARSE_TILE 1024u
struct TileRec {
    const uint2 *g;               /* records of this chunk, index = a - dict_len */
    uint2 *tile;                  /* LDS, ZK_PARSE_TILE entries */
    uint32_t dict_len, len, base, lane;
    __device__ void operator()(uint32_t p, uint32_t *f, uint32_t *q)
    {
        uint32_t i = p - dict_len;
        if (i - base >= ZK_PARSE_TILE) {
            base = i & ~(ZK_PARSE_TILE - 1u);
            __syncthreads();
            for (uint32_t k = lane * 2u; k < ZK_PARSE_TILE; k += 128u) {
                uint32_t idx = base + k;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (idx + 2u <= len) v = *(const uint4 *)(g + idx);
                else if (idx < len) { uint2 t = g[idx]; v.x = t.x; v.y = t.y; }
                *(uint4 *)(tile + k) = v;
            }
            __syncthreads();
        }
        uint2 v = tile[i - base];
        *f = v.x; *q = v.y;
    }
};
struct GSym { uint32_t *s; uint32_t lane; __device__ void operator()(uint32_t i, uint32_t w) const { if (lane == 0) s[i] = w; } };
struct GBlk {
    ZcBlock *b; uint32_t cap; uint32_t *err; uint32_t lane;
    __device__ void operator()(uint32_t i, const ZcBlock &v) const
    {
        if (lane != 0) return;
        if (i < cap) b[i] = v; else atomicOr(err, (uint32_t)ZE_BLOCK_OVERFLOW);
    }
};

__global__ __launch_bounds__(64) void k_parse(ZkJob J)
{
    __shared__ uint2 s_tile[ZK_PARSE_TILE];
    const uint32_t lc = blockIdx.x, lane = threadIdx.x;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    TileRec rec; rec.g = J.rec + (g.dstart + g.dict_len - J.rec_base); rec.tile = s_tile;
    rec.dict_len = g.dict_len; rec.len = g.len; rec.base = 0xffff0000u; rec.lane = lane;
    GSym sym; sym.s = J.sym + (uint64_t)lc * J.P.chunk_size; sym.lane = lane;
    GBlk blk; blk.b = J.blk + (uint64_t)lc * J.P.max_blocks; blk.cap = J.P.max_blocks; blk.err = J.err; blk.lane = lane;
    uint32_t nsym = 0, nb;
    if (J.P.parser == ZC_PARSE_HUFF) nb = zc_parse_huff(g.dict_len, g.len, g.final, J.P, blk);
    else if (J.P.parser == ZC_PARSE_STORED) nb = zc_parse_stored(g.dict_len, g.len, g.final, J.P, blk);
    else if (J.P.parser == ZC_PARSE_RLE) nb = zc_parse_rle(g.dict_len, g.len, g.final, J.P, rec, sym, blk, &nsym);
    else nb = zc_parse_slow(g.dict_len, g.len, g.final, J.P, rec, sym, blk, &nsym);
    if (lane == 0) {
        J.nblk[lc] = nb < J.P.max_blocks ? nb : J.P.max_blocks;
        J.nsym[lc] = nsym;
    }
}

/* ------------------------------------------------------------------ k_resolve / k_hop / k_expand
 * The lazy parser (deflate_slow) split into a parallel part, a short serial
 * part and a parallel part; see zc_resolve_pos in zamd_core.h. */
#define ZK_RESOLVE_TILE 256u
struct GRec2 {
    const uint2 *r; uint32_t dict_len;
    __device__ void operator()(uint32_t p, uint32_t *f, uint32_t *q) const { uint2 v = r[p - dict_len]; *f = v.x; *q = v.y; }
};

/* The records of three positions in a row in registers, the others' from memory */
struct GRec2Three {
    const uint2 *r; uint32_t dict_len, p0; uint2 a, b, c;
    __device__ void operator()(uint32_t p, uint32_t *f, uint32_t *q) const
    {
        const uint32_t d = p - p0;
        if (d == 0) { *f = a.x; *q = a.y; }
        else if (d == 1) { *f = b.x; *q = b.y; }
        else if (d == 2) { *f = c.x; *q = c.y; }
        else { const uint2 v = r[p - dict_len]; *f = v.x; *q = v.y; }
    }
};

/* A workgroup takes 8 192 positions of its chunk in a row: the chunk's geometry (64-bit arithmetic, the same for every
 * thread) costs more than one position's hop word, so it is worked out once for all of them.  A thread takes two
 * positions next to each other: one 16-byte load brings both records, the first position's lazy look finds the second's
 * record in a register and the second's finds the next thread's first (a DPP shift; lane 63 loads it), so that the
 * look most chains end with makes no trip to memory.  The hop word of a position where no match starts is the distance
 * to the next one where one does among the 128 positions of its wavefront, or to the end of those: the wavefronts share
 * nothing and never wait for each other. */
#define ZK_RESOLVE_SPAN 32u
__global__ __launch_bounds__(ZK_RESOLVE_TILE) void k_resolve(ZkJob J, uint32_t spans_per_chunk)
{
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t lc = blockIdx.x / spans_per_chunk, span = blockIdx.x % spans_per_chunk;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    const uint32_t E = g.dict_len + g.len;
    GRec2Three rec; rec.r = J.rec + (g.dstart + g.dict_len - J.rec_base); rec.dict_len = g.dict_len;
    uint32_t *const hop = J.hop + (uint64_t)lc * J.P.chunk_size;
    for (uint32_t t = 0; t < ZK_RESOLVE_SPAN / 2u; t++) {
        const uint32_t wi = (span * (ZK_RESOLVE_SPAN / 2u) + t) * (2u * ZK_RESOLVE_TILE) + 2u * (tid - lane);   /* the wavefront's first position */
        if (wi >= g.len) return;
        const uint32_t i0 = wi + 2u * lane;
        uint4 v = make_uint4(0, 0, 0, 0);
        uint2 e = make_uint2(0, 0);
        if (i0 + 2u <= g.len) v = *(const uint4 *)(rec.r + i0);
        else if (i0 < g.len) { const uint2 x = rec.r[i0]; v.x = x.x; v.y = x.y; }
        if (lane == 63u && i0 + 2u < g.len) e = rec.r[i0 + 2u];
        asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w), "+v"(e.x), "+v"(e.y));      /* all loads are made before one is looked at */
        uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.x, 0x130, 0xf, 0xf, true);    /* wave_shl:1: the next lane's */
        uint32_t ny = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v.y, 0x130, 0xf, 0xf, true);
        if (lane == 63u) { nx = e.x; ny = e.y; }
        rec.p0 = g.dict_len + i0; rec.a = make_uint2(v.x, v.y); rec.b = make_uint2(v.z, v.w); rec.c = make_uint2(nx, ny);
        uint32_t h0 = 0, h1 = 0;
        if (i0 < g.len) h0 = zc_resolve_pos(rec, g.dict_len + i0, E, J.P);
        if (i0 + 1u < g.len) h1 = zc_resolve_pos(rec, g.dict_len + i0 + 1u, E, J.P);
        /* positions without a match start: how far to the next one (position 2 l + e of the wavefront's 128 is lane l's e) */
        const unsigned long long s0 = __ballot((h0 & ZC_HOP_MATCH) != 0), s1 = __ballot((h1 & ZC_HOP_MATCH) != 0);
        const unsigned long long above = (~0ull << lane) << 1;
        const unsigned long long m0 = s0 & above, m1 = s1 & above;
        const uint32_t left = g.len - wi;
        uint32_t np = left < 128u ? left : 128u;
        if (m0) np = min(np, 2u * ((uint32_t)__ffsll((long long)m0) - 1u));
        if (m1) np = min(np, 2u * ((uint32_t)__ffsll((long long)m1) - 1u) + 1u);
        if (!(h1 & ZC_HOP_MATCH)) h1 = np - (2u * lane + 1u);
        if (!(h0 & ZC_HOP_MATCH)) h0 = (h1 & ZC_HOP_MATCH) ? 1u : np - 2u * lane;
        if (i0 + 1u < g.len) *(uint2 *)(hop + i0) = make_uint2(h0, h1);
        else if (i0 < g.len) hop[i0] = h0;
    }
}

#define ZK_HOP_TILE 512u
/* SPEC: the hop words come from k_follow: 0 = nobody has been at this position, literals are single steps */
template <bool SPEC>
__global__ __launch_bounds__(64) void k_hop(ZkJob J)
{
    __shared__ uint32_t s_tile[ZK_HOP_TILE];
    const uint32_t lc = blockIdx.x, lane = threadIdx.x;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    const uint32_t *hop = J.hop + (uint64_t)lc * J.P.chunk_size;
    uint32_t *hopw = J.hop + (uint64_t)lc * J.P.chunk_size;
    uint32_t *arr = J.arr + (uint64_t)lc * (J.P.chunk_size / 2);
    GlbWin gwin; gwin.b = J.in + g.dstart; gwin.n = J.n - g.dstart;
    GlbPd gpd; gpd.pd = J.pdist + (g.dstart - J.pdist_base); gpd.len = J.pdist_len - (g.dstart - J.pdist_base);
    uint32_t i = 0, H = 0, base = 0xffff0000u, nfall = 0;
    while (i < g.len) {
        if (i - base >= ZK_HOP_TILE) {
            base = i & ~(ZK_HOP_TILE - 1u);
            __syncthreads();
            for (uint32_t k = lane * 4u; k < ZK_HOP_TILE; k += 256u) {
                uint32_t idx = base + k;
                uint4 v = make_uint4(1, 1, 1, 1);
                if (idx + 4u <= g.len) v = *(const uint4 *)(hop + idx);
                else { if (idx < g.len) v.x = hop[idx]; if (idx + 1 < g.len) v.y = hop[idx + 1]; if (idx + 2 < g.len) v.z = hop[idx + 2]; }
                *(uint4 *)(s_tile + k) = v;
            }
            __syncthreads();
        }
        uint32_t h = s_tile[i - base];
        const bool fell = SPEC && h == 0;
        if (fell) {
            /* nobody has been here (k_follow): search as the parser would, from global memory; every lane the same */
            h = zc_hop_direct(gwin, gpd, g.dict_len + i, g.dict_len + g.len, g.corner, J.P);
            if (!h) h = 1;
            nfall++;
        }
        if (h & ZC_HOP_MATCH) {
            if (lane == 0) arr[H] = i;
            if (lane == 0 && fell) hopw[i] = h;      /* k_expand reads the hop word */
            H++;
            i += zc_hop_k(h) + zc_hop_len(h);
        } else if (!SPEC || h != 1) i += h;
        else {
            /* a run of single literals: every "1" that follows a "1" is on the parse as well */
            uint32_t j = i - base + 1u + lane;
            unsigned long long m = __ballot(j >= ZK_HOP_TILE || s_tile[j] != 1u);
            i += 1u + (m ? (uint32_t)__ffsll((long long)m) - 1u : 64u);
        }
    }
    if (lane == 0) { J.nhop[lc] = H; if (nfall) atomicAdd(J.err + 1, nfall); }
}

/* The same walk for the hop words of k_resolve (no searches of its own), 64 positions at a time: lane l holds the
 * offset that the hop word of position i + l leads to; the walk inside the 64 is a chain of v_readlane in scalar
 * registers (five instructions a hop) that leaves the set of positions it stood on as a bit mask, and the positions of
 * that set where a match starts are stored by their own lanes, ranked by the mask.  The hop words come through an LDS
 * tile that is fetched one tile ahead. */
__global__ __launch_bounds__(64) void k_hopw(ZkJob J)
{
    __shared__ uint32_t s_tile[ZK_HOP_TILE];
    const uint32_t lc = blockIdx.x, lane = threadIdx.x;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    const uint32_t *hop = J.hop + (uint64_t)lc * J.P.chunk_size;
    uint32_t *arr = J.arr + (uint64_t)lc * (J.P.chunk_size / 2);
    const uint32_t len = __builtin_amdgcn_readfirstlane(g.len);
    /* the hop words of one tile, eight per lane, on their way from memory: the tile behind the current one is fetched
     * while the current one is walked.  A chunk's row of hop words is a whole number of tiles long (the chunk size is a
     * multiple of 16 KiB), so a tile is loaded as it stands, without a branch that would wait for the load; the words
     * behind the chunk's last position are never stood on. */
    auto fetch = [&](uint32_t tb, uint4 &v0, uint4 &v1) {
        v0 = *(const uint4 *)(hop + tb + lane * 4u);
        v1 = *(const uint4 *)(hop + tb + 256u + lane * 4u);
    };
    uint32_t i = 0, H = 0, base = 0xffff0000u, pbase = 0xffff0000u;
    uint4 p0 = make_uint4(1, 1, 1, 1), p1 = p0;
    while (i < len) {
        if (i - base >= ZK_HOP_TILE) {
            base = i & ~(ZK_HOP_TILE - 1u);
            if (base != pbase) fetch(base, p0, p1);        /* the first tile, or a hop over a whole tile */
            __syncthreads();
            *(uint4 *)(s_tile + lane * 4u) = p0;
            *(uint4 *)(s_tile + 256u + lane * 4u) = p1;
            __syncthreads();
            pbase = base + ZK_HOP_TILE;
            if (pbase < len) fetch(pbase, p0, p1);
        }
        const uint32_t off = i - base;
        uint32_t wlen = ZK_HOP_TILE - off < 64u ? ZK_HOP_TILE - off : 64u;
        if (len - i < wlen) wlen = len - i;
        wlen = (uint32_t)__builtin_amdgcn_readfirstlane((int)wlen);      /* the walk's loop condition stays on the scalar unit */
        const uint32_t w = s_tile[off + lane < ZK_HOP_TILE ? off + lane : ZK_HOP_TILE - 1u];
        const bool ism = (w & ZC_HOP_MATCH) != 0;
        uint32_t step = ism ? zc_hop_k(w) + zc_hop_len(w) : w;
        if (step == 0) step = 1;                           /* no word of k_resolve is zero; a walk must move */
        const uint32_t to = lane + step;
        unsigned long long stood = 0;
        uint32_t cur = 0;
        do {
            stood |= 1ull << cur;
            cur = (uint32_t)__builtin_amdgcn_readlane((int)to, (int)cur);
        } while (cur < wlen);
        const unsigned long long mm = __ballot(ism) & stood;
        if ((mm >> lane) & 1ull) arr[H + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull))] = i + lane;
        H += (uint32_t)__popcll(mm);
        i += cur;
    }
    if (lane == 0) J.nhop[lc] = H;
}

/* The same walk for the hop words of k_resolve, eight chunks to a wavefront.  The walk is serial per chunk and a lone
 * wavefront per chunk spends its 64 lanes on one chain of dependent LDS reads, so that what bounds the kernel is the
 * instruction issue of 32 such wavefronts per CU: here a group of eight lanes owns a chunk (its lanes carry identical
 * values and stage the chunk's hop words through a tile of their own), and one instruction stream advances eight walks. */
#define ZK_EXPAND_THREADS 256
__global__ __launch_bounds__(ZK_EXPAND_THREADS) void k_expand(ZkJob J)
{
    __shared__ uint32_t s_off[ZK_EXPAND_THREADS + 1], s_lit0[ZK_EXPAND_THREADS], s_lits[ZK_EXPAND_THREADS];
    __shared__ uint32_t s_word[ZK_EXPAND_THREADS], s_mpos[ZK_EXPAND_THREADS], s_end[ZK_EXPAND_THREADS];
    __shared__ uint32_t s_wsum[ZK_EXPAND_THREADS / 64], s_wmax[ZK_EXPAND_THREADS / 64];
    __shared__ __attribute__((aligned(8))) uint16_t s_own[4 * ZK_EXPAND_THREADS];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t lc = blockIdx.x;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    const uint32_t E = g.dict_len + g.len, lim = J.P.lit_bufsize - 1;
    /* deflate_slow tallies the symbol that starts at pos in the iteration that stands at pos + 1, refills its window below
     * MIN_LOOKAHEAD and tallies a pending literal after its loop (deflate.c:1744, 1796-1829, 1840-1844); deflate_rle tallies
     * it in the iteration at pos, refills at MAX_MATCH or less and has nothing pending (:1874, 1906-1920) */
    const bool rle = J.P.parser == ZC_PARSE_RLE;
    const uint32_t top_off = rle ? 0u : 1u, look_thr = rle ? ZC_MAX_MATCH + 1u : ZC_MIN_LOOK;
    const uint32_t H = J.nhop[lc];
    const uint32_t *hop = J.hop + (uint64_t)lc * J.P.chunk_size;
    const uint32_t *arr = J.arr + (uint64_t)lc * (J.P.chunk_size / 2);
    uint32_t *sym = J.sym + (uint64_t)lc * J.P.chunk_size;
    ZcBlock *blk = J.blk + (uint64_t)lc * J.P.max_blocks;
    uint32_t base_sym = 0, endp = g.dict_len, tail_lits = 0;
    for (uint32_t t = 0; t <= H; t += ZK_EXPAND_THREADS) {
        const uint32_t j = t + tid;
        uint32_t a = E, k = 0, L = 0, D = 0, myend = E;
        if (j < H) {
            uint32_t ai = arr[j], h = hop[ai];
            a = g.dict_len + ai; k = zc_hop_k(h); L = zc_hop_len(h); D = zc_hop_dist(h);
            myend = a + k + L;
        }
        s_end[tid] = myend;
        __syncthreads();
        const uint32_t prev = tid ? s_end[tid - 1] : endp;
        uint32_t lits = 0, cnt = 0;
        if (j <= H) { lits = (a - prev) + k; cnt = lits + (j < H ? 1u : 0u); }
        if (j == H) tail_lits = lits;
        /* exclusive scan of the symbol counts */
        const uint32_t incl = zk_wave_scan(cnt);
        if (lane == 63) s_wsum[wv] = incl;
        __syncthreads();
        uint32_t woff = 0, total = 0;
        for (uint32_t w = 0; w < ZK_EXPAND_THREADS / 64; w++) { uint32_t x = s_wsum[w]; if (w < wv) woff += x; total += x; }
        s_off[tid] = woff + incl - cnt;
        if (tid == 0) s_off[ZK_EXPAND_THREADS] = total;
        s_lit0[tid] = prev; s_lits[tid] = lits;
        s_word[tid] = j < H ? zc_sym_match(L, D) : 0u; s_mpos[tid] = a + k;
        __syncthreads();
        /* which hop a symbol belongs to: every hop marks its first symbol, and a running maximum over the marks carries
         * the owner to the symbols behind it (1 024 symbols at a time, four per thread) */
        uint32_t carry = 0;                                   /* owner + 1 of the last symbol of the 1 024 before */
        for (uint32_t cx = 0; cx < total; cx += 4u * ZK_EXPAND_THREADS) {
            *(uint2 *)&s_own[4u * tid] = make_uint2(0, 0);
            __syncthreads();
            { const uint32_t so = s_off[tid] - cx; if (cnt != 0u && so < 4u * ZK_EXPAND_THREADS) s_own[so] = (uint16_t)(tid + 1u); }
            __syncthreads();
            const uint2 mk = *(const uint2 *)&s_own[4u * tid];
            uint32_t m[4] = {mk.x & 0xffffu, mk.x >> 16, mk.y & 0xffffu, mk.y >> 16};
            m[1] = max(m[1], m[0]); m[2] = max(m[2], m[1]); m[3] = max(m[3], m[2]);
            const uint32_t incl_m = zk_wave_scan_max(m[3]);
            if (lane == 63) s_wmax[wv] = incl_m;
            __syncthreads();
            uint32_t before = carry, all = carry;
            for (uint32_t w = 0; w < ZK_EXPAND_THREADS / 64; w++) { const uint32_t xm = s_wmax[w]; if (w < wv) before = max(before, xm); all = max(all, xm); }
            before = max(before, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl_m, 0x138, 0xf, 0xf, true));   /* wave_shr:1: the lanes before this one */
            carry = all;
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t x = cx + 4u * tid + u;
                if (x >= total) break;
                const uint32_t lo = max(m[u], before) - 1u;
                const uint32_t o = x - s_off[lo];
                uint32_t word, pos, endpos;
                if (o < s_lits[lo]) { pos = s_lit0[lo] + o; word = pos; endpos = pos + 1; }
                else { word = s_word[lo]; pos = s_mpos[lo]; endpos = pos + ((word >> 16) & 0xffu) + 3u; }
                const uint32_t sidx = base_sym + x;
                sym[sidx] = word;
                if ((sidx + 1) % lim == 0) {
                    uint32_t b = (sidx + 1) / lim - 1;
                    if (b < J.P.max_blocks) { blk[b].bits = endpos; blk[b].hdr_bits = pos + top_off; }
                }
            }
            __syncthreads();                                  /* s_own and s_wmax are written again */
        }
        base_sym += total;
        endp = s_end[ZK_EXPAND_THREADS - 1];
        __syncthreads();
    }
    /* the trailing literal run is owned by the thread that handled pseudo-hop H */
    __shared__ uint32_t s_tail;
    if ((H % ZK_EXPAND_THREADS) == tid) s_tail = tail_lits;
    __syncthreads();
    const uint32_t nsym = base_sym;
    uint32_t nb = zc_block_count(nsym, lim, g.final, !rle && s_tail != 0);
    if (nb > J.P.max_blocks) { if (tid == 0) atomicOr(J.err, (uint32_t)ZE_BLOCK_OVERFLOW); nb = J.P.max_blocks; }
    /* assemble the block table (FLUSH_BLOCK_ONLY, deflate.c:1538-1547) */
    for (uint32_t b0 = 0; b0 < nb; b0 += ZK_EXPAND_THREADS) {
        const uint32_t b = b0 + tid;
        uint32_t in_start = 0, in_end = 0, p_top = 0;
        if (b < nb) {
            in_start = b ? blk[b - 1].bits : g.dict_len;
            if (b + 1 == nb) { in_end = E; p_top = E; } else { in_end = blk[b].bits; p_top = blk[b].hdr_bits; }
        }
        __syncthreads();
        if (b < nb) {
            ZcBlock B;
            B.sym_start = b * lim; B.sym_count = (b + 1 < nb) ? lim : nsym - b * lim;
            B.in_start = in_start; B.in_len = in_end - in_start;
            B.flags = ((b + 1 == nb && g.final) ? ZC_BF_LAST : 0u) |
                      (in_start >= zc_window_base(p_top, E, look_thr, J.P.wsize) ? ZC_BF_CANSTORE : 0u);
            B.type = 0; B.bits = b + 1 == nb ? 0u : in_end; B.hdr_bits = 0; B.bit_off = 0;
            blk[b] = B;
        }
        __syncthreads();
    }
    if (tid == 0) { J.nblk[lc] = nb; J.nsym[lc] = nsym; }
}

/* ------------------------------------------------------------------ k_plan */
__global__ __launch_bounds__(ZK_PLAN_THREADS) void k_plan(ZkJob J)
{
    __shared__ ZcTree<ZC_L_CODES> lt;
    __shared__ ZcTree<ZC_D_CODES> dt;
    __shared__ ZcTree<ZC_BL_CODES> bt;
    __shared__ uint32_t cost[4];
    uint32_t *const hist = lt.pq;                 /* 288 + 32 counters; the queue takes the array over once the counts are in freq[] */
    const uint32_t tid = threadIdx.x;
    const uint32_t lc = blockIdx.y, j = blockIdx.x;
    if (j >= J.nblk[lc]) return;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    ZcBlock *bp = J.blk + (uint64_t)lc * J.P.max_blocks + j;
    ZcBlock b = *bp;
    if (b.flags & ZC_BF_STORED) return;          /* type 0 already; nothing to count or build */
    const uint8_t *src = J.in + g.dstart;
    const uint32_t *sym = J.sym + (uint64_t)lc * J.P.chunk_size + b.sym_start;
    for (uint32_t i = tid; i < 320; i += ZK_PLAN_THREADS) hist[i] = 0;
    if (tid < 4) cost[tid] = 0;
    __syncthreads();
    /* _tr_tally frequency counts (trees.c:1017-1030) */
    if (b.flags & ZC_BF_IMPLICIT) {
        for (uint32_t i = tid; i < b.sym_count; i += ZK_PLAN_THREADS) atomicAdd(&hist[src[b.in_start + i]], 1u);
    } else if (b.sym_count >= 8u) {
        /* eight consecutive symbols per lane at a time, loaded without a branch (the last lanes of the block step back so
         * that their eight end with it, and skip what the lanes before them have counted), then the eight literal bytes
         * (a match reads byte 0): two trips to memory per 512 symbols */
        for (uint32_t i0 = 0; i0 < b.sym_count; i0 += 8u * ZK_PLAN_THREADS) {
            const uint32_t want = i0 + tid * 8u, from = want + 8u <= b.sym_count ? want : b.sym_count - 8u;
            const uint4 sa = *(const uint4 *)(sym + from), sb = *(const uint4 *)(sym + from + 4u);
            const uint32_t s[8] = {sa.x, sa.y, sa.z, sa.w, sb.x, sb.y, sb.z, sb.w};
            uint32_t lit[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) lit[u] = src[(s[u] & ZC_SYM_MATCH) ? 0u : s[u]];
#pragma unroll
            for (uint32_t u = 0; u < 8; u++) {
                if (from + u < want) continue;
                if (!(s[u] & ZC_SYM_MATCH)) atomicAdd(&hist[lit[u]], 1u);
                else {
                    atomicAdd(&hist[257u + zc_len_code((s[u] >> 16) & 0xffu)], 1u);
                    atomicAdd(&hist[288u + zc_dist_code((s[u] & 0xffffu) - 1u)], 1u);
                }
            }
        }
    } else {
        for (uint32_t i = tid; i < b.sym_count; i += ZK_PLAN_THREADS) {
            const uint32_t s = sym[i];
            if (!(s & ZC_SYM_MATCH)) atomicAdd(&hist[src[s]], 1u);
            else {
                atomicAdd(&hist[257u + zc_len_code((s >> 16) & 0xffu)], 1u);
                atomicAdd(&hist[288u + zc_dist_code((s & 0xffffu) - 1u)], 1u);
            }
        }
    }
    __syncthreads();
    if (tid == 0 && j == 0 && J.chunk0 + lc == 0) {
        /* detect_data_type on the first block of the call (trees.c:920-921, 1120-1147): 1 + Z_BINARY / Z_TEXT */
        uint32_t dt_ = 0, n;
        for (n = 0; n < 32; n++) if (((0xf3ffc07fu >> n) & 1u) && hist[n]) break;
        if (n == 32) {
            if (hist[9] || hist[10] || hist[13]) dt_ = 1;
            else for (n = 32; n < 256; n++) if (hist[n]) { dt_ = 1; break; }
        }
        J.err[2] = 1u + dt_;
    }
    for (uint32_t i = tid; i < ZC_L_CODES; i += ZK_PLAN_THREADS) lt.freq[i] = (uint16_t)(i == ZC_END_BLOCK ? 1u : hist[i]);
    if (tid < ZC_D_CODES) dt.freq[tid] = (uint16_t)hist[288 + tid];
    __syncthreads();
    if (tid == 0) {
        uint32_t o = 0, s = 0;
        zc_build_tree(lt, 15, true, [](int n) { return zc_static_llen((uint32_t)n); },
                      [](int n) { return n >= 257 ? zc_len_extra((uint32_t)n - 257u) : 0u; }, &o, &s);
        cost[0] = o; cost[1] = s;
    } else if (tid == 32) {
        uint32_t o = 0, s = 0;
        zc_build_tree(dt, 15, true, [](int) { return 5u; },
                      [](int n) { return zc_dist_extra((uint32_t)n); }, &o, &s);
        cost[2] = o; cost[3] = s;
    }
    __syncthreads();
    if (tid == 0) {
        ZcBlockTab *tab = J.tab + (uint64_t)lc * J.P.max_blocks + j;
        zc_plan_block(lt, dt, bt, cost[0] + cost[2], cost[1] + cost[3], J.P.strategy, &b, tab);
        bp->type = b.type; bp->bits = b.bits; bp->hdr_bits = b.hdr_bits;
        if (lt.deep | dt.deep | bt.deep) atomicOr(J.err, (uint32_t)ZE_TREE_DEPTH);
    }
}

/* ------------------------------------------------------------------ k_layout / k_scan */
__global__ void k_layout(ZkJob J)
{
    uint32_t lc = blockIdx.x * blockDim.x + threadIdx.x;
    if (lc >= J.nchunks) return;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    ZcBlock *base = J.blk + (uint64_t)lc * J.P.max_blocks;
    uint64_t sync = 0;
    /* a non-final chunk that is the last of a call without finish still ends on a sync marker */
    const uint32_t pb = (J.chunk0 + lc == 0) ? J.prime_bits : 0u;
    if (pb) atomicOr(&J.out32[0], J.prime_val);
    const bool call_end = J.chunk0 + lc == J.last_chunk;
    uint32_t left = 0;
    J.chunk_bytes[lc] = zc_layout_chunk([&](uint32_t i) { return base + i; }, J.nblk[lc], g.final, &sync, pb,
                                        call_end ? J.P.tail : (uint32_t)ZC_TAIL_SYNC, &left);
    J.sync_bit[lc] = sync;
    if (call_end) J.err[3] = left;
    if (J.splits) {
        /* Where a reader may cut the chunk into split_parts parts that take about the same time to decode, which goes by
         * symbols, not bytes (a block of literals and a block of long matches take the same): part k starts at the first
         * block in front of which k / split_parts of the chunk's symbols lie (a stored block counts as one symbol per 64
         * bytes).  A record is (bit offset from the chunk's first byte) << 32 | offset of the block's first byte in the
         * chunk's data; strictly ascending, packed to the front, zeros behind. */
        uint64_t *sp = J.splits + (J.chunk0 + lc) * (uint64_t)(J.split_parts - 1u);
        const uint32_t nb = J.nblk[lc];
        auto weight = [&](uint32_t j) { return base[j].type == 0 ? base[j].in_len / 64u + 1u : base[j].sym_count + 1u; };
        uint64_t total = 0;
        for (uint32_t j = 0; j < nb; j++) total += weight(j);
        uint32_t j = 0, last_off = 0, w = 0;
        uint64_t before = 0;                                /* symbols in front of block j */
        for (uint32_t k = 1; k < J.split_parts; k++) {
            const uint64_t target = total * k / J.split_parts;
            while (j < nb && (before < target || j == 0)) { before += weight(j); j++; }
            if (j >= nb) break;
            const uint32_t off = base[j].in_start - g.dict_len;
            if (off > last_off && off < g.len) { sp[w++] = (base[j].bit_off << 32) | off; last_off = off; }
        }
        while (w < J.split_parts - 1u) sp[w++] = 0;
    }
}

__global__ __launch_bounds__(1024) void k_scan(ZkJob J)
{
    /* exclusive scan of the batch's chunk sizes on top of the running total */
    __shared__ uint64_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (J.nchunks + 1023) / 1024;
    uint64_t sum = 0;
    for (uint32_t k = 0; k < per; k++) { uint32_t i = tid * per + k; if (i < J.nchunks) sum += J.chunk_bytes[i]; }
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) { uint64_t run = *J.running_total; for (uint32_t i = 0; i < 1024; i++) { uint64_t v = part[i]; part[i] = run; run += v; }
                    *J.running_total = run; J.chunk_off[J.chunk0 + J.nchunks] = run;
                    if (run > J.out_cap) atomicOr(J.err, (uint32_t)ZE_OUT_OVERFLOW); }
    __syncthreads();
    uint64_t run = part[tid];
    for (uint32_t k = 0; k < per; k++) { uint32_t i = tid * per + k; if (i < J.nchunks) { J.chunk_off[J.chunk0 + i] = run; run += J.chunk_bytes[i]; } }
}

/* ------------------------------------------------------------------ k_emit */
__device__ inline void g_or_bits(uint32_t *out32, uint64_t bitpos, uint64_t v, uint32_t n)
{
    if (!n) return;
    uint64_t w = bitpos >> 5; uint32_t sh = (uint32_t)bitpos & 31u;
    uint32_t w0 = (uint32_t)(v << sh);
    uint64_t rest = sh ? (v >> (32u - sh)) : (v >> 32);
    uint32_t w1 = (uint32_t)rest, w2 = (uint32_t)(rest >> 32);
    if (w0) atomicOr(&out32[w], w0);
    if (w1) atomicOr(&out32[w + 1], w1);
    if (w2) atomicOr(&out32[w + 2], w2);
}
__device__ inline void lds_or_bits(uint32_t *st, uint32_t bitpos, uint64_t v, uint32_t n)
{
    if (!n) return;
    uint32_t w = bitpos >> 5, sh = bitpos & 31u;
    uint32_t w0 = (uint32_t)(v << sh);
    uint64_t rest = sh ? (v >> (32u - sh)) : (v >> 32);
    uint32_t w1 = (uint32_t)rest, w2 = (uint32_t)(rest >> 32);
    if (w0) atomicOr(&st[w], w0);
    if (w1) atomicOr(&st[w + 1], w1);
    if (w2) atomicOr(&st[w + 2], w2);
}

__global__ __launch_bounds__(ZK_EMIT_THREADS) void k_emit(ZkJob J)
{
    __shared__ uint32_t stage[ZK_STAGE_WORDS];
    __shared__ uint32_t wsum[ZK_EMIT_THREADS / 64];
    __shared__ uint32_t ltab[288], dtab[32];
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint32_t lc = blockIdx.y, j = blockIdx.x;
    if (*J.err) return;
    const uint32_t nb = J.nblk[lc];
    if (j > nb) return;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    const uint64_t cbit = J.chunk_off[J.chunk0 + lc] * 8ull;
    uint32_t *out32 = J.out32;
    if (j == nb) {
        const uint32_t tail = J.chunk0 + lc == J.last_chunk ? J.P.tail : (uint32_t)ZC_TAIL_SYNC;
        if (!g.final && tid == 0) {
            if (tail == ZC_TAIL_SYNC) {               /* Z_SYNC_FLUSH marker: 000, pad, 00 00 FF FF */
                uint64_t bit = cbit + J.sync_bit[lc] + 3;
                bit = (bit + 7) & ~7ull;
                g_or_bits(out32, bit + 16, 0xffffu, 16);
            } else if (tail == ZC_TAIL_ALIGN)         /* _tr_align: STATIC_TREES << 1 in 3 bits, then the all-zero code of END_BLOCK */
                g_or_bits(out32, cbit + J.sync_bit[lc], 2u, 10);
        }
        return;
    }
    const ZcBlock b = J.blk[(uint64_t)lc * J.P.max_blocks + j];
    const uint8_t *src = J.in + g.dstart;
    const uint64_t bit0 = cbit + b.bit_off;
    const uint32_t last = b.flags & ZC_BF_LAST;
    if (b.type == 0) {                            /* _tr_stored_block, trees.c:865-877 */
        uint64_t byte = ((bit0 + 3 + 7) >> 3);
        if (tid == 0) {
            g_or_bits(out32, bit0, last, 3);
            uint32_t l = b.in_len & 0xffffu;
            g_or_bits(out32, byte * 8, l | ((~l & 0xffffu) << 16), 32);
        }
        byte += 4;
        for (uint32_t i = tid * 4; i < b.in_len; i += ZK_EMIT_THREADS * 4) {
            uint32_t v = 0, k = 0;
            for (; k < 4 && i + k < b.in_len; k++) v |= (uint32_t)src[b.in_start + i + k] << (8 * k);
            g_or_bits(out32, (byte + i) * 8, v, 8 * k);
        }
        return;
    }
    const ZcBlockTab *tab = J.tab + (uint64_t)lc * J.P.max_blocks + j;
    for (uint32_t i = tid; i < 288; i += ZK_EMIT_THREADS) ltab[i] = tab->lc[i];
    if (tid < 32) dtab[tid] = tab->dc[tid];
    /* block header: 3 type bits, then the dynamic tree description (send_all_trees) */
    if (tid == 0) g_or_bits(out32, bit0, (b.type << 1) | last, 3);
    for (uint32_t i = tid; i * 32 < b.hdr_bits; i += ZK_EMIT_THREADS) {
        uint32_t nbits = b.hdr_bits - i * 32 < 32 ? b.hdr_bits - i * 32 : 32;
        uint32_t v = (uint32_t)tab->hdr[4 * i] | ((uint32_t)tab->hdr[4 * i + 1] << 8) |
                     ((uint32_t)tab->hdr[4 * i + 2] << 16) | ((uint32_t)tab->hdr[4 * i + 3] << 24);
        if (nbits < 32) v &= (1u << nbits) - 1u;
        g_or_bits(out32, bit0 + 3 + i * 32, v, nbits);
    }
    __syncthreads();
    const uint32_t *sym = J.sym + (uint64_t)lc * J.P.chunk_size + b.sym_start;
    uint64_t bitpos = bit0 + 3 + b.hdr_bits;
    const uint32_t total = b.sym_count + 1;       /* + END_BLOCK */
    /* The tile's symbols and literal bytes are on their way while the tile before it is coded: symbol words two tiles
     * ahead, the literal bytes they point at one tile ahead, loaded without a branch (past the block's end: its last
     * symbol, not used), and the barriers of the loop wait for LDS only. */
    const bool implicit = (b.flags & ZC_BF_IMPLICIT) != 0;
    const uint32_t last_sym = b.sym_count ? b.sym_count - 1u : 0u;
    auto load_sym = [&](uint32_t i) -> uint32_t { return (implicit || !b.sym_count) ? 0u : sym[i < last_sym ? i : last_sym]; };
    auto load_lit = [&](uint32_t i, uint32_t s) -> uint32_t {
        const uint32_t a = implicit ? b.in_start + (i < last_sym ? i : last_sym) : ((s & ZC_SYM_MATCH) ? 0u : s);
        return src[a];
    };
    uint32_t s_cur = load_sym(tid), s_n1 = load_sym(ZK_EMIT_THREADS + tid);
    uint32_t lit_cur = load_lit(tid, s_cur);
    for (uint32_t t = 0; t < total; t += ZK_EMIT_THREADS) {
        uint32_t i = t + tid;
        const uint32_t lit_n1 = load_lit(i + ZK_EMIT_THREADS, s_n1);
        const uint32_t s_n2 = load_sym(i + 2u * ZK_EMIT_THREADS);
        uint64_t v = 0; uint32_t n = 0;
        if (i < b.sym_count) n = zc_sym_bits(s_cur, lit_cur, ltab, dtab, &v);
        else if (i == b.sym_count) { v = ltab[ZC_END_BLOCK] & 0xffffu; n = ltab[ZC_END_BLOCK] >> 16; }
        s_cur = s_n1; s_n1 = s_n2; lit_cur = lit_n1;
        /* exclusive prefix sum of code lengths over the 256-symbol tile */
        const uint32_t incl = zk_wave_scan(n);
        if (lane == 63) wsum[wv] = incl;
        for (uint32_t k = tid; k < ZK_STAGE_WORDS; k += ZK_EMIT_THREADS) stage[k] = 0;
        ZK_LDS_BARRIER();
        uint32_t woff = 0, tile_bits = 0;
        for (uint32_t k = 0; k < ZK_EMIT_THREADS / 64; k++) { uint32_t ws = wsum[k]; if (k < wv) woff += ws; tile_bits += ws; }
        const uint32_t sh = (uint32_t)bitpos & 31u;
        lds_or_bits(stage, sh + woff + incl - n, v, n);
        ZK_LDS_BARRIER();
        const uint32_t nwords = (sh + tile_bits + 31) >> 5;
        const uint64_t w0 = bitpos >> 5;
        for (uint32_t k = tid; k < nwords; k += ZK_EMIT_THREADS) { uint32_t x = stage[k]; if (x) atomicOr(&out32[w0 + k], x); }
        bitpos += tile_bits;
        ZK_LDS_BARRIER();
    }
}

/* ------------------------------------------------------------------ checksums
 * adler32 (adler32.c:65-133) and crc32 (crc32.c:204-278) of one slice per
 * workgroup.  Each lane reduces a contiguous sub-slice; partial results are
 * merged with the combine identities (adler32.c:136, crc32.c:355). */
#define ZK_CRC_POLY 0xedb88320u
__device__ __host__ inline uint32_t gf2_mul(uint32_t a, uint32_t b)
{   /* product of two polynomials mod the CRC polynomial, bit 31 = x^0 */
    uint32_t p = 0;
    for (uint32_t m = 0x80000000u; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ ZK_CRC_POLY : b >> 1;
    }
    return p;
}
__device__ __host__ inline uint32_t gf2_xpow8n(uint64_t nbytes, const uint32_t *x2n)
{   /* x^(8*nbytes); x2n[k] = x^(2^k) */
    uint32_t p = 0x80000000u; uint32_t k = 3;
    while (nbytes) { if (nbytes & 1u) p = gf2_mul(x2n[k & 63u], p); nbytes >>= 1; k++; }
    return p;
}
struct ZkCrcConst { uint32_t tab[256]; uint32_t x2n[64]; };
__constant__ ZkCrcConst c_crc;
static ZkCrcConst h_crc; static bool h_crc_ready = false;
static void crc_host_init()
{
    if (h_crc_ready) return;
    for (uint32_t n = 0; n < 256; n++) { uint32_t c = n; for (int k = 0; k < 8; k++) c = (c & 1u) ? ZK_CRC_POLY ^ (c >> 1) : c >> 1; h_crc.tab[n] = c; }
    uint32_t p = 0x40000000u;                     /* x^1 */
    h_crc.x2n[0] = p;
    for (int k = 1; k < 64; k++) { p = gf2_mul(p, p); h_crc.x2n[k] = p; }
    h_crc_ready = true;
}

__global__ __launch_bounds__(256) void k_checksum(const uint8_t *in, uint64_t n, uint32_t slice, uint32_t *adler_out, uint32_t *crc_out)
{
    __shared__ uint32_t s_tab[256];
    __shared__ unsigned long long s_a[256], s_w[256];
    __shared__ uint32_t s_c[256];
    const uint32_t tid = threadIdx.x;
    const uint64_t start = (uint64_t)blockIdx.x * slice;
    const uint32_t L = (uint32_t)((n - start) < slice ? (n - start) : slice);
    const uint8_t *p = in + start;
    s_tab[tid] = c_crc.tab[tid];
    __syncthreads();
    uint32_t seg = (L + 255) / 256; seg = (seg + 15) & ~15u;
    uint32_t b0 = tid * seg, b1 = b0 + seg; if (b0 > L) b0 = L; if (b1 > L) b1 = L;
    unsigned long long A = 0, W = 0; uint32_t crc = 0;
    for (uint32_t o = b0; o < b1; ) {
        if (o + 16 <= b1 && (((uintptr_t)(p + o)) & 15u) == 0) {
            uint4 q = *(const uint4 *)(p + o);
            uint32_t w[4] = {q.x, q.y, q.z, q.w};
            uint32_t sb = 0, sj = 0;
            for (int k = 0; k < 4; k++) {
                uint32_t x = w[k];
                uint32_t y0 = x & 0xff, y1 = (x >> 8) & 0xff, y2 = (x >> 16) & 0xff, y3 = x >> 24;
                sb += y0 + y1 + y2 + y3;
                sj += (4 * k) * y0 + (4 * k + 1) * y1 + (4 * k + 2) * y2 + (4 * k + 3) * y3;
                crc ^= x;
                crc = s_tab[crc & 0xff] ^ (crc >> 8); crc = s_tab[crc & 0xff] ^ (crc >> 8);
                crc = s_tab[crc & 0xff] ^ (crc >> 8); crc = s_tab[crc & 0xff] ^ (crc >> 8);
            }
            A += sb; W += (unsigned long long)(L - o) * sb - sj;
            o += 16;
        } else {
            uint32_t y = p[o];
            A += y; W += (unsigned long long)(L - o) * y;
            crc = s_tab[(crc ^ y) & 0xff] ^ (crc >> 8);
            o++;
        }
    }
    /* shift this lane's raw CRC over the bytes that follow it */
    crc = gf2_mul(gf2_xpow8n(L - b1, c_crc.x2n), crc);
    s_a[tid] = A; s_w[tid] = W; s_c[tid] = crc;
    __syncthreads();
    for (uint32_t d = 128; d; d >>= 1) {
        if (tid < d) { s_a[tid] += s_a[tid + d]; s_w[tid] += s_w[tid + d]; s_c[tid] ^= s_c[tid + d]; }
        __syncthreads();
    }
    if (tid == 0) {
        if (adler_out) {
            uint32_t s1 = (uint32_t)((1ull + s_a[0]) % 65521ull);
            uint32_t s2 = (uint32_t)(((unsigned long long)L + s_w[0]) % 65521ull);
            adler_out[blockIdx.x] = s1 | (s2 << 16);
        }
        if (crc_out) crc_out[blockIdx.x] = gf2_mul(gf2_xpow8n(L, c_crc.x2n), 0xffffffffu) ^ s_c[0] ^ 0xffffffffu;
    }
}

/* adler32_combine (adler32.c:136-162) for kernels and host alike */
__device__ __host__ inline uint32_t zk_adler_combine(uint32_t a1, uint32_t a2, uint64_t len2)
{
    const uint32_t B = 65521u;
    uint32_t rem = (uint32_t)(len2 % B), s1 = a1 & 0xffffu, s2 = (uint32_t)(((uint64_t)rem * s1) % B);
    s1 += (a2 & 0xffffu) + B - 1;
    s2 += (a1 >> 16) + (a2 >> 16) + B - rem;
    if (s1 >= B) s1 -= B;
    if (s1 >= B) s1 -= B;
    if (s2 >= (B << 1)) s2 -= (B << 1);
    if (s2 >= B) s2 -= B;
    return s1 | (s2 << 16);
}

/* ------------------------------------------------------------------ whole-buffer checksums without any allocation
 * Per-slice partial checksums go to a fixed device array of the code object instead of a buffer from hipMalloc, and
 * are folded on the device, so that the call neither allocates nor copies to the host.  One array per GPU, shared by
 * every call on that GPU: calls that overlap on one GPU race. */
#define ZNA_SLICE      (1u << 20)
#define ZNA_MAX_SLICES 16384u                     /* inputs of up to 16 GiB */
static __device__ __align__(256) uint32_t scratch[2u * ZNA_MAX_SLICES];   /* [0, MAX): adler32 per slice, [MAX, 2 MAX): crc32 per slice */
#define ZNA_MAX_DEVICES 64
static uint32_t *g_na_scratch[ZNA_MAX_DEVICES];   /* device address of `scratch` on each GPU, from hipGetSymbolAddress */

/* Folds the per-slice values (slices of `slice` bytes, the last one shorter) onto the seeds: one workgroup; every
 * thread folds a run of consecutive slices, then the runs are folded pairwise in LDS. */
__global__ __launch_bounds__(1024) void k_ck_fold(const uint32_t *part_a, const uint32_t *part_c, uint32_t nslices, uint64_t n, uint32_t slice,
                                                  uint32_t seed_a, uint32_t seed_c, uint32_t *out2)
{
    __shared__ uint32_t s_a[1024], s_c[1024];
    __shared__ unsigned long long s_len[1024];
    const uint32_t tid = threadIdx.x;
    const uint32_t per = (nslices + 1023u) / 1024u;
    uint32_t a = 1u, c = 0u; unsigned long long len = 0;      /* checksums of the empty string */
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t i = tid * per + k;
        if (i >= nslices) break;
        const unsigned long long off = (unsigned long long)i * slice;
        const unsigned long long l = n - off < slice ? n - off : slice;
        a = zk_adler_combine(a, part_a[i], l);
        c = gf2_mul(gf2_xpow8n(l, c_crc.x2n), c) ^ part_c[i];
        len += l;
    }
    s_a[tid] = a; s_c[tid] = c; s_len[tid] = len;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        if ((tid & (2 * d - 1)) == 0) {
            const unsigned long long l2 = s_len[tid + d];
            if (l2) {
                s_a[tid] = zk_adler_combine(s_a[tid], s_a[tid + d], l2);
                s_c[tid] = gf2_mul(gf2_xpow8n(l2, c_crc.x2n), s_c[tid]) ^ s_c[tid + d];
                s_len[tid] += l2;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        out2[0] = zk_adler_combine(seed_a, s_a[0], s_len[0]);
        out2[1] = gf2_mul(gf2_xpow8n(s_len[0], c_crc.x2n), seed_c) ^ s_c[0];
    }
}

/* ------------------------------------------------------------------ host side */
#define HIPCHK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "zamd: %s failed: %s (%s:%d)\n", #x, hipGetErrorString(_e), __FILE__, __LINE__); return ZAMD_HIP_ERROR(_e); } } while (0)

struct ZeWorkspace {
    uint32_t cap_chunks = 0, chunk_size = 0, max_blocks = 0;
    bool has_match = false;       /* pdist, rec and hop are allocated (the parsers that search ahead of the parse use them) */
    uint16_t *pdist = nullptr; uint2 *rec = nullptr; uint32_t *sym = nullptr; uint32_t *hop = nullptr; uint32_t *nhop = nullptr;
    ZcBlock *blk = nullptr; ZcBlockTab *tab = nullptr;
    uint32_t *nblk = nullptr, *nsym = nullptr; uint64_t *chunk_bytes = nullptr, *sync_bit = nullptr;
    uint64_t pdist_len = 0;
    uint16_t *fhead = nullptr; uint64_t fhead_bytes = 0;
};

#define ZE_STREAMS 8            /* batches in flight at most */
struct ZiState;
struct zamd_engine {
    int device, cu_count;
    ZiState *inf;                 /* batch-inflate workspace, created on first use */
    hipStream_t st[ZE_STREAMS];
    void (*on_launched)(void *); void *on_launched_arg;   /* called once per deflate call when its last kernel is queued */
    uint8_t *h_stage[2][2];         /* pinned staging for zamd_dev_upload_staged / _download_staged: [direction][half], ZE_STAGE_BYTES each */
    uint32_t *h_res;                /* 64 bytes of pinned host memory: where a call's totals and flags arrive */
    hipStream_t st_up, st_down;     /* host copies run on a stream of their own (a plain hipMemcpy takes the null stream), so that one thread's
                                     * transfers and the kernels of another thread's deflate call overlap.  A process has GPU_MAX_HW_QUEUES
                                     * (4) hardware queues and streams beyond that share them: a copy queued behind this engine's kernels in
                                     * a shared queue was measured at a quarter of its speed.  Hence one stream for both directions, and a
                                     * caller that overlaps transfers with a deflate call runs that call on three streams
                                     * (zamd_engine_set_streams). */
    hipEvent_t ev_scan[ZE_STREAMS], ev_init;
    ZeWorkspace ws[ZE_STREAMS];
    int nstreams;
    int serialize;                  /* batches are cut as for nstreams in flight but issued to one stream: every kernel alone on the GPU */
    int spec_match;                 /* follow the parse (k_follow) instead of searching every position (k_match) */
    uint32_t prime_bits, prime_val;   /* deflatePrime residue for the next deflate call only */
    int wbits;                        /* windowBits of the following deflate calls (9..15) */
    uint32_t tail_mode;               /* ZC_TAIL_* for the next deflate call only */
    uint64_t *split_out; uint32_t split_parts;   /* where the next deflate call (only) records its chunks' split points */
    uint32_t pend_bits, pend_val;     /* bits the last call left behind its last complete byte */
    int data_type;                    /* detect_data_type of the last call's first block, or Z_UNKNOWN (2) */
    int tuned; uint32_t tune[4];   /* deflateTune values for the following calls (good, lazy, nice, chain) */
    uint64_t ws_budget;
    uint64_t *d_total; uint32_t *d_err; uint32_t last_fallback_hops;
    uint64_t *d_off_own; uint64_t d_off_cap;
    /* timing */
    hipEvent_t ev_call[2];
    struct Tm { hipEvent_t a, b; int which; };
    Tm *tm; int n_tm, cap_tm;
    double t_ms[12]; int t_n[12];
    /* last job for debugging */
    ZkJob last_job; bool have_last;
    uint32_t *d_ck_a, *d_ck_c; uint64_t ck_cap;
};

static void ws_free(ZeWorkspace &w)
{
    hipFree(w.pdist); hipFree(w.rec); hipFree(w.sym); hipFree(w.hop); hipFree(w.nhop); hipFree(w.blk); hipFree(w.tab);
    hipFree(w.nblk); hipFree(w.nsym); hipFree(w.chunk_bytes); hipFree(w.sync_bit); hipFree(w.fhead);
    w = ZeWorkspace();
}

/* with_match: links (2 B), records (8 B) and hop words (4 B) per position, for the parsers that search ahead of the parse */
static uint64_t ws_bytes_per_chunk(uint32_t chunk_size, uint32_t max_blocks, bool with_match)
{
    return (with_match ? 14ull : 0ull) * chunk_size + 4ull * chunk_size +
           (uint64_t)max_blocks * (sizeof(ZcBlock) + sizeof(ZcBlockTab)) + 64;
}

static int ws_ensure(ZeWorkspace &w, uint32_t chunks, uint32_t chunk_size, uint32_t max_blocks, bool with_match)
{
    if (w.cap_chunks >= chunks && w.chunk_size == chunk_size && w.max_blocks == max_blocks && (w.has_match || !with_match)) return 0;
    const bool keep_match = with_match || w.has_match;
    ws_free(w);
    if (keep_match) {
        w.pdist_len = (uint64_t)chunks * chunk_size + ZC_WSIZE + 64;
        HIPCHK(hipMalloc((void **)&w.pdist, w.pdist_len * 2));
        HIPCHK(hipMalloc((void **)&w.rec, (uint64_t)chunks * chunk_size * 8));
        HIPCHK(hipMalloc((void **)&w.hop, (uint64_t)chunks * chunk_size * 4));
        w.has_match = true;
    }
    HIPCHK(hipMalloc((void **)&w.sym, (uint64_t)chunks * chunk_size * 4));
    HIPCHK(hipMalloc((void **)&w.nhop, (uint64_t)chunks * 4));
    HIPCHK(hipMalloc((void **)&w.blk, (uint64_t)chunks * max_blocks * sizeof(ZcBlock)));
    HIPCHK(hipMalloc((void **)&w.tab, (uint64_t)chunks * max_blocks * sizeof(ZcBlockTab)));
    HIPCHK(hipMalloc((void **)&w.nblk, (uint64_t)chunks * 4));
    HIPCHK(hipMalloc((void **)&w.nsym, (uint64_t)chunks * 4));
    HIPCHK(hipMalloc((void **)&w.chunk_bytes, (uint64_t)chunks * 8));
    HIPCHK(hipMalloc((void **)&w.sync_bit, (uint64_t)chunks * 8));
    w.cap_chunks = chunks; w.chunk_size = chunk_size; w.max_blocks = max_blocks;
    return 0;
}

extern "C" const char *zamd_version(void) { return "zlib_amd 0.1 (deflate semantics of zlib 1.2.8)"; }

extern "C" int zamd_engine_create(zamd_engine **out, int device, uint64_t workspace_bytes)
{
    if (!out) return ZAMD_STREAM_ERROR;
    int count = 0;
    hipError_t ce = hipGetDeviceCount(&count);
    if (ce != hipSuccess || count <= 0) {
        fprintf(stderr, "zamd: no HIP device available (%s); this library has no CPU path\n", hipGetErrorString(ce));
        return ZAMD_HIP_ERROR(ce == hipSuccess ? hipErrorNoDevice : ce);
    }
    HIPCHK(hipSetDevice(device));
    zamd_engine *e = new (std::nothrow) zamd_engine();
    if (!e) return ZAMD_MEM_ERROR;
    memset((void *)e, 0, sizeof *e);
    for (int i = 0; i < ZE_STREAMS; i++) e->ws[i] = ZeWorkspace();
    e->nstreams = 4;                 /* one per hardware queue (GPU_MAX_HW_QUEUES defaults to 4) */
    e->wbits = 15; e->data_type = 2;
    e->on_launched = nullptr; e->on_launched_arg = nullptr;
    for (int d = 0; d < 2; d++) for (int h = 0; h < 2; h++) e->h_stage[d][h] = nullptr;
    e->spec_match = getenv("ZAMD_MATCH_FOLLOW") ? 1 : 0;
    if (const char *v = getenv("ZAMD_STREAMS")) { int k = atoi(v); if (k >= 1 && k <= ZE_STREAMS) e->nstreams = k; }
    e->serialize = getenv("ZAMD_SERIALIZE") ? 1 : 0;
    e->device = device;
    { hipDeviceProp_t pr; HIPCHK(hipGetDeviceProperties(&pr, device)); e->cu_count = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256; }
    e->ws_budget = workspace_bytes ? workspace_bytes : (48ull << 30);
    for (int i = 0; i < ZE_STREAMS; i++) {
        HIPCHK(hipStreamCreateWithFlags(&e->st[i], hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&e->ev_scan[i], hipEventDisableTiming));
    }
    HIPCHK(hipHostMalloc((void **)&e->h_res, 64, hipHostMallocDefault));
    HIPCHK(hipStreamCreateWithFlags(&e->st_up, hipStreamNonBlocking));
    e->st_down = e->st_up;         /* one stream for both directions: see ZE_STREAMS_BESIDE_COPIES */
    HIPCHK(hipEventCreateWithFlags(&e->ev_init, hipEventDisableTiming));
    for (int i = 0; i < 2; i++) HIPCHK(hipEventCreate(&e->ev_call[i]));
    HIPCHK(hipMalloc((void **)&e->d_total, 8));
    HIPCHK(hipMalloc((void **)&e->d_err, 16));       /* [0] error flags, [1] hops k_hop had to search itself */
    crc_host_init();
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(c_crc), &h_crc, sizeof h_crc));
    /* the fixed scratch array of the allocation-free checksum call: every GPU has its own copy at its own address */
    if (device >= 0 && device < ZNA_MAX_DEVICES && !g_na_scratch[device]) {
        uint32_t *p = nullptr;
        HIPCHK(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
        g_na_scratch[device] = p;
    }
    e->cap_tm = 512; e->tm = (zamd_engine::Tm *)calloc((size_t)e->cap_tm, sizeof(zamd_engine::Tm));
    for (int i = 0; i < e->cap_tm; i++) { HIPCHK(hipEventCreate(&e->tm[i].a)); HIPCHK(hipEventCreate(&e->tm[i].b)); }
    *out = e;
    return ZAMD_OK;
}

static void zi_free(zamd_engine *e);
extern "C" void zamd_engine_destroy(zamd_engine *e)
{
    if (!e) return;
    hipSetDevice(e->device);
    hipDeviceSynchronize();
    zi_free(e);
    for (int i = 0; i < ZE_STREAMS; i++) { ws_free(e->ws[i]); hipStreamDestroy(e->st[i]); hipEventDestroy(e->ev_scan[i]); }
    for (int i = 0; i < 2; i++) hipEventDestroy(e->ev_call[i]);
    hipStreamDestroy(e->st_up);
    hipHostFree(e->h_res);
    for (int d = 0; d < 2; d++) for (int h = 0; h < 2; h++) if (e->h_stage[d][h]) hipHostFree(e->h_stage[d][h]);
    hipEventDestroy(e->ev_init);
    for (int i = 0; i < e->cap_tm; i++) { hipEventDestroy(e->tm[i].a); hipEventDestroy(e->tm[i].b); }
    free(e->tm);
    hipFree(e->d_total); hipFree(e->d_err); hipFree(e->d_off_own); hipFree(e->d_ck_a); hipFree(e->d_ck_c);
    delete e;
}

static uint32_t max_blocks_for(uint32_t chunk_size, int mem_level)
{
    uint32_t lim = (1u << (mem_level + 6)) - 1;
    return (chunk_size + lim - 1) / lim + 2;
}

extern "C" int zamd_engine_tune(zamd_engine *e, int good_length, int max_lazy, int nice_length, int max_chain)
{
    if (!e) return ZAMD_STREAM_ERROR;
    if (good_length == 0 && max_lazy == 0 && nice_length == 0 && max_chain == 0) { e->tuned = 0; return ZAMD_OK; }
    /* a chain budget of 0 makes the reference's loop counter wrap; nothing sensible to be exact against */
    if (good_length < 0 || max_lazy < 0 || nice_length < 0 || max_chain < 1 || max_chain > 65535 || max_lazy > 65535 || nice_length > 65535 || good_length > 65535)
        return ZAMD_STREAM_ERROR;
    e->tune[0] = (uint32_t)good_length; e->tune[1] = (uint32_t)max_lazy; e->tune[2] = (uint32_t)nice_length; e->tune[3] = (uint32_t)max_chain;
    e->tuned = 1;
    return ZAMD_OK;
}

extern "C" int zamd_engine_set_streams(zamd_engine *e, int nstreams, int serialize)
{
    if (!e || nstreams < 1 || nstreams > ZE_STREAMS) return ZAMD_STREAM_ERROR;
    e->nstreams = nstreams; e->serialize = serialize ? 1 : 0;
    return ZAMD_OK;
}

extern "C" int zamd_engine_streams(zamd_engine *e) { return e ? e->nstreams : 0; }
extern "C" int zamd_engine_serialized(zamd_engine *e) { return e ? e->serialize : 0; }

extern "C" int zamd_engine_window_bits(zamd_engine *e, int window_bits)
{
    if (!e || window_bits < 9 || window_bits > 15) return ZAMD_STREAM_ERROR;
    e->wbits = window_bits;
    return ZAMD_OK;
}

extern "C" int zamd_engine_tail(zamd_engine *e, int mode)
{
    if (!e || mode < 0 || mode > 2) return ZAMD_STREAM_ERROR;
    e->tail_mode = (uint32_t)mode;
    return ZAMD_OK;
}

extern "C" int zamd_engine_splits(zamd_engine *e, uint64_t *d_splits, uint32_t parts)
{
    if (!e || (d_splits && (parts < 2 || parts > ZAMD_MAX_PARTS))) return ZAMD_STREAM_ERROR;
    e->split_out = d_splits; e->split_parts = d_splits ? parts : 0u;
    return ZAMD_OK;
}

extern "C" int zamd_engine_pending_bits(zamd_engine *e, int *bits, int *value)
{
    if (!e) return ZAMD_STREAM_ERROR;
    if (bits) *bits = (int)e->pend_bits;
    if (value) *value = (int)e->pend_val;
    return ZAMD_OK;
}

extern "C" int zamd_engine_data_type(zamd_engine *e) { return e ? e->data_type : 2; }

extern "C" int zamd_engine_prime(zamd_engine *e, int bits, int value)
{
    if (!e || bits < 0 || bits > 7) return ZAMD_STREAM_ERROR;
    e->prime_bits = (uint32_t)bits; e->prime_val = (uint32_t)value & ((1u << bits) - 1u);
    return ZAMD_OK;
}

extern "C" uint64_t zamd_deflate_bound(uint64_t n, uint32_t chunk_size, int mem_level)
{
    if (chunk_size == 0 || mem_level < 1 || mem_level > 9) return 0;
    uint64_t chunks = n ? (n + chunk_size - 1) / chunk_size : 1;
    /* worst case: every block stored (5 bytes each, and never larger than the
     * coded form + 4) plus the sync marker / final padding per chunk */
    /* level 0 with a small window cuts a stored block every MAX_DIST bytes (deflate.c:1607-1609): 5 bytes per 250 at windowBits 9 */
    uint64_t b = n + n / 50 + chunks * ((uint64_t)max_blocks_for(chunk_size, mem_level) * 5 + 16) + 64;
    return (b + 15) & ~15ull;
}

static int fill_params(ZcParams &P, int level, int strategy, int mem_level, uint32_t chunk_size, int wbits = 15)
{
    static const uint16_t cfg[10][4] = {{0, 0, 0, 0}, {4, 4, 8, 4}, {4, 5, 16, 8}, {4, 6, 32, 32}, {4, 4, 16, 16},
                                        {8, 16, 32, 32}, {8, 16, 128, 128}, {8, 32, 128, 256}, {32, 128, 258, 1024}, {32, 258, 258, 4096}};
    if (level == -1) level = 6;
    if (level < 0 || level > 9 || strategy < 0 || strategy > 4 || mem_level < 1 || mem_level > 9) return ZAMD_STREAM_ERROR;
    /* level 0 with Z_HUFFMAN_ONLY / Z_RLE pushes blocks beyond the reference's own pending buffer (trees.c:1224) */
    if (level == 0 && (strategy == ZC_STRAT_HUFF || strategy == ZC_STRAT_RLE)) return ZAMD_STREAM_ERROR;
    if (chunk_size < 32768 || chunk_size % 16384 != 0 || chunk_size > (64u << 20)) return ZAMD_STREAM_ERROR;
    memset(&P, 0, sizeof P);
    P.level = (uint32_t)level; P.strategy = (uint32_t)strategy; P.mem_level = (uint32_t)mem_level;
    P.good = cfg[level][0]; P.lazy = cfg[level][1]; P.nice = cfg[level][2]; P.chain = cfg[level][3];
    P.hash_bits = (uint32_t)mem_level + 7; P.hash_mask = (1u << P.hash_bits) - 1; P.hash_shift = (P.hash_bits + 2) / 3;
    P.lit_bufsize = 1u << (mem_level + 6);
    P.parser = strategy == ZC_STRAT_HUFF ? ZC_PARSE_HUFF : strategy == ZC_STRAT_RLE ? ZC_PARSE_RLE : level == 0 ? ZC_PARSE_STORED : level <= 3 ? ZC_PARSE_FAST : ZC_PARSE_SLOW;
    P.chunk_size = chunk_size;
    P.max_blocks = max_blocks_for(chunk_size, mem_level);
    P.wsize = 1u << wbits; P.max_dist = P.wsize - ZC_MIN_LOOK;
    if (level == 0) {       /* deflate_stored ends a block at max_block bytes or as soon as it spans MAX_DIST (deflate.c:1597-1609) */
        uint32_t mb = P.lit_bufsize * 4u - 5u; if (mb > 0xffffu) mb = 0xffffu; if (mb > P.max_dist) mb = P.max_dist;
        const uint32_t nb = chunk_size / mb + 4u;
        if (nb > P.max_blocks) P.max_blocks = nb;
    }
    return ZAMD_OK;
}

__global__ void k_fast(ZkJob J);
__global__ void k_fastw(ZkJob J);
__global__ void k_fastw8(ZkJob J);
__global__ void k_fastr(ZkJob J);
#define ZFR_WAYS 32u
#define ZF_ROW (65536u + 32768u)
#define ZFW_WAYS 4u
static void tm_begin(zamd_engine *e, hipStream_t s, int which, int *slot)
{
    *slot = -1;
    if (e->n_tm >= e->cap_tm) return;
    *slot = e->n_tm++;
    e->tm[*slot].which = which;
    hipEventRecord(e->tm[*slot].a, s);
}
static void tm_end(zamd_engine *e, hipStream_t s, int slot) { if (slot >= 0) hipEventRecord(e->tm[slot].b, s); }

extern "C" int zamd_deflate_shard_dev(zamd_engine *e, const void *d_in, uint64_t n, uint32_t head_dict_len, uint32_t chunk_size,
                                      int level, int strategy, int mem_level, int finish,
                                      void *d_out, uint64_t out_cap, uint64_t *d_chunk_off, uint64_t *total_out)
{
    if (!e || !d_out || (!d_in && n)) return ZAMD_STREAM_ERROR;
    if (((uintptr_t)d_in & 15u) || ((uintptr_t)d_out & 15u)) return ZAMD_STREAM_ERROR;
    ZcParams P;
    int rc = fill_params(P, level, strategy, mem_level, chunk_size, e->wbits);
    if (rc) return rc;
    if (head_dict_len > P.wsize) return ZAMD_STREAM_ERROR;
    if (e->tuned && level != 0) {
        P.good = e->tune[0]; P.lazy = e->tune[1]; P.nice = e->tune[2]; P.chain = e->tune[3];
        /* deflate_fast always searches with prev_length 2, so a good_length of 2 or less quarters every walk's budget
         * (deflate.c:1185-1187); a budget of zero wraps in the reference and is as good as unbounded */
        if (P.parser == ZC_PARSE_FAST && P.good <= ZC_MIN_MATCH - 1) { P.chain >>= 2; if (P.chain == 0) return ZAMD_STREAM_ERROR; }
        /* levels 1-3 compare the candidates of a walk one per lane */
        if (P.parser == ZC_PARSE_FAST && P.chain > 64) return ZAMD_STREAM_ERROR;
    }
    out_cap &= ~15ull;
    /* the call's arguments are in order: from here on it consumes the one-shot settings (tail, primed bits), both at once */
    P.tail = e->tail_mode; e->tail_mode = ZC_TAIL_SYNC;
    const uint32_t prime_bits = e->prime_bits, prime_val = e->prime_val;
    e->prime_bits = 0; e->prime_val = 0;
    uint64_t *const split_out = e->split_out; const uint32_t split_parts = e->split_parts;
    e->split_out = nullptr; e->split_parts = 0;
    e->pend_bits = 0; e->pend_val = 0; e->data_type = 2;
    const auto host_t0 = std::chrono::steady_clock::now();
    HIPCHK(hipSetDevice(e->device));
    const uint64_t total_chunks = n ? (n + chunk_size - 1) / chunk_size : 1;
    /* batch size from the workspace budget, two batches in flight */
    /* levels 1-3 keep their own tables per chunk: a bucket of the newest strings per hash value when the chain budget
     * fits one (k_fastw), else head[] and prev[] (k_fast) */
    static const bool fast_links = getenv("ZAMD_FAST_LINKS") != nullptr;
    const bool fast_ways = P.parser == ZC_PARSE_FAST && P.chain <= ZFW_WAYS && !fast_links;
    const bool fast_ring = P.parser == ZC_PARSE_FAST && !fast_ways && P.chain <= ZFR_WAYS && !fast_links;      /* k_fastr: a ring of 32 strings and a count per hash value */
    const uint64_t fast_row = P.parser != ZC_PARSE_FAST ? 0ull : fast_ways ? (uint64_t)(P.hash_mask + 1u) * 16u :
                              fast_ring ? (uint64_t)(P.hash_mask + 1u) * (ZFR_WAYS + 1u) * 4u : (uint64_t)ZF_ROW * 2u;
    const bool with_match = P.parser == ZC_PARSE_SLOW || P.parser == ZC_PARSE_RLE;
    uint64_t per = ws_bytes_per_chunk(chunk_size, P.max_blocks, with_match) + fast_row;
    const int NS = getenv("ZAMD_SERIAL") ? 1 : e->nstreams;
    uint64_t bc = e->ws_budget / (uint64_t)NS / per;
    if (bc < 1) bc = 1;
    if (bc > total_chunks) bc = total_chunks;
    /* fewer batches than streams: cut the work into one batch per stream so that the serial stages of one
     * batch (k_hop, k_plan) run beside the parallel stages of another; not worth it for a handful of chunks */
    static const uint64_t min_split = getenv("ZAMD_MIN_SPLIT") ? strtoull(getenv("ZAMD_MIN_SPLIT"), nullptr, 10) : 64;
    if (total_chunks >= min_split && total_chunks < (uint64_t)NS * bc) bc = (total_chunks + NS - 1) / NS;
    if (bc > 0x7fffffu) bc = 0x7fffffu;
    const uint32_t batch = (uint32_t)bc;
    const uint32_t nbatches = (uint32_t)((total_chunks + batch - 1) / batch);
    for (int i = 0; i < NS && (uint32_t)i < nbatches; i++) { rc = ws_ensure(e->ws[i], batch, chunk_size, P.max_blocks, with_match); if (rc) return rc; }
    uint64_t *d_off = d_chunk_off;
    if (!d_off) {
        if (e->d_off_cap < total_chunks + 1) { hipFree(e->d_off_own); e->d_off_own = nullptr; e->d_off_cap = 0;
            HIPCHK(hipMalloc((void **)&e->d_off_own, (total_chunks + 1) * 8)); e->d_off_cap = total_chunks + 1; }
        d_off = e->d_off_own;
    }
    e->n_tm = 0; e->have_last = false;
    hipStream_t s0 = e->st[0];
    HIPCHK(hipEventRecord(e->ev_call[0], s0));
    HIPCHK(hipMemsetAsync(e->d_total, 0, 8, s0));
    HIPCHK(hipMemsetAsync(e->d_err, 0, 16, s0));
    HIPCHK(hipMemsetAsync(d_out, 0, out_cap, s0));
    HIPCHK(hipEventRecord(e->ev_init, s0));
    HIPCHK(hipEventRecord(e->ev_scan[NS - 1], s0));     /* batch 0's predecessor "scan" */
    const uint32_t tiles = (chunk_size + ZK_TILE - 1) / ZK_TILE;
    for (uint32_t bi = 0; bi < nbatches; bi++) {
        const int si = (int)(bi % (uint32_t)NS);                /* workspace and scan-event slot */
        hipStream_t s = e->st[e->serialize ? 0 : si];
        ZeWorkspace &w = e->ws[si];
        ZkJob J; memset(&J, 0, sizeof J);
        J.in = (const uint8_t *)d_in - head_dict_len; J.n = n + head_dict_len; J.head = head_dict_len; J.chunk0 = (uint64_t)bi * batch;
        J.nchunks = (uint32_t)((total_chunks - J.chunk0) < batch ? (total_chunks - J.chunk0) : batch);
        J.finish = finish; J.P = P; J.prime_bits = prime_bits; J.prime_val = prime_val; J.last_chunk = total_chunks - 1;
        ZcChunkGeom g0 = zc_geom(J.n, chunk_size, J.chunk0, finish, J.head, P.wsize);
        J.pdist = w.pdist; J.pdist_base = g0.dstart; J.pdist_len = w.pdist_len;
        J.rec = w.rec; J.rec_base = head_dict_len + J.chunk0 * chunk_size;
        J.sym = w.sym; J.hop = w.hop; J.nhop = w.nhop; J.arr = (uint32_t *)w.rec; J.blk = w.blk; J.tab = w.tab; J.nblk = w.nblk; J.nsym = w.nsym;
        J.chunk_bytes = w.chunk_bytes; J.sync_bit = w.sync_bit;
        J.chunk_off = d_off; J.running_total = e->d_total;
        J.out32 = (uint32_t *)d_out; J.out_cap = out_cap; J.err = e->d_err;
        J.splits = split_out; J.split_parts = split_parts;
        int slot;
        if (bi >= 1 && bi < (uint32_t)NS) HIPCHK(hipStreamWaitEvent(s, e->ev_init, 0));   /* memsets on stream 0 */
        if (P.parser == ZC_PARSE_SLOW) {
            tm_begin(e, s, 0, &slot);
            hipLaunchKernelGGL(k_chain, dim3(J.nchunks), dim3(ZK_CHAIN_THREADS), 0, s, J, (int)ZK_CHAIN_REF);
            tm_end(e, s, slot);
        }
        /* k_match's records reproduce neither a nice length below the lazy limit nor a quarter budget of zero
         * (which wraps in the reference): those searches follow the parse (k_follow) */
        const bool dense = P.parser == ZC_PARSE_RLE || (P.parser == ZC_PARSE_SLOW && !(e->spec_match || P.nice < P.lazy || P.chain < 4));
        if (P.parser == ZC_PARSE_SLOW && !dense) {
            tm_begin(e, s, 1, &slot);
            HIPCHK(hipMemsetAsync(w.hop, 0, (uint64_t)J.nchunks * chunk_size * 4u, s));
            hipLaunchKernelGGL(k_follow, dim3(J.nchunks), dim3(ZF_THREADS), 0, s, J);
            tm_end(e, s, slot);
        }
        if (dense) {
            const int rle = P.parser == ZC_PARSE_RLE ? 1 : 0;
            /* Chain budgets from 256 on (levels 7 to 9) search along the four-byte chains (k_match4): three more passes over
             * the input set that up (how far the budget reaches, the nearest three-byte match, the chains themselves), which
             * costs what about 60 candidate steps per position cost k_match, and saves what the hash chain's members of
             * length 3 cost — most of a budget of 1 024 or 4 096, about as much as the set-up costs of one of 128 (measured:
             * level 6 comes out the same either way, level 7 a sixth faster, levels 8 and 9 half as fast again and more).
             * ZAMD_MATCH4_MIN_CHAIN moves the border (tests, measurements). */
            static const uint32_t four_from = getenv("ZAMD_MATCH4_MIN_CHAIN") ? (uint32_t)strtoul(getenv("ZAMD_MATCH4_MIN_CHAIN"), nullptr, 10) : 256u;
            const bool by_four = !rle && P.chain >= four_from && P.nice >= 4u;
            if (by_four) {
                /* per position: how far the chain budget reaches (from the hash chains), the nearest three-byte match (from
                 * the exact three-byte chains, built in the hash chains' place), then the four-byte chains in theirs */
                const uint32_t rtiles32 = (chunk_size + ZK_REACH_TILE - 1) / ZK_REACH_TILE;
                tm_begin(e, s, 10, &slot);
                hipLaunchKernelGGL(k_reach, dim3(J.nchunks * rtiles32), dim3(ZK_MATCH_THREADS), 0, s, J, rtiles32);
                tm_end(e, s, slot);
                tm_begin(e, s, 0, &slot);
                hipLaunchKernelGGL(k_chain, dim3(J.nchunks), dim3(ZK_CHAIN_THREADS), 0, s, J, (int)ZK_CHAIN_THREE);
                tm_end(e, s, slot);
                tm_begin(e, s, 11, &slot);
                hipLaunchKernelGGL(k_first3, dim3(J.nchunks * tiles), dim3(ZK_MATCH_THREADS), 0, s, J, tiles);
                tm_end(e, s, slot);
                tm_begin(e, s, 0, &slot);
                hipLaunchKernelGGL(k_chain, dim3(J.nchunks), dim3(ZK_CHAIN_THREADS), 0, s, J, (int)ZK_CHAIN_FOUR);
                tm_end(e, s, slot);
            }
            tm_begin(e, s, 1, &slot);
            if (by_four) hipLaunchKernelGGL(k_match4, dim3(J.nchunks * tiles), dim3(ZK_MATCH_THREADS), 0, s, J, tiles);
            else hipLaunchKernelGGL(k_match, dim3(J.nchunks * tiles), dim3(ZK_MATCH_THREADS), 0, s, J, tiles, rle);
            tm_end(e, s, slot);
        }
        tm_begin(e, s, 2, &slot);
        if (P.parser == ZC_PARSE_FAST) {
            const bool ways = fast_ways;
            const uint64_t row = fast_row;
            if (w.fhead_bytes < batch * row) {
                hipFree(w.fhead); w.fhead = nullptr; w.fhead_bytes = 0;
                HIPCHK(hipMalloc((void **)&w.fhead, batch * row)); w.fhead_bytes = batch * row;
            }
            J.fhead = w.fhead;
            HIPCHK(hipMemsetAsync(w.fhead, 0, J.nchunks * row, s));
            /* all the call's batches are on the GPU together: with more than 24 of their wavefronts to a CU, the build that fits eight to a SIMD */
            static const char *force8 = getenv("ZAMD_FASTW8");      /* tests: "1" = always the eight-wavefront build, "0" = never */
            const bool eight = force8 ? force8[0] == '1' : total_chunks > 24ull * (uint64_t)e->cu_count;
            if (ways && eight) hipLaunchKernelGGL(k_fastw8, dim3(J.nchunks), dim3(64), 0, s, J);
            else if (ways) hipLaunchKernelGGL(k_fastw, dim3(J.nchunks), dim3(64), 0, s, J);
            else if (fast_ring) hipLaunchKernelGGL(k_fastr, dim3(J.nchunks), dim3(64), 0, s, J);
            else hipLaunchKernelGGL(k_fast, dim3(J.nchunks), dim3(64), 0, s, J);
        }
        else if ((P.parser == ZC_PARSE_SLOW || P.parser == ZC_PARSE_RLE) && !(dense && getenv("ZAMD_SERIAL_PARSE"))) {
            const uint32_t rtiles = (chunk_size + ZK_RESOLVE_TILE * ZK_RESOLVE_SPAN - 1) / (ZK_RESOLVE_TILE * ZK_RESOLVE_SPAN);
            if (dense) hipLaunchKernelGGL(k_resolve, dim3(J.nchunks * rtiles), dim3(ZK_RESOLVE_TILE), 0, s, J, rtiles);
            if (dense) hipLaunchKernelGGL(k_hopw, dim3(J.nchunks), dim3(64), 0, s, J);
            else hipLaunchKernelGGL(k_hop<true>, dim3(J.nchunks), dim3(64), 0, s, J);
            hipLaunchKernelGGL(k_expand, dim3(J.nchunks), dim3(ZK_EXPAND_THREADS), 0, s, J);
        } else hipLaunchKernelGGL(k_parse, dim3(J.nchunks), dim3(64), 0, s, J);
        tm_end(e, s, slot);
        tm_begin(e, s, 3, &slot);
        hipLaunchKernelGGL(k_plan, dim3(P.max_blocks, J.nchunks), dim3(ZK_PLAN_THREADS), 0, s, J);
        tm_end(e, s, slot);
        tm_begin(e, s, 4, &slot);
        hipLaunchKernelGGL(k_layout, dim3((J.nchunks + 255) / 256), dim3(256), 0, s, J);
        /* output offsets chain from batch to batch through d_total */
        HIPCHK(hipStreamWaitEvent(s, e->ev_scan[(si + NS - 1) % NS], 0));
        hipLaunchKernelGGL(k_scan, dim3(1), dim3(1024), 0, s, J);
        HIPCHK(hipEventRecord(e->ev_scan[si], s));
        tm_end(e, s, slot);
        tm_begin(e, s, 5, &slot);
        hipLaunchKernelGGL(k_emit, dim3(P.max_blocks + 1, J.nchunks), dim3(ZK_EMIT_THREADS), 0, s, J);
        tm_end(e, s, slot);
        if (bi == 0) { e->last_job = J; e->have_last = true; }
    }
    HIPCHK(hipGetLastError());
    const auto host_t1 = std::chrono::steady_clock::now();
    if (e->on_launched) e->on_launched(e->on_launched_arg);
    for (int i = 1; i < NS && (uint32_t)i < nbatches; i++) HIPCHK(hipStreamSynchronize(e->st[i]));
    HIPCHK(hipEventRecord(e->ev_call[1], s0));
    /* the call's results come back through pinned memory on the call's own stream: a blocking hipMemcpy would queue
     * behind whatever large host copy another thread has in flight */
    HIPCHK(hipMemcpyAsync(e->h_res, e->d_total, 8, hipMemcpyDeviceToHost, s0));
    HIPCHK(hipMemcpyAsync(e->h_res + 2, e->d_err, 16, hipMemcpyDeviceToHost, s0));
    HIPCHK(hipStreamSynchronize(s0));
    uint64_t total = 0; uint32_t ev[4] = {0, 0, 0, 0};
    memcpy(&total, e->h_res, 8); memcpy(ev, e->h_res + 2, 16);
    const uint32_t err = ev[0];
    e->last_fallback_hops = ev[1];
    e->data_type = ev[2] ? (int)ev[2] - 1 : 2;
    if (!finish && P.tail != ZC_TAIL_SYNC && ev[3] && !(err & ~0u)) {
        /* the bits behind the last complete byte wait for the next call, as they would in bi_buf */
        HIPCHK(hipMemcpyAsync(e->h_res + 8, (const uint8_t *)d_out + total, 1, hipMemcpyDeviceToHost, s0));
        HIPCHK(hipStreamSynchronize(s0));
        const uint8_t last = (uint8_t)(e->h_res[8] & 0xffu);
        e->pend_bits = ev[3]; e->pend_val = last & ((1u << ev[3]) - 1u);
    }
    for (int k = 0; k < 12; k++) { e->t_ms[k] = 0; e->t_n[k] = 0; }
    for (int i = 0; i < e->n_tm; i++) { float ms = 0; if (hipEventElapsedTime(&ms, e->tm[i].a, e->tm[i].b) == hipSuccess) { e->t_ms[e->tm[i].which] += ms; e->t_n[e->tm[i].which]++; } }
    { float ms = 0; if (hipEventElapsedTime(&ms, e->ev_call[0], e->ev_call[1]) == hipSuccess) { e->t_ms[6] = ms; e->t_n[6] = 1; } }
    /* the host's side of the call: until the last kernel was queued, and from there to the results */
    e->t_ms[7] = std::chrono::duration<double, std::milli>(host_t1 - host_t0).count(); e->t_n[7] = 1;
    e->t_ms[8] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - host_t1).count(); e->t_n[8] = 1;
    if (total_out) *total_out = total;
    if (err & ZE_OUT_OVERFLOW) return ZAMD_BUF_ERROR;
    if (err) { fprintf(stderr, "zamd: device reported internal error flags 0x%x\n", err); return ZAMD_STREAM_ERROR; }
    return ZAMD_OK;
}

extern "C" int zamd_deflate_dev(zamd_engine *e, const void *d_in, uint64_t n, uint32_t chunk_size,
                                int level, int strategy, int mem_level, int finish,
                                void *d_out, uint64_t out_cap, uint64_t *d_chunk_off, uint64_t *total_out)
{
    return zamd_deflate_shard_dev(e, d_in, n, 0, chunk_size, level, strategy, mem_level, finish, d_out, out_cap, d_chunk_off, total_out);
}

extern "C" double zamd_last_timing_ms(zamd_engine *e, int which, int *launches)
{
    if (!e || which < 0 || which > 11) return 0;
    if (which == 9) { if (launches) *launches = 0; return (double)e->last_fallback_hops; }   /* a count, not a time */
    if (launches) *launches = e->t_n[which];
    return e->t_ms[which];
}

extern "C" int64_t zamd_debug_dump(zamd_engine *e, int what, uint32_t chunk, void *host_buf, uint64_t max_elems)
{
    if (!e || !e->have_last || chunk >= e->last_job.nchunks) return ZAMD_STREAM_ERROR;
    const ZkJob &J = e->last_job;
    ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + chunk, J.finish, J.head, J.P.wsize);
    uint64_t cnt = 0; const void *src = nullptr; size_t esz = 0;
    uint32_t nb = 0, ns = 0;
    hipMemcpy(&nb, J.nblk + chunk, 4, hipMemcpyDeviceToHost);
    hipMemcpy(&ns, J.nsym + chunk, 4, hipMemcpyDeviceToHost);
    switch (what) {
    case 0: src = J.pdist + (g.dstart + g.dict_len - J.pdist_base); cnt = g.len; esz = 2; break;
    case 1: src = J.rec + (g.dstart + g.dict_len - J.rec_base); cnt = g.len; esz = 8; break;
    case 2: src = J.sym + (uint64_t)chunk * J.P.chunk_size; cnt = ns; esz = 4; break;
    case 3: src = J.blk + (uint64_t)chunk * J.P.max_blocks; cnt = nb; esz = sizeof(ZcBlock); break;
    default: return ZAMD_STREAM_ERROR;
    }
    if (cnt > max_elems) cnt = max_elems;
    if (cnt && hipMemcpy(host_buf, src, cnt * esz, hipMemcpyDeviceToHost) != hipSuccess) return ZAMD_HIP_ERROR(1);
    return (int64_t)cnt;
}

/* ------------------------------------------------------------------ checksums (host entry points) */
extern "C" uint32_t zamd_adler32_combine(uint32_t a1, uint32_t a2, int64_t len2)
{   /* adler32.c:136-162 */
    const uint32_t B = 65521u;
    if (len2 < 0) return 0xffffffffu;
    uint32_t rem = (uint32_t)(len2 % B), s1 = a1 & 0xffffu, s2 = (uint32_t)(((uint64_t)rem * s1) % B);
    s1 += (a2 & 0xffffu) + B - 1;
    s2 += (a1 >> 16) + (a2 >> 16) + B - rem;
    if (s1 >= B) s1 -= B;
    if (s1 >= B) s1 -= B;
    if (s2 >= (B << 1)) s2 -= (B << 1);
    if (s2 >= B) s2 -= B;
    return s1 | (s2 << 16);
}

extern "C" uint32_t zamd_crc32_combine(uint32_t c1, uint32_t c2, int64_t len2)
{   /* crc32.c:327-408: advance c1 over len2 zero bytes, i.e. multiply by x^(8 len2) */
    if (len2 <= 0) return c1;
    crc_host_init();
    return gf2_mul(gf2_xpow8n((uint64_t)len2, h_crc.x2n), c1) ^ c2;
}

extern "C" int zamd_checksums_dev(zamd_engine *e, const void *d_in, uint64_t n, uint32_t chunk_size,
                                  uint32_t *d_adler, uint32_t *d_crc)
{
    if (!e || (!d_in && n) || chunk_size == 0) return ZAMD_STREAM_ERROR;
    if (n == 0) return ZAMD_OK;
    HIPCHK(hipSetDevice(e->device));
    uint64_t chunks = (n + chunk_size - 1) / chunk_size;
    if (chunks > 0x7fffffffull) return ZAMD_STREAM_ERROR;
    hipLaunchKernelGGL(k_checksum, dim3((uint32_t)chunks), dim3(256), 0, e->st[0], (const uint8_t *)d_in, n, chunk_size, d_adler, d_crc);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->st[0]));
    return ZAMD_OK;
}

static int whole_checksum(zamd_engine *e, const void *d_in, uint64_t n, int want_crc, uint32_t seed, uint32_t *result)
{
    if (!e || !result) return ZAMD_STREAM_ERROR;
    if (n == 0) { *result = seed; return ZAMD_OK; }
    const uint32_t slice = 1u << 20;
    uint64_t chunks = (n + slice - 1) / slice;
    if (e->ck_cap < chunks) {
        hipFree(e->d_ck_a); hipFree(e->d_ck_c); e->d_ck_a = e->d_ck_c = nullptr; e->ck_cap = 0;
        HIPCHK(hipMalloc((void **)&e->d_ck_a, chunks * 4)); HIPCHK(hipMalloc((void **)&e->d_ck_c, chunks * 4)); e->ck_cap = chunks;
    }
    int rc = zamd_checksums_dev(e, d_in, n, slice, want_crc ? nullptr : e->d_ck_a, want_crc ? e->d_ck_c : nullptr);
    if (rc) return rc;
    uint32_t *h = (uint32_t *)malloc(chunks * 4);
    if (!h) return ZAMD_MEM_ERROR;
    HIPCHK(hipMemcpy(h, want_crc ? e->d_ck_c : e->d_ck_a, chunks * 4, hipMemcpyDeviceToHost));
    uint32_t acc = seed;
    for (uint64_t i = 0; i < chunks; i++) {
        int64_t len = (int64_t)((n - i * slice) < slice ? (n - i * slice) : slice);
        acc = want_crc ? zamd_crc32_combine(acc, h[i], len) : zamd_adler32_combine(acc, h[i], len);
    }
    free(h);
    *result = acc;
    return ZAMD_OK;
}

extern "C" int zamd_checksums_noalloc_dev(zamd_engine *e, uint32_t adler, uint32_t crc, const void *d_in, uint64_t n, uint32_t *d_result)
{
    if (!e || !d_result || (!d_in && n)) return ZAMD_STREAM_ERROR;
    const uint64_t nslices = (n + ZNA_SLICE - 1) / ZNA_SLICE;
    if (nslices > ZNA_MAX_SLICES) {
        fprintf(stderr, "zamd_checksums_noalloc_dev: %llu bytes exceed the limit of %llu bytes of the fixed scratch array\n",
                (unsigned long long)n, (unsigned long long)ZNA_MAX_SLICES * ZNA_SLICE);
        return ZAMD_BUF_ERROR;
    }
    if (e->device < 0 || e->device >= ZNA_MAX_DEVICES || !g_na_scratch[e->device]) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device));
    uint32_t *part = g_na_scratch[e->device];       /* the device address, never the host-side name of the array */
    hipStream_t s = e->st[0];
    if (nslices)
        hipLaunchKernelGGL(k_checksum, dim3((uint32_t)nslices), dim3(256), 0, s, (const uint8_t *)d_in, n, ZNA_SLICE, part, part + ZNA_MAX_SLICES);
    hipLaunchKernelGGL(k_ck_fold, dim3(1), dim3(1024), 0, s, part, part + ZNA_MAX_SLICES, (uint32_t)nslices, n, ZNA_SLICE, adler, crc, d_result);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    return ZAMD_OK;
}

extern "C" int zamd_adler32_dev(zamd_engine *e, uint32_t adler, const void *d_in, uint64_t n, uint32_t *result)
{ return whole_checksum(e, d_in, n, 0, adler, result); }
extern "C" int zamd_crc32_dev(zamd_engine *e, uint32_t crc, const void *d_in, uint64_t n, uint32_t *result)
{ return whole_checksum(e, d_in, n, 1, crc, result); }

#include "zamd_inflate_impl.h"
#include "zamd_fast_impl.h"

/* ------------------------------------------------------------------ device-memory helpers */
/* Exclusive scan of n chunk lengths on top of `base`: off[i] = base + len[0] + ... + len[i-1], off[n] = the total. */
__global__ __launch_bounds__(1024) void k_offsets(const uint64_t *len, uint64_t n, uint64_t base, uint64_t *off)
{
    __shared__ uint64_t part[1024];
    const uint32_t tid = threadIdx.x;
    const uint64_t per = (n + 1023) / 1024;
    uint64_t sum = 0;
    for (uint64_t k = 0; k < per; k++) { const uint64_t i = tid * per + k; if (i < n) sum += len[i]; }
    part[tid] = sum;
    __syncthreads();
    if (tid == 0) { uint64_t run = base; for (uint32_t i = 0; i < 1024; i++) { const uint64_t v = part[i]; part[i] = run; run += v; } off[n] = run; }
    __syncthreads();
    uint64_t run = part[tid];
    for (uint64_t k = 0; k < per; k++) { const uint64_t i = tid * per + k; if (i < n) { off[i] = run; run += len[i]; } }
}

extern "C" int zamd_chunk_offsets_dev(zamd_engine *e, const uint64_t *d_lengths, uint64_t n, uint64_t base, uint64_t *d_offsets)
{
    if (!e || !d_lengths || !d_offsets || d_lengths == d_offsets) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_offsets, dim3(1), dim3(1024), 0, e->st[0], d_lengths, n, base, d_offsets);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->st[0]));
    return ZAMD_OK;
}

/* lengths[i] = offsets[i + 1] - offsets[i] for i < n, 0 for n <= i < pad_to */
__global__ __launch_bounds__(256) void k_lengths(const uint64_t *off, uint64_t n, uint64_t pad_to, uint64_t *len)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < pad_to) len[i] = i < n ? off[i + 1] - off[i] : 0ull;
}

extern "C" int zamd_chunk_lengths_dev(zamd_engine *e, const uint64_t *d_offsets, uint64_t n, uint64_t *d_lengths, uint64_t pad_to)
{
    if (!e || !d_lengths || (!d_offsets && n)) return ZAMD_STREAM_ERROR;
    if (pad_to < n) pad_to = n;
    if (pad_to == 0) return ZAMD_OK;
    HIPCHK(hipSetDevice(e->device));
    hipLaunchKernelGGL(k_lengths, dim3((uint32_t)((pad_to + 255) / 256)), dim3(256), 0, e->st[0], d_offsets, n, pad_to, d_lengths);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->st[0]));
    return ZAMD_OK;
}

extern "C" int zamd_engine_on_launched(zamd_engine *e, void (*fn)(void *), void *arg)
{
    if (!e) return ZAMD_STREAM_ERROR;
    e->on_launched = fn; e->on_launched_arg = arg;
    return ZAMD_OK;
}
extern "C" int zamd_dev_alloc(zamd_engine *e, uint64_t bytes, void **d_ptr)
{
    if (!e || !d_ptr) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device));
    hipError_t r = hipMalloc(d_ptr, bytes ? bytes : 16);
    return r == hipSuccess ? ZAMD_OK : (r == hipErrorOutOfMemory ? ZAMD_MEM_ERROR : ZAMD_HIP_ERROR(r));
}
extern "C" int zamd_dev_free(zamd_engine *e, void *d_ptr)
{
    if (!e) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device)); HIPCHK(hipFree(d_ptr)); return ZAMD_OK;
}
extern "C" int zamd_dev_upload(zamd_engine *e, void *d_dst, const void *h_src, uint64_t bytes)
{
    if (!e) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device));
    if (bytes) { HIPCHK(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, e->st_up)); HIPCHK(hipStreamSynchronize(e->st_up)); }
    return ZAMD_OK;
}
extern "C" int zamd_dev_download(zamd_engine *e, void *h_dst, const void *d_src, uint64_t bytes)
{
    if (!e) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device));
    if (bytes) { HIPCHK(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, e->st_down)); HIPCHK(hipStreamSynchronize(e->st_down)); }
    return ZAMD_OK;
}
/* The same transfers through pinned staging memory of the engine, the host's own memcpy on one side and the copy
 * engine on the other, two pieces in flight.  Beside running kernels the runtime's path for pageable memory was
 * measured four to five times slower than alone; this one keeps its pace.  One upload and one download may run at the
 * same time (from two threads), not two of a kind; both directions share one stream, in the order they were queued. */
#define ZE_STAGE_BYTES (32ull << 20)
/* The host's side of a staged transfer, several threads to a piece: one thread's memcpy runs at half of what the copy
 * engine moves (a 3.5 GiB download: 130 ms with one thread).  The helpers live as long as the transfer that made them;
 * ZAMD_COPY_THREADS (default 4) is the number of threads to a piece, the caller's own included. */
struct ZeParCopy {
    std::vector<std::thread> th;
    std::mutex mu; std::condition_variable cv_go, cv_done;
    uint64_t gen = 0; int pending = 0; bool quit = false;
    uint8_t *dst = nullptr; const uint8_t *src = nullptr; uint64_t len = 0, slice = 0;
    ZeParCopy()
    {
        static const int want = [] { const char *v = getenv("ZAMD_COPY_THREADS"); const int k = v ? atoi(v) : 4; return k < 1 ? 1 : k > 16 ? 16 : k; }();
        try {
            for (int i = 1; i < want; i++) th.emplace_back([this, i] { work(i); });
        } catch (...) { /* fewer helpers, or none: the caller copies the rest itself */ }
    }
    ~ZeParCopy()
    {
        { std::lock_guard<std::mutex> g(mu); quit = true; }
        cv_go.notify_all();
        for (std::thread &t : th) t.join();
    }
    void part(int i) const
    {
        const uint64_t a = (uint64_t)i * slice;
        if (a < len) memcpy(dst + a, src + a, len - a < slice ? len - a : slice);
    }
    void work(int i)
    {
        uint64_t seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> g(mu); cv_go.wait(g, [&] { return quit || gen != seen; }); if (quit) return; seen = gen; }
            part(i);
            { std::lock_guard<std::mutex> g(mu); pending--; }
            cv_done.notify_one();
        }
    }
    void copy(void *d, const void *s, uint64_t l)
    {
        if (th.empty() || l < (4u << 20)) { memcpy(d, s, l); return; }
        const uint64_t n = th.size() + 1;
        { std::lock_guard<std::mutex> g(mu); dst = (uint8_t *)d; src = (const uint8_t *)s; len = l; slice = ((l + n - 1) / n + 63) & ~63ull; pending = (int)th.size(); gen++; }
        cv_go.notify_all();
        part(0);
        std::unique_lock<std::mutex> g(mu);
        cv_done.wait(g, [&] { return pending == 0; });
    }
};
static int stage_ready(zamd_engine *e, int dir)
{
    for (int h = 0; h < 2; h++)
        if (!e->h_stage[dir][h]) HIPCHK(hipHostMalloc((void **)&e->h_stage[dir][h], ZE_STAGE_BYTES, hipHostMallocDefault));
    return ZAMD_OK;
}
extern "C" int zamd_dev_upload_staged(zamd_engine *e, void *d_dst, const void *h_src, uint64_t bytes)
{
    if (!e) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device));
    if (int rc = stage_ready(e, 0)) return rc;
    hipEvent_t ev[2];
    for (int h = 0; h < 2; h++) HIPCHK(hipEventCreateWithFlags(&ev[h], hipEventDisableTiming));
    int rc = ZAMD_OK;
    uint64_t piece = 0;
    ZeParCopy pc;
    for (uint64_t at = 0; at < bytes && rc == ZAMD_OK; at += ZE_STAGE_BYTES, piece++) {
        const int h = (int)(piece & 1);
        const uint64_t len = bytes - at < ZE_STAGE_BYTES ? bytes - at : ZE_STAGE_BYTES;
        if (piece >= 2 && hipEventSynchronize(ev[h]) != hipSuccess) { rc = ZAMD_HIP_ERROR(1); break; }   /* the half is free again */
        pc.copy(e->h_stage[0][h], (const uint8_t *)h_src + at, len);
        if (hipMemcpyAsync((uint8_t *)d_dst + at, e->h_stage[0][h], len, hipMemcpyHostToDevice, e->st_up) != hipSuccess ||
            hipEventRecord(ev[h], e->st_up) != hipSuccess) rc = ZAMD_HIP_ERROR(1);
    }
    if (hipStreamSynchronize(e->st_up) != hipSuccess) rc = ZAMD_HIP_ERROR(1);
    for (int h = 0; h < 2; h++) hipEventDestroy(ev[h]);
    return rc;
}
extern "C" int zamd_dev_download_staged(zamd_engine *e, void *h_dst, const void *d_src, uint64_t bytes)
{
    if (!e) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device));
    if (int rc = stage_ready(e, 1)) return rc;
    hipEvent_t ev[2];
    for (int h = 0; h < 2; h++) HIPCHK(hipEventCreateWithFlags(&ev[h], hipEventDisableTiming));
    int rc = ZAMD_OK;
    const uint64_t pieces = (bytes + ZE_STAGE_BYTES - 1) / ZE_STAGE_BYTES;
    ZeParCopy pc;
    /* piece k + 1 is on its way while the host copies piece k out */
    for (uint64_t k = 0; k <= pieces && rc == ZAMD_OK; k++) {
        if (k < pieces) {
            const uint64_t at = k * ZE_STAGE_BYTES, len = bytes - at < ZE_STAGE_BYTES ? bytes - at : ZE_STAGE_BYTES;
            if (hipMemcpyAsync(e->h_stage[1][k & 1], (const uint8_t *)d_src + at, len, hipMemcpyDeviceToHost, e->st_down) != hipSuccess ||
                hipEventRecord(ev[k & 1], e->st_down) != hipSuccess) { rc = ZAMD_HIP_ERROR(1); break; }
        }
        if (k >= 1) {
            const uint64_t at = (k - 1) * ZE_STAGE_BYTES, len = bytes - at < ZE_STAGE_BYTES ? bytes - at : ZE_STAGE_BYTES;
            if (hipEventSynchronize(ev[(k - 1) & 1]) != hipSuccess) { rc = ZAMD_HIP_ERROR(1); break; }
            pc.copy((uint8_t *)h_dst + at, e->h_stage[1][(k - 1) & 1], len);
        }
    }
    if (hipStreamSynchronize(e->st_down) != hipSuccess) rc = ZAMD_HIP_ERROR(1);
    for (int h = 0; h < 2; h++) hipEventDestroy(ev[h]);
    return rc;
}
extern "C" int zamd_dev_memset(zamd_engine *e, void *d_dst, int value, uint64_t bytes)
{
    if (!e) return ZAMD_STREAM_ERROR;
    HIPCHK(hipSetDevice(e->device)); if (bytes) HIPCHK(hipMemset(d_dst, value, bytes)); return ZAMD_OK;
}
