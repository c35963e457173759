/*
 * zamd_fast_impl.h — deflate_fast (levels 1-3, deflate.c:1628-1722) on the GPU.
 * Included by zamd_engine.hip.
 *
 * The hash chains of these levels depend on the parse itself (strings inside a
 * match longer than max_insert_length are never inserted), so a chunk is parsed
 * by ONE wavefront that owns the chunk's head[] and prev[] as 16-bit window
 * indices, exactly as the reference lays them out.  The state machine
 * (zc_parse_fast) is wave-uniform; what keeps it off the memory latency:
 *   - a lone wavefront issues at most one instruction every four cycles, so what
 *     counts is how many chunks a CU parses at once: the LDS holds only prev[] of
 *     the newest 2 Ki positions (4 KiB: a CU's 32 wavefront slots are all usable);
 *     older links, head[] and the input bytes are read from global memory;
 *   - head[] is never on the critical path: the hashes
 *     of the next 128 positions are computed by the lanes and their head entries
 *     gathered one batch ahead; an insertion patches the gathered values of later
 *     positions with the same hash in registers, so a position's head entry is a
 *     v_readlane away when the parse gets there;
 *   - the candidates of a chain walk are compared with the current string all at
 *     once, one per lane.
 */

/* Links of the newest ZF_NEAR positions are kept in LDS.  4 KiB per wavefront lets a CU hold its 32 wavefronts, and
 * chunks in flight are what this kernel lives on (measured on 8 GiB of text at level 1: 16 KiB 5.3 GB/s, 8 KiB 7.3,
 * 4 KiB 8.0, 2 KiB 7.8 — the links further back come from global memory). */
#ifndef ZF_NEAR
#define ZF_NEAR 2048u
#endif

struct ZkFastCtx {
    const uint8_t *b;             /* a-coordinate 0 */
    uint64_t readable;            /* bytes of b that may be read */
    uint32_t E;                   /* end of the chunk's data (a-coordinate) */
    uint16_t *prev_l;             /* LDS: prev[] of the newest ZF_NEAR window indices */
    uint16_t *prev_g, *fhead;     /* global: all of prev[] (32768 entries), head[] */
    uint32_t hshift, hmask, hsize, nice, chain, lane;
    uint32_t wsize, wmask, near, max_dist;    /* window size (deflate.c:284), its mask, LDS-resident links (<= wsize), wsize - MIN_LOOKAHEAD */
    uint32_t q0;                  /* first position of the current batch */
    uint32_t h_cur, hv_cur, h_nxt, hv_nxt;   /* per lane: hash and head entry of q0 + lane, q0 + 64 + lane */

    __device__ __forceinline__ uint32_t bytes4(uint32_t a) const
    {
        uint32_t w = 0;
        if ((uint64_t)a + 4u <= readable) __builtin_memcpy(&w, b + a, 4);
        else for (uint32_t i = 0; i < 4 && (uint64_t)a + i < readable; i++) w |= (uint32_t)b[a + i] << (8u * i);
        return w;
    }
    /* hashes and head entries of the 64 positions from qb on */
    __device__ __forceinline__ void load_batch(uint32_t qb, uint32_t &h, uint32_t &hv)
    {
        const uint32_t a = qb + lane;
        const bool ok = a + ZC_MIN_MATCH <= E;
        const uint32_t w = ok ? bytes4(a) : 0u;
        h = ok ? zc_hash3(w & 0xffu, (w >> 8) & 0xffu, (w >> 16) & 0xffu, hshift, hmask) : 0xffffffffu;
        hv = ok ? (uint32_t)fhead[h] : 0u;
    }
    __device__ __forceinline__ void start_batches(uint32_t q)
    {
        __threadfence_block();                 /* earlier head stores have landed */
        q0 = q;
        load_batch(q0, h_cur, hv_cur);
        load_batch(q0 + 64u, h_nxt, hv_nxt);
    }
    __device__ __forceinline__ uint32_t insert(uint32_t p, uint32_t B)
    {
        if (p - q0 >= 64u) {
            if (p - q0 < 128u) {
                h_cur = h_nxt; hv_cur = hv_nxt; q0 += 64u;
                __threadfence_block();
                load_batch(q0 + 64u, h_nxt, hv_nxt);
            } else start_batches(p);
        }
        const uint32_t k = p - q0, wi = p - B;
        const uint32_t hq = __builtin_amdgcn_readlane(h_cur, k);
        const uint32_t e = __builtin_amdgcn_readlane(hv_cur, k);
        hv_cur = (lane > k && h_cur == hq) ? wi : hv_cur;
        hv_nxt = h_nxt == hq ? wi : hv_nxt;
        if (lane == 0) { prev_l[wi & (near - 1u)] = (uint16_t)e; prev_g[wi & wmask] = (uint16_t)e; fhead[hq] = (uint16_t)wi; }
        return e;
    }
    __device__ __forceinline__ void rebase(uint16_t *t, uint32_t n)
    {
        for (uint32_t i = lane * 8u; i < n; i += 512u) {
            uint4 q = *(uint4 *)(t + i);
            uint32_t v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t lo = v[j] & 0xffffu, hi = v[j] >> 16;
                v[j] = (lo >= wsize ? lo - wsize : 0u) | ((hi >= wsize ? hi - wsize : 0u) << 16);
            }
            *(uint4 *)(t + i) = make_uint4(v[0], v[1], v[2], v[3]);
        }
    }
    __device__ __forceinline__ void slide()
    {
        __threadfence_block();
        rebase(fhead, hsize);
        rebase(prev_g, wsize);
        for (uint32_t i = lane; i < near; i += 64) { const uint32_t m = prev_l[i]; prev_l[i] = (uint16_t)(m >= wsize ? m - wsize : 0u); }
        hv_cur = hv_cur >= wsize ? hv_cur - wsize : 0u;
        hv_nxt = hv_nxt >= wsize ? hv_nxt - wsize : 0u;
        __threadfence_block();
    }
    /* longest_match (deflate.c:1148-1289) with prev_length 2: the chain is walked
     * first (at most max_chain links, every one above the distance limit), then
     * lane k measures candidate k, then the reference's rule picks: strictly
     * longer wins, stop at the first one that reaches nice_match. */
    __device__ __forceinline__ void longest(uint32_t p, uint32_t B, uint32_t cur, uint32_t look, uint32_t *mlen, uint32_t *mdist)
    {
        const uint32_t sw = p - B, limit = sw > max_dist ? sw - max_dist : 0u;
        const uint32_t cap = look < ZC_MAX_MATCH ? look : ZC_MAX_MATCH, nice_eff = nice < look ? nice : look;
        uint32_t candv = 0, n = 0;
        do {
            candv = lane == n ? cur : candv;
            n++;
            cur = sw - cur < near ? zi_uni(prev_l[cur & (near - 1u)]) : zi_uni(prev_g[cur & wmask]);
        } while (cur > limit && n < chain);
        uint32_t len = 0;
        if (lane < n) {
            const uint8_t *m = b + B + candv, *s = b + p;
            const bool wide = (uint64_t)p + cap + 16u <= readable;      /* candidates lie below p */
            while (len < cap) {
                if (wide) {
                    uint64_t a0, a1, c0, c1;
                    __builtin_memcpy(&a0, m + len, 8); __builtin_memcpy(&a1, m + len + 8, 8);
                    __builtin_memcpy(&c0, s + len, 8); __builtin_memcpy(&c1, s + len + 8, 8);
                    const uint64_t x0 = a0 ^ c0, x1 = a1 ^ c1;
                    if (x0) { len += (uint32_t)__builtin_ctzll(x0) >> 3; break; }
                    if (x1) { len += 8u + ((uint32_t)__builtin_ctzll(x1) >> 3); break; }
                    len += 16;
                } else {
                    if (m[len] != s[len]) break;
                    len++;
                }
            }
            if (len > cap) len = cap;
        }
        uint32_t best = 2, bk = 0;
        for (uint32_t k = 0; k < n; k++) {
            const uint32_t l = __builtin_amdgcn_readlane(len, k);
            if (l > best) { best = l; bk = k; if (best >= nice_eff) break; }
        }
        if (best >= ZC_MIN_MATCH) { *mlen = best; *mdist = sw - __builtin_amdgcn_readlane(candv, bk); }
    }
};

/* global tables of one chunk: head[65536] then prev[32768], 16-bit window indices */
#ifndef ZF_ROW
#define ZF_ROW (65536u + 32768u)
#endif

struct GSymRow {
    uint32_t *s; uint32_t lane; uint32_t *row;
    __device__ void operator()(uint32_t i, uint32_t w) const
    {
        if (lane == (i & 63u)) *row = w;
        if ((i & 63u) == 63u) s[(i & ~63u) + lane] = *row;
    }
};

__global__ __launch_bounds__(64) void k_fast(ZkJob J)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_prev[ZF_NEAR];
    const uint32_t lc = blockIdx.x, lane = threadIdx.x;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    ZkFastCtx cx;
    cx.b = J.in + g.dstart; cx.readable = J.n - g.dstart; cx.E = g.dict_len + g.len;
    cx.prev_l = s_prev; cx.fhead = J.fhead + (uint64_t)lc * ZF_ROW; cx.prev_g = cx.fhead + 65536u;
    cx.hshift = J.P.hash_shift; cx.hmask = J.P.hash_mask; cx.hsize = J.P.hash_mask + 1u;
    cx.nice = J.P.nice; cx.chain = J.P.chain; cx.lane = lane;
    cx.wsize = J.P.wsize; cx.wmask = J.P.wsize - 1u; cx.near = J.P.wsize < ZF_NEAR ? J.P.wsize : ZF_NEAR; cx.max_dist = J.P.max_dist;
    for (uint32_t i = lane; i < ZF_NEAR; i += 64) s_prev[i] = 0;
    __syncthreads();
    cx.start_batches(0);
    /* symbols leave in rows of 64: lane i & 63 keeps symbol i until the row is full.  A store per symbol would sit in
     * front of the parse's next load (loads and stores of a wavefront complete in order) */
    uint32_t symrow = 0;
    GSymRow sym; sym.s = J.sym + (uint64_t)lc * J.P.chunk_size; sym.lane = lane; sym.row = &symrow;
    GBlk blk; blk.b = J.blk + (uint64_t)lc * J.P.max_blocks; blk.cap = J.P.max_blocks; blk.err = J.err; blk.lane = lane;
    uint32_t nsym = 0;
    const uint32_t nb = zc_parse_fast(g.dict_len, g.len, g.final, J.P, cx, sym, blk, &nsym);
    if (lane < (nsym & 63u)) sym.s[(nsym & ~63u) + lane] = symrow;          /* the last, partial row */
    if (lane == 0) {
        J.nblk[lc] = nb < J.P.max_blocks ? nb : J.P.max_blocks;
        J.nsym[lc] = nsym;
    }
}

/* ------------------------------------------------------------------ k_fastw
 * The same parser for chain budgets of at most ZFW_WAYS candidates (level 1: max_chain 4).  A chain walk of
 * longest_match is then no walk at all: head[h] and the prev[] links behind it are the strings that were inserted
 * under hash value h, newest first (prev[c] is what head[h] held when c went in, deflate.c:186-189), so a table of
 * the ZFW_WAYS newest strings per hash value holds every candidate a search can reach.  The table keeps a-coordinates
 * (32 bits), which need no rebasing when the window slides: a string the reference has turned into NIL, or whose
 * link has been overwritten since, lies at or below the distance limit, where the walk ends anyway
 * (deflate.c:1158-1159, 1284, 1431-1448).
 *
 * One wavefront per chunk, wave-uniform state machine as in k_fast.  The buckets of the next 128 positions are
 * gathered a batch ahead (16 bytes per lane); an insertion shifts the new string into the gathered buckets of later
 * positions with the same hash in registers, and a batch's insertions reach the table together when the batch is
 * left: of the strings inserted under one hash value the last one's lane holds the bucket as it stands then.  No
 * link array, no LDS, no store per insertion. */
#ifndef ZFW_WAYS
#define ZFW_WAYS 4u
#endif
#define ZFW_NONE 0xffffffffu
#ifndef ZFW_SGPRS
#define ZFW_SGPRS 64
#endif

#define ZFW_UNI(x) __builtin_amdgcn_readfirstlane(x)
/* position of the lowest set bit, 0xffffffff for 0 (what the instruction returns; __builtin_ctz leaves 0 undefined) */
__device__ __forceinline__ uint32_t zfw_ffbl(uint32_t x) { uint32_t r; asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x)); return r; }
__device__ __forceinline__ uint32_t zfw_min3(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
/* v = wave-uniform x in lane `LANE` (a constant), unchanged elsewhere */
#define ZFW_SETLANE(v, x, LANE) asm("v_writelane_b32 %0, %1, " #LANE : "+v"(v) : "s"(x))
/* a wave's own loads see its own earlier stores: order for the compiler, nothing for the hardware to wait for */
#define ZFW_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront")

__device__ __forceinline__ void zfw_chunk(const ZkJob &J)
{
    const uint32_t lc = blockIdx.x, lane = threadIdx.x;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    const uint8_t *const b = J.in + g.dstart;
    const uint64_t readable = J.n - g.dstart;
    const uint32_t dict_len = g.dict_len, E = g.dict_len + g.len;
    uint4 *const tab = (uint4 *)J.fhead + (uint64_t)lc * (J.P.hash_mask + 1u);
    const uint32_t hshift = J.P.hash_shift, hmask = J.P.hash_mask, nice = J.P.nice, max_dist = J.P.max_dist;
    const uint32_t ways = J.P.chain < ZFW_WAYS ? J.P.chain : ZFW_WAYS;
    const uint32_t W = J.P.wsize, max_insert = J.P.lazy, lim = J.P.lit_bufsize - 1u;
    uint32_t *const symout = J.sym + (uint64_t)lc * J.P.chunk_size;
    GBlk blk; blk.b = J.blk + (uint64_t)lc * J.P.max_blocks; blk.cap = J.P.max_blocks; blk.err = J.err; blk.lane = lane;
    /* a comparison may read 16 bytes at a time while 258 + 16 bytes from p on lie inside the buffer */
    const uint32_t wide_end = readable > 274u ? (uint32_t)(readable - 274u < 0xfffffff0ull ? readable - 274u : 0xfffffff0ull) : 0u;
    const bool in_ways = lane < ways;
    const uint32_t inv_lane = 3u - lane, is0 = lane == 0 ? 1u : 0u;

    /* ---- the batch: hashes and buckets of positions q0 .. q0 + 63, one position per lane */
    uint32_t q0 = 0, h_cur, last_cur = ZFW_NONE;
    uint4 c_cur;
    uint32_t wb = 0x80000000u, wreg = 0;                 /* no a-coordinate is within 256 bytes of this */
    auto load_batch = [&](uint32_t qb, uint32_t &h, uint4 &c) {
        const uint32_t a = qb + lane;
        const bool ok = a + ZC_MIN_MATCH <= E;
        /* the batch's bytes come from a register that holds the 256 bytes from wb on (lane l: the dword at wb + 4 l, zero
         * past the buffer's end) and is loaded again when a batch would reach past it: one trip to memory for three
         * batches or so instead of one for each */
        if (qb - wb > 188u) {
            wb = qb & ~3u;
            const uint64_t wa = (uint64_t)wb + 4u * lane;
            uint32_t t = 0;
            if (wa + 4u <= readable) __builtin_memcpy(&t, b + wa, 4);
            else for (uint32_t i = 0; i < 4u && wa + i < readable; i++) t |= (uint32_t)b[wa + i] << (8u * i);
            wreg = t;
        }
        const uint32_t wo = qb - wb + lane, wd = wo >> 2;
        const uint32_t w = __builtin_amdgcn_alignbyte((uint32_t)__builtin_amdgcn_ds_bpermute((int)((wd + 1u) << 2), (int)wreg),
                                                      (uint32_t)__builtin_amdgcn_ds_bpermute((int)(wd << 2), (int)wreg), wo);
        h = ok ? zc_hash3(w & 0xffu, (w >> 8) & 0xffu, (w >> 16) & 0xffu, hshift, hmask) : ZFW_NONE;
        c = make_uint4(0, 0, 0, 0);
        if (ok) c = tab[h];
    };
    /* p has left the current batch: its insertions go to the table, and the 64 positions from p on are the next batch.
     * (Gathering a batch further ahead buys nothing: an insertion has to be shifted into every gathered bucket of its
     * hash value at once, so the gathered data is needed at the next insertion anyway.) */
    auto advance = [&](uint32_t p) {
        if (last_cur == lane) tab[h_cur] = c_cur;
        last_cur = ZFW_NONE;
        ZFW_WAVE_FENCE();
        q0 = p;
        load_batch(q0, h_cur, c_cur);
    };
    /* INSERT_STRING (deflate.c:186-189): the string at p enters the buckets that later positions of the batch
     * hold for its hash value, and its own lane's, which becomes the table entry when the batch is left */
    auto enter = [&](uint32_t p, uint32_t k, uint32_t hq) {
        const bool same = h_cur == hq;
        const bool mine = same && lane >= k;
        c_cur.w = mine ? c_cur.z : c_cur.w; c_cur.z = mine ? c_cur.y : c_cur.z;
        c_cur.y = mine ? c_cur.x : c_cur.y; c_cur.x = mine ? p : c_cur.x;
        last_cur = same ? k : last_cur;
    };
    auto insert_only = [&](uint32_t p) {
        if (p - q0 >= 64u) advance(p);
        const uint32_t k = p - q0;
        enter(p, k, __builtin_amdgcn_readlane(h_cur, k));
    };

    load_batch(0u, h_cur, c_cur);
    /* deflateSetDictionary inserts every dictionary string; the last two wait for the first input bytes
     * (deflate.c:364-381, 1471-1489) */
    for (uint32_t a = 0; a < dict_len && a + ZC_MIN_MATCH <= E; a++) insert_only(a);

    uint32_t B = 0, wend = dict_len, p = dict_len;                    /* fill_window's state, see ZcWindow */
    uint32_t nsym = 0, blk_sym0 = 0, bstart = dict_len, nblk = 0;     /* _tr_tally's and the block table's, see ZcSink */
    uint32_t symrow = 0;
    auto flush = [&](uint32_t last) {                                 /* FLUSH_BLOCK_ONLY, deflate.c:1538-1547 */
        ZcBlock kb;
        kb.sym_start = blk_sym0; kb.sym_count = nsym - blk_sym0;
        kb.in_start = bstart; kb.in_len = p - bstart;
        kb.flags = (last ? ZC_BF_LAST : 0u) | (bstart >= B ? ZC_BF_CANSTORE : 0u);
        kb.type = 0; kb.bits = 0; kb.hdr_bits = 0; kb.bit_off = 0;
        blk(nblk++, kb);
        blk_sym0 = nsym; bstart = p;
    };
    /* ---- searches, sixteen positions at a time.
     * One comparison of a search is a trip to memory that the parse has to wait for.  A round makes that trip for the
     * searches of positions r0 .. r0 + 15 together: lane 4 j + w measures candidate w of position r0 + j, from the
     * bucket that position's lane holds now.  A step at a position of the round whose bucket has not changed since
     * (no string of its hash value was inserted from r0 on) finds its answer in the round; at level 1 on text three
     * steps in four do.  Comparisons that run past 16 bytes, positions past the batch and the chunk's last 262 bytes
     * take the search below, one position at a time. */
    const uint32_t jl = lane >> 2, way = lane & 3u, inv_way = 3u - way, quad0 = lane & ~3u;
    const bool way_ok = way < ways, w0 = way == 0, w1 = way == 1, w2 = way == 2;
    uint32_t r0 = ZFW_NONE, rlen = 0, rkey = 0, rcand = 0;
    auto lcp16 = [&](uint32_t ca, uint32_t cb) {      /* common prefix of the 16 bytes at a-coordinates ca and cb, in bytes */
        uint4 a, c;
        __builtin_memcpy(&a, b + ca, 16); __builtin_memcpy(&c, b + cb, 16);
        return zfw_min3(zfw_min3(zfw_ffbl(a.x ^ c.x), zfw_ffbl(a.y ^ c.y) | 32u, zfw_ffbl(a.z ^ c.z) | 64u),
                        zfw_ffbl(a.w ^ c.w) | 96u, 128u) >> 3;
    };
    auto round = [&](uint32_t p, uint32_t k, uint32_t B, uint32_t thr) {
        const int src = (int)((k + jl) << 2);
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)c_cur.x), c1 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)c_cur.y);
        const uint32_t c2 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)c_cur.z), c3 = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)c_cur.w);
        const uint32_t candv = w0 ? c0 : w1 ? c1 : w2 ? c2 : c3;
        const uint32_t x = p + jl;
        const bool far = x - B > max_dist;
        const uint32_t lo = far ? x - max_dist : B;
        /* the head of the chain may be MAX_DIST away, the strings behind it must be nearer (deflate.c:1660, 1284);
         * window index 0 is NIL for all of them */
        const bool ok = way_ok & (k + jl < 64u) & (candv + ((w0 & far) ? 1u : 0u) > lo);
        uint32_t len = 0;
        if (ok) len = lcp16(candv, x);
        const uint32_t mine4 = (uint32_t)(__ballot(len >= thr) >> quad0) & 15u;     /* which of this position's candidates end the walk */
        const uint32_t cut = zfw_ffbl(mine4 | 8u);
        uint32_t key = way <= cut ? ((len << 2) | inv_way) : 0u;
        key = max(key, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0xB1, 0xf, 0xf, false));   /* quad_perm 1,0,3,2 */
        key = max(key, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x4E, 0xf, 0xf, false));   /* quad_perm 2,3,0,1 */
        rkey = key; rcand = candv;
        r0 = p; rlen = 64u - k < 16u ? 64u - k : 16u;
    };
    /* longest_match (deflate.c:1148-1289) with prev_length 2 for the one position p.  The walk goes on while the
     * window index of the next string is above the limit (:1158-1159, 1284): lo is the a-coordinate of that limit.
     * The bucket's strings grow older lane by lane, so the ones in reach are the first lanes.  Lane w measures
     * candidate w, then the reference's rule picks: strictly longer wins, stop at the first one of nice length */
    auto search = [&](uint32_t p, uint32_t k, uint32_t s0, uint32_t lo, uint32_t cap, uint32_t thr, uint32_t &mlen, uint32_t &mdist) {
        const uint32_t s1 = __builtin_amdgcn_readlane(c_cur.y, k), s2 = __builtin_amdgcn_readlane(c_cur.z, k), s3 = __builtin_amdgcn_readlane(c_cur.w, k);
        uint32_t candv = s0;
        ZFW_SETLANE(candv, s1, 1); ZFW_SETLANE(candv, s2, 2); ZFW_SETLANE(candv, s3, 3);
        const bool valid = in_ways & (candv + is0 > lo);         /* lane 0 has passed the head's test */
        uint32_t len = 0;
        if (valid) {
            if (p < wide_end) {
                len = lcp16(candv, p);
                if (__ballot(len == 16u) != 0ull) {             /* somebody's first 16 bytes are equal */
                    uint32_t off = 16u;
                    do {
                        const uint32_t r = lcp16(candv + off, p + off);
                        len = len == off ? off + r : len;      /* a comparison that has ended keeps its length */
                        off += 16u;
                    } while (off < cap && __ballot(len == off) != 0ull);
                }
            } else {
                const uint8_t *m = b + candv, *sp = b + p;
#pragma unroll 1
                while (len < cap && m[len] == sp[len]) len++;
            }
            len = len < cap ? len : cap;
        }
        const uint32_t cut = (uint32_t)__builtin_ctzll(__ballot(len >= thr) | 8ull);
        uint32_t key = lane <= cut ? ((len << 2) | inv_lane) : 0u;     /* lanes past the bucket hold length 0 */
        key = max(key, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0xB1, 0xf, 0xf, false));
        key = max(key, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)key, 0x4E, 0xf, 0xf, false));
        const uint32_t top = ZFW_UNI(key), best = top >> 2;
        if (best >= ZC_MIN_MATCH) { mlen = best; mdist = p - __builtin_amdgcn_readlane(candv, 3u - (top & 3u)); }
    };
    uint32_t last_step = 1;                           /* how far the last step moved p */
    /* One iteration of deflate_fast's loop (deflate.c:1641-1711) at p with `look` bytes of lookahead.  `full`: the
     * lookahead is MIN_LOOKAHEAD or more, so no length is cut short by it and every string has its three bytes. */
    auto step = [&](const bool full, const uint32_t look) {
        uint32_t mlen = 0, mdist = 0;
        if (full || look >= ZC_MIN_MATCH) {
            if (p - q0 >= 64u) { advance(p); r0 = ZFW_NONE; }
            const uint32_t k = p - q0;
            const uint32_t hq = __builtin_amdgcn_readlane(h_cur, k);
            const uint32_t s0 = __builtin_amdgcn_readlane(c_cur.x, k);
            /* the head of the chain must not be NIL and at most MAX_DIST away (deflate.c:1660) */
            const bool far = p - B > max_dist;
            const uint32_t lo = far ? p - max_dist : B;
            if (s0 >= (far ? lo : lo + 1u)) {
                const uint32_t cap = full ? ZC_MAX_MATCH : (look < ZC_MAX_MATCH ? look : ZC_MAX_MATCH);
                uint32_t thr = full ? (nice < ZC_MIN_LOOK ? nice : ZC_MIN_LOOK) : (nice < look ? nice : look);
                thr = thr < ZC_MIN_MATCH ? ZC_MIN_MATCH : thr;           /* nothing shorter than 3 beats best_len 2 and ends the walk */
                bool served = false;
                if (full) {
                    uint32_t j = p - r0;
                    /* a string inserted since the round began is the newest of its bucket */
                    bool fresh = j < rlen && s0 < r0;
                    if (!fresh && last_step < 16u && p + 16u <= wide_end) { round(p, k, B, thr); j = 0; fresh = true; }
                    if (fresh) {
                        const uint32_t top = __builtin_amdgcn_readlane(rkey, j << 2), best = top >> 2;
                        if (best < 16u) {
                            served = true;
                            if (best >= ZC_MIN_MATCH) { mlen = best; mdist = p - __builtin_amdgcn_readlane(rcand, (j << 2) + 3u - (top & 3u)); }
                        }
                    }
                }
                if (!served) search(p, k, s0, lo, cap, thr, mlen, mdist);
            }
            enter(p, k, hq);
        }
        uint32_t word;
        const uint32_t p0 = p;
        if (mlen >= ZC_MIN_MATCH) {
            word = zc_sym_match(mlen, mdist);
            if (mlen <= max_insert && (full || look - mlen >= ZC_MIN_MATCH))
                for (uint32_t i = 1; i < mlen; i++) insert_only(p0 + i);
            p = p0 + mlen;
        } else { word = p0; p = p0 + 1u; }
        last_step = p - p0;
        /* _tr_tally (trees.c:1010-1050): symbols leave in rows of 64, lane i & 63 keeps symbol i until the row is full */
        symrow = lane == (nsym & 63u) ? word : symrow;
        if ((nsym & 63u) == 63u) symout[(nsym & ~63u) + lane] = symrow;
        nsym++;
        if (nsym - blk_sym0 == lim) flush(0);
    };
    for (;;) {
        while (wend - p >= ZC_MIN_LOOK) step(true, ZC_MIN_LOOK);
        /* fill_window, deflate.c:1390-1494 */
        if (p - B >= W + (W - ZC_MIN_LOOK)) { B += W; r0 = ZFW_NONE; }          /* a round's limits hold for the window base it saw */
        const uint64_t cap = (uint64_t)B + 2u * W;
        wend = E < cap ? E : (uint32_t)cap;
        const uint32_t look = wend - p;
        if (look == 0) break;
        if (look < ZC_MIN_LOOK) step(false, look);
    }
    if (g.final) flush(1);
    else if (nsym != blk_sym0) flush(0);
    if (lane < (nsym & 63u)) symout[(nsym & ~63u) + lane] = symrow;    /* the last, partial row */
    if (lane == 0) {
        J.nblk[lc] = nblk < J.P.max_blocks ? nblk : J.P.max_blocks;
        J.nsym[lc] = nsym;
    }
}
/* Two builds of the same parser.  A SIMD holds eight wavefronts only if each takes at most 64 scalar registers (a
 * wavefront's allocation is its count plus 32, in steps of 16, out of 800; measured: 106 registers leave room for
 * six, 78 for seven); the compiler then keeps part of the state machine in lanes of a vector register, which costs
 * instructions.  With more chunks in flight than 24 per CU the eighth wavefront is worth it (8 GiB of text at
 * level 1: 22.4 -> 24.8 GB/s, the last 1 024 chunks no longer start when the first ones are done); with fewer, the
 * leaner code is (4 GiB: 16.9 against 15.3 GB/s). */
__global__ __launch_bounds__(64) void k_fastw(ZkJob J) { zfw_chunk(J); }
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(8, 8), amdgpu_num_sgpr(ZFW_SGPRS))) void k_fastw8(ZkJob J) { zfw_chunk(J); }


/* ------------------------------------------------------------------ k_fastr
 * deflate_fast for chain budgets of 5 to 32 (levels 2 and 3).  The same fact as in k_fastw — head[h] and the links
 * behind it are the strings inserted under hash value h, newest first — with buckets too large for registers: per hash
 * value a ring of the ZFR_WAYS newest strings (a-coordinates) and a count of all that were ever inserted, in global
 * memory.  A search reads the whole ring with one load (lane j takes slot j, lane ZFR_WAYS the count), knows every
 * slot's age from the count, measures all candidates at once and picks by age: no walk along links, of which level 3
 * made up to 32 dependent trips to memory per search.  An insertion is one store into the ring and one of the count;
 * the strings inside a short match go in together (their counts gathered with one load, strings of one hash value
 * ranked among themselves). */
#ifndef ZFR_WAYS
#define ZFR_WAYS 32u
#endif

__device__ __forceinline__ void zfr_chunk(const ZkJob &J)
{
    const uint32_t lc = blockIdx.x, lane = threadIdx.x;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    const uint8_t *const b = J.in + g.dstart;
    const uint64_t readable = J.n - g.dstart;
    const uint32_t dict_len = g.dict_len, E = g.dict_len + g.len;
    const uint32_t hsize = J.P.hash_mask + 1u;
    uint32_t *const ring = (uint32_t *)J.fhead + (uint64_t)lc * hsize * (ZFR_WAYS + 1u);
    uint32_t *const cnt = ring + (uint64_t)hsize * ZFR_WAYS;
    const uint32_t hshift = J.P.hash_shift, hmask = J.P.hash_mask, nice = J.P.nice, max_dist = J.P.max_dist;
    const uint32_t chain = J.P.chain < ZFR_WAYS ? J.P.chain : ZFR_WAYS;
    const uint32_t W = J.P.wsize, max_insert = J.P.lazy, lim = J.P.lit_bufsize - 1u;
    uint32_t *const symout = J.sym + (uint64_t)lc * J.P.chunk_size;
    GBlk blk; blk.b = J.blk + (uint64_t)lc * J.P.max_blocks; blk.cap = J.P.max_blocks; blk.err = J.err; blk.lane = lane;
    /* a comparison may read 16 bytes at a time while 258 + 16 bytes from p on lie inside the buffer */
    const uint32_t wide_end = readable > 274u ? (uint32_t)(readable - 274u < 0xfffffff0ull ? readable - 274u : 0xfffffff0ull) : 0u;

    /* The input from the parse's position on is kept in a register: lane l holds the dword at a-coordinate wb + 4 l (zero
     * past the buffer's end), 256 bytes in all, so that the 16 bytes a step starts from are five v_readlane and four
     * v_alignbyte instead of a trip to memory; the register is loaded again when the position leaves its first 240 bytes. */
    uint32_t wb = 0x80000000u, wreg = 0;                 /* no a-coordinate is within 240 bytes of this */
    auto window16 = [&](uint32_t x) {
        if (x - wb > 239u) {
            wb = x & ~3u;
            const uint64_t a = (uint64_t)wb + 4u * lane;
            uint32_t w = 0;
            if (a + 4u <= readable) __builtin_memcpy(&w, b + a, 4);
            else for (uint32_t i = 0; i < 4u && a + i < readable; i++) w |= (uint32_t)b[a + i] << (8u * i);
            wreg = w;
        }
        const uint32_t o = x - wb, d = o >> 2;
        const uint32_t s0 = __builtin_amdgcn_readlane(wreg, d), s1 = __builtin_amdgcn_readlane(wreg, d + 1u), s2 = __builtin_amdgcn_readlane(wreg, d + 2u),
                       s3 = __builtin_amdgcn_readlane(wreg, d + 3u), s4 = __builtin_amdgcn_readlane(wreg, d + 4u);
        return make_uint4(__builtin_amdgcn_alignbyte(s1, s0, o), __builtin_amdgcn_alignbyte(s2, s1, o),
                          __builtin_amdgcn_alignbyte(s3, s2, o), __builtin_amdgcn_alignbyte(s4, s3, o));
    };
    /* hash of the string that starts i bytes into v (i per lane, at most 12) */
    auto hash_at = [&](const uint4 &v, uint32_t i) {
        const uint32_t d = i >> 2;
        const uint32_t lo = d == 0 ? v.x : d == 1 ? v.y : d == 2 ? v.z : v.w;
        const uint32_t hi = d == 0 ? v.y : d == 1 ? v.z : d == 2 ? v.w : 0u;
        const uint32_t w = __builtin_amdgcn_alignbyte(hi, lo, i & 3u);
        return zc_hash3(w & 0xffu, (w >> 8) & 0xffu, (w >> 16) & 0xffu, hshift, hmask);
    };
    /* INSERT_STRING (deflate.c:186-189) for the m strings at x0 .. x0 + m - 1, lane i holding the hash of x0 + i: strings of
     * one hash value take consecutive slots in position order, the last of them leaves the count */
    auto insert_many = [&](uint32_t x0, uint32_t m, uint32_t h) {
        const bool act = lane < m;
        uint32_t c = 0;
        if (act) c = cnt[h];
        uint32_t rank = 0, later = 0;
        for (uint32_t k = 0; k < m; k++) {
            const uint32_t hk = __builtin_amdgcn_readlane(h, k);
            const bool same = act && h == hk;
            rank += (same && lane > k) ? 1u : 0u;
            later |= (same && lane < k) ? 1u : 0u;
        }
        if (act) {
            ring[(uint64_t)h * ZFR_WAYS + ((c + rank) & (ZFR_WAYS - 1u))] = x0 + lane;
            if (!later) cnt[h] = c + rank + 1u;
        }
        ZFW_WAVE_FENCE();
    };
    auto wave_max32 = [&](uint32_t x) {                    /* over lanes 0 .. 31 */
        x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xf, 0xf, false));     /* quad_perm 1,0,3,2 */
        x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xf, 0xf, false));     /* quad_perm 2,3,0,1 */
        x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xf, 0xf, false));    /* row_half_mirror */
        x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xf, 0xf, false));    /* row_mirror */
        const uint32_t a = __builtin_amdgcn_readlane(x, 0), c = __builtin_amdgcn_readlane(x, 16);
        return a > c ? a : c;
    };
    auto lcp16v = [&](uint32_t ca, const uint4 &c) {       /* common prefix of the 16 bytes at a-coordinate ca with c */
        uint4 a;
        __builtin_memcpy(&a, b + ca, 16);
        return zfw_min3(zfw_min3(zfw_ffbl(a.x ^ c.x), zfw_ffbl(a.y ^ c.y) | 32u, zfw_ffbl(a.z ^ c.z) | 64u),
                        zfw_ffbl(a.w ^ c.w) | 96u, 128u) >> 3;
    };

    /* deflateSetDictionary inserts every dictionary string; the last two wait for the first input bytes
     * (deflate.c:364-381, 1471-1489) */
    for (uint32_t a = 0; a < dict_len && a + ZC_MIN_MATCH <= E; a += 64u) {
        uint32_t m = dict_len - a < 64u ? dict_len - a : 64u;
        if (a + m + ZC_MIN_MATCH - 1u > E) m = E - (ZC_MIN_MATCH - 1u) - a;
        const uint32_t x = a + lane;
        uint32_t w = 0;
        if (lane < m) w = (uint32_t)b[x] | ((uint32_t)b[x + 1] << 8) | ((uint32_t)b[x + 2] << 16);
        insert_many(a, m, zc_hash3(w & 0xffu, (w >> 8) & 0xffu, (w >> 16) & 0xffu, hshift, hmask));
    }

    uint32_t B = 0, wend = dict_len, p = dict_len;                    /* fill_window's state, see ZcWindow */
    uint32_t nsym = 0, blk_sym0 = 0, bstart = dict_len, nblk = 0;     /* _tr_tally's and the block table's, see ZcSink */
    uint32_t symrow = 0;
    auto flush = [&](uint32_t last) {                                 /* FLUSH_BLOCK_ONLY, deflate.c:1538-1547 */
        ZcBlock kb;
        kb.sym_start = blk_sym0; kb.sym_count = nsym - blk_sym0;
        kb.in_start = bstart; kb.in_len = p - bstart;
        kb.flags = (last ? ZC_BF_LAST : 0u) | (bstart >= B ? ZC_BF_CANSTORE : 0u);
        kb.type = 0; kb.bits = 0; kb.hdr_bits = 0; kb.bit_off = 0;
        blk(nblk++, kb);
        blk_sym0 = nsym; bstart = p;
    };
    /* One iteration of deflate_fast's loop (deflate.c:1641-1711) at p with `look` bytes of lookahead.  `full`: the
     * lookahead is MIN_LOOKAHEAD or more, so no length is cut short by it and every string has its three bytes. */
    auto step = [&](const bool full, const uint32_t look) {
        uint32_t mlen = 0, mdist = 0;
        uint4 pv = make_uint4(0, 0, 0, 0);
        if (full || look >= ZC_MIN_MATCH) {
            pv = window16(p);
            const uint32_t h0 = ZFW_UNI(hash_at(pv, 0u));
            /* the bucket as it stands before this string goes in: the slots by lanes 0 .. 31, the count by lane 32 */
            uint32_t val = 0;
            if (lane <= ZFR_WAYS) val = lane < ZFR_WAYS ? ring[(uint64_t)h0 * ZFR_WAYS + lane] : cnt[h0];
            const uint32_t c = __builtin_amdgcn_readlane(val, ZFR_WAYS);
            if (lane < 2u) { if (lane == 0) ring[(uint64_t)h0 * ZFR_WAYS + (c & (ZFR_WAYS - 1u))] = p; else cnt[h0] = c + 1u; }
            ZFW_WAVE_FENCE();
            /* longest_match (deflate.c:1148-1289) with prev_length 2: the head of the chain must not be NIL and at most
             * MAX_DIST away (:1660), the strings behind it must be nearer (:1158-1159, 1284), the walk takes at most
             * max_chain of them; they grow older (and lie further back) slot by slot down from the newest */
            const uint32_t age = (c - 1u - lane) & (ZFR_WAYS - 1u);
            const bool far = p - B > max_dist;
            const uint32_t lo = far ? p - max_dist : B;
            const uint32_t nent = c < chain ? c : chain;
            const bool valid = lane < ZFR_WAYS && age < nent && val + ((age == 0u && far) ? 1u : 0u) > lo;
            const uint32_t vm = (uint32_t)__ballot(valid);
            if (c != 0u && ((vm >> ((c - 1u) & (ZFR_WAYS - 1u))) & 1u)) {
                const uint32_t cap = full ? ZC_MAX_MATCH : (look < ZC_MAX_MATCH ? look : ZC_MAX_MATCH);
                uint32_t thr = full ? (nice < ZC_MIN_LOOK ? nice : ZC_MIN_LOOK) : (nice < look ? nice : look);
                thr = thr < ZC_MIN_MATCH ? ZC_MIN_MATCH : thr;           /* nothing shorter than 3 beats best_len 2 and ends the walk */
                uint32_t len = 0;
                if (valid) {
                    if (p < wide_end) {
                        len = lcp16v(val, pv);
                        if (__ballot(len == 16u) != 0ull) {             /* somebody's first 16 bytes are equal */
                            uint32_t off = 16u;
                            do {
                                uint4 cc; __builtin_memcpy(&cc, b + (p + off), 16);
                                const uint32_t r = lcp16v(val + off, cc);
                                len = len == off ? off + r : len;      /* a comparison that has ended keeps its length */
                                off += 16u;
                            } while (off < cap && __ballot(len == off) != 0ull);
                        }
                    } else {
                        const uint8_t *m = b + val, *sp = b + p;
#pragma unroll 1
                        while (len < cap && m[len] == sp[len]) len++;
                    }
                    len = len < cap ? len : cap;
                }
                /* the reference's rule by age: strictly longer wins, the walk ends at the first string of nice length.
                 * Bit a of `ends`: the string of age a ends the walk (slot s holds age (c - 1 - s) mod 32) */
                const uint32_t gm = (uint32_t)__ballot(valid && len >= thr);
                const uint32_t rv = __brev(gm), sh = c & (ZFR_WAYS - 1u);
                const uint32_t ends = sh ? ((rv << sh) | (rv >> (32u - sh))) : rv;
                const uint32_t cut = ends ? (uint32_t)__builtin_ctz(ends) : ZFR_WAYS - 1u;
                const uint32_t key = (valid && age <= cut) ? ((len << 6) | (63u - age)) : 0u;
                const uint32_t top = wave_max32(key), best = top >> 6;
                if (best >= ZC_MIN_MATCH) {
                    const uint32_t bage = 63u - (top & 63u);
                    mlen = best; mdist = p - __builtin_amdgcn_readlane(val, (c - 1u - bage) & (ZFR_WAYS - 1u));
                }
            }
        }
        uint32_t word;
        const uint32_t p0 = p;
        if (mlen >= ZC_MIN_MATCH) {
            word = zc_sym_match(mlen, mdist);
            if (mlen <= max_insert && (full || look - mlen >= ZC_MIN_MATCH)) {
                /* the strings inside the match: the first twelve have their bytes in pv, the rest are fetched */
                const uint32_t m = mlen - 1u;
                uint32_t done = m < 12u ? m : 12u;
                insert_many(p0 + 1u, done, hash_at(pv, lane < 12u ? lane + 1u : 0u));
                while (done < m) {
                    const uint32_t take = m - done < 13u ? m - done : 13u;
                    const uint4 v = window16(p0 + 1u + done);
                    insert_many(p0 + 1u + done, take, hash_at(v, lane < 13u ? lane : 0u));
                    done += take;
                }
            }
            p = p0 + mlen;
        } else { word = p0; p = p0 + 1u; }
        /* _tr_tally (trees.c:1010-1050): symbols leave in rows of 64, lane i & 63 keeps symbol i until the row is full */
        symrow = lane == (nsym & 63u) ? word : symrow;
        if ((nsym & 63u) == 63u) symout[(nsym & ~63u) + lane] = symrow;
        nsym++;
        if (nsym - blk_sym0 == lim) flush(0);
    };
    for (;;) {
        while (wend - p >= ZC_MIN_LOOK) step(true, ZC_MIN_LOOK);
        /* fill_window, deflate.c:1390-1494 */
        if (p - B >= W + (W - ZC_MIN_LOOK)) B += W;
        const uint64_t cap = (uint64_t)B + 2u * W;
        wend = E < cap ? E : (uint32_t)cap;
        const uint32_t look = wend - p;
        if (look == 0) break;
        if (look < ZC_MIN_LOOK) step(false, look);
    }
    if (g.final) flush(1);
    else if (nsym != blk_sym0) flush(0);
    if (lane < (nsym & 63u)) symout[(nsym & ~63u) + lane] = symrow;    /* the last, partial row */
    if (lane == 0) {
        J.nblk[lc] = nblk < J.P.max_blocks ? nblk : J.P.max_blocks;
        J.nsym[lc] = nsym;
    }
}
/* (a second build that fits eight wavefronts to a SIMD, as k_fastw has one, made no difference here: the trips to memory
 * per step and twice the instructions, not the wavefronts in flight, are what this kernel runs at) */
__global__ __launch_bounds__(64) void k_fastr(ZkJob J) { zfr_chunk(J); }
