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

struct ZkFastWCtx {
    const uint8_t *b; uint64_t readable; uint32_t E;
    uint4 *tab;                   /* global: one bucket per hash value: a-coordinates of its newest strings, newest first, 0 = none */
    uint32_t hshift, hmask, nice, chain, lane, max_dist;
    uint32_t q0;                  /* first position of the current batch */
    uint32_t h_cur, h_nxt;        /* per lane: hash of q0 + lane, of q0 + 64 + lane (ZFW_NONE: no string starts there) */
    uint4 c_cur, c_nxt;           /* per lane: the bucket of that hash as the string at that position will find it */
    uint32_t last_cur;            /* per lane: the lane of the current batch that inserted last under this lane's hash */
    uint32_t s0, s1, s2, s3;      /* wave-uniform: the bucket the last insertion found */

    __device__ __forceinline__ uint32_t bytes4(uint32_t a) const
    {
        uint32_t w = 0;
        if ((uint64_t)a + 4u <= readable) __builtin_memcpy(&w, b + a, 4);
        else for (uint32_t i = 0; i < 4 && (uint64_t)a + i < readable; i++) w |= (uint32_t)b[a + i] << (8u * i);
        return w;
    }
    __device__ __forceinline__ void load_batch(uint32_t qb, uint32_t &h, uint4 &c)
    {
        const uint32_t a = qb + lane;
        const bool ok = a + ZC_MIN_MATCH <= E;
        const uint32_t w = ok ? bytes4(a) : 0u;
        h = ok ? zc_hash3(w & 0xffu, (w >> 8) & 0xffu, (w >> 16) & 0xffu, hshift, hmask) : ZFW_NONE;
        c = make_uint4(0, 0, 0, 0);
        if (ok) c = tab[h];
    }
    /* the current batch's insertions go to the table */
    __device__ __forceinline__ void flush_cur()
    {
        if (last_cur == lane) tab[h_cur] = c_cur;
        last_cur = ZFW_NONE;
        __threadfence_block();
    }
    __device__ __forceinline__ void start_batches(uint32_t q)
    {
        q0 = q;
        load_batch(q0, h_cur, c_cur);
        load_batch(q0 + 64u, h_nxt, c_nxt);
    }
    __device__ __forceinline__ uint32_t insert(uint32_t p, uint32_t B)
    {
        if (p - q0 >= 64u) {
            flush_cur();
            if (p - q0 < 128u) {
                h_cur = h_nxt; c_cur = c_nxt; q0 += 64u;
                load_batch(q0 + 64u, h_nxt, c_nxt);
            } else start_batches(p);
        }
        const uint32_t k = p - q0;
        const uint32_t hq = __builtin_amdgcn_readlane(h_cur, k);
        s0 = __builtin_amdgcn_readlane(c_cur.x, k); s1 = __builtin_amdgcn_readlane(c_cur.y, k);
        s2 = __builtin_amdgcn_readlane(c_cur.z, k); s3 = __builtin_amdgcn_readlane(c_cur.w, k);
        const bool same = h_cur == hq;
        if (same && lane >= k) c_cur = make_uint4(p, c_cur.x, c_cur.y, c_cur.z);
        last_cur = same ? k : last_cur;
        if (h_nxt == hq) c_nxt = make_uint4(p, c_nxt.x, c_nxt.y, c_nxt.z);
        return s0 > B ? s0 - B : 0u;
    }
    __device__ __forceinline__ void slide() {}
    /* longest_match (deflate.c:1148-1289) with prev_length 2 over the bucket insert() found: lane k measures
     * candidate k, then the reference's rule picks: strictly longer wins, stop at the first one of nice length. */
    __device__ __forceinline__ void longest(uint32_t p, uint32_t B, uint32_t, uint32_t look, uint32_t *mlen, uint32_t *mdist)
    {
        /* the walk goes on while the window index of the next string is above the limit (deflate.c:1158-1159, 1284) */
        const uint32_t lo = p - B > max_dist ? p - max_dist : B;
        const uint32_t cap = look < ZC_MAX_MATCH ? look : ZC_MAX_MATCH, nice_eff = nice < look ? nice : look;
        const uint32_t candv = lane == 0 ? s0 : lane == 1 ? s1 : lane == 2 ? s2 : s3;
        const bool valid = lane == 0 || (lane < ZFW_WAYS && lane < chain && candv > lo);
        const uint32_t n = (uint32_t)__builtin_ctzll(~__ballot(valid));
        uint32_t len = 0;
        if (lane < n) {
            const uint8_t *m = b + candv, *s = b + p;
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
        if (best >= ZC_MIN_MATCH) { *mlen = best; *mdist = p - __builtin_amdgcn_readlane(candv, bk); }
    }
};

__global__ __launch_bounds__(64) void k_fastw(ZkJob J)
{
    const uint32_t lc = blockIdx.x, lane = threadIdx.x;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    ZkFastWCtx cx;
    cx.b = J.in + g.dstart; cx.readable = J.n - g.dstart; cx.E = g.dict_len + g.len;
    cx.tab = (uint4 *)J.fhead + (uint64_t)lc * (J.P.hash_mask + 1u);
    cx.hshift = J.P.hash_shift; cx.hmask = J.P.hash_mask;
    cx.nice = J.P.nice; cx.chain = J.P.chain; cx.lane = lane; cx.max_dist = J.P.max_dist;
    cx.last_cur = ZFW_NONE; cx.s0 = cx.s1 = cx.s2 = cx.s3 = 0;
    cx.start_batches(0);
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
