/*
 * zamd_fast_impl.h — deflate_fast (levels 1-3, deflate.c:1628-1722) on the GPU.
 * Included by zamd_engine.hip.
 *
 * The hash chains of these levels depend on the parse itself (strings inside a
 * match longer than max_insert_length are never inserted), so a chunk is parsed
 * by ONE wavefront that owns the chunk's head[] and prev[] as 16-bit window
 * indices, exactly as the reference lays them out.  The state machine
 * (zc_parse_fast) is wave-uniform; what keeps it off the memory latency:
 *   - the last 64 KiB of input live in an LDS ring (filled 1 KiB at a time, well
 *     ahead of the parse), prev[] lives in LDS too;
 *   - head[] lives in global memory, but never on the critical path: the hashes
 *     of the next 128 positions are computed by the lanes and their head entries
 *     gathered one batch ahead; an insertion patches the gathered values of later
 *     positions with the same hash in registers, so a position's head entry is a
 *     v_readlane away when the parse gets there;
 *   - the candidates of a chain walk are compared with the current string all at
 *     once, one per lane, out of the ring.
 */

#define ZF_RING      65536u
#define ZF_RING_PAD  288u       /* the first bytes are mirrored behind the end: reads of <= 272 bytes never wrap */

struct ZkFastCtx {
    const uint8_t *b;             /* a-coordinate 0 */
    uint64_t readable;            /* bytes of b that may be read */
    uint32_t E;                   /* end of the chunk's data (a-coordinate) */
    uint8_t *ring; uint16_t *prev; uint16_t *fhead;
    uint32_t hshift, hmask, hsize, nice, chain, lane;
    uint32_t ring_hi;             /* ring holds a-coordinates [ring_hi - ZF_RING, ring_hi) */
    uint32_t q0;                  /* first position of the current batch */
    uint32_t h_cur, hv_cur, h_nxt, hv_nxt;   /* per lane: hash and head entry of q0 + lane, q0 + 64 + lane */
    uint32_t curB;                /* window base the gathered entries are relative to */

    __device__ __forceinline__ void fill_ring(uint32_t upto)
    {
        while (ring_hi < upto) {
            const uint32_t a = ring_hi + lane * 16u;
            uint4 v = make_uint4(0, 0, 0, 0);
            if ((uint64_t)a + 16u <= readable) v = *(const uint4 *)(b + a);
            else if (a < readable) {
                uint32_t t[4] = {0, 0, 0, 0};
                for (uint32_t i = 0; i < 16 && (uint64_t)a + i < readable; i++) t[i >> 2] |= (uint32_t)b[a + i] << (8u * (i & 3u));
                v = make_uint4(t[0], t[1], t[2], t[3]);
            }
            const uint32_t r = a & (ZF_RING - 1u);
            *(uint4 *)(ring + r) = v;
            if (r < ZF_RING_PAD) *(uint4 *)(ring + ZF_RING + r) = v;
            ring_hi += 1024u;
        }
    }
    __device__ __forceinline__ uint32_t load4(uint32_t a) const
    {
        const uint32_t r = a & (ZF_RING - 1u), i = r >> 2;
        const uint32_t *w = (const uint32_t *)ring;
        return __builtin_amdgcn_alignbyte(w[i + 1], w[i], r & 3u);
    }
    /* hashes and head entries of the 64 positions from qb on */
    __device__ __forceinline__ void load_batch(uint32_t qb, uint32_t &h, uint32_t &hv)
    {
        const uint32_t a = qb + lane;
        const bool ok = a + ZC_MIN_MATCH <= E;
        const uint32_t w = load4(a);
        h = ok ? zc_hash3(w & 0xffu, (w >> 8) & 0xffu, (w >> 16) & 0xffu, hshift, hmask) : 0xffffffffu;
        hv = ok ? (uint32_t)fhead[h] : 0u;
    }
    __device__ __forceinline__ void start_batches(uint32_t q, uint32_t B)
    {
        fill_ring(q + 2048u);
        __threadfence_block();                 /* earlier head stores have landed */
        q0 = q; curB = B;
        load_batch(q0, h_cur, hv_cur);
        load_batch(q0 + 64u, h_nxt, hv_nxt);
    }
    __device__ __forceinline__ uint32_t insert(uint32_t p, uint32_t B)
    {
        if (p - q0 >= 64u) {
            if (p - q0 < 128u) {
                if (ring_hi < p + 1024u) fill_ring(p + 8192u);
                h_cur = h_nxt; hv_cur = hv_nxt; q0 += 64u;
                __threadfence_block();
                load_batch(q0 + 64u, h_nxt, hv_nxt);
            } else start_batches(p, B);
        }
        const uint32_t k = p - q0, wi = p - B;
        const uint32_t hq = __builtin_amdgcn_readlane(h_cur, k);
        const uint32_t e = __builtin_amdgcn_readlane(hv_cur, k);
        hv_cur = (lane > k && h_cur == hq) ? wi : hv_cur;
        hv_nxt = h_nxt == hq ? wi : hv_nxt;
        if (lane == 0) { prev[wi & 32767u] = (uint16_t)e; fhead[hq] = (uint16_t)wi; }
        return e;
    }
    __device__ __forceinline__ void slide()
    {
        __threadfence_block();
        for (uint32_t i = lane * 8u; i < hsize; i += 512u) {
            uint4 q = *(uint4 *)(fhead + i);
            uint32_t t[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const uint32_t lo = t[j] & 0xffffu, hi = t[j] >> 16;
                t[j] = (lo >= 32768u ? lo - 32768u : 0u) | ((hi >= 32768u ? hi - 32768u : 0u) << 16);
            }
            *(uint4 *)(fhead + i) = make_uint4(t[0], t[1], t[2], t[3]);
        }
        for (uint32_t i = lane; i < 32768u; i += 64) { const uint32_t m = prev[i]; prev[i] = (uint16_t)(m >= 32768u ? m - 32768u : 0u); }
        hv_cur = hv_cur >= 32768u ? hv_cur - 32768u : 0u;
        hv_nxt = hv_nxt >= 32768u ? hv_nxt - 32768u : 0u;
        curB += 32768u;
        __threadfence_block();
    }
    /* longest_match (deflate.c:1148-1289) with prev_length 2: the chain is walked
     * first (at most max_chain links, every one above the distance limit), then
     * lane k measures candidate k, then the reference's rule picks: strictly
     * longer wins, stop at the first one that reaches nice_match. */
    __device__ __forceinline__ void longest(uint32_t p, uint32_t B, uint32_t cur, uint32_t look, uint32_t *mlen, uint32_t *mdist)
    {
        const uint32_t sw = p - B, limit = sw > ZC_MAX_DIST ? sw - ZC_MAX_DIST : 0u;
        const uint32_t cap = look < ZC_MAX_MATCH ? look : ZC_MAX_MATCH, nice_eff = nice < look ? nice : look;
        if (ring_hi < p + 512u) fill_ring(p + 8192u);
        uint32_t candv = 0, n = 0;
        do {
            candv = lane == n ? cur : candv;
            n++;
            cur = zi_uni(prev[cur & 32767u]);
        } while (cur > limit && n < chain);
        uint32_t len = 0;
        if (lane < n) {
            const uint32_t m = B + candv;
            while (len < cap) {
                const uint32_t x0 = load4(m + len) ^ load4(p + len);
                if (x0) { len += zc_ctz32(x0) >> 3; break; }
                const uint32_t x1 = load4(m + len + 4u) ^ load4(p + len + 4u);
                if (x1) { len += 4u + (zc_ctz32(x1) >> 3); break; }
                len += 8;
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

__global__ __launch_bounds__(64) void k_fast(ZkJob J)
{
    __shared__ __attribute__((aligned(16))) uint8_t s_ring[ZF_RING + ZF_RING_PAD];
    __shared__ uint16_t s_prev[32768];
    const uint32_t lc = blockIdx.x, lane = threadIdx.x;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head);
    ZkFastCtx cx;
    cx.b = J.in + g.dstart; cx.readable = J.n - g.dstart; cx.E = g.dict_len + g.len;
    cx.ring = s_ring; cx.prev = s_prev; cx.fhead = J.fhead + (uint64_t)lc * 65536u;
    cx.hshift = J.P.hash_shift; cx.hmask = J.P.hash_mask; cx.hsize = J.P.hash_mask + 1u;
    cx.nice = J.P.nice; cx.chain = J.P.chain; cx.lane = lane;
    cx.ring_hi = 0;
    for (uint32_t i = lane; i < 32768u; i += 64) s_prev[i] = 0;
    __syncthreads();
    cx.start_batches(0, 0);
    GSym sym; sym.s = J.sym + (uint64_t)lc * J.P.chunk_size; sym.lane = lane;
    GBlk blk; blk.b = J.blk + (uint64_t)lc * J.P.max_blocks; blk.cap = J.P.max_blocks; blk.err = J.err; blk.lane = lane;
    uint32_t nsym = 0;
    const uint32_t nb = zc_parse_fast(g.dict_len, g.len, g.final, J.P, cx, sym, blk, &nsym);
    if (lane == 0) {
        J.nblk[lc] = nb < J.P.max_blocks ? nb : J.P.max_blocks;
        J.nsym[lc] = nsym;
    }
}
