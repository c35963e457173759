/*
 * zamd_fast_impl.h — deflate_fast (levels 1-3, deflate.c:1628-1722) on the GPU.
 * Included by zamd_engine.hip after zamd_inflate_impl.h (uses its register-window
 * byte reader).
 *
 * The hash chains of these levels depend on the parse itself (strings inside a
 * match longer than max_insert_length are never inserted), so a chunk is parsed
 * by ONE wavefront that owns head[] and prev[] in LDS as 16-bit window indices,
 * exactly as the reference lays them out (128 KiB: one chunk per CU).  The state
 * machine (zc_parse_fast) is wave-uniform; the lanes' width is used where the
 * reference loops: the candidates of a chain walk are compared with the current
 * string all at once, one per lane, and the tables are rebased by all lanes when
 * the window slides.  The next 256 bytes of input sit in a VGPR (one dword per
 * lane), so hashing a position touches no memory but the LDS tables.
 */

struct ZkFastCtx {
    const uint8_t *b;             /* a-coordinate 0 */
    uint64_t readable;            /* bytes of b that may be read */
    uint16_t *head, *prev;
    uint32_t hshift, hmask, hsize, nice, chain, lane;
    ZiBits win;

    __device__ __forceinline__ uint32_t insert(uint32_t p, uint32_t B)
    {
        win.bitpos = (uint64_t)p * 8u;
        const uint32_t w = (uint32_t)zi_peek(win, lane);
        const uint32_t h = zc_hash3(w & 0xffu, (w >> 8) & 0xffu, (w >> 16) & 0xffu, hshift, hmask);
        const uint32_t e = zi_uni(head[h]);
        if (lane == 0) { prev[(p - B) & 32767u] = (uint16_t)e; head[h] = (uint16_t)(p - B); }
        return e;
    }
    __device__ __forceinline__ void slide()
    {
        for (uint32_t i = lane; i < hsize; i += 64) { uint32_t m = head[i]; head[i] = (uint16_t)(m >= 32768u ? m - 32768u : 0u); }
        for (uint32_t i = lane; i < 32768u; i += 64) { uint32_t m = prev[i]; prev[i] = (uint16_t)(m >= 32768u ? m - 32768u : 0u); }
    }
    /* longest_match (deflate.c:1148-1289) with prev_length 2: the chain is walked
     * first (at most max_chain links, every one above the distance limit), then
     * lane k measures candidate k, then the reference's rule picks: strictly
     * longer wins, stop at the first one that reaches nice_match. */
    __device__ __forceinline__ void longest(uint32_t p, uint32_t B, uint32_t cur, uint32_t look, uint32_t *mlen, uint32_t *mdist)
    {
        const uint32_t sw = p - B, limit = sw > ZC_MAX_DIST ? sw - ZC_MAX_DIST : 0u;
        const uint32_t cap = look < ZC_MAX_MATCH ? look : ZC_MAX_MATCH, nice_eff = nice < look ? nice : look;
        uint32_t candv = 0, n = 0;
        do {
            candv = lane == n ? cur : candv;
            n++;
            cur = zi_uni(prev[cur & 32767u]);
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

__global__ __launch_bounds__(64) void k_fast(ZkJob J)
{
    __shared__ uint16_t s_head[32768];
    __shared__ uint16_t s_prev[32768];
    const uint32_t lc = blockIdx.x, lane = threadIdx.x;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head);
    ZkFastCtx cx;
    cx.b = J.in + g.dstart; cx.readable = J.n - g.dstart;
    cx.head = s_head; cx.prev = s_prev;
    cx.hshift = J.P.hash_shift; cx.hmask = J.P.hash_mask; cx.hsize = J.P.hash_mask + 1u;
    cx.nice = J.P.nice; cx.chain = J.P.chain; cx.lane = lane;
    zi_bits_init(cx.win, (const uint32_t *)cx.b, (cx.readable + 3) >> 2, 0, lane);
    for (uint32_t i = lane; i < cx.hsize; i += 64) s_head[i] = 0;
    for (uint32_t i = lane; i < 32768u; i += 64) s_prev[i] = 0;
    __syncthreads();
    GSym sym; sym.s = J.sym + (uint64_t)lc * J.P.chunk_size; sym.lane = lane;
    GBlk blk; blk.b = J.blk + (uint64_t)lc * J.P.max_blocks; blk.cap = J.P.max_blocks; blk.err = J.err; blk.lane = lane;
    uint32_t nsym = 0;
    const uint32_t nb = zc_parse_fast(g.dict_len, g.len, g.final, J.P, cx, sym, blk, &nsym);
    if (lane == 0) {
        J.nblk[lc] = nb < J.P.max_blocks ? nb : J.P.max_blocks;
        J.nsym[lc] = nsym;
    }
}
