/* k_match8: k_match over a window image in which every candidate peek is ONE 8-byte LDS read.
 *
 * k_match is bound by the LDS array: per candidate it reads two aligned dwords (the three peek bytes may straddle
 * a dword) and a 16-bit link from addresses that are random across the wavefront, and a 4-byte LDS read is banked
 * over 32 banks per 32 lanes — about 3.5 cycles per lane group where a conflict-free read takes one, so the pair of
 * dwords costs about 14 LDS cycles and the link 7 (the measured 20 cycles per wavefront step, times the steps, is the
 * kernel's run time).  An 8-byte read is banked over 64 banks and fetches both dwords in about 7 cycles, but must be
 * 8-byte aligned.  Here the window bytes are therefore stored as overlapping 8-byte slots — slot j holds the bytes
 * 4j..4j+7 of the window — so that any two neighbouring dwords are one aligned slot.  That doubles the window image;
 * the tile shrinks from 16 384 to 7 936 positions to make room (40 704 positions with the 32 KiB halo: 2 x 41 024
 * bytes of slots + 2 x 40 704 bytes of links = 163 456 of the CU's 163 840 bytes).
 *
 * The walk itself is k_match's, statement for statement (zc_match_init / zc_match_peek / zc_match_extend /
 * zc_match_advance of zamd_core.h; deflate.c:1148-1289).  Included by zamd_engine.hip behind k_match. */

#define ZK8_TILE      7936u
#define ZK8_POS       (ZK_HALO + ZK8_TILE)                 /* positions whose links are in LDS */
#define ZK8_WIN_BYTES (ZK8_POS + 320u)                     /* window bytes imaged (compares read up to 258 + 16 bytes on) */
#define ZK8_SLOT_BYTES (2u * ZK8_WIN_BYTES)
#define ZK8_LDS_BYTES (ZK8_SLOT_BYTES + 2u * ZK8_POS)

struct LdsWin8 {            /* Win of zamd_core.h over the slot image */
    const uint2 *w; uint32_t a0;
    __device__ uint32_t load4(uint32_t a) const
    {
        uint32_t off = a - a0;
        uint2 q = w[off >> 2];
        return __builtin_amdgcn_alignbyte(q.y, q.x, off);
    }
};

template <int THRESH>
__global__ __launch_bounds__(ZK_MATCH_THREADS) void k_match8(ZkJob J, uint32_t tiles_per_chunk)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_lds[ZK8_LDS_BYTES / 4];
    __shared__ uint32_t s_next;
    uint2 *s_slot = (uint2 *)s_lds;
    uint16_t *s_pd = (uint16_t *)(s_lds + ZK8_SLOT_BYTES / 4);
    const uint32_t tid = threadIdx.x;
    const uint32_t lc = blockIdx.x / tiles_per_chunk, tile = blockIdx.x % tiles_per_chunk;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head);
    if ((uint64_t)tile * ZK8_TILE >= g.len) return;
    const uint32_t E = g.dict_len + g.len;
    const uint32_t t0 = g.dict_len + tile * ZK8_TILE;
    const uint32_t tlen = (E - t0) < ZK8_TILE ? (E - t0) : ZK8_TILE;
    const int64_t a0 = (int64_t)t0 - (int64_t)ZK_HALO;      /* a-coordinate of window byte 0 */
    /* the plain bytes are staged where the links will lie, spread into slots, then the links are loaded over them */
    {
        uint32_t *stage = (uint32_t *)s_pd;
        zk_load_window(J, g, a0, stage, ZK8_WIN_BYTES + 16u, tid);
        __syncthreads();
        for (uint32_t j = tid; j < ZK8_WIN_BYTES / 4; j += ZK_MATCH_THREADS) s_slot[j] = make_uint2(stage[j], stage[j + 1]);
        __syncthreads();
        zk_load_links(J, g, a0, s_pd, ZK8_POS, tid);
    }
    if (tid == 0) s_next = 0;
    __syncthreads();
    LdsWin8 win; win.w = s_slot; win.a0 = (uint32_t)a0;
    LdsPd pd; pd.pd = s_pd; pd.a0 = (uint32_t)a0;
    uint2 *rec = J.rec + (g.dstart + t0 - J.rec_base);
    const uint32_t lane = tid & 63u;
    const uint32_t nice_match = J.P.nice, max_chain = J.P.chain, qb = max_chain >> 2;
    const uint32_t la0 = (uint32_t)a0;
    const uint8_t *s_b = (const uint8_t *)s_lds;
    uint32_t p = t0, cur = t0, best = 2, bdist = 0, qrec = 0, scan3 = 0, cap = 0, nice = 0;
    int32_t kk = 0;                           /* candidates examined minus max_chain: negative while budget remains */
    int32_t walking = 0, parked = 0;          /* sign masks: -1 / 0 */
    uint32_t wcur = 0, wend = 0;              /* this wave's ticket of unclaimed tile positions */
    bool exhausted = false;
    for (;;) {
        /* ---- full comparisons of parked candidates (zc_match_extend, then zc_match_advance) */
        if (parked < 0) {
            uint32_t len, dnext;
            {
                const uint32_t oc = cur - la0, op = p - la0, ic = oc >> 2, ip = op >> 2;
                dnext = pd(cur);
                const uint2 ca = s_slot[ic], cb = s_slot[ic + 2], cc = s_slot[ic + 4];
                const uint2 qa = s_slot[ip], qb2 = s_slot[ip + 2], qc = s_slot[ip + 4];
                const uint32_t x0 = __builtin_amdgcn_alignbyte(ca.y, ca.x, oc) ^ __builtin_amdgcn_alignbyte(qa.y, qa.x, op);
                const uint32_t x1 = __builtin_amdgcn_alignbyte(cb.x, ca.y, oc) ^ __builtin_amdgcn_alignbyte(qb2.x, qa.y, op);
                const uint32_t x2 = __builtin_amdgcn_alignbyte(cb.y, cb.x, oc) ^ __builtin_amdgcn_alignbyte(qb2.y, qb2.x, op);
                const uint32_t x3 = __builtin_amdgcn_alignbyte(cc.x, cb.y, oc) ^ __builtin_amdgcn_alignbyte(qc.x, qb2.y, op);
                len = x0 ? zc_ctz32(x0) >> 3 : x1 ? 4u + (zc_ctz32(x1) >> 3) : x2 ? 8u + (zc_ctz32(x2) >> 3) : x3 ? 12u + (zc_ctz32(x3) >> 3) : 16u;
            }
            if (len == 16u) {
                while (len < cap) {
                    uint32_t x = win.load4(cur + len) ^ win.load4(p + len);
                    if (x) { len += zc_ctz32(x) >> 3; break; }
                    len += 4;
                }
            }
            if (len > cap) len = cap;
            bool fin = false;
            const uint32_t k = (uint32_t)(kk + (int32_t)max_chain);     /* the step that parked this walk counted the candidate */
            if (len > best) {
                best = len; bdist = p - cur;
                if (k <= qb) qrec = best | (bdist << 16);
                if (len >= nice) fin = true;
                else scan3 = win.load4(p + len - 2u) & 0x00ffffffu;
            }
            if (!fin) {
                const uint32_t d = dnext;
                if (kk >= 0 || d == 0 || d >= cur) fin = true;
                else { cur -= d; fin = p - cur >= ZC_MAX_DIST; }
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
                if (zc_match_init(win, pd, t0 + idx, E, g.corner, nice_match, st)) {
                    p = st.p; cur = st.cur; kk = -(int32_t)max_chain; best = 2; bdist = 0; qrec = 0;
                    scan3 = st.scan3; cap = st.cap; nice = st.nice;
                    walking = -1;
                } else rec[idx] = make_uint2(0u, 0u);
            }
            wcur += take;
        }
        if (__ballot(walking < 0) == 0ull) { if (exhausted && wcur == wend) break; continue; }
        /* ---- candidate steps until enough lanes need service */
        const uint32_t full_now = best >= 3u ? (best | (bdist << 16)) : 0u;   /* best is fixed while walking */
        const uint32_t pout = p - t0;
        const int32_t lim = (int32_t)(p < ZC_MAX_DIST ? p : ZC_MAX_DIST);      /* stop once the distance reaches this */
        /* the loop runs on c = candidate + (best - 2) - la0, the window offset of the candidate's bytes best-2.. */
        const uint32_t peek_off = (best - 2u) - la0;
        const uint16_t *link = s_pd - (best - 2u);   /* link[c] is the link of the candidate */
        uint32_t c = cur + peek_off;
        const int32_t plim = (int32_t)p - lim + (int32_t)peek_off;   /* the next candidate c' is in reach while plim - c' < 0 */
        /* a walk that has stopped stays stopped without being told (same peek, same link, same distance, count frozen);
         * lanes that are not walking get a count that says "budget spent" */
        int32_t wm = walking;
        int32_t kc = kk & walking;
        uint32_t xlast = 1u;
#pragma unroll 1
        for (int it = 0; it < 8; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint2 q = *(const uint2 *)(s_b + ((c & ~3u) << 1));      /* the slot of the dword that holds byte c */
                const uint32_t v = __builtin_amdgcn_alignbyte(q.y, q.x, c);
                const int32_t d = (int32_t)link[c];
                const uint32_t x = (v & 0x00ffffffu) ^ scan3;
                const int32_t nohit = -(int32_t)x;               /* negative unless the peek matched */
                xlast = x;
                kc -= wm;                                         /* count the candidate */
                const int32_t go = ((plim - (int32_t)c) + d) & kc & nohit;   /* negative iff the walk continues */
                wm = go >> 31;
                c += (uint32_t)__mul24(d, wm);                    /* c -= d where the walk continues */
                asm("" : "+v"(c));
            }
            if (__popcll(__ballot(wm == 0)) >= THRESH) break;
        }
        if (walking < 0) kk = kc;
        cur = c - peek_off;
        parked = xlast == 0u ? (walking & ~wm) : 0;
        if ((walking & ~wm & ~parked) < 0) {
            uint32_t k = (uint32_t)(kk + (int32_t)max_chain);
            uint32_t rq = qb == 0 ? 0u : (k > qb ? qrec : full_now);
            rec[pout] = make_uint2(full_now, rq);
        }
        walking = wm | parked;                /* parked walks are still owned by the lane */
    }
}
