/* k_match2: k_match with TWO chain walks per lane.
 *
 * k_match keeps one workgroup of 16 wavefronts on a CU (its LDS image of window bytes and links leaves no room
 * for a second), four per SIMD, and every candidate step of a walk depends on the one before it (the link read
 * gives the next address): with four wavefronts to switch between, a SIMD waits on LDS round trips for much of
 * the time, and neither the vector unit nor the LDS array is busy (measured: about a third and two thirds).  Here
 * every lane owns two independent walks, A and B, and the candidate loop advances both in each step, so that the
 * LDS reads of one are in flight while the other's are consumed.  Everything else — what a step decides, parking
 * of candidates that pass the three-byte peek, the records written — is k_match's, statement for statement
 * (zc_match_init / zc_match_peek / zc_match_extend / zc_match_advance of zamd_core.h, deflate.c:1148-1289).
 * Included by zamd_engine.hip behind k_match. */

struct ZkWalk {
    uint32_t p, cur, best, bdist, qrec, scan3, cap, nice;
    int32_t kk;                      /* candidates examined minus max_chain: negative while budget remains */
    int32_t walking, parked;         /* sign masks: -1 / 0 */
};

template <int THRESH>
__global__ __launch_bounds__(ZK_MATCH_THREADS) void k_match2(ZkJob J, uint32_t tiles_per_chunk)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_lds[ZK_LDS_BYTES / 4];
    __shared__ uint32_t s_next;
    uint32_t *s_win = s_lds;
    uint16_t *s_pd = (uint16_t *)(s_lds + ZK_PD_OFF / 4);
    const uint32_t tid = threadIdx.x;
    const uint32_t lc = blockIdx.x / tiles_per_chunk, tile = blockIdx.x % tiles_per_chunk;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head);
    if ((uint64_t)tile * ZK_TILE >= g.len) return;
    const uint32_t E = g.dict_len + g.len;
    const uint32_t t0 = g.dict_len + tile * ZK_TILE;
    const uint32_t tlen = (E - t0) < ZK_TILE ? (E - t0) : ZK_TILE;
    const int64_t a0 = (int64_t)t0 - (int64_t)ZK_HALO;      /* a-coordinate of LDS byte 0 */
    zk_load_window(J, g, a0, s_win, ZK_WIN_BYTES, tid);
    zk_load_links(J, g, a0, s_pd, ZK_HALO + ZK_TILE, tid);
    if (tid == 0) s_next = 0;
    __syncthreads();
    LdsWin win; win.w = s_win; win.a0 = (uint32_t)a0;
    LdsPd pd; pd.pd = s_pd; pd.a0 = (uint32_t)a0;
    uint2 *rec = J.rec + (g.dstart + t0 - J.rec_base);
    const uint32_t lane = tid & 63u;
    const uint32_t nice_match = J.P.nice, max_chain = J.P.chain, qb = max_chain >> 2;
    const uint32_t la0 = (uint32_t)a0;
    ZkWalk A, B;
    A.p = A.cur = t0; A.best = 2; A.bdist = A.qrec = A.scan3 = A.cap = A.nice = 0; A.kk = 0; A.walking = A.parked = 0;
    B = A;
    uint32_t wcur = 0, wend = 0;              /* this wave's ticket of unclaimed tile positions */
    bool exhausted = false;

    /* full comparison of a parked candidate (zc_match_extend, then zc_match_advance) */
    auto service = [&](ZkWalk &w) {
        if (w.parked < 0) {
            uint32_t len, dnext;
            {
                const uint32_t oc = w.cur - la0, op = w.p - la0, ic = oc >> 2, ip = op >> 2;
                dnext = pd(w.cur);
                const uint32_t c0 = s_win[ic], c1 = s_win[ic + 1], c2 = s_win[ic + 2], c3 = s_win[ic + 3], c4 = s_win[ic + 4];
                const uint32_t q0 = s_win[ip], q1 = s_win[ip + 1], q2 = s_win[ip + 2], q3 = s_win[ip + 3], q4 = s_win[ip + 4];
                const uint32_t x0 = __builtin_amdgcn_alignbyte(c1, c0, oc) ^ __builtin_amdgcn_alignbyte(q1, q0, op);
                const uint32_t x1 = __builtin_amdgcn_alignbyte(c2, c1, oc) ^ __builtin_amdgcn_alignbyte(q2, q1, op);
                const uint32_t x2 = __builtin_amdgcn_alignbyte(c3, c2, oc) ^ __builtin_amdgcn_alignbyte(q3, q2, op);
                const uint32_t x3 = __builtin_amdgcn_alignbyte(c4, c3, oc) ^ __builtin_amdgcn_alignbyte(q4, q3, op);
                len = x0 ? zc_ctz32(x0) >> 3 : x1 ? 4u + (zc_ctz32(x1) >> 3) : x2 ? 8u + (zc_ctz32(x2) >> 3) : x3 ? 12u + (zc_ctz32(x3) >> 3) : 16u;
            }
            if (len == 16u) {
                while (len < w.cap) {
                    uint32_t x = win.load4(w.cur + len) ^ win.load4(w.p + len);
                    if (x) { len += zc_ctz32(x) >> 3; break; }
                    len += 4;
                }
            }
            if (len > w.cap) len = w.cap;
            bool fin = false;
            const uint32_t k = (uint32_t)(w.kk + (int32_t)max_chain);     /* the step that parked the walk counted the candidate */
            if (len > w.best) {
                w.best = len; w.bdist = w.p - w.cur;
                if (k <= qb) w.qrec = w.best | (w.bdist << 16);
                if (len >= w.nice) fin = true;
                else w.scan3 = win.load4(w.p + len - 2u) & 0x00ffffffu;
            }
            if (!fin) {
                const uint32_t d = dnext;
                if (w.kk >= 0 || d == 0 || d >= w.cur) fin = true;
                else { w.cur -= d; fin = w.p - w.cur >= ZC_MAX_DIST; }
            }
            w.parked = 0; w.walking = -1;
            if (fin) {
                uint32_t full = w.best >= 3u ? (w.best | (w.bdist << 16)) : 0u;
                uint32_t rq = qb == 0 ? 0u : (k > qb ? w.qrec : full);
                rec[w.p - t0] = make_uint2(full, rq);
                w.walking = 0;
            }
        }
    };
    /* idle walks of this kind draw positions from the wave's ticket (64 positions per LDS atomic) */
    auto refill = [&](ZkWalk &w) {
        for (int pass = 0; pass < 2; pass++) {
            unsigned long long idle = __ballot(w.walking == 0);
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
            if (w.walking == 0 && rank < take) {
                const uint32_t idx = wcur + rank;
                ZcMatchState st;
                if (zc_match_init(win, pd, t0 + idx, E, g.corner, nice_match, st)) {
                    w.p = st.p; w.cur = st.cur; w.kk = -(int32_t)max_chain; w.best = 2; w.bdist = 0; w.qrec = 0;
                    w.scan3 = st.scan3; w.cap = st.cap; w.nice = st.nice;
                    w.walking = -1;
                } else rec[idx] = make_uint2(0u, 0u);
            }
            wcur += take;
        }
    };
    /* what the candidate loop leaves behind: parked if the walk stands on a candidate that passed the peek,
     * otherwise, if it stopped, its record */
    auto settle = [&](ZkWalk &w, uint32_t c, uint32_t peek_off, int32_t kc, int32_t wm, uint32_t xlast) {
        const uint32_t full_now = w.best >= 3u ? (w.best | (w.bdist << 16)) : 0u;
        if (w.walking < 0) w.kk = kc;
        w.cur = c - peek_off;
        w.parked = xlast == 0u ? (w.walking & ~wm) : 0;
        if ((w.walking & ~wm & ~w.parked) < 0) {
            uint32_t k = (uint32_t)(w.kk + (int32_t)max_chain);
            uint32_t rq = qb == 0 ? 0u : (k > qb ? w.qrec : full_now);
            rec[w.p - t0] = make_uint2(full_now, rq);
        }
        w.walking = wm | w.parked;
    };

    for (;;) {
        service(A); service(B);
        refill(A); refill(B);
        if (__ballot((A.walking | B.walking) < 0) == 0ull) { if (exhausted && wcur == wend) break; continue; }
        /* ---- candidate steps of both walks until enough of them need service; the step is k_match's */
        const uint32_t offA = (A.best - 2u) - la0, offB = (B.best - 2u) - la0;
        const uint16_t *linkA = s_pd - (A.best - 2u), *linkB = s_pd - (B.best - 2u);
        uint32_t cA = A.cur + offA, cB = B.cur + offB;
        const int32_t plimA = (int32_t)A.p - (int32_t)(A.p < ZC_MAX_DIST ? A.p : ZC_MAX_DIST) + (int32_t)offA;
        const int32_t plimB = (int32_t)B.p - (int32_t)(B.p < ZC_MAX_DIST ? B.p : ZC_MAX_DIST) + (int32_t)offB;
        const uint32_t scanA = A.scan3, scanB = B.scan3;
        int32_t wmA = A.walking, wmB = B.walking;
        int32_t kcA = A.kk & A.walking, kcB = B.kk & B.walking;
        uint32_t xA = 1u, xB = 1u;
#pragma unroll 1
        for (int it = 0; it < 8; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const uint32_t iA = cA >> 2, iB = cB >> 2;
                const uint32_t vA = __builtin_amdgcn_alignbyte(s_win[iA + 1], s_win[iA], cA);
                const int32_t dA = (int32_t)linkA[cA];
                const uint32_t vB = __builtin_amdgcn_alignbyte(s_win[iB + 1], s_win[iB], cB);
                const int32_t dB = (int32_t)linkB[cB];
                xA = (vA & 0x00ffffffu) ^ scanA;
                xB = (vB & 0x00ffffffu) ^ scanB;
                kcA -= wmA; kcB -= wmB;
                const int32_t goA = ((plimA - (int32_t)cA) + dA) & kcA & -(int32_t)xA;
                const int32_t goB = ((plimB - (int32_t)cB) + dB) & kcB & -(int32_t)xB;
                wmA = goA >> 31; wmB = goB >> 31;
                cA += (uint32_t)__mul24(dA, wmA);
                cB += (uint32_t)__mul24(dB, wmB);
                asm("" : "+v"(cA));
                asm("" : "+v"(cB));
            }
            if (__popcll(__ballot(wmA == 0)) + __popcll(__ballot(wmB == 0)) >= 2 * THRESH) break;
        }
        settle(A, cA, offA, kcA, wmA, xA);
        settle(B, cB, offB, kcB, wmB, xB);
    }
}
