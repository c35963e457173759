/* Match search over four-byte chains (levels whose chain budget fits eight bits).
 *
 * longest_match() (deflate.c:1148-1289) walks "all earlier strings with the hash of p, newest first" and, after
 * the first real match, rejects almost every one of them on the byte behind the best length so far.  Which
 * candidates can change the result is known without the parse: a candidate matters only if it matches MORE than
 * the best so far, so after the first true three-byte match (c0) only strings that agree with p in four bytes
 * matter.  Two kernels use that:
 *
 *   k_link4   for every string p: the nearest earlier member of p's hash chain that agrees with p in four bytes
 *             (link4, a distance) and the number of hash-chain steps that lie between them (skip) — the chain
 *             budget of the reference counts those steps — plus, for p's own search, the first member that
 *             agrees in three bytes (distance and step number of c0).  One short walk along the hash chain per
 *             string, ended by the first four-byte agreement or the budget.
 *   k_match4  the search proper, along link4 with the skipped steps added up: the candidates it examines are
 *             exactly the members of the reference's walk that agree with p in four bytes, in the same order,
 *             with their step numbers; its records (best within the whole budget, best within a quarter of it)
 *             are those of k_match.  If no four-byte candidate is in reach the answer is c0 with length 3.
 *
 * Both follow the rules of zc_match_init / zc_match_advance (zamd_core.h) for what ends a walk: the head
 * candidate may lie MAX_DIST back, later ones less; window index 0 is NIL; `chain` candidates at most.
 * A link is a fact about the input alone (it does not depend on the chunk a string is seen from), so the links of a
 * chunk's dictionary are those its predecessor wrote; the first chunk of a batch computes its dictionary's itself. */
#define ZK4_TILE     8192u
#define ZK4_HALO     32512u                   /* MAX_DIST rounded up to a multiple of 256 */
#define ZK4_WIN      (ZK4_HALO + ZK4_TILE + 320u)
#define ZK4_NONE     0xffffu
#define ZK4_NOSEARCH 0xffu
#define ZK4_MAX_CHAIN 254u

template <int THRESH>
__global__ __launch_bounds__(ZK_MATCH_THREADS) void k_link4(ZkJob J, uint32_t tiles_per_chunk)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_lds[ZK_LDS_BYTES / 4];
    __shared__ uint32_t s_next;
    uint32_t *s_win = s_lds;
    uint16_t *s_pd = (uint16_t *)(s_lds + ZK_PD_OFF / 4);
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t lc = blockIdx.x / tiles_per_chunk, tile = blockIdx.x % tiles_per_chunk;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head);
    const uint32_t E = g.dict_len + g.len;
    const uint32_t from = lc == 0 ? 0u : g.dict_len;
    if ((uint64_t)from + (uint64_t)tile * ZK_TILE >= E) return;
    const uint32_t t0 = from + tile * ZK_TILE;
    const uint32_t tlen = (E - t0) < ZK_TILE ? (E - t0) : ZK_TILE;
    const int64_t a0 = (int64_t)t0 - (int64_t)ZK_HALO;
    zk_load_window(J, g, a0, s_win, ZK_WIN_BYTES, tid);
    zk_load_links(J, g, a0, s_pd, ZK_HALO + ZK_TILE, tid);
    if (tid == 0) s_next = 0;
    __syncthreads();
    LdsWin win; win.w = s_win; win.a0 = (uint32_t)a0;
    LdsPd pd; pd.pd = s_pd; pd.a0 = (uint32_t)a0;
    const uint64_t obase = g.dstart + t0 - J.pdist_base;
    uint16_t *o_l4 = J.l4 + obase, *o_f3d = J.f3d + obase;
    uint8_t *o_sk = J.sk + obase, *o_f3s = J.f3s + obase;
    const uint32_t max_chain = J.P.chain;
    const uint32_t la0 = (uint32_t)a0;
    uint32_t p = t0, cur = t0, scan4 = 0, msk = 0x00ffffffu, f3 = 0;
    bool own = false;
    int32_t kk = 0, walking = 0, parked = 0;
    uint32_t wcur = 0, wend = 0;
    bool exhausted = false;
    auto finish = [&](uint32_t l4, uint32_t skip) {
        const uint32_t i = p - t0;
        o_l4[i] = (uint16_t)l4; o_sk[i] = (uint8_t)skip;
        o_f3d[i] = (uint16_t)(f3 & 0xffffu); o_f3s[i] = (uint8_t)(f3 ? (f3 >> 16) : ZK4_NOSEARCH);
    };
    for (;;) {
        /* ---- walks that stopped on a candidate agreeing with p under the lane's mask */
        if (parked < 0) {
            const uint32_t x = win.load4(cur) ^ scan4;
            const uint32_t k = (uint32_t)(kk + (int32_t)max_chain);      /* the step that stopped here counted the candidate */
            if (msk == 0x00ffffffu) {                                     /* first true three-byte match: c0 of p's own search */
                if (own) f3 = (p - cur) | (k << 16);
                msk = 0xffffffffu;
            }
            bool fin = false;
            uint32_t l4 = ZK4_NONE, skip = 255u;
            if (x == 0u) { l4 = p - cur; skip = k; fin = true; }
            else {
                const uint32_t d = pd(cur);
                if (kk >= 0 || d == 0 || d >= cur) fin = true;
                else { cur -= d; fin = p - cur >= ZC_MAX_DIST; }
            }
            parked = 0; walking = -1;
            if (fin) { finish(l4, skip); walking = 0; }
        }
        /* ---- refill idle lanes */
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
                p = t0 + wcur + rank;
                const uint32_t d0 = pd(p);
                f3 = 0; msk = 0x00ffffffu;
                const bool ok = d0 != 0 && d0 <= ZC_MAX_DIST && d0 < p && !(p == g.corner && d0 == ZC_MAX_DIST);
                if (ok) {
                    own = E - p >= ZC_MIN_MATCH;
                    cur = p - d0; kk = -(int32_t)max_chain; scan4 = win.load4(p);
                    walking = -1;
                } else finish(ZK4_NONE, 255u);
            }
            wcur += take;
        }
        if (__ballot(walking < 0) == 0ull) { if (exhausted && wcur == wend) break; continue; }
        /* ---- chain steps, as in k_match: straight-line arithmetic on sign masks */
        const int32_t lim = (int32_t)(p < ZC_MAX_DIST ? p : ZC_MAX_DIST);
        const uint32_t peek_off = 0u - la0;
        const uint16_t *link = s_pd - la0;
        uint32_t c = cur;
        const int32_t plim = (int32_t)p - lim;
        int32_t wm = walking;
        uint32_t xlast = 1u;
#pragma unroll 1
        for (int it = 0; it < 8; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t ac = c + peek_off;
                const uint32_t wi = ac >> 2;
                const uint32_t v = __builtin_amdgcn_alignbyte(s_win[wi + 1], s_win[wi], ac);
                const int32_t d = (int32_t)link[c];
                const uint32_t x = (v ^ scan4) & msk;
                const int32_t nohit = (int32_t)(x | (0u - x));
                xlast = x;
                kk -= wm;
                const int32_t go = ((plim - (int32_t)c) + d) & kk & nohit & wm;
                c = go < 0 ? c - (uint32_t)d : c;
                wm = go >> 31;
            }
            if (__popcll(__ballot(wm == 0)) >= THRESH) break;
        }
        cur = c;
        parked = xlast == 0u ? (walking & ~wm) : 0;
        if ((walking & ~wm & ~parked) < 0) finish(ZK4_NONE, 255u);
        walking = wm | parked;
    }
}

#define ZK4_LDS_BYTES (ZK4_WIN + 3u * (ZK4_HALO + ZK4_TILE))

template <int THRESH>
__global__ __launch_bounds__(ZK_MATCH_THREADS) void k_match4(ZkJob J, uint32_t tiles_per_chunk)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_lds[ZK4_LDS_BYTES / 4];
    __shared__ uint32_t s_next;
    uint32_t *s_win = s_lds;
    uint16_t *s_l4 = (uint16_t *)(s_lds + ZK4_WIN / 4);
    uint8_t *s_sk = (uint8_t *)(s_l4 + (ZK4_HALO + ZK4_TILE));
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t lc = blockIdx.x / tiles_per_chunk, tile = blockIdx.x % tiles_per_chunk;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head);
    if ((uint64_t)tile * ZK4_TILE >= g.len) return;
    const uint32_t E = g.dict_len + g.len;
    const uint32_t t0 = g.dict_len + tile * ZK4_TILE;
    const uint32_t tlen = (E - t0) < ZK4_TILE ? (E - t0) : ZK4_TILE;
    const int64_t a0 = (int64_t)t0 - (int64_t)ZK4_HALO;
    zk_load_window(J, g, a0, s_win, ZK4_WIN, tid);
    for (uint32_t v = tid; v < (ZK4_HALO + ZK4_TILE) / 8; v += ZK_MATCH_THREADS) {
        const int64_t a = a0 + 8 * (int64_t)v;
        uint4 q = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
        uint2 k = make_uint2(0xffffffffu, 0xffffffffu);
        if (a >= 0) {
            const uint64_t idx = g.dstart + (uint64_t)a - J.pdist_base;
            if (idx + 8 <= J.pdist_len) { q = *(const uint4 *)(J.l4 + idx); k = *(const uint2 *)(J.sk + idx); }
        }
        ((uint4 *)s_l4)[v] = q;
        ((uint2 *)s_sk)[v] = k;
    }
    if (tid == 0) s_next = 0;
    __syncthreads();
    LdsWin win; win.w = s_win; win.a0 = (uint32_t)a0;
    uint2 *rec = J.rec + (g.dstart + t0 - J.rec_base);
    const uint16_t *g_f3d = J.f3d + (g.dstart + t0 - J.pdist_base);
    const uint8_t *g_f3s = J.f3s + (g.dstart + t0 - J.pdist_base);
    const uint32_t nice_match = J.P.nice, max_chain = J.P.chain, qb = max_chain >> 2;
    const uint32_t la0 = (uint32_t)a0;
    const uint16_t *link = s_l4 - la0;
    const uint8_t *skp = s_sk - la0;
    uint32_t p = t0, cur = t0, best = 3, bdist = 0, qrec = 0, scan3 = 0, cap = 0, nice = 0, c0rec = 0, c0q = 0;
    int32_t kk = 0;                           /* chain steps so far minus (max_chain + 1): negative while the candidate is within the budget */
    int32_t walking = 0, parked = 0;
    uint32_t wcur = 0, wend = 0;
    bool exhausted = false;
    for (;;) {
        /* ---- refill idle lanes; a fresh walk starts parked on its first four-byte candidate */
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
                const uint32_t q = t0 + idx;
                const uint32_t s3 = g_f3s[idx], d3 = g_f3d[idx];
                const uint32_t look = E - q;
                uint2 r = make_uint2(0u, 0u);
                if (s3 != ZK4_NOSEARCH && s3 <= max_chain) {
                    const uint32_t r3 = 3u | (d3 << 16);
                    const uint32_t r3q = s3 <= qb ? r3 : 0u;
                    r = make_uint2(r3, r3q);
                    const uint32_t d = link[q];
                    const uint32_t s = skp[q];
                    const uint32_t lim = q < ZC_MAX_DIST ? q : ZC_MAX_DIST;
                    if (look < 4u) {}                                      /* nice length 3: the walk ends on c0 */
                    else if (s3 == 1u && d3 == ZC_MAX_DIST) {
                        /* the head candidate at exactly MAX_DIST: it is the only one in reach */
                        const uint32_t cp = look < ZC_MAX_MATCH ? look : ZC_MAX_MATCH;
                        uint32_t len = 0;
                        while (len < cp) {
                            uint32_t x = win.load4(q - d3 + len) ^ win.load4(q + len);
                            if (x) { len += zc_ctz32(x) >> 3; break; }
                            len += 4;
                        }
                        if (len > cp) len = cp;
                        r = make_uint2(len | (d3 << 16), len | (d3 << 16));
                    }
                    else if (d != ZK4_NONE && s <= max_chain && d < lim) {
                        p = q; cur = q - d; kk = (int32_t)s - (int32_t)max_chain - 1;
                        best = 3; bdist = 0; qrec = 0; c0rec = r3; c0q = r3q;
                        cap = look < ZC_MAX_MATCH ? look : ZC_MAX_MATCH;
                        nice = nice_match < look ? nice_match : look;
                        scan3 = win.load4(q + 1u) & 0x00ffffffu;
                        walking = -1; parked = -1;                         /* a four-byte candidate always beats length 3 */
                    }
                }
                if (walking == 0) rec[idx] = r;
            }
            wcur += take;
        }
        /* ---- full comparisons of parked candidates */
        if (parked < 0) {
            uint32_t len = 0;
            while (len < cap) {
                uint32_t x = win.load4(cur + len) ^ win.load4(p + len);
                if (x) { len += zc_ctz32(x) >> 3; break; }
                len += 4;
            }
            if (len > cap) len = cap;
            bool fin = false;
            const uint32_t k = (uint32_t)(kk + (int32_t)max_chain + 1);   /* step number of this candidate in the reference's walk */
            if (len > best) {
                best = len; bdist = p - cur;
                if (k <= qb) qrec = best | (bdist << 16);
                if (len >= nice) fin = true;
                else scan3 = win.load4(p + len - 2u) & 0x00ffffffu;
            }
            if (!fin) {
                const uint32_t d = link[cur];
                const int32_t k2 = kk + (int32_t)skp[cur];
                if (k2 >= 0 || d == ZK4_NONE || d >= cur) fin = true;
                else { cur -= d; kk = k2; fin = p - cur >= ZC_MAX_DIST; }
            }
            parked = 0; walking = -1;
            if (fin) {
                rec[p - t0] = make_uint2(bdist ? (best | (bdist << 16)) : c0rec, qrec ? qrec : c0q);
                walking = 0;
            }
        }
        if (__ballot(walking < 0) == 0ull) { if (exhausted && wcur == wend) break; continue; }
        /* ---- candidate steps */
        const int32_t lim = (int32_t)(p < ZC_MAX_DIST ? p : ZC_MAX_DIST);
        const uint32_t peek_off = (best - 2u) - la0;
        uint32_t c = cur;
        const int32_t plim = (int32_t)p - lim;
        int32_t wm = walking & ~parked;
        const int32_t wstart = wm;
        uint32_t xlast = 1u;
#pragma unroll 1
        for (int it = 0; it < 8; it++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t ac = c + peek_off;
                const uint32_t wi = ac >> 2;
                const uint32_t v = __builtin_amdgcn_alignbyte(s_win[wi + 1], s_win[wi], ac);
                const int32_t d = (int32_t)link[c];
                const int32_t k2 = kk + (int32_t)skp[c];
                const uint32_t x = (v & 0x00ffffffu) ^ scan3;
                const int32_t nohit = -(int32_t)x;
                xlast = x;
                const int32_t go = ((plim - (int32_t)c) + d) & k2 & nohit & wm;
                c = go < 0 ? c - (uint32_t)d : c;
                kk = go < 0 ? k2 : kk;
                wm = go >> 31;
            }
            if (__popcll(__ballot(wm == 0)) >= THRESH) break;
        }
        cur = c;
        const int32_t newly = xlast == 0u ? (wstart & ~wm) : 0;
        if ((wstart & ~wm & ~newly) < 0)
            rec[p - t0] = make_uint2(bdist ? (best | (bdist << 16)) : c0rec, qrec ? qrec : c0q);
        parked |= newly;
        walking = wm | parked;
    }
}
