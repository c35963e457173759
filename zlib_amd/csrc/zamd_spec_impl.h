/*
 * zamd_spec_impl.h — k_spec: match search driven by the parse (levels 4-9).
 *
 * deflate_slow (deflate.c:1730-1853) calls longest_match only where its parse
 * arrives: not inside a match it has taken, with a quarter of the chain budget
 * once the previous match is good enough, not at all once it is long enough.  On
 * compressible data that is a fifth of the chain steps a search at every position
 * costs.  Which positions are examined is a serial property of the chunk, so the
 * kernel guesses: parses start at every ZK_SPEC_SEG-th position of a tile as if
 * nothing were pending there (from such a state the next step, a "hop": k lazy
 * literals and a match, or one literal, depends on the data alone, see
 * zc_hop_direct in zamd_core.h) and follow their own hops.  Parses that start
 * inside a match of the real parse fall in with it within a hop or two; a bitmap
 * of the positions some parse has been to stops the later arrival, so no hop is
 * computed twice.  The real parse of the chunk is then the chain of recorded hops
 * from the chunk's first byte (k_hop), which computes the rare hop nobody
 * recorded (a parse that crosses into the next tile and has not met that tile's
 * parses after ZK_SPEC_OVER positions) by itself.
 *
 * Measured on MI355X this is slower than searching every position (k_match): a
 * tile that fits the LDS gives each of its 1024 lanes sixteen positions, so every
 * lane owns one short serial parse and a wavefront waits for its longest one,
 * while k_match deals single positions to whichever lane is free.  The engine
 * therefore uses k_spec only where k_match's precomputed records cannot express
 * the search (deflateTune with nice_length < max_lazy or max_chain < 4), or on
 * request (ZAMD_MATCH_SPEC=1).
 *
 * One lane = one parse; the chain walk of a lane is the loop of k_match (same
 * window and link images in LDS, same straight-line candidate step, candidates
 * that pass the three-byte peek parked for a joint full comparison).
 * Included by zamd_engine.hip after k_match.
 */
#ifndef ZAMD_SPEC_IMPL_H
#define ZAMD_SPEC_IMPL_H

#define ZK_SPEC_OVER  256u                                   /* positions past the tile a parse may still process */
#define ZK_SPEC_SEG   16u                                    /* distance of the speculative starting points */
#define ZK_SPEC_SPAN  (ZK_TILE + ZK_SPEC_OVER)
#define ZK_SPEC_WIN   (ZK_HALO + ZK_SPEC_SPAN + 640u)        /* window bytes: + 255 lazy steps + 258 + compare slack */
#define ZK_SPEC_PD    (ZK_HALO + ZK_SPEC_SPAN + 264u)        /* links */
#define ZK_SPEC_LDS   (ZK_SPEC_WIN + 2u * ZK_SPEC_PD)

enum { ZS_ACQ = 0, ZS_RES = 2, ZS_DONE = 3 };
#ifdef ZAMD_SPEC_STATS
__device__ unsigned long long g_spec_stats[8];
#define ZS_STAT(i, v) (st_##i += (v))
#else
#define ZS_STAT(i, v) ((void)0)
#endif

template <int THRESH>
__global__ __launch_bounds__(ZK_MATCH_THREADS) void k_spec(ZkJob J, uint32_t tiles_per_chunk, uint32_t SEGV)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_lds[ZK_SPEC_LDS / 4];
    __shared__ uint32_t s_known[ZK_SPEC_SPAN / 32 + 1];
    __shared__ uint32_t s_next;
    uint32_t *s_win = s_lds;
    uint16_t *s_pd = (uint16_t *)(s_lds + ZK_SPEC_WIN / 4);
    const uint32_t tid = threadIdx.x;
    const uint32_t lc = blockIdx.x / tiles_per_chunk, tile = blockIdx.x % tiles_per_chunk;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    if ((uint64_t)tile * ZK_TILE >= g.len) return;
    const uint32_t E = g.dict_len + g.len;
    const uint32_t t0 = g.dict_len + tile * ZK_TILE;
    const uint32_t tlen = (E - t0) < ZK_TILE ? (E - t0) : ZK_TILE;
    const uint32_t limit_end = (E - t0) < tlen + ZK_SPEC_OVER ? E : t0 + tlen + ZK_SPEC_OVER;
    const uint32_t nseg = (tlen + SEGV - 1u) / SEGV;
    const int64_t a0 = (int64_t)t0 - (int64_t)ZK_HALO;      /* a-coordinate of LDS byte 0 */
    for (uint32_t v = tid; v < ZK_SPEC_WIN / 16; v += blockDim.x) {
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
    /* chain links; "none" (0) is stored as 0xffff so that adding it to a distance always exceeds MAX_DIST */
    for (uint32_t v = tid; v < ZK_SPEC_PD / 8; v += blockDim.x) {
        int64_t a = a0 + 8 * (int64_t)v;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (a >= 0) {
            uint64_t idx = g.dstart + (uint64_t)a - J.pdist_base;
            if (idx + 8 <= J.pdist_len) q = *(const uint4 *)(J.pdist + idx);
            else if (idx < J.pdist_len) {
                uint32_t t[4] = {0, 0, 0, 0};
                for (uint32_t k = 0; k < 8 && idx + k < J.pdist_len; k++) t[k >> 1] |= (uint32_t)J.pdist[idx + k] << (16 * (k & 1));
                q = make_uint4(t[0], t[1], t[2], t[3]);
            }
        }
        uint32_t t[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((t[k] & 0xffffu) == 0) t[k] |= 0xffffu;
            if ((t[k] >> 16) == 0) t[k] |= 0xffff0000u;
        }
        ((uint4 *)s_pd)[v] = make_uint4(t[0], t[1], t[2], t[3]);
    }
    for (uint32_t v = tid; v < ZK_SPEC_SPAN / 32 + 1; v += blockDim.x) s_known[v] = 0;
    if (tid == 0) s_next = 0;
    __syncthreads();
    LdsWin win; win.w = s_win; win.a0 = (uint32_t)a0;
    LdsPd pd; pd.pd = s_pd; pd.a0 = (uint32_t)a0;
    uint32_t *hop = J.hop + (uint64_t)lc * J.P.chunk_size;      /* index = a-coordinate - dict_len */
    const uint32_t nice_match = J.P.nice, max_chain = J.P.chain, good = J.P.good, lazy = J.P.lazy, max_dist = J.P.max_dist;
    const uint32_t qbudget = (max_chain >> 2) ? (max_chain >> 2) : 0x7fffffffu;   /* zero wraps in the reference's do-while */
    const bool filt = J.P.strategy == ZC_STRAT_FILTERED;
    const uint32_t la0 = (uint32_t)a0;
    /* the parse of this lane: the hop in progress started at hp with nothing pending and has so far
     * k lazy literals and the match (L, D); L == 0: the first search, at hp itself, is under way */
    uint32_t hp = 0, k = 0, L = 0, D = 0, st = ZS_ACQ, reach = 0;
    /* the walk of this lane */
    uint32_t p = t0, cur = t0, best = 2, plen = 2, bdist = 0, scan3 = 0, cap = 0, nice = 0;
    int32_t kk = 0;                           /* candidates examined minus the budget: negative while budget remains */
    int32_t walking = 0, parked = 0;          /* sign masks: -1 / 0 */

#ifdef ZAMD_SPEC_STATS
    unsigned long long st_0 = 0, st_1 = 0, st_2 = 0, st_3 = 0, st_4 = 0, st_5 = 0, st_6 = 0, st_7 = 0;
    const long long t_begin = clock64();
#endif
    auto start_walk = [&](uint32_t q, uint32_t pl) -> bool {
        ZcMatchState ms;
        if (!zc_match_init(win, pd, q, E, g.corner, nice_match, ms, max_dist)) return false;
        if (pl >= ms.cap) return false;
        ZS_STAT(2, 1);
        p = q; cur = ms.cur; cap = ms.cap; nice = ms.nice; best = pl; plen = pl; bdist = 0;
        scan3 = pl > 2u ? (win.load4(q + pl - 2u) & 0x00ffffffu) : ms.scan3;
        kk = -(int32_t)(pl >= good ? qbudget : max_chain);
        walking = -1;
        return true;
    };

    for (;;) {
        /* ---- full comparisons of parked candidates */
        if (parked < 0) {
            ZS_STAT(6, 1);
            uint32_t len = 0;
            while (len < cap) {
                uint32_t x = win.load4(cur + len) ^ win.load4(p + len);
                if (x) { len += zc_ctz32(x) >> 3; break; }
                len += 4;
            }
            if (len > cap) len = cap;
            bool fin = false;
            if (len > best) {
                best = len; bdist = p - cur;
                if (len >= nice) fin = true;
                else scan3 = win.load4(p + len - 2u) & 0x00ffffffu;
            }
            if (!fin) {
                uint32_t d = pd(cur);
                if (kk >= 0 || d == 0 || d >= cur) fin = true;
                else { cur -= d; fin = p - cur >= max_dist; }
            }
            parked = 0; walking = fin ? 0 : -1;
        }
        /* ---- parse steps of the lanes whose walk is over, as straight-line code on predicates */
        for (int pass = 0; pass < 2; pass++) {
            const bool sv = walking == 0 && st != ZS_DONE;
            if (__ballot(sv) == 0ull) break;
            if ((tid & 63u) == 0) ZS_STAT(4, 1);
            /* 1. the search's result (deflate.c:1771-1815) */
            const bool res = sv && st == ZS_RES;
            const bool acc = res && best > plen && !(best <= 5u && (filt || (best == ZC_MIN_MATCH && bdist > ZC_TOO_FAR)));
            k += (acc && L != 0u) ? 1u : 0u;
            L = acc ? best : L; D = acc ? bdist : D;
            const uint32_t q1 = hp + k + 1u;
            const bool probe = acc && L < lazy && E - q1 >= ZC_MIN_MATCH;      /* the lazy search at the next position */
            const bool hop_done = res && !probe;
            /* 2. a complete hop is recorded */
            if (hop_done) hop[hp - g.dict_len] = L ? zc_hop_pack(k, L, D) : 1u;
            if (hop_done) ZS_STAT(3, 1);
            const uint32_t cnd = L ? hp + k + L : hp + 1u;
            /* 3. where the parse goes on: behind the hop unless somebody has been there or it is out of reach,
             *    else at the next speculative starting point */
            const bool acq = hop_done || (sv && st == ZS_ACQ);
            bool have = false; uint32_t nhp = 0;
            if (hop_done && cnd < reach) {
                const uint32_t i = cnd - t0;
                if (!((atomicOr(&s_known[i >> 5], 1u << (i & 31u)) >> (i & 31u)) & 1u)) { have = true; nhp = cnd; }
            }
            if (acq && !have) {
                const uint32_t seg = atomicAdd(&s_next, 1u);
                st = ZS_ACQ;
                if (seg >= nseg) st = ZS_DONE;
                else {
                    const uint32_t i = seg * SEGV;
                    if (!((atomicOr(&s_known[i >> 5], 1u << (i & 31u)) >> (i & 31u)) & 1u)) { have = true; nhp = t0 + i; }
                    /* a parse that has not met another one ZK_SPEC_OVER positions past its own stretch gives up:
                     * where matches are that long, searches are cheap and k_hop does the few it needs itself */
                    reach = t0 + i + SEGV + ZK_SPEC_OVER;
                    if (reach > limit_end) reach = limit_end;
                }
            }
            if (have) { hp = nhp; L = 0; k = 0; D = 0; st = ZS_RES; }
            /* 4. the next search: the lazy one with prev_length = L, or the first of a new hop with prev_length = 2
             *    (none at all unless max_lazy_match is above that, deflate.c:1765); a search that cannot start
             *    counts as one that found nothing */
            if (probe || have) {
                const uint32_t q = probe ? q1 : hp, pl = probe ? L : ZC_MIN_MATCH - 1u;
                if (!(lazy >= ZC_MIN_MATCH && start_walk(q, pl))) { best = pl; plen = pl; }
            }
        }
        if (__ballot(walking < 0) == 0ull) { if (__ballot(st != ZS_DONE) == 0ull) break; continue; }
        /* ---- candidate steps until enough lanes need service */
        const uint32_t live = (uint32_t)__popcll(__ballot(st != ZS_DONE));
        const uint32_t thr = live / 2u < (uint32_t)THRESH ? (live / 2u ? live / 2u : 1u) : (uint32_t)THRESH;
        const int32_t lim = (int32_t)(p < max_dist ? p : max_dist);      /* stop once the distance reaches this */
        const uint32_t peek_off = (best - 2u) - la0;
        const uint16_t *link = s_pd - la0;
        uint32_t c = cur;                     /* candidate */
        const int32_t plim = (int32_t)p - lim;       /* the next candidate c' is in reach while plim - c' < 0 */
        int32_t wm = walking;
#ifdef ZAMD_SPEC_STATS
        const long long t_w = clock64();
#endif
#pragma unroll 1
        for (int it = 0; it < 32; it++) {
#pragma unroll
            for (int u = 0; u < 2; u++) {
                ZS_STAT(0, wm < 0 ? 1 : 0); ZS_STAT(1, 1);
                const uint32_t ac = c + peek_off;
                const uint32_t wi = ac >> 2;
                const uint32_t v = __builtin_amdgcn_alignbyte(s_win[wi + 1], s_win[wi], ac & 3u);
                const int32_t d = (int32_t)link[c];
                const uint32_t x = (v & 0x00ffffffu) ^ scan3;
                const int32_t nohit = -(int32_t)x;               /* negative unless the peek matched */
                const int32_t hit = (int32_t)(x - 1u) & wm;      /* negative iff walking and matched */
                kk -= wm;                                         /* count the candidate */
                const int32_t go = ((plim - (int32_t)c) + d) & kk & nohit & wm;   /* negative iff the walk continues */
                c = go < 0 ? c - (uint32_t)d : c;
                parked |= hit;
                wm = go >> 31;
            }
            if ((uint32_t)__popcll(__ballot(wm == 0 && st != ZS_DONE)) >= thr) break;
        }
#ifdef ZAMD_SPEC_STATS
        if ((tid & 63u) == 0) st_5 += (unsigned long long)(clock64() - t_w);
#endif
        cur = c;
        parked >>= 31;
        walking = wm | parked;                /* parked walks are still owned by the lane */
    }
#ifdef ZAMD_SPEC_STATS
    if ((tid & 63u) == 0) st_7 = (unsigned long long)(clock64() - t_begin);
    atomicAdd(&g_spec_stats[0], st_0); atomicAdd(&g_spec_stats[1], st_1); atomicAdd(&g_spec_stats[2], st_2); atomicAdd(&g_spec_stats[3], st_3);
    atomicAdd(&g_spec_stats[4], st_4); atomicAdd(&g_spec_stats[5], st_5); atomicAdd(&g_spec_stats[6], st_6); atomicAdd(&g_spec_stats[7], st_7);
#endif
}
#ifdef ZAMD_SPEC_STATS
extern "C" int zamd_debug_spec_stats(unsigned long long *out8, int reset)
{
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_spec_stats), 64) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_spec_stats), z, 64) != hipSuccess) return -1; }
    return 0;
}
#endif

/* Accessors of k_hop's own searches: window bytes and links straight from global memory. */
struct GlbWin {
    const uint8_t *b; uint64_t n;             /* b = a-coordinate 0, n = bytes of input from there */
    __device__ uint32_t load4(uint32_t a) const
    {
        uint32_t v = 0;
        for (uint32_t k = 0; k < 4; k++) if ((uint64_t)a + k < n) v |= (uint32_t)b[a + k] << (8 * k);
        return v;
    }
};
struct GlbPd {
    const uint16_t *pd; uint64_t len;         /* pd = link of a-coordinate 0 */
    __device__ uint32_t operator()(uint32_t a) const { return a < len ? pd[a] : 0u; }
};

#endif /* ZAMD_SPEC_IMPL_H */
