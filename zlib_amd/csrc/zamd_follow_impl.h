/*
 * zamd_follow_impl.h — k_follow: match search driven by the parse (levels 4-9), one workgroup per chunk.
 *
 * deflate_slow (deflate.c:1730-1853) calls longest_match only where its parse arrives: not inside a match it has
 * taken, with a quarter of the chain budget once the previous match is good enough, not at all once it is long
 * enough.  On compressible data that is a fifth of the chain steps a search at every position costs.  Which positions
 * those are is a serial property of the chunk, so the kernel guesses: parses ("contexts") start at every ZF_SEG-th
 * position as if nothing were pending there, follow their own hops (a hop: k lazy literals and a match, or one
 * literal; from a state with nothing pending it depends on the data alone, zc_hop_direct in zamd_core.h), and stop
 * where another parse has been (a bitmap of visited hop starts), so no hop is computed twice.  The real parse is the
 * chain of recorded hops from the chunk's first byte (k_hop).
 *
 * The workgroup stays with its chunk and the LDS holds a RING of the chunk's newest ZF_R positions (window bytes and
 * chain links), loaded in blocks of ZF_BK.  A lane whose parse has ended takes the next stretch of the chunk; a block of the ring
 * is overwritten with the next block of the chunk as soon as no parse can look back into it: every search position
 * is at least 32 KiB + one block ahead of it.  That is tracked with a count of live search positions per block; a
 * lane that needs bytes which are not loaded yet waits (uncounted: it is far ahead of the blocks that matter).
 *
 * Ring coordinates: a-coordinate a lives at offset a % ZF_R; the first ZF_MIRROR bytes of the ring are repeated behind
 * its end, so that comparisons run off the end without wrapping; candidate offsets wrap in the walk itself.
 *
 * Measured on MI355X this is three times slower than searching every position (k_match): the ring holds 16 Ki
 * positions where searches can run, i.e. one stretch of ZF_SEG positions for each of the 1 024 lanes, so a lane that
 * has finished its stretch mostly finds the next one beyond the loaded data and waits (60 % of the lanes of a
 * candidate round, by the kernel's counters) until the slowest parse of the oldest block lets the ring move on; a
 * chain of dependent searches per lane is bound by latency where k_match is bound by throughput.  The engine
 * therefore uses k_follow only where k_match's precomputed records cannot express the search (deflateTune with
 * nice_length < max_lazy or max_chain < 4), or on request (ZAMD_MATCH_FOLLOW=1).
 *
 * One lane = one parse; the chain walk of a lane is the loop of k_match (same straight-line candidate step,
 * candidates that pass the three-byte peek parked for a joint full comparison).
 * Included by zamd_engine.hip after k_match.
 */
#ifndef ZAMD_FOLLOW_IMPL_H
#define ZAMD_FOLLOW_IMPL_H

#define ZF_R       49152u                 /* positions in the ring: 32 KiB of history + 16 KiB where searches run */
#define ZF_BK      4096u                  /* positions per block */
#define ZF_NB      (ZF_R / ZF_BK)
#define ZF_MIRROR  320u
#define ZF_SEG     16u                    /* distance of the speculative starting points */
#define ZF_LOOK    288u                   /* bytes behind a search position that must be loaded: MAX_MATCH + compare slack */
#define ZF_THREADS 1024
#define ZF_NOBLK   0xffffffffu
#define ZF_SPIN_CAP (1u << 21)            /* rounds a wavefront may wait without walking before it reports itself stuck */

enum { ZE_FOLLOW_STUCK = 32 };
#ifdef ZAMD_SPEC_STATS
__device__ unsigned long long g_spec_stats[8];
#define ZS_STAT(i, v) (st_##i += (v))
#else
#define ZS_STAT(i, v) ((void)0)
#endif
enum { ZFS_ACQ = 0, ZFS_WAIT = 1, ZFS_RES = 2, ZFS_DONE = 3 };

struct RingWin {
    const uint32_t *w;
    __device__ uint32_t at(uint32_t off) const { const uint32_t i = off >> 2; return __builtin_amdgcn_alignbyte(w[i + 1], w[i], off); }   /* ring offset */
    __device__ uint32_t load4(uint32_t a) const { return at(a % ZF_R); }                                                             /* a-coordinate */
};
struct RingPd {
    const uint16_t *pd;
    __device__ uint32_t operator()(uint32_t a) const { const uint32_t v = pd[a % ZF_R]; return v == ZK_PD_NONE ? 0u : v; }
};

/* One block of the chunk into the ring: bytes (zero outside the input), links ("none" as 0xffff), visited bits cleared. */
__device__ inline void zf_load_block(const ZkJob &J, const ZcChunkGeom &g, uint32_t a0, uint32_t *s_win, uint16_t *s_pd, uint32_t *s_known,
                                     uint32_t t, uint32_t nt)
{
    const uint32_t r0 = a0 % ZF_R;
    for (uint32_t v = t; v < ZF_BK / 16; v += nt) {
        const uint32_t a = a0 + 16u * v;
        const uint64_t abs = g.dstart + a;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (abs + 16 <= J.n) q = *(const uint4 *)(J.in + abs);
        else if (abs < J.n) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (uint32_t k = 0; k < 16 && abs + k < J.n; k++) w[k >> 2] |= (uint32_t)J.in[abs + k] << (8 * (k & 3));
            q = make_uint4(w[0], w[1], w[2], w[3]);
        }
        ((uint4 *)s_win)[(r0 >> 4) + v] = q;
        if (r0 == 0 && 16u * v < ZF_MIRROR) ((uint4 *)s_win)[(ZF_R >> 4) + v] = q;
    }
    for (uint32_t v = t; v < ZF_BK / 8; v += nt) {
        const uint64_t idx = g.dstart + a0 + 8u * v - J.pdist_base;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (idx + 8 <= J.pdist_len) q = *(const uint4 *)(J.pdist + idx);
        else if (idx < J.pdist_len) {
            uint32_t w[4] = {0, 0, 0, 0};
            for (uint32_t k = 0; k < 8 && idx + k < J.pdist_len; k++) w[k >> 1] |= (uint32_t)J.pdist[idx + k] << (16 * (k & 1));
            q = make_uint4(w[0], w[1], w[2], w[3]);
        }
        uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if ((w[k] & 0xffffu) == 0) w[k] |= 0xffffu;
            if ((w[k] >> 16) == 0) w[k] |= 0xffff0000u;
        }
        ((uint4 *)s_pd)[(r0 >> 3) + v] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    for (uint32_t v = t; v < ZF_BK / 32; v += nt) s_known[(r0 >> 5) + v] = 0;
}

__global__ __launch_bounds__(ZF_THREADS) void k_follow(ZkJob J)
{
    __shared__ __attribute__((aligned(16))) uint32_t s_win[(ZF_R + ZF_MIRROR) / 4];
    __shared__ __attribute__((aligned(16))) uint16_t s_pd[ZF_R];
    __shared__ uint32_t s_known[ZF_R / 32];
    __shared__ uint32_t s_cnt[ZF_NB];               /* live search positions per block (index: block number mod ZF_NB) */
    __shared__ uint32_t s_next, s_pending, s_hi, s_loading;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t lc = blockIdx.x;
    const ZcChunkGeom g = zc_geom(J.n, J.P.chunk_size, J.chunk0 + lc, J.finish, J.head, J.P.wsize);
    const uint32_t E = g.dict_len + g.len;
    const uint32_t nseg = (g.len + ZF_SEG - 1u) / ZF_SEG;
    const uint32_t hi_final = ((E + ZF_LOOK + 32u + ZF_BK - 1u) / ZF_BK) * ZF_BK;
    const uint32_t hi0 = hi_final < ZF_R ? hi_final : ZF_R;
    for (uint32_t b = 0; b < hi0 / ZF_BK; b++) zf_load_block(J, g, b * ZF_BK, s_win, s_pd, s_known, tid, ZF_THREADS);
    if (tid < ZF_NB) s_cnt[tid] = 0;
    if (tid == 0) { s_next = 0; s_pending = 0; s_hi = hi0; s_loading = 0; }
    __syncthreads();
    volatile uint32_t *v_hi = &s_hi, *v_cnt = s_cnt, *v_next = &s_next, *v_pending = &s_pending;
    RingWin win; win.w = s_win;
    RingPd pd; pd.pd = s_pd;
    uint32_t *hop = J.hop + (uint64_t)lc * J.P.chunk_size;      /* index = a-coordinate - dict_len */
    const uint32_t nice_match = J.P.nice, max_chain = J.P.chain, good = J.P.good, lazy = J.P.lazy, max_dist = J.P.max_dist;
    const uint32_t qbudget = (max_chain >> 2) ? (max_chain >> 2) : 0x7fffffffu;   /* zero wraps in the reference's do-while */
    const bool filt = J.P.strategy == ZC_STRAT_FILTERED;
    /* the parse of this lane: the hop in progress started at hp with nothing pending and has so far
     * k lazy literals and the match (L, D); L == 0: the first search, at hp itself, is under way */
    uint32_t hp = 0, k = 0, L = 0, D = 0, st = ZFS_ACQ;
    /* the walk of this lane: p the search position, pidx / cidx ring offsets of its string and of the candidate,
     * room = how much further back the walk may go (distance limit minus the candidate's distance) */
    uint32_t p = 0, pidx = 0, cidx = 0, room = 0, lim = 0, best = 2, plen = 2, bdist = 0, scan3 = 0, cap = 0, nice = 0;
    int32_t kk = 0;                           /* candidates examined minus the budget: negative while budget remains */
    int32_t walking = 0, parked = 0;          /* sign masks: -1 / 0 */
    uint32_t rb = ZF_NOBLK;                   /* the block this lane's search position is counted in */
    bool unseen = false;                      /* the hop start hp lies beyond the loaded data: its visited bit is tested when the data is there */
    uint32_t spins = 0;
#ifdef ZAMD_SPEC_STATS
    unsigned long long st_0 = 0, st_1 = 0, st_2 = 0, st_3 = 0, st_4 = 0, st_5 = 0, st_6 = 0, st_7 = 0;
    const long long t_begin = clock64();
#endif

    auto leave_block = [&]() { if (rb != ZF_NOBLK) { atomicSub(&s_cnt[rb % ZF_NB], 1u); rb = ZF_NOBLK; } };
    /* starts the search at q with prev_length pl; 0: it cannot find anything, 1: under way, 2: the data is not there yet */
    auto start_walk = [&](uint32_t q, uint32_t pl, uint32_t hi) -> int {
        if (q + ZF_LOOK > hi && hi < hi_final) { leave_block(); st = ZFS_WAIT; p = q; plen = pl; return 2; }
        const uint32_t nb = q / ZF_BK;
        if (nb != rb) { atomicAdd(&s_cnt[nb % ZF_NB], 1u); leave_block(); rb = nb; }     /* counted at the new place before it leaves the old one */
        st = ZFS_RES;
        ZS_STAT(2, 1);
        ZcMatchState ms;
        if (!zc_match_init(win, pd, q, E, g.corner, nice_match, ms, max_dist)) return 0;
        if (pl >= ms.cap) return 0;
        p = q; pidx = q % ZF_R; cap = ms.cap; nice = ms.nice; best = pl; plen = pl; bdist = 0;
        const uint32_t d0 = q - ms.cur;
        cidx = pidx >= d0 ? pidx - d0 : pidx + ZF_R - d0;
        lim = q < max_dist ? q : max_dist;
        room = lim - d0;
        scan3 = pl > 2u ? (win.at(pidx + pl - 2u) & 0x00ffffffu) : ms.scan3;
        kk = -(int32_t)(pl >= good ? qbudget : max_chain);
        walking = -1;
        return 1;
    };

    for (;;) {
        /* ---- full comparisons of parked candidates */
        if (parked < 0) {
            uint32_t len;
            {
                const uint32_t ic = cidx >> 2, ip = pidx >> 2;
                const uint32_t c0 = s_win[ic], c1 = s_win[ic + 1], c2 = s_win[ic + 2], c3 = s_win[ic + 3], c4 = s_win[ic + 4];
                const uint32_t q0 = s_win[ip], q1 = s_win[ip + 1], q2 = s_win[ip + 2], q3 = s_win[ip + 3], q4 = s_win[ip + 4];
                const uint32_t x0 = __builtin_amdgcn_alignbyte(c1, c0, cidx) ^ __builtin_amdgcn_alignbyte(q1, q0, pidx);
                const uint32_t x1 = __builtin_amdgcn_alignbyte(c2, c1, cidx) ^ __builtin_amdgcn_alignbyte(q2, q1, pidx);
                const uint32_t x2 = __builtin_amdgcn_alignbyte(c3, c2, cidx) ^ __builtin_amdgcn_alignbyte(q3, q2, pidx);
                const uint32_t x3 = __builtin_amdgcn_alignbyte(c4, c3, cidx) ^ __builtin_amdgcn_alignbyte(q4, q3, pidx);
                len = x0 ? zc_ctz32(x0) >> 3 : x1 ? 4u + (zc_ctz32(x1) >> 3) : x2 ? 8u + (zc_ctz32(x2) >> 3) : x3 ? 12u + (zc_ctz32(x3) >> 3) : 16u;
            }
            if (len == 16u) {
                while (len < cap) {
                    uint32_t x = win.at(cidx + len) ^ win.at(pidx + len);
                    if (x) { len += zc_ctz32(x) >> 3; break; }
                    len += 4;
                }
            }
            if (len > cap) len = cap;
            bool fin = false;
            if (len > best) {
                best = len; bdist = lim - room;
                if (len >= nice) fin = true;
                else scan3 = win.at(pidx + len - 2u) & 0x00ffffffu;
            }
            if (!fin) {
                const uint32_t d = s_pd[cidx];            /* "none" is 0xffff: beyond any room */
                if (kk >= 0 || d >= room) fin = true;
                else { cidx = cidx >= d ? cidx - d : cidx + ZF_R - d; room -= d; }
            }
            parked = 0; walking = fin ? 0 : -1;
        }
        /* ---- parse steps of the lanes whose walk is over, as straight-line code on predicates */
        for (int pass = 0; pass < 2; pass++) {
            const bool sv = walking == 0 && st != ZFS_DONE;
            if (__ballot(sv) == 0ull) break;
            /* 1. the search's result (deflate.c:1771-1815) */
            const bool res = sv && st == ZFS_RES;
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
            /* 3. where the parse goes on: behind the hop unless somebody has been there, else at the next starting point */
            bool have = false; uint32_t nhp = 0;
            const uint32_t hi_now = *v_hi;
            if (hop_done && cnd < E) {
                /* the visited bits of positions beyond the loaded data still belong to an older block */
                if (cnd + ZF_LOOK > hi_now && hi_now < hi_final) { have = true; nhp = cnd; unseen = true; }
                else {
                    const uint32_t i = cnd % ZF_R;
                    if (!((atomicOr(&s_known[i >> 5], 1u << (i & 31u)) >> (i & 31u)) & 1u)) { have = true; nhp = cnd; }
                }
            }
            bool raised = false;
            if ((hop_done || (sv && st == ZFS_ACQ)) && !have) {
                /* between taking a stretch and being counted at its place the lane is "pending": the loader must not
                 * conclude from the counts alone that nobody is about to search in an old block */
                atomicAdd(&s_pending, 1u); raised = true;
                const uint32_t seg = atomicAdd(&s_next, 1u);
                st = ZFS_ACQ;
                if (seg >= nseg) { st = ZFS_DONE; leave_block(); }
                else {
                    const uint32_t a = g.dict_len + seg * ZF_SEG, i = a % ZF_R;
                    if (a + ZF_LOOK > hi_now && hi_now < hi_final) { have = true; nhp = a; unseen = true; }
                    else if (!((atomicOr(&s_known[i >> 5], 1u << (i & 31u)) >> (i & 31u)) & 1u)) { have = true; nhp = a; }
                }
            }
            if (have) { hp = nhp; L = 0; k = 0; D = 0; st = ZFS_RES; }
            /* 4. the next search: the lazy one with prev_length = L, the first of a new hop with prev_length = 2
             *    (none at all unless max_lazy_match is above that, deflate.c:1765), or the one that waited for its data;
             *    a search that cannot start counts as one that found nothing */
            const bool retry = sv && st == ZFS_WAIT;
            if (probe || have || retry) {
                const uint32_t q = retry ? p : probe ? q1 : hp, pl = retry ? plen : probe ? L : ZC_MIN_MATCH - 1u;
                bool go_on = true;
                if (unseen && !(q + ZF_LOOK > hi_now && hi_now < hi_final)) {
                    /* the data of this hop start has arrived: has another parse been here meanwhile? */
                    unseen = false;
                    const uint32_t i = hp % ZF_R;
                    if ((atomicOr(&s_known[i >> 5], 1u << (i & 31u)) >> (i & 31u)) & 1u) { go_on = false; leave_block(); st = ZFS_ACQ; }
                }
                if (go_on) {
                    int r = 0;
                    if (lazy >= ZC_MIN_MATCH || retry) r = start_walk(q, pl, hi_now);
                    else st = ZFS_RES;
                    if (r == 0) { best = pl; plen = pl; }
                }
            }
            if (raised) atomicSub(&s_pending, 1u);
        }
        /* ---- the next block of the chunk over the oldest one, once nobody can look back into that */
        {
            uint32_t doit = 0, hi = 0;
            if (lane == 0) {
                hi = *v_hi;
                if (hi < hi_final && atomicCAS(&s_loading, 0u, 1u) == 0u) {
                    hi = *v_hi;
                    bool ok = hi < hi_final;
                    if (ok) {
                        const uint32_t lo = hi - ZF_R, bound = lo + 32768u + ZF_BK;
                        ok = *v_pending == 0u;
                        const uint32_t nx = *v_next;
                        ok = ok && (nx >= nseg || g.dict_len + nx * ZF_SEG >= bound);
                        for (uint32_t b = lo / ZF_BK; ok && b <= (lo + 32768u) / ZF_BK; b++) ok = v_cnt[b % ZF_NB] == 0u;
                    }
                    if (ok) doit = 1; else atomicExch(&s_loading, 0u);
                }
            }
            doit = __shfl(doit, 0); hi = __shfl(hi, 0);
            if (doit) {
                zf_load_block(J, g, hi, s_win, s_pd, s_known, lane, 64u);
                __threadfence_block();
                if (lane == 0) { *v_hi = hi + ZF_BK; __threadfence_block(); atomicExch(&s_loading, 0u); }
            }
        }
        if (__ballot(walking < 0) == 0ull) {
            if (__ballot(st != ZFS_DONE) == 0ull) break;
            if (++spins > ZF_SPIN_CAP) { if (lane == 0) atomicOr(J.err, (uint32_t)ZE_FOLLOW_STUCK); leave_block(); break; }
            if (lane == 0) ZS_STAT(4, 1);
            __builtin_amdgcn_s_sleep(4);
            continue;
        }
        spins = 0;
        /* ---- candidate steps until enough lanes need service */
        /* lanes that wait for data are not helped by a service round */
        const bool serviceable = st != ZFS_DONE && st != ZFS_WAIT;
        const uint32_t live = (uint32_t)__popcll(__ballot(serviceable));
        const uint32_t thr = live / 2u < 28u ? (live / 2u ? live / 2u : 1u) : 28u;
        const uint32_t peek_off = best - 2u;
        uint32_t c = cidx;
        uint32_t rm = room;
        int32_t wm = walking;
        int32_t kc = kk & walking;
        uint32_t xlast = 1u;
#ifdef ZAMD_SPEC_STATS
        const long long t_w = clock64();
        st_6 += st == ZFS_WAIT ? 1 : 0;
#endif
#pragma unroll 1
        for (int it = 0; it < 64 / ZK_MATCH_STEPS; it++) {
#pragma unroll
            for (int u = 0; u < ZK_MATCH_STEPS; u++) {
                ZS_STAT(1, 1);
                if (wm >= 0) continue;                            /* stopped lanes are masked off, their LDS reads with them */
                ZS_STAT(0, 1);
                const uint32_t ac = c + peek_off;
                const uint32_t wi = ac >> 2;
                const uint32_t v = __builtin_amdgcn_alignbyte(s_win[wi + 1], s_win[wi], ac);
                const int32_t d = (int32_t)s_pd[c];
                const uint32_t x = (v & 0x00ffffffu) ^ scan3;
                const int32_t nohit = -(int32_t)x;               /* negative unless the peek matched */
                xlast = x;
                kc -= wm;                                         /* count the candidate */
                const int32_t go = (d - (int32_t)rm) & kc & nohit;   /* negative iff the walk continues: in reach, in budget, no hit */
                wm = go >> 31;
                const int32_t dd = d & wm;
                rm -= (uint32_t)dd;
                int32_t cn = (int32_t)c - dd;
                cn += (cn >> 31) & (int32_t)ZF_R;                 /* wrap */
                c = (uint32_t)cn;
            }
            if ((uint32_t)__popcll(__ballot(wm == 0 && serviceable)) >= thr) break;
        }
#ifdef ZAMD_SPEC_STATS
        if (lane == 0) st_5 += (unsigned long long)(clock64() - t_w);
#endif
        if (walking < 0) { kk = kc; cidx = c; room = rm; }
        /* a walk that stopped in this round stands on the candidate that stopped it: it is parked if that candidate passed
         * the peek; otherwise it is over */
        parked = xlast == 0u ? (walking & ~wm) : 0;
        walking = wm | parked;                /* parked walks are still owned by the lane */
    }
#ifdef ZAMD_SPEC_STATS
    if (lane == 0) st_7 = (unsigned long long)(clock64() - t_begin);
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

#endif /* ZAMD_FOLLOW_IMPL_H */
