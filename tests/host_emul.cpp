/*
 * host_emul.cpp — runs the engine's per-lane building blocks (zamd_core.h) on
 * the host, chained the way the HIP kernels chain them (chain links -> match
 * records -> parse -> block plan -> layout -> emit), and compares the bytes
 * with the CPU oracle.  Test infrastructure: it checks the decomposition, it is
 * not a CPU code path of the product.
 *
 * build: g++ -O2 -I zlib_amd/csrc -I oracle tests/host_emul.cpp oracle/zoracle.c -o tests/host_emul
 * usage: host_emul [iterations] [seed]
 */
#include "zamd_core.h"
extern "C" {
#include "zoracle.h"
}
#include <cstdio>
#include <cstdlib>
#include <vector>

static int g_hops = 0;   /* 1: decomposed lazy parse (resolve / hop / expand) instead of the serial replay */
struct HostWin {
    const uint8_t *b; uint64_t n;    /* b = a-coordinate 0, n = bytes readable */
    uint32_t load4(uint32_t a) const { uint32_t v = 0; for (int k = 0; k < 4; k++) if ((uint64_t)a + k < n) v |= (uint32_t)b[a + k] << (8 * k); return v; }
};

static void or_bits(std::vector<uint8_t> &out, uint64_t bit, uint64_t v, uint32_t n)
{
    for (uint32_t i = 0; i < n; i++, bit++) if ((v >> i) & 1u) out[bit >> 3] |= (uint8_t)(1u << (bit & 7));
}

/* head[]/prev[] of deflate_fast held the way the kernel holds them: 16-bit window indices */
static int g_wbits = 15;      /* windowBits of the case at hand */
struct HostFast {
    const uint8_t *b; const ZcParams *P;
    std::vector<uint16_t> head, prev;
    uint32_t insert(uint32_t p, uint32_t B)
    {
        uint32_t h = zc_hash3(b[p], b[p + 1], b[p + 2], P->hash_shift, P->hash_mask), e = head[h];
        prev[(p - B) & (P->wsize - 1u)] = (uint16_t)e; head[h] = (uint16_t)(p - B);
        return e;
    }
    void slide()
    {
        for (auto &m : head) m = m >= P->wsize ? m - P->wsize : 0;
        for (auto &m : prev) m = m >= P->wsize ? m - P->wsize : 0;
    }
    void longest(uint32_t p, uint32_t B, uint32_t cur, uint32_t look, uint32_t *mlen, uint32_t *mdist)
    {   /* lane-parallel shape of the kernel: collect the candidates, measure them all, then pick in chain order */
        uint32_t sw = p - B, limit = sw > P->max_dist ? sw - P->max_dist : 0, cap = look < ZC_MAX_MATCH ? look : ZC_MAX_MATCH;
        uint32_t nice = P->nice < look ? P->nice : look, cand[32], lens[32], n = 0;
        do { cand[n++] = cur; cur = prev[cur & (P->wsize - 1u)]; } while (cur > limit && n < P->chain);
        for (uint32_t k = 0; k < n; k++) { uint32_t l = 0; const uint8_t *m = b + B + cand[k], *s = b + p; while (l < cap && m[l] == s[l]) l++; lens[k] = l; }
        uint32_t best = 2, bk = 0;
        for (uint32_t k = 0; k < n; k++) { if (lens[k] > best) { best = lens[k]; bk = k; if (best >= nice) break; } }
        if (best >= 3) { *mlen = best; *mdist = sw - cand[bk]; }
    }
};

static const int *g_tune;     /* deflateTune values or NULL */
static int g_tuned_wild;      /* tuned values outside what the record-based matcher reproduces */
static std::vector<uint8_t> emul_chunked(const uint8_t *in, uint64_t n, uint32_t chunk_size, int level, int strategy, int mem_level,
                                         std::vector<uint64_t> &chunk_off)
{
    static const uint16_t cfg[10][4] = {{0, 0, 0, 0}, {4, 4, 8, 4}, {4, 5, 16, 8}, {4, 6, 32, 32}, {4, 4, 16, 16},
                                        {8, 16, 32, 32}, {8, 16, 128, 128}, {8, 32, 128, 256}, {32, 128, 258, 1024}, {32, 258, 258, 4096}};
    ZcParams P; memset(&P, 0, sizeof P);
    P.level = level; P.strategy = strategy; P.mem_level = mem_level;
    P.good = cfg[level][0]; P.lazy = cfg[level][1]; P.nice = cfg[level][2]; P.chain = cfg[level][3];
    if (g_tune && level) { P.good = g_tune[0]; P.lazy = g_tune[1]; P.nice = g_tune[2]; P.chain = g_tune[3]; }
    P.hash_bits = mem_level + 7; P.hash_mask = (1u << P.hash_bits) - 1; P.hash_shift = (P.hash_bits + 2) / 3;
    P.lit_bufsize = 1u << (mem_level + 6);
    P.parser = strategy == ZC_STRAT_HUFF ? ZC_PARSE_HUFF : strategy == ZC_STRAT_RLE ? ZC_PARSE_RLE : level == 0 ? ZC_PARSE_STORED : level < 4 ? ZC_PARSE_FAST : ZC_PARSE_SLOW;
    P.chunk_size = chunk_size; P.max_blocks = chunk_size / (P.lit_bufsize - 1) + 3;
    P.wsize = 1u << g_wbits; P.max_dist = P.wsize - ZC_MIN_LOOK;
    if (level == 0) P.max_blocks += chunk_size / P.max_dist + 4;

    /* k_chain: global links */
    std::vector<uint16_t> pdist(n + 4, 0);
    {
        std::vector<uint64_t> head((size_t)1 << P.hash_bits, 0);
        for (uint64_t a = 0; a + 3 <= n; a++) {
            uint32_t h = zc_hash3(in[a], in[a + 1], in[a + 2], P.hash_shift, P.hash_mask);
            uint64_t old = head[h]; head[h] = a + 1;
            uint64_t d = old ? a + 1 - old : 0;
            pdist[a] = (uint16_t)(d > 32767 ? 0 : d);
        }
    }
    /* k_chain in its other two modes: links of the exact three-byte chains and of the four-byte chains */
    std::vector<uint16_t> pdist3e(n + 4, 0);
    {
        std::vector<uint64_t> head((size_t)1 << 15, 0);
        for (uint64_t a = 0; a + 3 <= n; a++) {
            uint32_t w = (uint32_t)in[a] | ((uint32_t)in[a + 1] << 8) | ((uint32_t)in[a + 2] << 16);
            uint32_t h = zc_hash3e(w);
            uint64_t old = head[h]; head[h] = a + 1;
            uint64_t d = old ? a + 1 - old : 0;
            pdist3e[a] = (uint16_t)(d > 32767 ? 0 : d);
        }
    }
    std::vector<uint16_t> pdist4(n + 4, 0);
    {
        std::vector<uint64_t> head((size_t)1 << 15, 0);
        for (uint64_t a = 0; a + 4 <= n; a++) {
            uint32_t w = (uint32_t)in[a] | ((uint32_t)in[a + 1] << 8) | ((uint32_t)in[a + 2] << 16) | ((uint32_t)in[a + 3] << 24);
            uint32_t h = zc_hash4(w);
            uint64_t old = head[h]; head[h] = a + 1;
            uint64_t d = old ? a + 1 - old : 0;
            pdist4[a] = (uint16_t)(d > 32767 ? 0 : d);
        }
    }
    const bool by_four = P.parser == ZC_PARSE_SLOW && !(P.nice < P.lazy || P.chain < 4 || P.nice < 4);   /* what the engine sends to k_lim + k_match */
    uint64_t chunks = n ? (n + chunk_size - 1) / chunk_size : 1;
    std::vector<uint8_t> out(n + n / 8 + chunks * 600 + 4096, 0);
    uint64_t obyte = 0;
    chunk_off.clear();
    for (uint64_t c = 0; c < chunks; c++) {
        ZcChunkGeom g = zc_geom(n, chunk_size, c, 1, 0, P.wsize);
        uint32_t E = g.dict_len + g.len;
        HostWin win; win.b = in + g.dstart; win.n = n - g.dstart;
        const uint16_t *pdc = pdist.data() + g.dstart;
        auto pd = [&](uint32_t a) { return (uint32_t)pdc[a]; };
        /* k_match */
        std::vector<uint32_t> rf(g.len + 1), rq(g.len + 1);
        if (P.parser == ZC_PARSE_SLOW)
            for (uint32_t i = 0; i < g.len; i++) zc_match_pos(win, pd, g.dict_len + i, E, g.corner, P.nice, P.chain, &rf[i], &rq[i], P.max_dist);
        if (by_four) {
            /* k_reach + k_first3 + k_match: the same records from the four-byte chains and the two words per position */
            const uint16_t *pdc4 = pdist4.data() + g.dstart, *pdc3 = pdist3e.data() + g.dstart;
            auto pd4 = [&](uint32_t a) { return (uint32_t)pdc4[a]; };
            auto pd3e = [&](uint32_t a) { return (uint32_t)pdc3[a]; };
            for (uint32_t i = 0; i < g.len; i++) {
                const uint32_t p = g.dict_len + i;
                const uint32_t f3 = E - p >= ZC_MIN_MATCH ? zc_first3(win, pd3e, p, P.max_dist) : 0u;
                const uint32_t lim = zc_lim_word(p, pd(p), zc_chain_reach(pd, p, P.chain), zc_chain_reach(pd, p, P.chain >> 2), P.max_dist, g.corner);
                uint32_t f, q;
                zc_match_pos4(win, pd4, p, E, P.nice, P.max_dist, lim, f3, &f, &q);
                if (f != rf[i] || q != rq[i]) {
                    fprintf(stderr, "four-byte chains differ at chunk %llu position %u: %08x %08x, hash chain %08x %08x (lim %08x f3 %08x)\n",
                            (unsigned long long)c, i, f, q, rf[i], rq[i], lim, f3);
                    exit(5);
                }
            }
        }
        else if (P.parser == ZC_PARSE_RLE)
            for (uint32_t i = 0; i < g.len; i++) rf[i] = rq[i] = zc_rle_pos(win, g.dict_len + i, E);
        /* k_parse */
        std::vector<uint32_t> sym(g.len + 1);
        std::vector<ZcBlock> blk(P.max_blocks);
        auto rec = [&](uint32_t p, uint32_t *f, uint32_t *q) { *f = rf[p - g.dict_len]; *q = rq[p - g.dict_len]; };
        auto symst = [&](uint32_t i, uint32_t w) { sym[i] = w; };
        auto blkst = [&](uint32_t i, const ZcBlock &b) { if (i >= blk.size()) { fprintf(stderr, "block table overflow\n"); exit(2); } blk[i] = b; };
        uint32_t nsym = 0, nb;
        if (P.parser == ZC_PARSE_FAST) {
            HostFast cx; cx.b = in + g.dstart; cx.P = &P;
            cx.head.assign((size_t)1 << P.hash_bits, 0); cx.prev.assign(P.wsize, 0);
            nb = zc_parse_fast(g.dict_len, g.len, g.final, P, cx, symst, blkst, &nsym);
        } else
        if (P.parser == ZC_PARSE_STORED) nb = zc_parse_stored(g.dict_len, g.len, g.final, P, blkst);
        else if (P.parser == ZC_PARSE_HUFF) nb = zc_parse_huff(g.dict_len, g.len, g.final, P, blkst);
        else if (P.parser == ZC_PARSE_RLE) nb = zc_parse_rle(g.dict_len, g.len, g.final, P, rec, symst, blkst, &nsym);
        else if (!g_hops) nb = zc_parse_slow(g.dict_len, g.len, g.final, P, rec, symst, blkst, &nsym);
        else {
            /* k_resolve: hop word per position, literal skips to the next match inside a 1024-position tile */
            std::vector<uint32_t> hop(g.len + 1, 0);
            if (g_hops == 2) {
                /* k_follow: hops are computed only where a parse arrives; parses start at every 16th position and
                 * run until they meet a position somebody has been to (0 = nobody) */
                for (uint32_t s0 = 0; s0 < g.len; s0 += 16)
                    for (uint32_t i = s0; i < g.len && hop[i] == 0;) {
                        uint32_t h = zc_hop_direct(win, pd, g.dict_len + i, E, g.corner, P);
                        if (!h) h = 1;
                        hop[i] = h;
                        i += (h & ZC_HOP_MATCH) ? zc_hop_k(h) + zc_hop_len(h) : 1u;
                    }
                /* every hop equals what the record-based resolver says */
                for (uint32_t i = 0; i < g.len; i++) if (hop[i]) {
                    uint32_t w = zc_resolve_pos(rec, g.dict_len + i, E, P);
                    if ((w ? w : 1u) != hop[i] && !g_tuned_wild) { fprintf(stderr, "hop mismatch at %u: direct %08x records %08x\n", i, hop[i], w); exit(4); }
                }
            } else {
            for (uint32_t i = 0; i < g.len; i++) hop[i] = zc_resolve_pos(rec, g.dict_len + i, E, P);
            for (uint32_t t = 0; t < g.len; t += 1024) {
                uint32_t te = t + 1024 < g.len ? t + 1024 : g.len, next = te;
                for (uint32_t i = te; i-- > t;) { if (hop[i] & ZC_HOP_MATCH) next = i; else hop[i] = next - i; }
            }
            }
            /* k_hop: the serial walk */
            std::vector<uint32_t> arr;
            for (uint32_t i = 0; i < g.len;) { uint32_t h = hop[i]; if (h & ZC_HOP_MATCH) { arr.push_back(i); i += zc_hop_k(h) + zc_hop_len(h); } else i += h; }
            /* k_expand: symbols, then blocks from the symbol positions */
            std::vector<uint32_t> spos;       /* start position of every symbol (a-coordinates) */
            uint32_t endp = g.dict_len;
            for (size_t hI = 0; hI <= arr.size(); hI++) {
                uint32_t a = hI < arr.size() ? g.dict_len + arr[hI] : E;
                uint32_t h = hI < arr.size() ? hop[arr[hI]] : 0;
                uint32_t lits = (a - endp) + (hI < arr.size() ? zc_hop_k(h) : 0);
                for (uint32_t o = 0; o < lits; o++) { sym[nsym] = endp + o; spos.push_back(endp + o); nsym++; }
                if (hI < arr.size()) {
                    uint32_t m = a + zc_hop_k(h);
                    sym[nsym] = zc_sym_match(zc_hop_len(h), zc_hop_dist(h)); spos.push_back(m); nsym++;
                    endp = m + zc_hop_len(h);
                }
            }
            const uint32_t lim = P.lit_bufsize - 1;
            bool tail_lit = nsym && !(sym[nsym - 1] & ZC_SYM_MATCH) && sym[nsym - 1] == E - 1;
            nb = zc_block_count(nsym, lim, g.final, tail_lit);
            uint32_t in_start = g.dict_len;
            for (uint32_t b = 0; b < nb; b++) {
                ZcBlock B; memset(&B, 0, sizeof B);
                B.sym_start = b * lim; B.sym_count = (b + 1 < nb) ? lim : nsym - b * lim;
                bool lastb = b + 1 == nb;
                uint32_t in_end, p_top;
                if (lastb) { in_end = E; p_top = E; }
                else {
                    uint32_t sI = (b + 1) * lim - 1, w = sym[sI];
                    in_end = (w & ZC_SYM_MATCH) ? spos[sI] + ((w >> 16) & 0xff) + 3 : spos[sI] + 1;
                    p_top = spos[sI] + 1;
                }
                B.in_start = in_start; B.in_len = in_end - in_start;
                B.flags = ((lastb && g.final) ? ZC_BF_LAST : 0u) | (in_start >= zc_window_base(p_top, E, ZC_MIN_LOOK, P.wsize) ? ZC_BF_CANSTORE : 0u);
                blkst(b, B); in_start = in_end;
            }
        }
        /* k_plan */
        std::vector<ZcBlockTab> tabs(nb);
        for (uint32_t j = 0; j < nb; j++) {
            static ZcTree<ZC_L_CODES> lt; static ZcTree<ZC_D_CODES> dt; static ZcTree<ZC_BL_CODES> bt;
            uint32_t hist[320] = {0};
            ZcBlock &b = blk[j];
            if (b.flags & ZC_BF_STORED) continue;
            for (uint32_t i = 0; i < b.sym_count; i++) {
                if (b.flags & ZC_BF_IMPLICIT) { hist[win.b[b.in_start + i]]++; continue; }
                uint32_t s = sym[b.sym_start + i];
                if (!(s & ZC_SYM_MATCH)) hist[win.b[s]]++;
                else { hist[257 + zc_len_code((s >> 16) & 0xff)]++; hist[288 + zc_dist_code((s & 0xffff) - 1)]++; }
            }
            hist[256] = 1;
            for (int i = 0; i < ZC_L_CODES; i++) lt.freq[i] = (uint16_t)hist[i];
            for (int i = 0; i < ZC_D_CODES; i++) dt.freq[i] = (uint16_t)hist[288 + i];
            uint32_t o1 = 0, s1 = 0, o2 = 0, s2 = 0;
            zc_build_tree(lt, 15, true, [](int k) { return zc_static_llen((uint32_t)k); },
                          [](int k) { return k >= 257 ? zc_len_extra((uint32_t)k - 257u) : 0u; }, &o1, &s1);
            zc_build_tree(dt, 15, true, [](int) { return 5u; }, [](int k) { return zc_dist_extra((uint32_t)k); }, &o2, &s2);
            zc_plan_block(lt, dt, bt, o1 + o2, s1 + s2, P.strategy, &b, &tabs[j]);
        }
        /* k_layout */
        uint64_t sync = 0;
        uint64_t cbytes = zc_layout_chunk([&](uint32_t i) { return &blk[i]; }, nb, g.final, &sync);
        chunk_off.push_back(obyte);
        uint64_t cbit = obyte * 8;
        /* k_emit */
        for (uint32_t j = 0; j < nb; j++) {
            const ZcBlock &b = blk[j];
            uint64_t bit = cbit + b.bit_off; uint32_t last = b.flags & ZC_BF_LAST;
            if (b.type == 0) {
                or_bits(out, bit, last, 3);
                uint64_t byte = (bit + 3 + 7) >> 3; uint32_t l = b.in_len & 0xffff;
                or_bits(out, byte * 8, l | ((~l & 0xffffu) << 16), 32);
                for (uint32_t i = 0; i < b.in_len; i++) out[byte + 4 + i] |= win.b[b.in_start + i];
                continue;
            }
            or_bits(out, bit, (b.type << 1) | last, 3); bit += 3;
            for (uint32_t i = 0; i < b.hdr_bits; i++) if ((tabs[j].hdr[i >> 3] >> (i & 7)) & 1) out[(bit + i) >> 3] |= (uint8_t)(1u << ((bit + i) & 7));
            bit += b.hdr_bits;
            for (uint32_t i = 0; i <= b.sym_count; i++) {
                uint64_t v; uint32_t nbits;
                if (i == b.sym_count) { v = tabs[j].lc[256] & 0xffff; nbits = tabs[j].lc[256] >> 16; }
                else {
                    uint32_t s, lit = 0;
                    if (b.flags & ZC_BF_IMPLICIT) { s = 0; lit = win.b[b.in_start + i]; }
                    else { s = sym[b.sym_start + i]; if (!(s & ZC_SYM_MATCH)) lit = win.b[s]; }
                    nbits = zc_sym_bits(s, lit, tabs[j].lc, tabs[j].dc, &v);
                }
                or_bits(out, bit, v, nbits); bit += nbits;
            }
            if (bit - (cbit + b.bit_off) != b.bits) { fprintf(stderr, "block %u: emitted %llu bits, planned %u\n", j, (unsigned long long)(bit - cbit - b.bit_off), b.bits); exit(3); }
        }
        if (!g.final) { uint64_t bit = (cbit + sync + 3 + 7) & ~7ull; or_bits(out, bit + 16, 0xffff, 16); }
        obyte += cbytes;
    }
    chunk_off.push_back(obyte);
    out.resize(obyte);
    return out;
}

static uint64_t rs = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 32); }

int main(int argc, char **argv)
{
    long iters = argc > 1 ? atol(argv[1]) : 60;
    if (argc > 2) rs = strtoull(argv[2], 0, 10) | 1;
    if (argc > 3) g_hops = atoi(argv[3]);
    /* closed-form code tables vs the reference's tables (via the oracle's trace is indirect; check self-consistency) */
    for (uint32_t lc = 0; lc < 256; lc++) { uint32_t c = zc_len_code(lc); if (lc < zc_len_base(c) || lc - zc_len_base(c) >= (1u << zc_len_extra(c)) + (c == 28)) { printf("len code table broken at %u\n", lc); return 1; } }
    for (uint32_t d = 0; d < 32768; d++) { uint32_t c = zc_dist_code(d); if (c >= 30 || d < zc_dist_base(c) || d - zc_dist_base(c) >= (1u << zc_dist_extra(c))) { printf("dist code table broken at %u\n", d); return 1; } }
    long bad = 0;
    static const uint32_t chunk_sizes[] = {32768, 49152, 65536, 131072, 262144};
    static const uint64_t tails[] = {0, 1, 2, 3, 5, 261, 262, 263, 16383, 32506, 32509, 32600, 32767, 65274 - 32768, 65374 - 32768};
    for (long it = 0; it < iters; it++) {
        uint32_t cs = chunk_sizes[rnd() % 5];
        uint64_t n = (uint64_t)(rnd() % 4) * cs + ((rnd() & 1) ? tails[rnd() % 15] % cs : rnd() % cs);
        if (it == 0) n = 0;
        if (it == 1) { cs = 65536; n = 65374; }      /* tail-mode slide geometry */
        int level = rnd() % 10, strategy = rnd() % 5, mem = (rnd() % 3) ? 8 : 1 + rnd() % 9;
        if (rnd() & 1) strategy = 0;
        if (level == 0 && (strategy == 2 || strategy == 3)) strategy = 0;
        if (level == 9 && n > 300000) level = 8;
        std::vector<uint8_t> in(n + 8);
        int kind = rnd() % 7;
        if (kind < 5) zo_corpus_fill(in.data(), n, kind, rnd() | 1);
        else if (kind == 5) for (uint64_t i = 0; i < n; i++) in[i] = (uint8_t)('a' + rnd() % (1 + (it % 16)));
        else { uint32_t per = 1 + rnd() % 300; for (uint64_t i = 0; i < n; i++) in[i] = (uint8_t)(i < per ? rnd() : in[i - per]); for (uint64_t i = 0; i < n / 211; i++) in[rnd() % n] ^= 4; }
        g_wbits = (rnd() % 3) ? 15 : 9 + (int)(rnd() % 6);
        int tune[4]; g_tune = NULL; g_tuned_wild = 0;
        if (g_hops == 2 && level >= 4 && rnd() % 3 == 0) {
            static const int lazies[] = {0, 1, 2, 3, 4, 5, 8, 16, 64, 258, 300}, nices[] = {3, 4, 8, 9, 32, 100, 258, 259, 1000}, chains[] = {1, 2, 3, 5, 8, 17, 40, 64, 100, 500};
            tune[0] = 1 + (int)(rnd() % 40); tune[1] = lazies[rnd() % 11]; tune[2] = nices[rnd() % 9]; tune[3] = chains[rnd() % 10];
            g_tune = tune; g_tuned_wild = tune[2] < tune[1] || tune[3] < 4;
        }
        std::vector<uint64_t> off;
        std::vector<uint8_t> got = emul_chunked(in.data(), n, cs, level, strategy, mem, off);
        std::vector<uint8_t> want(n + n / 4 + 65536);
        std::vector<uint32_t> clen(n / cs + 2);
        size_t wl;
        {   /* oracle chunk loop with the same chunk size */
            size_t o = 0, offi = 0, c = 0;
            do {
                const size_t wsz = (size_t)1 << g_wbits;
                size_t len = n - offi < cs ? n - offi : cs, dict = offi < wsz ? offi : wsz;
                size_t r = zo_deflate_chunk_tuned(in.data() + offi - dict, dict, len, level, strategy, mem, g_wbits, offi + len >= n, g_tune, want.data() + o, want.size() - o, NULL);
                if (r == (size_t)-1) { printf("oracle failed\n"); return 1; }
                clen[c++] = (uint32_t)r; o += r; offi += len;
            } while (offi < n);
            wl = o;
        }
        bool ok = got.size() == wl && memcmp(got.data(), want.data(), wl) == 0;
        if (!ok) {
            bad++;
            size_t d = 0; while (d < wl && d < got.size() && got[d] == want[d]) d++;
            printf("MISMATCH it=%ld n=%llu cs=%u L=%d S=%d M=%d W=%d kind=%d got=%zu want=%zu first diff %zu", it, (unsigned long long)n, cs, level, strategy, mem, g_wbits, kind, got.size(), wl, d);
            if (g_tune) printf(" tune=%d,%d,%d,%d", g_tune[0], g_tune[1], g_tune[2], g_tune[3]);
            printf("\n");
            if (bad > 5) break;
        }
    }
    printf("host_emul: %ld iterations, %ld mismatches\n", iters, bad);
    return bad != 0;
}
