/*
 * zamd_core.h — shared per-lane building blocks of the MI355X DEFLATE engine.
 *
 * Everything here is `__host__ __device__`: the HIP kernels in zamd_engine.hip
 * call these functions from device code, and tests/host_emul.cpp runs the same
 * functions on the host to check the pipeline's decomposition without a GPU.
 *
 * Coordinates: a chunk is addressed in "a-coordinates": a = 0 is the first byte
 * of the chunk's dictionary (the 32 KiB of input before the chunk, or the chunk
 * itself when it is the first one), the chunk's own bytes are
 * [dict_len, dict_len + len).  Window index 0 of the reference (deflate.c:104,
 * NIL) is therefore a = 0 until the first window slide.
 *
 * Citations are file:line in zlib 1.2.8.
 */
#ifndef ZAMD_CORE_H
#define ZAMD_CORE_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ZC_HD __host__ __device__ inline
#else
#define ZC_HD inline
#endif

#define ZC_WSIZE     32768u                 /* windowBits 15: the largest window; smaller ones (deflate.c:284) come in ZcParams */
#define ZC_MIN_MATCH 3u                     /* zutil.h:75 */
#define ZC_MAX_MATCH 258u                   /* zutil.h:76 */
#define ZC_MIN_LOOK  262u                   /* deflate.h:281 */
#define ZC_MAX_DIST  (ZC_WSIZE - ZC_MIN_LOOK)   /* deflate.h:286 */
#define ZC_TOO_FAR   4096u                  /* deflate.c:107 */
#define ZC_L_CODES   286
#define ZC_D_CODES   30
#define ZC_BL_CODES  19
#define ZC_END_BLOCK 256
#define ZC_NO_CORNER 0xffffffffu

enum { ZC_PARSE_SLOW = 0, ZC_PARSE_FAST = 1, ZC_PARSE_RLE = 2, ZC_PARSE_HUFF = 3, ZC_PARSE_STORED = 4 };
enum { ZC_STRAT_DEFAULT = 0, ZC_STRAT_FILTERED = 1, ZC_STRAT_HUFF = 2, ZC_STRAT_RLE = 3, ZC_STRAT_FIXED = 4 };

/* block flags */
#define ZC_BF_LAST      1u    /* BFINAL */
#define ZC_BF_CANSTORE  2u    /* block start still inside the window (deflate.c:1539) */
#define ZC_BF_IMPLICIT  4u    /* symbols are the literal bytes of the block, no symbol buffer */
#define ZC_BF_STORED    8u    /* level 0: the block is stored whatever it would cost coded (trees.c:945-948) */

struct ZcParams {
    uint32_t level, strategy, mem_level;
    uint32_t good, lazy, nice, chain;          /* configuration_table, deflate.c:131-143 */
    uint32_t hash_bits, hash_shift, hash_mask; /* deflate.c:284-291 */
    uint32_t lit_bufsize;                      /* deflate.c:299 */
    uint32_t parser;                           /* ZC_PARSE_* */
    uint32_t chunk_size;
    uint32_t max_blocks;                       /* per-chunk capacity of the block table */
    uint32_t wsize, max_dist;                  /* 1 << windowBits, wsize - MIN_LOOKAHEAD (deflate.c:284, deflate.h:286) */
    uint32_t tail;                             /* how the call's last chunk ends when it is not final: ZC_TAIL_* */
};
enum { ZC_TAIL_SYNC = 0,    /* Z_SYNC_FLUSH / Z_FULL_FLUSH: empty stored block, byte aligned (deflate.c:926-927) */
       ZC_TAIL_ALIGN = 1,   /* Z_PARTIAL_FLUSH: _tr_align, an empty static block of 10 bits (trees.c:892-901) */
       ZC_TAIL_NONE = 2 };  /* Z_BLOCK: the last block is complete, nothing follows it (deflate.c:923-925) */

struct ZcBlock {                 /* one _tr_flush_block() call */
    uint32_t sym_start, sym_count;
    uint32_t in_start, in_len;   /* a-coordinates / stored_len */
    uint32_t flags;
    uint32_t type;               /* 0 stored, 1 static, 2 dynamic (set by the planner) */
    uint32_t bits;               /* non-stored: 3 + coded bits; stored: unused */
    uint32_t hdr_bits;           /* dynamic: bits of the tree header after the 3 type bits */
    uint64_t bit_off;            /* absolute bit offset in the output buffer */
};

struct ZcChunkGeom {
    uint64_t dstart;             /* absolute offset of a = 0 */
    uint32_t dict_len, len, final, corner;
};

/* Chunk c of an n-byte input cut every chunk_size bytes.  `corner` is the one
 * position whose head candidate at distance exactly MAX_DIST is lost to a
 * window slide in tail mode (fill_window called with no input left,
 * deflate.c:1419-1451), or ZC_NO_CORNER. */
ZC_HD ZcChunkGeom zc_geom(uint64_t n, uint32_t chunk_size, uint64_t c, int finish, uint32_t head = 0, uint32_t wsize = ZC_WSIZE)
{
    /* `head` bytes (at most wsize) in front of chunk 0 belong to the buffer and are
     * chunk 0's dictionary: a shard that continues somebody else's stream, or a preset dictionary */
    const uint32_t maxd = wsize - ZC_MIN_LOOK;
    ZcChunkGeom g;
    uint64_t start = head + c * (uint64_t)chunk_size;
    uint64_t rem = n - start;
    g.len = (uint32_t)(rem < chunk_size ? rem : chunk_size);
    g.dict_len = (uint32_t)(start < wsize ? start : wsize);
    g.dstart = start - g.dict_len;
    g.final = (finish && start + g.len >= n) ? 1u : 0u;
    uint32_t E = g.dict_len + g.len;
    uint32_t bt = E <= 2 * wsize ? 0u : ((E - 2 * wsize + wsize - 1) / wsize) * wsize;
    uint32_t tail = E - bt;       /* data end as a window index in the last window */
    g.corner = (tail >= wsize + maxd + 3 && tail <= 2 * wsize - 1) ? bt + wsize + maxd : ZC_NO_CORNER;
    return g;
}

ZC_HD uint32_t zc_hash3(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t shift, uint32_t mask)
{   /* three UPDATE_HASH steps, deflate.c:167 */
    return ((((b0 << shift) ^ b1) << shift) ^ b2) & mask;
}

ZC_HD uint32_t zc_ctz32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__ffs((int)x) - 1u;
#else
    return (uint32_t)__builtin_ctz(x);
#endif
}
ZC_HD uint32_t zc_clz32(uint32_t x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)__clz((int)x);
#else
    return (uint32_t)__builtin_clz(x);
#endif
}
ZC_HD uint32_t zc_bitrev(uint32_t v, uint32_t n)        /* bi_reverse, trees.c:1154 */
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __brev(v) >> (32u - n);
#else
    uint32_t r = 0;
    for (uint32_t i = 0; i < n; i++) { r = (r << 1) | (v & 1u); v >>= 1; }
    return r;
#endif
}

/* ------------------------------------------------------------------ symbol codes
 * Closed forms of _length_code / _dist_code / base_length / base_dist /
 * extra_lbits / extra_dbits (trees.c:62-66, 257-288). */
ZC_HD uint32_t zc_len_code(uint32_t lc)           /* lc = length - 3, 0..255 */
{
    if (lc < 8) return lc;
    if (lc == 255) return 28;
    uint32_t e = 29u - zc_clz32(lc);              /* floor(log2 lc) - 2 */
    return 4u * e + 4u + ((lc >> e) & 3u);
}
ZC_HD uint32_t zc_len_extra(uint32_t code) { return (code < 8 || code == 28) ? 0u : (code >> 2) - 1u; }
ZC_HD uint32_t zc_len_base(uint32_t code)
{
    if (code < 8) return code;
    if (code == 28) return 255;
    return (4u + (code & 3u)) << ((code >> 2) - 1u);
}
ZC_HD uint32_t zc_dist_code(uint32_t d)           /* d = distance - 1, 0..32767 */
{
    if (d < 4) return d;
    uint32_t hb = 31u - zc_clz32(d);
    return 2u * hb + ((d >> (hb - 1u)) & 1u);
}
ZC_HD uint32_t zc_dist_extra(uint32_t code) { return code < 4 ? 0u : (code >> 1) - 1u; }
ZC_HD uint32_t zc_dist_base(uint32_t code) { return code < 4 ? code : (2u + (code & 1u)) << ((code >> 1) - 1u); }

/* static literal/length tree (trees.c:291-301): returns code | len<<16 */
ZC_HD uint32_t zc_static_lcode(uint32_t n)
{
    uint32_t len, c;
    if (n <= 143) { len = 8; c = 0x30 + n; }
    else if (n <= 255) { len = 9; c = 0x190 + (n - 144); }
    else if (n <= 279) { len = 7; c = n - 256; }
    else { len = 8; c = 0xC0 + (n - 280); }
    return zc_bitrev(c, len) | (len << 16);
}
ZC_HD uint32_t zc_static_llen(uint32_t n) { return n <= 143 ? 8u : n <= 255 ? 9u : n <= 279 ? 7u : 8u; }
ZC_HD uint32_t zc_static_dcode(uint32_t n) { return zc_bitrev(n, 5) | (5u << 16); }

/* Symbol word written by the parsers: literal = a-coordinate of the byte
 * (bit 31 clear); match = bit 31 | (length-3) << 16 | distance. */
#define ZC_SYM_MATCH 0x80000000u
ZC_HD uint32_t zc_sym_match(uint32_t len, uint32_t dist) { return ZC_SYM_MATCH | ((len - 3u) << 16) | dist; }

/* compress_block() for one symbol (trees.c:1072-1096): bits LSB-first in *v.
 * ltab/dtab hold code | len<<16 per symbol. */
template <class LTab, class DTab>
ZC_HD uint32_t zc_sym_bits(uint32_t sym, uint32_t lit_byte, const LTab &ltab, const DTab &dtab, uint64_t *v)
{
    if (!(sym & ZC_SYM_MATCH)) {
        uint32_t e = ltab[lit_byte];
        *v = e & 0xffffu;
        return e >> 16;
    }
    uint32_t lc = (sym >> 16) & 0xffu, d = (sym & 0xffffu) - 1u;
    uint32_t c = zc_len_code(lc), e = ltab[c + 257u];
    uint64_t acc = e & 0xffffu; uint32_t n = e >> 16;
    uint32_t xb = zc_len_extra(c);
    acc |= (uint64_t)(lc - zc_len_base(c)) << n; n += xb;
    c = zc_dist_code(d); e = dtab[c];
    acc |= (uint64_t)(e & 0xffffu) << n; n += e >> 16;
    xb = zc_dist_extra(c);
    acc |= (uint64_t)(d - zc_dist_base(c)) << n; n += xb;
    *v = acc;
    return n;
}

/* ------------------------------------------------------------------ match search
 * longest_match() (deflate.c:1148-1289) evaluated without knowing the parser's
 * state.  For levels 4-9 every string with three bytes of lookahead is inserted
 * in order (deflate.c:1757, 1806-1810), so the chain of p is "all earlier
 * positions with p's hash, newest first" and can be walked ahead of the parse.
 * The walk starts from best_len = 2 and keeps the last strictly-longer
 * candidate seen within each of the two chain budgets the parser can ask for
 * (max_chain, or max_chain >> 2 when prev_length >= good_match, :1185-1187),
 * stopping at the first candidate of nice length (:1276).  The parser then
 * accepts the recorded match iff it is longer than its prev_length, which is
 * what starting the walk from best_len = prev_length does.
 *
 * Win::load4(a) returns the 4 input bytes at a-coordinate a (little endian),
 * pd(a) the distance from a to the previous string with the same hash (0 =
 * none within 32767).  Result words: len | dist << 16, len = 0 for "no match".
 */
struct ZcMatchState {
    uint32_t p, cur, k, best, bdist, qrec, cap, nice, scan3, qset, maxd;
};

/* Sets up the walk for position p.  Returns false when p has no usable head
 * candidate (its record is then 0, 0). */
template <class Win, class Pd>
ZC_HD bool zc_match_init(const Win &win, const Pd &pd, uint32_t p, uint32_t E, uint32_t corner,
                         uint32_t nice_match, ZcMatchState &s, uint32_t max_dist = ZC_MAX_DIST)
{
    s.maxd = max_dist;
    uint32_t look = E - p;
    if (look < ZC_MIN_MATCH) return false;                 /* no INSERT_STRING, deflate.c:1756 */
    uint32_t d0 = pd(p);
    if (d0 == 0 || d0 > max_dist) return false;            /* hash_head == NIL or too far, :1765-1766 */
    if (d0 >= p) return false;                             /* window index 0 is NIL */
    if (p == corner && d0 == max_dist) return false;       /* slid onto index 0 in tail mode */
    s.p = p; s.cur = p - d0; s.k = 0; s.best = ZC_MIN_MATCH - 1; s.bdist = 0; s.qrec = 0; s.qset = 0;
    s.cap = look < ZC_MAX_MATCH ? look : ZC_MAX_MATCH;
    s.nice = nice_match < look ? nice_match : look;        /* :1191 */
    s.scan3 = win.load4(p) & 0x00ffffffu;
    return true;
}

/* One chain step in three pieces, so that a kernel can overlap the loads of
 * several walks and batch the rare long comparisons:
 *   zc_match_peek     loads the three bytes that decide whether s.cur can beat
 *                     s.best, and the link to the next chain member
 *   zc_match_extend   full comparison of a candidate that passed the peek;
 *                     returns true when it reached nice length (walk over)
 *   zc_match_advance  chain budget, quarter-budget snapshot, link and distance
 *                     limit; returns true when the walk is over */
template <class Win, class Pd>
ZC_HD void zc_match_peek(const Win &win, const Pd &pd, const ZcMatchState &s, uint32_t *c3, uint32_t *d)
{
    *c3 = win.load4(s.cur + s.best - 2u) & 0x00ffffffu;
    *d = pd(s.cur);
}

template <class Win>
ZC_HD bool zc_match_extend(const Win &win, ZcMatchState &s)
{
    uint32_t len = 0;
    while (len < s.cap) {
        uint32_t x = win.load4(s.cur + len) ^ win.load4(s.p + len);
        if (x) { len += zc_ctz32(x) >> 3; break; }
        len += 4;
    }
    if (len > s.cap) len = s.cap;
    if (len > s.best) {
        s.best = len; s.bdist = s.p - s.cur;
        if (len >= s.nice) { s.k++; return true; }         /* :1276 */
        s.scan3 = win.load4(s.p + len - 2u) & 0x00ffffffu;
    }
    return false;
}

ZC_HD bool zc_match_advance(ZcMatchState &s, uint32_t d, uint32_t max_chain)
{
    s.k++;
    if (s.k == (max_chain >> 2)) { s.qrec = s.best >= ZC_MIN_MATCH ? (s.best | (s.bdist << 16)) : 0; s.qset = 1; }
    if (s.k >= max_chain) return true;
    if (d == 0 || d >= s.cur) return true;                 /* end of chain, or window index <= 0 */
    s.cur -= d;
    return s.p - s.cur >= s.maxd;                          /* cur <= limit, :1158, 1284 */
}

/* Examines the candidate s.cur and moves to the next chain member.  Returns
 * true when the walk is over. */
template <class Win, class Pd>
ZC_HD bool zc_match_step(const Win &win, const Pd &pd, uint32_t max_chain, ZcMatchState &s)
{
    uint32_t c3, d;
    zc_match_peek(win, pd, s, &c3, &d);
    if (c3 == s.scan3 && zc_match_extend(win, s)) return true;
    return zc_match_advance(s, d, max_chain);
}

ZC_HD void zc_match_result(const ZcMatchState &s, uint32_t max_chain, uint32_t *rec_full, uint32_t *rec_quarter)
{
    uint32_t full = s.best >= ZC_MIN_MATCH ? (s.best | (s.bdist << 16)) : 0;
    uint32_t qbudget = max_chain >> 2;
    *rec_full = full;
    *rec_quarter = qbudget == 0 ? 0u : ((s.qset && s.k > qbudget) ? s.qrec : full);
}

template <class Win, class Pd>
ZC_HD void zc_match_pos(const Win &win, const Pd &pd, uint32_t p, uint32_t E, uint32_t corner,
                        uint32_t nice_match, uint32_t max_chain, uint32_t *rec_full, uint32_t *rec_quarter,
                        uint32_t max_dist = ZC_MAX_DIST)
{
    ZcMatchState s;
    *rec_full = 0; *rec_quarter = 0;
    if (!zc_match_init(win, pd, p, E, corner, nice_match, s, max_dist)) return;
    while (!zc_match_step(win, pd, max_chain, s)) {}
    zc_match_result(s, max_chain, rec_full, rec_quarter);
}

/* ------------------------------------------------------------------ the same search along four-byte chains
 * longest_match() examines every member of the hash chain (strings that share three bytes, give or take a hash
 * collision), and a member costs a step whether it can improve the match or not.  What it returns can be had from
 * far fewer steps:
 *   - a candidate of length 4 or more shares four bytes with the string, so it is a member of the string's FOUR-BYTE
 *     chain (strings with the same hash of four bytes, newest first: zc_hash4, links built like the hash chain's);
 *   - if no such candidate is within reach, the result is the nearest candidate of length 3 — the nearest earlier
 *     string with the same three bytes, found along a chain of its own (zc_hash3e: a hash that spreads three-byte
 *     strings evenly, so that the nearest member is the match but for rare collisions) — or nothing;
 *   - "within reach" is the chain budget (max_chain candidates, a quarter of it from good_match on, deflate.c:1185-1187)
 *     and the distance limit (:1158-1159, 1284).  The budget counts hash-chain members, so it is turned into a
 *     distance once per position: the distance of the string's max_chain-th (max_chain/4-th) predecessor in the hash
 *     chain.  A candidate is within the budget iff it is no further away than that.
 * The walk is over four-byte-chain members no further than min(budget distance, MAX_DIST - 1) and keeps
 * longest_match's rules (strictly longer wins, the first candidate of nice length ends it).  The hash-chain head alone
 * may lie at exactly MAX_DIST (:1660, 1766); nothing lies behind it then, and it is the only candidate.
 *
 * Per position two words say all this (k_reach and k_first3 write them, k_match reads them):
 *   lim   bits 0-14: the largest distance a candidate may have (full budget); bit 15: the hash-chain head lies at
 *         exactly MAX_DIST and is admissible; bits 16-31: the largest distance within a quarter of the budget
 *   f3    distance of the nearest earlier string with the same three bytes, 0 = none within MAX_DIST */
ZC_HD uint32_t zc_hash4(uint32_t w) { return (w * 0x9E3779B1u) >> 17; }
ZC_HD uint32_t zc_hash3e(uint32_t w) { return ((w & 0x00ffffffu) * 0x9E3779B1u) >> 17; }

/* pd3e(a) = distance to the previous string with a's zc_hash3e, 0 = none within 32767 */
template <class Win, class Pd>
ZC_HD uint32_t zc_first3(const Win &win, const Pd &pd3e, uint32_t p, uint32_t max_dist)
{
    const uint32_t scan3 = win.load4(p) & 0x00ffffffu;
    uint32_t cur = p;
    for (;;) {
        const uint32_t d = pd3e(cur);
        if (d == 0 || d > cur) return 0;
        cur -= d;
        if (p - cur > max_dist) return 0;
        if ((win.load4(cur) & 0x00ffffffu) == scan3) return p - cur;
    }
}

/* Distance from p to its steps-th predecessor in the hash chain, 0xffff when the chain ends before it or the distance
 * exceeds 32767 (k_reach gets the same number by pointer jumping). */
template <class Pd>
ZC_HD uint32_t zc_chain_reach(const Pd &pd, uint32_t p, uint32_t steps)
{
    uint32_t cur = p;
    for (uint32_t k = 0; k < steps; k++) {
        const uint32_t d = pd(cur);
        if (d == 0 || d > cur) return 0xffffu;
        cur -= d;
        if (p - cur > 32767u) return 0xffffu;
    }
    return p - cur;
}

/* head_dist: the position's own hash-chain link (0 = none) */
ZC_HD uint32_t zc_lim_word(uint32_t p, uint32_t head_dist, uint32_t reach_full, uint32_t reach_quarter, uint32_t max_dist, uint32_t corner)
{
    uint32_t wf = reach_full < max_dist - 1u ? reach_full : max_dist - 1u;      /* later candidates need cur > limit, :1284 */
    const uint32_t nil = p ? p - 1u : 0u;                                      /* a-coordinate 0 is NIL */
    if (wf > nil) wf = nil;
    const uint32_t wq = reach_quarter < wf ? reach_quarter : wf;
    /* hash_head at exactly MAX_DIST (:1660, 1766), unless it is window index 0 or has slid onto it in tail mode */
    const uint32_t lone = (head_dist == max_dist && p > max_dist && p != corner) ? 0x8000u : 0u;
    return wf | lone | (wq << 16);
}

/* longest_match for position p from its two words and the four-byte chain (pd4(a) = distance to the previous string
 * with a's four-byte hash, 0 = none within 32767).  Same result words as zc_match_pos. */
template <class Win, class Pd4>
ZC_HD void zc_match_pos4(const Win &win, const Pd4 &pd4, uint32_t p, uint32_t E, uint32_t nice_match, uint32_t max_dist, uint32_t lim, uint32_t f3,
                         uint32_t *rec_full, uint32_t *rec_quarter)
{
    *rec_full = 0; *rec_quarter = 0;
    const uint32_t look = E - p, d3 = f3 & 0x7fffu;
    if (look < ZC_MIN_MATCH || d3 == 0) return;
    uint32_t wf = lim & 0x7fffu, wq = lim >> 16;
    const uint32_t cap = look < ZC_MAX_MATCH ? look : ZC_MAX_MATCH, nice = nice_match < look ? nice_match : look;
    uint32_t best, full = 0, qrec = 0, cur, d;
    if (lim & 0x8000u) {
        /* the hash-chain head at MAX_DIST is the only candidate there is; it is measured in full if its three bytes match */
        if (d3 != max_dist) return;
        best = ZC_MIN_MATCH - 1u; cur = p; d = max_dist; wf = max_dist; wq = max_dist;
    } else {
        if (d3 > wf) return;
        full = ZC_MIN_MATCH | (d3 << 16); qrec = d3 <= wq ? full : 0u;
        best = ZC_MIN_MATCH; cur = p; d = pd4(p);
        if (cap == ZC_MIN_MATCH) d = 0;
    }
    while (d != 0 && (p - cur) + d <= wf) {
        cur -= d;
        if (((win.load4(cur + best - 2u) ^ win.load4(p + best - 2u)) & 0x00ffffffu) == 0) {    /* the peek of k_match */
            uint32_t len = 0;
            while (len < cap) {
                const uint32_t x = win.load4(cur + len) ^ win.load4(p + len);
                if (x) { len += zc_ctz32(x) >> 3; break; }
                len += 4;
            }
            if (len > cap) len = cap;
            if (len > best) {
                best = len; full = len | ((p - cur) << 16);
                if (p - cur <= wq) qrec = full;
                if (len >= nice) break;                /* :1276 */
            }
        }
        if (lim & 0x8000u) break;
        d = pd4(cur);
    }
    *rec_full = full; *rec_quarter = qrec;
}

/* Z_RLE (deflate.c:1884-1898): run of bytes equal to the previous byte. */
template <class Win>
ZC_HD uint32_t zc_rle_pos(const Win &win, uint32_t p, uint32_t E)
{
    uint32_t look = E - p;
    if (look < ZC_MIN_MATCH || p == 0) return 0;
    uint32_t cap = look < ZC_MAX_MATCH ? look : ZC_MAX_MATCH;
    uint32_t c = win.load4(p - 1) & 0xffu, pat = c * 0x01010101u, len = 0;
    while (len < cap) {
        uint32_t x = win.load4(p + len) ^ pat;
        if (x) { len += zc_ctz32(x) >> 3; break; }
        len += 4;
    }
    if (len > cap) len = cap;
    return len >= ZC_MIN_MATCH ? (len | (1u << 16)) : 0;
}

/* ------------------------------------------------------------------ parsers
 * Window bookkeeping shared by all parsers: fill_window() for a call that hands
 * over the whole chunk (deflate.c:1390-1494).  B = a-coordinate of window index
 * 0, wend = end of the data currently in the window. */
struct ZcWindow {
    uint32_t B, wend, E, W;       /* W = window size */
    ZC_HD void refill(uint32_t p)
    {
        if (p - B >= W + (W - ZC_MIN_LOOK)) B += W;
        uint64_t cap = (uint64_t)B + 2u * W;
        wend = E < cap ? E : (uint32_t)cap;
    }
};

/* Common symbol/block sink.  Sym(i, word) stores symbol i, Blk(i, ZcBlock) block i. */
template <class Sym, class Blk>
struct ZcSink {
    Sym sym; Blk blk;
    uint32_t nsym, blk_sym0, bstart, nblk, lim;
    ZC_HD ZcSink(Sym s, Blk b, uint32_t start, uint32_t lit_bufsize)
        : sym(s), blk(b), nsym(0), blk_sym0(0), bstart(start), nblk(0), lim(lit_bufsize - 1) {}
    ZC_HD bool emit(uint32_t word)                 /* _tr_tally, trees.c:1010-1050 */
    {
        sym(nsym++, word);
        return nsym - blk_sym0 == lim;
    }
    ZC_HD void flush(uint32_t p, uint32_t last, const ZcWindow &w, uint32_t extra_flags = 0)
    {   /* FLUSH_BLOCK_ONLY, deflate.c:1538-1547 */
        ZcBlock b;
        b.sym_start = blk_sym0; b.sym_count = nsym - blk_sym0;
        b.in_start = bstart; b.in_len = p - bstart;
        b.flags = (last ? ZC_BF_LAST : 0u) | (bstart >= w.B ? ZC_BF_CANSTORE : 0u) | extra_flags;
        b.type = 0; b.bits = 0; b.hdr_bits = 0; b.bit_off = 0;
        blk(nblk++, b);
        blk_sym0 = nsym; bstart = p;
    }
};

/* deflate_slow() (deflate.c:1730-1853) replayed over the precomputed match
 * records.  Rec(p, &full, &quarter) loads the record of a-coordinate p.
 * Returns the number of blocks; *nsym_out receives the symbol count. */
template <class Rec, class Sym, class Blk>
ZC_HD uint32_t zc_parse_slow(uint32_t dict_len, uint32_t len, uint32_t final_, const ZcParams &P,
                             Rec rec, Sym sym, Blk blk, uint32_t *nsym_out)
{
    ZcWindow w; w.B = 0; w.E = dict_len + len; w.wend = dict_len; w.W = P.wsize;
    ZcSink<Sym, Blk> out(sym, blk, dict_len, P.lit_bufsize);
    uint32_t p = dict_len, mlen = ZC_MIN_MATCH - 1, mdist = 0, avail = 0;
    for (;;) {
        if (w.wend - p < ZC_MIN_LOOK) { w.refill(p); if (w.wend == p) break; }
        uint32_t plen = mlen, pdist = mdist;
        mlen = ZC_MIN_MATCH - 1;
        if (plen < P.lazy && w.wend - p >= ZC_MIN_MATCH) {
            uint32_t rf, rq; rec(p, &rf, &rq);
            uint32_t v = plen >= P.good ? rq : rf;
            uint32_t l = v & 0xffffu, d = v >> 16;
            if (l > plen &&
                !(l <= 5 && (P.strategy == ZC_STRAT_FILTERED || (l == ZC_MIN_MATCH && d > ZC_TOO_FAR)))) {
                mlen = l; mdist = d;                       /* :1771-1785 */
            }
        }
        if (plen >= ZC_MIN_MATCH && mlen <= plen) {        /* :1790-1815 */
            bool f = out.emit(zc_sym_match(plen, pdist));
            p += plen - 1; avail = 0; mlen = ZC_MIN_MATCH - 1;
            if (f) out.flush(p, 0, w);
        } else if (avail) {                                /* :1817-1829 */
            if (out.emit(p - 1)) out.flush(p, 0, w);
            p++;
        } else { avail = 1; p++; }                         /* :1830-1837 */
    }
    if (avail) out.emit(p - 1);                            /* :1840-1844 */
    if (final_) out.flush(p, 1, w);
    else if (out.nsym != out.blk_sym0) out.flush(p, 0, w);
    *nsym_out = out.nsym;
    return out.nblk;
}

/* ------------------------------------------------------------------ lazy parse, decomposed
 * The replay above is serial in the position.  Most of it is not: whenever
 * deflate_slow reaches a position with nothing pending (after a match was
 * emitted, or while it is only carrying a literal), what it does next depends on
 * the match records alone.  zc_resolve_pos() computes that for every position in
 * parallel: the hop word says "k lazy literals, then the match (L, D)"; the only
 * serial part left is following the hops (zc_hop_next), and the symbol stream and
 * block table are expanded from the hop list in parallel.
 *
 * hop word: bit 31 set = match hop: D in bits 0-14, L-3 in bits 15-22, k in bits
 * 23-30; bit 31 clear = no match starts here: the value is how many positions to
 * step over as literals (filled in by the caller, 0 from zc_resolve_pos). */
#define ZC_HOP_MATCH 0x80000000u
ZC_HD uint32_t zc_hop_pack(uint32_t k, uint32_t L, uint32_t D) { return ZC_HOP_MATCH | (k << 23) | ((L - 3u) << 15) | D; }
ZC_HD uint32_t zc_hop_k(uint32_t h) { return (h >> 23) & 0xffu; }
ZC_HD uint32_t zc_hop_len(uint32_t h) { return ((h >> 15) & 0xffu) + 3u; }
ZC_HD uint32_t zc_hop_dist(uint32_t h) { return h & 0x7fffu; }

/* longest_match's result as the caller sees it (deflate.c:1771-1785): the record
 * chosen by prev_length, accepted only if longer, then the TOO_FAR / Z_FILTERED rule */
ZC_HD bool zc_accept(uint32_t rf, uint32_t rq, uint32_t plen, const ZcParams &P, uint32_t *L, uint32_t *D)
{
    uint32_t v = plen >= P.good ? rq : rf;
    uint32_t l = v & 0xffffu, d = v >> 16;
    if (l <= plen) return false;
    if (l <= 5 && (P.strategy == ZC_STRAT_FILTERED || (l == ZC_MIN_MATCH && d > ZC_TOO_FAR))) return false;
    *L = l; *D = d;
    return true;
}

template <class Rec>
ZC_HD uint32_t zc_resolve_pos(Rec rec, uint32_t p, uint32_t E, const ZcParams &P)
{
    uint32_t rf, rq, L, D;
    if (P.parser == ZC_PARSE_RLE) {                        /* deflate_rle (deflate.c:1861-1928): a run of three or more is taken as it stands */
        if (E - p < ZC_MIN_MATCH) return 0;
        rec(p, &rf, &rq);
        const uint32_t l = rf & 0xffffu;
        return l >= ZC_MIN_MATCH ? zc_hop_pack(0, l, 1) : 0u;
    }
    if (E - p < ZC_MIN_MATCH || P.lazy < ZC_MIN_MATCH) return 0;      /* prev_length (2) < max_lazy_match, deflate.c:1765 */
    rec(p, &rf, &rq);
    if (!zc_accept(rf, rq, ZC_MIN_MATCH - 1, P, &L, &D)) return 0;
    uint32_t k = 0;
    for (;;) {                                             /* deflate.c:1762-1837 at position p + k + 1 */
        uint32_t q = p + k + 1, L2, D2;
        if (L >= P.lazy || E - q < ZC_MIN_MATCH) break;
        rec(q, &rf, &rq);
        if (!zc_accept(rf, rq, L, P, &L2, &D2)) break;
        L = L2; D = D2; k++;
    }
    return zc_hop_pack(k, L, D);
}

/* The same hop computed from the data instead of from precomputed records: the
 * searches deflate_slow itself makes when it arrives at p with nothing pending.
 * zc_walk_direct is longest_match as the parser calls it (deflate.c:1148-1289):
 * best_len starts at prev_length, the chain budget is quartered from good_match on
 * (a budget of zero wraps in the reference's do-while and is as good as unbounded).
 * Only the positions the parser examines are searched. */
template <class Win, class Pd>
ZC_HD bool zc_walk_direct(const Win &win, const Pd &pd, uint32_t q, uint32_t E, uint32_t corner, const ZcParams &P,
                          uint32_t plen, uint32_t *len, uint32_t *dist)
{
    ZcMatchState s;
    if (!zc_match_init(win, pd, q, E, corner, P.nice, s, P.max_dist)) return false;
    if (plen >= s.cap) return false;                       /* nothing can be longer than prev_length */
    if (plen > ZC_MIN_MATCH - 1) { s.best = plen; s.scan3 = win.load4(q + plen - 2u) & 0x00ffffffu; }
    uint32_t budget = plen >= P.good ? P.chain >> 2 : P.chain;
    if (budget == 0) budget = 0x7fffffffu;
    while (!zc_match_step(win, pd, budget, s)) {}
    if (s.best <= plen) return false;
    *len = s.best; *dist = s.bdist;
    return true;
}
ZC_HD bool zc_filtered_out(uint32_t l, uint32_t d, const ZcParams &P)    /* deflate.c:1776-1785 */
{
    return l <= 5 && (P.strategy == ZC_STRAT_FILTERED || (l == ZC_MIN_MATCH && d > ZC_TOO_FAR));
}
template <class Win, class Pd>
ZC_HD uint32_t zc_hop_direct(const Win &win, const Pd &pd, uint32_t p, uint32_t E, uint32_t corner, const ZcParams &P)
{
    uint32_t L, D, L2, D2, k = 0;
    if (P.lazy < ZC_MIN_MATCH) return 0;                   /* prev_length (2) < max_lazy_match, deflate.c:1765 */
    if (!zc_walk_direct(win, pd, p, E, corner, P, ZC_MIN_MATCH - 1, &L, &D) || zc_filtered_out(L, D, P)) return 0;
    for (;;) {
        uint32_t q = p + k + 1;
        if (L >= P.lazy || E - q < ZC_MIN_MATCH) break;
        if (!zc_walk_direct(win, pd, q, E, corner, P, L, &L2, &D2) || zc_filtered_out(L2, D2, P)) break;
        L = L2; D = D2; k++;
    }
    return zc_hop_pack(k, L, D);
}

/* Window base (a-coordinate of window index 0) in force at the iteration that
 * starts at p_top, for a chunk that ends at E: fill_window slides by 32 KiB at
 * the first iteration top that is both short of lookahead (< look_thr bytes
 * left in the window) and at window index >= 65274 (deflate.c:1419, 1744). */
ZC_HD uint32_t zc_window_base(uint32_t p_top, uint32_t E, uint32_t look_thr, uint32_t wsize = ZC_WSIZE)
{
    /* the base in force at p_top is the smallest multiple B of wsize with p_top < thr(B), and thr(B) <= B + 2 wsize - 261:
     * every B below (p_top / wsize - 2) * wsize fails, so the search may start there (a small window slides thousands
     * of times in a chunk) */
    uint32_t B = p_top >= 3u * wsize ? (p_top / wsize - 3u) * wsize : 0u;
    for (;;) {
        uint64_t cap = (uint64_t)B + 2u * wsize;
        uint32_t wend = E < cap ? E : (uint32_t)cap;
        uint32_t t1 = wend >= look_thr - 1 ? wend - (look_thr - 1) : 0, t2 = B + wsize + (wsize - ZC_MIN_LOOK);
        uint32_t thr = t1 > t2 ? t1 : t2;
        if (p_top < thr) return B;
        B += wsize;
    }
}

/* Number of blocks a chunk of nsym symbols is cut into (lit_bufsize rule,
 * trees.c:1050) and block b's symbol range.  tail_literal: the last symbol is
 * the literal deflate_slow tallies after its loop (deflate.c:1840-1844), whose
 * "buffer full" result is ignored. */
ZC_HD uint32_t zc_block_count(uint32_t nsym, uint32_t lim, uint32_t final_, bool tail_literal)
{
    uint32_t full = nsym / lim, rem = nsym % lim;
    if (rem == 0 && nsym != 0 && tail_literal) return full;
    return full + ((final_ || rem) ? 1u : 0u);
}

/* deflate_rle() (deflate.c:1861-1928) over precomputed run lengths. */
template <class Rec, class Sym, class Blk>
ZC_HD uint32_t zc_parse_rle(uint32_t dict_len, uint32_t len, uint32_t final_, const ZcParams &P,
                            Rec rec, Sym sym, Blk blk, uint32_t *nsym_out)
{
    ZcWindow w; w.B = 0; w.E = dict_len + len; w.wend = dict_len; w.W = P.wsize;
    ZcSink<Sym, Blk> out(sym, blk, dict_len, P.lit_bufsize);
    uint32_t p = dict_len;
    for (;;) {
        if (w.wend - p <= ZC_MAX_MATCH) { w.refill(p); if (w.wend == p) break; }
        uint32_t l = 0;
        if (w.wend - p >= ZC_MIN_MATCH && p > w.B) { uint32_t rf, rq; rec(p, &rf, &rq); l = rf & 0xffffu; }
        bool f;
        if (l >= ZC_MIN_MATCH) { f = out.emit(zc_sym_match(l, 1)); p += l; }
        else { f = out.emit(p); p++; }
        if (f) out.flush(p, 0, w);
    }
    if (final_) out.flush(p, 1, w);
    else if (out.nsym != out.blk_sym0) out.flush(p, 0, w);
    *nsym_out = out.nsym;
    return out.nblk;
}

/* deflate_fast() (deflate.c:1628-1722), levels 1-3.  Which strings enter the
 * hash chains depends on the parse (the inside of a match longer than
 * max_insert_length is skipped, :1680-1697), so the chains cannot be built ahead
 * of it: this parser owns head[]/prev[] exactly as the reference does, through
 * Ctx:
 *   cx.insert(p, B)   INSERT_STRING: returns the previous head entry, a window
 *                     index (a-coordinate minus B), 0 = NIL
 *   cx.slide()        the window moved by 32 KiB: every entry m becomes
 *                     m >= 32768 ? m - 32768 : NIL (deflate.c:1431-1448)
 *   cx.longest(p, B, head, look, &len, &dist)   longest_match with prev_length 2
 * P.lazy holds max_insert_length for these levels (deflate.h:179-183). */
template <class Ctx, class Sym, class Blk>
ZC_HD uint32_t zc_parse_fast(uint32_t dict_len, uint32_t len, uint32_t final_, const ZcParams &P,
                             Ctx &cx, Sym sym, Blk blk, uint32_t *nsym_out)
{
    ZcWindow w; w.B = 0; w.E = dict_len + len; w.wend = dict_len; w.W = P.wsize;
    ZcSink<Sym, Blk> out(sym, blk, dict_len, P.lit_bufsize);
    /* deflateSetDictionary inserts every dictionary string; the last two wait for
     * the first input bytes (deflate.c:364-381, 1471-1489) */
    for (uint32_t a = 0; a < dict_len && a + ZC_MIN_MATCH <= w.E; a++) cx.insert(a, 0u);
    uint32_t p = dict_len;
    for (;;) {
        if (w.wend - p < ZC_MIN_LOOK) {
            uint32_t b0 = w.B;
            w.refill(p);
            if (w.B != b0) cx.slide();
            if (w.wend == p) break;
        }
        uint32_t look = w.wend - p, hh = 0, mlen = 0, mdist = 0;
        if (look >= ZC_MIN_MATCH) hh = cx.insert(p, w.B);
        if (hh != 0 && (p - w.B) - hh <= P.max_dist) cx.longest(p, w.B, hh, look, &mlen, &mdist);
        bool f;
        if (mlen >= ZC_MIN_MATCH) {
            f = out.emit(zc_sym_match(mlen, mdist));
            if (mlen <= P.lazy && look - mlen >= ZC_MIN_MATCH)
                for (uint32_t i = 1; i < mlen; i++) cx.insert(p + i, w.B);
            p += mlen;
        } else { f = out.emit(p); p++; }
        if (f) out.flush(p, 0, w);
    }
    if (final_) out.flush(p, 1, w);
    else if (out.nsym != out.blk_sym0) out.flush(p, 0, w);
    *nsym_out = out.nsym;
    return out.nblk;
}

/* deflate_stored() (deflate.c:1564-1619), level 0: no symbols, only the block
 * boundaries.  fill_window tops the window up to 2 * w_size whenever the lookahead
 * is used up (one deflate call hands over the whole chunk); a block ends at
 * min(0xffff, pending_buf_size - 5) bytes (:1571-1576, 1597-1603) or as soon as
 * it spans MAX_DIST (:1607-1609). */
template <class Blk>
ZC_HD uint32_t zc_parse_stored(uint32_t dict_len, uint32_t len, uint32_t final_, const ZcParams &P, Blk blk)
{
    ZcWindow w; w.B = 0; w.E = dict_len + len; w.wend = dict_len; w.W = P.wsize;
    uint32_t max_block = P.lit_bufsize * 4u - 5u;
    if (max_block > 0xffffu) max_block = 0xffffu;
    uint32_t p = dict_len, look = 0, bstart = dict_len, nblk = 0;
    auto flush = [&](uint32_t last) {
        ZcBlock b; b.sym_start = 0; b.sym_count = 0; b.in_start = bstart; b.in_len = p - bstart;
        b.flags = ZC_BF_CANSTORE | ZC_BF_STORED | (last ? ZC_BF_LAST : 0u); b.type = 0; b.bits = 0; b.hdr_bits = 0; b.bit_off = 0;
        blk(nblk++, b); bstart = p;
    };
    for (;;) {
        if (look <= 1) { w.refill(p); look = w.wend - p; if (look == 0) break; }
        p += look; look = 0;
        const uint32_t max_start = bstart + max_block;
        if (p >= max_start) { look = p - max_start; p = max_start; flush(0); }
        if (p - bstart >= P.max_dist) flush(0);
    }
    if (final_) flush(1);
    else if (p > bstart) flush(0);
    return nblk;
}

/* deflate_huff() (deflate.c:1934-1967): every byte a literal, so the block
 * table follows from the chunk length alone. */
template <class Blk>
ZC_HD uint32_t zc_parse_huff(uint32_t dict_len, uint32_t len, uint32_t final_, const ZcParams &P, Blk blk)
{
    /* A block can be stored only while its start is still inside the window (deflate.c:1539-1541).  fill_window runs
     * when the lookahead is used up (:1942), so the base in force is that of the iteration in which the flush happens:
     * the one of the block's last byte, or, for the block flushed after the loop, the one that found no more input. */
    const uint32_t lim = P.lit_bufsize - 1, E = dict_len + len;
    uint32_t nblk = 0, done = 0;
    while (len - done >= lim) {
        ZcBlock b; b.sym_start = done; b.sym_count = lim; b.in_start = dict_len + done; b.in_len = lim;
        b.flags = ZC_BF_IMPLICIT | (b.in_start >= zc_window_base(b.in_start + lim - 1u, E, 1u, P.wsize) ? ZC_BF_CANSTORE : 0u);
        b.type = 0; b.bits = 0; b.hdr_bits = 0; b.bit_off = 0;
        blk(nblk++, b); done += lim;
    }
    if (final_ || len != done) {
        ZcBlock b; b.sym_start = done; b.sym_count = len - done; b.in_start = dict_len + done; b.in_len = len - done;
        b.flags = ZC_BF_IMPLICIT | (final_ ? ZC_BF_LAST : 0u) | (b.in_start >= zc_window_base(E, E, 1u, P.wsize) ? ZC_BF_CANSTORE : 0u);
        b.type = 0; b.bits = 0; b.hdr_bits = 0; b.bit_off = 0;
        blk(nblk++, b);
    }
    return nblk;
}

/* ------------------------------------------------------------------ Huffman
 * build_tree / gen_bitlen (trees.c:488-565, 617-699) executed by one lane.
 * N = number of leaves, HEAP = 2N+1.  freq[] holds leaf frequencies on entry
 * (internal nodes are written during construction); len[] receives the code
 * lengths, bl_count[] the per-length counts for gen_codes. */
template <int N>
struct ZcTree {
    enum { HEAP = 2 * N + 1 };
    uint32_t pq[HEAP + 1];         /* from the bottom: the priority queue of build_tree, freq << 16 | depth << 10 | node (see
                                    * zc_build_tree); from the top: the nodes in the order they left the queue, read by gen_bitlen —
                                    * one array for both, as in the reference (heap[] and heap_max, trees.c:664, 682) */
    uint16_t freq[N];              /* of the leaves; an inner node's frequency lives in its queue word only */
    uint16_t dad[HEAP];
    uint8_t  len[HEAP + 3];
    uint16_t bl_count[16];
    int32_t  max_code;
    uint32_t deep;                 /* set when a subtree grew deeper than the packed key can hold (it cannot: see below) */
};

/* The queue holds one word per node: frequency (16 bits: a block has at most lit_bufsize symbols), depth of the subtree
 * (6 bits: a subtree of depth d weighs at least Fibonacci(d), so depth stays below 24 for frequencies that add up to
 * less than 65536) and node number (10 bits).  smaller(n, m) of trees.c:443-445 — frequency, then depth, "<=" on
 * ties — is then one comparison of the words without their node bits, and pqdownheap never touches freq[] or
 * depth[]: the serial dependent loads per level drop from six to one. */
#define ZC_PQ_KEY(e) ((e) >> 10)
template <int N>
ZC_HD void zc_sift(ZcTree<N> &t, int heap_len, int k)               /* pqdownheap, trees.c:453 */
{
    const uint32_t v = t.pq[k];
    int j = k << 1;
    while (j <= heap_len) {
        uint32_t a = t.pq[j];
        /* key(x) <= key(a) is x <= (a | 1023): one OR instead of two shifts */
        if (j < heap_len) { const uint32_t b = t.pq[j + 1]; if (b <= (a | 1023u)) { j++; a = b; } }
        if (v <= (a | 1023u)) break;
        t.pq[k] = a; k = j; j <<= 1;
    }
    t.pq[k] = v;
}

/* stlen(n) = static code length of symbol n (or 0 when the tree has no static
 * counterpart), xbits(n) = extra bits of symbol n. */
template <int N, class StLen, class XBits>
ZC_HD void zc_build_tree(ZcTree<N> &t, int max_length, bool has_static, StLen stlen, XBits xbits,
                         uint32_t *opt_len, uint32_t *static_len)
{
    const int HEAP = ZcTree<N>::HEAP;
    int heap_len = 0, heap_max = HEAP, max_code = -1, n, m, node;
    t.deep = 0;
    for (n = 0; n < N; n++) {
        if (t.freq[n] != 0) { t.pq[++heap_len] = ((uint32_t)t.freq[n] << 16) | (uint32_t)(max_code = n); }
        else t.len[n] = 0;
    }
    while (heap_len < 2) {                                           /* trees.c:648-654 */
        node = (max_code < 2 ? ++max_code : 0);
        t.pq[++heap_len] = (1u << 16) | (uint32_t)node;
        t.freq[node] = 1;
        (*opt_len)--; if (has_static) *static_len -= stlen(node);
    }
    t.max_code = max_code;
    for (n = heap_len / 2; n >= 1; n--) zc_sift(t, heap_len, n);
    node = N;
    do {
        const uint32_t en = t.pq[1];
        t.pq[1] = t.pq[heap_len--]; zc_sift(t, heap_len, 1);
        const uint32_t em = t.pq[1];
        n = (int)(en & 1023u); m = (int)(em & 1023u);
        t.pq[--heap_max] = (uint32_t)n; t.pq[--heap_max] = (uint32_t)m;      /* the queue has shrunk below heap_max: heap_len + 2 (iterations so far) < HEAP */
        const uint32_t f = (en >> 16) + (em >> 16);
        const uint32_t dn = (en >> 10) & 63u, dm = (em >> 10) & 63u;
        uint32_t d = (dn >= dm ? dn : dm) + 1u;
        if (d > 63u) { d = 63u; t.deep = 1; }
        t.dad[n] = t.dad[m] = (uint16_t)node;
        t.pq[1] = (f << 16) | (d << 10) | (uint32_t)node;
        node++;
        zc_sift(t, heap_len, 1);
    } while (heap_len >= 2);
    { const uint32_t root = t.pq[1] & 1023u; t.pq[--heap_max] = root; }

    /* gen_bitlen */
    int h, bits, overflow = 0;
    for (bits = 0; bits <= 15; bits++) t.bl_count[bits] = 0;
    t.len[t.pq[heap_max]] = 0;
    for (h = heap_max + 1; h < HEAP; h++) {
        n = (int)t.pq[h];
        bits = t.len[t.dad[n]] + 1;
        if (bits > max_length) { bits = max_length; overflow++; }
        t.len[n] = (uint8_t)bits;
        if (n > max_code) continue;
        t.bl_count[bits]++;
        uint32_t xb = xbits(n), f = t.freq[n];
        *opt_len += f * ((uint32_t)bits + xb);
        if (has_static) *static_len += f * (stlen(n) + xb);
    }
    if (overflow == 0) return;
    do {
        bits = max_length - 1;
        while (t.bl_count[bits] == 0) bits--;
        t.bl_count[bits]--; t.bl_count[bits + 1] += 2; t.bl_count[max_length]--;
        overflow -= 2;
    } while (overflow > 0);
    for (bits = max_length; bits != 0; bits--) {
        n = t.bl_count[bits];
        while (n != 0) {
            m = (int)t.pq[--h];
            if (m > max_code) continue;
            if ((int)t.len[m] != bits) {
                *opt_len += (uint32_t)((bits - (int)t.len[m]) * (int)t.freq[m]);
                t.len[m] = (uint8_t)bits;
            }
            n--;
        }
    }
}

/* gen_codes (trees.c:575-607): code | len<<16 for symbols 0..max_code into tab[]. */
template <int N, class Tab>
ZC_HD void zc_assign_codes(const ZcTree<N> &t, Tab &tab, int count)
{
    uint32_t next[16], c = 0;
    for (int b = 1; b <= 15; b++) { c = (c + t.bl_count[b - 1]) << 1; next[b] = c; }
    for (int n = 0; n < count; n++) {
        uint32_t l = n <= t.max_code ? t.len[n] : 0u;
        tab[n] = l ? (zc_bitrev(next[l]++, l) | (l << 16)) : 0u;
    }
}

/* LSB-first bit writer (send_bits, trees.c:213) into 32-bit words, little endian: bits gather in a register and leave a
 * word at a time, so the buffer is only ever written (and need not start zeroed); flush() writes the last, partial word
 * with zeros above its bits.  v holds nothing above its n bits. */
struct ZcBitW {
    uint32_t *buf; uint32_t nbits;
    uint64_t acc; uint32_t fill, widx;
    ZC_HD void begin(void *words) { buf = (uint32_t *)words; nbits = 0; acc = 0; fill = 0; widx = 0; }
    ZC_HD void put(uint32_t v, uint32_t n)
    {
        acc |= (uint64_t)v << fill; fill += n; nbits += n;
        if (fill >= 32u) { buf[widx++] = (uint32_t)acc; acc >>= 32; fill -= 32u; }
    }
    ZC_HD void flush() { if (fill) buf[widx] = (uint32_t)acc; }
};

/* scan_tree / send_tree (trees.c:705-795).  With w == nullptr counts the
 * bit-length symbols into blfreq[], else writes them with the codes in bltab. */
template <class Len>
ZC_HD void zc_walk_lengths(Len len, int max_code, uint16_t *blfreq, const uint32_t *bltab, ZcBitW *w)
{
    int prevlen = -1, curlen, nextlen = (int)len(0), count = 0, maxc = 7, minc = 4;
    if (nextlen == 0) { maxc = 138; minc = 3; }
    for (int n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = (n == max_code) ? 0xffff : (int)len(n + 1);
        if (++count < maxc && curlen == nextlen) continue;
        if (count < minc) {
            if (w) { do w->put(bltab[curlen] & 0xffffu, bltab[curlen] >> 16); while (--count); }
            else blfreq[curlen] = (uint16_t)(blfreq[curlen] + count);
        } else if (curlen != 0) {
            if (curlen != prevlen) {
                if (w) { w->put(bltab[curlen] & 0xffffu, bltab[curlen] >> 16); count--; }
                else blfreq[curlen]++;
            }
            if (w) { w->put(bltab[16] & 0xffffu, bltab[16] >> 16); w->put((uint32_t)(count - 3), 2); }
            else blfreq[16]++;
        } else if (count <= 10) {
            if (w) { w->put(bltab[17] & 0xffffu, bltab[17] >> 16); w->put((uint32_t)(count - 3), 3); }
            else blfreq[17]++;
        } else {
            if (w) { w->put(bltab[18] & 0xffffu, bltab[18] >> 16); w->put((uint32_t)(count - 11), 7); }
            else blfreq[18]++;
        }
        count = 0; prevlen = curlen;
        if (nextlen == 0) { maxc = 138; minc = 3; }
        else if (curlen == nextlen) { maxc = 6; minc = 3; }
        else { maxc = 7; minc = 4; }
    }
}

#define ZC_HDR_BYTES 320     /* 14 + 19*3 + 316*7 bits = 2283 bits < 2560 */

/* Per-block coding tables produced by the planner and read by the emitter. */
struct ZcBlockTab {
    uint32_t lc[288];        /* code | len<<16 */
    uint32_t dc[32];
    uint8_t  hdr[ZC_HDR_BYTES];   /* the tree description (send_all_trees) as little-endian 32-bit words; bits past hdr_bits are not defined */
};

/* The serial tail of _tr_flush_block() (trees.c:938-989) once the literal and
 * distance trees exist: bit-length tree, cost comparison, block type, header
 * bits and code tables.  Runs on one lane. */
ZC_HD void zc_plan_block(ZcTree<ZC_L_CODES> &lt, ZcTree<ZC_D_CODES> &dt, ZcTree<ZC_BL_CODES> &bt,
                         uint32_t opt_len, uint32_t static_len, uint32_t strategy,
                         ZcBlock *b, ZcBlockTab *tab)
{
    static const uint8_t bl_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    for (int i = 0; i < ZC_BL_CODES; i++) bt.freq[i] = 0;
    /* build_bl_tree, trees.c:801-829 */
    zc_walk_lengths([&](int n) { return lt.len[n]; }, lt.max_code, bt.freq, nullptr, nullptr);
    zc_walk_lengths([&](int n) { return dt.len[n]; }, dt.max_code, bt.freq, nullptr, nullptr);
    uint32_t dummy = 0;
    zc_build_tree(bt, 7, false, [](int) { return 0u; },
                  [](int n) { return n == 16 ? 2u : n == 17 ? 3u : n == 18 ? 7u : 0u; }, &opt_len, &dummy);
    int max_bl;
    for (max_bl = ZC_BL_CODES - 1; max_bl >= 3; max_bl--) if (bt.len[bl_order[max_bl]] != 0) break;
    opt_len += 3u * ((uint32_t)max_bl + 1u) + 5u + 5u + 4u;
    uint32_t opt_lenb = (opt_len + 3 + 7) >> 3, static_lenb = (static_len + 3 + 7) >> 3;
    if (static_lenb <= opt_lenb) opt_lenb = static_lenb;                       /* trees.c:948 */
    b->hdr_bits = 0;
    if (b->in_len + 4 <= opt_lenb && (b->flags & ZC_BF_CANSTORE)) {            /* trees.c:958 */
        b->type = 0; b->bits = 0;
    } else if (strategy == ZC_STRAT_FIXED || static_lenb == opt_lenb) {        /* trees.c:972 */
        b->type = 1; b->bits = 3 + static_len;
        for (uint32_t n = 0; n < 288; n++) tab->lc[n] = zc_static_lcode(n);
        for (uint32_t n = 0; n < 30; n++) tab->dc[n] = zc_static_dcode(n);
    } else {
        b->type = 2; b->bits = 3 + opt_len;
        zc_assign_codes(lt, tab->lc, 288);
        zc_assign_codes(dt, tab->dc, 32);
        uint32_t bltab[ZC_BL_CODES];
        zc_assign_codes(bt, bltab, ZC_BL_CODES);
        ZcBitW w; w.begin(tab->hdr);
        /* send_all_trees, trees.c:836-860 */
        w.put((uint32_t)(lt.max_code + 1 - 257), 5);
        w.put((uint32_t)(dt.max_code + 1 - 1), 5);
        w.put((uint32_t)(max_bl + 1 - 4), 4);
        for (int r = 0; r <= max_bl; r++) w.put(bt.len[bl_order[r]], 3);
        zc_walk_lengths([&](int n) { return lt.len[n]; }, lt.max_code, nullptr, bltab, &w);
        zc_walk_lengths([&](int n) { return dt.len[n]; }, dt.max_code, nullptr, bltab, &w);
        w.flush();
        b->hdr_bits = w.nbits;
    }
}

/* Bit layout of one chunk: walks the block table, assigns bit offsets relative
 * to the chunk start and returns the chunk's byte length.  Stored blocks pad to
 * a byte boundary after their 3 header bits (copy_block, trees.c:1205-1219);
 * a non-final chunk ends with the empty stored block of Z_SYNC_FLUSH
 * (deflate.c:926-927), a final one is padded by bi_windup (trees.c:996-1001). */
template <class BlkAt>
ZC_HD uint64_t zc_layout_chunk(BlkAt blk_at, uint32_t nblk, uint32_t final_, uint64_t *sync_bit_off, uint32_t start_bit = 0,
                               uint32_t tail = ZC_TAIL_SYNC, uint32_t *left_bits = nullptr)
{
    /* tail != ZC_TAIL_SYNC: the chunk ends inside a byte; the return value counts its complete bytes and *left_bits
     * the bits of the next one (bi_flush keeps them in the bit buffer, trees.c:1169-1181) */
    uint64_t bit = start_bit;      /* bits deflatePrime left in the bit buffer (deflate.c:464-487) */
    for (uint32_t i = 0; i < nblk; i++) {
        ZcBlock *b = blk_at(i);
        b->bit_off = bit;
        if (b->type == 0) { bit += 3; bit = (bit + 7) & ~7ull; bit += 32 + 8ull * b->in_len; }
        else bit += b->bits;
    }
    if (left_bits) *left_bits = 0;
    if (final_) bit = (bit + 7) & ~7ull;
    else if (tail == ZC_TAIL_SYNC) { *sync_bit_off = bit; bit += 3; bit = (bit + 7) & ~7ull; bit += 32; }
    else {
        *sync_bit_off = bit;
        if (tail == ZC_TAIL_ALIGN) bit += 10;      /* 3 header bits of a static block + the 7-bit end-of-block code */
        if (left_bits) *left_bits = (uint32_t)(bit & 7u);
    }
    return bit >> 3;
}

#endif /* ZAMD_CORE_H */
