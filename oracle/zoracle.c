/*
 * zoracle.c — CPU oracle (TEST INFRASTRUCTURE ONLY; never linked into the product).
 *
 * Restates, in absolute buffer coordinates, what zlib 1.2.8 computes for one
 * deflate() call over a dictionary-primed chunk.  See zoracle.h for the parity
 * pin.  Citations are file:line in the reference tree.
 */
#include "zoracle.h"
#include <stdlib.h>
#include <string.h>

#define MINM 3          /* MIN_MATCH, zutil.h:75 */
#define MAXM 258        /* MAX_MATCH, zutil.h:76 */
#define MIN_LOOK (MAXM + MINM + 1)   /* MIN_LOOKAHEAD, deflate.h:281 */
#define TOO_FAR 4096    /* deflate.c:107-109 */
#define L_CODES 286
#define D_CODES 30
#define BL_CODES 19
#define HEAP_SZ (2 * L_CODES + 1)
#define END_BLOCK 256

/* ------------------------------------------------------------------ tables */
/* trees.c:62-72 */
static const uint8_t x_lbits[29] = {0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0};
static const uint8_t x_dbits[30] = {0,0,0,0,1,1,2,2,3,3,4,4,5,5,6,6,7,7,8,8,9,9,10,10,11,11,12,12,13,13};
static const uint8_t x_blbits[19] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,2,3,7};
static const uint8_t bl_order[19] = {16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1,15};

static uint8_t  len_code[256], dist_code[512];
static uint16_t base_len[29], base_dist[30];
static uint16_t st_lcode[288], st_dcode[30];
static uint8_t  st_llen[288], st_dlen[30];
static int tables_ready;

static unsigned bitrev(unsigned v, int n)          /* trees.c:1154 */
{
    unsigned r = 0;
    while (n-- > 0) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

/* canonical code assignment, trees.c:575-607 */
static void assign_codes(uint16_t *code, const uint8_t *len, int max_code, const uint16_t *bl_count)
{
    uint16_t next[16]; unsigned c = 0;
    for (int b = 1; b <= 15; b++) { c = (c + bl_count[b - 1]) << 1; next[b] = (uint16_t)c; }
    for (int n = 0; n <= max_code; n++)
        if (len[n]) code[n] = (uint16_t)bitrev(next[len[n]]++, len[n]);
}

static void build_tables(void)                     /* trees.c:234-314 */
{
    if (tables_ready) return;
    int length = 0, code, dist = 0;
    for (code = 0; code < 28; code++) {
        base_len[code] = (uint16_t)length;
        for (int n = 0; n < (1 << x_lbits[code]); n++) len_code[length++] = (uint8_t)code;
    }
    len_code[255] = 28; base_len[28] = 255;  /* base_length[28] is never read with extra bits */
    for (code = 0; code < 16; code++) {
        base_dist[code] = (uint16_t)dist;
        for (int n = 0; n < (1 << x_dbits[code]); n++) dist_code[dist++] = (uint8_t)code;
    }
    dist >>= 7;
    for (; code < D_CODES; code++) {
        base_dist[code] = (uint16_t)(dist << 7);
        for (int n = 0; n < (1 << (x_dbits[code] - 7)); n++) dist_code[256 + dist++] = (uint8_t)code;
    }
    uint16_t cnt[16] = {0};
    for (int n = 0; n < 288; n++) {
        st_llen[n] = n <= 143 ? 8 : n <= 255 ? 9 : n <= 279 ? 7 : 8;
        cnt[st_llen[n]]++;
    }
    assign_codes(st_lcode, st_llen, 287, cnt);
    for (int n = 0; n < D_CODES; n++) { st_dlen[n] = 5; st_dcode[n] = (uint16_t)bitrev((unsigned)n, 5); }
    tables_ready = 1;
}

static inline unsigned d_code(unsigned d)          /* deflate.h:305 */
{
    return d < 256 ? dist_code[d] : dist_code[256 + (d >> 7)];
}

/* deflate.c:131-143 : good_length, max_lazy, nice_length, max_chain, parser */
enum { P_STORED, P_FAST, P_SLOW };
static const struct { uint16_t good, lazy, nice, chain; int parser; } cfg[10] = {
    {0, 0, 0, 0, P_STORED}, {4, 4, 8, 4, P_FAST}, {4, 5, 16, 8, P_FAST}, {4, 6, 32, 32, P_FAST},
    {4, 4, 16, 16, P_SLOW}, {8, 16, 32, 32, P_SLOW}, {8, 16, 128, 128, P_SLOW},
    {8, 32, 128, 256, P_SLOW}, {32, 128, 258, 1024, P_SLOW}, {32, 258, 258, 4096, P_SLOW}};

/* ------------------------------------------------------------------ state */
typedef struct {
    uint16_t freq[HEAP_SZ], dad[HEAP_SZ], code[HEAP_SZ];
    uint8_t  len[HEAP_SZ + 1];
    int max_code;
} tree_t;

typedef struct {
    const uint8_t *stlen;   /* static lengths or NULL */
    const uint8_t *extra; int extra_base, elems, max_length;
} tdesc_t;

static const tdesc_t L_DESC  = {st_llen, x_lbits, 257, L_CODES, 15};
static const tdesc_t D_DESC  = {st_dlen, x_dbits, 0, D_CODES, 15};
static const tdesc_t BL_DESC = {NULL, x_blbits, 0, BL_CODES, 7};

typedef struct {
    const uint8_t *buf; uint32_t dict_len, end;
    int level, strategy;
    uint32_t wsize, wmask, max_dist, hbits, hmask, hshift, lit_bufsize;
    uint32_t good, lazy, nice, chain;
    uint32_t base, wend, pending_ins;        /* window model */
    uint32_t *head, *prev;                   /* entries hold position+1, 0 = empty */
    uint16_t *d_buf; uint8_t *l_buf; uint32_t *p_buf; uint32_t last_lit, matches, block_start;
    tree_t lt, dt, bt;
    int heap[HEAP_SZ], heap_len, heap_max; uint8_t depth[HEAP_SZ]; uint16_t bl_count[16];
    uint64_t opt_len, static_len;
    uint8_t *out; size_t out_cap, out_n; uint64_t bitbuf; int bitcnt; int overflow;
    uint64_t bits_total;
    int data_type;                           /* strm->data_type: 2 = Z_UNKNOWN until the first block at level > 0 */
    zo_trace *tr; size_t tr_sym_base;
} enc_t;

/* ------------------------------------------------------------------ bit output
 * send_bits / bi_flush / bi_windup (trees.c:213-225, 1169-1199) reduce to plain
 * LSB-first concatenation. */
static void put_bits(enc_t *s, unsigned v, int n)
{
    s->bitbuf |= (uint64_t)v << s->bitcnt; s->bitcnt += n; s->bits_total += (unsigned)n;
    while (s->bitcnt >= 8) {
        if (s->out_n < s->out_cap) s->out[s->out_n++] = (uint8_t)s->bitbuf; else s->overflow = 1;
        s->bitbuf >>= 8; s->bitcnt -= 8;
    }
}
static void byte_align(enc_t *s)
{
    if (s->bitcnt) { int pad = 8 - s->bitcnt; put_bits(s, 0, pad); }
}

/* ------------------------------------------------------------------ Huffman
 * smaller(), pqdownheap(): trees.c:443-476 */
#define SMALLER(t, n, m) ((t)->freq[n] < (t)->freq[m] || ((t)->freq[n] == (t)->freq[m] && s->depth[n] <= s->depth[m]))
static void sift_down(enc_t *s, tree_t *t, int k)
{
    int v = s->heap[k], j = k << 1;
    while (j <= s->heap_len) {
        if (j < s->heap_len && SMALLER(t, s->heap[j + 1], s->heap[j])) j++;
        if (SMALLER(t, v, s->heap[j])) break;
        s->heap[k] = s->heap[j]; k = j; j <<= 1;
    }
    s->heap[k] = v;
}

/* gen_bitlen(): trees.c:488-565 */
static void limit_lengths(enc_t *s, tree_t *t, const tdesc_t *d)
{
    int h, n, m, bits, xbits, overflow = 0, maxl = d->max_length;
    memset(s->bl_count, 0, sizeof s->bl_count);
    t->len[s->heap[s->heap_max]] = 0;
    for (h = s->heap_max + 1; h < HEAP_SZ; h++) {
        n = s->heap[h];
        bits = t->len[t->dad[n]] + 1;
        if (bits > maxl) { bits = maxl; overflow++; }
        t->len[n] = (uint8_t)bits;
        if (n > t->max_code) continue;
        s->bl_count[bits]++;
        xbits = n >= d->extra_base ? d->extra[n - d->extra_base] : 0;
        s->opt_len += (uint64_t)t->freq[n] * (unsigned)(bits + xbits);
        if (d->stlen) s->static_len += (uint64_t)t->freq[n] * (unsigned)(d->stlen[n] + xbits);
    }
    if (!overflow) return;
    do {
        bits = maxl - 1;
        while (s->bl_count[bits] == 0) bits--;
        s->bl_count[bits]--; s->bl_count[bits + 1] += 2; s->bl_count[maxl]--;
        overflow -= 2;
    } while (overflow > 0);
    for (bits = maxl; bits != 0; bits--) {
        n = s->bl_count[bits];
        while (n != 0) {
            m = s->heap[--h];
            if (m > t->max_code) continue;
            if (t->len[m] != bits) {
                s->opt_len += (uint64_t)(((int64_t)bits - (int64_t)t->len[m]) * (int64_t)t->freq[m]);
                t->len[m] = (uint8_t)bits;
            }
            n--;
        }
    }
}

/* build_tree(): trees.c:617-699 */
static void make_tree(enc_t *s, tree_t *t, const tdesc_t *d)
{
    int n, m, node, max_code = -1;
    s->heap_len = 0; s->heap_max = HEAP_SZ;
    for (n = 0; n < d->elems; n++) {
        if (t->freq[n]) { s->heap[++s->heap_len] = max_code = n; s->depth[n] = 0; }
        else t->len[n] = 0;
    }
    while (s->heap_len < 2) {
        node = s->heap[++s->heap_len] = (max_code < 2 ? ++max_code : 0);
        t->freq[node] = 1; s->depth[node] = 0;
        s->opt_len--; if (d->stlen) s->static_len -= d->stlen[node];
    }
    t->max_code = max_code;
    for (n = s->heap_len / 2; n >= 1; n--) sift_down(s, t, n);
    node = d->elems;
    do {
        n = s->heap[1]; s->heap[1] = s->heap[s->heap_len--]; sift_down(s, t, 1);
        m = s->heap[1];
        s->heap[--s->heap_max] = n; s->heap[--s->heap_max] = m;
        t->freq[node] = (uint16_t)(t->freq[n] + t->freq[m]);
        s->depth[node] = (uint8_t)((s->depth[n] >= s->depth[m] ? s->depth[n] : s->depth[m]) + 1);
        t->dad[n] = t->dad[m] = (uint16_t)node;
        s->heap[1] = node++;
        sift_down(s, t, 1);
    } while (s->heap_len >= 2);
    s->heap[--s->heap_max] = s->heap[1];
    limit_lengths(s, t, d);
    assign_codes(t->code, t->len, max_code, s->bl_count);
}

/* scan_tree()/send_tree(): trees.c:705-795.  emit==0 counts into bt.freq,
 * emit==1 writes the codes. */
static void walk_lengths(enc_t *s, const tree_t *t, int max_code, int emit)
{
    int prevlen = -1, curlen, nextlen = t->len[0], count = 0, maxc = 7, minc = 4;
    if (nextlen == 0) { maxc = 138; minc = 3; }
    for (int n = 0; n <= max_code; n++) {
        curlen = nextlen; nextlen = (n == max_code) ? 0xffff : t->len[n + 1];
        if (++count < maxc && curlen == nextlen) continue;
        if (count < minc) {
            if (emit) { do put_bits(s, s->bt.code[curlen], s->bt.len[curlen]); while (--count); }
            else s->bt.freq[curlen] = (uint16_t)(s->bt.freq[curlen] + count);
        } else if (curlen != 0) {
            if (curlen != prevlen) {
                if (emit) { put_bits(s, s->bt.code[curlen], s->bt.len[curlen]); count--; }
                else s->bt.freq[curlen]++;
            }
            if (emit) { put_bits(s, s->bt.code[16], s->bt.len[16]); put_bits(s, (unsigned)(count - 3), 2); }
            else s->bt.freq[16]++;
        } else if (count <= 10) {
            if (emit) { put_bits(s, s->bt.code[17], s->bt.len[17]); put_bits(s, (unsigned)(count - 3), 3); }
            else s->bt.freq[17]++;
        } else {
            if (emit) { put_bits(s, s->bt.code[18], s->bt.len[18]); put_bits(s, (unsigned)(count - 11), 7); }
            else s->bt.freq[18]++;
        }
        count = 0; prevlen = curlen;
        if (nextlen == 0) { maxc = 138; minc = 3; }
        else if (curlen == nextlen) { maxc = 6; minc = 3; }
        else { maxc = 7; minc = 4; }
    }
}

static void reset_block(enc_t *s)                  /* init_block(), trees.c:409 */
{
    memset(s->lt.freq, 0, L_CODES * sizeof(uint16_t));
    memset(s->dt.freq, 0, D_CODES * sizeof(uint16_t));
    memset(s->bt.freq, 0, BL_CODES * sizeof(uint16_t));
    s->lt.freq[END_BLOCK] = 1;
    s->opt_len = s->static_len = 0; s->last_lit = s->matches = 0;
}

/* compress_block(): trees.c:1060-1105 */
static void emit_symbols(enc_t *s, const uint16_t *lc_, const uint8_t *ll_, const uint16_t *dc_, const uint8_t *dl_)
{
    for (uint32_t i = 0; i < s->last_lit; i++) {
        unsigned dist = s->d_buf[i], lc = s->l_buf[i];
        if (dist == 0) { put_bits(s, lc_[lc], ll_[lc]); continue; }
        unsigned c = len_code[lc];
        put_bits(s, lc_[c + 257], ll_[c + 257]);
        if (x_lbits[c]) put_bits(s, lc - base_len[c], x_lbits[c]);
        dist--; c = d_code(dist);
        put_bits(s, dc_[c], dl_[c]);
        if (x_dbits[c]) put_bits(s, dist - base_dist[c], x_dbits[c]);
    }
    put_bits(s, lc_[END_BLOCK], ll_[END_BLOCK]);
}

static void trace_block(enc_t *s, uint32_t in_len, int type, int last, int can_store, uint64_t bit_start)
{
    zo_trace *t = s->tr; if (!t) return;
    if (t->n_blocks == t->cap_blocks) {
        t->cap_blocks = t->cap_blocks ? 2 * t->cap_blocks : 64;
        t->blocks = (zo_block *)realloc(t->blocks, t->cap_blocks * sizeof(zo_block));
    }
    zo_block *b = &t->blocks[t->n_blocks++];
    b->sym_start = (uint32_t)s->tr_sym_base; b->sym_count = s->last_lit;
    b->in_start = s->block_start; b->in_len = in_len; b->type = (uint32_t)type; b->last = (uint32_t)last;
    b->can_store = (uint32_t)can_store; b->opt_len = s->opt_len; b->static_len = s->static_len;
    b->bit_start = bit_start; b->bit_end = s->bits_total;
    if (t->n_syms + s->last_lit > t->cap_syms) {
        t->cap_syms = 2 * (t->n_syms + s->last_lit) + 1024;
        t->syms = (zo_sym *)realloc(t->syms, t->cap_syms * sizeof(zo_sym));
    }
    for (uint32_t i = 0; i < s->last_lit; i++) {
        zo_sym *y = &t->syms[t->n_syms++];
        y->pos = s->p_buf[i]; y->dist = s->d_buf[i]; y->lc = s->l_buf[i];
    }
    s->tr_sym_base = t->n_syms;
}

/* _tr_flush_block() + FLUSH_BLOCK_ONLY: trees.c:907-1004, deflate.c:1538-1547.
 * `p` is strstart at flush time. */
static void flush_block(enc_t *s, uint32_t p, int last)
{
    uint32_t stored_len = p - s->block_start;
    int can_store = s->block_start >= s->base;      /* block_start >= 0 in window terms */
    uint64_t bit_start = s->bits_total;
    int type;
    int max_bl = 2;
    uint64_t opt_lenb, static_lenb;
    if (s->level > 0) {
        if (s->data_type == 2) {                    /* detect_data_type(), trees.c:920-921, 1120-1147 */
            unsigned long black_mask = 0xf3ffc07fUL; int n;
            s->data_type = -1;
            for (n = 0; n <= 31; n++, black_mask >>= 1)
                if ((black_mask & 1) && s->lt.freq[n] != 0) { s->data_type = 0; break; }
            if (s->data_type < 0) {
                if (s->lt.freq[9] != 0 || s->lt.freq[10] != 0 || s->lt.freq[13] != 0) s->data_type = 1;
                else for (n = 32; n < 256; n++) if (s->lt.freq[n] != 0) { s->data_type = 1; break; }
            }
            if (s->data_type < 0) s->data_type = 0;
        }
        make_tree(s, &s->lt, &L_DESC);
        make_tree(s, &s->dt, &D_DESC);
        /* build_bl_tree(): trees.c:801-829 */
        walk_lengths(s, &s->lt, s->lt.max_code, 0);
        walk_lengths(s, &s->dt, s->dt.max_code, 0);
        make_tree(s, &s->bt, &BL_DESC);
        for (max_bl = BL_CODES - 1; max_bl >= 3; max_bl--) if (s->bt.len[bl_order[max_bl]]) break;
        s->opt_len += 3 * ((uint64_t)max_bl + 1) + 5 + 5 + 4;
        opt_lenb = (s->opt_len + 3 + 7) >> 3; static_lenb = (s->static_len + 3 + 7) >> 3;
        if (static_lenb <= opt_lenb) opt_lenb = static_lenb;
    } else opt_lenb = static_lenb = (uint64_t)stored_len + 5;      /* force a stored block, trees.c:945-948 */

    if ((uint64_t)stored_len + 4 <= opt_lenb && can_store) {
        type = 0;                                   /* _tr_stored_block, trees.c:865 */
        uint64_t ol = s->opt_len, sl = s->static_len;
        put_bits(s, (0 << 1) + (unsigned)last, 3);
        byte_align(s);
        put_bits(s, stored_len & 0xffff, 16); put_bits(s, (~stored_len) & 0xffff, 16);
        for (uint32_t i = 0; i < stored_len; i++) put_bits(s, s->buf[s->block_start + i], 8);
        s->opt_len = ol; s->static_len = sl;
    } else if (s->strategy == ZO_FIXED || static_lenb == opt_lenb) {
        type = 1;
        put_bits(s, (1 << 1) + (unsigned)last, 3);
        emit_symbols(s, st_lcode, st_llen, st_dcode, st_dlen);
    } else {
        type = 2;
        put_bits(s, (2 << 1) + (unsigned)last, 3);
        /* send_all_trees(): trees.c:836-860 */
        put_bits(s, (unsigned)(s->lt.max_code + 1 - 257), 5);
        put_bits(s, (unsigned)(s->dt.max_code + 1 - 1), 5);
        put_bits(s, (unsigned)(max_bl + 1 - 4), 4);
        for (int r = 0; r <= max_bl; r++) put_bits(s, s->bt.len[bl_order[r]], 3);
        walk_lengths(s, &s->lt, s->lt.max_code, 1);
        walk_lengths(s, &s->dt, s->dt.max_code, 1);
        emit_symbols(s, s->lt.code, s->lt.len, s->dt.code, s->dt.len);
    }
    if (last) byte_align(s);                        /* bi_windup, trees.c:996-1001 */
    trace_block(s, stored_len, type, last, can_store, bit_start);
    reset_block(s);
    s->block_start = p;
}

/* _tr_tally(): trees.c:1010-1055 (deflate.h:323-339) */
static int tally(enc_t *s, uint32_t pos, unsigned dist, unsigned lc)
{
    s->p_buf[s->last_lit] = pos;
    s->d_buf[s->last_lit] = (uint16_t)dist; s->l_buf[s->last_lit++] = (uint8_t)lc;
    if (dist == 0) s->lt.freq[lc]++;
    else { s->matches++; s->lt.freq[len_code[lc] + 257]++; s->dt.freq[d_code(dist - 1)]++; }
    return s->last_lit == s->lit_bufsize - 1;
}

/* ------------------------------------------------------------------ window model */
static inline uint32_t hash3(const enc_t *s, uint32_t p)   /* UPDATE_HASH x3, deflate.c:167 */
{
    const uint8_t *b = s->buf + p;
    return ((((uint32_t)b[0] << s->hshift) ^ b[1]) << s->hshift ^ b[2]) & s->hmask;
}
/* INSERT_STRING(), deflate.c:186-189: returns the previous head entry */
static inline uint32_t insert_pos(enc_t *s, uint32_t p)
{
    uint32_t h = hash3(s, p), e = s->head[h];
    s->prev[p & s->wmask] = e; s->head[h] = p + 1;
    return e;
}

/* fill_window() for a call that supplies the whole chunk at once
 * (deflate.c:1390-1532): optional slide, top-up, deferred dictionary inserts. */
static void refill(enc_t *s, uint32_t p)
{
    if (p - s->base >= s->wsize + s->max_dist) s->base += s->wsize;
    uint64_t cap = (uint64_t)s->base + 2ull * s->wsize;
    s->wend = s->end < cap ? s->end : (uint32_t)cap;
    while (s->pending_ins) {
        uint32_t q = p - s->pending_ins;
        if (q + MINM > s->wend) break;
        insert_pos(s, q); s->pending_ins--;
    }
}

/* An entry read from head[]/prev[] names a usable candidate only if it is
 * non-empty and lies strictly above window index 0 (NIL == 0, deflate.c:104). */
static inline int live(const enc_t *s, uint32_t e) { return e != 0 && e - 1 > s->base; }

/* longest_match(): deflate.c:1148-1289 */
static uint32_t longest(enc_t *s, uint32_t p, uint32_t cur, uint32_t prev_len, uint32_t *mstart)
{
    uint32_t chain = s->chain, look = s->wend - p, best = prev_len;
    uint32_t nice = s->nice < look ? s->nice : look;
    uint32_t cap = look < MAXM ? look : MAXM;
    uint32_t limit = (p - s->base > s->max_dist) ? p - s->max_dist : s->base;
    const uint8_t *scan = s->buf + p;
    if (prev_len >= s->good) chain >>= 2;
    do {
        const uint8_t *m = s->buf + cur;
        if (best < cap && m[best] == scan[best]) {   /* cannot beat `best` otherwise */
            uint32_t len = 0;
            while (len < cap && m[len] == scan[len]) len++;
            if (len > best) { *mstart = cur; best = len; if (len >= nice) break; }
        }
        uint32_t e = s->prev[cur & s->wmask];
        if (e == 0) break;
        cur = e - 1;
    } while (cur > limit && --chain != 0);
    return best <= look ? best : look;
}

/* deflate_fast(): deflate.c:1628-1722 */
static uint32_t parse_fast(enc_t *s)
{
    uint32_t p = s->dict_len, mlen = MINM - 1, mstart = 0;
    for (;;) {
        if (s->wend - p < MIN_LOOK) { refill(s, p); if (s->wend == p) break; }
        uint32_t hh = 0; int bflush;
        if (s->wend - p >= MINM) hh = insert_pos(s, p);
        if (live(s, hh) && p - (hh - 1) <= s->max_dist) mlen = longest(s, p, hh - 1, MINM - 1, &mstart);
        if (mlen >= MINM) {
            bflush = tally(s, p, p - mstart, mlen - MINM);
            uint32_t look = s->wend - p - mlen;
            if (mlen <= s->lazy /* max_insert_length */ && look >= MINM) {
                mlen--; do { p++; insert_pos(s, p); } while (--mlen != 0);
                p++;
            } else { p += mlen; mlen = 0; }
        } else {
            bflush = tally(s, p, 0, s->buf[p]); p++;
        }
        if (bflush) flush_block(s, p, 0);
    }
    return p;
}

/* deflate_slow(): deflate.c:1730-1853 */
static uint32_t parse_slow(enc_t *s)
{
    uint32_t p = s->dict_len, mlen = MINM - 1, mstart = 0, avail = 0;
    for (;;) {
        if (s->wend - p < MIN_LOOK) { refill(s, p); if (s->wend == p) break; }
        uint32_t hh = 0;
        if (s->wend - p >= MINM) hh = insert_pos(s, p);
        uint32_t plen = mlen, pmatch = mstart;
        mlen = MINM - 1;
        if (live(s, hh) && plen < s->lazy && p - (hh - 1) <= s->max_dist) {
            mlen = longest(s, p, hh - 1, plen, &mstart);
            if (mlen <= 5 && (s->strategy == ZO_FILTERED || (mlen == MINM && p - mstart > TOO_FAR)))
                mlen = MINM - 1;
        }
        if (plen >= MINM && mlen <= plen) {
            uint32_t max_insert = s->wend - MINM;
            int bflush = tally(s, p - 1, p - 1 - pmatch, plen - MINM);
            uint32_t n = plen - 2;
            do { if (++p <= max_insert) insert_pos(s, p); } while (--n != 0);
            avail = 0; mlen = MINM - 1; p++;
            if (bflush) flush_block(s, p, 0);
        } else if (avail) {
            if (tally(s, p - 1, 0, s->buf[p - 1])) flush_block(s, p, 0);
            p++;
        } else { avail = 1; p++; }
    }
    if (avail) tally(s, p - 1, 0, s->buf[p - 1]);
    return p;
}

/* deflate_rle(): deflate.c:1861-1928 */
static uint32_t parse_rle(enc_t *s)
{
    uint32_t p = s->dict_len;
    for (;;) {
        if (s->wend - p <= MAXM) { refill(s, p); if (s->wend == p) break; }
        uint32_t look = s->wend - p, mlen = 0; int bflush;
        if (look >= MINM && p > s->base) {
            uint8_t c = s->buf[p - 1];
            uint32_t cap = look < MAXM ? look : MAXM;
            while (mlen < cap && s->buf[p + mlen] == c) mlen++;
        }
        if (mlen >= MINM) { bflush = tally(s, p, 1, mlen - MINM); p += mlen; }
        else { bflush = tally(s, p, 0, s->buf[p]); p++; }
        if (bflush) flush_block(s, p, 0);
    }
    return p;
}

/* deflate_huff(): deflate.c:1934-1967 */
static uint32_t parse_huff(enc_t *s)
{
    uint32_t p = s->dict_len;
    for (;;) {
        if (s->wend == p) { refill(s, p); if (s->wend == p) break; }
        int bflush = tally(s, p, 0, s->buf[p]); p++;
        if (bflush) flush_block(s, p, 0);
    }
    return p;
}

/* deflate_stored(): deflate.c:1564-1619 (level 0).  fill_window() tops the
 * window up to its full 2 * w_size whenever the lookahead is used up; blocks end
 * at min(0xffff, pending_buf_size - 5) bytes or as soon as they span MAX_DIST. */
static uint32_t parse_stored(enc_t *s)
{
    uint32_t max_block = 0xffff, p = s->dict_len, look = 0;
    if (max_block > s->lit_bufsize * 4u - 5u) max_block = s->lit_bufsize * 4u - 5u;
    for (;;) {
        if (look <= 1) {
            refill(s, p);
            look = s->wend - p;
            if (look == 0) break;
        }
        p += look; look = 0;
        uint32_t max_start = s->block_start + max_block;
        if (p >= max_start) { look = p - max_start; p = max_start; flush_block(s, p, 0); }
        if (p - s->block_start >= s->max_dist) flush_block(s, p, 0);
    }
    return p;
}

void zo_trace_free(zo_trace *t)
{
    if (!t) return;
    free(t->syms); free(t->blocks); memset(t, 0, sizeof *t);
}

size_t zo_deflate_chunk(const uint8_t *buf, size_t dict_len, size_t n,
                        int level, int strategy, int mem_level, int wbits,
                        int final, uint8_t *out, size_t out_cap, zo_trace *trace)
{
    return zo_deflate_chunk_tuned(buf, dict_len, n, level, strategy, mem_level, wbits, final, NULL, out, out_cap, trace);
}

/* the same after deflateTune(good_length, max_lazy, nice_length, max_chain) (deflate.c:531-547); tune == NULL: table values */
size_t zo_deflate_chunk_tuned(const uint8_t *buf, size_t dict_len, size_t n,
                              int level, int strategy, int mem_level, int wbits,
                              int final, const int *tune, uint8_t *out, size_t out_cap, zo_trace *trace)
{
    return zo_deflate_chunk_primed(buf, dict_len, n, level, strategy, mem_level, wbits, final, tune, 0, 0, out, out_cap, trace);
}

/* the same with prime_bits (0..7) bits of prime_val already waiting in the bit buffer, as deflatePrime
 * leaves them (deflate.c:464-487): the first block starts behind them */
size_t zo_deflate_chunk_primed(const uint8_t *buf, size_t dict_len, size_t n,
                               int level, int strategy, int mem_level, int wbits,
                               int final, const int *tune, int prime_bits, int prime_val,
                               uint8_t *out, size_t out_cap, zo_trace *trace)
{
    return zo_deflate_chunk_ex(buf, dict_len, n, level, strategy, mem_level, wbits, final ? ZO_FLUSH_FINISH : ZO_FLUSH_SYNC,
                               tune, prime_bits, prime_val, out, out_cap, trace, NULL, NULL, NULL);
}

/* the general form: `flush` is what the one deflate() call is given (deflate.c:923-945); with Z_PARTIAL_FLUSH and
 * Z_BLOCK the stream can end inside a byte: the return value counts the whole bytes (what deflate() hands out) and
 * pend_bits / pend_val the bits left in the bit buffer (what deflatePending() reports, deflate.c:450-461) */
size_t zo_deflate_chunk_ex(const uint8_t *buf, size_t dict_len, size_t n,
                           int level, int strategy, int mem_level, int wbits,
                           int flush, const int *tune, int prime_bits, int prime_val,
                           uint8_t *out, size_t out_cap, zo_trace *trace,
                           int *pend_bits, int *pend_val, int *data_type)
{
    const int final = flush == ZO_FLUSH_FINISH;
    if (flush < 0 || flush > 3) return (size_t)-1;
    if (prime_bits < 0 || prime_bits > 7) return (size_t)-1;
    if (level == -1) level = 6;                                    /* deflate.c:258 */
    if (level < 0 || level > 9 || strategy < 0 || strategy > 4 || mem_level < 1 || mem_level > 9 ||
        wbits < 9 || wbits > 15 || dict_len > (1u << wbits) || dict_len + n > 0xfffffff0u)
        return (size_t)-1;
    build_tables();
    enc_t *s = (enc_t *)calloc(1, sizeof *s);
    s->buf = buf; s->dict_len = (uint32_t)dict_len; s->end = (uint32_t)(dict_len + n);
    s->level = level; s->strategy = strategy;
    s->wsize = 1u << wbits; s->wmask = s->wsize - 1; s->max_dist = s->wsize - MIN_LOOK;
    s->hbits = (uint32_t)mem_level + 7; s->hmask = (1u << s->hbits) - 1;       /* deflate.c:284-291 */
    s->hshift = (s->hbits + MINM - 1) / MINM;
    s->lit_bufsize = 1u << (mem_level + 6);                                    /* deflate.c:299 */
    s->good = cfg[level].good; s->lazy = cfg[level].lazy; s->nice = cfg[level].nice; s->chain = cfg[level].chain;
    if (tune) { s->good = (uint32_t)tune[0]; s->lazy = (uint32_t)tune[1]; s->nice = (uint32_t)tune[2]; s->chain = (uint32_t)tune[3]; }
    s->head = (uint32_t *)calloc((size_t)1 << s->hbits, sizeof(uint32_t));
    s->prev = (uint32_t *)calloc(s->wsize, sizeof(uint32_t));
    s->d_buf = (uint16_t *)malloc(s->lit_bufsize * sizeof(uint16_t));
    s->l_buf = (uint8_t *)malloc(s->lit_bufsize);
    s->p_buf = (uint32_t *)malloc(s->lit_bufsize * sizeof(uint32_t));
    s->out = out; s->out_cap = out_cap; s->tr = trace; s->data_type = 2;
    if (trace) s->tr_sym_base = trace->n_syms;
    reset_block(s);
    if (prime_bits) put_bits(s, (unsigned)prime_val & ((1u << prime_bits) - 1u), prime_bits);

    /* deflateSetDictionary(): deflate.c:323-389.  All dictionary strings with
     * three bytes available are inserted; the last (up to) two are deferred. */
    s->base = 0; s->wend = s->dict_len; s->block_start = s->dict_len;
    if (s->dict_len >= MINM)
        for (uint32_t q = 0; q + MINM <= s->dict_len; q++) insert_pos(s, q);
    s->pending_ins = s->dict_len < MINM - 1 ? s->dict_len : MINM - 1;
    if (s->dict_len == 0) s->pending_ins = 0;

    uint32_t p;
    if (strategy == ZO_HUFFMAN_ONLY) p = parse_huff(s);                       /* deflate.c:903-905 */
    else if (strategy == ZO_RLE) p = parse_rle(s);
    else if (cfg[level].parser == P_STORED) p = parse_stored(s);
    else if (cfg[level].parser == P_FAST) p = parse_fast(s);
    else p = parse_slow(s);

    if (final) flush_block(s, p, 1);
    else {
        if (s->last_lit || (s->level == 0 && p > s->block_start)) flush_block(s, p, 0);   /* deflate.c:1616, 1720, 1850 */
        if (flush == ZO_FLUSH_SYNC) {
            /* Z_SYNC_FLUSH: empty stored block, deflate.c:926-927, trees.c:865-877 */
            put_bits(s, 0, 3); byte_align(s); put_bits(s, 0x0000, 16); put_bits(s, 0xffff, 16);
        } else if (flush == ZO_FLUSH_PARTIAL) {
            /* Z_PARTIAL_FLUSH: _tr_align, trees.c:892-901: an empty static block, then bi_flush (whole bytes only) */
            put_bits(s, 1 << 1, 3); put_bits(s, st_lcode[END_BLOCK], st_llen[END_BLOCK]);
        }                                           /* Z_BLOCK: nothing more, deflate.c:923-925 */
    }
    if (pend_bits) *pend_bits = s->bitcnt;
    if (pend_val) *pend_val = (int)(s->bitbuf & ((1u << s->bitcnt) - 1u));
    if (data_type) *data_type = s->data_type;
    size_t res = s->overflow ? (size_t)-1 : s->out_n;
    free(s->head); free(s->prev); free(s->d_buf); free(s->l_buf); free(s->p_buf); free(s);
    return res;
}

size_t zo_deflate_chunked(const uint8_t *in, size_t n, size_t chunk_size,
                          int level, int strategy, int mem_level,
                          uint8_t *out, size_t out_cap,
                          uint32_t *chunk_len, size_t max_chunks)
{
    return zo_deflate_chunked_w(in, n, chunk_size, level, strategy, mem_level, 15, out, out_cap, chunk_len, max_chunks);
}

size_t zo_deflate_chunked_w(const uint8_t *in, size_t n, size_t chunk_size,
                            int level, int strategy, int mem_level, int wbits,
                            uint8_t *out, size_t out_cap,
                            uint32_t *chunk_len, size_t max_chunks)
{
    size_t off = 0, o = 0, c = 0;
    if (wbits < 9 || wbits > 15) return (size_t)-1;
    const size_t wsize = (size_t)1 << wbits;
    do {
        size_t len = n - off < chunk_size ? n - off : chunk_size;
        size_t dict = off < wsize ? off : wsize;
        int final = off + len >= n;
        size_t r = zo_deflate_chunk(in + off - dict, dict, len, level, strategy, mem_level, wbits,
                                    final, out + o, out_cap - o, NULL);
        if (r == (size_t)-1 || c >= max_chunks) return (size_t)-1;
        chunk_len[c++] = (uint32_t)r; o += r; off += len;
    } while (off < n);
    return o;
}

/* ------------------------------------------------------------------ checksums */
#define ADLER_BASE 65521u
#define ADLER_NMAX 5552

uint32_t zo_adler32(uint32_t adler, const uint8_t *buf, size_t len)   /* adler32.c:65-133 */
{
    if (!buf) return 1;
    uint32_t a = adler & 0xffff, b = adler >> 16;
    while (len) {
        size_t k = len < ADLER_NMAX ? len : ADLER_NMAX; len -= k;
        while (k--) { a += *buf++; b += a; }
        a %= ADLER_BASE; b %= ADLER_BASE;
    }
    return a | (b << 16);
}

uint32_t zo_adler32_combine(uint32_t a1, uint32_t a2, int64_t len2)    /* adler32.c:136-162 */
{
    if (len2 < 0) return 0xffffffffu;
    uint32_t rem = (uint32_t)(len2 % ADLER_BASE);
    uint32_t s1 = a1 & 0xffff, s2 = (rem * s1) % ADLER_BASE;
    s1 += (a2 & 0xffff) + ADLER_BASE - 1;
    s2 += (a1 >> 16) + (a2 >> 16) + ADLER_BASE - rem;
    if (s1 >= ADLER_BASE) s1 -= ADLER_BASE;
    if (s1 >= ADLER_BASE) s1 -= ADLER_BASE;
    if (s2 >= (ADLER_BASE << 1)) s2 -= (ADLER_BASE << 1);
    if (s2 >= ADLER_BASE) s2 -= ADLER_BASE;
    return s1 | (s2 << 16);
}

static uint32_t crc_tab[256]; static int crc_ready;
static void crc_init(void)                                             /* crc32.c:90-138 */
{
    if (crc_ready) return;
    for (uint32_t n = 0; n < 256; n++) {
        uint32_t c = n;
        for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
        crc_tab[n] = c;
    }
    crc_ready = 1;
}

uint32_t zo_crc32(uint32_t crc, const uint8_t *buf, size_t len)        /* crc32.c:204-234 */
{
    if (!buf) return 0;
    crc_init();
    uint32_t c = crc ^ 0xffffffffu;
    while (len--) c = crc_tab[(c ^ *buf++) & 0xff] ^ (c >> 8);
    return c ^ 0xffffffffu;
}

/* crc32_combine_(): crc32.c:327-408 — GF(2) operator for len2 zero bytes */
static uint32_t gf2_times(const uint32_t *mat, uint32_t vec)
{
    uint32_t sum = 0;
    for (; vec; vec >>= 1, mat++) if (vec & 1) sum ^= *mat;
    return sum;
}
static void gf2_square(uint32_t *sq, const uint32_t *mat)
{
    for (int n = 0; n < 32; n++) sq[n] = gf2_times(mat, mat[n]);
}
uint32_t zo_crc32_combine(uint32_t c1, uint32_t c2, int64_t len2)
{
    uint32_t even[32], odd[32], row = 1;
    if (len2 <= 0) return c1;
    odd[0] = 0xedb88320u;
    for (int n = 1; n < 32; n++) { odd[n] = row; row <<= 1; }
    gf2_square(even, odd); gf2_square(odd, even);
    do {
        gf2_square(even, odd);
        if (len2 & 1) c1 = gf2_times(even, c1);
        len2 >>= 1;
        if (len2 == 0) break;
        gf2_square(odd, even);
        if (len2 & 1) c1 = gf2_times(odd, c1);
        len2 >>= 1;
    } while (len2 != 0);
    return c1 ^ c2;
}

/* ------------------------------------------------------------------ corpus
 * Fixture-data definition (xorshift64 word salad / noise / low entropy /
 * image-like deltas / 64 KiB mixed segments). */
typedef struct { uint64_t s; } rng_t;
static uint32_t r32(rng_t *g) { uint64_t s = g->s; s ^= s << 13; s ^= s >> 7; s ^= s << 17; g->s = s; return (uint32_t)(s >> 32); }
static const char *const voc[32] = {"the","of","and","to","in","is","that","for","it","as","was","with","be","by","on","not",
    "he","this","are","or","his","from","at","which","but","have","an","had","they","you","were","compression"};
static void fill_kind(uint8_t *buf, size_t n, int kind, rng_t *g, size_t *phase)
{
    size_t p = 0;
    switch (kind) {
    case 0: { unsigned w = 0;
        while (p < n) { const char *s = voc[r32(g) & 31]; size_t l = strlen(s); if (l > n - p) l = n - p;
            memcpy(buf + p, s, l); p += l; if (p < n) buf[p++] = (++w % 12 == 0) ? '\n' : ' '; }
        break; }
    case 1: while (p < n) buf[p++] = (uint8_t)(r32(g) & 0xff); break;
    case 2: while (p < n) { uint32_t a = r32(g), b = r32(g), c = r32(g); buf[p++] = (uint8_t)(a & b & c & 0xff); } break;
    case 3: while (p < n) { size_t i = *phase; unsigned x = (unsigned)((i / 3) % 1024), y = (unsigned)(i / 3 / 1024), c = (unsigned)(i % 3);
            unsigned cur = (x * (c + 1) + y * (3 - c) + (r32(g) & 3)) & 0xff;
            unsigned prv = x ? (((x - 1) * (c + 1) + y * (3 - c)) & 0xff) : 0;
            buf[p++] = (uint8_t)((cur - prv) & 0xff); (*phase)++; }
        break;
    }
}
void zo_corpus_fill(uint8_t *buf, size_t n, int kind, uint64_t seed)
{
    rng_t g; g.s = seed; size_t phase = 0;
    if (kind != 4) { fill_kind(buf, n, kind, &g, &phase); return; }
    size_t p = 0; unsigned k = 0;
    while (p < n) { size_t l = n - p < 65536 ? n - p : 65536; unsigned m = k++ % 5;
        if (m == 4) memset(buf + p, 0, l); else fill_kind(buf + p, l, (int)m, &g, &phase); p += l; }
}

/* ------------------------------------------------------------------ raw inflate
 * Restates the block layer of inflate() (inflate.c:826-1173), the table rules of
 * inflate_table() (inftrees.c:107-138: over-subscribed and incomplete sets) and
 * the copy semantics of inflate_fast() (inffast.c:185-281) with a canonical
 * bit-serial decoder.  Error texts are the reference's. */
typedef struct {
    const uint8_t *in; size_t in_len, pos; uint64_t hold; int bits; int starved;
} bitrd_t;

static int need(bitrd_t *b, int n)
{
    while (b->bits < n) {
        if (b->pos >= b->in_len) { b->starved = 1; return 0; }
        b->hold |= (uint64_t)b->in[b->pos++] << b->bits; b->bits += 8;
    }
    return 1;
}
static unsigned take(bitrd_t *b, int n) { unsigned v = (unsigned)(b->hold & ((1ull << n) - 1)); b->hold >>= n; b->bits -= n; return v; }

typedef struct { uint16_t count[16], symbol[288]; int max; } huff_t;

/* returns 0 ok, -1 over-subscribed or incomplete (inftrees.c:131-138) */
static int huff_build(huff_t *h, const uint8_t *lens, int n, int is_codes)
{
    int len, left = 1; uint16_t offs[16];
    memset(h->count, 0, sizeof h->count);
    for (int s = 0; s < n; s++) h->count[lens[s]]++;
    for (h->max = 15; h->max >= 1; h->max--) if (h->count[h->max]) break;
    if (h->max == 0) return 0;                       /* no codes: every code is invalid */
    for (len = 1; len <= 15; len++) { left <<= 1; left -= h->count[len]; if (left < 0) return -1; }
    offs[1] = 0;
    for (len = 1; len < 15; len++) offs[len + 1] = (uint16_t)(offs[len] + h->count[len]);
    for (int s = 0; s < n; s++) if (lens[s]) h->symbol[offs[lens[s]]++] = (uint16_t)s;
    if (left > 0 && (is_codes || h->max != 1)) return -1;
    return 0;
}

/* returns symbol, -1 invalid code, -2 out of input */
static int huff_decode(bitrd_t *b, const huff_t *h)
{
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
        if (!need(b, 1)) return -2;
        code |= (int)take(b, 1);
        int count = h->count[len];
        if (code - count < first) return h->symbol[index + (code - first)];
        index += count; first += count; first <<= 1; code <<= 1;
        if (len >= h->max) break;
    }
    return -1;
}

static int inflate_raw_from(const uint8_t *in, size_t in_len, uint64_t start_bit, const uint8_t *dict, size_t dict_len,
                            uint8_t *out, size_t out_cap, size_t *consumed, size_t *produced,
                            const char **msg, int stop_at_sync);
int zo_inflate_raw(const uint8_t *in, size_t in_len, const uint8_t *dict, size_t dict_len,
                   uint8_t *out, size_t out_cap, size_t *consumed, size_t *produced,
                   const char **msg, int stop_at_sync)
{
    return inflate_raw_from(in, in_len, 0, dict, dict_len, out, out_cap, consumed, produced, msg, stop_at_sync);
}
/* The same from a block that starts at bit start_bit of the input (what inflatePrime with the first byte's upper bits
 * sets up, inflate.c:232-253): stored blocks still align to the input's byte boundaries. */
int zo_inflate_raw_at(const uint8_t *in, size_t in_len, uint64_t start_bit, const uint8_t *dict, size_t dict_len,
                      uint8_t *out, size_t out_cap, size_t *consumed, size_t *produced,
                      const char **msg, int stop_at_sync)
{
    return inflate_raw_from(in, in_len, start_bit, dict, dict_len, out, out_cap, consumed, produced, msg, stop_at_sync);
}
static int inflate_raw_from(const uint8_t *in, size_t in_len, uint64_t start_bit, const uint8_t *dict, size_t dict_len,
                            uint8_t *out, size_t out_cap, size_t *consumed, size_t *produced,
                            const char **msg, int stop_at_sync)
{
    static const uint16_t lbase[29] = {3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258};
    static const uint16_t dbase[30] = {1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577};
    build_tables();
    bitrd_t b = {in, in_len, (size_t)(start_bit >> 3), 0, 0, 0};
    size_t o = 0; int rc = 0, last = 0; const char *m = NULL;
    if (start_bit >> 3 > in_len) { if (msg) *msg = NULL; return -5; }
    if (start_bit & 7) { if (!need(&b, (int)(start_bit & 7))) { if (msg) *msg = NULL; return -5; } take(&b, (int)(start_bit & 7)); }
    huff_t *lh = (huff_t *)malloc(sizeof(huff_t)), *dh = (huff_t *)malloc(sizeof(huff_t)), *ch = (huff_t *)malloc(sizeof(huff_t));
    uint8_t lens[320];
#define FAIL(t) do { m = (t); rc = -3; goto done; } while (0)
#define STARVE() do { rc = -5; goto done; } while (0)
    while (!last) {
        if (!need(&b, 3)) STARVE();
        last = (int)take(&b, 1);
        unsigned type = take(&b, 2);
        if (type == 3) FAIL("invalid block type");
        if (type == 0) {
            take(&b, b.bits & 7);
            if (!need(&b, 32)) STARVE();
            unsigned len = take(&b, 16), nlen = take(&b, 16);
            if (len != (nlen ^ 0xffff)) FAIL("invalid stored block lengths");
            /* bytes still in the bit buffer are whole bytes of the stream */
            while (len) {
                if (!need(&b, 8)) STARVE();
                if (o >= out_cap) STARVE();
                out[o++] = (uint8_t)take(&b, 8); len--;
            }
            if (stop_at_sync && nlen == 0xffff && !last) break;
            continue;
        }
        if (type == 1) {
            for (int s = 0; s < 288; s++) lens[s] = st_llen[s];
            huff_build(lh, lens, 288, 0);
            for (int s = 0; s < 30; s++) lens[s] = 5;
            huff_build(dh, lens, 30, 0);
        } else {
            if (!need(&b, 14)) STARVE();
            unsigned nlen = take(&b, 5) + 257, ndist = take(&b, 5) + 1, ncode = take(&b, 4) + 4;
            if (nlen > 286 || ndist > 30) FAIL("too many length or distance symbols");
            memset(lens, 0, 19);
            for (unsigned i = 0; i < ncode; i++) { if (!need(&b, 3)) STARVE(); lens[bl_order[i]] = (uint8_t)take(&b, 3); }
            if (huff_build(ch, lens, 19, 1)) FAIL("invalid code lengths set");
            unsigned have = 0;
            while (have < nlen + ndist) {
                int sym = huff_decode(&b, ch);
                if (sym == -2) STARVE();
                if (sym < 0) FAIL("invalid code lengths set");   /* unreachable for complete sets */
                if (sym < 16) { lens[have++] = (uint8_t)sym; continue; }
                unsigned len = 0, copy;
                if (sym == 16) {
                    if (!need(&b, 2)) STARVE();
                    if (have == 0) FAIL("invalid bit length repeat");
                    len = lens[have - 1]; copy = 3 + take(&b, 2);
                } else if (sym == 17) { if (!need(&b, 3)) STARVE(); copy = 3 + take(&b, 3); }
                else { if (!need(&b, 7)) STARVE(); copy = 11 + take(&b, 7); }
                if (have + copy > nlen + ndist) FAIL("invalid bit length repeat");
                while (copy--) lens[have++] = (uint8_t)len;
            }
            if (lens[256] == 0) FAIL("invalid code -- missing end-of-block");
            if (huff_build(lh, lens, (int)nlen, 0)) FAIL("invalid literal/lengths set");
            if (huff_build(dh, lens + nlen, (int)ndist, 0)) FAIL("invalid distances set");
        }
        for (;;) {
            int sym = huff_decode(&b, lh);
            if (sym == -2) STARVE();
            if (sym < 0) FAIL("invalid literal/length code");
            if (sym < 256) { if (o >= out_cap) STARVE(); out[o++] = (uint8_t)sym; continue; }
            if (sym == 256) break;
            sym -= 257;
            if (sym >= 29) FAIL("invalid literal/length code");
            if (!need(&b, x_lbits[sym])) STARVE();
            unsigned len = lbase[sym] + take(&b, x_lbits[sym]);
            int ds = huff_decode(&b, dh);
            if (ds == -2) STARVE();
            if (ds < 0 || ds >= 30) FAIL("invalid distance code");
            if (!need(&b, x_dbits[ds])) STARVE();
            size_t dist = dbase[ds] + take(&b, x_dbits[ds]);
            if (dist > o + dict_len) FAIL("invalid distance too far back");
            if (o + len > out_cap) STARVE();
            for (unsigned k = 0; k < len; k++, o++)
                out[o] = dist > o ? dict[dict_len - (dist - o)] : out[o - dist];
        }
    }
done:
    /* whole unread bytes go back to the caller, as inflate_fast does (inffast.c:309-313) */
    { size_t back = (size_t)(b.bits >> 3); b.pos -= back; }
    if (consumed) *consumed = b.pos;
    if (produced) *produced = o;
    if (msg) *msg = m;
    free(lh); free(dh); free(ch);
    return rc;
#undef FAIL
#undef STARVE
}
