/*
 * fuzz_vs_ref.c — pins the oracle (zoracle.c) against the unmodified reference
 * build (oracle/_ref/libzref.so).  TEST INFRASTRUCTURE ONLY.
 *
 * For random (data, level, strategy, memLevel, dictionary, flush) tuples it runs
 * the reference recipe  deflateInit2 / deflateSetDictionary / one deflate() call
 * and compares the bytes with zo_deflate_chunk().  Also cross-checks checksums
 * and their combine functions, and inflates every stream with zo_inflate_raw.
 *
 * usage: fuzz_vs_ref [iterations] [seed] [max_len]
 */
#include "zoracle.h"
#include <stdio.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>
#include <stdlib.h>
#include <string.h>
#include "zlib.h"   /* the reference's own header, found via -I/root/reference */

static uint64_t rs = 88172645463325252ull;
static uint32_t rnd(void) { rs ^= rs << 13; rs ^= rs >> 7; rs ^= rs << 17; return (uint32_t)(rs >> 32); }

static void gen(uint8_t *b, size_t n)
{
    int mode = rnd() % 9;
    if (mode < 5) { zo_corpus_fill(b, n, mode, rnd() | 1); return; }
    if (mode == 5) { unsigned a = 1 + rnd() % 16; for (size_t i = 0; i < n; i++) b[i] = (uint8_t)('a' + rnd() % a); return; }
    if (mode == 6) { unsigned per = 1 + rnd() % 300; for (size_t i = 0; i < n; i++) b[i] = (uint8_t)(i < per ? rnd() : b[i - per]);
                     for (size_t i = 0; i < n / 97; i++) { b[rnd() % (n ? n : 1)] ^= 1; } return; }
    if (mode == 7) { memset(b, rnd() & 0xff, n); for (size_t i = 0; i < n / 1000; i++) b[rnd() % (n ? n : 1)] = (uint8_t)rnd(); return; }
    /* mode 8: copies of earlier material at assorted distances, incl. near MAX_DIST */
    size_t i = 0;
    while (i < n) {
        if (i > 8 && rnd() % 3) {
            size_t d = (rnd() % 4 == 0) ? 32400 + rnd() % 400 : 1 + rnd() % 40000; if (d > i) d = 1 + rnd() % i;
            size_t l = 3 + rnd() % 300; for (size_t k = 0; k < l && i < n; k++, i++) b[i] = b[i - d];
        } else { size_t l = 1 + rnd() % 20; for (size_t k = 0; k < l && i < n; k++, i++) b[i] = (uint8_t)('a' + rnd() % 6); }
    }
}

static const int *g_tune;   /* deflateTune values for the next reference call, or NULL */
static int g_prime_bits, g_prime_val;   /* deflatePrime in front of the next reference call */
static int g_wbits = 15;                /* windowBits of the next reference call */
static int g_flush = -1;                /* flush value of the next reference call when not final: -1 = Z_SYNC_FLUSH */
static int g_ref_pend_bits, g_ref_pend_val, g_ref_data_type;   /* results of the last reference call */
static size_t ref_chunk(const uint8_t *buf, size_t dict, size_t n, int level, int strat, int mem, int final,
                        uint8_t *out, size_t cap)
{
    z_stream z; memset(&z, 0, sizeof z);
    if (deflateInit2(&z, level, Z_DEFLATED, -g_wbits, mem, strat) != Z_OK) return (size_t)-1;
    if (g_tune) deflateTune(&z, g_tune[0], g_tune[1], g_tune[2], g_tune[3]);
    if (g_prime_bits) deflatePrime(&z, g_prime_bits, g_prime_val);
    if (dict && deflateSetDictionary(&z, buf, (uInt)dict) != Z_OK) { deflateEnd(&z); return (size_t)-1; }
    z.next_in = (Bytef *)(buf + dict); z.avail_in = (uInt)n; z.next_out = out; z.avail_out = (uInt)cap;
    int r = deflate(&z, final ? Z_FINISH : g_flush >= 0 ? g_flush : Z_SYNC_FLUSH);
    size_t res = (r == (final ? Z_STREAM_END : Z_OK) && z.avail_in == 0) ? z.total_out : (size_t)-1;
    unsigned pend = 0; int bits = 0;
    deflatePending(&z, &pend, &bits);
    g_ref_pend_bits = bits; g_ref_pend_val = 0; g_ref_data_type = z.data_type;
    if (bits) {            /* the value of the bits left in bi_buf: flush them out with a second call and look at the next byte */
        Bytef extra[16]; z.next_out = extra; z.avail_out = sizeof extra;
        if (deflate(&z, Z_FINISH) == Z_STREAM_END) g_ref_pend_val = extra[0] & ((1 << bits) - 1);
    }
    deflateEnd(&z);
    return res;
}

/* zlib 1.2.8's overlay overrun (see below) can also corrupt its heap: combinations at risk run
 * in a child process whose death counts like corrupt output. */
static size_t ref_chunk_guarded(const uint8_t *buf, size_t dict, size_t n, int level, int strat, int mem, int final,
                                uint8_t *out, size_t cap, int *died)
{
    *died = 0;
    if (!(mem <= 3 && (strat == 4 || level <= 3))) return ref_chunk(buf, dict, n, level, strat, mem, final, out, cap);
    size_t *sh = mmap(NULL, cap + 64, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (sh == MAP_FAILED) return ref_chunk(buf, dict, n, level, strat, mem, final, out, cap);
    sh[0] = (size_t)-1;
    pid_t pid = fork();
    if (pid == 0) { sh[0] = ref_chunk(buf, dict, n, level, strat, mem, final, (uint8_t *)(sh + 8), cap); sh[1] = (size_t)g_ref_pend_bits; sh[2] = (size_t)g_ref_pend_val; sh[3] = (size_t)g_ref_data_type; _exit(0); }
    int st = 0; waitpid(pid, &st, 0);
    size_t r = (size_t)-1;
    if (WIFEXITED(st) && WEXITSTATUS(st) == 0) { r = sh[0]; if (r != (size_t)-1) memcpy(out, sh + 8, r); g_ref_pend_bits = (int)sh[1]; g_ref_pend_val = (int)sh[2]; g_ref_data_type = (int)sh[3]; }
    else *died = 1;
    munmap(sh, cap + 64);
    return r;
}

int main(int argc, char **argv)
{
    long iters = argc > 1 ? atol(argv[1]) : 300;
    if (argc > 2) rs = strtoull(argv[2], 0, 10) | 1;
    size_t max_len = argc > 3 ? (size_t)atol(argv[3]) : 400000;
    uint8_t *buf = malloc(32768 + max_len + 16), *o1 = malloc(2 * max_len + 65536), *o2 = malloc(2 * max_len + 65536);
    uint8_t *back = malloc(max_len + 16);
    long bad = 0, ref_corrupt = 0;
    static const size_t special[] = {0, 1, 2, 3, 4, 261, 262, 263, 16383, 16384, 32506, 32768 - 262 - 3, 32509, 32767, 32768, 32769,
                                     65274, 65277, 65374, 65535, 65536, 65537, 98000, 131072};
    for (long it = 0; it < iters; it++) {
        size_t n;
        unsigned r = rnd() % 10;
        if (r < 3) n = special[rnd() % (sizeof special / sizeof *special)];
        else if (r < 6) n = rnd() % 70000;
        else n = rnd() % (max_len + 1);
        if (n > max_len) n = max_len;
        /* windowBits: mostly 15, else 9..14 (8 is raised to 9 by deflateInit2_, deflate.c:276) */
        g_wbits = (rnd() % 4) ? 15 : 9 + (int)(rnd() % 6);
        const size_t wsz = (size_t)1 << g_wbits;
        size_t dict = (rnd() % 3) ? wsz : (rnd() % 4 ? 0 : rnd() % (wsz + 1));
        if (g_wbits < 15 && n > 200000) n = 200000;
        int level = rnd() % 10, strat = rnd() % 5, mem = (rnd() % 3) ? 8 : 1 + rnd() % 9, final = rnd() & 1;
        /* level 0 with Z_RLE / Z_HUFFMAN_ONLY stores whole symbol buffers (up to 16383 * 258 bytes) through the 64 KiB
         * pending buffer in zlib 1.2.8 (trees.c:1224 overruns): nothing to pin there */
        if (level == 0 && (strat == 2 || strat == 3)) strat = 0;
        if (rnd() % 2) strat = 0;
        gen(buf, dict + n);
        size_t cap = 2 * max_len + 65536;
        int tune[4]; g_tune = NULL;
        if (level > 0 && rnd() % 4 == 0) {        /* deflateTune: chain lengths, lazy limits and nice lengths off the table */
            static const int lazies[] = {0, 1, 2, 3, 4, 5, 8, 16, 64, 258, 300}, nices[] = {3, 4, 8, 9, 32, 100, 258, 259, 1000}, chains[] = {1, 2, 3, 5, 8, 17, 40, 64, 100, 500};
            tune[0] = 1 + (int)(rnd() % 40); tune[1] = lazies[rnd() % 11]; tune[2] = nices[rnd() % 9]; tune[3] = chains[rnd() % (level <= 3 ? 8 : 10)];
            g_tune = tune;
        }
        g_prime_bits = 0; g_prime_val = 0;
        if (rnd() % 6 == 0) { g_prime_bits = 1 + (int)(rnd() % 7); g_prime_val = (int)(rnd() & 0xffff); }
        /* the flush value of a chunk that is not the last: mostly Z_SYNC_FLUSH, else Z_PARTIAL_FLUSH, Z_BLOCK or Z_FULL_FLUSH */
        g_flush = -1;
        int oflush = final ? ZO_FLUSH_FINISH : ZO_FLUSH_SYNC;
        if (!final && rnd() % 3 == 0) {
            unsigned k = rnd() % 3;
            g_flush = k == 0 ? Z_PARTIAL_FLUSH : k == 1 ? Z_BLOCK : Z_FULL_FLUSH;
            oflush = k == 0 ? ZO_FLUSH_PARTIAL : k == 1 ? ZO_FLUSH_BLOCK : ZO_FLUSH_SYNC;
        }
        int died = 0;
        size_t a = ref_chunk_guarded(buf, dict, n, level, strat, mem, final, o1, cap, &died);
        if (died) { ref_corrupt++; fprintf(stderr, "reference crashed: it=%ld n=%zu L=%d S=%d M=%d\n", it, n, level, strat, mem); continue; }
        zo_trace tr; memset(&tr, 0, sizeof tr);
        int opb = 0, opv = 0, odt = 2;
        size_t b = zo_deflate_chunk_ex(buf, dict, n, level, strat, mem, g_wbits, oflush, g_tune, g_prime_bits, g_prime_val, o2, cap, &tr, &opb, &opv, &odt);
        int ok = a == b && a != (size_t)-1 && memcmp(o1, o2, a) == 0;
        if (ok && (opb != g_ref_pend_bits || opv != g_ref_pend_val)) { ok = 0; fprintf(stderr, "pending bits: ref %d (0x%x) oracle %d (0x%x)\n", g_ref_pend_bits, g_ref_pend_val, opb, opv); }
        if (ok && odt != g_ref_data_type) { ok = 0; fprintf(stderr, "data_type: ref %d oracle %d\n", g_ref_data_type, odt); }
        const int whole = opb == 0 && (oflush == ZO_FLUSH_SYNC || oflush == ZO_FLUSH_FINISH);
        /* trace must tile the chunk */
        if (ok && !(level == 0 && tr.n_syms == 0)) { uint32_t pos = (uint32_t)dict; for (size_t i = 0; i < tr.n_syms; i++) { if (tr.syms[i].pos != pos) { ok = 0; break; } pos += tr.syms[i].dist ? tr.syms[i].lc + 3u : 1u; } if (pos != dict + n) ok = 0; }
        if (ok && !g_prime_bits && whole) {   /* inflate with the oracle's decoder */
            size_t used = 0, made = 0; const char *msg = 0;
            int rc = zo_inflate_raw(o2, b, buf, dict, back, n, &used, &made, &msg, !final);
            if (rc != 0 || made != n || used != b || memcmp(back, buf + dict, n)) { ok = 0; fprintf(stderr, "inflate rc=%d made=%zu/%zu used=%zu/%zu msg=%s\n", rc, made, n, used, b, msg ? msg : "-"); }
        }
        if (!ok && a != (size_t)-1) {
            /* zlib 1.2.8 can overrun the pending_buf/d_buf overlay (deflate.h:217-245) when a block
             * codes to more bytes than the overlay leaves room for (Z_FIXED, tiny memLevel); its output
             * is then corrupt.  Such cases pin nothing: skip them if the reference cannot inflate itself. */
            z_stream iz; memset(&iz, 0, sizeof iz); inflateInit2(&iz, -15);
            if (!whole) { o1[a] = (uint8_t)g_ref_pend_val; a++; }      /* the decoder needs the bits the encoder still holds */
            if (dict) inflateSetDictionary(&iz, buf, (uInt)dict);
            iz.next_in = o1; iz.avail_in = (uInt)a; iz.next_out = back; iz.avail_out = (uInt)n;
            if (g_prime_bits && a) { inflatePrime(&iz, 8 - g_prime_bits, o1[0] >> g_prime_bits); iz.next_in++; iz.avail_in--; }
            int ir = inflate(&iz, Z_SYNC_FLUSH);
            int self_ok = (ir == Z_OK || ir == Z_STREAM_END || ir == Z_BUF_ERROR) && iz.total_out == n && memcmp(back, buf + dict, n) == 0;
            inflateEnd(&iz);
            if (!self_ok) { ref_corrupt++; ok = 1; fprintf(stderr, "reference output corrupt (does not inflate): it=%ld n=%zu L=%d S=%d M=%d\n", it, n, level, strat, mem); }
        }
        if (!ok) {
            bad++;
            fprintf(stderr, "MISMATCH it=%ld n=%zu dict=%zu L=%d S=%d M=%d W=%d final=%d flush=%d ref=%zu ora=%zu\n", it, n, dict, level, strat, mem, g_wbits, final, g_flush, a, b);
            { size_t d = 0; while (d < a && d < b && o1[d] == o2[d]) d++; fprintf(stderr, "  first diff at byte %zu\n", d);
              for (size_t i = 0; i < tr.n_blocks && i < 12; i++) fprintf(stderr, "  blk %zu: syms=%u in=[%u,+%u) type=%u can_store=%u opt=%llu static=%llu bits=[%llu,%llu)\n", i, tr.blocks[i].sym_count, tr.blocks[i].in_start, tr.blocks[i].in_len, tr.blocks[i].type, tr.blocks[i].can_store, (unsigned long long)tr.blocks[i].opt_len, (unsigned long long)tr.blocks[i].static_len, (unsigned long long)tr.blocks[i].bit_start, (unsigned long long)tr.blocks[i].bit_end); }
            if (bad > 10) break;
        }
        zo_trace_free(&tr);
        /* checksums */
        uint32_t seedv = rnd();
        size_t cut = n ? rnd() % (n + 1) : 0;
        if (adler32(seedv & 0xffffffff, buf, (uInt)n) != zo_adler32(seedv, buf, n) && (seedv & 0xffff) < 65521 && (seedv >> 16) < 65521) { bad++; fprintf(stderr, "adler mismatch\n"); }
        if (crc32(seedv, buf, (uInt)n) != zo_crc32(seedv, buf, n)) { bad++; fprintf(stderr, "crc mismatch\n"); }
        uint32_t a1 = zo_adler32(1, buf, cut), a2 = zo_adler32(1, buf + cut, n - cut);
        uint32_t c1 = zo_crc32(0, buf, cut), c2 = zo_crc32(0, buf + cut, n - cut);
        if (zo_adler32_combine(a1, a2, (int64_t)(n - cut)) != adler32_combine(a1, a2, (z_off_t)(n - cut)) ||
            zo_adler32_combine(a1, a2, (int64_t)(n - cut)) != zo_adler32(1, buf, n)) { bad++; fprintf(stderr, "adler combine mismatch\n"); }
        if (zo_crc32_combine(c1, c2, (int64_t)(n - cut)) != crc32_combine(c1, c2, (z_off_t)(n - cut)) ||
            zo_crc32_combine(c1, c2, (int64_t)(n - cut)) != zo_crc32(0, buf, n)) { bad++; fprintf(stderr, "crc combine mismatch\n"); }
    }
    printf("fuzz_vs_ref: %ld iterations, %ld mismatches, %ld skipped (reference corrupted its own output)\n", iters, bad, ref_corrupt);
    return bad != 0;
}
