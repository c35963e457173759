/*
 * gen_golden.c — writes tests/golden/deflate_golden.json from the UNMODIFIED
 * reference build (oracle/_ref/libzref.so).  TEST INFRASTRUCTURE ONLY.
 *
 * Every record is "reference output for stated input": the chunk recipe
 * (raw deflate, 1 MiB chunks, 32 KiB dictionary, Z_SYNC_FLUSH / Z_FINISH) on
 * the deterministic corpus generator, plus short known-answer vectors.
 * Only sizes, CRC-32s and short hex strings are stored.
 *
 * usage: gen_golden > ../tests/golden/deflate_golden.json
 */
#include "zoracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "zlib.h"   /* reference header, -I/root/reference */

static size_t ref_chunk(const uint8_t *buf, size_t dict, size_t n, int level, int strat, int mem, int final,
                        uint8_t *out, size_t cap)
{
    z_stream z; memset(&z, 0, sizeof z);
    if (deflateInit2(&z, level, Z_DEFLATED, -15, mem, strat) != Z_OK) exit(2);
    if (dict && deflateSetDictionary(&z, buf, (uInt)dict) != Z_OK) exit(3);
    z.next_in = (Bytef *)(buf + dict); z.avail_in = (uInt)n; z.next_out = out; z.avail_out = (uInt)cap;
    int r = deflate(&z, final ? Z_FINISH : Z_SYNC_FLUSH);
    if (r != (final ? Z_STREAM_END : Z_OK) || z.avail_in) exit(4);
    size_t res = z.total_out;
    deflateEnd(&z);
    return res;
}

static int first = 1;
static void chunked_row(int kind, uint64_t seed, size_t n, size_t chunk, int level, int strat, int mem)
{
    uint8_t *in = malloc(n + 1), *out = malloc(n + n / 4 + 65536);
    zo_corpus_fill(in, n, kind, seed);
    printf("%s  {\"kind\": %d, \"seed\": %llu, \"n\": %zu, \"chunk\": %zu, \"level\": %d, \"strategy\": %d, \"mem_level\": %d,\n",
           first ? "" : ",\n", kind, (unsigned long long)seed, n, chunk, level, strat, mem);
    first = 0;
    printf("   \"in_adler\": \"%08lx\", \"in_crc\": \"%08lx\", \"head\": \"", adler32(1, in, (uInt)n), crc32(0, in, (uInt)n));
    for (int i = 0; i < 8 && (size_t)i < n; i++) printf("%02x", in[i]);
    printf("\",\n   \"chunk_len\": [");
    size_t off = 0, o = 0; int c = 0;
    char crcs[65536]; size_t cp = 0; crcs[0] = 0;
    do {
        size_t len = n - off < chunk ? n - off : chunk, dict = off < 32768 ? off : 32768;
        size_t r = ref_chunk(in + off - dict, dict, len, level, strat, mem, off + len >= n, out + o, n + n / 4 + 65536 - o);
        printf("%s%zu", c ? ", " : "", r);
        cp += (size_t)snprintf(crcs + cp, sizeof crcs - cp, "%s\"%08lx\"", c ? ", " : "", crc32(0, out + o, (uInt)r));
        o += r; off += len; c++;
    } while (off < n);
    printf("],\n   \"chunk_crc\": [%s],\n   \"total\": %zu, \"out_crc\": \"%08lx\"}", crcs, o, crc32(0, out, (uInt)o));
    free(in); free(out);
}

static void kat(const char *name, const uint8_t *in, size_t n, int level, int strat)
{
    uint8_t out[65536];
    size_t r = ref_chunk(in, 0, n, level, strat, 8, 1, out, sizeof out);
    printf("%s  {\"name\": \"%s\", \"n\": %zu, \"level\": %d, \"strategy\": %d, \"in_crc\": \"%08lx\", \"out_len\": %zu, \"out_crc\": \"%08lx\", \"out_hex\": \"",
           first ? "" : ",\n", name, n, level, strat, crc32(0, in, (uInt)n), r, crc32(0, out, (uInt)r));
    first = 0;
    for (size_t i = 0; i < r && i < 256; i++) printf("%02x", out[i]);
    printf("\"}");
}

/* One chunk with a dictionary of any length, any windowBits and any flush value: the reference's bytes, the bits it
 * keeps in its bit buffer afterwards (deflatePending) and strm->data_type. */
static void boundary_row(int kind, uint64_t seed, size_t n, size_t dict, int level, int strat, int wbits, int flush)
{
    uint8_t *buf = malloc(dict + n + 1), *out = malloc(n + n / 4 + 65536);
    zo_corpus_fill(buf, dict, kind, seed + 1000);
    zo_corpus_fill(buf + dict, n, kind, seed);
    z_stream z; memset(&z, 0, sizeof z);
    if (deflateInit2(&z, level, Z_DEFLATED, -wbits, 8, strat) != Z_OK) exit(5);
    if (dict && deflateSetDictionary(&z, buf, (uInt)dict) != Z_OK) exit(6);
    z.next_in = buf + dict; z.avail_in = (uInt)n; z.next_out = out; z.avail_out = (uInt)(n + n / 4 + 65536);
    int r = deflate(&z, flush);
    if (r != (flush == Z_FINISH ? Z_STREAM_END : Z_OK) || z.avail_in) exit(7);
    size_t res = z.total_out; unsigned pend = 0; int bits = 0, val = 0, dt = z.data_type;
    deflatePending(&z, &pend, &bits);
    if (bits) { Bytef extra[16]; z.next_out = extra; z.avail_out = sizeof extra; if (deflate(&z, Z_FINISH) != Z_STREAM_END) exit(8); val = extra[0] & ((1 << bits) - 1); }
    deflateEnd(&z);
    printf("%s  {\"kind\": %d, \"seed\": %llu, \"n\": %zu, \"dict\": %zu, \"level\": %d, \"strategy\": %d, \"wbits\": %d, \"flush\": %d, "
           "\"out_len\": %zu, \"out_crc\": \"%08lx\", \"pend_bits\": %d, \"pend_val\": %d, \"data_type\": %d}",
           first ? "" : ",\n", kind, (unsigned long long)seed, n, dict, level, strat, wbits, flush, res, crc32(0, out, (uInt)res), bits, val, dt);
    first = 0;
    free(buf); free(out);
}

int main(void)
{
    const size_t N = 4u * 1048576u + 12345u, C = 1048576u;
    printf("{\"generator\": \"oracle/gen_golden.c against zlib %s built from the reference sources\",\n\"chunked\": [\n", zlibVersion());
    chunked_row(0, 1, N, C, 6, 0, 8); chunked_row(0, 1, N, C, 1, 0, 8); chunked_row(0, 1, N, C, 9, 0, 8);
    chunked_row(1, 2, N, C, 6, 0, 8);
    chunked_row(2, 3, N, C, 6, 2, 8); chunked_row(2, 3, N, C, 9, 0, 8);
    chunked_row(3, 4, N, C, 6, 1, 8); chunked_row(3, 4, N, C, 6, 3, 8); chunked_row(3, 4, N, C, 9, 0, 8);
    chunked_row(4, 5, N, C, 6, 0, 8); chunked_row(4, 5, N, C, 1, 0, 8); chunked_row(4, 5, N, C, 6, 4, 8);
    chunked_row(4, 5, N, C, 6, 0, 9); chunked_row(4, 5, N, C, 6, 0, 1);
    /* more levels / strategies on the mixed corpus, and the tail-mode geometry */
    for (int l = 2; l <= 8; l++) if (l != 6) chunked_row(4, 5, N, C, l, 0, 8);
    chunked_row(4, 5, N, C, 3, 1, 8); chunked_row(4, 5, N, C, 9, 1, 8); chunked_row(4, 5, N, C, 1, 3, 8); chunked_row(4, 5, N, C, 1, 4, 8);
    chunked_row(0, 7, 2u * 1048576u + 65374u, C, 6, 0, 8); chunked_row(0, 7, 2u * 1048576u + 32600u, C, 6, 0, 8);
    /* level 0 (deflate_stored) */
    chunked_row(4, 5, N, C, 0, 0, 8); chunked_row(1, 2, 2u * 1048576u + 777u, C, 0, 0, 1); chunked_row(0, 1, 1048576u + 65531u, C, 0, 4, 9);
    chunked_row(4, 9, 3u * 262144u + 777u, 262144u, 6, 0, 8); chunked_row(0, 9, 3u * 65536u + 5u, 65536u, 9, 0, 8);
    printf("\n],\n\"kat\": [\n"); first = 1;
    uint8_t b[20000]; size_t n;
    memset(b, 'a', 40); kat("a40", b, 40, 6, 0);
    kat("empty", b, 0, 6, 0);
    n = 0; for (int i = 0; i < 64; i++) for (int k = 0; k < 4; k++) b[n++] = (uint8_t)i; kat("ties64x4", b, n, 6, 2);
    { unsigned fib[19] = {1, 1}; for (int i = 2; i < 19; i++) fib[i] = fib[i - 1] + fib[i - 2];
      n = 0; for (int i = 0; i < 17; i++) for (unsigned k = 0; k < fib[i]; k++) b[n++] = (uint8_t)i; kat("fib17", b, n, 6, 2);
      n = 0; for (int i = 0; i < 19; i++) for (unsigned k = 0; k < fib[i]; k++) b[n++] = (uint8_t)i; kat("fib19", b, n, 6, 2); }
    for (int i = 0; i < 256; i++) b[i] = (uint8_t)i; kat("range256", b, 256, 6, 2);
    memcpy(b, "hello, hello!", 14); kat("hello", b, 14, 6, 0); kat("hello_fixed", b, 14, 6, 4);
    for (int i = 0; i < 3000; i++) { b[2 * i] = 'a'; b[2 * i + 1] = 'b'; } b[6000] = 'c'; kat("ab3000c", b, 6001, 9, 0);
    memset(b, 'a', 40); kat("a40_rle", b, 40, 6, 3);
    memcpy(b, "abababababab", 12); memset(b + 12, 'z', 300); kat("ab6z300_rle", b, 312, 6, 3);
    /* the boundary cases of the stream API: dictionaries of any length, windowBits 9..14, every flush value, data_type */
    printf("\n],\n\"boundary\": [\n"); first = 1;
    { static const size_t dl[] = {1, 2, 3, 15, 17, 1001, 32767, 32768};
      for (int i = 0; i < 8; i++) { boundary_row(0, 11, 200000, dl[i], 6, 0, 15, Z_SYNC_FLUSH); boundary_row(4, 12, 150001, dl[i], 1, 0, 15, Z_FINISH); }
      boundary_row(3, 13, 100000, 50, 9, 1, 15, Z_SYNC_FLUSH); boundary_row(2, 14, 90000, 777, 6, 3, 15, Z_SYNC_FLUSH); boundary_row(0, 15, 70000, 4097, 4, 4, 15, Z_FINISH); }
    for (int w = 9; w <= 14; w++) {
        boundary_row(0, 20 + w, 300000, (size_t)1 << w, 6, 0, w, Z_SYNC_FLUSH); boundary_row(4, 30 + w, 200000, 0, 9, 0, w, Z_FINISH);
        boundary_row(3, 40 + w, 150000, ((size_t)1 << w) - 5, 1, 0, w, Z_SYNC_FLUSH); boundary_row(2, 50 + w, 100000, 100, 3, 0, w, Z_FINISH);
        boundary_row(0, 60 + w, 80000, (size_t)1 << w, 6, 3, w, Z_SYNC_FLUSH); boundary_row(4, 70 + w, 65536 + 300, (size_t)1 << w, 0, 0, w, Z_SYNC_FLUSH);
    }
    { static const int fl[] = {Z_PARTIAL_FLUSH, Z_BLOCK, Z_FULL_FLUSH};
      for (int f = 0; f < 3; f++) for (int k = 0; k < 5; k++) {
          boundary_row(k, 80 + k, 100000 + 7 * k, k & 1 ? 32768 : 0, k == 2 ? 1 : 6, 0, 15, fl[f]);
          boundary_row(k, 90 + k, 3 * k, 0, 6, 0, 15, fl[f]); } }
    for (int k = 0; k < 5; k++) { boundary_row(k, 100 + k, 50000, 0, 6, 0, 15, Z_FINISH); boundary_row(k, 110 + k, 50000, 0, 0, 0, 15, Z_FINISH); boundary_row(k, 120 + k, 40000, 0, 6, 2, 15, Z_FINISH); }
    printf("\n]}\n");
    return 0;
}
