/*
 * cpu_baseline.c — times the chunk recipe on host cores (TEST/BENCH
 * INFRASTRUCTURE ONLY).  Links the UNMODIFIED reference build
 * (oracle/_ref/libzref.so): deflateInit2(-15) + deflateSetDictionary(32 KiB) +
 * one deflate(Z_SYNC_FLUSH | Z_FINISH) per chunk, one chunk per task over
 * `threads` pthreads, inputs resident in RAM, file I/O excluded.
 *
 * usage: cpu_baseline kind seed total_bytes chunk_size level strategy mem_level threads [piece_bytes] [port]
 *        piece_bytes > 0: piece i of the corpus is generated with seed + i
 *        "port" as last argument times the oracle restatement instead of the reference
 * After the timed deflate pass the same threads inflate the chunks again
 * (inflateInit2(-15) + inflateSetDictionary + inflate per chunk, reference only)
 * and one thread compresses the first `single_bytes` (64 MiB unless the sample is
 * smaller) as ONE stream, which gives the ratio the chunked output is compared to.
 * prints one JSON object.
 */
#define _POSIX_C_SOURCE 200809L
#include "zoracle.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "zlib.h"

static uint8_t *g_in; static size_t g_n, g_chunk; static int g_level, g_strat, g_mem, g_threads, g_port;
static size_t *g_out_len; static volatile long g_next;
static uint8_t **g_out; static int g_phase; static long g_bad;

static void *worker(void *arg)
{
    (void)arg;
    size_t nchunks = (g_n + g_chunk - 1) / g_chunk;
    uint8_t *out = malloc(g_chunk + g_chunk / 4 + 65536);
    for (;;) {
        long c = __sync_fetch_and_add(&g_next, 1);
        if ((size_t)c >= nchunks) break;
        size_t off = (size_t)c * g_chunk, len = g_n - off < g_chunk ? g_n - off : g_chunk, dict = off < 32768 ? off : 32768;
        int final = off + len >= g_n;
        if (g_phase == 1) {           /* inflate pass */
            z_stream z; memset(&z, 0, sizeof z);
            inflateInit2(&z, -15);
            if (dict) inflateSetDictionary(&z, g_in + off - dict, (uInt)dict);
            z.next_in = g_out[c]; z.avail_in = (uInt)g_out_len[c]; z.next_out = out; z.avail_out = (uInt)(g_chunk + 64);
            int r = inflate(&z, Z_SYNC_FLUSH);
            if ((r != Z_OK && r != Z_STREAM_END) || z.total_out != len || memcmp(out, g_in + off, len)) __sync_fetch_and_add(&g_bad, 1);
            inflateEnd(&z);
            continue;
        }
        if (g_port) {
            g_out_len[c] = zo_deflate_chunk(g_in + off - dict, dict, len, g_level, g_strat, g_mem, 15, final, out, g_chunk + g_chunk / 4 + 65536, NULL);
        } else {
            z_stream z; memset(&z, 0, sizeof z);
            deflateInit2(&z, g_level, Z_DEFLATED, -15, g_mem, g_strat);
            if (dict) deflateSetDictionary(&z, g_in + off - dict, (uInt)dict);
            z.next_in = g_in + off; z.avail_in = (uInt)len; z.next_out = out; z.avail_out = (uInt)(g_chunk + g_chunk / 4 + 65536);
            deflate(&z, final ? Z_FINISH : Z_SYNC_FLUSH);
            g_out_len[c] = z.total_out;
            deflateEnd(&z);
            g_out[c] = malloc(z.total_out + 8);
            memcpy(g_out[c], out, z.total_out);
        }
    }
    free(out);
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 9) { fprintf(stderr, "usage: see source\n"); return 2; }
    int kind = atoi(argv[1]); uint64_t seed = strtoull(argv[2], 0, 10);
    g_n = strtoull(argv[3], 0, 10); g_chunk = strtoull(argv[4], 0, 10);
    g_level = atoi(argv[5]); g_strat = atoi(argv[6]); g_mem = atoi(argv[7]); g_threads = atoi(argv[8]);
    size_t piece = argc > 9 ? strtoull(argv[9], 0, 10) : 0;
    g_port = argc > 10 && !strcmp(argv[10], "port");
    g_in = malloc(g_n + 16);
    if (piece) { for (size_t i = 0, off = 0; off < g_n; i++, off += piece) zo_corpus_fill(g_in + off, g_n - off < piece ? g_n - off : piece, kind, seed + i); }
    else zo_corpus_fill(g_in, g_n, kind, seed);
    size_t nchunks = (g_n + g_chunk - 1) / g_chunk;
    g_out_len = calloc(nchunks + 1, sizeof(size_t));
    g_out = calloc(nchunks + 1, sizeof(uint8_t *));
    struct timespec t0, t1;
    pthread_t th[256];
    if (g_threads > 256) g_threads = 256;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < g_threads; i++) pthread_create(&th[i], NULL, worker, NULL);
    for (int i = 0; i < g_threads; i++) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    size_t total = 0; for (size_t c = 0; c < nchunks; c++) total += g_out_len[c];
    double inf_sec = 0;
    if (!g_port) {
        g_phase = 1; g_next = 0;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        for (int i = 0; i < g_threads; i++) pthread_create(&th[i], NULL, worker, NULL);
        for (int i = 0; i < g_threads; i++) pthread_join(th[i], NULL);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        inf_sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    }
    /* single-stream reference ratio on a bounded prefix, and the chunked size of the same prefix */
    size_t single = g_n < ((size_t)64 << 20) ? g_n : ((size_t)64 << 20), single_out = 0, chunked_same = 0;
    single = single / g_chunk * g_chunk;
    if (!g_port && single) {
        size_t cap = single + single / 4 + 65536;
        uint8_t *o = malloc(cap);
        z_stream z; memset(&z, 0, sizeof z);
        deflateInit2(&z, g_level, Z_DEFLATED, -15, g_mem, g_strat);
        z.next_in = g_in; z.avail_in = (uInt)single; z.next_out = o; z.avail_out = (uInt)cap;
        deflate(&z, Z_FINISH);
        single_out = z.total_out;
        deflateEnd(&z); free(o);
        for (size_t c = 0; c < single / g_chunk; c++) chunked_same += g_out_len[c];
    }
    printf("{\"bytes\": %zu, \"out_bytes\": %zu, \"seconds\": %.6f, \"threads\": %d, \"kind\": \"%s\", \"zlib\": \"%s\", "
           "\"inflate_seconds\": %.6f, \"inflate_bad\": %ld, \"single_bytes\": %zu, \"single_out\": %zu, \"chunked_out_same_prefix\": %zu}\n",
           g_n, total, sec, g_threads, g_port ? "port" : "reference", zlibVersion(), inf_sec, g_bad, single, single_out, chunked_same);
    return 0;
}
