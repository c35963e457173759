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
    struct timespec t0, t1;
    pthread_t th[256];
    if (g_threads > 256) g_threads = 256;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (int i = 0; i < g_threads; i++) pthread_create(&th[i], NULL, worker, NULL);
    for (int i = 0; i < g_threads; i++) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    size_t total = 0; for (size_t c = 0; c < nchunks; c++) total += g_out_len[c];
    printf("{\"bytes\": %zu, \"out_bytes\": %zu, \"seconds\": %.6f, \"threads\": %d, \"kind\": \"%s\", \"zlib\": \"%s\"}\n",
           g_n, total, sec, g_threads, g_port ? "port" : "reference", zlibVersion());
    return 0;
}
