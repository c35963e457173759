/*
 * zoracle.h — CPU oracle for the chunked-DEFLATE path (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C restatement of the algorithms of zlib 1.2.8 that sit on the
 * accelerated path: LZ77 parsing (deflate.c), Huffman block coding (trees.c),
 * adler32/crc32 with combine (adler32.c, crc32.c) and raw inflate (inflate.c,
 * inffast.c, inftrees.c).  It works in absolute buffer coordinates instead of a
 * sliding 64 KiB window and exists only so that tests, the smoke check and the
 * CPU-baseline leg of the benchmark have something to compare the HIP kernels
 * against.  Nothing in the product library links, loads or calls it.
 *
 * Parity pin: every function here is checked against a build of the unmodified
 * reference sources (oracle/_ref/libzref.so, built by oracle/Makefile from the
 * files where they lie) by oracle/fuzz_vs_ref.c and tests/test_oracle_vs_ref.py,
 * and against the golden vectors committed under tests/golden/.
 */
#ifndef ZORACLE_H
#define ZORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* strategy values, zlib.h:193-197 */
enum { ZO_DEFAULT = 0, ZO_FILTERED = 1, ZO_HUFFMAN_ONLY = 2, ZO_RLE = 3, ZO_FIXED = 4 };

/* One LZ77 symbol as tallied by _tr_tally (trees.c:1010): dist==0 -> literal `lc`,
 * else match of length lc+3 at distance dist.  `pos` = absolute position of the
 * first input byte the symbol covers (dictionary included). */
typedef struct { uint32_t pos; uint16_t dist; uint16_t lc; } zo_sym;

/* One emitted block (one _tr_flush_block call, trees.c:907). */
typedef struct {
    uint32_t sym_start, sym_count;   /* range in the symbol trace              */
    uint32_t in_start, in_len;       /* input bytes covered (stored_len)       */
    uint32_t type;                   /* 0 stored, 1 static, 2 dynamic          */
    uint32_t last;                   /* BFINAL                                 */
    uint32_t can_store;              /* block_start still inside the window    */
    uint64_t opt_len, static_len;    /* bit costs that drove the choice        */
    uint64_t bit_start, bit_end;     /* position in the chunk's output         */
} zo_block;

typedef struct {
    zo_sym   *syms;   size_t n_syms,   cap_syms;
    zo_block *blocks; size_t n_blocks, cap_blocks;
} zo_trace;

void zo_trace_free(zo_trace *t);

/*
 * Compress one chunk exactly as the reference does for
 *   deflateInit2(level, Z_DEFLATED, -wbits, mem_level, strategy);
 *   if (dict_len) deflateSetDictionary(buf, dict_len);
 *   deflate(next_in = buf + dict_len, avail_in = n, final ? Z_FINISH : Z_SYNC_FLUSH);
 * buf[0 .. dict_len) is the dictionary (at most 1<<wbits bytes, the caller
 * passes the tail), buf[dict_len .. dict_len+n) the chunk.  Returns the number
 * of output bytes, or (size_t)-1 if out_cap is too small or a parameter is out
 * of range.  `trace` may be NULL.
 */
size_t zo_deflate_chunk_tuned(const uint8_t *buf, size_t dict_len, size_t n,
                              int level, int strategy, int mem_level, int wbits,
                              int final, const int *tune /* good, lazy, nice, chain or NULL */,
                              uint8_t *out, size_t out_cap, zo_trace *trace);
/* ... and with prime_bits (0..7) low bits of prime_val in front of the first block (deflatePrime, deflate.c:464-487) */
size_t zo_deflate_chunk_primed(const uint8_t *buf, size_t dict_len, size_t n,
                               int level, int strategy, int mem_level, int wbits,
                               int final, const int *tune, int prime_bits, int prime_val,
                               uint8_t *out, size_t out_cap, zo_trace *trace);
size_t zo_deflate_chunk(const uint8_t *buf, size_t dict_len, size_t n,
                        int level, int strategy, int mem_level, int wbits,
                        int final, uint8_t *out, size_t out_cap, zo_trace *trace);
/* ... and the general form: `flush` is the flush value of the one deflate() call (deflate.c:923-945).  Z_PARTIAL_FLUSH
 * (_tr_align, trees.c:892-901) and Z_BLOCK can end inside a byte: the return value counts whole bytes, *pend_bits /
 * *pend_val the bits left in the bit buffer (deflatePending, deflate.c:450-461), *data_type what strm->data_type is
 * afterwards (trees.c:920-921, 1120-1147; 2 = Z_UNKNOWN).  The three result pointers may be NULL. */
enum { ZO_FLUSH_SYNC = 0, ZO_FLUSH_FINISH = 1, ZO_FLUSH_PARTIAL = 2, ZO_FLUSH_BLOCK = 3 };
size_t zo_deflate_chunk_ex(const uint8_t *buf, size_t dict_len, size_t n,
                           int level, int strategy, int mem_level, int wbits,
                           int flush, const int *tune, int prime_bits, int prime_val,
                           uint8_t *out, size_t out_cap, zo_trace *trace,
                           int *pend_bits, int *pend_val, int *data_type);

/* Whole-buffer chunked compression (the recipe of the accelerated path):
 * chunk i is primed with the previous min(32768, offset) bytes, every chunk but
 * the last ends with Z_SYNC_FLUSH, the last with Z_FINISH.  chunk_len[i]
 * receives each chunk's compressed size.  Returns total bytes or (size_t)-1. */
size_t zo_deflate_chunked(const uint8_t *in, size_t n, size_t chunk_size,
                          int level, int strategy, int mem_level,
                          uint8_t *out, size_t out_cap,
                          uint32_t *chunk_len, size_t max_chunks);
/* the same with a window of 1 << wbits bytes (9..15): the dictionary of a chunk is the previous 1 << wbits bytes */
size_t zo_deflate_chunked_w(const uint8_t *in, size_t n, size_t chunk_size,
                            int level, int strategy, int mem_level, int wbits,
                            uint8_t *out, size_t out_cap,
                            uint32_t *chunk_len, size_t max_chunks);

/* adler32.c:65, crc32.c:204 and their combine functions (adler32.c:136, crc32.c:355) */
uint32_t zo_adler32(uint32_t adler, const uint8_t *buf, size_t len);
uint32_t zo_crc32(uint32_t crc, const uint8_t *buf, size_t len);
uint32_t zo_adler32_combine(uint32_t a1, uint32_t a2, int64_t len2);
uint32_t zo_crc32_combine(uint32_t c1, uint32_t c2, int64_t len2);

/* Raw inflate of a complete deflate stream (inflate.c:605 in raw mode, one call
 * with Z_FINISH semantics).  `dict`/`dict_len` prime the window as
 * inflateSetDictionary does.  Returns 0 (Z_OK/STREAM_END reached), -3 data
 * error (msg set), -5 output/input exhausted before the final block ended.
 * *consumed / *produced receive byte counts; *msg the reference's error text. */
int zo_inflate_raw(const uint8_t *in, size_t in_len, const uint8_t *dict, size_t dict_len,
                   uint8_t *out, size_t out_cap, size_t *consumed, size_t *produced,
                   const char **msg, int stop_at_sync /* stop after an empty stored block */);
/* the same from a block that starts at bit start_bit of `in` */
int zo_inflate_raw_at(const uint8_t *in, size_t in_len, uint64_t start_bit, const uint8_t *dict, size_t dict_len,
                      uint8_t *out, size_t out_cap, size_t *consumed, size_t *produced,
                      const char **msg, int stop_at_sync);

/* Deterministic corpus generator shared by fixtures and tests
 * (kinds: 0 text, 1 binary, 2 lowent, 3 image, 4 mixed). */
void zo_corpus_fill(uint8_t *buf, size_t n, int kind, uint64_t seed);

#ifdef __cplusplus
}
#endif
#endif
