/*
 * zamd.h — C ABI of the MI355X DEFLATE engine (libzamd.so).
 *
 * Two layers are exported:
 *   1. the engine layer below (zamd_*): device-resident, batch oriented — what a
 *      host program that already keeps its data in HBM calls, and what the
 *      zlib-compatible layer is built on;
 *   2. the zlib 1.2.8 stream API (include/zamd_zlib.h): z_stream, deflateInit2_,
 *      deflate, inflate, compress2, uncompress, adler32, crc32, ... with the
 *      reference's signatures, so the library links where libz links.
 *
 * All pointers named d_* are device pointers valid on the engine's GPU; every
 * other pointer is host memory.  No torch or C++ types appear in signatures.
 * Return values are zlib's (zlib.h:173-181): 0 Z_OK, -2 Z_STREAM_ERROR,
 * -4 Z_MEM_ERROR, -5 Z_BUF_ERROR; -100 - hipError for a HIP failure.
 *
 * File:line citations name the reference interface each entry replaces.
 */
#ifndef ZAMD_H
#define ZAMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct zamd_engine zamd_engine;

#define ZAMD_OK            0
#define ZAMD_STREAM_ERROR (-2)
#define ZAMD_DATA_ERROR   (-3)
#define ZAMD_MEM_ERROR    (-4)
#define ZAMD_BUF_ERROR    (-5)
#define ZAMD_HIP_ERROR(e) (-100 - (int)(e))

/* Engine life cycle.  One engine per (process, GPU); it owns two HIP streams and
 * a workspace of at most `workspace_bytes` (0 = default of 48 GiB, shrunk to fit
 * the input).  Plays the role of the allocations deflateInit2_ makes
 * (deflate.c:293-303) for all streams at once. */
int  zamd_engine_create(zamd_engine **out, int device, uint64_t workspace_bytes);
void zamd_engine_destroy(zamd_engine *e);

/* Upper bound of the compressed size of n bytes cut into chunk_size chunks
 * (replaces deflateBound, deflate.c:566-621, which does not cover Z_SYNC_FLUSH
 * output, zlib.h:708-710). */
uint64_t zamd_deflate_bound(uint64_t n, uint32_t chunk_size, int mem_level);

/*
 * Chunked raw DEFLATE of a device buffer.  The input is cut every chunk_size
 * bytes (multiple of 16384, >= 32768); chunk i is compressed exactly as
 *   deflateInit2(level, Z_DEFLATED, -15, mem_level, strategy)       deflate.c:213
 *   deflateSetDictionary(previous 32 KiB of input)   (i > 0)        deflate.c:323
 *   deflate(Z_SYNC_FLUSH), or Z_FINISH for the last chunk if finish deflate.c:665
 * and the chunks are concatenated into d_out, giving one valid raw deflate
 * stream.  d_in and d_out must be 16-byte aligned.  d_chunk_off (device,
 * n_chunks + 1 entries, may be NULL) receives each chunk's byte offset in d_out
 * followed by the total; *total_out (host) receives the total.
 * level 1..9, strategy 0..4 (zlib.h:193-197), mem_level 1..9.
 * Blocks the calling thread until the result is in d_out.
 */
int zamd_deflate_dev(zamd_engine *e, const void *d_in, uint64_t n, uint32_t chunk_size,
                     int level, int strategy, int mem_level, int finish,
                     void *d_out, uint64_t out_cap, uint64_t *d_chunk_off, uint64_t *total_out);

/*
 * The same for one shard of a larger input (multi-GPU sharding by chunk range):
 * the head_dict_len (0 or 32768) bytes in front of d_in are device-resident too
 * and are the dictionary of the shard's first chunk, exactly as the previous
 * 32 KiB are for every later chunk.  Pass finish = 1 only on the shard that
 * holds the end of the input.
 */
int zamd_deflate_shard_dev(zamd_engine *e, const void *d_in, uint64_t n, uint32_t head_dict_len, uint32_t chunk_size,
                           int level, int strategy, int mem_level, int finish,
                           void *d_out, uint64_t out_cap, uint64_t *d_chunk_off, uint64_t *total_out);

/*
 * Per-chunk checksums of a device buffer: adler32 (adler32.c:65) and crc32
 * (crc32.c:204) of every chunk_size slice, each started from the initial value
 * (1 / 0).  d_adler / d_crc receive n_chunks values (either may be NULL).
 * Combine them with zamd_adler32_combine / zamd_crc32_combine.
 */
int zamd_checksums_dev(zamd_engine *e, const void *d_in, uint64_t n, uint32_t chunk_size,
                       uint32_t *d_adler, uint32_t *d_crc);

/* Whole-buffer checksums of a device buffer, continuing from `adler` / `crc`
 * (the value semantics of adler32() / crc32(), zlib.h:1569-1617). */
int zamd_adler32_dev(zamd_engine *e, uint32_t adler, const void *d_in, uint64_t n, uint32_t *result);
int zamd_crc32_dev(zamd_engine *e, uint32_t crc, const void *d_in, uint64_t n, uint32_t *result);

/* adler32_combine (adler32.c:136-162) and crc32_combine (crc32.c:327-408). */
uint32_t zamd_adler32_combine(uint32_t adler1, uint32_t adler2, int64_t len2);
uint32_t zamd_crc32_combine(uint32_t crc1, uint32_t crc2, int64_t len2);

/*
 * Batch inflate of a chunk-indexed raw deflate stream produced by
 * zamd_deflate_dev (or by the reference with the same chunk recipe): chunk i
 * occupies d_comp[chunk_off[i], chunk_off[i+1]) and inflates to chunk_size bytes
 * (the last one to the remainder of n_out).  Replaces a loop of
 * inflateSetDictionary + inflate(Z_SYNC_FLUSH) per chunk (inflate.c:605, 1291;
 * the access pattern of examples/zran.c:249-349).  Returns ZAMD_DATA_ERROR and
 * the failing chunk in *bad_chunk if a stream is invalid.
 */
int zamd_inflate_dev(zamd_engine *e, const void *d_comp, const uint64_t *d_chunk_off, uint64_t n_chunks,
                     uint32_t chunk_size, void *d_out, uint64_t n_out, int64_t *bad_chunk);

/* Timing of the last zamd_deflate_dev call, measured with HIP events on the
 * engine's own streams: `which` 0 = chain build, 1 = match search, 2 = parse,
 * 3 = block planning, 4 = layout, 5 = bit emission, 6 = whole call.
 * Returns the summed milliseconds; *launches receives the number of launches. */
double zamd_last_timing_ms(zamd_engine *e, int which, int *launches);

/* Deterministic synthetic corpus used by tests and bench (host memory):
 * kind 0 text, 1 binary, 2 low-entropy, 3 image-like, 4 mixed 64 KiB segments. */
void zamd_corpus_fill(uint8_t *buf, uint64_t n, int kind, uint64_t seed);

/* Debug/inspection: copies an intermediate buffer of the LAST deflate call's
 * first batch for local chunk index `chunk` to host memory.  what: 0 = chain
 * links (u16 per chunk byte), 1 = match records (2 x u32 per chunk byte),
 * 2 = symbols (u32 each), 3 = block table (struct ZcBlock).  Returns the number
 * of elements copied or a negative error. */
int64_t zamd_debug_dump(zamd_engine *e, int what, uint32_t chunk, void *host_buf, uint64_t max_elems);

/* Large corpora: piece i (piece_bytes long, 0 = 16 MiB) is zamd_corpus_fill(kind, seed + i),
 * generated by `threads` host threads. */
void zamd_corpus_fill_pieces(uint8_t *buf, uint64_t n, int kind, uint64_t seed, uint64_t piece_bytes, int threads);

/* Thin device-memory helpers so that a C or ctypes caller needs no HIP headers
 * (hipMalloc / hipFree / hipMemcpy on the engine's device). */
int zamd_dev_alloc(zamd_engine *e, uint64_t bytes, void **d_ptr);
int zamd_dev_free(zamd_engine *e, void *d_ptr);
int zamd_dev_upload(zamd_engine *e, void *d_dst, const void *h_src, uint64_t bytes);
int zamd_dev_download(zamd_engine *e, void *h_dst, const void *d_src, uint64_t bytes);
int zamd_dev_memset(zamd_engine *e, void *d_dst, int value, uint64_t bytes);

const char *zamd_version(void);

#ifdef __cplusplus
}
#endif
#endif
