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
 * level 0..9 (0 = deflate_stored, deflate.c:1564), strategy 0..4 (zlib.h:193-197),
 * mem_level 1..9.
 * Blocks the calling thread until the result is in d_out.
 */
int zamd_deflate_dev(zamd_engine *e, const void *d_in, uint64_t n, uint32_t chunk_size,
                     int level, int strategy, int mem_level, int finish,
                     void *d_out, uint64_t out_cap, uint64_t *d_chunk_off, uint64_t *total_out);

/*
 * The same for one shard of a larger input (multi-GPU sharding by chunk range):
 * the head_dict_len bytes (any length up to the window size, 32768 at windowBits 15) in front of d_in are
 * device-resident too and are the dictionary of the shard's first chunk, exactly
 * as the previous 32 KiB are for every later chunk — the tail of the previous
 * shard, or a preset dictionary (deflateSetDictionary, deflate.c:323).  Pass
 * finish = 1 only on the shard that holds the end of the input.  d_in itself must be 16-byte aligned; a
 * head_dict_len that is not a multiple of 16 leaves the dictionary's first byte unaligned, and the kernels' 16-byte
 * loads of the first chunk's window then run in the hardware's unaligned mode (slower for that chunk, same bytes).
 */
int zamd_deflate_shard_dev(zamd_engine *e, const void *d_in, uint64_t n, uint32_t head_dict_len, uint32_t chunk_size,
                           int level, int strategy, int mem_level, int finish,
                           void *d_out, uint64_t out_cap, uint64_t *d_chunk_off, uint64_t *total_out);

/* deflateTune (deflate.c:531-547) for the following deflate calls of this engine:
 * good_length, max_lazy (max_insert_length at levels 1-3), nice_length, max_chain
 * instead of the level's table values; all four 0 = back to the table.  max_chain
 * must be >= 1, and <= 64 at levels 1-3.  Values the precomputed-record matcher cannot
 * express (nice_length < max_lazy, max_chain < 4) switch levels 4-9 to the slower
 * matcher that follows the parse; the output stays the reference's. */
int zamd_engine_tune(zamd_engine *e, int good_length, int max_lazy, int nice_length, int max_chain);

/* windowBits of the following deflate calls of this engine: 9..15, default 15 (deflate.c:271-291: w_size = 1 << windowBits,
 * MAX_DIST = w_size - 262; a chunk's dictionary is then the previous w_size bytes, and head_dict_len at most w_size). */
int zamd_engine_window_bits(zamd_engine *e, int window_bits);

/* How the next deflate call of this engine (only) ends when finish == 0: what deflate() does after its last block for
 * Z_SYNC_FLUSH / Z_FULL_FLUSH (0: the empty stored block 00 00 FF FF, deflate.c:926-927), Z_PARTIAL_FLUSH (1: _tr_align,
 * an empty static block of 10 bits, trees.c:892-901) or Z_BLOCK (2: nothing, deflate.c:923-925).  Chunks inside the call
 * still end on sync markers.  With 1 and 2 the output can end inside a byte: *total_out counts the complete bytes and
 * zamd_engine_pending_bits returns the bits (fewer than 8) behind them, which the reference keeps in its bit buffer
 * (deflatePending, deflate.c:450-461); hand them to the next call with zamd_engine_prime. */
#define ZAMD_TAIL_SYNC  0
#define ZAMD_TAIL_ALIGN 1
#define ZAMD_TAIL_NONE  2
int zamd_engine_tail(zamd_engine *e, int mode);
int zamd_engine_pending_bits(zamd_engine *e, int *bits, int *value);

/* strm->data_type as the last deflate call would have set it (trees.c:920-921, 1120-1147: detect_data_type on the
 * literal counts of the first block at level > 0): 0 Z_BINARY, 1 Z_TEXT, 2 Z_UNKNOWN (no block, or level 0). */
int zamd_engine_data_type(zamd_engine *e);

/* Split points for the next deflate call of this engine (only): besides the chunk offsets the call records, for every
 * chunk, up to parts - 1 block boundaries that cut the chunk into parts of about equal output, so that a reader can put
 * `parts` wavefronts on one chunk (zamd_inflate_split_dev).  d_splits (device) holds parts - 1 records per chunk, indexed
 * by chunk number as d_chunk_off is: (bit offset of the block from the chunk's first byte) << 32 | offset of the block's
 * first byte in the chunk's data, ascending, zeros behind the last one (a chunk of few blocks has fewer).  2 <= parts
 * <= ZAMD_MAX_PARTS; d_splits = NULL cancels.  The reference has no counterpart: its access points are whole-window
 * snapshots taken while inflating (examples/zran.c:249-349). */
#define ZAMD_MAX_PARTS 16
int zamd_engine_splits(zamd_engine *e, uint64_t *d_splits, uint32_t parts);

/* deflatePrime (deflate.c:464-487) for the next deflate call of this engine only: `bits`
 * (0..7) low bits of `value` precede the first block of the call's first chunk, which
 * starts inside output byte 0 as it does in the reference after deflatePrime. */
int zamd_engine_prime(zamd_engine *e, int bits, int value);

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

/*
 * zamd_adler32_dev and zamd_crc32_dev in one call that allocates nothing: no counterpart in the reference; it is for
 * callers that cannot allocate (tight serving loops, harnesses that forbid hipMalloc).  It launches on the engine's own
 * stream and waits for it, so it cannot itself be recorded into a HIP graph.
 * The per-slice partial checksums go to a fixed array inside the library's code object (128 KiB per GPU, held until the
 * process ends) instead of a hipMalloc'ed buffer, and are folded on the GPU: d_result (DEVICE memory, 2 x uint32) receives
 * adler32(adler, d_in, n) and crc32(crc, d_in, n).  n is at most 16 GiB (16384 slices of 1 MiB): larger inputs are
 * refused with ZAMD_BUF_ERROR and a message on stderr.  Every call on a GPU shares that GPU's array: calls that overlap
 * on one GPU race.  Blocks the calling thread until the result is in d_result.
 */
int zamd_checksums_noalloc_dev(zamd_engine *e, uint32_t adler, uint32_t crc, const void *d_in, uint64_t n, uint32_t *d_result);

/* adler32_combine (adler32.c:136-162) and crc32_combine (crc32.c:327-408). */
uint32_t zamd_adler32_combine(uint32_t adler1, uint32_t adler2, int64_t len2);
uint32_t zamd_crc32_combine(uint32_t crc1, uint32_t crc2, int64_t len2);

/*
 * Batch inflate of a chunk-indexed raw deflate stream produced by
 * zamd_deflate_dev (or by the reference with the same chunk recipe): chunk i
 * occupies d_comp[chunk_off[i], chunk_off[i+1]) and inflates to chunk_size bytes
 * (the last one to the remainder of n_out).  Replaces a loop of
 * inflateSetDictionary + inflate(Z_SYNC_FLUSH) per chunk (inflate.c:605, 1291;
 * the access pattern of examples/zran.c:249-349).  d_comp must be 4-byte aligned
 * and readable up to the next multiple of 4 bytes, d_out 16-byte aligned,
 * chunk_size a multiple of 8.  Returns ZAMD_DATA_ERROR and the failing chunk in
 * *bad_chunk if a stream is invalid (zamd_inflate_last_error gives the reason).
 */
int zamd_inflate_dev(zamd_engine *e, const void *d_comp, const uint64_t *d_chunk_off, uint64_t n_chunks,
                     uint32_t chunk_size, void *d_out, uint64_t n_out, int64_t *bad_chunk);

/* The same for one shard of a larger stream: d_chunk_off holds offsets relative
 * to d_comp (comp_bytes long); d_head_window (device, 32768 bytes, or NULL for
 * the shard that starts the stream) is the inflated data in front of the shard,
 * i.e. the last 32 KiB produced by the previous shard. */
int zamd_inflate_shard_dev(zamd_engine *e, const void *d_comp, uint64_t comp_bytes, const uint64_t *d_chunk_off, uint64_t n_chunks,
                           uint32_t chunk_size, const void *d_head_window, void *d_out, uint64_t n_out, int64_t *bad_chunk);

/* The same with `parts` wavefronts to a chunk: d_splits (device, parts - 1 records per chunk) is what
 * zamd_engine_splits made the deflate call record.  The records are hints of where blocks start: one that is out of
 * order or outside its chunk ends its chunk's list, one that is not a block boundary makes its chunk fail
 * (ZAMD_DATA_ERROR, *bad_chunk) like any other invalid data.  d_splits = NULL or parts < 2: zamd_inflate_shard_dev.
 * chunk_size must be a multiple of 16. */
int zamd_inflate_split_dev(zamd_engine *e, const void *d_comp, uint64_t comp_bytes, const uint64_t *d_chunk_off, uint64_t n_chunks,
                           uint32_t chunk_size, const uint64_t *d_splits, uint32_t parts, const void *d_head_window,
                           void *d_out, uint64_t n_out, int64_t *bad_chunk);

/*
 * The same in three steps, for ranks that hold consecutive shards of ONE stream and so do not have the window in
 * front of their shard until the rank before them has produced it (the inflate side of chunk-range sharding; the
 * reference's counterpart is the window updatewindow() carries from call to call, inflate.c:379-427):
 *   begin   decodes every chunk and resolves the last 32 KiB of each as far as they depend on this shard alone;
 *           bytes that come from the window in front of the shard stay references (has_head_window = 0 on the
 *           shard that starts the stream).  chunk_size >= 32768, n_chunks < 32768.
 *   tail    d_tail_window (device, 32768 bytes) receives the last 32 KiB of (d_head_window ++ this shard's output):
 *           the next shard's window, a 32 KiB gather that does not wait for the bulk of the output.
 *           d_head_window is the previous shard's tail window (ignored when has_head_window was 0).
 *   finish  completes d_out.
 * Rank r: begin; receive the window from r-1; tail; send it to r+1; finish.
 * The three calls share the engine's inflate workspace with every other inflate entry point: any other inflate call
 * of this engine between begin and finish ends the sharded inflate, and its tail / finish calls then return
 * ZAMD_STREAM_ERROR.
 */
int zamd_inflate_shard_begin_dev(zamd_engine *e, const void *d_comp, uint64_t comp_bytes, const uint64_t *d_chunk_off, uint64_t n_chunks,
                                 uint32_t chunk_size, int has_head_window, void *d_out, uint64_t n_out, int64_t *bad_chunk);
int zamd_inflate_shard_tail_dev(zamd_engine *e, const void *d_head_window, void *d_tail_window);
int zamd_inflate_shard_finish_dev(zamd_engine *e, const void *d_head_window);

/* Number of batches one deflate call keeps in flight on separate HIP streams (1..8, default 4).  serialize != 0:
 * the batches are cut the same way but all issued to one stream, so that every kernel runs alone on the GPU, which
 * is what per-kernel timing needs (zamd_last_timing_ms then returns durations that add up, not overlapping spans). */
int zamd_engine_set_streams(zamd_engine *e, int nstreams, int serialize);
/* The number of streams deflate calls are spread over at present (0: no engine), and the serialize flag. */
int zamd_engine_streams(zamd_engine *e);
int zamd_engine_serialized(zamd_engine *e);

/*
 * General form: inflate a list of segments of one raw deflate stream, one GPU
 * wavefront per segment.  A segment starts at bit in_bit (a block boundary) and
 * may read up to bit in_end_bit; it decodes whole blocks until the final block,
 * the end of its input, or (mode & ZAMD_INF_SYNCSTOP) an empty stored block.
 * With d_out == NULL only lengths are computed; otherwise segment i writes
 * d_out[out_off, out_off + out_len) (out_off ascending, regions disjoint,
 * out_len <= out_cap) and may refer to the win_avail (<= 32768) bytes in front of
 * out_off — which other segments of the same call produce, or d_head_window
 * holds for positions before d_out[0].  On return every segment carries
 * out_len / end_bit of the last complete block, final_seen, and status:
 * 0 = stopped at in_end_bit, 1 = final block done, 2 = stopped at a sync marker,
 * 3 = input ended inside a block, >= 16 = data error (zamd_inflate_message).
 * This is what inflate() of include/zamd_zlib.h is built on (inflate.c:605).
 */
typedef struct zamd_inflate_seg {
    uint64_t in_bit, in_end_bit, out_off, out_cap;
    uint32_t win_avail, mode;
    uint64_t out_len, end_bit;          /* results */
    uint32_t status, final_seen;        /* results */
    uint64_t stop_bit;                  /* ZAMD_INF_BLOCKSTOP: stop at the first block boundary at or behind this bit */
} zamd_inflate_seg;
#define ZAMD_INF_SYNCSTOP 1u
#define ZAMD_INF_EXACT    2u            /* out_len must come out equal to out_cap */
#define ZAMD_INF_BLOCKSTOP 4u           /* status 2 (as for a sync marker) when a block ends at or behind stop_bit */
int zamd_inflate_segments_dev(zamd_engine *e, const void *d_comp, uint64_t comp_bytes, zamd_inflate_seg *segs, uint64_t nseg,
                              const void *d_head_window, void *d_out, uint64_t out_bytes);
/* Block boundaries from a counting pass: the next zamd_inflate_segments_dev call with d_out == NULL (and no other)
 * fills h_marks (host, nseg rows of ZAMD_INF_MARKS entries) with boundaries between the blocks of every segment, as
 * (bits from the segment's in_bit) << 32 | bytes of output in front of the boundary, ascending, unused entries 0: one
 * about every step_bytes of output, and further apart once a segment has reported ZAMD_INF_MARKS - 1 of them (every
 * other one is then dropped).  With them the pass that writes the output can cut a segment into parts (the access
 * points examples/zran.c:195-223 collects with Z_BLOCK).  h_marks == NULL withdraws the request. */
#define ZAMD_INF_MARKS 16u
int zamd_inflate_block_marks(zamd_engine *e, unsigned long long *h_marks, uint32_t step_bytes);
/* Offsets just behind every 00 00 FF FF in d_in[from, n) (the pattern inflateSync looks for,
 * inflate.c:1354-1375): candidate starting points for zamd_inflate_segments_dev.  h_list (host)
 * receives at most cap offsets in no particular order; *count above cap means the list is
 * incomplete. */
int zamd_find_sync_dev(zamd_engine *e, const void *d_in, uint64_t from, uint64_t n, uint64_t *h_list, uint32_t cap, uint32_t *count);
/* Starting points inside a stream that has no sync markers: for every h_targets_bit[i] (a bit position in
 * d_in) the first bit position in [target, target + window_bits) at which the header of a dynamic-Huffman block
 * that is not the last one can be read completely and passes every check inflate makes before the block's first
 * symbol (inflate.c:894-1012: counts in range, a complete code-length code, repeats in range, an end-of-block
 * code, complete literal/length and distance codes), or ~0.  A hit is a guess, to be confirmed by a segment that
 * ends there (ZAMD_INF_BLOCKSTOP).  The reference has no counterpart: inflateSync looks for 00 00 FF FF only. */
int zamd_find_blocks_dev(zamd_engine *e, const void *d_in, uint64_t n, const uint64_t *h_targets_bit, uint32_t ntargets,
                         uint64_t window_bits, uint64_t *h_found_bit);
const char *zamd_inflate_message(uint32_t status);      /* the reference's msg strings, inflate.c:859-1202 */
int zamd_inflate_last_error(zamd_engine *e, int64_t *segment, uint32_t *status);

/* Timing of the last zamd_deflate_dev call, measured with HIP events on the
 * engine's own streams: `which` 0 = chain build, 1 = match search, 2 = parse,
 * 3 = block planning, 4 = layout, 5 = bit emission, 6 = whole call; of the last
 * inflate call: 7 = decode, 8 = window resolution.
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

/* Where every chunk of a stream that several GPUs compress goes: d_offsets[i] = base + the sum of d_lengths[0..i),
 * d_offsets[n] = base + the sum of all (device arrays; d_offsets has n + 1 entries and must not be d_lengths).  A
 * serial writer accumulates these in total_out as it flushes (deflate.c:642-662); ranks that compress shards side by
 * side all-gather their chunks' lengths (d_chunk_off[i + 1] - d_chunk_off[i] of their own calls) and scan them here. */
int zamd_chunk_offsets_dev(zamd_engine *e, const uint64_t *d_lengths, uint64_t n, uint64_t base, uint64_t *d_offsets);

/* The lengths that go into that scan, from a call's own offsets: d_lengths[i] = d_offsets[i + 1] - d_offsets[i] for
 * i < n, and 0 for n <= i < pad_to (ranks contribute equally many entries to an all-gather; the ones that hold fewer
 * chunks pad with chunks of no bytes, which the scan passes over). */
int zamd_chunk_lengths_dev(zamd_engine *e, const uint64_t *d_offsets, uint64_t n, uint64_t *d_lengths, uint64_t pad_to);

/*
 * One deflate call on several GPUs of one node, one process: no counterpart in the reference, whose deflate() is one
 * thread (deflate.c:665-976).  zamd_multi_create makes one engine per entry of `devices`; the input is cut into
 * contiguous ranges of chunks (zamd_multi_partition), each range primed with the 32 KiB of input in front of it, and
 * the ranges' outputs laid end to end are byte for byte the single-device stream.  The devices exchange the chunks'
 * compressed lengths with ncclAllGather (librccl.so, opened at run time; one communicator per device through
 * ncclCommInitAll) and every device scans the table (zamd_chunk_offsets_dev).  Without librccl, with ZAMD_NO_RCCL set,
 * or with a device listed twice (RCCL refuses that), the lengths go through host memory and the host scans them
 * (zamd_multi_offsets_host): zamd_multi_uses_rccl tells which.
 */
typedef struct zamd_multi zamd_multi;
int zamd_multi_create(zamd_multi **out, const int *devices, int n_devices, uint64_t workspace_bytes_per_device);
void zamd_multi_destroy(zamd_multi *m);
int zamd_multi_devices(zamd_multi *m);
zamd_engine *zamd_multi_engine(zamd_multi *m, int index);          /* e.g. for zamd_engine_tune on every device */
int zamd_multi_uses_rccl(zamd_multi *m);
/* Device `index` of n_devices takes chunks [*first, *first + *count) of total_chunks: *per_device = ceil(total /
 * n_devices) chunks each, in device order, the last devices fewer or none.  In a table of n_devices * per_device
 * entries laid out device after device, chunk c is then entry c and the padding lies behind the last chunk. */
int zamd_multi_partition(uint64_t total_chunks, int n_devices, int index, uint64_t *first, uint64_t *count, uint64_t *per_device);
int zamd_multi_offsets_host(const uint64_t *lengths, uint64_t n, uint64_t base, uint64_t *offsets);
/* Ranges that are on their devices already: device i holds n[i] bytes (the partition's share) at d_in[i], with
 * head_len[i] bytes of the input in front of them, and writes its part of the stream to d_out[i].  d_table[i] (on
 * device i, n_devices * per_device + 1 entries) receives the place of EVERY chunk in the whole stream, h_table (host,
 * may be NULL) the same; device i's bytes belong at table[i * per_device]. */
int zamd_deflate_multi_dev(zamd_multi *m, const void *const *d_in, const uint64_t *n, const uint32_t *head_len, uint32_t chunk_size,
                           int level, int strategy, int mem_level, int finish, void *const *d_out, const uint64_t *out_cap,
                           uint64_t *const *d_table, uint64_t *h_table, uint64_t *total_out);
/* The same from host memory to host memory (what deflate() of zamd_zlib.h calls when ZAMD_DEVICES names more than one
 * device): h_dict / dict_len is the input in front of h_in (at most 32768 bytes), h_chunk_off (may be NULL) receives
 * total_chunks + 1 offsets, *adler / *crc (may be NULL) are continued over h_in as adler32() / crc32() would. */
int zamd_deflate_multi(zamd_multi *m, const void *h_in, uint64_t n, const void *h_dict, uint32_t dict_len, uint32_t chunk_size,
                       int level, int strategy, int mem_level, int finish, void *h_out, uint64_t out_cap,
                       uint64_t *h_chunk_off, uint64_t *total_out, uint32_t *adler, uint32_t *crc);

/* `fn(arg)` is called from inside every following deflate call of this engine, on the calling thread, when the call's last
 * kernel has been queued and before the call starts to wait for the GPU (NULL: no call).  A copy between pageable host
 * memory and the device holds a lock of the GPU runtime for its whole length, and kernel launches of other threads
 * wait for it: a caller that overlaps the transfers of one batch with the kernels of another starts the transfers
 * from here.  No counterpart in the reference. */
int zamd_engine_on_launched(zamd_engine *e, void (*fn)(void *), void *arg);

/* Thin device-memory helpers so that a C or ctypes caller needs no HIP headers
 * (hipMalloc / hipFree / hipMemcpy on the engine's device). */
int zamd_dev_alloc(zamd_engine *e, uint64_t bytes, void **d_ptr);
int zamd_dev_free(zamd_engine *e, void *d_ptr);
int zamd_dev_upload(zamd_engine *e, void *d_dst, const void *h_src, uint64_t bytes);
int zamd_dev_download(zamd_engine *e, void *h_dst, const void *d_src, uint64_t bytes);
/* The same two transfers through pinned staging buffers of the engine (2 x 32 MiB per direction, allocated at first
 * use): for transfers that run beside the kernels of a deflate call, where the runtime's own path for pageable memory
 * slows down several times.  One staged upload and one staged download may run at a time, from different threads. */
int zamd_dev_upload_staged(zamd_engine *e, void *d_dst, const void *h_src, uint64_t bytes);
int zamd_dev_download_staged(zamd_engine *e, void *h_dst, const void *d_src, uint64_t bytes);
int zamd_dev_memset(zamd_engine *e, void *d_dst, int value, uint64_t bytes);

const char *zamd_version(void);

#ifdef __cplusplus
}
#endif
#endif
