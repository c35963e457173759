/*
 * zamd_zlib.h — the zlib 1.2.8 stream API as exported by libzamd_z.so.
 *
 * Declarations are binary compatible with the reference's zlib.h (types
 * zconf.h:368-490, z_stream zlib.h:85-104, gz_header zlib.h:109-129, constants
 * zlib.h:161-209): a program compiled against the reference header can be linked
 * against this library instead of libz.  Only the entry points listed here are
 * provided; every one names the reference function it replaces.
 *
 * Compression runs on the GPU through the engine in zamd.h: deflate() buffers
 * input on the host and compresses it as dictionary-primed chunks (default
 * 1 MiB, each ending on a Z_SYNC_FLUSH marker), so the stream is valid zlib /
 * gzip / raw deflate and every chunk equals the reference's output for that
 * chunk, but a multi-chunk stream is not byte-identical to a single reference
 * stream.  Inputs of at most one chunk compress to exactly the reference's bytes.
 */
#ifndef ZAMD_ZLIB_H
#define ZAMD_ZLIB_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZLIB_VERSION "1.2.8"
#define ZLIB_VERNUM 0x1280

typedef unsigned char Byte;
typedef unsigned int uInt;
typedef unsigned long uLong;
typedef Byte Bytef;
typedef void *voidpf;
typedef void *voidp;
typedef uLong uLongf;
typedef long z_off_t;
typedef long long z_off64_t;
typedef unsigned int z_crc_t;

typedef voidpf (*alloc_func)(voidpf opaque, uInt items, uInt size);
typedef void (*free_func)(voidpf opaque, voidpf address);
struct internal_state;

typedef struct z_stream_s {            /* zlib.h:85-104 */
    const Bytef *next_in; uInt avail_in; uLong total_in;
    Bytef *next_out; uInt avail_out; uLong total_out;
    const char *msg;
    struct internal_state *state;
    alloc_func zalloc; free_func zfree; voidpf opaque;
    int data_type; uLong adler; uLong reserved;
} z_stream;
typedef z_stream *z_streamp;

typedef struct gz_header_s {           /* zlib.h:109-129 */
    int text; uLong time; int xflags; int os;
    Bytef *extra; uInt extra_len; uInt extra_max;
    Bytef *name; uInt name_max;
    Bytef *comment; uInt comm_max;
    int hcrc; int done;
} gz_header;
typedef gz_header *gz_headerp;

/* zlib.h:161-209 */
#define Z_NO_FLUSH 0
#define Z_PARTIAL_FLUSH 1
#define Z_SYNC_FLUSH 2
#define Z_FULL_FLUSH 3
#define Z_FINISH 4
#define Z_BLOCK 5
#define Z_TREES 6
#define Z_OK 0
#define Z_STREAM_END 1
#define Z_NEED_DICT 2
#define Z_ERRNO (-1)
#define Z_STREAM_ERROR (-2)
#define Z_DATA_ERROR (-3)
#define Z_MEM_ERROR (-4)
#define Z_BUF_ERROR (-5)
#define Z_VERSION_ERROR (-6)
#define Z_NO_COMPRESSION 0
#define Z_BEST_SPEED 1
#define Z_BEST_COMPRESSION 9
#define Z_DEFAULT_COMPRESSION (-1)
#define Z_FILTERED 1
#define Z_HUFFMAN_ONLY 2
#define Z_RLE 3
#define Z_FIXED 4
#define Z_DEFAULT_STRATEGY 0
#define Z_BINARY 0
#define Z_TEXT 1
#define Z_UNKNOWN 2
#define Z_DEFLATED 8
#define Z_NULL 0

const char *zlibVersion(void);                                                  /* zutil.c:30 */
uLong zlibCompileFlags(void);                                                   /* zutil.c:35 */
const char *zError(int err);                                                    /* zutil.c:136 */

/* deflate side: deflate.c:201-1067 */
int deflateInit_(z_streamp strm, int level, const char *version, int stream_size);                  /* :201 */
int deflateInit2_(z_streamp strm, int level, int method, int windowBits, int memLevel,
                  int strategy, const char *version, int stream_size);                              /* :213 */
int deflate(z_streamp strm, int flush);                                                             /* :665 */
int deflateEnd(z_streamp strm);                                                                     /* :979 */
int deflateReset(z_streamp strm);                                                                   /* :427 */
int deflateResetKeep(z_streamp strm);                                                               /* :392 */
uLong deflateBound(z_streamp strm, uLong sourceLen);                                                /* :566 */
int deflateParams(z_streamp strm, int level, int strategy);                                         /* :490 */
int deflatePending(z_streamp strm, unsigned *pending, int *bits);                                   /* :450 */
int deflateSetHeader(z_streamp strm, gz_headerp head);                                              /* :439 */
int deflateCopy(z_streamp dest, z_streamp source);                                                  /* :1014 */
/* The dictionary primes the first chunk.  The engine takes it in multiples of 16
 * bytes: for other lengths the oldest dictLength % 16 bytes are not matched against
 * (the stream stays valid for the full dictionary, dictid covers all of it). */
int deflateSetDictionary(z_streamp strm, const Bytef *dictionary, uInt dictLength);                 /* :323 */
/* Holds for all input compressed after the call (input still buffered included) until
 * deflateReset or a level change.  max_chain must be >= 1, and <= 64 at levels 1-3. */
int deflateTune(z_streamp strm, int good_length, int max_lazy, int nice_length, int max_chain);     /* :531 */
int deflatePrime(z_streamp strm, int bits, int value);              /* :464; bits 0..32 */

/* inflate side: inflate.c:104-1512 */
int inflateInit_(z_streamp strm, const char *version, int stream_size);                             /* :222 */
int inflateInit2_(z_streamp strm, int windowBits, const char *version, int stream_size);            /* :180 */
int inflate(z_streamp strm, int flush);                                                             /* :605 */
int inflateEnd(z_streamp strm);                                                                     /* :1254 */
int inflateReset(z_streamp strm);                                                                   /* :129 */
int inflateReset2(z_streamp strm, int windowBits);                                                  /* :141 */
int inflateResetKeep(z_streamp strm);                                                               /* :104 */
int inflateSetDictionary(z_streamp strm, const Bytef *dictionary, uInt dictLength);                 /* :1291 */
int inflateGetDictionary(z_streamp strm, Bytef *dictionary, uInt *dictLength);                      /* :1268 */
int inflateGetHeader(z_streamp strm, gz_headerp head);                                              /* :1326 */
int inflateSync(z_streamp strm);                                                                    /* :1377 */
int inflateSyncPoint(z_streamp strm);                                                               /* :1428 */
int inflateCopy(z_streamp dest, z_streamp source);                                                  /* :1438 */
int inflateUndermine(z_streamp strm, int subvert);                                                  /* :1485 */
long inflateMark(z_streamp strm);                                                                   /* :1502 */
/* On deflate data (raw streams, or once the wrapper header is through) before any input of
 * it has been handed over; Z_STREAM_ERROR otherwise. */
int inflatePrime(z_streamp strm, int bits, int value);                                              /* :230 */

/* call-back interface for raw deflate data: infback.c:28-640.  Unused input is reported in next_in/avail_in
 * as far as it lies in the last buffer in() delivered. */
typedef unsigned (*in_func)(void *, const unsigned char **);
typedef int (*out_func)(void *, unsigned char *, unsigned);
int inflateBackInit_(z_streamp strm, int windowBits, unsigned char *window, const char *version, int stream_size);    /* :28 */
int inflateBack(z_streamp strm, in_func in, void *in_desc, out_func out, void *out_desc);                             /* :250 */
int inflateBackEnd(z_streamp strm);                                                                                  /* :631 */
#define inflateBackInit(strm, windowBits, window) \
    inflateBackInit_((strm), (windowBits), (window), ZLIB_VERSION, (int)sizeof(z_stream))

/* one-shot utilities: compress.c:22-80, uncompr.c:24-59 */
int compress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen);
int compress2(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen, int level);
uLong compressBound(uLong sourceLen);
int uncompress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen);

/* checksums: adler32.c:65-179, crc32.c:204-425 */
uLong adler32(uLong adler, const Bytef *buf, uInt len);
uLong crc32(uLong crc, const Bytef *buf, uInt len);
uLong adler32_combine(uLong adler1, uLong adler2, z_off_t len2);
uLong adler32_combine64(uLong adler1, uLong adler2, z_off64_t len2);
uLong crc32_combine(uLong crc1, uLong crc2, z_off_t len2);
uLong crc32_combine64(uLong crc1, uLong crc2, z_off64_t len2);
const z_crc_t *get_crc_table(void);

/* gz* file layer: gzlib.c, gzread.c, gzwrite.c, gzclose.c (declarations zlib.h:1224-1552, 1670-1728).
 * Same behaviour and return values; the buffers default to 32 MiB (gzbuffer() or the
 * environment variable ZAMD_GZ_BUF_MIB change that) because every call into the
 * codec that makes progress is a GPU round trip. */
#include <stdarg.h>
typedef struct gzFile_s *gzFile;
struct gzFile_s { unsigned have; unsigned char *next; z_off64_t pos; };                             /* zlib.h:1670-1674 */
gzFile gzopen(const char *path, const char *mode);                                                  /* gzlib.c:268 */
gzFile gzopen64(const char *path, const char *mode);                                                /* gzlib.c:276 */
gzFile gzdopen(int fd, const char *mode);                                                           /* gzlib.c:284 */
int gzbuffer(gzFile file, unsigned size);                                                           /* gzlib.c:314 */
int gzsetparams(gzFile file, int level, int strategy);                                              /* gzwrite.c:513 */
int gzread(gzFile file, voidp buf, unsigned len);                                                   /* gzread.c:288 */
int gzwrite(gzFile file, const void *buf, unsigned len);                                            /* gzwrite.c:165 */
int gzprintf(gzFile file, const char *format, ...);                                                 /* gzwrite.c:375 */
int gzvprintf(gzFile file, const char *format, va_list va);                                         /* gzwrite.c:311 */
int gzputs(gzFile file, const char *s);                                                             /* gzwrite.c:294 */
char *gzgets(gzFile file, char *buf, int len);                                                      /* gzread.c:483 */
int gzputc(gzFile file, int c);                                                                     /* gzwrite.c:246 */
int gzgetc(gzFile file);                                                                            /* gzread.c:387 */
int gzgetc_(gzFile file);                                                                           /* gzread.c:416 */
int gzungetc(int c, gzFile file);                                                                   /* gzread.c:423 */
int gzflush(gzFile file, int flush);                                                                /* gzwrite.c:478 */
z_off_t gzseek(gzFile file, z_off_t offset, int whence);                                            /* gzlib.c:439 */
z_off64_t gzseek64(gzFile file, z_off64_t offset, int whence);                                      /* gzlib.c:362 */
int gzrewind(gzFile file);                                                                          /* gzlib.c:339 */
z_off_t gztell(gzFile file);                                                                        /* gzlib.c:468 */
z_off64_t gztell64(gzFile file);                                                                    /* gzlib.c:451 */
z_off_t gzoffset(gzFile file);                                                                      /* gzlib.c:501 */
z_off64_t gzoffset64(gzFile file);                                                                  /* gzlib.c:478 */
int gzeof(gzFile file);                                                                             /* gzlib.c:511 */
int gzdirect(gzFile file);                                                                          /* gzread.c:547 */
int gzclose(gzFile file);                                                                           /* gzclose.c:11 */
int gzclose_r(gzFile file);                                                                         /* gzread.c:567 */
int gzclose_w(gzFile file);                                                                         /* gzwrite.c:555 */
const char *gzerror(gzFile file, int *errnum);                                                      /* gzlib.c:528 */
void gzclearerr(gzFile file);                                                                       /* gzlib.c:549 */
#define gzgetc(g) ((g)->have ? ((g)->have--, (g)->pos++, *((g)->next)++) : gzgetc(g))               /* zlib.h:1684-1686 */

/* ---- chunk index: this library only, no counterpart in the reference (examples/zran.c:67-72 builds
 * comparable access points by inflating a file once).  deflate() cuts its input into chunks that each end on a
 * byte boundary; the index lists, per chunk in stream order, { compressed bytes, inflated bytes }. */
int zamd_deflate_index(z_streamp strm, int enable);         /* after deflateInit/Reset, before the first input */
int zamd_deflate_index_get(z_streamp strm, const unsigned **pairs, unsigned *n_chunks);
                                                             /* after Z_STREAM_END, before deflateReset; Z_DATA_ERROR: the
                                                                stream has no byte-aligned chunk grid (deflatePrime) */
int zamd_inflate_index_hint(z_streamp strm, const unsigned *pairs, unsigned n_chunks);
                                                             /* after inflateInit/Reset: the index of the deflate data that
                                                                follows; inflate() then places whole chunks in one pass.  A
                                                                hint that does not fit the data is dropped, not trusted */
/* Block boundaries inside the chunks, recorded with the index: per_chunk records for every chunk of the index, each
 * (bit offset of a block from the chunk's first byte) << 32 | offset of the block's first byte in the chunk's data,
 * ascending, zeros behind the last one.  With them inflate() puts several wavefronts on a chunk, and a round of chunks
 * is decoded, checked and copied to the caller's buffer in overlapped groups.  Hints like the index itself. */
int zamd_deflate_index_splits(z_streamp strm, const unsigned long long **recs, unsigned *per_chunk);
                                                             /* with zamd_deflate_index_get */
int zamd_inflate_index_hint_splits(z_streamp strm, const unsigned long long *recs, unsigned per_chunk);
                                                             /* directly after zamd_inflate_index_hint */
/* adler32() and crc32() (zlib.h:1569-1617) have no way to report a failure, and this library can fail where the
 * reference cannot (no GPU, no device memory): such a call writes a line to stderr, hands its running value back
 * unchanged and leaves its error code here.  Returns 0 or that code, and clears it.  No counterpart in the reference. */
int zamd_z_last_error(void);
/* The same index kept in a gzip file: gzopen(path, "wbI") (or ZAMD_GZ_INDEX=1) appends it as empty gzip members
 * that every gzip reader skips; gzopen for reading finds and uses it (ZAMD_GZ_INDEX=0: ignores it). */
unsigned zamd_deflate_chunk_bytes(z_streamp strm);          /* the chunk size of this stream's grid (input bytes) */
/* What deflate() collects before the engine runs: input handed over in pieces of exactly this size, with nothing
 * buffered in front, is compressed where it lies.  No counterpart in the reference (deflate.c:1409-1413 sizes the
 * reference's own unit of work, the window). */
unsigned long long zamd_deflate_batch_bytes(z_streamp strm);
/* Sets up the GPU engine now, on the calling thread, instead of at the first call that compresses or inflates (Z_OK, or
 * what that call would have reported).  For callers that want the seconds of set-up outside their own timing, and for
 * the gz layer, whose helper threads must not be the ones that initialise the GPU runtime (what that runtime registers
 * for the end of the process has to be registered before the layer's own hook).  No counterpart in the reference. */
int zamd_z_engine_ready(void);
unsigned long long zamd_inflate_hinted_chunks(z_streamp strm);   /* chunks inflate() has placed by their index entries so far */
/* Random access (the reference keeps such access points in memory while it inflates a file once, examples/zran.c:67-72,
 * 249-349): a file written with a 'W' in the mode string (or ZAMD_GZ_WINDOWS=n in the environment: every n-th chunk) also
 * stores, in front of the index, the 32 KiB of data in front of every n-th chunk, each in an empty gzip member of its
 * own.  gzseek() on such a file reads the one window it needs and inflates from that chunk on; without windows it
 * inflates from the start of the file, as the reference's does (gzlib.c:362-436).  The data's CRC-32 is not checked on a
 * pass that began with such a seek.  Cost: 32 KiB per window, 3 % of the data at the default chunk size with n = 1. */
typedef struct zamd_gz_index {
    unsigned long long member_offset; unsigned n_chunks; unsigned *pairs;
    unsigned long long window_offset; unsigned window_every, n_windows;      /* file offset of the first window member; 0, 0, 0: none */
    unsigned long long *splits; unsigned splits_per_chunk;                   /* split records of the chunks (zamd_deflate_index_splits), or NULL, 0 */
} zamd_gz_index;
/* chunks placed by index entries since the file was opened, and seeks that started from a stored window */
int zamd_gz_seek_stats(gzFile file, unsigned long long *chunks_decoded, unsigned long long *window_seeks);
int zamd_gz_index_load(int fd, zamd_gz_index *out);          /* 0 = found, -1 = the file carries no index */
void zamd_gz_index_free(zamd_gz_index *ix);

#define deflateInit(strm, level) deflateInit_((strm), (level), ZLIB_VERSION, (int)sizeof(z_stream))
#define deflateInit2(strm, level, method, windowBits, memLevel, strategy) \
    deflateInit2_((strm), (level), (method), (windowBits), (memLevel), (strategy), ZLIB_VERSION, (int)sizeof(z_stream))
#define inflateInit(strm) inflateInit_((strm), ZLIB_VERSION, (int)sizeof(z_stream))
#define inflateInit2(strm, windowBits) inflateInit2_((strm), (windowBits), ZLIB_VERSION, (int)sizeof(z_stream))

#ifdef __cplusplus
}
#endif
#endif
