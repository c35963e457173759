// zlib 1.2.8 stream API (include/zamd_zlib.h) on top of the GPU engine (include/zamd.h).
//
// Host code only: argument checking, wrapper headers/trailers, buffering between
// the caller's z_stream cursors and the engine's batch calls.  All byte work
// (matching, Huffman coding, checksums, decoding) happens in libzamd.so on the
// GPU; when no GPU engine can be created every stream call fails with
// Z_STREAM_ERROR / Z_MEM_ERROR instead of falling back to the CPU.
//
// Reference behaviour mirrored here: deflate.c:201-320 (init), :392-436 (reset),
// :665-976 (deflate), :979-1007 (end), compress.c, uncompr.c, inflate.c:104-249,
// :605-1252 (wrapper handling of inflate), zutil.c:17-40, 136.
#include "../../include/zamd.h"
#include "../../include/zamd_zlib.h"

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

namespace {

const char *const k_errmsg[10] = {   // zutil.c:17-27
    "need dictionary", "stream end", "", "file error", "stream error",
    "data error", "insufficient memory", "buffer error", "incompatible version", ""};
inline const char *err_msg(int err) { return k_errmsg[Z_NEED_DICT - err]; }

// ------------------------------------------------------------------ engine singleton
struct Gpu {
    std::mutex mu;           // serialises every GPU call made through this API
    zamd_engine *eng = nullptr;
    int rc = Z_OK;
    bool tried = false;
    uint8_t *d_in = nullptr;  uint64_t in_cap = 0;    // staging: [32 KiB history][batch]
    uint8_t *d_out = nullptr; uint64_t out_cap = 0;
    uint64_t *d_off = nullptr; uint64_t off_cap = 0;
};
Gpu g_gpu;

// Caller holds g_gpu.mu.
zamd_engine *engine_locked() {
    if (!g_gpu.tried) {
        g_gpu.tried = true;
        const char *dev = getenv("ZAMD_DEVICE");
        g_gpu.rc = zamd_engine_create(&g_gpu.eng, dev ? atoi(dev) : 0, 0);
        if (g_gpu.rc != 0) g_gpu.eng = nullptr;
    }
    return g_gpu.eng;
}

int grow(zamd_engine *e, uint8_t **p, uint64_t *cap, uint64_t need) {
    if (*cap >= need) return Z_OK;
    if (*p) zamd_dev_free(e, *p);
    *p = nullptr; *cap = 0;
    uint64_t want = need + (need >> 2) + 4096;
    void *q = nullptr;
    if (zamd_dev_alloc(e, want, &q) != 0) {
        if (zamd_dev_alloc(e, need, &q) != 0) return Z_MEM_ERROR;
        want = need;
    }
    *p = (uint8_t *)q; *cap = want;
    return Z_OK;
}

uint64_t env_u64(const char *name, uint64_t dflt) {
    const char *v = getenv(name);
    if (!v || !*v) return dflt;
    return strtoull(v, nullptr, 10);
}

voidpf default_alloc(voidpf, uInt items, uInt size) { return malloc((size_t)items * size); }   // zutil.c:304
void default_free(voidpf, voidpf p) { free(p); }                                               // zutil.c:314

// CRC-32 of the few gzip header bytes covered by FHCRC (deflate.c:728-730, 853-860);
// payload checksums are computed on the GPU.
uint32_t header_crc(uint32_t crc, const uint8_t *p, size_t n) {
    crc = ~crc;
    for (size_t i = 0; i < n; i++) {
        crc ^= p[i];
        for (int k = 0; k < 8; k++) crc = (crc >> 1) ^ (0xedb88320u & (0u - (crc & 1)));
    }
    return ~crc;
}

const uint32_t DEFLATE_MAGIC = 0x5a414d44u;   // distinguishes our states from inflate states
const uint32_t INFLATE_MAGIC = 0x5a414d49u;
const uint32_t HIST = 32768;

enum { ST_INIT = 42, ST_BUSY = 113, ST_FINISH = 666 };   // deflate.h:51-57 values

struct DState {
    uint32_t magic;
    int status, wrap, level, strategy, mem_level, w_bits, last_flush;
    uint32_t chunk_size;
    uint64_t batch_bytes;
    gz_headerp gzhead;
    bool trailer_done;
    uint32_t hist_len;                 // bytes of history kept in `hist` (== HIST once warm)
    uint64_t since_reset;              // input bytes compressed since the last history reset
    std::vector<uint8_t> hist;         // last 32 KiB of compressed input
    std::vector<uint8_t> in;           // input accepted from the caller, not yet compressed
    std::vector<uint8_t> out;          // produced output not yet handed to the caller
    size_t out_pos;
};

inline DState *dstate(z_streamp strm) {
    if (strm == Z_NULL || strm->state == Z_NULL) return nullptr;
    DState *s = (DState *)strm->state;
    return s->magic == DEFLATE_MAGIC ? s : nullptr;
}

inline int rank_of(int f) { return (f << 1) - (f > 4 ? 9 : 0); }   // deflate.c:153

void put_be32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
void put_le32(std::vector<uint8_t> &v, uint32_t x) {
    v.push_back(x); v.push_back(x >> 8); v.push_back(x >> 16); v.push_back(x >> 24);
}

void flush_pending(z_streamp strm, DState *s) {    // deflate.c:642-662
    size_t len = s->out.size() - s->out_pos;
    if (len > strm->avail_out) len = strm->avail_out;
    if (len == 0) return;
    memcpy(strm->next_out, s->out.data() + s->out_pos, len);
    strm->next_out += len; strm->avail_out -= (uInt)len; strm->total_out += len;
    s->out_pos += len;
    if (s->out_pos == s->out.size()) { s->out.clear(); s->out_pos = 0; }
}

void write_header(z_streamp strm, DState *s) {     // deflate.c:690-867
    std::vector<uint8_t> &o = s->out;
    int xfl = s->level == 9 ? 2 : ((s->strategy >= Z_HUFFMAN_ONLY || s->level < 2) ? 4 : 0);
    if (s->wrap == 2) {
        size_t beg = o.size();
        o.push_back(31); o.push_back(139); o.push_back(8);
        gz_headerp h = s->gzhead;
        if (h == Z_NULL) {
            for (int i = 0; i < 5; i++) o.push_back(0);
            o.push_back((uint8_t)xfl);
            o.push_back(3);                              // OS_CODE for Unix, zutil.h:179-181
        } else {
            o.push_back((h->text ? 1 : 0) + (h->hcrc ? 2 : 0) + (h->extra == Z_NULL ? 0 : 4) +
                        (h->name == Z_NULL ? 0 : 8) + (h->comment == Z_NULL ? 0 : 16));
            put_le32(o, (uint32_t)h->time);
            o.push_back((uint8_t)xfl);
            o.push_back(h->os & 0xff);
            if (h->extra != Z_NULL) {
                o.push_back(h->extra_len & 0xff); o.push_back((h->extra_len >> 8) & 0xff);
                o.insert(o.end(), h->extra, h->extra + (h->extra_len & 0xffff));
            }
            if (h->name != Z_NULL) { const Bytef *p = h->name; do o.push_back(*p); while (*p++); }
            if (h->comment != Z_NULL) { const Bytef *p = h->comment; do o.push_back(*p); while (*p++); }
            if (h->hcrc) {
                uint32_t c = header_crc(0, o.data() + beg, o.size() - beg);
                o.push_back(c & 0xff); o.push_back((c >> 8) & 0xff);
            }
        }
        strm->adler = 0;                                  // crc32(0, NULL, 0)
    } else {
        uInt header = (Z_DEFLATED + ((s->w_bits - 8) << 4)) << 8;
        uInt level_flags;
        if (s->strategy >= Z_HUFFMAN_ONLY || s->level < 2) level_flags = 0;
        else if (s->level < 6) level_flags = 1;
        else if (s->level == 6) level_flags = 2;
        else level_flags = 3;
        header |= level_flags << 6;
        header += 31 - (header % 31);
        o.push_back(header >> 8); o.push_back(header & 0xff);
        strm->adler = 1;                                  // adler32(0, NULL, 0)
    }
    s->status = ST_BUSY;
}

// Compress `n` bytes at `src` (host) as chunks primed with the stream's history;
// append the result to s->out and fold the bytes into strm->adler.
int run_batch(z_streamp strm, DState *s, const uint8_t *src, uint64_t n, int finish) {
    std::lock_guard<std::mutex> lk(g_gpu.mu);
    zamd_engine *e = engine_locked();
    if (!e) return g_gpu.rc == ZAMD_MEM_ERROR ? Z_MEM_ERROR : Z_STREAM_ERROR;
    uint32_t head = s->hist_len == HIST ? HIST : 0;
    uint64_t bound = zamd_deflate_bound(n, s->chunk_size, s->mem_level);
    if (grow(e, &g_gpu.d_in, &g_gpu.in_cap, HIST + n + 64) != Z_OK) return Z_MEM_ERROR;
    if (grow(e, &g_gpu.d_out, &g_gpu.out_cap, bound + 64) != Z_OK) return Z_MEM_ERROR;
    int rc = 0;
    if (head) rc = zamd_dev_upload(e, g_gpu.d_in, s->hist.data(), HIST);
    if (rc == 0 && n) rc = zamd_dev_upload(e, g_gpu.d_in + HIST, src, n);
    uint64_t total = 0;
    if (rc == 0)
        rc = zamd_deflate_shard_dev(e, g_gpu.d_in + HIST, n, head, s->chunk_size, s->level, s->strategy,
                                    s->mem_level, finish, g_gpu.d_out, bound, nullptr, &total);
    if (rc == 0 && s->wrap && n) {
        uint32_t v = (uint32_t)strm->adler;
        rc = s->wrap == 2 ? zamd_crc32_dev(e, v, g_gpu.d_in + HIST, n, &v)
                          : zamd_adler32_dev(e, v, g_gpu.d_in + HIST, n, &v);
        strm->adler = v;
    }
    if (rc == 0 && total) {
        size_t at = s->out.size();
        s->out.resize(at + total);
        rc = zamd_dev_download(e, s->out.data() + at, g_gpu.d_out, total);
    }
    if (rc != 0) return rc == ZAMD_MEM_ERROR ? Z_MEM_ERROR : (rc == ZAMD_BUF_ERROR ? Z_BUF_ERROR : Z_STREAM_ERROR);
    // history for the next batch = last 32 KiB of everything compressed since the last reset
    s->since_reset += n;
    if (n >= HIST) {
        s->hist.assign(src + n - HIST, src + n);
        s->hist_len = HIST;
    } else if (n) {
        s->hist.insert(s->hist.end(), src, src + n);
        if (s->hist.size() > HIST) s->hist.erase(s->hist.begin(), s->hist.end() - HIST);
        s->hist_len = (uint32_t)s->hist.size();
    }
    return Z_OK;
}

void reset_history(DState *s) { s->hist.clear(); s->hist_len = 0; s->since_reset = 0; }

int finish_batch(z_streamp strm, DState *s, int flush) {
    int rc = run_batch(strm, s, s->in.data(), s->in.size(), flush == Z_FINISH);
    if (rc != Z_OK) return rc;
    s->in.clear();
    if (flush == Z_FULL_FLUSH) reset_history(s);          // deflate.c:930-937: forget the history
    if (flush == Z_FINISH) {
        s->status = ST_FINISH;
        if (s->wrap == 2) { put_le32(s->out, (uint32_t)strm->adler); put_le32(s->out, (uint32_t)strm->total_in); }
        else if (s->wrap == 1) put_be32(s->out, (uint32_t)strm->adler);
        s->trailer_done = true;
    }
    return Z_OK;
}

}  // namespace

extern "C" {

const char *zlibVersion(void) { return ZLIB_VERSION; }

uLong zlibCompileFlags(void) {   // zutil.c:35-113: type sizes only, no optional-feature bits
    uLong flags = 0;
    auto code = [](size_t sz) -> uLong { return sz == 2 ? 0 : sz == 4 ? 1 : sz == 8 ? 2 : 3; };
    flags += code(sizeof(uInt));
    flags += code(sizeof(uLong)) << 2;
    flags += code(sizeof(voidpf)) << 4;
    flags += code(sizeof(z_off_t)) << 6;
    return flags;
}

const char *zError(int err) { return (err < -6 || err > 2) ? "" : err_msg(err); }

int deflateInit_(z_streamp strm, int level, const char *version, int stream_size) {
    return deflateInit2_(strm, level, Z_DEFLATED, 15, 8, Z_DEFAULT_STRATEGY, version, stream_size);
}

int deflateInit2_(z_streamp strm, int level, int method, int windowBits, int memLevel, int strategy,
                  const char *version, int stream_size) {
    if (version == Z_NULL || version[0] != ZLIB_VERSION[0] || stream_size != (int)sizeof(z_stream))
        return Z_VERSION_ERROR;
    if (strm == Z_NULL) return Z_STREAM_ERROR;
    strm->msg = Z_NULL;
    if (strm->zalloc == (alloc_func)0) { strm->zalloc = default_alloc; strm->opaque = (voidpf)0; }
    if (strm->zfree == (free_func)0) strm->zfree = default_free;
    if (level == Z_DEFAULT_COMPRESSION) level = 6;
    int wrap = 1;
    if (windowBits < 0) { wrap = 0; windowBits = -windowBits; }
    else if (windowBits > 15) { wrap = 2; windowBits -= 16; }
    if (memLevel < 1 || memLevel > 9 || method != Z_DEFLATED || windowBits < 8 || windowBits > 15 ||
        level < 0 || level > 9 || strategy < 0 || strategy > Z_FIXED)
        return Z_STREAM_ERROR;
    // The engine implements the 32 KiB window only, and no stored-only level.
    if (windowBits != 15 || level == 0) { strm->msg = "parameter not supported by the GPU engine"; return Z_STREAM_ERROR; }
    void *mem = strm->zalloc(strm->opaque, 1, (uInt)sizeof(DState));
    if (mem == Z_NULL) return Z_MEM_ERROR;
    DState *s = new (mem) DState();
    s->magic = DEFLATE_MAGIC;
    s->wrap = wrap; s->level = level; s->strategy = strategy; s->mem_level = memLevel; s->w_bits = windowBits;
    s->gzhead = Z_NULL;
    uint64_t chunk = env_u64("ZAMD_Z_CHUNK", 1u << 20);
    if (chunk < 32768 || chunk > (64u << 20) || (chunk & 16383)) chunk = 1u << 20;
    s->chunk_size = (uint32_t)chunk;
    uint64_t batch = env_u64("ZAMD_Z_BATCH_MIB", 128) << 20;
    if (batch < chunk) batch = chunk;
    s->batch_bytes = batch / chunk * chunk;
    strm->state = (struct internal_state *)s;
    return deflateReset(strm);
}

int deflateResetKeep(z_streamp strm) {
    DState *s = dstate(strm);
    if (!s || strm->zalloc == (alloc_func)0 || strm->zfree == (free_func)0) return Z_STREAM_ERROR;
    strm->total_in = strm->total_out = 0;
    strm->msg = Z_NULL;
    strm->data_type = Z_UNKNOWN;
    s->out.clear(); s->out_pos = 0;
    s->status = s->wrap ? ST_INIT : ST_BUSY;
    strm->adler = s->wrap == 2 ? 0 : 1;
    s->last_flush = Z_NO_FLUSH;
    s->trailer_done = false;
    return Z_OK;
}

int deflateReset(z_streamp strm) {
    int ret = deflateResetKeep(strm);
    if (ret == Z_OK) {
        DState *s = dstate(strm);
        s->in.clear();
        reset_history(s);
    }
    return ret;
}

int deflateSetHeader(z_streamp strm, gz_headerp head) {
    DState *s = dstate(strm);
    if (!s || s->wrap != 2) return Z_STREAM_ERROR;
    s->gzhead = head;
    return Z_OK;
}

int deflatePending(z_streamp strm, unsigned *pending, int *bits) {
    DState *s = dstate(strm);
    if (!s) return Z_STREAM_ERROR;
    if (pending != Z_NULL) *pending = (unsigned)(s->out.size() - s->out_pos);
    if (bits != Z_NULL) *bits = 0;     // output is handed over in whole bytes
    return Z_OK;
}

int deflateSetDictionary(z_streamp strm, const Bytef *, uInt) {
    if (strm) strm->msg = "preset dictionaries are not supported by the GPU engine";
    return Z_STREAM_ERROR;
}
int deflateTune(z_streamp strm, int, int, int, int) {
    if (strm) strm->msg = "deflateTune is not supported by the GPU engine";
    return Z_STREAM_ERROR;
}
int deflatePrime(z_streamp strm, int, int) {
    if (strm) strm->msg = "deflatePrime is not supported by the GPU engine";
    return Z_STREAM_ERROR;
}

int deflateParams(z_streamp strm, int level, int strategy) {
    DState *s = dstate(strm);
    if (!s) return Z_STREAM_ERROR;
    if (level == Z_DEFAULT_COMPRESSION) level = 6;
    if (level < 1 || level > 9 || strategy < 0 || strategy > Z_FIXED) return Z_STREAM_ERROR;
    int err = Z_OK;
    if ((level != s->level || strategy != s->strategy) && !s->in.empty()) {
        // Input accepted so far is compressed with the old parameters, as the
        // reference does with deflate(Z_BLOCK) (deflate.c:512-518); the output
        // waits in the pending buffer.
        if (s->status == ST_INIT) write_header(strm, s);
        err = finish_batch(strm, s, Z_BLOCK);
        if (err != Z_OK) return err;
        if (strm->next_out != Z_NULL) flush_pending(strm, s);
    }
    s->level = level;
    s->strategy = strategy;
    return err;
}

uLong deflateBound(z_streamp strm, uLong sourceLen) {
    DState *s = dstate(strm);
    uint32_t chunk = s ? s->chunk_size : (1u << 20);
    // one engine call per batch, each possibly ending in a sync marker, plus flush markers
    uLong body = (uLong)zamd_deflate_bound(sourceLen, chunk, s ? s->mem_level : 8) + 16;
    uLong wraplen = 6;
    if (s) {
        if (s->wrap == 0) wraplen = 0;
        else if (s->wrap == 2) {
            wraplen = 18;
            gz_headerp h = s->gzhead;
            if (h != Z_NULL) {
                if (h->extra != Z_NULL) wraplen += 2 + h->extra_len;
                if (h->name != Z_NULL) wraplen += strlen((const char *)h->name) + 1;
                if (h->comment != Z_NULL) wraplen += strlen((const char *)h->comment) + 1;
                if (h->hcrc) wraplen += 2;
            }
        }
    }
    return body + wraplen;
}

int deflate(z_streamp strm, int flush) {
    DState *s = dstate(strm);
    if (!s || flush > Z_BLOCK || flush < 0) return Z_STREAM_ERROR;
    if (strm->next_out == Z_NULL || (strm->next_in == Z_NULL && strm->avail_in != 0) ||
        (s->status == ST_FINISH && flush != Z_FINISH)) {
        strm->msg = err_msg(Z_STREAM_ERROR);
        return Z_STREAM_ERROR;
    }
    if (strm->avail_out == 0) { strm->msg = err_msg(Z_BUF_ERROR); return Z_BUF_ERROR; }
    int old_flush = s->last_flush;
    s->last_flush = flush;

    if (s->status == ST_INIT) write_header(strm, s);

    if (s->out.size() != s->out_pos) {                   // deflate.c:869-881
        flush_pending(strm, s);
        if (strm->avail_out == 0) { s->last_flush = -1; return Z_OK; }
    } else if (strm->avail_in == 0 && rank_of(flush) <= rank_of(old_flush) && flush != Z_FINISH) {
        strm->msg = err_msg(Z_BUF_ERROR);                // deflate.c:887-890
        return Z_BUF_ERROR;
    }
    if (s->status == ST_FINISH && strm->avail_in != 0) { strm->msg = err_msg(Z_BUF_ERROR); return Z_BUF_ERROR; }

    // Accept input; compress whenever a full batch is available.  Input is only
    // accepted while the pending output is empty, which bounds host memory.
    while (strm->avail_in != 0) {
        int rc;
        if (s->in.empty() && strm->avail_in >= s->batch_bytes) {
            rc = run_batch(strm, s, strm->next_in, s->batch_bytes, 0);
            if (rc != Z_OK) { strm->msg = err_msg(rc); return rc; }
            strm->next_in += s->batch_bytes; strm->avail_in -= (uInt)s->batch_bytes; strm->total_in += s->batch_bytes;
        } else {
            uint64_t take = s->batch_bytes - s->in.size();
            if (take > strm->avail_in) take = strm->avail_in;
            s->in.insert(s->in.end(), strm->next_in, strm->next_in + take);
            strm->next_in += take; strm->avail_in -= (uInt)take; strm->total_in += take;
            if (s->in.size() < s->batch_bytes) break;
            rc = run_batch(strm, s, s->in.data(), s->in.size(), 0);
            if (rc != Z_OK) { strm->msg = err_msg(rc); return rc; }
            s->in.clear();
        }
        flush_pending(strm, s);
        if (strm->avail_out == 0) { s->last_flush = -1; return Z_OK; }
    }

    if (flush != Z_NO_FLUSH && s->status != ST_FINISH) {
        int rc = finish_batch(strm, s, flush);
        if (rc != Z_OK) { strm->msg = err_msg(rc); return rc; }
        flush_pending(strm, s);
        if (strm->avail_out == 0 && s->out.size() != s->out_pos) { s->last_flush = -1; return Z_OK; }
    }
    if (flush != Z_FINISH) return Z_OK;
    flush_pending(strm, s);
    if (s->out.size() != s->out_pos) { s->last_flush = -1; return Z_OK; }
    return Z_STREAM_END;
}

int deflateEnd(z_streamp strm) {
    DState *s = dstate(strm);
    if (!s) return Z_STREAM_ERROR;
    int status = s->status;
    if (status != ST_INIT && status != ST_BUSY && status != ST_FINISH) return Z_STREAM_ERROR;
    s->magic = 0;
    s->~DState();
    strm->zfree(strm->opaque, s);
    strm->state = Z_NULL;
    return status == ST_BUSY ? Z_DATA_ERROR : Z_OK;      // deflate.c:1006
}

int deflateCopy(z_streamp dest, z_streamp source) {
    DState *ss = dstate(source);
    if (!ss || dest == Z_NULL) return Z_STREAM_ERROR;
    memcpy((void *)dest, (const void *)source, sizeof(z_stream));
    void *mem = dest->zalloc(dest->opaque, 1, (uInt)sizeof(DState));
    if (mem == Z_NULL) { dest->state = Z_NULL; return Z_MEM_ERROR; }
    DState *ds = new (mem) DState(*ss);
    dest->state = (struct internal_state *)ds;
    return Z_OK;
}

// ------------------------------------------------------------------ one-shot helpers
uLong compressBound(uLong sourceLen) {   // compress.c:76-80 plus the per-chunk markers of this engine
    uLong ref = sourceLen + (sourceLen >> 12) + (sourceLen >> 14) + (sourceLen >> 25) + 13;
    uLong own = deflateBound(Z_NULL, sourceLen);
    return own > ref ? own : ref;
}

int compress2(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen, int level) {   // compress.c:22-59
    z_stream stream;
    memset(&stream, 0, sizeof(stream));
    stream.next_in = source;
    stream.avail_in = (uInt)sourceLen;
    if ((uLong)stream.avail_in != sourceLen) return Z_BUF_ERROR;
    stream.next_out = dest;
    stream.avail_out = (uInt)*destLen;
    if ((uLong)stream.avail_out != *destLen) return Z_BUF_ERROR;
    int err = deflateInit_(&stream, level, ZLIB_VERSION, (int)sizeof(z_stream));
    if (err != Z_OK) return err;
    err = deflate(&stream, Z_FINISH);
    if (err != Z_STREAM_END) {
        deflateEnd(&stream);
        return err == Z_OK ? Z_BUF_ERROR : err;
    }
    *destLen = stream.total_out;
    return deflateEnd(&stream);
}

int compress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen) {
    return compress2(dest, destLen, source, sourceLen, Z_DEFAULT_COMPRESSION);
}

// ------------------------------------------------------------------ checksums
static uLong checksum_host_buf(int crc, uLong start, const Bytef *buf, uInt len) {
    if (buf == Z_NULL) return crc ? 0UL : 1UL;
    if (len == 0) return start;
    std::lock_guard<std::mutex> lk(g_gpu.mu);
    zamd_engine *e = engine_locked();
    if (!e) return crc ? 0UL : 1UL;                       // no GPU: no result (the initial value)
    if (grow(e, &g_gpu.d_in, &g_gpu.in_cap, (uint64_t)len + 64) != Z_OK) return crc ? 0UL : 1UL;
    uint32_t v = (uint32_t)start;
    if (zamd_dev_upload(e, g_gpu.d_in, buf, len) != 0) return crc ? 0UL : 1UL;
    int rc = crc ? zamd_crc32_dev(e, v, g_gpu.d_in, len, &v) : zamd_adler32_dev(e, v, g_gpu.d_in, len, &v);
    return rc == 0 ? (uLong)v : (crc ? 0UL : 1UL);
}

uLong adler32(uLong adler, const Bytef *buf, uInt len) { return checksum_host_buf(0, adler, buf, len); }
uLong crc32(uLong crc, const Bytef *buf, uInt len) { return checksum_host_buf(1, crc, buf, len); }
uLong adler32_combine(uLong a1, uLong a2, z_off_t len2) { return zamd_adler32_combine((uint32_t)a1, (uint32_t)a2, len2); }
uLong adler32_combine64(uLong a1, uLong a2, z_off64_t len2) { return zamd_adler32_combine((uint32_t)a1, (uint32_t)a2, len2); }
uLong crc32_combine(uLong c1, uLong c2, z_off_t len2) { return zamd_crc32_combine((uint32_t)c1, (uint32_t)c2, len2); }
uLong crc32_combine64(uLong c1, uLong c2, z_off64_t len2) { return zamd_crc32_combine((uint32_t)c1, (uint32_t)c2, len2); }

const z_crc_t *get_crc_table(void) {      // crc32.c:190-199: the byte-wise table (first of the reference's eight)
    static z_crc_t table[256];
    static std::once_flag once;
    std::call_once(once, [] {
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1)));
            table[n] = c;
        }
    });
    return table;
}

}  // extern "C"
