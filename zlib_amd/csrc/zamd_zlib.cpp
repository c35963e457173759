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
#ifndef _GNU_SOURCE
#define _GNU_SOURCE
#endif
#include "../../include/zamd.h"
#include "../../include/zamd_zlib.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <string>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <condition_variable>
#include <new>
#include <vector>

namespace {

const char *const k_errmsg[10] = {   // zutil.c:17-27
    "need dictionary", "stream end", "", "file error", "stream error",
    "data error", "insufficient memory", "buffer error", "incompatible version", ""};
inline const char *err_msg(int err) { return k_errmsg[Z_NEED_DICT - err]; }

// ------------------------------------------------------------------ engines
// A lane is an engine with its staging buffers; a call that needs the GPU takes a free lane for as long as it runs, so
// streams used from different threads at once (zlib.h:143-145) run side by side on the GPU up to the number of lanes
// (ZAMD_Z_LANES, 1..8, default 4), and queue for a lane beyond that.  A program with one thread only ever sees lane 0.
struct Gpu {
    std::mutex mu;           // held by the call that is using the lane
    zamd_engine *eng = nullptr;
    int rc = Z_OK;
    bool tried = false;
    uint8_t *d_in = nullptr;  uint64_t in_cap = 0;    // staging: [32 KiB history][batch]
    uint8_t *d_out = nullptr; uint64_t out_cap = 0;
    uint8_t *d_in_b = nullptr; uint64_t in_cap_b = 0;    // the second pair, used while batches of one call overlap
    uint8_t *d_out_b = nullptr; uint64_t out_cap_b = 0;
    uint64_t *d_off = nullptr; uint64_t off_cap = 0;
};
const int MAX_LANES = 8;
Gpu g_lanes[MAX_LANES];
thread_local Gpu *t_lane = nullptr;      // the lane this thread holds (helper threads of a call are given their call's lane)
#define g_gpu (*t_lane)
int lane_count() {
    static const int n = [] { const char *v = getenv("ZAMD_Z_LANES"); int k = v ? atoi(v) : 4; return k < 1 ? 1 : k > MAX_LANES ? MAX_LANES : k; }();
    return n;
}
struct LaneLock {
    Gpu *prev;
    LaneLock() : prev(t_lane) {
        const int n = lane_count();
        for (int i = 0; i < n; i++) if (g_lanes[i].mu.try_lock()) { t_lane = &g_lanes[i]; return; }
        // all busy: wait for one; which one is spread over the threads so that they do not all queue at lane 0
        static std::atomic<unsigned> turn{0};
        Gpu *g = &g_lanes[turn.fetch_add(1) % (unsigned)n];
        g->mu.lock();
        t_lane = g;
    }
    ~LaneLock() { t_lane->mu.unlock(); t_lane = prev; }
    LaneLock(const LaneLock &) = delete;
    LaneLock &operator=(const LaneLock &) = delete;
};

// ZAMD_DEVICES=0,1,...: deflate() calls that bring several batches at once spread them over these devices
// (zamd_deflate_multi), one engine each besides the lanes' own.  Caller holds g_multi_mu.
std::mutex g_multi_mu;
zamd_multi *g_multi = nullptr; bool g_multi_tried = false;
zamd_multi *multi_locked() {
    if (!g_multi_tried) {
        g_multi_tried = true;
        const char *v = getenv("ZAMD_DEVICES");
        std::vector<int> devs;
        for (const char *p = v; p && *p;) {
            char *end = nullptr;
            const long d = strtol(p, &end, 10);
            if (end == p) break;
            devs.push_back((int)d);
            p = *end == ',' ? end + 1 : end;
        }
        if (devs.size() >= 2 && zamd_multi_create(&g_multi, devs.data(), (int)devs.size(), 0) != 0) {
            fprintf(stderr, "zamd_z: ZAMD_DEVICES=%s: the devices could not be set up; one device is used\n", v);
            g_multi = nullptr;
        }
    }
    return g_multi;
}

// Caller holds a lane (LaneLock).
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

// Large downloads go through the engine's pinned staging (copy engine and host memcpy overlapped: 25 GB/s where the
// runtime's path for pageable memory gives 16).
int download(zamd_engine *e, void *h_dst, const void *d_src, uint64_t bytes) {
    return bytes >= (32u << 20) ? zamd_dev_download_staged(e, h_dst, d_src, bytes) : zamd_dev_download(e, h_dst, d_src, bytes);
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

// The buffers of a stream's state come from the stream's own zalloc/zfree, like the state itself (a caller that
// limits or counts memory through them, as test/infcover.c:55-234 does, then sees all of it).  zalloc takes 32-bit
// items and sizes: requests are made in units of 64 KiB.
template <class T> struct ZAlloc {
    typedef T value_type;
    alloc_func za; free_func zf; voidpf opaque;
    ZAlloc() : za(default_alloc), zf(default_free), opaque(Z_NULL) {}
    explicit ZAlloc(z_streamp strm) : za(strm->zalloc), zf(strm->zfree), opaque(strm->opaque) {}
    template <class U> ZAlloc(const ZAlloc<U> &o) : za(o.za), zf(o.zf), opaque(o.opaque) {}
    T *allocate(size_t n) {
        const uint64_t bytes = (uint64_t)n * sizeof(T);
        void *p = bytes <= 65536 ? za(opaque, 1, (uInt)(bytes ? bytes : 1)) : za(opaque, (uInt)((bytes + 65535) >> 16), 65536);
        if (p == Z_NULL) throw std::bad_alloc();
        return (T *)p;
    }
    void deallocate(T *p, size_t) { zf(opaque, p); }
    // resize() leaves new elements as they are (bytes, words: a buffer that a download or a copy is about to fill is not
    // cleared first; every resize() of these vectors either shrinks them or is followed by exactly that)
    template <class U> void construct(U *p) { ::new ((void *)p) U; }
    template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&... a) { ::new ((void *)p) U(std::forward<A0>(a0), std::forward<A>(a)...); }
    template <class U> bool operator==(const ZAlloc<U> &o) const { return za == o.za && zf == o.zf && opaque == o.opaque; }
    template <class U> bool operator!=(const ZAlloc<U> &o) const { return !(*this == o); }
};
typedef std::vector<uint8_t, ZAlloc<uint8_t>> Bytes;
typedef std::vector<uint32_t, ZAlloc<uint32_t>> Words;
typedef std::vector<uint64_t, ZAlloc<uint64_t>> Longs;

const uint32_t DEFLATE_MAGIC = 0x5a414d44u;   // distinguishes our states from inflate states
const uint32_t INFLATE_MAGIC = 0x5a414d49u;
const uint32_t HIST = 32768;
const uint32_t IDX_PARTS = 8;                 // parts a chunk of the index can be cut into: IDX_PARTS - 1 split records per chunk

struct Trace {      // ZAMD_Z_TRACE=1: phase times of every engine round trip on stderr
    bool on; std::chrono::steady_clock::time_point t; std::string line;
    Trace() : on(getenv("ZAMD_Z_TRACE") != nullptr), t(std::chrono::steady_clock::now()) {}
    void lap(const char *what) {
        if (!on) return;
        auto n = std::chrono::steady_clock::now();
        char b[64]; snprintf(b, sizeof b, " %s %.1f", what, std::chrono::duration<double, std::milli>(n - t).count());
        line += b; t = n;
    }
    void done(const char *who) { if (on) fprintf(stderr, "zamd_z %s:%s\n", who, line.c_str()); }
};

enum { ST_INIT = 42, ST_BUSY = 113, ST_FINISH = 666 };   // deflate.h:51-57 values

struct DState {
    uint32_t magic;
    int status, wrap, level, strategy, mem_level, w_bits, last_flush;
    int flush_done = -1;     // the flush whose blocks are written but whose output the caller is still collecting (-1: none)
    uint32_t chunk_size;
    uint64_t batch_bytes;
    gz_headerp gzhead;
    bool trailer_done, have_dict;
    uint32_t prime_bits, prime_val;    // deflatePrime bits (< 8) that precede the next compressed block
    bool tuned; int tune[4];           // deflateTune values, kept until deflateReset or a level change
    uint32_t dictid;
    uint32_t hist_len;                 // bytes of history kept in `hist` (== HIST once warm)
    uint64_t since_reset;              // input bytes compressed since the last history reset
    Bytes hist;                        // last 32 KiB of compressed input
    Bytes in;                          // input accepted from the caller, not yet compressed
    Bytes out;                         // produced output not yet handed to the caller
    size_t out_pos;
    bool index_on, index_ok;           // chunk index wanted (zamd_deflate_index); still describes the whole stream
    Words index;                       // per chunk since the last reset: compressed bytes, inflated bytes
    Longs splits;                      // and IDX_PARTS - 1 split records (block boundaries inside the chunk, zamd_engine_splits)
    explicit DState(z_streamp strm) : hist(ZAlloc<uint8_t>(strm)), in(ZAlloc<uint8_t>(strm)), out(ZAlloc<uint8_t>(strm)), index(ZAlloc<uint32_t>(strm)),
                                      splits(ZAlloc<uint64_t>(strm)) {}
};

inline DState *dstate(z_streamp strm) {
    if (strm == Z_NULL || strm->state == Z_NULL) return nullptr;
    DState *s = (DState *)strm->state;
    return s->magic == DEFLATE_MAGIC ? s : nullptr;
}

inline int rank_of(int f) { return (f << 1) - (f > 4 ? 9 : 0); }   // deflate.c:153

void put_be32(Bytes &v, uint32_t x) {
    v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x);
}
void put_le32(Bytes &v, uint32_t x) {
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
    Bytes &o = s->out;
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
        if (s->have_dict) header |= 0x20;                 // PRESET_DICT, deflate.c:750
        header += 31 - (header % 31);
        o.push_back(header >> 8); o.push_back(header & 0xff);
        if (s->have_dict) put_be32(o, s->dictid);         // deflate.c:757-760
        strm->adler = 1;                                  // adler32(0, NULL, 0)
    }
    s->status = ST_BUSY;
}

// One batch of input on its way through the GPU: upload, engine call, download.  The three steps are separate so
// that consecutive batches of one deflate() call can overlap them (run_batches_piped); run_batch does them in a row.
struct Batch {
    const uint8_t *src; uint64_t n;          // input of the batch (host)
    const uint8_t *head_src; uint32_t head;  // the history that primes it: `head` bytes ending at head_src + head
    int finish, tail, slot;                  // slot: which pair of staging buffers
    uint64_t bound, total;                   // room reserved for the output, bytes produced
};

inline uint8_t *&slot_in(int k) { return k ? g_gpu.d_in_b : g_gpu.d_in; }
inline uint64_t &slot_in_cap(int k) { return k ? g_gpu.in_cap_b : g_gpu.in_cap; }
inline uint8_t *&slot_out(int k) { return k ? g_gpu.d_out_b : g_gpu.d_out; }
inline uint64_t &slot_out_cap(int k) { return k ? g_gpu.out_cap_b : g_gpu.out_cap; }

int batch_reserve(zamd_engine *e, DState *s, Batch &b) {
    b.bound = zamd_deflate_bound(b.n, s->chunk_size, s->mem_level);
    if (grow(e, &slot_in(b.slot), &slot_in_cap(b.slot), HIST + b.n + 64) != Z_OK) return Z_MEM_ERROR;
    if (grow(e, &slot_out(b.slot), &slot_out_cap(b.slot), b.bound + 64) != Z_OK) return Z_MEM_ERROR;
    return Z_OK;
}
// `beside`: the transfer runs while another batch is on the GPU
int batch_upload(zamd_engine *e, const Batch &b, bool beside = false) {
    uint8_t *d = slot_in(b.slot);
    int (*const up)(zamd_engine *, void *, const void *, uint64_t) = beside ? zamd_dev_upload_staged : zamd_dev_upload;
    int rc = 0;
    if (b.head && b.head_src + b.head == b.src) return up(e, d + (HIST - b.head), b.head_src, (uint64_t)b.head + b.n);   // one piece of host memory
    if (b.head) rc = zamd_dev_upload(e, d + (HIST - b.head), b.head_src, b.head);
    if (rc == 0 && b.n) rc = up(e, d + HIST, b.src, b.n);
    return rc;
}
// The engine call of a batch and what it leaves in the stream state: chunk index, pending bits, data type, check value.
int batch_compute(z_streamp strm, DState *s, zamd_engine *e, Batch &b, Trace &tr) {
    // fault injection for tests of the callers' error paths (the allocator fake of test/infcover.c:55-234 plays this
    // part for the reference): ZAMD_Z_FAIL_DEFLATE_AFTER=k lets k engine calls through, then reports out of memory
    if (const char *fv = getenv("ZAMD_Z_FAIL_DEFLATE_AFTER")) {
        static long calls = 0;
        if (calls++ >= atol(fv)) return ZAMD_MEM_ERROR;
    }
    const uint64_t n = b.n;
    uint8_t *d_in = slot_in(b.slot), *d_out = slot_out(b.slot);
    int rc = 0;
    b.total = 0;
    const uint64_t nch = n ? (n + s->chunk_size - 1) / s->chunk_size : 1;
    uint64_t *d_tab = nullptr;
    const bool odd_tail = !b.finish && b.tail != ZAMD_TAIL_SYNC;
    // everything that can fail without the engine comes first: the engine is shared by all streams, and what is set on
    // it below is one-shot state that the call which follows must consume
    if (s->index_on && s->index_ok && !odd_tail) {
        const uint64_t tab_bytes = (nch + 1) * 8 + nch * (IDX_PARTS - 1) * 8;      // offsets, then the split records
        if (grow(e, (uint8_t **)&g_gpu.d_off, &g_gpu.off_cap, tab_bytes > HIST ? tab_bytes : HIST) != Z_OK) return ZAMD_MEM_ERROR;
        d_tab = g_gpu.d_off;
    }
    struct EngineReset {       // whatever happens below, the next stream finds the engine as it expects it
        zamd_engine *e;
        ~EngineReset() { zamd_engine_tune(e, 0, 0, 0, 0); zamd_engine_window_bits(e, 15); zamd_engine_prime(e, 0, 0); zamd_engine_tail(e, ZAMD_TAIL_SYNC);
                         zamd_engine_splits(e, nullptr, 0); }
    } engine_reset = {e};
    const uint32_t prime_bits = s->prime_bits, prime_val = s->prime_val;
    if (s->tuned) rc = zamd_engine_tune(e, s->tune[0], s->tune[1], s->tune[2], s->tune[3]);
    if (rc == 0) rc = zamd_engine_window_bits(e, s->w_bits);
    if (rc == 0 && prime_bits) rc = zamd_engine_prime(e, (int)prime_bits, (int)prime_val);
    if (rc == 0 && odd_tail) { rc = zamd_engine_tail(e, b.tail); s->index_ok = false; }
    if (rc == 0 && d_tab) rc = zamd_engine_splits(e, d_tab + nch + 1, IDX_PARTS);
    if (rc == 0)
        rc = zamd_deflate_shard_dev(e, d_in + HIST, n, b.head, s->chunk_size, s->level, s->strategy,
                                    s->mem_level, b.finish, d_out, b.bound, d_tab, &b.total);
    if (rc == 0 && d_tab) {
        // chunks end on byte boundaries (a sync marker or the final block), so byte counts describe them
        std::vector<uint64_t> off(nch + 1 + nch * (IDX_PARTS - 1));
        rc = zamd_dev_download(e, off.data(), d_tab, off.size() * 8);
        for (uint64_t c = 0; rc == 0 && c < nch; c++) {
            const uint64_t raw = n ? ((c + 1) * s->chunk_size <= n ? s->chunk_size : n - c * s->chunk_size) : 0;
            s->index.push_back((uint32_t)(off[c + 1] - off[c]));
            s->index.push_back((uint32_t)raw);
        }
        if (rc == 0) s->splits.insert(s->splits.end(), off.begin() + nch + 1, off.end());
    }
    if (rc == 0) {
        // the stream's pending bits went out in front of this call's first chunk; the bits behind the call's last whole
        // byte stay in the bit buffer (bi_flush, trees.c:1169-1181): they lead the next block.  A call that failed
        // leaves the stream's bits where they were.
        int pb = 0, pv = 0;
        zamd_engine_pending_bits(e, &pb, &pv);
        s->prime_bits = (uint32_t)pb; s->prime_val = (uint32_t)pv;
        // strm->data_type comes from the first block flushed at level > 0 (trees.c:920-921)
        if (strm->data_type == Z_UNKNOWN) { int dt = zamd_engine_data_type(e); if (dt != Z_UNKNOWN) strm->data_type = dt; }
    }
    tr.lap("deflate");
    if (tr.on) {       // the engine's own clock: spans of its kernels by stage, and of the whole call
        char buf[240]; int nl = 0;
        snprintf(buf, sizeof buf, " [chain %.1f match %.1f parse %.1f plan %.1f emit %.1f call %.1f; host: queueing %.1f waiting %.1f]", zamd_last_timing_ms(e, 0, &nl),
                 zamd_last_timing_ms(e, 1, &nl), zamd_last_timing_ms(e, 2, &nl), zamd_last_timing_ms(e, 3, &nl), zamd_last_timing_ms(e, 5, &nl), zamd_last_timing_ms(e, 6, &nl),
                 zamd_last_timing_ms(e, 7, &nl), zamd_last_timing_ms(e, 8, &nl));
        tr.line += buf;
    }
    if (rc == 0 && s->wrap && n) {
        uint32_t v = (uint32_t)strm->adler;
        rc = s->wrap == 2 ? zamd_crc32_dev(e, v, d_in + HIST, n, &v)
                          : zamd_adler32_dev(e, v, d_in + HIST, n, &v);
        strm->adler = v;
    }
    tr.lap("check");
    return rc;
}
// history for the next batch = the last window of everything compressed since the last reset
void batch_history(DState *s, const uint8_t *src, uint64_t n) {
    const uint32_t wsize = 1u << s->w_bits;
    s->since_reset += n;
    if (n >= wsize) {
        s->hist.assign(src + n - wsize, src + n);
        s->hist_len = wsize;
    } else if (n) {
        s->hist.insert(s->hist.end(), src, src + n);
        if (s->hist.size() > wsize) s->hist.erase(s->hist.begin(), s->hist.end() - wsize);
        s->hist_len = (uint32_t)s->hist.size();
    }
}
inline int z_code(int rc) { return rc == 0 ? Z_OK : rc == ZAMD_MEM_ERROR ? Z_MEM_ERROR : (rc == ZAMD_BUF_ERROR ? Z_BUF_ERROR : Z_STREAM_ERROR); }

// Compress `n` bytes at `src` (host) as chunks primed with the stream's history;
// append the result to s->out and fold the bytes into strm->adler.
// `tail`: how the output ends when finish == 0 (ZAMD_TAIL_*: sync marker, _tr_align, nothing).
int run_batch(z_streamp strm, DState *s, const uint8_t *src, uint64_t n, int finish, int tail = ZAMD_TAIL_SYNC) {
    LaneLock lk;
    zamd_engine *e = engine_locked();
    if (!e) return g_gpu.rc == ZAMD_MEM_ERROR ? Z_MEM_ERROR : Z_STREAM_ERROR;
    Batch b;
    b.src = src; b.n = n; b.head = s->hist_len; b.head_src = s->hist.data(); b.finish = finish; b.tail = tail; b.slot = 0; b.total = 0;
    if (batch_reserve(e, s, b) != Z_OK) return Z_MEM_ERROR;
    Trace tr;
    int rc = batch_upload(e, b);
    tr.lap("upload");
    if (rc == 0) rc = batch_compute(strm, s, e, b, tr);
    if (rc == 0 && b.total) {
        if (s->out.size() == s->out_pos && strm->avail_out >= b.total) {
            // nothing pending and room enough: straight into the caller's buffer
            rc = download(e, strm->next_out, g_gpu.d_out, b.total);
            if (rc == 0) { strm->next_out += b.total; strm->avail_out -= (uInt)b.total; strm->total_out += b.total; }
        } else {
            size_t at = s->out.size();
            s->out.resize(at + b.total);
            rc = download(e, s->out.data() + at, g_gpu.d_out, b.total);
        }
    }
    tr.lap("download"); tr.done("deflate batch");
    if (rc != 0) return z_code(rc);
    batch_history(s, src, n);
    return Z_OK;
}

// `count` full batches straight from the caller's input to the caller's output, three at a time in flight: while the
// engine compresses batch i, one thread uploads batch i + 1 into the other pair of staging buffers and another
// downloads the output of batch i - 1.  The caller has checked that nothing is pending and that the output has room
// for the bound of every batch.  Consumes the input and advances the stream counters itself.
// `last_flush` other than Z_NO_FLUSH: the rest of the input (shorter than a batch, possibly empty) follows as the
// last batch of the row and ends the way finish_batch would end it.
int run_batches_piped(z_streamp strm, DState *s, uint64_t full, int last_flush) {
    LaneLock lk;
    zamd_engine *e = engine_locked();
    if (!e) return g_gpu.rc == ZAMD_MEM_ERROR ? Z_MEM_ERROR : Z_STREAM_ERROR;
    const uint64_t nb = s->batch_bytes;
    const uint32_t wsize = 1u << s->w_bits;
    // How the row is cut into engine calls.  The stream does not depend on it (chunks are what it is made of).  An engine
    // call of two batches runs a tenth faster per byte than two calls of one (its serial kernels take as long either
    // way), but nothing overlaps the upload of the row's first call or the download of its last: those two are half a
    // batch, the calls between them two batches (levels 4-9; 3.5 GiB: 338 -> about 300 ms).
    const uint64_t total = last_flush != Z_NO_FLUSH ? strm->avail_in : full * nb;
    std::vector<uint64_t> sizes;
    {
        static const bool wide = !getenv("ZAMD_Z_EVEN_BATCHES");
        const uint64_t small = (nb / 2) / s->chunk_size * s->chunk_size, big = 2 * nb;
        uint64_t left = total;
        if (wide && s->level >= 4 && small >= wsize && small >= s->chunk_size && total >= big + 2 * small) {
            sizes.push_back(small); left -= small;
            while (left > big + small) { sizes.push_back(big); left -= big; }
            if (left > small) {                               // only the row's last call may end inside a chunk
                const uint64_t mid = (left - small) / s->chunk_size * s->chunk_size;
                if (mid) { sizes.push_back(mid); left -= mid; }
            }
            if (left) sizes.push_back(left);
        } else {
            for (uint64_t i = 0; i < full; i++) { sizes.push_back(nb); left -= nb; }
            if (last_flush != Z_NO_FLUSH) sizes.push_back(left);
        }
    }
    const uint64_t count = sizes.size();
    std::vector<Batch> bs(count);
    uint64_t at_in = 0, max_n = 0;
    for (uint64_t v : sizes) if (v > max_n) max_n = v;
    for (uint64_t i = 0; i < count; i++) {
        Batch &b = bs[i];
        b.src = strm->next_in + at_in; b.n = sizes[i]; b.finish = 0; b.tail = ZAMD_TAIL_SYNC; b.slot = (int)(i & 1); b.total = 0;
        at_in += b.n;
        if (i + 1 == count && last_flush != Z_NO_FLUSH) {
            b.finish = last_flush == Z_FINISH;
            b.tail = last_flush == Z_PARTIAL_FLUSH ? ZAMD_TAIL_ALIGN : last_flush == Z_BLOCK ? ZAMD_TAIL_NONE : ZAMD_TAIL_SYNC;   // deflate.c:923-945
        }
        if (i == 0) { b.head = s->hist_len; b.head_src = s->hist.data(); }
        else { b.head = wsize; b.head_src = b.src - wsize; }             // a batch is longer than the window
        if (i < 2) {                                                       // the buffers are sized once, for the largest call, before any thread uses them
            Batch widest = b; widest.n = max_n;
            if (batch_reserve(e, s, widest) != Z_OK) return Z_MEM_ERROR;
        }
        b.bound = zamd_deflate_bound(b.n, s->chunk_size, s->mem_level);
    }
    std::mutex mu; std::condition_variable cv;
    uint64_t uploaded = 0, launched = 0, computed = 0, downloaded = 0;
    int err = 0;
    // transfers start when the engine call they run beside has queued its kernels (see zamd_engine_on_launched)
    struct Hook { std::mutex *mu; std::condition_variable *cv; uint64_t *launched; } hook = {&mu, &cv, &launched};
    const int streams_before = zamd_engine_streams(e), serialized_before = zamd_engine_serialized(e);
    if (streams_before > 3) zamd_engine_set_streams(e, 3, serialized_before);        // the fourth hardware queue is the transfers'
    zamd_engine_on_launched(e, [](void *p) { Hook *h = (Hook *)p; { std::lock_guard<std::mutex> g(*h->mu); ++*h->launched; } h->cv->notify_all(); }, &hook);
    uint8_t *const dst0 = strm->next_out;
    uint64_t out_total = 0;
    Trace tr;
    std::vector<double> up_ms(count, 0.0), down_ms(count, 0.0);     // for the trace: how long each transfer took beside the kernels
    Gpu *const lane = t_lane;
    std::thread up([&] {
        t_lane = lane;
        for (uint64_t i = 0; i < count; i++) {
            { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return err || (computed + 2 > i && launched + 1 > i); }); if (err) return; }   // batch i - 2 is done with the slot, batch i - 1 is on the GPU
            const auto t0 = std::chrono::steady_clock::now();
            int rc = batch_upload(e, bs[i], i > 0);
            up_ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            { std::lock_guard<std::mutex> g(mu); if (rc) err = err ? err : rc; uploaded = i + 1; }
            cv.notify_all();
            if (rc) return;
        }
    });
    // should the second thread not start, the first is told to stop and joined before the exception travels on
    struct JoinOnUnwind {
        std::thread &t; std::mutex &mu; std::condition_variable &cv; int &err; zamd_engine *e; int streams, serialized; bool armed;
        ~JoinOnUnwind() {
            if (!armed) return;
            if (t.joinable()) { { std::lock_guard<std::mutex> g(mu); if (!err) err = ZAMD_MEM_ERROR; } cv.notify_all(); t.join(); }
            zamd_engine_on_launched(e, nullptr, nullptr);
            zamd_engine_set_streams(e, streams, serialized);
        }
    } up_guard = {up, mu, cv, err, e, streams_before, serialized_before, true};
    std::thread down([&] {
        t_lane = lane;
        uint64_t at = 0;
        for (uint64_t i = 0; i < count; i++) {
            { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return err || (computed > i && (launched > i + 1 || i + 1 >= count)); }); if (computed <= i) return; }
            const auto t0 = std::chrono::steady_clock::now();
            int rc = bs[i].total ? (i + 1 < count ? zamd_dev_download_staged : zamd_dev_download)(e, dst0 + at, slot_out(bs[i].slot), bs[i].total) : 0;
            down_ms[i] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            at += bs[i].total;
            { std::lock_guard<std::mutex> g(mu); if (rc) err = err ? err : rc; downloaded = i + 1; out_total = at; }
            cv.notify_all();
            if (rc) return;
        }
    });
    up_guard.armed = false;
    uint64_t done = 0;
    for (uint64_t i = 0; i < count; i++) {
        { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return err || (uploaded > i && downloaded + 2 > i); }); if (err) break; }
        tr.lap("wait");
        int rc = batch_compute(strm, s, e, bs[i], tr);
        { std::lock_guard<std::mutex> g(mu); if (rc) err = err ? err : rc; else computed = i + 1; }
        cv.notify_all();
        if (rc) break;
        done = i + 1;
    }
    { std::lock_guard<std::mutex> g(mu); if (done < count && !err) err = ZAMD_STREAM_ERROR; }
    cv.notify_all();
    up.join(); down.join();
    zamd_engine_on_launched(e, nullptr, nullptr);
    zamd_engine_set_streams(e, streams_before, serialized_before);
    if (tr.on) for (uint64_t i = 0; i < count; i++) { char b[64]; snprintf(b, sizeof b, " {up %.1f down %.1f}", up_ms[i], down_ms[i]); tr.line += b; }
    tr.done("deflate batches, pipelined");
    // what was compressed and downloaded before a failure stays in the stream, as after a failing run_batch in a row of them
    const uint64_t good = downloaded < done ? downloaded : done;
    if (good) {
        uint64_t bytes = 0;
        for (uint64_t i = 0; i < good; i++) bytes += bs[i].total;
        strm->next_out += bytes; strm->avail_out -= (uInt)bytes; strm->total_out += bytes;
        uint64_t taken = 0;
        for (uint64_t i = 0; i < good; i++) {
            if (i + 2 >= good) batch_history(s, bs[i].src, bs[i].n); else s->since_reset += bs[i].n;   // the last two decide the history
            taken += bs[i].n;
        }
        strm->next_in += taken; strm->avail_in -= (uInt)taken; strm->total_in += taken;
    }
    return err ? z_code(err) : Z_OK;
}

// `n` bytes of the caller's input straight to the caller's output on all the devices of ZAMD_DEVICES at once: contiguous
// ranges of chunks, one per device (zamd_deflate_multi).  Returns NO_MULTI when the call is not one for that path.
const int NO_MULTI = 200;
int run_multi(z_streamp strm, DState *s, uint64_t n, int finish) {
    if (s->prime_bits || s->w_bits != 15 || s->hist_len > HIST) return NO_MULTI;
    std::lock_guard<std::mutex> lk(g_multi_mu);
    zamd_multi *m = multi_locked();
    if (!m) return NO_MULTI;
    const int nd = zamd_multi_devices(m);
    struct Reset { zamd_multi *m; int nd; ~Reset() { for (int i = 0; i < nd; i++) zamd_engine_tune(zamd_multi_engine(m, i), 0, 0, 0, 0); } } reset = {m, nd};
    if (s->tuned) for (int i = 0; i < nd; i++) if (zamd_engine_tune(zamd_multi_engine(m, i), s->tune[0], s->tune[1], s->tune[2], s->tune[3]) != 0) return Z_STREAM_ERROR;
    Trace tr;
    const uint64_t nch = n ? (n + s->chunk_size - 1) / s->chunk_size : 1;
    std::vector<uint64_t> off(s->index_on && s->index_ok ? nch + 1 : 0);
    uint64_t total = 0;
    uint32_t a = (uint32_t)strm->adler, c = (uint32_t)strm->adler;
    int rc = zamd_deflate_multi(m, strm->next_in, n, s->hist.data(), s->hist_len, s->chunk_size, s->level, s->strategy, s->mem_level, finish,
                                strm->next_out, strm->avail_out, off.empty() ? nullptr : off.data(), &total,
                                s->wrap == 1 ? &a : nullptr, s->wrap == 2 ? &c : nullptr);
    tr.lap("deflate on all devices"); tr.done("deflate, several devices");
    if (rc != 0) return z_code(rc);
    if (s->wrap == 1) strm->adler = a; else if (s->wrap == 2) strm->adler = c;
    for (uint64_t k = 0; k + 1 < off.size(); k++) {
        s->index.push_back((uint32_t)(off[k + 1] - off[k]));
        s->index.push_back((uint32_t)(n ? ((k + 1) * s->chunk_size <= n ? s->chunk_size : n - k * s->chunk_size) : 0));
    }
    if (strm->data_type == Z_UNKNOWN) { int dt = zamd_engine_data_type(zamd_multi_engine(m, 0)); if (dt != Z_UNKNOWN) strm->data_type = dt; }
    batch_history(s, strm->next_in, n);
    strm->next_in += n; strm->avail_in -= (uInt)n; strm->total_in += n;
    strm->next_out += total; strm->avail_out -= (uInt)total; strm->total_out += total;
    return Z_OK;
}

void reset_history(DState *s) { s->hist.clear(); s->hist_len = 0; s->since_reset = 0; }

// `compressed`: the engine call that ends the flush has run already (as the last of a row of overlapped batches)
int finish_batch(z_streamp strm, DState *s, int flush, bool compressed = false) {
    const int tail = flush == Z_PARTIAL_FLUSH ? ZAMD_TAIL_ALIGN : flush == Z_BLOCK ? ZAMD_TAIL_NONE : ZAMD_TAIL_SYNC;   // deflate.c:923-945
    int rc = compressed ? Z_OK : run_batch(strm, s, s->in.data(), s->in.size(), flush == Z_FINISH, tail);
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

// No exception leaves the library: the host-side buffers are std::vectors, and an allocation that fails inside one
// of them comes back as Z_MEM_ERROR, as a failing zalloc does in the reference.
template <class F> int guarded(z_streamp strm, F f) {
    try { return f(); }
    catch (const std::bad_alloc &) { if (strm) strm->msg = err_msg(Z_MEM_ERROR); return Z_MEM_ERROR; }
    catch (...) { if (strm) strm->msg = err_msg(Z_STREAM_ERROR); return Z_STREAM_ERROR; }
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
    if (windowBits == 8) windowBits = 9;      // until 256-byte window bug fixed, deflate.c:276
    // level 0 with Z_HUFFMAN_ONLY / Z_RLE overruns the reference's own pending buffer (trees.c:1224) and is refused.
    if (level == 0 && (strategy == Z_HUFFMAN_ONLY || strategy == Z_RLE)) { strm->msg = "parameter not supported by the GPU engine"; return Z_STREAM_ERROR; }
    void *mem = strm->zalloc(strm->opaque, 1, (uInt)sizeof(DState));
    if (mem == Z_NULL) return Z_MEM_ERROR;
    DState *s = new (mem) DState(strm);
    s->magic = DEFLATE_MAGIC;
    s->wrap = wrap; s->level = level; s->strategy = strategy; s->mem_level = memLevel; s->w_bits = windowBits;
    s->gzhead = Z_NULL;
    uint64_t chunk = env_u64("ZAMD_Z_CHUNK", 1u << 20);
    if (chunk < 32768 || chunk > (64u << 20) || (chunk & 16383)) chunk = 1u << 20;
    s->chunk_size = (uint32_t)chunk;
    // deflate_fast is one wavefront per chunk: what it runs at is the number of chunks in one engine call (a 512 MiB batch
    // of level 1 occupies two wavefront slots of a CU and takes as long as one of 6 GiB), so levels 1-3 take larger batches
    const bool fast_levels = level >= 1 && level <= 3 && strategy != Z_HUFFMAN_ONLY && strategy != Z_RLE;
    uint64_t batch = env_u64("ZAMD_Z_BATCH_MIB", fast_levels ? 1024 : 512) << 20;
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
    s->flush_done = -1;
    s->trailer_done = false;
    s->have_dict = false; s->dictid = 0;
    s->prime_bits = 0; s->prime_val = 0;
    s->index.clear(); s->splits.clear(); s->index_ok = true;
    return Z_OK;
}

int deflateReset(z_streamp strm) {
    int ret = deflateResetKeep(strm);
    if (ret == Z_OK) {
        DState *s = dstate(strm);
        s->in.clear();
        reset_history(s);
        s->tuned = false;                      // lm_init takes the level's table again, deflate.c:1108-1111
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
    if (bits != Z_NULL) *bits = (int)s->prime_bits;     // compressed data ends every call on a byte boundary; only deflatePrime leaves bits
    return Z_OK;
}

static int deflateSetDictionary_body(z_streamp strm, const Bytef *dictionary, uInt dictLength) {    // deflate.c:323-389
    DState *s = dstate(strm);
    if (!s || dictionary == Z_NULL) return Z_STREAM_ERROR;
    if (s->wrap == 2 || (s->wrap == 1 && s->status != ST_INIT) || !s->in.empty() || s->since_reset != 0) return Z_STREAM_ERROR;
    if (s->wrap == 1) {
        uLong id = adler32(strm->adler, dictionary, dictLength);
        strm->adler = id; s->dictid = (uint32_t)id; s->have_dict = true;
    }
    // the window keeps the last w_size bytes (deflate.c:347-356); they prime the first chunk
    const uInt wsize = 1u << s->w_bits;
    uInt keep = dictLength > wsize ? wsize : dictLength;
    s->hist.assign(dictionary + (dictLength - keep), dictionary + dictLength);
    s->hist_len = keep;
    return Z_OK;
}
int deflateSetDictionary(z_streamp strm, const Bytef *dictionary, uInt dictLength) { return guarded(strm, [&] { return deflateSetDictionary_body(strm, dictionary, dictLength); }); }
int deflateTune(z_streamp strm, int good_length, int max_lazy, int nice_length, int max_chain) {    // deflate.c:531-547
    DState *s = dstate(strm);
    if (!s) return Z_STREAM_ERROR;
    // The values hold for input compressed from now on, buffered input included.  A chain budget
    // below 1 has no defined meaning in the reference's loop; levels 1-3 walk at most 64 candidates.
    if (good_length < 0 || max_lazy < 0 || nice_length < 0 || max_chain < 1 || good_length > 65535 ||
        max_lazy > 65535 || nice_length > 65535 || max_chain > 65535 ||
        (s->level >= 1 && s->level <= 3 && max_chain > 64)) {
        strm->msg = "deflateTune values not supported by the GPU engine";
        return Z_STREAM_ERROR;
    }
    s->tune[0] = good_length; s->tune[1] = max_lazy; s->tune[2] = nice_length; s->tune[3] = max_chain;
    s->tuned = true;
    return Z_OK;
}
static int deflatePrime_body(z_streamp strm, int bits, int value) {     // deflate.c:464-487
    DState *s = dstate(strm);
    if (!s || bits < 0 || bits > 32) return Z_STREAM_ERROR;
    // whole bytes go to the pending output at once (_tr_flush_bits, trees.c:882-886), the rest
    // waits for the next block
    uint64_t hold = (uint64_t)s->prime_val | ((uint64_t)((uint32_t)value & (bits == 32 ? 0xffffffffu : ((1u << bits) - 1u))) << s->prime_bits);
    uint32_t have = s->prime_bits + (uint32_t)bits;
    while (have >= 8) { s->out.push_back((uint8_t)hold); hold >>= 8; have -= 8; }
    s->prime_bits = have; s->prime_val = (uint32_t)hold;
    s->index_ok = false;                   // inserted bits are not part of any chunk
    return Z_OK;
}
int deflatePrime(z_streamp strm, int bits, int value) { return guarded(strm, [&] { return deflatePrime_body(strm, bits, value); }); }

static int deflateParams_body(z_streamp strm, int level, int strategy) {
    DState *s = dstate(strm);
    if (!s) return Z_STREAM_ERROR;
    if (level == Z_DEFAULT_COMPRESSION) level = 6;
    if (level < 0 || level > 9 || strategy < 0 || strategy > Z_FIXED) return Z_STREAM_ERROR;
    if (level == 0 && (strategy == Z_HUFFMAN_ONLY || strategy == Z_RLE)) return Z_STREAM_ERROR;
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
    if (level != s->level) s->tuned = false;   // a new level brings its own table values, deflate.c:519-525
    s->level = level;
    s->strategy = strategy;
    return err;
}
int deflateParams(z_streamp strm, int level, int strategy) { return guarded(strm, [&] { return deflateParams_body(strm, level, strategy); }); }

uLong deflateBound(z_streamp strm, uLong sourceLen) {
    DState *s = dstate(strm);
    uint32_t chunk = s ? s->chunk_size : (1u << 20);
    // one engine call per batch, each possibly ending in a sync marker, plus flush markers
    uLong body = (uLong)zamd_deflate_bound(sourceLen, chunk, s ? s->mem_level : 8) + 16;
    // batches that overlap on the GPU are downloaded straight into the caller's buffer at their bounds' spacing: the
    // constant part of every batch's bound counts once per batch
    body += (uLong)(sourceLen / (s && s->batch_bytes >= 65536 ? s->batch_bytes : 65536) + 2) * 160;
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

static int deflate_body(z_streamp strm, int flush) {
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
    // A flush is complete when deflate() returns with room left in the output (zlib.h:277-282).  Its blocks are written by
    // one engine call, however long the caller takes to collect them: calls that only collect must not flush again
    // (the reference writes the marker once the block in front of it has gone out, deflate.c:904-945).
    const bool collecting = s->flush_done == flush && strm->avail_in == 0 && flush != Z_NO_FLUSH;
    s->flush_done = -1;

    if (s->status == ST_INIT) write_header(strm, s);

    if (s->out.size() != s->out_pos) {                   // deflate.c:869-881
        flush_pending(strm, s);
        if (strm->avail_out == 0) { s->last_flush = -1; if (collecting) s->flush_done = flush; return Z_OK; }
    } else if (strm->avail_in == 0 && rank_of(flush) <= rank_of(old_flush) && flush != Z_FINISH) {
        strm->msg = err_msg(Z_BUF_ERROR);                // deflate.c:887-890
        return Z_BUF_ERROR;
    }
    if (s->status == ST_FINISH && strm->avail_in != 0) { strm->msg = err_msg(Z_BUF_ERROR); return Z_BUF_ERROR; }

    // Accept input; compress whenever a full batch is available.  Input is only
    // accepted while the pending output is empty, which bounds host memory.
    bool flushed = false;                  // the flush's own engine call has run inside a row of overlapped batches
    // Levels 1-3 with everything at hand (compress2, a one-shot deflate(Z_FINISH)): the whole input is one engine call.
    // Their parser is one wavefront per chunk, so the call takes about as long for 6 000 chunks as for 500, and cutting
    // the input into batches that overlap their transfers loses more than the overlap gains.
    if (flush == Z_FINISH && s->status != ST_FINISH && s->level >= 1 && s->level <= 3 && s->strategy != Z_HUFFMAN_ONLY &&
        s->strategy != Z_RLE && s->in.empty() && s->out.size() == s->out_pos && strm->avail_in > s->batch_bytes &&
        strm->avail_out >= zamd_deflate_bound(strm->avail_in, s->chunk_size, s->mem_level) + 64) {
        const uint64_t n = strm->avail_in;
        int rc = run_batch(strm, s, strm->next_in, n, 1);
        if (rc != Z_OK) { strm->msg = err_msg(rc); return rc; }
        strm->next_in += n; strm->avail_in = 0; strm->total_in += n;
        flushed = true;
    }
    while (strm->avail_in != 0) {
        int rc;
        // whole batches that can go straight from the caller's input to the caller's output, overlapped on the GPU
        uint64_t full = s->in.empty() ? strm->avail_in / s->batch_bytes : 0;
        const bool flushing = flush != Z_NO_FLUSH && s->status != ST_FINISH;
        bool with_last = flushing;
        // the flush belongs to the last batch that has input: input of exactly so many batches must not leave it an empty
        // one (that would be a block, or a marker, which deflate() with all the input at once does not write)
        if (flushing && full && strm->avail_in % s->batch_bytes == 0) full--;
        if (full >= 1) {
            // room for the bound of what the row takes (the bounds of its engine calls add up to that of their input, and
            // a few bytes per call)
            auto need = [&](uint64_t bytes, uint64_t calls) { return zamd_deflate_bound(bytes, s->chunk_size, s->mem_level) + 96 * calls + 64; };
            if (with_last && strm->avail_out < need(strm->avail_in, full + 1)) with_last = false;
            if (!with_last) while (full && strm->avail_out < need(full * s->batch_bytes, full)) full--;
        }
        // several devices: the whole row of batches as one call on all of them (flushes that end inside a byte stay on one)
        static const bool multi_on = getenv("ZAMD_DEVICES") != nullptr;
        const bool row = full + (with_last ? 1 : 0) >= 2 && full >= 1 && s->out.size() == s->out_pos;
        rc = NO_MULTI;
        if (multi_on && row && !(with_last && (flush == Z_PARTIAL_FLUSH || flush == Z_BLOCK))) {
            const uint64_t n = with_last ? strm->avail_in : full * s->batch_bytes;
            if (strm->avail_out >= zamd_deflate_bound(n, s->chunk_size, s->mem_level) + 64) rc = run_multi(strm, s, n, with_last && flush == Z_FINISH);
            if (rc != NO_MULTI && rc != Z_OK) { strm->msg = err_msg(rc); return rc; }
            if (rc == Z_OK && with_last) { flushed = true; break; }
        }
        if (rc == Z_OK) { /* done on all devices */ }
        else if (row && s->batch_bytes >= 65536 && !getenv("ZAMD_Z_NO_PIPELINE")) {
            rc = run_batches_piped(strm, s, full, with_last ? flush : Z_NO_FLUSH);
            if (rc != Z_OK) { strm->msg = err_msg(rc); return rc; }
            if (with_last) { flushed = true; break; }
        } else if (s->in.empty() && strm->avail_in >= s->batch_bytes) {
            const bool last = flushing && strm->avail_in == s->batch_bytes;       // the flush's own batch
            const int tail = flush == Z_PARTIAL_FLUSH ? ZAMD_TAIL_ALIGN : flush == Z_BLOCK ? ZAMD_TAIL_NONE : ZAMD_TAIL_SYNC;   // deflate.c:923-945
            rc = last ? run_batch(strm, s, strm->next_in, s->batch_bytes, flush == Z_FINISH, tail) : run_batch(strm, s, strm->next_in, s->batch_bytes, 0);
            if (rc != Z_OK) { strm->msg = err_msg(rc); return rc; }
            strm->next_in += s->batch_bytes; strm->avail_in -= (uInt)s->batch_bytes; strm->total_in += s->batch_bytes;
            if (last) { flushed = true; break; }
        } else {
            uint64_t take = s->batch_bytes - s->in.size();
            if (take > strm->avail_in) take = strm->avail_in;
            // a caller that writes by the megabyte fills the batch: room for all of it at once, not by doubling and copying
            if (s->in.capacity() < s->batch_bytes && s->in.size() + take > (16u << 20)) {
                try { s->in.reserve(s->batch_bytes); } catch (...) { /* an allocator that has no such piece: grow as needed */ }
            }
            s->in.insert(s->in.end(), strm->next_in, strm->next_in + take);
            strm->next_in += take; strm->avail_in -= (uInt)take; strm->total_in += take;
            if (s->in.size() < s->batch_bytes || (flushing && strm->avail_in == 0)) break;     // (a full batch that is the flush's own: finish_batch)
            rc = run_batch(strm, s, s->in.data(), s->in.size(), 0);
            if (rc != Z_OK) { strm->msg = err_msg(rc); return rc; }
            s->in.clear();
        }
        flush_pending(strm, s);
        if (strm->avail_out == 0) { s->last_flush = -1; return Z_OK; }
    }

    if (flush != Z_NO_FLUSH && s->status != ST_FINISH) {
        if (!collecting) {
            int rc = finish_batch(strm, s, flush, flushed);
            if (rc != Z_OK) { strm->msg = err_msg(rc); return rc; }
            flush_pending(strm, s);
        }
        if (strm->avail_out == 0 && s->out.size() != s->out_pos) { s->last_flush = -1; s->flush_done = flush; return Z_OK; }
    }
    if (flush != Z_FINISH) return Z_OK;
    flush_pending(strm, s);
    if (s->out.size() != s->out_pos) { s->last_flush = -1; return Z_OK; }
    return Z_STREAM_END;
}
int deflate(z_streamp strm, int flush) { return guarded(strm, [&] { return deflate_body(strm, flush); }); }

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

static int deflateCopy_body(z_streamp dest, z_streamp source) {
    DState *ss = dstate(source);
    if (!ss || dest == Z_NULL) return Z_STREAM_ERROR;
    memcpy((void *)dest, (const void *)source, sizeof(z_stream));
    dest->state = Z_NULL;                          // never the source's state, whatever fails below (deflate.c:1036-1053)
    void *mem = dest->zalloc(dest->opaque, 1, (uInt)sizeof(DState));
    if (mem == Z_NULL) return Z_MEM_ERROR;
    DState *ds;
    // the copy's buffers come from the caller's zalloc and may be refused: give the state's block back then
    try { ds = new (mem) DState(*ss); }
    catch (...) { dest->zfree(dest->opaque, mem); throw; }
    dest->state = (struct internal_state *)ds;
    return Z_OK;
}
int deflateCopy(z_streamp dest, z_streamp source) { return guarded(dest, [&] { return deflateCopy_body(dest, source); }); }

// Chunk index of the stream being written (no counterpart in the reference; the access points of
// examples/zran.c:67-72 serve the same end): one entry per chunk, in stream order.
int zamd_deflate_index(z_streamp strm, int enable) {
    DState *s = dstate(strm);
    if (!s) return Z_STREAM_ERROR;
    if (s->since_reset != 0 || !s->in.empty()) return Z_STREAM_ERROR;     // before the first input only
    s->index_on = enable != 0;
    s->index.clear(); s->splits.clear(); s->index_ok = true;
    return Z_OK;
}
int zamd_deflate_index_splits(z_streamp strm, const unsigned long long **recs, unsigned *per_chunk) {
    DState *s = dstate(strm);
    if (!s || !recs || !per_chunk) return Z_STREAM_ERROR;
    if (!s->index_on) return Z_STREAM_ERROR;
    if (!s->index_ok || s->splits.size() != s->index.size() / 2 * (IDX_PARTS - 1)) return Z_DATA_ERROR;
    static_assert(sizeof(unsigned long long) == sizeof(uint64_t), "split records are 64 bits wide");
    *recs = (const unsigned long long *)s->splits.data(); *per_chunk = IDX_PARTS - 1;
    return Z_OK;
}
unsigned zamd_deflate_chunk_bytes(z_streamp strm) { DState *s = dstate(strm); return s ? s->chunk_size : 0u; }
int zamd_z_engine_ready(void) {
    try { LaneLock lk; return engine_locked() ? Z_OK : (g_gpu.rc == ZAMD_MEM_ERROR ? Z_MEM_ERROR : Z_STREAM_ERROR); }
    catch (...) { return Z_MEM_ERROR; }
}
unsigned long long zamd_deflate_batch_bytes(z_streamp strm) { DState *s = dstate(strm); return s ? s->batch_bytes : 0ull; }
int zamd_deflate_index_get(z_streamp strm, const unsigned **pairs, unsigned *n_chunks) {
    DState *s = dstate(strm);
    if (!s || pairs == Z_NULL || n_chunks == Z_NULL) return Z_STREAM_ERROR;
    *pairs = Z_NULL; *n_chunks = 0;
    if (!s->index_on) return Z_STREAM_ERROR;
    if (!s->index_ok) return Z_DATA_ERROR;
    *pairs = s->index.data(); *n_chunks = (unsigned)(s->index.size() / 2);
    return Z_OK;
}

// ------------------------------------------------------------------ inflate side
}  // extern "C"

namespace {

enum IMode { IM_HEAD, IM_DICT, IM_BODY, IM_TRAILER, IM_DONE, IM_BAD, IM_MEM };

struct IState {
    uint32_t magic;
    int wrap;                       // 0 raw, bit 0 zlib allowed, bit 1 gzip allowed (inflate.c:153-164)
    unsigned wbits;
    IMode mode;
    bool gzip, havedict, last, synced;
    gz_headerp head;
    Bytes hdr;                      // wrapper header bytes seen so far
    Bytes in;                       // compressed bytes from the next block start on
    uint32_t bit_off;               // bit inside in[0] where that block starts
    size_t scanned;                 // in[0, scanned) has been searched for sync markers
    size_t tried;                   // size of `in` at the last attempt that made no progress
    Longs cands;                    // byte offsets in `in` just behind a 00 00 FF FF pattern
    Bytes out; size_t out_pos;
    Bytes hist;                     // last <= 32 KiB of output (the window)
    uint32_t check; uint64_t total; // running check value and length of the inflated data
    uint32_t dictid;
    uint64_t out_batch;
    uint64_t prime_hold; unsigned prime_bits;   // inflatePrime bits waiting in front of the first input byte
    unsigned tail_partial;          // inflatePrime bits behind buffered input: valid low bits of the last byte of `in` (0: it is whole)
    Words hints; size_t hint_pos;   // chunk index of the deflate data ahead (zamd_inflate_index_hint): compressed bytes, inflated bytes per chunk
    Longs hint_splits; unsigned hint_per;   // and hint_per split records per chunk (zamd_inflate_index_hint_splits), or none
    uint64_t hinted_chunks;         // chunks placed by their index entries since inflateInit (zamd_inflate_hinted_chunks)
    size_t call_taken;              // bytes the current inflate() call has moved from next_in to the end of `in`
    explicit IState(z_streamp strm) : hdr(ZAlloc<uint8_t>(strm)), in(ZAlloc<uint8_t>(strm)), cands(ZAlloc<uint64_t>(strm)), out(ZAlloc<uint8_t>(strm)),
                                      hist(ZAlloc<uint8_t>(strm)), hints(ZAlloc<uint32_t>(strm)), hint_splits(ZAlloc<uint64_t>(strm)), hint_per(0), hinted_chunks(0) { tail_partial = 0; }
};

// Hands the bytes at the end of `in` that the current call took (at most `want` of them) back to the caller.
// Bytes of earlier calls cannot go back: the caller's next_in no longer points behind them.
size_t unread_tail(z_streamp strm, IState *s, size_t want) {
    size_t mine = s->call_taken < s->in.size() ? s->call_taken : s->in.size();
    size_t k = want < mine ? want : mine;
    if (k == 0) return 0;
    strm->next_in -= k; strm->avail_in += (uInt)k; strm->total_in -= k;
    s->in.resize(s->in.size() - k);
    s->call_taken -= k;
    while (!s->cands.empty() && s->cands.back() > s->in.size()) s->cands.pop_back();
    if (s->scanned > s->in.size()) s->scanned = s->in.size();
    s->tried = 0;
    return k;
}

inline IState *istate(z_streamp strm) {
    if (strm == Z_NULL || strm->state == Z_NULL) return nullptr;
    IState *s = (IState *)strm->state;
    return s->magic == INFLATE_MAGIC ? s : nullptr;
}

int ibad(z_streamp strm, IState *s, const char *msg) { strm->msg = msg; s->mode = IM_BAD; return Z_DATA_ERROR; }

// Parses the wrapper header collected in s->hdr.  Returns 1 when complete, 0 when
// more bytes are needed, Z_DATA_ERROR on a bad header (msg set).  (inflate.c:640-825)
int parse_header(z_streamp strm, IState *s) {
    const Bytes &h = s->hdr;
    size_t n = h.size();
    if (n < 2) return 0;
    if ((s->wrap & 2) && h[0] == 0x1f && h[1] == 0x8b) {
        if (n < 4) return 0;
        if (h[2] != Z_DEFLATED) return ibad(strm, s, "unknown compression method");
        unsigned flags = h[3];
        if (flags & 0xe0) return ibad(strm, s, "unknown header flags set");
        if (n < 10) return 0;
        size_t pos = 10;
        size_t extra_at = 0, extra_len = 0, name_at = 0, comm_at = 0;
        if (flags & 4) {
            if (n < pos + 2) return 0;
            extra_len = h[pos] | (h[pos + 1] << 8);
            pos += 2; extra_at = pos;
            if (n < pos + extra_len) return 0;
            pos += extra_len;
        }
        if (flags & 8) {
            name_at = pos;
            while (pos < n && h[pos]) pos++;
            if (pos >= n) return 0;
            pos++;
        }
        if (flags & 16) {
            comm_at = pos;
            while (pos < n && h[pos]) pos++;
            if (pos >= n) return 0;
            pos++;
        }
        if (flags & 2) {
            if (n < pos + 2) return 0;
            uint32_t c = header_crc(0, h.data(), pos) & 0xffff;
            if ((unsigned)(h[pos] | (h[pos + 1] << 8)) != c) return ibad(strm, s, "header crc mismatch");
            pos += 2;
        }
        if (n != pos) return 0;      // bytes are appended one at a time, so this cannot overshoot
        gz_headerp g = s->head;
        if (g != Z_NULL) {
            g->text = flags & 1;
            g->time = (uLong)h[4] | ((uLong)h[5] << 8) | ((uLong)h[6] << 16) | ((uLong)h[7] << 24);
            g->xflags = h[8]; g->os = h[9];
            if (flags & 4) {
                g->extra_len = (uInt)extra_len;
                if (g->extra != Z_NULL) memcpy(g->extra, h.data() + extra_at, extra_len > g->extra_max ? g->extra_max : extra_len);
            } else g->extra = Z_NULL;
            if (flags & 8) {
                if (g->name != Z_NULL) for (uInt i = 0; i < g->name_max; i++) { g->name[i] = h[name_at + i]; if (!h[name_at + i]) break; }
            } else g->name = Z_NULL;
            if (flags & 16) {
                if (g->comment != Z_NULL) for (uInt i = 0; i < g->comm_max; i++) { g->comment[i] = h[comm_at + i]; if (!h[comm_at + i]) break; }
            } else g->comment = Z_NULL;
            g->hcrc = (flags >> 1) & 1;
            g->done = 1;
        }
        s->gzip = true;
        strm->adler = s->check = 0;
        s->mode = IM_BODY;
        return 1;
    }
    if (s->head != Z_NULL) s->head->done = -1;
    if (!(s->wrap & 1) || (((unsigned)h[0] << 8) + h[1]) % 31) return ibad(strm, s, "incorrect header check");
    if ((h[0] & 15) != Z_DEFLATED) return ibad(strm, s, "unknown compression method");
    unsigned len = (h[0] >> 4) + 8;
    if (s->wbits == 0) s->wbits = len;
    else if (len > s->wbits) return ibad(strm, s, "invalid window size");
    s->gzip = false;
    if (h[1] & 0x20) {
        if (n < 6) return 0;
        s->dictid = ((uint32_t)h[2] << 24) | ((uint32_t)h[3] << 16) | ((uint32_t)h[4] << 8) | h[5];
        strm->adler = s->check = s->dictid;
        s->mode = IM_DICT;
        return 1;
    }
    strm->adler = s->check = 1;
    s->mode = IM_BODY;
    return 1;
}

void idrain(z_streamp strm, IState *s) {
    size_t len = s->out.size() - s->out_pos;
    if (len > strm->avail_out) len = strm->avail_out;
    if (len == 0) return;
    memcpy(strm->next_out, s->out.data() + s->out_pos, len);
    strm->next_out += len; strm->avail_out -= (uInt)len; strm->total_out += len;
    s->out_pos += len;
    if (s->out_pos == s->out.size()) { s->out.clear(); s->out_pos = 0; }
}

// Places the segments of `chain` (ZAMD_INF_EXACT, their outputs back to back from offset 0) straight into the caller's
// buffer in groups of about `group_out` bytes of output: while the GPU decodes group k, one helper thread downloads
// group k - 1 and (when the compressed data is not on the device yet: `d_comp` null) another uploads the input of
// group k + 1.  The window in front of a group is the end of the group before it, still on the device.  `good` receives
// the number of leading segments that were placed, verified and downloaded; a segment that does not end where and as it
// should stops the row there (the caller decides what that means).  Returns 0, or the engine's error code.
int place_piped(zamd_engine *e, IState *s, const uint8_t *ip, size_t in_n, const uint8_t *d_comp, std::vector<zamd_inflate_seg> &chain,
                uint8_t *dst, uint64_t group_out, size_t *good, uint32_t *bad_status, Trace &tr) {
    struct Grp { size_t s0, s1; uint64_t out_off, out_len, in0, in_len; };
    std::vector<Grp> gs;
    for (size_t i = 0; i < chain.size();) {
        Grp g; g.s0 = i; g.out_off = chain[i].out_off; g.out_len = 0;
        while (i < chain.size() && (g.out_len < group_out || chain.size() - i < 2)) { g.out_len += chain[i].out_cap; i++; }
        g.s1 = i;
        g.in0 = chain[g.s0].in_bit >> 3; g.in_len = ((chain[g.s1 - 1].in_end_bit + 7) >> 3) - g.in0;
        gs.push_back(g);
    }
    *good = 0; *bad_status = 0;
    const size_t ng = gs.size();
    uint64_t max_in = 0, max_out = 0;
    for (const Grp &g : gs) { if (g.in_len > max_in) max_in = g.in_len; if (g.out_len > max_out) max_out = g.out_len; }
    for (int k = 0; k < 2; k++) {
        if (!d_comp && grow(e, &slot_in(k), &slot_in_cap(k), max_in + 64) != Z_OK) return ZAMD_MEM_ERROR;
        if (grow(e, &slot_out(k), &slot_out_cap(k), max_out + 64) != Z_OK) return ZAMD_MEM_ERROR;
    }
    uint8_t *d_head0 = nullptr;
    if (!s->hist.empty()) {
        if (grow(e, (uint8_t **)&g_gpu.d_off, &g_gpu.off_cap, HIST) != Z_OK) return ZAMD_MEM_ERROR;
        d_head0 = (uint8_t *)g_gpu.d_off;
        if (zamd_dev_upload(e, d_head0 + (HIST - s->hist.size()), s->hist.data(), s->hist.size()) != 0) return ZAMD_MEM_ERROR;
    }
    std::mutex mu; std::condition_variable cv;
    size_t uploaded = d_comp ? ng : 0, computed = 0, downloaded = 0;
    int err = 0;
    std::thread up, down;
    const auto t_start = std::chrono::steady_clock::now();
    auto now_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
    std::vector<double> t_up(ng, 0), t_comp0(ng, 0), t_comp(ng, 0), t_down0(ng, 0), t_down(ng, 0);      // ZAMD_Z_TRACE: when each step of each group ended
    std::vector<double> t_dec(ng, 0), t_res(ng, 0), t_seg(ng, 0);
    struct Joiner {            // whatever happens, the helpers are told to stop and joined before their captures go away
        std::thread &a, &b; std::mutex &mu; std::condition_variable &cv; int &err; size_t &computed; size_t ng;
        ~Joiner() {
            { std::lock_guard<std::mutex> g(mu); if (computed < ng && !err) err = ZAMD_STREAM_ERROR; }
            cv.notify_all();
            if (a.joinable()) a.join();
            if (b.joinable()) b.join();
        }
    } joiner = {up, down, mu, cv, err, computed, ng};
    Gpu *const lane = t_lane;
    if (!d_comp)
        up = std::thread([&] {
            t_lane = lane;
            for (size_t k = 0; k < ng; k++) {
                { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return err || computed + 2 > k; }); if (err) return; }
                int rc = zamd_dev_upload_staged(e, slot_in((int)(k & 1)), ip + gs[k].in0, gs[k].in_len);
                t_up[k] = now_ms();
                { std::lock_guard<std::mutex> g(mu); if (rc) err = err ? err : rc; uploaded = k + 1; }
                cv.notify_all();
                if (rc) return;
            }
        });
    down = std::thread([&] {
        t_lane = lane;
        for (size_t k = 0; k < ng; k++) {
            { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return err || computed > k; }); if (computed <= k) return; }
            t_down0[k] = now_ms();
            int rc = zamd_dev_download_staged(e, dst + gs[k].out_off, slot_out((int)(k & 1)), gs[k].out_len);
            t_down[k] = now_ms();
            { std::lock_guard<std::mutex> g(mu); if (rc) err = err ? err : rc; downloaded = k + 1; }
            cv.notify_all();
            if (rc) return;
        }
    });
    std::vector<zamd_inflate_seg> segs;
    size_t verified = 0;
    for (size_t k = 0; k < ng; k++) {
        { std::unique_lock<std::mutex> g(mu); cv.wait(g, [&] { return err || (uploaded > k && downloaded + 2 > k); }); if (err) break; }
        const Grp &g = gs[k];
        t_comp0[k] = now_ms();
        segs.assign(chain.begin() + g.s0, chain.begin() + g.s1);
        const uint64_t bit0 = d_comp ? 0 : g.in0 * 8;
        for (zamd_inflate_seg &q : segs) { q.in_bit -= bit0; q.in_end_bit -= bit0; q.out_off -= g.out_off; }
        const uint8_t *d_head = k == 0 ? d_head0 : slot_out((int)((k - 1) & 1)) + gs[k - 1].out_len - HIST;
        uint8_t *d_out = slot_out((int)(k & 1));
        int rc = zamd_inflate_segments_dev(e, d_comp ? d_comp : slot_in((int)(k & 1)), d_comp ? in_n : g.in_len, segs.data(), segs.size(), d_head, d_out, g.out_len);
        t_seg[k] = now_ms();
        { int nl = 0; t_dec[k] = zamd_last_timing_ms(e, 7, &nl); t_res[k] = zamd_last_timing_ms(e, 8, &nl); }
        bool fits = rc == 0;
        for (size_t i = 0; fits && i < segs.size(); i++) {
            if (segs[i].status >= 3) { *bad_status = segs[i].status; fits = false; }
            if (segs[i].status == 1 && !(k + 1 == ng && i + 1 == segs.size())) { *bad_status = 1; fits = false; }     // the final block may only close the last segment
        }
        if (fits) for (size_t i = 0; i < segs.size(); i++) { chain[g.s0 + i].status = segs[i].status; chain[g.s0 + i].end_bit = segs[i].end_bit + bit0; }
        if (fits && s->wrap) {
            uint32_t v = s->check;
            rc = s->gzip ? zamd_crc32_dev(e, v, d_out, g.out_len, &v) : zamd_adler32_dev(e, v, d_out, g.out_len, &v);
            if (rc == 0) s->check = v;
        }
        t_comp[k] = now_ms();
        { std::lock_guard<std::mutex> g2(mu); if (rc) err = err ? err : rc; else if (fits) computed = k + 1; }
        cv.notify_all();
        if (rc || !fits) break;
        verified = g.s1;
    }
    { std::lock_guard<std::mutex> g(mu); if (computed < ng && !err) err = 1; }     // 1: stop, nothing wrong with the GPU
    cv.notify_all();
    if (up.joinable()) up.join();
    if (down.joinable()) down.join();
    tr.lap("place+download, pipelined");
    if (tr.on) {
        char b[200];
        for (size_t k = 0; k < ng; k++) {
            snprintf(b, sizeof b, " [g%zu up %.0f kern %.0f-%.0f (segments call until %.0f: decode %.1f resolve %.1f) down %.0f-%.0f]", k, t_up[k], t_comp0[k], t_comp[k],
                     t_seg[k], t_dec[k], t_res[k], t_down0[k], t_down[k]);
            tr.line += b;
        }
    }
    if (err && err != 1) return err;
    // what counts is what has arrived in the caller's buffer
    *good = downloaded >= ng ? verified : (downloaded ? gs[downloaded - 1].s1 : 0);
    if (*good > verified) *good = verified;
    return 0;
}

// One decode attempt over the buffered input, or, with `ext`, over the caller's own input buffer
// (nothing is buffered then: what the attempt consumes is taken off next_in, the rest stays with the
// caller).  Returns Z_OK (with or without progress), NO_PROGRESS (ext only: not one whole block in
// there), Z_DATA_ERROR (msg set) or Z_MEM_ERROR.
const int NO_PROGRESS = 100, HINT_MISFIT = 101;
inline void drop_hints(IState *s) { s->hints.clear(); s->hint_pos = 0; s->hint_splits.clear(); s->hint_per = 0; }
int decode_some(z_streamp strm, IState *s, const uint8_t *ext = nullptr, size_t ext_n = 0) {
    const bool view = ext != nullptr;
    // view with bytes in the buffer (the rest of a chunk that an earlier call left there): the input is those bytes and,
    // behind them, the caller's; it becomes one piece on the device, and nothing is copied on the host
    const size_t pre_n = view ? s->in.size() : 0;
    const uint8_t *const ip = view ? ext : s->in.data();
    const size_t in_n = view ? pre_n + ext_n : s->in.size();
    Longs view_cands(s->cands.get_allocator()); size_t view_scanned = 0;
    Longs &cands = view ? view_cands : s->cands;
    size_t &scanned = view ? view_scanned : s->scanned;
    // bits that inflatePrime put behind the buffered input end inside the last byte
    const uint64_t end_bits = in_n * 8ull - ((!view && s->tail_partial) ? 8u - s->tail_partial : 0u);
    if (end_bits <= s->bit_off) { if (view) return NO_PROGRESS; s->tried = in_n; return Z_OK; }
    Trace tr;
    LaneLock lk;
    zamd_engine *e = engine_locked();
    if (!e) return Z_MEM_ERROR;
    // the input goes to the device when somebody needs it there: a round that places chunks by their index in
    // overlapped groups uploads group by group
    bool on_device = false;
    auto need_input = [&]() -> int {
        if (on_device) return Z_OK;
        if (grow(e, &g_gpu.d_in, &g_gpu.in_cap, in_n + 64) != Z_OK) return Z_MEM_ERROR;
        if (pre_n && zamd_dev_upload(e, g_gpu.d_in, s->in.data(), pre_n) != 0) return Z_MEM_ERROR;
        if ((in_n - pre_n >= (32u << 20) ? zamd_dev_upload_staged : zamd_dev_upload)(e, g_gpu.d_in + pre_n, ip, in_n - pre_n) != 0) return Z_MEM_ERROR;
        on_device = true;
        tr.lap("upload");
        return Z_OK;
    };
    // what a round may produce: the output batch, or what the caller's buffer takes when the round can write straight into it
    uint64_t out_batch = s->out_batch;
    if (s->out.size() == s->out_pos && strm->avail_out >= (64u << 20) && strm->avail_out < out_batch) out_batch = strm->avail_out;
    std::vector<zamd_inflate_seg> chain;
    uint64_t produced = 0, end_bit = s->bit_off;
    bool final_block = false;
    // With a chunk index for the data ahead, the chunks that are complete in the input are placed directly:
    // no marker search and no counting pass.  An index that turns out not to fit the data is dropped and
    // the attempt starts over without it.
    bool hinted = false;
    size_t hints_used = 0;
    std::vector<uint32_t> chunk_end;        // hinted: chain[chunk_end[k] - 1] is the last segment of the k-th chunk taken
    std::vector<unsigned long long> marks;  // not hinted: block marks of the counting pass, a row per segment, and the segments' first bits
    std::vector<uint64_t> mark_bits;
    bool cut = false;                       // some segment of the chain was cut at its marks
    if (s->hint_pos < s->hints.size() && s->bit_off == 0) {
        uint64_t pos = 0;
        const unsigned per = s->hint_splits.size() == s->hints.size() / 2 * s->hint_per ? s->hint_per : 0u;
        for (size_t h = s->hint_pos; h + 1 < s->hints.size() && chunk_end.size() < 8192 && chain.size() + per + 1 <= 32768; h += 2) {
            const uint64_t clen = s->hints[h], raw = s->hints[h + 1];
            if (raw == 0 || pos + clen > in_n || (produced + raw > out_batch && !chain.empty())) break;
            // the chunk's split records cut it into parts, one wavefront each; a record that is out of order or outside
            // the chunk ends the chunk's list (one that is no block boundary shows when its part does not decode)
            uint64_t pbit = pos * 8, pout = 0;
            auto part = [&](uint64_t bit_end, uint64_t out_end) {
                zamd_inflate_seg g; memset(&g, 0, sizeof g);
                g.in_bit = pbit; g.in_end_bit = bit_end; g.out_off = produced + pout; g.out_cap = out_end - pout; g.out_len = g.out_cap;
                g.win_avail = (uint32_t)((s->hist.size() + g.out_off) < HIST ? (s->hist.size() + g.out_off) : HIST);
                g.mode = ZAMD_INF_EXACT;
                chain.push_back(g);
                pbit = bit_end; pout = out_end;
            };
            for (unsigned k = 0; k < per; k++) {
                const uint64_t rec = s->hint_splits[h / 2 * per + k];
                const uint64_t bit = pos * 8 + (rec >> 32), off = rec & 0xffffffffull;
                if (rec == 0 || off <= pout || off >= raw || bit <= pbit || bit >= (pos + clen) * 8) break;
                part(bit, off);
            }
            part((pos + clen) * 8, raw);
            chunk_end.push_back((uint32_t)chain.size());
            pos += clen; produced += raw; hints_used += 2;
        }
        hinted = !chain.empty();
        if (hinted) end_bit = pos * 8;
        else if (s->hints[s->hint_pos + 1] != 0 && s->hints[s->hint_pos + 1] <= out_batch) {
            // the next chunk is not whole in the input yet: wait for the rest rather than leave the chunk grid
            if (view) return NO_PROGRESS;
            s->tried = in_n;
            return Z_OK;
        }
        else { drop_hints(s); }         // an empty or oversized chunk: the general path takes over
    }
    if (!hinted) {
        if (need_input() != Z_OK) return Z_MEM_ERROR;
        // sync-marker candidates: independent starting points for parallel wavefronts.  The bytes that are
        // new since the last attempt are searched on the GPU (a marker may straddle the old end).
        {
            const uint64_t from = scanned < 3 ? 0 : scanned - 3;
            std::vector<uint64_t> found(1u << 18);
            uint32_t cnt = 0;
            if (zamd_find_sync_dev(e, g_gpu.d_in, from, in_n, found.data(), (uint32_t)found.size(), &cnt) != 0) return Z_MEM_ERROR;
            if (cnt <= found.size()) {
                found.resize(cnt);
                std::sort(found.begin(), found.end());
            } else {                                        // more markers than the list holds: search on the host
                if (pre_n) return NO_PROGRESS;              // (in two pieces there: the caller buffers them into one)
                found.clear();
                static const uint8_t pat[4] = {0, 0, 0xff, 0xff};
                const uint8_t *base = ip, *end = base + in_n;
                for (const uint8_t *q = base + from; q + 4 <= end;) {
                    const uint8_t *hit = (const uint8_t *)memmem(q, (size_t)(end - q), pat, 4);
                    if (!hit) break;
                    found.push_back((uint64_t)(hit - base) + 4);
                    q = hit + 1;
                }
            }
            for (uint64_t c : found) if (cands.empty() || cands.back() < c) cands.push_back(c);
            scanned = in_n;
        }
        tr.lap("scan");
        std::vector<zamd_inflate_seg> segs;
        // A candidate is a guess: what starts there may be the middle of a stored block, so its
        // wavefront gets a bounded amount of input and output (a real segment that is longer is
        // continued from where it stopped, as the known start of the next attempt).
        const uint64_t spec_bits = 96ull << 23, spec_out = 96ull << 20;
        auto add = [&](uint64_t bit, uint32_t win, bool guess) {
            zamd_inflate_seg g; memset(&g, 0, sizeof g);
            g.in_bit = bit; g.in_end_bit = (guess && bit + spec_bits < end_bits) ? bit + spec_bits : end_bits;
            g.out_off = 0; g.out_cap = guess ? spec_out : (~0ull >> 2); g.win_avail = win; g.mode = ZAMD_INF_SYNCSTOP;
            segs.push_back(g);
        };
        // Stretches without sync markers (streams of other producers have none at all): guessed starts of
        // dynamic blocks, one per STEP of input, found on the GPU.  A segment stops at the first block boundary
        // at or behind the next guess; a guess that is no block start just ends the chain there for this round.
        std::vector<uint64_t> bc;
        {
            // (a round of a few megabytes is as slow as its longest segment, 4 ms of one wavefront per deflate block of
            // some 35 KiB: guesses 16 KiB apart find every block, 32 KiB apart leave segments of two)
            static const uint64_t step_env = getenv("ZAMD_Z_BLOCKSCAN_KIB") ? strtoull(getenv("ZAMD_Z_BLOCKSCAN_KIB"), nullptr, 10) * 8192ull : ~0ull;
            const uint64_t STEP = step_env != ~0ull ? step_env : (in_n <= (32ull << 20) ? 16 : 32) * 8192ull;
            if (STEP && end_bits > s->bit_off + 2 * STEP) {
                // only where sync markers are further apart than GAP.  A wavefront inflates about 20 MB a second: a round of
                // few segments takes as long as its longest one, so a small round is cut finer than its sync markers cut it
                // (a 1 MiB chunk alone takes 50 ms, eight pieces of it 7); a large round has segments enough to fill the GPU,
                // and the search for block starts would cost more than the finer cut saves (87 ms on 1.7 GB of input)
                static const uint64_t gap_env = getenv("ZAMD_Z_BLOCKSCAN_GAP_KIB") ? strtoull(getenv("ZAMD_Z_BLOCKSCAN_GAP_KIB"), nullptr, 10) : 0;
                const uint64_t GAP = (gap_env ? gap_env : in_n <= (512ull << 20) ? 256 : 2048) * 8192ull;
                std::vector<uint64_t> tg;
                size_t ci = 0;                                   // first marker behind t
                for (uint64_t t = s->bit_off + STEP; t + STEP / 2 <= end_bits && tg.size() < (1u << 16); t += STEP) {
                    while (ci < cands.size() && cands[ci] * 8 <= t) ci++;
                    const bool m0 = ci > 0 && cands[ci - 1] * 8 > s->bit_off, m1 = ci < cands.size() && cands[ci] * 8 < end_bits;
                    const uint64_t before = m0 ? cands[ci - 1] * 8 : s->bit_off;
                    const uint64_t behind = m1 ? cands[ci] * 8 : end_bits;
                    if ((m0 || m1) && behind - before < GAP) continue;    // (a short stream without any marker is searched)
                    if (t - before < STEP / 2 || behind - t < STEP / 2) continue;        // a marker close by is a start already
                    tg.push_back(t);
                }
                if (!tg.empty()) {
                    std::vector<uint64_t> found(tg.size());
                    if (zamd_find_blocks_dev(e, g_gpu.d_in, in_n, tg.data(), (uint32_t)tg.size(), STEP, found.data()) != 0) return Z_MEM_ERROR;
                    for (uint64_t f : found) if (f != ~0ull && f > s->bit_off && f < end_bits && (bc.empty() || bc.back() < f)) bc.push_back(f);
                }
            }
        }
        tr.lap("blockscan");
        {
            std::vector<uint64_t> starts;
            for (uint64_t c : cands) if (c * 8 > s->bit_off && c * 8 < end_bits) starts.push_back(c * 8);
            starts.insert(starts.end(), bc.begin(), bc.end());
            std::sort(starts.begin(), starts.end());
            starts.erase(std::unique(starts.begin(), starts.end()), starts.end());
            add(s->bit_off, (uint32_t)s->hist.size(), false);
            for (uint64_t b : starts) if (segs.size() < (1u << 20)) add(b, HIST, true);
            size_t bi = 0;
            for (zamd_inflate_seg &g : segs) {
                while (bi < bc.size() && bc[bi] <= g.in_bit) bi++;
                if (bi < bc.size()) { g.mode |= ZAMD_INF_BLOCKSTOP; g.stop_bit = bc[bi]; }
            }
        }

        // the counting pass notes block boundaries inside its segments (one about every MARK_STEP bytes of output): the pass
        // that writes the output cuts the segments there, a wavefront to a part (a 1 MiB chunk is some 50 ms of one
        // wavefront's time)
        // (in a small round every block boundary is a mark: its parts are single blocks, and the round is as slow as its longest part)
        static const uint32_t mark_env = getenv("ZAMD_Z_INF_MARK_KIB") ? (uint32_t)env_u64("ZAMD_Z_INF_MARK_KIB", 128) << 10 : ~0u;
        const uint32_t MARK_STEP = mark_env != ~0u ? mark_env : in_n <= (32ull << 20) ? (32u << 10) : (128u << 10);
        if (MARK_STEP) {
            marks.assign(segs.size() * ZAMD_INF_MARKS, 0ull);
            for (const zamd_inflate_seg &g : segs) mark_bits.push_back(g.in_bit);
            if (zamd_inflate_block_marks(e, marks.data(), MARK_STEP) != 0) return Z_MEM_ERROR;
        }
        if (zamd_inflate_segments_dev(e, g_gpu.d_in, in_n, segs.data(), segs.size(), nullptr, nullptr, 0) != 0) return Z_MEM_ERROR;

        tr.lap("count");
        // A guess that is no block start leaves the segment in front of it ending where no segment starts.  Such
        // ends get segments of their own, all of them in one more pass (and again, should one of those end the same way).
        for (int round = 0; round < 6 && !bc.empty(); round++) {
            std::vector<uint64_t> loose;
            for (const zamd_inflate_seg &g : segs) {
                if (g.status != 2 || g.end_bit >= end_bits) continue;
                size_t lo = 0, hi = segs.size();
                while (lo < hi) { size_t mid = (lo + hi) / 2; if (segs[mid].in_bit < g.end_bit) lo = mid + 1; else hi = mid; }
                if (lo >= segs.size() || segs[lo].in_bit != g.end_bit) loose.push_back(g.end_bit);
            }
            if (loose.empty()) break;
            std::sort(loose.begin(), loose.end());
            loose.erase(std::unique(loose.begin(), loose.end()), loose.end());
            std::vector<zamd_inflate_seg> fix;
            size_t bi = 0;
            for (uint64_t b : loose) {
                zamd_inflate_seg g; memset(&g, 0, sizeof g);
                g.in_bit = b; g.in_end_bit = b + spec_bits < end_bits ? b + spec_bits : end_bits;
                g.out_cap = spec_out; g.win_avail = HIST; g.mode = ZAMD_INF_SYNCSTOP;
                while (bi < bc.size() && bc[bi] <= b) bi++;
                if (bi < bc.size()) { g.mode |= ZAMD_INF_BLOCKSTOP; g.stop_bit = bc[bi]; }
                fix.push_back(g);
            }
            if (zamd_inflate_segments_dev(e, g_gpu.d_in, in_n, fix.data(), fix.size(), nullptr, nullptr, 0) != 0) return Z_MEM_ERROR;
            std::vector<zamd_inflate_seg> merged(segs.size() + fix.size());
            std::merge(segs.begin(), segs.end(), fix.begin(), fix.end(), merged.begin(),
                       [](const zamd_inflate_seg &a, const zamd_inflate_seg &b) { return a.in_bit < b.in_bit; });
            segs.swap(merged);
        }
        tr.lap("repair");
        // follow the chain of segments from the known block start
        size_t idx = 0;
        for (;;) {
            zamd_inflate_seg g = segs[idx];
            bool prefix = false;
            if (g.out_len > out_batch && g.out_len > 0 && chain.empty()) {
                // one segment larger than the output batch: take a whole-block prefix of it
                zamd_inflate_seg one = g; one.out_cap = out_batch; one.status = 0;
                if (zamd_inflate_segments_dev(e, g_gpu.d_in, in_n, &one, 1, nullptr, nullptr, 0) != 0) return Z_MEM_ERROR;
                if (one.out_len == 0) { strm->msg = "deflate block too large for the GPU engine"; s->mode = IM_MEM; return Z_MEM_ERROR; }
                g = one; g.status = 3; prefix = true;
            }
            if ((produced + g.out_len > out_batch || chain.size() + ZAMD_INF_MARKS > 32768) && !chain.empty()) break;
            if (g.end_bit > g.in_bit || g.status == 1) {
                // the segment's block marks cut it into parts; marks that are out of order or outside the segment end the list
                const unsigned long long *row = nullptr;
                if (!prefix && !mark_bits.empty()) {
                    const size_t r = (size_t)(std::lower_bound(mark_bits.begin(), mark_bits.end(), g.in_bit) - mark_bits.begin());
                    if (r < mark_bits.size() && mark_bits[r] == g.in_bit) row = &marks[r * ZAMD_INF_MARKS];
                }
                uint64_t pbit = g.in_bit, pout = 0;
                auto part = [&](uint64_t bit_end, uint64_t out_end, uint32_t status) {
                    zamd_inflate_seg q = g;
                    q.in_bit = pbit; q.in_end_bit = q.end_bit = bit_end; q.out_off = produced + pout; q.out_cap = q.out_len = out_end - pout;
                    q.win_avail = (uint32_t)((s->hist.size() + q.out_off) < HIST ? (s->hist.size() + q.out_off) : HIST);
                    q.mode = ZAMD_INF_EXACT; q.status = status;
                    chain.push_back(q);
                    pbit = bit_end; pout = out_end;
                };
                for (unsigned k = 0; row && k < ZAMD_INF_MARKS; k++) {
                    const uint64_t bit = g.in_bit + (row[k] >> 32), off = row[k] & 0xffffffffull;
                    if (row[k] == 0 || off <= pout || off >= g.out_len || bit <= pbit || bit >= g.end_bit) break;
                    part(bit, off, 0);
                    cut = true;
                }
                part(g.end_bit, g.out_len, g.status);
                produced += g.out_len; end_bit = g.end_bit;
            }
            if (g.status >= 16) {
                // valid blocks in front of the error are delivered first; the error (or, for a guess that hit
                // its output bound, the rest of the segment) comes back on the next attempt
                if (chain.empty()) return ibad(strm, s, zamd_inflate_message(g.status));
                break;
            }
            if (g.status == 1) { final_block = true; break; }
            if (g.status != 2) break;                            // input (or the output batch) ran out inside this segment
            // stopped at a sync marker: the next segment is the candidate that starts there
            uint64_t nb = g.end_bit;
            size_t lo = idx + 1, hi = segs.size();
            while (lo < hi) { size_t mid = (lo + hi) / 2; if (segs[mid].in_bit < nb) lo = mid + 1; else hi = mid; }
            if (lo >= segs.size() || segs[lo].in_bit != nb) break;
            idx = lo;
        }
    }
    if (chain.empty()) { if (view) return NO_PROGRESS; s->tried = in_n; return Z_OK; }
    if (!hinted) { drop_hints(s); }      // progress off the chunk grid: an index no longer applies
    tr.lap("chain");

    // Large rounds whose segments are cut into parts (indexed chunks that come with split points, counted segments with their
    // block marks) and can go straight into the caller's buffer are placed in groups whose transfers overlap the next group's
    // kernels (place_piped); the input of indexed chunks is uploaded group by group as well.
    // (A group's kernels take as long as its slowest segment: whole chunks, at some 60 ms each, make groups a loss.)
    // (A group's kernels take some 20 ms at 512 MiB and 27 at 1 GiB, its download 12 ms per 512 MiB: large rounds take large
    // groups, 3.5 GiB in groups of 512 MiB 139 ms, of 1 GiB 128.)
    static const uint64_t group_env = env_u64("ZAMD_Z_INF_GROUP_MIB", 0) << 20;
    const uint64_t group_out = group_env ? group_env : produced >= (3ull << 30) ? (1ull << 30) : (512ull << 20);
    const bool piped = (hinted ? s->hint_per != 0 : cut) && produced >= 2 * group_out && chain.size() >= 2 && s->out.size() == s->out_pos && strm->avail_out >= produced &&
                       group_out >= HIST && !getenv("ZAMD_Z_NO_PIPELINE");
    if (produced && piped) {
        size_t good = 0; uint32_t bad = 0;
        const uint32_t check0 = s->check;
        if (pre_n && need_input() != Z_OK) return Z_MEM_ERROR;        // two pieces on the host: the groups take their input from the device
        int rc = place_piped(e, s, ip, in_n, on_device ? g_gpu.d_in : nullptr, chain, strm->next_out, group_out, &good, &bad, tr);
        if (rc != 0) { s->check = check0; return Z_MEM_ERROR; }
        if (good == 0) {
            if (hinted) { drop_hints(s); return HINT_MISFIT; }
            return ibad(strm, s, zamd_inflate_message(bad));
        }
        if (good < chain.size()) {
            // what lies in front of the failure is delivered (with an index: the whole chunks in front of it); the failure
            // comes back on the next attempt, without the index should it be the index that did not fit
            if (hinted) {
                size_t k = 0;
                while (k < chunk_end.size() && chunk_end[k] <= good) k++;
                if (k == 0) { drop_hints(s); return HINT_MISFIT; }
                good = chunk_end[k - 1];
                hints_used = 2 * k;
                s->hints.resize(s->hint_pos + hints_used);
                if (s->hint_per) s->hint_splits.resize((s->hint_pos + hints_used) / 2 * s->hint_per);
            }
            chain.resize(good);
            final_block = false;
        }
        produced = chain.back().out_off + chain.back().out_cap;
        end_bit = hinted ? chain.back().in_end_bit : chain.back().end_bit;
        if (good == chain.size() && chain.back().status == 1) final_block = true;
        if (hinted) { s->hint_pos += hints_used; s->hinted_chunks += hints_used / 2; }
        if (s->wrap) strm->adler = s->check;
        const uint8_t *got = strm->next_out;
        strm->next_out += produced; strm->avail_out -= (uInt)produced; strm->total_out += produced;
        s->total += produced;
        const size_t wlim = s->wbits >= 8 && s->wbits < 15 ? (size_t)1 << s->wbits : HIST;
        s->hist.assign(got + produced - wlim, got + produced);
    } else if (produced) {
        if (need_input() != Z_OK) return Z_MEM_ERROR;
        if (grow(e, &g_gpu.d_out, &g_gpu.out_cap, produced + 64) != Z_OK) return Z_MEM_ERROR;
        uint8_t *d_head = nullptr;
        if (!s->hist.empty()) {
            if (grow(e, (uint8_t **)&g_gpu.d_off, &g_gpu.off_cap, HIST) != Z_OK) return Z_MEM_ERROR;
            d_head = (uint8_t *)g_gpu.d_off;
            if (zamd_dev_upload(e, d_head + (HIST - s->hist.size()), s->hist.data(), s->hist.size()) != 0) return Z_MEM_ERROR;
        }
        if (zamd_inflate_segments_dev(e, g_gpu.d_in, in_n, chain.data(), chain.size(), d_head, g_gpu.d_out, produced) != 0)
            return Z_MEM_ERROR;
        for (const zamd_inflate_seg &g : chain)
            if (g.status >= 3) {
                if (hinted) { drop_hints(s); return HINT_MISFIT; }
                return ibad(strm, s, zamd_inflate_message(g.status));
            }
        if (hinted) {
            // the final block may only close the last chunk taken
            for (size_t i = 0; i + 1 < chain.size(); i++)
                if (chain[i].status == 1) { drop_hints(s); return HINT_MISFIT; }
            final_block = chain.back().status == 1;
            { s->hint_pos += hints_used; s->hinted_chunks += hints_used / 2; }
        }
        tr.lap("place");
        if (s->wrap) {
            uint32_t v = s->check;
            int rc = s->gzip ? zamd_crc32_dev(e, v, g_gpu.d_out, produced, &v) : zamd_adler32_dev(e, v, g_gpu.d_out, produced, &v);
            if (rc != 0) return Z_MEM_ERROR;
            strm->adler = s->check = v;
        }
        tr.lap("check");
        const uint8_t *got;
        if (s->out.size() == s->out_pos && strm->avail_out >= produced) {
            // nothing pending and room enough: straight into the caller's buffer
            if (download(e, strm->next_out, g_gpu.d_out, produced) != 0) return Z_MEM_ERROR;
            got = strm->next_out;
            strm->next_out += produced; strm->avail_out -= (uInt)produced; strm->total_out += produced;
        } else {
            size_t at = s->out.size();
            s->out.resize(at + produced);
            if (download(e, s->out.data() + at, g_gpu.d_out, produced) != 0) return Z_MEM_ERROR;
            got = s->out.data() + at;
        }
        tr.lap("download");
        s->total += produced;
        // the window carried from round to round holds 1 << windowBits bytes (updatewindow, inflate.c:379-427): a
        // distance that reaches further back than that into an earlier round is "too far back" (inflate.c:1122-1134)
        const size_t wlim = s->wbits >= 8 && s->wbits < 15 ? (size_t)1 << s->wbits : HIST;
        if (produced >= wlim) s->hist.assign(got + produced - wlim, got + produced);
        else {
            s->hist.insert(s->hist.end(), got, got + produced);
            if (s->hist.size() > wlim) s->hist.erase(s->hist.begin(), s->hist.end() - wlim);
        }
    }
    // drop the consumed input
    size_t drop = (size_t)(end_bit >> 3);
    if (final_block) { drop = (size_t)((end_bit + 7) >> 3); s->bit_off = 0; s->last = true; s->mode = IM_TRAILER; }
    else s->bit_off = (uint32_t)(end_bit & 7);
    if (view) {
        const size_t from_buf = drop < pre_n ? drop : pre_n, from_caller = drop - from_buf;
        if (from_buf) { s->in.erase(s->in.begin(), s->in.begin() + from_buf); s->cands.clear(); s->scanned = 0; }   // what the buffer's own rounds knew about it
        strm->next_in += from_caller; strm->avail_in -= (uInt)from_caller; strm->total_in += from_caller;
    } else {
        s->in.erase(s->in.begin(), s->in.begin() + drop);
        if (s->in.empty()) s->tail_partial = 0;
        Longs keep(cands.get_allocator());
        for (uint64_t c : cands) if (c > drop) keep.push_back(c - drop);
        cands.swap(keep);
        scanned = scanned > drop ? scanned - drop : 0;
    }
    s->tried = 0;
    tr.lap("tidy"); tr.done("inflate round");
    return Z_OK;
}

int istate_reset(z_streamp strm, IState *s, bool keep_window) {
    strm->total_in = strm->total_out = 0;
    s->total = 0;
    strm->msg = Z_NULL;
    if (s->wrap) strm->adler = s->wrap & 1;
    s->mode = s->wrap ? IM_HEAD : IM_BODY;
    s->last = s->havedict = s->gzip = s->synced = false;
    s->head = Z_NULL;
    s->hdr.clear(); s->in.clear(); s->cands.clear(); s->out.clear();
    s->bit_off = 0; s->scanned = 0; s->tried = 0; s->out_pos = 0; s->check = 0; s->dictid = 0;
    s->prime_hold = 0; s->prime_bits = 0; s->tail_partial = 0;
    drop_hints(s);
    if (!keep_window) s->hist.clear();
    return Z_OK;
}

}  // namespace

extern "C" {

int inflateResetKeep(z_streamp strm) {
    IState *s = istate(strm);
    if (!s) return Z_STREAM_ERROR;
    return istate_reset(strm, s, true);
}

int inflateReset(z_streamp strm) {
    IState *s = istate(strm);
    if (!s) return Z_STREAM_ERROR;
    return istate_reset(strm, s, false);
}

int inflateReset2(z_streamp strm, int windowBits) {
    IState *s = istate(strm);
    if (!s) return Z_STREAM_ERROR;
    int wrap;
    if (windowBits < 0) { wrap = 0; windowBits = -windowBits; }
    else { wrap = (windowBits >> 4) + 1; if (windowBits < 48) windowBits &= 15; }
    if (windowBits && (windowBits < 8 || windowBits > 15)) return Z_STREAM_ERROR;
    s->wrap = wrap;
    s->wbits = (unsigned)windowBits;
    return inflateReset(strm);
}

int inflateInit2_(z_streamp strm, int windowBits, const char *version, int stream_size) {
    if (version == Z_NULL || version[0] != ZLIB_VERSION[0] || stream_size != (int)sizeof(z_stream)) return Z_VERSION_ERROR;
    if (strm == Z_NULL) return Z_STREAM_ERROR;
    strm->msg = Z_NULL;
    if (strm->zalloc == (alloc_func)0) { strm->zalloc = default_alloc; strm->opaque = (voidpf)0; }
    if (strm->zfree == (free_func)0) strm->zfree = default_free;
    void *mem = strm->zalloc(strm->opaque, 1, (uInt)sizeof(IState));
    if (mem == Z_NULL) return Z_MEM_ERROR;
    IState *s = new (mem) IState(strm);
    s->magic = INFLATE_MAGIC;
    // A round costs what its slowest segment costs, however many segments it has (up to the GPU's 6 000 wavefront
    // slots): large rounds are cheap rounds.
    s->out_batch = env_u64("ZAMD_Z_OUT_BATCH_MIB", 4096) << 20;
    strm->state = (struct internal_state *)s;
    int ret = inflateReset2(strm, windowBits);
    if (ret != Z_OK) {
        s->magic = 0; s->~IState();
        strm->zfree(strm->opaque, mem);
        strm->state = Z_NULL;
    }
    return ret;
}

int inflateInit_(z_streamp strm, const char *version, int stream_size) { return inflateInit2_(strm, 15, version, stream_size); }

int inflateEnd(z_streamp strm) {
    IState *s = istate(strm);
    if (!s || strm->zfree == (free_func)0) return Z_STREAM_ERROR;
    s->magic = 0;
    s->~IState();
    strm->zfree(strm->opaque, s);
    strm->state = Z_NULL;
    return Z_OK;
}

static int inflate_body(z_streamp strm, int flush) {
    IState *s = istate(strm);
    if (!s || strm->next_out == Z_NULL || (strm->next_in == Z_NULL && strm->avail_in != 0)) return Z_STREAM_ERROR;
    if (s->mode == IM_MEM) return Z_MEM_ERROR;
    const uInt in0 = strm->avail_in, out0 = strm->avail_out;
    int ret = Z_OK;
    s->call_taken = 0;
    Trace tc;
    struct Done { Trace &t; ~Done() { t.done("inflate call"); } } done_tc = {tc};
    for (;;) {
        idrain(strm, s);
        tc.lap("drain");
        if (s->mode == IM_TRAILER) {
            // Checked in the same call that decoded the final block, so that bytes
            // behind the stream can be handed back to that call's next_in.
            size_t need = s->wrap ? (s->gzip ? 8 : 4) : 0;
            while (s->in.size() < need && strm->avail_in != 0) {
                s->in.push_back(*strm->next_in++); strm->avail_in--; strm->total_in++; s->call_taken++;
            }
            if (need && s->in.size() >= 4) {               // the check value is compared as soon as it is there (inflate.c:1176-1192)
                const uint8_t *t = s->in.data();
                uint32_t got = s->gzip ? ((uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24))
                                       : (((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3]);
                if (got != s->check) { ibad(strm, s, "incorrect data check"); need = 4; }
            }
            if (s->in.size() < need) break;
            if (s->mode != IM_BAD && s->gzip) {
                const uint8_t *t = s->in.data();
                uint32_t isize = (uint32_t)t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
                if (isize != (uint32_t)(s->total & 0xffffffffu)) ibad(strm, s, "incorrect length check");
            }
            // what follows the stream goes back to the caller: every such byte came in with this call, because a call
            // that returns with output pending hands its unused input back (below)
            unread_tail(strm, s, s->in.size() - need);
            s->in.clear(); s->cands.clear(); s->scanned = 0;
            if (s->mode != IM_BAD) s->mode = IM_DONE;
            continue;
        }
        if (s->out.size() != s->out_pos) break;            // caller must make room first; no input is taken meanwhile
        if (s->mode == IM_DONE) { ret = Z_STREAM_END; break; }
        if (s->mode == IM_BAD) { ret = Z_DATA_ERROR; break; }
        if (s->mode == IM_DICT) {
            if (!s->havedict) return Z_NEED_DICT;
            strm->adler = s->check = 1;
            s->mode = IM_BODY;
        }
        if (s->mode == IM_HEAD) {
            int r = 0;
            while (strm->avail_in != 0 && r == 0) {
                s->hdr.push_back(*strm->next_in++); strm->avail_in--; strm->total_in++;
                r = parse_header(strm, s);
            }
            if (r == 0) break;
            continue;
        }
        if (s->mode == IM_BODY) {
            if (s->prime_bits) {
                // primed bits become the top bits of the bytes in front of the input
                const unsigned nb = (s->prime_bits + 7) >> 3, off = nb * 8 - s->prime_bits;
                const uint64_t v = s->prime_hold << off;
                for (unsigned i = 0; i < nb; i++) s->in.push_back((uint8_t)(v >> (8 * i)));
                s->bit_off = off; s->prime_bits = 0; s->prime_hold = 0;
            }
            // No room at all, and the chunk index says that what comes next inflates to something: a round now would only
            // park a chunk or two in the stream's own buffer (a round of one chunk takes 9 ms, as long as one of a
            // hundred); the caller's next call, with room, takes them along.  (Without the index the rest may be an
            // empty last block and the trailer, which must not wait for room: inflate.c:1165-1210 needs none either.)
            if (strm->avail_out == 0 && s->bit_off == 0 && s->hint_pos + 1 < s->hints.size() && s->hints[s->hint_pos + 1] != 0) break;
            // The same without an index when this call has filled the room it was given and a megabyte or more of the
            // caller's input is left (the end of a stream is a few bytes): that input stays with the caller, whose next
            // call brings room.  (A call that comes without room is served as before.)
            if (strm->avail_out == 0 && out0 != 0 && s->in.empty() && strm->avail_in >= (1u << 20)) break;
            bool tried_all = false;
            if ((s->in.empty() || (s->in.size() < (4u << 20) && strm->avail_in >= (8u << 20) && !s->tail_partial)) && strm->avail_in >= (1u << 20)) {
                // a large input is decoded where it lies; what a round does not use stays with the caller
                static const size_t view_cap = (size_t)env_u64("ZAMD_Z_IN_VIEW_MIB", 2048) << 20;
                const size_t n = strm->avail_in < view_cap ? strm->avail_in : view_cap;
                int rc = decode_some(strm, s, strm->next_in, n);
                tc.lap("round on the caller's input");
                if (rc == Z_MEM_ERROR) { s->mode = IM_MEM; return Z_MEM_ERROR; }
                if (rc != NO_PROGRESS) continue;
                tried_all = n == strm->avail_in;            // not one whole block in there: buffer it and wait for more
            }
            if (s->tail_partial && strm->avail_in) {
                strm->msg = "input behind inflatePrime bits that do not complete a byte";
                return Z_STREAM_ERROR;
            }
            const size_t before = s->in.size();
            uint64_t room = (512ull << 20) > before ? (512ull << 20) - before : 0;
            uint64_t take = strm->avail_in < room ? strm->avail_in : room;
            if (take) {
                s->in.insert(s->in.end(), strm->next_in, strm->next_in + take);
                strm->next_in += take; strm->avail_in -= (uInt)take; strm->total_in += take; s->call_taken += take;
            }
            if (tried_all) s->tried = s->in.size();
            tc.lap("input buffered");
            if (s->in.size() == s->tried) break;             // nothing new since the last fruitless attempt
            int rc = decode_some(strm, s);
            tc.lap("round on buffered input");
            if (rc == Z_MEM_ERROR) { s->mode = IM_MEM; return Z_MEM_ERROR; }
            continue;
        }
        break;
    }
    // A round can stop before the buffered input is used up (output batch, segment count): the call then returns with
    // output waiting, and whatever it took and did not decode goes back to the caller.  The end of the stream may be
    // in there, with bytes of whatever follows it, and those can only be returned by the call that took them.
    if (s->out.size() != s->out_pos && s->mode == IM_BODY) unread_tail(strm, s, s->in.size());
    strm->data_type = (int)s->bit_off + (s->last ? 64 : 0) + (s->mode == IM_BODY ? 128 : 0);
    if (((in0 == strm->avail_in && out0 == strm->avail_out) || flush == Z_FINISH) && ret == Z_OK) ret = Z_BUF_ERROR;
    return ret;
}
int inflate(z_streamp strm, int flush) { return guarded(strm, [&] { return inflate_body(strm, flush); }); }

static int inflateSetDictionary_body(z_streamp strm, const Bytef *dictionary, uInt dictLength) {
    IState *s = istate(strm);
    if (!s || (dictionary == Z_NULL && dictLength)) return Z_STREAM_ERROR;
    if (s->wrap != 0 && s->mode != IM_DICT) return Z_STREAM_ERROR;
    if (s->mode == IM_DICT) {
        uLong id = adler32(1L, dictionary, dictLength);
        if (id != s->check) return Z_DATA_ERROR;
    }
    s->hist.insert(s->hist.end(), dictionary, dictionary + dictLength);
    const size_t wlim = s->wbits >= 8 && s->wbits < 15 ? (size_t)1 << s->wbits : HIST;
    if (s->hist.size() > wlim) s->hist.erase(s->hist.begin(), s->hist.end() - wlim);
    s->havedict = true;
    return Z_OK;
}
int inflateSetDictionary(z_streamp strm, const Bytef *dictionary, uInt dictLength) { return guarded(strm, [&] { return inflateSetDictionary_body(strm, dictionary, dictLength); }); }

// The chunk index of the deflate data that follows (entries as zamd_deflate_index_get hands them out):
// inflate() then places whole chunks directly instead of searching for their boundaries.  A hint that
// does not fit the data costs time, not correctness: it is dropped and the data decoded without it.
static int zamd_inflate_index_hint_body(z_streamp strm, const unsigned *pairs, unsigned n_chunks) {
    IState *s = istate(strm);
    if (!s || (pairs == Z_NULL && n_chunks)) return Z_STREAM_ERROR;
    if (s->mode != IM_HEAD && !(s->mode == IM_BODY && s->total == 0 && s->in.empty())) return Z_STREAM_ERROR;
    s->hints.assign(pairs, pairs + 2 * (size_t)n_chunks);
    s->hint_pos = 0;
    s->hint_splits.clear(); s->hint_per = 0;
    return Z_OK;
}
int zamd_inflate_index_hint(z_streamp strm, const unsigned *pairs, unsigned n_chunks) { return guarded(strm, [&] { return zamd_inflate_index_hint_body(strm, pairs, n_chunks); }); }
// The split records that go with the index just given (zamd_deflate_index_splits): per_chunk records for each of its chunks.
static int zamd_inflate_index_hint_splits_body(z_streamp strm, const unsigned long long *recs, unsigned per_chunk) {
    IState *s = istate(strm);
    if (!s || recs == Z_NULL || per_chunk == 0 || per_chunk > 15) return Z_STREAM_ERROR;
    if (s->hints.empty() || s->hint_pos != 0 || s->total != 0) return Z_STREAM_ERROR;
    s->hint_splits.assign((const uint64_t *)recs, (const uint64_t *)recs + s->hints.size() / 2 * per_chunk);
    s->hint_per = per_chunk;
    return Z_OK;
}
int zamd_inflate_index_hint_splits(z_streamp strm, const unsigned long long *recs, unsigned per_chunk) { return guarded(strm, [&] { return zamd_inflate_index_hint_splits_body(strm, recs, per_chunk); }); }

unsigned long long zamd_inflate_hinted_chunks(z_streamp strm) { IState *s = istate(strm); return s ? s->hinted_chunks : 0ull; }

int inflateGetDictionary(z_streamp strm, Bytef *dictionary, uInt *dictLength) {
    IState *s = istate(strm);
    if (!s) return Z_STREAM_ERROR;
    if (!s->hist.empty() && dictionary != Z_NULL) memcpy(dictionary, s->hist.data(), s->hist.size());
    if (dictLength != Z_NULL) *dictLength = (uInt)s->hist.size();
    return Z_OK;
}

int inflateGetHeader(z_streamp strm, gz_headerp head) {
    IState *s = istate(strm);
    if (!s || (s->wrap & 2) == 0) return Z_STREAM_ERROR;
    s->head = head;
    head->done = 0;
    return Z_OK;
}

int inflateSync(z_streamp strm) {          // inflate.c:1377-1418
    IState *s = istate(strm);
    if (!s) return Z_STREAM_ERROR;
    if (strm->avail_in == 0 && s->in.empty()) return Z_BUF_ERROR;
    // search the buffered bytes, then the caller's
    Bytes &b = s->in;
    size_t from = s->bit_off ? 1 : 0;
    b.insert(b.end(), strm->next_in, strm->next_in + strm->avail_in);
    size_t taken = strm->avail_in, found = 0;
    for (size_t p = from + 4; p <= b.size(); p++)
        if (b[p - 4] == 0 && b[p - 3] == 0 && b[p - 2] == 0xff && b[p - 1] == 0xff) { found = p; break; }
    strm->next_in += taken; strm->total_in += taken; strm->avail_in = 0;
    if (!found) { b.clear(); s->bit_off = 0; s->cands.clear(); s->scanned = 0; s->tried = 0; return Z_DATA_ERROR; }
    // bytes behind the marker go back to the caller when they came from this call
    size_t behind = b.size() - found;
    size_t give = behind < taken ? behind : taken;
    strm->next_in -= give; strm->avail_in += (uInt)give; strm->total_in -= give;
    Bytes rest(b.begin() + found, b.end() - give, b.get_allocator());
    uLong tin = strm->total_in, tout = strm->total_out;
    istate_reset(strm, s, false);
    strm->total_in = tin; strm->total_out = tout;
    s->in.swap(rest);
    s->mode = IM_BODY;
    return Z_OK;
}

int inflateSyncPoint(z_streamp strm) {
    IState *s = istate(strm);
    if (!s) return Z_STREAM_ERROR;
    return 0;      // input is only ever consumed in whole blocks, never up to the length bytes of a stored block
}

static int inflateCopy_body(z_streamp dest, z_streamp source) {
    IState *ss = istate(source);
    if (dest == Z_NULL || !ss || source->zalloc == (alloc_func)0 || source->zfree == (free_func)0) return Z_STREAM_ERROR;
    void *mem = source->zalloc(source->opaque, 1, (uInt)sizeof(IState));
    if (mem == Z_NULL) return Z_MEM_ERROR;
    IState *ds;
    // dest is written only once the copy exists (inflate.c:1455-1466 frees what it got and leaves dest alone)
    try { ds = new (mem) IState(*ss); }
    catch (...) { source->zfree(source->opaque, mem); throw; }
    memcpy((void *)dest, (const void *)source, sizeof(z_stream));
    dest->state = (struct internal_state *)ds;
    return Z_OK;
}
int inflateCopy(z_streamp dest, z_streamp source) { return guarded(dest, [&] { return inflateCopy_body(dest, source); }); }

static int inflatePrime_body(z_streamp strm, int bits, int value) {     // inflate.c:230-250
    IState *s = istate(strm);
    if (!s) return Z_STREAM_ERROR;
    if (bits < 0) { s->prime_hold = 0; s->prime_bits = 0; return Z_OK; }
    if (bits > 16 || s->prime_bits + (unsigned)bits > 32) return Z_STREAM_ERROR;
    if (s->mode != IM_BODY) {
        strm->msg = "inflatePrime is supported on deflate data only";
        return Z_STREAM_ERROR;
    }
    const uint32_t v = (unsigned)value & ((1u << bits) - 1u);
    if (!s->in.empty()) {
        // In the middle of a stream, with input waiting for the rest of its block: the bits go behind that input (the
        // reference puts them into its bit accumulator, in front of the bytes that have not been handed over yet).  The last
        // byte of `in` may then be a partial one; input can follow once further bits have completed it, because stored
        // blocks align to the input's own byte boundaries and one buffer has only one set of those.
        uint32_t left = (uint32_t)bits, val = v;
        while (left) {
            if (s->tail_partial == 0) { s->in.push_back(0); }
            const unsigned have = s->tail_partial, take = 8u - have < left ? 8u - have : left;
            s->in.back() = (uint8_t)(s->in.back() | ((val & ((1u << take) - 1u)) << have));
            s->tail_partial = (have + take) & 7u;
            val >>= take; left -= take;
        }
        s->tried = 0;                      // there is something new to look at
        return Z_OK;
    }
    s->prime_hold += (uint64_t)v << s->prime_bits;
    s->prime_bits += (unsigned)bits;
    return Z_OK;
}
int inflatePrime(z_streamp strm, int bits, int value) { return guarded(strm, [&] { return inflatePrime_body(strm, bits, value); }); }

int inflateUndermine(z_streamp strm, int) {      // inflate.c:1485-1500 without INFLATE_ALLOW_INVALID_DISTANCE_TOOFAR_ARRR
    if (!istate(strm)) return Z_STREAM_ERROR;
    return Z_DATA_ERROR;
}

long inflateMark(z_streamp strm) {
    if (!istate(strm)) return -(1L << 16);
    return -(1L << 16);                          // always between blocks: back == -1, no copy in progress
}

int uncompress(Bytef *dest, uLongf *destLen, const Bytef *source, uLong sourceLen) {    // uncompr.c:24-59
    z_stream stream;
    memset(&stream, 0, sizeof(stream));
    stream.next_in = source;
    stream.avail_in = (uInt)sourceLen;
    if ((uLong)stream.avail_in != sourceLen) return Z_BUF_ERROR;
    stream.next_out = dest;
    stream.avail_out = (uInt)*destLen;
    if ((uLong)stream.avail_out != *destLen) return Z_BUF_ERROR;
    int err = inflateInit_(&stream, ZLIB_VERSION, (int)sizeof(z_stream));
    if (err != Z_OK) return err;
    err = inflate(&stream, Z_FINISH);
    if (err != Z_STREAM_END) {
        // Input is consumed ahead of the output here, so "input used up" alone does not mean
        // a truncated stream: inflated data still waiting for room is the Z_BUF_ERROR case.
        IState *s = istate(&stream);
        bool waiting = s && (s->out.size() != s->out_pos || s->mode == IM_DONE);
        inflateEnd(&stream);
        if (err == Z_NEED_DICT || (err == Z_BUF_ERROR && stream.avail_in == 0 && !waiting)) return Z_DATA_ERROR;
        return err;
    }
    *destLen = stream.total_out;
    return inflateEnd(&stream);
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
// adler32() and crc32() cannot report an error through their return value (zlib.h:1569-1617), and this library can
// fail where the reference cannot (no GPU, no device memory).  A failure is therefore made loud: a line on stderr,
// a sticky error code (zamd_z_last_error), and the caller's running value comes back unchanged.
static std::atomic<int> g_sticky_error{0};
static uLong checksum_failed(int crc, uLong start, int code, const char *what) {
    g_sticky_error = code;
    fprintf(stderr, "zamd_z: %s() failed (%s, code %d): the returned value is the running value it was given, not a checksum\n",
            crc ? "crc32" : "adler32", what, code);
    return start;
}
static uLong checksum_host_buf(int crc, uLong start, const Bytef *buf, uInt len) {
    if (buf == Z_NULL) return crc ? 0UL : 1UL;
    if (len == 0) return start;
    LaneLock lk;
    zamd_engine *e = engine_locked();
    if (!e) return checksum_failed(crc, start, g_gpu.rc ? g_gpu.rc : Z_STREAM_ERROR, "no GPU engine");
    if (grow(e, &g_gpu.d_in, &g_gpu.in_cap, (uint64_t)len + 64) != Z_OK) return checksum_failed(crc, start, Z_MEM_ERROR, "device memory");
    uint32_t v = (uint32_t)start;
    int rc = zamd_dev_upload(e, g_gpu.d_in, buf, len);
    if (rc != 0) return checksum_failed(crc, start, rc, "upload");
    rc = crc ? zamd_crc32_dev(e, v, g_gpu.d_in, len, &v) : zamd_adler32_dev(e, v, g_gpu.d_in, len, &v);
    return rc == 0 ? (uLong)v : checksum_failed(crc, start, rc, "kernel");
}

// 0, or the code of the last failure of a call that has no way to report one (adler32, crc32); reading clears it.
int zamd_z_last_error(void) { return g_sticky_error.exchange(0); }

uLong adler32(uLong adler, const Bytef *buf, uInt len) { return checksum_host_buf(0, adler, buf, len); }
uLong crc32(uLong crc, const Bytef *buf, uInt len) { return checksum_host_buf(1, crc, buf, len); }
uLong adler32_combine(uLong a1, uLong a2, z_off_t len2) { return zamd_adler32_combine((uint32_t)a1, (uint32_t)a2, len2); }
uLong adler32_combine64(uLong a1, uLong a2, z_off64_t len2) { return zamd_adler32_combine((uint32_t)a1, (uint32_t)a2, len2); }
uLong crc32_combine(uLong c1, uLong c2, z_off_t len2) { return zamd_crc32_combine((uint32_t)c1, (uint32_t)c2, len2); }
uLong crc32_combine64(uLong c1, uLong c2, z_off64_t len2) { return zamd_crc32_combine((uint32_t)c1, (uint32_t)c2, len2); }

const z_crc_t *get_crc_table(void) {
    // crc32.c:190-199 hands out all of crc_table[8][256] (crc32.c:90-125): table k < 4 holds the CRC of byte n followed
    // by k zero bytes, tables 4-7 the same values byte-swapped (the word-at-a-time tables of big-endian machines).
    // Callers that slice by four read all of them.
    static z_crc_t table[8][256];
    static std::once_flag once;
    std::call_once(once, [] {
        auto swap = [](uint32_t x) { return (x >> 24) | ((x >> 8) & 0xff00u) | ((x << 8) & 0xff0000u) | (x << 24); };
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1)));
            table[0][n] = c;
        }
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = table[0][n];
            table[4][n] = swap(c);
            for (int k = 1; k < 4; k++) {
                c = table[0][c & 0xffu] ^ (c >> 8);            // one more zero byte behind it
                table[k][n] = c; table[k + 4][n] = swap(c);
            }
        }
    });
    return &table[0][0];
}

// ------------------------------------------------------------------ inflateBack (infback.c:28-640)
// The call-back flavour of inflate for raw deflate data: input is pulled with in(), output is handed
// to out() a window at a time.  Here it is a loop around inflate() of this library; the caller's
// window is the output buffer of every inflate() call.
namespace {
const uint32_t BACK_MAGIC = 0x5a414d42u;
struct BState { uint32_t magic; unsigned char *window; unsigned wsize; };
}

int inflateBackInit_(z_streamp strm, int windowBits, unsigned char *window, const char *version, int stream_size) {   // infback.c:28-72
    if (version == Z_NULL || version[0] != ZLIB_VERSION[0] || stream_size != (int)sizeof(z_stream)) return Z_VERSION_ERROR;
    if (strm == Z_NULL || window == Z_NULL || windowBits < 8 || windowBits > 15) return Z_STREAM_ERROR;
    strm->msg = Z_NULL;
    if (strm->zalloc == (alloc_func)0) { strm->zalloc = default_alloc; strm->opaque = (voidpf)0; }
    if (strm->zfree == (free_func)0) strm->zfree = default_free;
    BState *b = (BState *)strm->zalloc(strm->opaque, 1, (uInt)sizeof(BState));
    if (b == Z_NULL) return Z_MEM_ERROR;
    b->magic = BACK_MAGIC; b->window = window; b->wsize = 1u << windowBits;
    strm->state = (struct internal_state *)b;
    return Z_OK;
}

static int inflateBack_body(z_streamp strm, in_func in, void *in_desc, out_func out, void *out_desc) {             // infback.c:250-629
    if (strm == Z_NULL || strm->state == Z_NULL || ((BState *)strm->state)->magic != BACK_MAGIC) return Z_STREAM_ERROR;
    BState *b = (BState *)strm->state;
    strm->msg = Z_NULL;
    z_stream z;
    memset(&z, 0, sizeof z);
    if (inflateInit2_(&z, -15, ZLIB_VERSION, (int)sizeof(z_stream)) != Z_OK) return Z_MEM_ERROR;
    z.next_in = strm->next_in; z.avail_in = strm->next_in != Z_NULL ? strm->avail_in : 0;
    z.next_out = b->window; z.avail_out = b->wsize;
    int ret;
    for (;;) {
        if (z.avail_in == 0) {
            const unsigned char *p = Z_NULL;
            unsigned n = in(in_desc, &p);
            if (n != 0) { z.next_in = p; z.avail_in = n; }
        }
        const bool starved = z.avail_in == 0;
        ret = inflate(&z, Z_NO_FLUSH);
        if (ret == Z_DATA_ERROR || ret == Z_MEM_ERROR || ret == Z_NEED_DICT || ret == Z_STREAM_ERROR) {
            strm->msg = z.msg;
            if (ret == Z_NEED_DICT || ret == Z_STREAM_ERROR) ret = Z_DATA_ERROR;
            break;
        }
        if (z.avail_out == 0 || ret == Z_STREAM_END) {          // a full window, or what is left at the end
            unsigned have = b->wsize - z.avail_out;
            if (have && out(out_desc, b->window, have)) { ret = Z_BUF_ERROR; break; }
            z.next_out = b->window; z.avail_out = b->wsize;
        }
        if (ret == Z_STREAM_END) break;
        if (starved && ret == Z_BUF_ERROR) { z.next_in = Z_NULL; z.avail_in = 0; break; }     // in() has nothing more: infback.c:117-124
    }
    strm->next_in = z.next_in; strm->avail_in = z.avail_in;     // what the stream did not use
    z.msg = Z_NULL;
    inflateEnd(&z);
    return ret;
}
int inflateBack(z_streamp strm, in_func in, void *in_desc, out_func out, void *out_desc) { return guarded(strm, [&] { return inflateBack_body(strm, in, in_desc, out, out_desc); }); }

int inflateBackEnd(z_streamp strm) {                                                                    // infback.c:631-640
    if (strm == Z_NULL || strm->state == Z_NULL || strm->zfree == (free_func)0 || ((BState *)strm->state)->magic != BACK_MAGIC)
        return Z_STREAM_ERROR;
    ((BState *)strm->state)->magic = 0;
    strm->zfree(strm->opaque, strm->state);
    strm->state = Z_NULL;
    return Z_OK;
}

}  // extern "C"
