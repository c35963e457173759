/*
 * zamd_gz.cpp — the gz* file layer of zlib 1.2.8 (gzlib.c, gzread.c, gzwrite.c,
 * gzclose.c) on top of the stream API of this library (zamd_zlib.cpp), so that a
 * program written against gzopen/gzread/gzwrite runs its compression on the GPU.
 *
 * Behaviour follows the reference function by function (cited below); what
 * differs is sizing.  Every deflate()/inflate() call that makes progress is a GPU
 * round trip, so the buffers default to 32 MiB (the reference: 8 KiB, gzguts.h:147)
 * — gzbuffer() and ZAMD_GZ_BUF_MIB change that — and reads hand inflate() the whole
 * input buffer at once.
 *
 * Chunk index (no counterpart in the reference's gz layer; examples/zran.c:67-72 keeps such access points in
 * memory): a file written with an 'I' in the mode string, or with ZAMD_GZ_INDEX=1 in the environment, is
 * followed by empty gzip members whose FEXTRA field lists, per chunk of the deflate data, its compressed and
 * its inflated length.  Any gzip reader skips them (they inflate to nothing); this reader finds them from the
 * end of the file and hands them to inflate(), which then places whole chunks in one pass.  Layout: see
 * zamd_gz_index_load below and the integration notes.
 */
#include "../../include/zamd_zlib.h"

#include <errno.h>
#include <fcntl.h>
#include <limits.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

extern "C" int zamd_gz_index_load(int fd, zamd_gz_index *out);
extern "C" void zamd_gz_index_free(zamd_gz_index *ix);

namespace {

enum { GZ_NONE = 0, GZ_READ = 7247, GZ_WRITE = 31153, GZ_APPEND = 1 };   /* gzguts.h:150-153 */
enum { LOOK = 0, COPY = 1, GZIP = 2 };                                    /* gzguts.h:155-157 */

/* Allocator of the input buffer: resize() leaves new bytes as they are (a buffer grown to a gigabyte for one large
 * gzread() would otherwise be cleared first, 200 ms that the file's bytes then overwrite). */
template <class T> struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { typedef NoInitAlloc<U> other; };
    NoInitAlloc() {}
    template <class U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <class U> void construct(U *p) { ::new ((void *)p) U; }
    template <class U, class A0, class... A> void construct(U *p, A0 &&a0, A &&... a) { ::new ((void *)p) U(std::forward<A0>(a0), std::forward<A>(a)...); }
};

struct GzFile {
    gzFile_s x;                 /* what the gzgetc() macro of zlib.h reads: must come first (gzguts.h:160-162) */
    int mode, fd;
    std::string path;
    unsigned size, want;        /* buffer size in use (0: nothing allocated yet), requested size */
    int hinting;                /* reading: the member being inflated is the one the file's chunk index describes */
    unsigned fill;              /* reading: how much input a refill may hold (0: size); a large gzread() raises it for its own calls */
    std::vector<unsigned char, NoInitAlloc<unsigned char> > in;
    std::vector<unsigned char> out;
    int direct;
    /* reading */
    int how;
    z_off64_t start;
    int eof, past;
    /* writing */
    int level, strategy;
    int want_index, members;    /* chunk index asked for; gzip members finished so far */
    uint64_t base, produced, member_at;   /* file offset at open, deflate() output so far, where the open member started */
    std::vector<unsigned> index; uint64_t index_member;
    std::vector<unsigned long long> isplits; unsigned isplits_per;   /* the member's split records (block boundaries inside its chunks), per chunk */
    unsigned win_every, chunk_bytes;      /* windows wanted in front of every win_every-th chunk (0: none); the chunk grid */
    uint64_t upos;                        /* input bytes of the open member handed to deflate() so far */
    std::vector<unsigned char> wtail;     /* the last 32 KiB of them */
    std::vector<std::vector<unsigned char> > windows;
    /* reading */
    std::vector<unsigned> ridx; uint64_t ridx_member; int have_ridx;
    std::vector<unsigned long long> rsplits; unsigned rsplits_per;   /* split records of the index, rsplits_per to a chunk (0: none) */
    uint64_t ridx_woff; unsigned ridx_wevery, ridx_nwin;   /* where the window members start, their spacing and number */
    uint64_t data_start;                  /* file offset of the indexed member's deflate data (0: not looked up yet) */
    int raw_mode;                         /* inflating from a stored window: raw deflate, the member's trailer is skipped by hand */
    unsigned skip_trailer; uint64_t read_limit;   /* trailer bytes still to skip; compressed bytes the next refills may load (0: no limit) */
    uint64_t out_limit;                   /* and what they inflate to: the next fetch asks for no more (0: no limit) */
    uint64_t window_seeks;
    /* seek request */
    z_off64_t skip; int seek;
    /* error */
    int err; std::string msg;
    z_stream strm;
    /* Writing behind the caller's back: writes of a megabyte or more are collected in one of two buffers of deflate()'s
     * batch size; a full one goes to a thread of this file's own, which runs comp() on it (deflate() takes a whole batch
     * where it lies) while the caller fills the other.  One job at a time.  While a job is out, strm, out, err, msg, the
     * file position and the index fields belong to that thread: the caller's side touches `carry`, `cur`, `behind_job`,
     * x.pos and the seek request only, and everything else after wait_idle(). */
    unsigned char *carry[2]; size_t carry_n[2], carry_cap; int cur;
    std::vector<unsigned char, NoInitAlloc<unsigned char> > wout;   /* room for what a whole batch compresses to: deflate() downloads straight into it */
    bool behind_job;                      /* a job is out and has not been waited for */
    std::thread worker; bool worker_on;
    std::mutex wmu; std::condition_variable wcv;
    const unsigned char *job_p; size_t job_n; bool job_ready, job_done, job_out, wquit;   /* under wmu; job_out: handed over and not finished */
    /* Reading ahead of the caller: a reader that has taken AHEAD_AFTER bytes in small requests gets its data inflated
     * ahead_bytes at a time (a round of that size runs at several times the speed of one buffer's worth) by the same
     * kind of thread, into one of two buffers, while it copies from the other.  While a job is out, strm, in, fill, eof,
     * how and the limits belong to the thread; it leaves its result in ra_n / ra_rc / ra_err / ra_msg and touches
     * neither x nor err / msg (the gzgetc() macro of zlib.h reads x without a call).  `ra_out`: a job has been started
     * and its result is not taken yet; the caller's side takes it when x.have reaches 0. */
    std::vector<unsigned char, NoInitAlloc<unsigned char> > ra_buf[2];
    unsigned ahead_bytes; uint64_t seq;   /* 0: never; bytes handed out through the buffer since the last positioning */
    bool ra_out; int ra_k, ra_cur;        /* the job's buffer; the buffer x.next points into (-1: the ordinary one) */
    unsigned ra_n; int ra_rc, ra_err; std::string ra_msg;
};
thread_local bool t_ahead = false;        /* this thread is a file's read-ahead thread */

inline GzFile *gz(gzFile f) { return (GzFile *)f; }
inline bool fatal(const GzFile *s) { const int e = t_ahead ? s->ra_err : s->err; return e != Z_OK && e != Z_BUF_ERROR; }

void set_error(GzFile *s, int err, const char *msg)      /* gz_error, gzlib.c:575-615 */
{
    if (t_ahead) {                                      /* an error ahead of the caller: it shows when the caller gets there */
        s->ra_err = err;
        if (msg == NULL) s->ra_msg.clear();
        else if (err == Z_MEM_ERROR) s->ra_msg = "out of memory";
        else s->ra_msg = s->path + ": " + msg;
        return;
    }
    if (err != Z_OK && err != Z_BUF_ERROR) s->x.have = 0;
    s->err = err;
    if (msg == NULL) s->msg.clear();
    else if (err == Z_MEM_ERROR) s->msg = "out of memory";
    else s->msg = s->path + ": " + msg;
}

void wait_idle(GzFile *s);
int ahead_start(GzFile *s);
int ahead_take(GzFile *s, bool more);

void reset(GzFile *s)                                    /* gz_reset, gzlib.c:75-88 */
{
    s->x.have = 0;
    if (s->mode == GZ_READ) { s->eof = 0; s->past = 0; s->how = LOOK; s->skip_trailer = 0; s->read_limit = 0; s->out_limit = 0; }
    s->seek = 0;
    s->ra_out = false; s->ra_cur = -1; s->seq = 0;      /* (the caller has waited for a job that was out) */
    set_error(s, Z_OK, NULL);
    s->x.pos = 0;
    s->strm.avail_in = 0;
}

unsigned default_want()
{
    const char *v = getenv("ZAMD_GZ_BUF_MIB");
    unsigned long mib = v ? strtoul(v, NULL, 10) : 32;
    if (mib < 1) mib = 1;
    if (mib > 1024) mib = 1024;
    return (unsigned)(mib << 20);
}

static gzFile open_file_body(const char *path, int fd, const char *mode)      /* gz_open, gzlib.c:91-265 */
{
    if (path == NULL) return NULL;
    GzFile *s = new (std::nothrow) GzFile();
    if (s == NULL) return NULL;
    memset(&s->x, 0, sizeof s->x);
    memset(&s->strm, 0, sizeof s->strm);
    s->size = 0; s->want = default_want(); s->fill = 0; s->hinting = 0;
    s->mode = GZ_NONE; s->level = Z_DEFAULT_COMPRESSION; s->strategy = Z_DEFAULT_STRATEGY; s->direct = 0;
    s->err = Z_OK; s->how = LOOK; s->eof = s->past = 0; s->seek = 0; s->skip = 0; s->start = 0;
    s->want_index = 0; s->members = 0; s->base = s->produced = s->member_at = s->index_member = 0; s->have_ridx = 0; s->ridx_member = 0; s->isplits_per = 0; s->rsplits_per = 0;
    s->win_every = 0; s->chunk_bytes = 0; s->upos = 0;
    s->carry[0] = s->carry[1] = NULL; s->carry_n[0] = s->carry_n[1] = 0; s->carry_cap = 0; s->cur = 0; s->behind_job = false;
    s->worker_on = false; s->job_p = NULL; s->job_n = 0; s->job_ready = s->job_done = s->job_out = s->wquit = false;
    s->ra_out = false; s->ra_k = 0; s->ra_cur = -1; s->ra_n = 0; s->ra_rc = 0; s->ra_err = Z_OK; s->seq = 0;
    { const char *v = getenv("ZAMD_GZ_AHEAD_MIB"); unsigned long m = v ? strtoul(v, NULL, 10) : 256; s->ahead_bytes = (unsigned)((m > 1024 ? 1024 : m) << 20); }
    s->ridx_woff = 0; s->ridx_wevery = s->ridx_nwin = 0; s->data_start = 0; s->raw_mode = 0; s->skip_trailer = 0; s->read_limit = 0; s->out_limit = 0; s->window_seeks = 0;
    { const char *v = getenv("ZAMD_GZ_INDEX"); if (v && *v && *v != '0') s->want_index = 1; }
    { const char *v = getenv("ZAMD_GZ_WINDOWS"); if (v && atoi(v) > 0) { s->win_every = (unsigned)atoi(v); s->want_index = 1; } }
    int cloexec = 0, exclusive = 0;
    for (; *mode; mode++) {
        if (*mode >= '0' && *mode <= '9') { s->level = *mode - '0'; continue; }
        switch (*mode) {
        case 'r': s->mode = GZ_READ; break;
        case 'w': s->mode = GZ_WRITE; break;
        case 'a': s->mode = GZ_APPEND; break;
        case '+': delete s; return NULL;               /* no reading and writing at once */
        case 'e': cloexec = 1; break;
        case 'x': exclusive = 1; break;
        case 'f': s->strategy = Z_FILTERED; break;
        case 'h': s->strategy = Z_HUFFMAN_ONLY; break;
        case 'R': s->strategy = Z_RLE; break;
        case 'F': s->strategy = Z_FIXED; break;
        case 'T': s->direct = 1; break;
        case 'I': s->want_index = 1; break;            /* this library only: append a chunk index */
        case 'W': s->want_index = 1; if (!s->win_every) s->win_every = 1; break;   /* and the window in front of every chunk */
        default: break;                                /* 'b' and anything unknown */
        }
    }
    if (s->mode == GZ_NONE) { delete s; return NULL; }
    if (s->mode == GZ_READ) {
        if (s->direct) { delete s; return NULL; }      /* transparent reading cannot be forced */
        s->direct = 1;                                 /* what an empty file is */
    }
    s->path = path;
    int oflag = (cloexec ? O_CLOEXEC : 0) |
                (s->mode == GZ_READ ? O_RDONLY
                                    : (O_WRONLY | O_CREAT | (exclusive ? O_EXCL : 0) | (s->mode == GZ_WRITE ? O_TRUNC : O_APPEND)));
#ifdef O_LARGEFILE
    oflag |= O_LARGEFILE;
#endif
    s->fd = fd > -1 ? fd : open(path, oflag, 0666);
    if (s->fd == -1) { delete s; return NULL; }
    const bool append = s->mode == GZ_APPEND;
    if (s->mode == GZ_APPEND) s->mode = GZ_WRITE;
    if (s->mode == GZ_READ) {
        s->start = lseek(s->fd, 0, SEEK_CUR);
        if (s->start == -1) s->start = 0;
        const char *v = getenv("ZAMD_GZ_INDEX");
        if (!(v && *v == '0')) {
            zamd_gz_index ix;
            if (zamd_gz_index_load(s->fd, &ix) == 0) {
                s->ridx.assign(ix.pairs, ix.pairs + 2 * (size_t)ix.n_chunks);
                s->ridx_member = ix.member_offset; s->have_ridx = 1;
                s->ridx_woff = ix.window_offset; s->ridx_wevery = ix.window_every; s->ridx_nwin = ix.n_windows;
                if (ix.splits && ix.splits_per_chunk) { s->rsplits.assign(ix.splits, ix.splits + (size_t)ix.n_chunks * ix.splits_per_chunk); s->rsplits_per = ix.splits_per_chunk; }
                zamd_gz_index_free(&ix);
            }
            lseek(s->fd, s->start, SEEK_SET);
        }
    } else {
        z_off64_t at = lseek(s->fd, 0, append ? SEEK_END : SEEK_CUR);
        s->base = at == -1 ? 0 : (uint64_t)at;
    }
    reset(s);
    return (gzFile)s;
}
gzFile open_file(const char *path, int fd, const char *mode)
{
    try { return open_file_body(path, fd, mode); }
    catch (...) { return NULL; }      /* an allocation failed: the call fails, nothing is thrown at a C caller */
}

/* ------------------------------------------------------------------ reading */
static double now_ms()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}
int load(GzFile *s, unsigned char *buf, unsigned len, unsigned *have)    /* gz_load, gzread.c:20-42 */
{
    ssize_t r = 0;
    *have = 0;
    while (*have < len) {
        r = read(s->fd, buf + *have, len - *have);
        if (r <= 0) break;
        *have += (unsigned)r;
    }
    if (r < 0) { set_error(s, Z_ERRNO, strerror(errno)); return -1; }
    if (r == 0) s->eof = 1;
    return 0;
}

int refill(GzFile *s)                                   /* gz_avail, gzread.c:51-75 */
{
    if (fatal(s)) return -1;
    if (s->eof == 0) {
        z_stream &z = s->strm;
        if (z.avail_in) memmove(s->in.data(), z.next_in, z.avail_in);
        const unsigned room = s->fill > s->size && s->in.size() >= s->fill ? s->fill : s->size;
        unsigned got, want = room - z.avail_in;
        if (s->read_limit && want > s->read_limit) want = (unsigned)s->read_limit;     /* after a seek: just the chunks asked for */
        if (load(s, s->in.data() + z.avail_in, want, &got) == -1) return -1;
        if (s->read_limit) s->read_limit -= got < s->read_limit ? got : s->read_limit;
        z.avail_in += got;
        z.next_in = s->in.data();
    }
    return 0;
}

int look(GzFile *s)                                     /* gz_look, gzread.c:86-165 */
{
    z_stream &z = s->strm;
    if (s->size == 0) {
        try { s->in.resize(s->want); s->out.resize((size_t)s->want << 1); }
        catch (...) { s->in.clear(); s->out.clear(); set_error(s, Z_MEM_ERROR, "out of memory"); return -1; }
        s->size = s->want;
        z.zalloc = Z_NULL; z.zfree = Z_NULL; z.opaque = Z_NULL; z.avail_in = 0; z.next_in = Z_NULL;
        if (inflateInit2(&z, 15 + 16) != Z_OK) {
            s->in.clear(); s->out.clear(); s->size = 0;
            set_error(s, Z_MEM_ERROR, "out of memory");
            return -1;
        }
    }
    while (s->skip_trailer) {                           /* crc and length of a member that was inflated as raw deflate data */
        if (z.avail_in == 0) { if (refill(s) == -1) return -1; if (z.avail_in == 0) break; }
        const unsigned n = z.avail_in < s->skip_trailer ? z.avail_in : s->skip_trailer;
        z.next_in += n; z.avail_in -= n; s->skip_trailer -= n;
    }
    if (z.avail_in < 2) {
        if (refill(s) == -1) return -1;
        if (z.avail_in == 0) return 0;
    }
    if (z.avail_in > 1 && z.next_in[0] == 31 && z.next_in[1] == 139) {
        if (s->have_ridx && s->ridx_nwin) {
            /* the window members and the index behind them hold no data: reading ends where they start */
            const z_off64_t at = lseek(s->fd, 0, SEEK_CUR);
            if (at != -1 && (uint64_t)at - z.avail_in == s->ridx_woff) { z.avail_in = 0; s->eof = 1; s->x.have = 0; lseek(s->fd, 0, SEEK_END); return 0; }
        }
        if (s->raw_mode) { inflateReset2(&z, 15 + 16); s->raw_mode = 0; }
        else inflateReset(&z);
        s->hinting = 0;
        if (s->have_ridx) {
            z_off64_t at = lseek(s->fd, 0, SEEK_CUR);
            if (at != -1 && (uint64_t)at - z.avail_in == s->ridx_member)
                s->hinting = zamd_inflate_index_hint(&z, s->ridx.data(), (unsigned)(s->ridx.size() / 2)) == Z_OK;
            if (s->hinting && s->rsplits_per) (void)zamd_inflate_index_hint_splits(&z, s->rsplits.data(), s->rsplits_per);
        }
        s->how = GZIP; s->direct = 0;
        return 0;
    }
    if (s->direct == 0) {                               /* bytes behind a gzip stream that are no gzip stream: ignored */
        z.avail_in = 0; s->eof = 1; s->x.have = 0;
        return 0;
    }
    s->x.next = s->out.data(); s->ra_cur = -1;          /* not gzip at all: hand the bytes through */
    if (z.avail_in) { memcpy(s->x.next, z.next_in, z.avail_in); s->x.have = z.avail_in; z.avail_in = 0; }
    s->how = COPY; s->direct = 1;
    return 0;
}

const uint64_t AHEAD_AFTER = [] { const char *v = getenv("ZAMD_GZ_AHEAD_AFTER_KIB"); return (v ? strtoull(v, NULL, 10) : 65536ull) << 10; }();
const unsigned AHEAD_PAD = 64;                          /* room in front of a buffer's data for gzungetc() */

int decomp(GzFile *s)                                   /* gz_decomp, gzread.c:172-218 */
{
    z_stream &z = s->strm;
    int ret = Z_OK;
    const unsigned had = z.avail_out;
    static const bool trace = getenv("ZAMD_Z_TRACE") != nullptr;
    do {
        const double t0 = trace ? now_ms() : 0;
        /* (a large request tops up a buffer that holds less than half of its input first: the input then reaches inflate()
         * in one piece, and the round is as large as the request) */
        if ((z.avail_in == 0 || (s->fill && z.avail_in < s->fill / 2 && !s->eof)) && refill(s) == -1) return -1;
        const double t1 = trace ? now_ms() : 0;
        /* inflate() of this library takes input ahead of its output, so an empty input buffer at the end of
         * the file is the end of the data only once inflate() itself has nothing left to hand over */
        const bool starved = z.avail_in == 0;
        const unsigned in0 = z.avail_in, out0 = z.avail_out;
        ret = inflate(&z, Z_NO_FLUSH);
        if (trace) fprintf(stderr, "zamd_gz decomp: refill %.1f ms, inflate %.1f ms (in %u -> %u, out room %u -> %u) rc %d\n", t1 - t0, now_ms() - t1, in0, z.avail_in, out0, z.avail_out, ret);
        if (starved && ret == Z_BUF_ERROR) { set_error(s, Z_BUF_ERROR, "unexpected end of file"); break; }
        if (ret == Z_STREAM_ERROR || ret == Z_NEED_DICT) { set_error(s, Z_STREAM_ERROR, "internal error: inflate stream corrupt"); return -1; }
        if (ret == Z_MEM_ERROR) { set_error(s, Z_MEM_ERROR, "out of memory"); return -1; }
        if (ret == Z_DATA_ERROR) { set_error(s, Z_DATA_ERROR, z.msg == NULL ? "compressed data error" : z.msg); return -1; }
    } while (z.avail_out && ret != Z_STREAM_END);
    if (t_ahead) s->ra_n = had - z.avail_out;
    else { s->x.have = had - z.avail_out; s->x.next = z.next_out - s->x.have; }
    if (ret == Z_STREAM_END) { s->how = LOOK; if (s->raw_mode) s->skip_trailer = 8; }   /* another member may follow */
    return 0;
}

/* inflate() of this library runs at what one call holds: a request of hundreds of megabytes gets input to match (about
 * what it inflates from, the request's own size at most), not one buffer at a time.  Sets s->fill for the decomp() that
 * follows (the caller clears it). */
void size_input(GzFile *s, unsigned len)
{
    z_stream &z = s->strm;
    static const uint64_t big_cap = [] { const char *v = getenv("ZAMD_GZ_READ_MIB"); uint64_t m = v ? strtoull(v, nullptr, 10) : 1024; return (m > 2047 ? 2047 : m) << 20; }();
    /* how much input the request inflates from: by the chunk index when the file has one (inflate() then places
     * exactly those chunks, straight into the caller's memory), else by what the data has compressed to so far.
     * Input beyond the request would be inflated ahead into a buffer of inflate()'s own and copied from there. */
    uint64_t want_in = len;
    const uint64_t done = zamd_inflate_hinted_chunks(&z);
    if (s->have_ridx && s->hinting && done < s->ridx.size() / 2) {
        uint64_t comp = 0, raw = 0;
        for (size_t k = (size_t)done; k < s->ridx.size() / 2 && raw < len; k++) { comp += s->ridx[2 * k]; raw += s->ridx[2 * k + 1]; }
        want_in = comp + 64;
    } else if (z.total_in > (1u << 20) && z.total_out > z.total_in) {
        want_in = (uint64_t)((double)len * ((double)z.total_in / (double)z.total_out) * 1.02) + 65536;
    }
    if (want_in > big_cap) want_in = big_cap;
    if (want_in > s->size && big_cap) {
        if (s->in.size() < want_in) {
            /* a new buffer with some room to spare (the next request's chunks may be a little larger); only the
             * bytes still waiting move over */
            uint64_t cap = want_in + (want_in >> 3);
            if (cap > big_cap) cap = big_cap;
            try {
                std::vector<unsigned char, NoInitAlloc<unsigned char> > bigger((size_t)cap);
                if (z.avail_in) memcpy(bigger.data(), z.next_in, z.avail_in);
                s->in.swap(bigger);
                z.next_in = s->in.data();
            } catch (...) { want_in = s->in.size(); }
        }
        s->fill = (unsigned)want_in;
    }
}

/* `more`: the caller reads in small requests, so reading ahead of it pays */
int fetch(GzFile *s, bool more = true)                  /* gz_fetch, gzread.c:226-253 */
{
    z_stream &z = s->strm;
    if (s->ra_out) {                                    /* what the thread has inflated ahead comes first */
        if (ahead_take(s, more) == -1) return -1;
        if (s->x.have) return 0;
    }
    do {
        switch (s->how) {
        case LOOK:
            if (look(s) == -1) return -1;
            if (s->how == LOOK) return 0;
            break;
        case COPY:
            if (load(s, s->out.data(), s->size << 1, &s->x.have) == -1) return -1;
            s->x.next = s->out.data();
            return 0;
        case GZIP:
            if (more && s->ahead_bytes > (s->size << 1) && s->seq >= AHEAD_AFTER && !s->out_limit && !s->read_limit && s->err == Z_OK && ahead_start(s) == 0) {
                if (ahead_take(s, true) == -1) return -1;
                break;
            }
            z.avail_out = s->size << 1; z.next_out = s->out.data();
            if (s->out_limit) { if (s->out_limit < z.avail_out) z.avail_out = (unsigned)s->out_limit; s->out_limit = 0; }
            if (decomp(s) == -1) return -1;
            s->ra_cur = -1; s->seq += s->x.have;
        }
    } while (s->x.have == 0 && (!s->eof || z.avail_in));
    return 0;
}

/* Nothing more can come: the file is read, the input buffer is empty, and no gzip member is open inside
 * inflate() (or the open one has already been found cut short). */
inline bool at_end(const GzFile *s) { return s->eof && s->strm.avail_in == 0 && (s->how != GZIP || s->err == Z_BUF_ERROR); }

int skip_out(GzFile *s, z_off64_t len)                  /* gz_skip, gzread.c:256-285 */
{
    while (len) {
        if (s->x.have) {
            unsigned n = (z_off64_t)s->x.have > len ? (unsigned)len : s->x.have;
            s->x.have -= n; s->x.next += n; s->x.pos += n; len -= n;
        } else if (!s->ra_out && at_end(s)) break;
        else if (fetch(s) == -1) return -1;
    }
    return 0;
}

/* ------------------------------------------------------------------ writing */
int winit(GzFile *s)                                    /* gz_init, gzwrite.c:15-66 */
{
    z_stream &z = s->strm;
    try { s->in.resize(s->want); if (!s->direct) s->out.resize(s->want); }
    catch (...) { s->in.clear(); s->out.clear(); set_error(s, Z_MEM_ERROR, "out of memory"); return -1; }
    if (!s->direct) {
        z.zalloc = Z_NULL; z.zfree = Z_NULL; z.opaque = Z_NULL;
        if (deflateInit2(&z, s->level, Z_DEFLATED, 15 + 16, 8, s->strategy) != Z_OK) {
            s->in.clear(); s->out.clear();
            set_error(s, Z_MEM_ERROR, "out of memory");
            return -1;
        }
        if (s->want_index) zamd_deflate_index(&z, 1);
        s->chunk_bytes = zamd_deflate_chunk_bytes(&z);
        /* collecting whole batches pays where the caller's pieces are smaller than one (ZAMD_GZ_BEHIND=0: never) */
        const char *v = getenv("ZAMD_GZ_BEHIND");
        const unsigned long long batch = zamd_deflate_batch_bytes(&z);
        s->carry_cap = (v && *v == '0') || batch < 2ull * s->want || batch > (1ull << 30) ? 0 : (size_t)batch;
    }
    s->size = s->want;
    if (!s->direct) { z.avail_out = s->size; z.next_out = s->out.data(); s->x.next = z.next_out; }
    return 0;
}

int write_all(GzFile *s, const unsigned char *p, size_t n)
{
    while (n) {
        ssize_t w = write(s->fd, p, n);
        if (w < 0) { if (errno == EINTR) continue; set_error(s, Z_ERRNO, strerror(errno)); return -1; }
        p += w; n -= (size_t)w;
    }
    return 0;
}

/* The input passes by once: what lies in front of every win_every-th chunk boundary is kept (the windows of the index). */
void note_input(GzFile *s, const unsigned char *p, size_t n)
{
    const uint64_t step = (uint64_t)s->chunk_bytes * s->win_every;
    if (step < 32768) return;
    for (uint64_t b = (s->upos / step + 1) * step; b <= s->upos + n; b += step) {
        const size_t from_p = (size_t)(b - s->upos);            /* bytes of p in front of the boundary */
        std::vector<unsigned char> w(32768);
        if (from_p >= 32768) memcpy(w.data(), p + from_p - 32768, 32768);
        else {
            const size_t need = 32768 - from_p;                 /* the rest is the end of what passed before */
            memcpy(w.data(), s->wtail.data() + s->wtail.size() - need, need);
            memcpy(w.data() + need, p, from_p);
        }
        s->windows.push_back(std::move(w));
    }
    if (n >= 32768) s->wtail.assign(p + n - 32768, p + n);
    else {
        s->wtail.insert(s->wtail.end(), p, p + n);
        if (s->wtail.size() > 32768) s->wtail.erase(s->wtail.begin(), s->wtail.end() - 32768);
    }
    s->upos += n;
}

int comp(GzFile *s, int flush)                          /* gz_comp, gzwrite.c:74-131 */
{
    z_stream &z = s->strm;
    if (s->size == 0 && winit(s) == -1) return -1;
    /* at this point the bytes at next_in are new: a comp() that returns without an error has handed all of them over */
    if (s->win_every && !s->direct && s->members == 0 && z.avail_in) note_input(s, z.next_in, z.avail_in);
    if (s->direct) {
        if (write_all(s, z.next_in, z.avail_in) == -1) return -1;
        z.avail_in = 0;
        return 0;
    }
    int ret = Z_OK;
    unsigned made;
    /* a whole batch (from the thread behind the caller): its output goes to a buffer that holds all of it, so that
     * deflate() downloads into it directly instead of into a vector of its own that is then copied out 32 MiB at a time */
    const bool whole = s->carry_cap && z.avail_in == s->carry_cap && flush == Z_NO_FLUSH && !s->wout.empty();
    if (whole) {
        const size_t have = (size_t)(z.next_out - s->x.next);
        if (have && write_all(s, s->x.next, have) == -1) return -1;
        z.next_out = s->wout.data(); z.avail_out = (uInt)s->wout.size(); s->x.next = z.next_out;
    }
    struct Back {       /* whatever happens, the stream is back on the file's ordinary buffer when comp() returns */
        GzFile *s; bool on;
        ~Back() {
            if (!on) return;
            z_stream &z = s->strm;
            if (z.next_out >= s->wout.data() && z.next_out <= s->wout.data() + s->wout.size()) {
                const size_t have = (size_t)(z.next_out - s->x.next);
                if (have && s->err == Z_OK) (void)write_all(s, s->x.next, have);
                z.avail_out = s->size; z.next_out = s->out.data(); s->x.next = z.next_out;
            }
        }
    } back = {s, whole};
    do {
        /* the buffer goes out when it is full, and on a flush (for Z_FINISH: once the stream has ended) */
        if (z.avail_out == 0 || (flush != Z_NO_FLUSH && (flush != Z_FINISH || ret == Z_STREAM_END))) {
            size_t have = (size_t)(z.next_out - s->x.next);
            if (have && write_all(s, s->x.next, have) == -1) return -1;
            if (z.avail_out == 0) { z.avail_out = s->size; z.next_out = s->out.data(); }
            s->x.next = z.next_out;
        }
        made = z.avail_out;
        ret = deflate(&z, flush);
        /* unlike the reference's, this deflate() can run out of (device) memory or fail in the engine: anything but
         * progress, the end of the stream, or "nothing to do" (Z_BUF_ERROR with the input used up) loses data */
        if (!(ret == Z_OK || ret == Z_STREAM_END || (ret == Z_BUF_ERROR && z.avail_in == 0))) {
            z.avail_in = 0;                             /* the caller's bytes are not ours to come back to */
            if (ret == Z_MEM_ERROR) set_error(s, Z_MEM_ERROR, "out of memory");
            else set_error(s, ret == Z_BUF_ERROR ? Z_BUF_ERROR : Z_STREAM_ERROR, "internal error: deflate stream corrupt");
            return -1;
        }
        made -= z.avail_out;
        s->produced += made;
    } while (made);
    if (flush == Z_FINISH) {
        if (s->want_index) {                            /* the index of the member that just ended, before the reset forgets it */
            const unsigned *pairs; unsigned n;
            s->index.clear();
            if (s->members == 0 && zamd_deflate_index_get(&z, &pairs, &n) == Z_OK) {
                s->index.assign(pairs, pairs + 2 * (size_t)n); s->index_member = s->base + s->member_at;
                const unsigned long long *recs; unsigned per = 0;
                s->isplits.clear(); s->isplits_per = 0;
                if (zamd_deflate_index_splits(&z, &recs, &per) == Z_OK && per) { s->isplits.assign(recs, recs + (size_t)per * n); s->isplits_per = per; }
            }
        }
        s->members++;
        s->member_at = s->produced;
        deflateReset(&z);                               /* the next write starts a new member */
        if (s->want_index) zamd_deflate_index(&z, 1);
    }
    return 0;
}

/* ------------------------------------------------------------------ the thread that compresses behind the caller */
void worker_loop(GzFile *s)
{
    for (;;) {
        const unsigned char *p; size_t n;
        {
            std::unique_lock<std::mutex> g(s->wmu);
            s->wcv.wait(g, [&] { return s->job_ready || s->wquit; });
            if (s->wquit) return;
            s->job_ready = false; p = s->job_p; n = s->job_n;
        }
        if (s->mode == GZ_READ) {
            /* ahead_bytes more of the inflated data into the job's buffer, the way a large gzread() gets them */
            t_ahead = true;
            s->ra_n = 0; s->ra_err = Z_OK; s->ra_msg.clear();
            try {
                /* With the chunk index the request ends where a chunk ends: what inflate() still holds of the chunks it
                 * has placed, and whole chunks behind that.  A request that ends inside a chunk costs a round of its own
                 * for that one chunk (some 9 ms), and so does every request after it. */
                unsigned len = s->ahead_bytes;
                z_stream &z = s->strm;
                const uint64_t done = zamd_inflate_hinted_chunks(&z);
                if (s->have_ridx && s->hinting && !s->raw_mode && done < s->ridx.size() / 2) {
                    uint64_t placed = 0;
                    for (size_t k = 0; k < (size_t)done; k++) placed += s->ridx[2 * k + 1];
                    if (placed >= z.total_out && placed - z.total_out < len) {
                        uint64_t acc = placed - z.total_out;
                        for (size_t k = (size_t)done; k < s->ridx.size() / 2 && acc + s->ridx[2 * k + 1] <= s->ahead_bytes; k++) acc += s->ridx[2 * k + 1];
                        if (acc) len = (unsigned)acc;
                    }
                }
                size_input(s, len);
                s->strm.avail_out = len; s->strm.next_out = s->ra_buf[s->ra_k].data() + AHEAD_PAD;
                s->ra_rc = decomp(s);
            } catch (...) { s->ra_rc = -1; s->ra_n = 0; s->ra_err = Z_MEM_ERROR; s->ra_msg = "out of memory"; }
            s->fill = 0;
        } else {
            s->strm.next_in = (Bytef *)p; s->strm.avail_in = (uInt)n;
            try { (void)comp(s, Z_NO_FLUSH); }
            catch (...) { s->strm.avail_in = 0; set_error(s, Z_MEM_ERROR, "out of memory"); }
        }
        { std::lock_guard<std::mutex> g(s->wmu); s->job_done = true; s->job_out = false; }
        s->wcv.notify_all();
    }
}
/* A process that ends with such files still open (exit() without gzclose()) must not take the GPU runtime down under a
 * job that is still running: the files that have a thread are listed, and exit() waits for their jobs. */
std::mutex g_files_mu;
std::vector<GzFile *> g_files;
std::atomic<bool> g_forked{false};      /* this process is a fork()ed child: the threads and their jobs stayed with the parent */
void in_forked_child() { g_forked.store(true); }
void wait_at_exit()
{
    if (g_forked.load()) return;
    const bool trace = getenv("ZAMD_Z_TRACE") != nullptr;
    std::lock_guard<std::mutex> l(g_files_mu);
    for (GzFile *s : g_files) {
        std::unique_lock<std::mutex> g(s->wmu);
        if (trace) fprintf(stderr, "zamd_gz exit: %s has a job out: %d\n", s->path.c_str(), (int)s->job_out);
        s->wcv.wait(g, [&] { return !s->job_out; });
        if (trace) fprintf(stderr, "zamd_gz exit: %s is idle\n", s->path.c_str());
    }
}
void start_worker(GzFile *s)                            /* throws what std::thread and the list throw */
{
    /* the GPU runtime is set up first, on this thread: what it registers for the end of the process then runs after
     * the hook below (handlers run in the reverse order), not under a job that is still using it */
    if (zamd_z_engine_ready() != Z_OK) throw std::bad_alloc();
    std::lock_guard<std::mutex> l(g_files_mu);
    static const bool hooked = atexit(wait_at_exit) == 0 && pthread_atfork(NULL, NULL, in_forked_child) == 0;
    (void)hooked;
    g_files.push_back(s);
    try { s->worker = std::thread(worker_loop, s); }
    catch (...) { g_files.pop_back(); throw; }
    s->worker_on = true;
}
/* After this the file's state is the caller's to read again. */
void wait_idle(GzFile *s)
{
    if (!s->behind_job) return;
    std::unique_lock<std::mutex> g(s->wmu);
    s->wcv.wait(g, [&] { return s->job_done; });
    s->behind_job = false;
}
/* The same without waiting: true when no job is out (any more). */
bool idle(GzFile *s)
{
    if (!s->behind_job) return true;
    std::lock_guard<std::mutex> g(s->wmu);
    if (!s->job_done) return false;
    s->behind_job = false;
    return true;
}
inline bool behind(const GzFile *s) { return s->behind_job || s->carry_n[s->cur] != 0; }
/* Hand the buffer in use to the thread (when the one before is done) and go on in the other one.  -1: the file is in
 * error (the buffer's bytes are dropped like everything else that is written then). */
int submit(GzFile *s)
{
    wait_idle(s);
    const int k = s->cur;
    const size_t n = s->carry_n[k];
    s->carry_n[k] = 0;
    if (s->err != Z_OK) return -1;
    if (n == 0) return 0;
    if (!s->worker_on) {
        try { start_worker(s); }
        catch (...) { s->worker_on = false; }
    }
    if (!s->worker_on || !s->carry[k ^ 1]) {
        /* no thread, or no second buffer yet and none to be had: this thread does the work */
        if (s->worker_on && (s->carry[k ^ 1] = (unsigned char *)malloc(s->carry_cap)) != NULL) { /* go on below */ }
        else {
            s->strm.next_in = s->carry[k]; s->strm.avail_in = (uInt)n;
            return comp(s, Z_NO_FLUSH);
        }
    }
    if (s->wout.empty() && n == s->carry_cap) {
        try { s->wout.resize((size_t)deflateBound(&s->strm, (uLong)s->carry_cap) + 4096); }
        catch (...) { /* then through the ordinary buffer */ }
    }
    { std::lock_guard<std::mutex> g(s->wmu); s->job_p = s->carry[k]; s->job_n = n; s->job_ready = true; s->job_done = false; s->job_out = true; }
    s->behind_job = true;
    s->wcv.notify_all();
    s->cur = k ^ 1;
    return 0;
}
/* Everything collected so far goes through deflate(), and the state is the caller's again.  -1: the file is in error. */
int settle(GzFile *s)
{
    if (!behind(s)) return s->err == Z_OK ? 0 : -1;
    int rc = submit(s);
    wait_idle(s);
    return rc == -1 || s->err != Z_OK ? -1 : 0;
}
/* Reading: start a job into the buffer the caller is not copying from.  The file must be idle.  -1: no thread or no
 * memory to be had (the caller reads on the ordinary way). */
int ahead_start(GzFile *s)
{
    const int k = s->ra_cur == 0 ? 1 : 0;
    try {
        if (s->ra_buf[k].size() < (size_t)s->ahead_bytes + AHEAD_PAD) s->ra_buf[k].resize((size_t)s->ahead_bytes + AHEAD_PAD);
        if (!s->worker_on) start_worker(s);
    } catch (...) { return -1; }
    s->ra_k = k;
    { std::lock_guard<std::mutex> g(s->wmu); s->job_ready = true; s->job_done = false; s->job_out = true; }
    s->behind_job = true; s->ra_out = true;
    s->wcv.notify_all();
    return 0;
}
/* Take the job's result (waiting for it if need be): its bytes become what the caller reads next, its error the
 * file's.  `more`: the next job starts right away.  -1: the job failed. */
int ahead_take(GzFile *s, bool more)
{
    wait_idle(s);
    s->ra_out = false;
    if (s->ra_err != Z_OK) {
        s->err = s->ra_err; s->msg = s->ra_msg;
        if (fatal(s)) s->x.have = 0;
    }
    if (s->ra_rc == -1) return -1;
    s->x.next = s->ra_buf[s->ra_k].data() + AHEAD_PAD; s->x.have = s->ra_n; s->ra_cur = s->ra_k;
    s->seq += s->ra_n;
    if (more && s->ra_n && s->how == GZIP && s->err == Z_OK) (void)ahead_start(s);
    return 0;
}

void stop_worker(GzFile *s)
{
    wait_idle(s);
    if (s->worker_on) {
        { std::lock_guard<std::mutex> g(s->wmu); s->wquit = true; }
        s->wcv.notify_all();
        s->worker.join();
        s->worker_on = false;
        std::lock_guard<std::mutex> l(g_files_mu);
        for (size_t i = 0; i < g_files.size(); i++) if (g_files[i] == s) { g_files.erase(g_files.begin() + i); break; }
    }
    free(s->carry[0]); free(s->carry[1]);
    s->carry[0] = s->carry[1] = NULL; s->carry_n[0] = s->carry_n[1] = 0;
}

int zeros(GzFile *s, z_off64_t len)                     /* gz_zero, gzwrite.c:134-162 */
{
    z_stream &z = s->strm;
    if (z.avail_in && comp(s, Z_NO_FLUSH) == -1) return -1;
    bool first = true;
    while (len) {
        unsigned n = (z_off64_t)s->size > len ? (unsigned)len : s->size;
        if (first) { memset(s->in.data(), 0, n); first = false; }
        z.avail_in = n; z.next_in = s->in.data();
        s->x.pos += n;
        if (comp(s, Z_NO_FLUSH) == -1) return -1;
        len -= n;
    }
    return 0;
}

/* ------------------------------------------------------------------ chunk index
 * One index member = a gzip member with no data: 1f 8b 08 04, mtime 0, xfl 0, os 3, XLEN, one FEXTRA subfield
 * 'Z' 'I', then the empty final block 03 00, crc 0, isize 0.  Subfield payload (little endian):
 *   u8 version (1), u8 flags (bit 0: last index member), u16 0, u32 n,
 *   u64 offset in the file of the gzip member that is described, u64 number of the first chunk listed here,
 *   n x { u32 compressed bytes, u32 inflated bytes }   chunks of that member's deflate data, in order,
 *   and in the last index member only: u64 offset in the file of the first index member, "ZAMI".
 * So the last 22 bytes of an indexed file are: that offset, "ZAMI", 03 00, eight zero bytes. */
const unsigned IDX_PER_MEMBER = 8000;
void put16(std::vector<unsigned char> &v, unsigned x) { v.push_back((unsigned char)x); v.push_back((unsigned char)(x >> 8)); }
void put32(std::vector<unsigned char> &v, uint32_t x) { put16(v, x & 0xffffu); put16(v, x >> 16); }
void put64(std::vector<unsigned char> &v, uint64_t x) { put32(v, (uint32_t)x); put32(v, (uint32_t)(x >> 32)); }

/* One window member: the same empty gzip member with the subfield 'Z' 'W': u8 version (1), u8 0, u16 0, u32 number of
 * the chunk the window lies in front of, 32768 bytes.  All window members have the same size (WIN_MEMBER), so the j-th
 * one is found without reading the others; they stand between the data and the index members, and the last index member
 * says where they start: with flags bit 1 set its payload holds, in front of the closing 12 bytes,
 * u64 offset in the file of the first window member, u32 spacing in chunks, u32 number of windows. */
const unsigned WIN_MEMBER = 10 + 2 + 4 + 8 + 32768 + 10;
/* Split members: the same empty gzip member with the subfield 'Z' 'S': u8 version (1), u8 records per chunk, u16 0, u32 n,
 * u64 number of the first chunk listed, n x records-per-chunk x u64 split records (zamd_deflate_index_splits: block
 * boundaries inside the chunk, (bit offset in the chunk) << 32 | offset in the chunk's data, unused ones 0), at most
 * SPLITS_PER_MEMBER chunks to a member (a subfield holds 65 531 bytes).  They stand behind the window members; with flags
 * bit 2 set the last index member's payload holds, behind the window fields, u64 offset in the file of the first split
 * member, u32 records per chunk, u32 number of split members.  With them inflate() puts several wavefronts on a chunk. */
const unsigned SPLITS_PER_MEMBER = 1000;
int write_index(GzFile *s)
{
    const size_t nch = s->index.size() / 2;
    std::vector<unsigned char> o;
    static const unsigned char head[10] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 3};
    /* windows in front of chunks that exist */
    size_t nwin = 0;
    if (s->win_every && nch > 1) { nwin = (nch - 1) / s->win_every; if (nwin > s->windows.size()) nwin = s->windows.size(); }
    const uint64_t first_window = s->base + s->produced;
    for (size_t j = 0; j < nwin; j++) {
        o.insert(o.end(), head, head + 10);
        put16(o, 4u + 8u + 32768u);
        o.push_back('Z'); o.push_back('W'); put16(o, 8u + 32768u);
        o.push_back(1); o.push_back(0); put16(o, 0); put32(o, (uint32_t)((j + 1) * s->win_every));
        o.insert(o.end(), s->windows[j].begin(), s->windows[j].end());
        o.push_back(3); o.push_back(0);
        for (int k = 0; k < 8; k++) o.push_back(0);
        if (o.size() >= (8u << 20)) { if (write_all(s, o.data(), o.size()) == -1) return -1; o.clear(); }
    }
    const uint64_t first_split = first_window + (uint64_t)nwin * WIN_MEMBER;
    const unsigned per = s->isplits_per && s->isplits_per <= 8 && s->isplits.size() == nch * (size_t)s->isplits_per ? s->isplits_per : 0u;
    uint64_t split_bytes = 0; unsigned nsplit = 0;
    for (size_t c0 = 0; per && c0 < nch; c0 += SPLITS_PER_MEMBER) {
        const size_t n = nch - c0 < SPLITS_PER_MEMBER ? nch - c0 : SPLITS_PER_MEMBER;
        const unsigned payload = 16u + 8u * per * (unsigned)n;
        o.insert(o.end(), head, head + 10);
        put16(o, payload + 4u);
        o.push_back('Z'); o.push_back('S'); put16(o, payload);
        o.push_back(1); o.push_back((unsigned char)per); put16(o, 0); put32(o, (uint32_t)n); put64(o, c0);
        for (size_t k = c0 * per; k < (c0 + n) * per; k++) put64(o, s->isplits[k]);
        o.push_back(3); o.push_back(0);
        for (int k = 0; k < 8; k++) o.push_back(0);
        split_bytes += 12u + payload + 4u + 10u; nsplit++;
        if (o.size() >= (8u << 20)) { if (write_all(s, o.data(), o.size()) == -1) return -1; o.clear(); }
    }
    const uint64_t first_member = first_split + split_bytes;
    for (size_t c0 = 0; c0 < nch; c0 += IDX_PER_MEMBER) {
        const size_t n = nch - c0 < IDX_PER_MEMBER ? nch - c0 : IDX_PER_MEMBER;
        const bool last = c0 + n == nch;
        const bool wins = last && nwin != 0, spl = last && nsplit != 0;
        const unsigned payload = 24u + 8u * (unsigned)n + (wins ? 16u : 0u) + (spl ? 16u : 0u) + (last ? 12u : 0u);
        o.insert(o.end(), head, head + 10);
        put16(o, payload + 4u);
        o.push_back('Z'); o.push_back('I'); put16(o, payload);
        o.push_back(1); o.push_back((last ? 1 : 0) | (wins ? 2 : 0) | (spl ? 4 : 0)); put16(o, 0); put32(o, (uint32_t)n);
        put64(o, s->index_member); put64(o, c0);
        for (size_t c = c0; c < c0 + n; c++) { put32(o, s->index[2 * c]); put32(o, s->index[2 * c + 1]); }
        if (wins) { put64(o, first_window); put32(o, s->win_every); put32(o, (uint32_t)nwin); }
        if (spl) { put64(o, first_split); put32(o, per); put32(o, nsplit); }
        if (last) { put64(o, first_member); o.push_back('Z'); o.push_back('A'); o.push_back('M'); o.push_back('I'); }
        o.push_back(3); o.push_back(0);
        for (int k = 0; k < 8; k++) o.push_back(0);
    }
    return write_all(s, o.data(), o.size());
}

/* Where the deflate data of the gzip member at `member` starts: behind the fixed header and whatever optional fields its
 * flags announce (RFC 1952; deflate.c:690-867 writes them, inflate.c:653-799 reads them).  0: not a gzip header. */
uint64_t gzip_data_start(int fd, uint64_t member)
{
    unsigned char h[12];
    if (pread(fd, h, 12, (off_t)member) < 10 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8) return 0;
    uint64_t at = member + 10;
    const unsigned flg = h[3];
    if (flg & 4) { at += 2u + ((unsigned)h[10] | ((unsigned)h[11] << 8)); }
    for (int field = 0; field < 2; field++) {            /* name, comment: up to a zero byte each */
        if (!(flg & (field == 0 ? 8 : 16))) continue;
        for (;;) {
            unsigned char b[4096];
            const ssize_t r = pread(fd, b, sizeof b, (off_t)at);
            if (r <= 0) return 0;
            const void *z0 = memchr(b, 0, (size_t)r);
            if (z0) { at += (uint64_t)((const unsigned char *)z0 - b) + 1; break; }
            at += (uint64_t)r;
        }
    }
    if (flg & 2) at += 2;
    return at;
}

/* gzseek() to `target` (an offset in the inflated data) through the stored windows: 1 = the file is positioned at the
 * chunk boundary in front of the target with inflate() set up to start there, 0 = not possible or not worth it (the
 * caller goes on the ordinary way), -1 = error. */
int window_seek(GzFile *s, z_off64_t target)
{
    if (!s->have_ridx || !s->ridx_nwin || !s->ridx_wevery || s->ridx.size() < 4) return 0;
    const size_t nch = s->ridx.size() / 2;
    const uint64_t chunk0 = s->ridx[1];
    if (chunk0 < 32768) return 0;
    uint64_t total = 0;
    for (size_t c = 0; c < nch; c++) { if (c + 1 < nch && s->ridx[2 * c + 1] != chunk0) return 0; total += s->ridx[2 * c + 1]; }
    if ((uint64_t)target >= total) return 0;
    const size_t c = (size_t)((uint64_t)target / chunk0);
    const size_t a = c / s->ridx_wevery * s->ridx_wevery;          /* the chunk with a window in front of it, at or before c */
    if (a == 0 || a / s->ridx_wevery - 1 >= s->ridx_nwin) return 0;
    /* not worth it when the data in between is about to be inflated anyway */
    if (target >= s->x.pos && (uint64_t)a * chunk0 <= (uint64_t)s->x.pos + s->x.have + (s->ra_out ? s->ra_n : 0u)) return 0;
    if (s->size == 0 && look(s) == -1) return -1;                  /* buffers and the inflate state */
    unsigned char wh[24];
    std::vector<unsigned char> win(32768);
    const uint64_t wat = s->ridx_woff + (uint64_t)(a / s->ridx_wevery - 1) * WIN_MEMBER;
    if (pread(s->fd, wh, 24, (off_t)wat) != 24 || wh[0] != 0x1f || wh[1] != 0x8b || wh[12] != 'Z' || wh[13] != 'W' || wh[16] != 1) return 0;
    if (((uint32_t)wh[20] | ((uint32_t)wh[21] << 8) | ((uint32_t)wh[22] << 16) | ((uint32_t)wh[23] << 24)) != (uint32_t)a) return 0;
    if (pread(s->fd, win.data(), 32768, (off_t)(wat + 24)) != 32768) return 0;
    if (s->data_start == 0) s->data_start = gzip_data_start(s->fd, s->ridx_member);
    if (s->data_start == 0) return 0;
    uint64_t comp_at = s->data_start, need = 0;
    for (size_t k = 0; k < a; k++) comp_at += s->ridx[2 * k];
    const size_t upto = c + 1 < nch ? c + 1 : c;                   /* and the chunk behind it: a short read may run on into it */
    for (size_t k = a; k <= upto; k++) need += s->ridx[2 * k];
    if (lseek(s->fd, (off_t)comp_at, SEEK_SET) == -1) return -1;
    z_stream &z = s->strm;
    if (inflateReset2(&z, -15) != Z_OK || inflateSetDictionary(&z, win.data(), 32768) != Z_OK ||
        zamd_inflate_index_hint(&z, s->ridx.data() + 2 * a, (unsigned)(nch - a)) != Z_OK) {
        set_error(s, Z_STREAM_ERROR, "internal error: inflate stream corrupt");
        return -1;
    }
    if (s->rsplits_per) (void)zamd_inflate_index_hint_splits(&z, s->rsplits.data() + a * (size_t)s->rsplits_per, s->rsplits_per);
    z.avail_in = 0; z.next_in = s->in.data();
    s->x.have = 0; s->eof = 0; s->past = 0; s->seek = 0; s->skip_trailer = 0;
    s->ra_out = false; s->ra_cur = -1; s->seq = 0;                 /* what was inflated ahead is dropped */
    set_error(s, Z_OK, NULL);
    s->how = GZIP; s->direct = 0; s->raw_mode = 1;
    s->read_limit = need;                                          /* the first read brings just the chunks up to the target's */
    s->out_limit = 0;
    for (size_t k = a; k <= upto; k++) s->out_limit += s->ridx[2 * k + 1];
    s->x.pos = (z_off64_t)((uint64_t)a * chunk0);
    s->window_seeks++;
    return 1;
}

}  // namespace

extern "C" {

/* Reads the chunk index at the end of the file behind fd (which must be seekable; its position is left
 * anywhere).  0 = found: out->pairs holds n_chunks x { compressed bytes, inflated bytes } (free with
 * zamd_gz_index_free), out->member_offset the file offset of the gzip member they describe.  -1 = no index. */
int zamd_gz_index_load(int fd, zamd_gz_index *out)
{
    if (out == NULL) return -1;
    memset(out, 0, sizeof *out);
    const off_t end = lseek(fd, 0, SEEK_END);
    if (end == (off_t)-1 || end < 22 + 10 + 2 + 4 + 24) return -1;
    unsigned char tail[22];
    if (pread(fd, tail, 22, end - 22) != 22) return -1;
    static const unsigned char fin[10] = {3, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (memcmp(tail + 8, "ZAMI", 4) != 0 || memcmp(tail + 12, fin, 10) != 0) return -1;
    auto rd = [](const unsigned char *p, int n) { uint64_t v = 0; for (int i = n - 1; i >= 0; i--) v = (v << 8) | p[i]; return v; };
    uint64_t at = rd(tail, 8);
    if (at >= (uint64_t)end) return -1;
    std::vector<unsigned> pairs;
    uint64_t member = 0, split_at = 0; unsigned split_per = 0, split_members = 0;
    for (;;) {
        unsigned char h[16 + 24];
        if (at + sizeof h > (uint64_t)end || pread(fd, h, sizeof h, (off_t)at) != (ssize_t)sizeof h) return -1;
        if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || h[3] != 4 || h[12] != 'Z' || h[13] != 'I' || h[16] != 1) return -1;
        const unsigned xlen = (unsigned)rd(h + 10, 2), plen = (unsigned)rd(h + 14, 2);
        const unsigned flags = h[17];
        const uint64_t n = rd(h + 20, 4), first = rd(h + 32, 8);
        const unsigned ext = ((flags & 2) ? 16u : 0u) + ((flags & 4) ? 16u : 0u);
        if (plen + 4u != xlen || plen != 24u + 8u * n + ext + ((flags & 1) ? 12u : 0u) || first != pairs.size() / 2) return -1;
        if ((flags & 6) && !(flags & 1)) return -1;
        if (pairs.empty()) member = rd(h + 24, 8); else if (member != rd(h + 24, 8)) return -1;
        std::vector<unsigned char> body(8 * n + ext);
        if (!body.empty() && pread(fd, body.data(), body.size(), (off_t)(at + sizeof h)) != (ssize_t)body.size()) return -1;
        for (uint64_t i = 0; i < 2 * n; i++) pairs.push_back((unsigned)rd(body.data() + 4 * i, 4));
        if (flags & 4) {
            const unsigned char *q = body.data() + 8 * n + ((flags & 2) ? 16u : 0u);
            split_at = rd(q, 8); split_per = (unsigned)rd(q + 8, 4); split_members = (unsigned)rd(q + 12, 4);
        }
        if (flags & 2) {
            out->window_offset = rd(body.data() + 8 * n, 8);
            out->window_every = (unsigned)rd(body.data() + 8 * n + 8, 4);
            out->n_windows = (unsigned)rd(body.data() + 8 * n + 12, 4);
            if (out->window_every == 0 || out->window_offset + (uint64_t)out->n_windows * WIN_MEMBER > (uint64_t)end) { out->window_offset = 0; out->window_every = out->n_windows = 0; }
        }
        at += 12 + xlen + 10;
        if (flags & 1) break;
    }
    if (at != (uint64_t)end) return -1;
    out->pairs = (unsigned *)malloc(pairs.size() * sizeof(unsigned) + 1);
    if (out->pairs == NULL) return -1;
    if (!pairs.empty()) memcpy(out->pairs, pairs.data(), pairs.size() * sizeof(unsigned));
    out->n_chunks = (unsigned)(pairs.size() / 2);
    out->member_offset = member;
    /* the split records: hints, so anything about them that does not add up just leaves them out */
    if (split_members && split_per && split_per <= 8) {
        std::vector<unsigned long long> recs;
        uint64_t sat = split_at; bool ok = true;
        for (unsigned m = 0; ok && m < split_members; m++) {
            unsigned char h[16 + 16];
            ok = sat + sizeof h <= (uint64_t)end && pread(fd, h, sizeof h, (off_t)sat) == (ssize_t)sizeof h && h[0] == 0x1f && h[1] == 0x8b && h[3] == 4 &&
                 h[12] == 'Z' && h[13] == 'S' && h[16] == 1 && h[17] == split_per;
            if (!ok) break;
            const unsigned xlen = (unsigned)rd(h + 10, 2), plen = (unsigned)rd(h + 14, 2);
            const uint64_t n = rd(h + 20, 4), first = rd(h + 24, 8);
            ok = plen + 4u == xlen && plen == 16u + 8u * split_per * n && first * split_per == recs.size();
            if (!ok) break;
            std::vector<unsigned char> body(8 * (size_t)split_per * n);
            ok = body.empty() || pread(fd, body.data(), body.size(), (off_t)(sat + sizeof h)) == (ssize_t)body.size();
            for (uint64_t i = 0; ok && i < split_per * n; i++) recs.push_back(rd(body.data() + 8 * i, 8));
            sat += 12 + xlen + 10;
        }
        if (ok && recs.size() == (size_t)out->n_chunks * split_per) {
            out->splits = (unsigned long long *)malloc(recs.size() * sizeof(unsigned long long) + 1);
            if (out->splits) { if (!recs.empty()) memcpy(out->splits, recs.data(), recs.size() * sizeof(unsigned long long)); out->splits_per_chunk = split_per; }
        }
    }
    return 0;
}

void zamd_gz_index_free(zamd_gz_index *ix) { if (ix) { free(ix->pairs); free(ix->splits); ix->pairs = NULL; ix->splits = NULL; ix->n_chunks = 0; ix->splits_per_chunk = 0; } }

int zamd_gz_seek_stats(gzFile file, unsigned long long *chunks_decoded, unsigned long long *window_seeks)
{
    GzFile *s = gz(file);
    if (s == NULL || s->mode != GZ_READ) return -1;
    wait_idle(s);
    if (chunks_decoded) *chunks_decoded = s->size ? zamd_inflate_hinted_chunks(&s->strm) : 0ull;
    if (window_seeks) *window_seeks = s->window_seeks;
    return 0;
}

gzFile gzopen(const char *path, const char *mode) { return open_file(path, -1, mode); }       /* gzlib.c:268 */
gzFile gzopen64(const char *path, const char *mode) { return open_file(path, -1, mode); }     /* gzlib.c:276 */

gzFile gzdopen(int fd, const char *mode)                                                        /* gzlib.c:284-301 */
{
    if (fd == -1) return NULL;
    char name[32];
    snprintf(name, sizeof name, "<fd:%d>", fd);
    return open_file(name, fd, mode);
}

int gzbuffer(gzFile file, unsigned size)                                                        /* gzlib.c:314-336 */
{
    GzFile *s = gz(file);
    if (s == NULL || (s->mode != GZ_READ && s->mode != GZ_WRITE)) return -1;
    if (s->size != 0) return -1;                        /* too late: buffers exist */
    if (size < 2) size = 2;
    if ((size << 1) < size) return -1;                  /* the read side needs twice as much */
    s->want = size;
    return 0;
}

int gzrewind(gzFile file)                                                                       /* gzlib.c:339-359 */
{
    GzFile *s = gz(file);
    if (s == NULL || s->mode != GZ_READ) return -1;
    wait_idle(s);
    if (fatal(s)) return -1;
    if (lseek(s->fd, s->start, SEEK_SET) == -1) return -1;
    reset(s);
    return 0;
}

static z_off64_t gzseek64_body(gzFile file, z_off64_t offset, int whence)                                   /* gzlib.c:362-436 */
{
    GzFile *s = gz(file);
    if (s == NULL || (s->mode != GZ_READ && s->mode != GZ_WRITE)) return -1;
    if (s->mode == GZ_WRITE) (void)settle(s);
    else wait_idle(s);
    if (fatal(s)) return -1;
    if (whence != SEEK_SET && whence != SEEK_CUR) return -1;
    if (whence == SEEK_SET) offset -= s->x.pos;
    else if (s->seek) offset += s->skip;
    s->seek = 0;
    if (s->mode == GZ_READ && s->how == COPY && s->x.pos + offset >= 0) {      /* plain file: seek the descriptor */
        if (lseek(s->fd, offset - s->x.have, SEEK_CUR) == -1) return -1;
        s->x.have = 0; s->eof = 0; s->past = 0; s->seek = 0;
        set_error(s, Z_OK, NULL);
        s->strm.avail_in = 0;
        s->x.pos += offset;
        return s->x.pos;
    }
    if (s->mode == GZ_READ && s->x.pos + offset >= 0 && s->ridx_nwin) {
        /* a file that carries windows: start from the one in front of the target's chunk */
        const z_off64_t target = s->x.pos + offset;
        const int r = window_seek(s, target);
        if (r == -1) return -1;
        if (r == 1) {
            const z_off64_t rest = target - s->x.pos;
            if (rest) { s->seek = 1; s->skip = rest; }
            return target;
        }
    }
    if (offset < 0) {                                   /* backwards: only by reading again from the start */
        if (s->mode != GZ_READ) return -1;
        offset += s->x.pos;
        if (offset < 0) return -1;
        if (gzrewind(file) == -1) return -1;
    }
    if (s->mode == GZ_READ) {
        unsigned n = (z_off64_t)s->x.have > offset ? (unsigned)offset : s->x.have;
        s->x.have -= n; s->x.next += n; s->x.pos += n; offset -= n;
    }
    if (offset) { s->seek = 1; s->skip = offset; }
    return s->x.pos + offset;
}
z_off64_t gzseek64(gzFile file, z_off64_t offset, int whence)
{
    try { return gzseek64_body(file, offset, whence); }
    catch (...) { return -1; }      /* an allocation failed: the call fails, nothing is thrown at a C caller */
}

z_off_t gzseek(gzFile file, z_off_t offset, int whence)
{
    z_off64_t r = gzseek64(file, (z_off64_t)offset, whence);
    return r == (z_off_t)r ? (z_off_t)r : -1;
}

z_off64_t gztell64(gzFile file)                                                                 /* gzlib.c:451-465 */
{
    GzFile *s = gz(file);
    if (s == NULL || (s->mode != GZ_READ && s->mode != GZ_WRITE)) return -1;
    return s->x.pos + (s->seek ? s->skip : 0);
}
z_off_t gztell(gzFile file) { z_off64_t r = gztell64(file); return r == (z_off_t)r ? (z_off_t)r : -1; }

z_off64_t gzoffset64(gzFile file)                                                               /* gzlib.c:478-498 */
{
    GzFile *s = gz(file);
    if (s == NULL || (s->mode != GZ_READ && s->mode != GZ_WRITE)) return -1;
    wait_idle(s);                                       /* (what is collected and not yet compressed is in no file) */
    z_off64_t off = lseek(s->fd, 0, SEEK_CUR);
    if (off == -1) return -1;
    if (s->mode == GZ_READ) off -= s->strm.avail_in;
    return off;
}
z_off_t gzoffset(gzFile file) { z_off64_t r = gzoffset64(file); return r == (z_off_t)r ? (z_off_t)r : -1; }

int gzeof(gzFile file)                                                                          /* gzlib.c:511-525 */
{
    GzFile *s = gz(file);
    if (s == NULL || (s->mode != GZ_READ && s->mode != GZ_WRITE)) return 0;
    return s->mode == GZ_READ ? s->past : 0;
}

const char *gzerror(gzFile file, int *errnum)                                                   /* gzlib.c:528-546 */
{
    GzFile *s = gz(file);
    if (s == NULL || (s->mode != GZ_READ && s->mode != GZ_WRITE)) return NULL;
    if (s->mode == GZ_WRITE) wait_idle(s);
    if (errnum != NULL) *errnum = s->err;
    return s->err == Z_MEM_ERROR ? "out of memory" : s->msg.c_str();
}

void gzclearerr(gzFile file)                                                                    /* gzlib.c:549-567 */
{
    GzFile *s = gz(file);
    if (s == NULL || (s->mode != GZ_READ && s->mode != GZ_WRITE)) return;
    wait_idle(s);
    if (s->mode == GZ_READ) { s->eof = 0; s->past = 0; }
    set_error(s, Z_OK, NULL);
}

/* ------------------------------------------------------------------ gzread.c */
/* Requests that go through the buffers: those below two buffers' worth, and, once the reader has shown that it reads on
 * and on, those that are small beside what the read-ahead thread inflates at a time (one round of that size is faster
 * than four of a quarter). */
static inline bool small(const GzFile *s, unsigned len)
{
    return len < (s->size << 1) || (s->ahead_bytes > (s->size << 1) && len <= s->ahead_bytes / 4 && s->seq >= AHEAD_AFTER);
}
static int gzread_body(gzFile file, voidp buf, unsigned len)                                                /* gzread.c:288-379 */
{
    GzFile *s = gz(file);
    if (s == NULL || s->mode != GZ_READ || fatal(s)) return -1;
    if ((int)len < 0) { set_error(s, Z_DATA_ERROR, "requested length does not fit in int"); return -1; }
    if (len == 0) return 0;
    if (s->seek) { s->seek = 0; if (skip_out(s, s->skip) == -1) return -1; }
    z_stream &z = s->strm;
    unsigned got = 0, n;
    do {
        if (s->x.have) {                                /* what is already inflated */
            n = s->x.have > len ? len : s->x.have;
            memcpy(buf, s->x.next, n);
            s->x.next += n; s->x.have -= n;
        } else if (s->ra_out) {                         /* what was inflated ahead comes first, whatever the request's size */
            if (ahead_take(s, small(s, len)) == -1) return -1;
            continue;
        } else if (at_end(s)) { s->past = 1; break; }
        else if (s->how == LOOK && len >= ((s->size ? s->size : s->want) << 1)) {      /* a large request at a member's start: find out what follows, */
            if (look(s) == -1) return -1;                                  /* then straight into the caller's memory as well */
            if (s->how == LOOK && s->x.have == 0 && s->strm.avail_in == 0 && s->eof) { s->past = 1; break; }
            continue;
        } else if (s->how == LOOK || small(s, len)) {   /* small request or new member: through the buffer */
            if (fetch(s) == -1) return -1;
            continue;
        } else if (s->how == COPY) {                            /* large request: straight into the caller's memory */
            if (load(s, (unsigned char *)buf, len, &n) == -1) return -1;
        } else {
            /* inflate() of this library runs at what one call holds: a request of hundreds of megabytes gets input to
             * match (about what it inflates from, the request's own size at most), not one buffer of 32 MiB at a time —
             * that is a round of some seventy chunks, as slow as a round of thousands */
            size_input(s, len);
            z.avail_out = len; z.next_out = (Bytef *)buf;
            const int rc = decomp(s);
            s->fill = 0;
            if (rc == -1) return -1;
            n = s->x.have; s->x.have = 0;
            s->seq += n;
        }
        len -= n; buf = (char *)buf + n; got += n; s->x.pos += n;
    } while (len);
    return (int)got;
}
int gzread(gzFile file, voidp buf, unsigned len)
{
    static const bool trace = getenv("ZAMD_Z_TRACE") != nullptr;
    const double t0 = trace ? now_ms() : 0;
    struct Done { double t0; unsigned len; bool on; ~Done() { if (on) fprintf(stderr, "zamd_gz gzread(%u): %.1f ms\n", len, now_ms() - t0); } } done = {t0, len, trace};
    try { return gzread_body(file, buf, len); }
    catch (...) { return -1; }      /* an allocation failed: the call fails, nothing is thrown at a C caller */
}

#undef gzgetc
static int gzgetc_body(gzFile file)                                                                         /* gzread.c:387-414 */
{
    GzFile *s = gz(file);
    if (s == NULL || s->mode != GZ_READ || fatal(s)) return -1;
    if (s->x.have) { s->x.have--; s->x.pos++; return *(s->x.next)++; }
    unsigned char c;
    return gzread(file, &c, 1) < 1 ? -1 : c;
}
int gzgetc(gzFile file)
{
    try { return gzgetc_body(file); }
    catch (...) { return -1; }      /* an allocation failed: the call fails, nothing is thrown at a C caller */
}
int gzgetc_(gzFile file) { return gzgetc(file); }

static int gzungetc_body(int c, gzFile file)                                                                /* gzread.c:423-480 */
{
    GzFile *s = gz(file);
    if (s == NULL || s->mode != GZ_READ || fatal(s)) return -1;
    if (s->seek) { s->seek = 0; if (skip_out(s, s->skip) == -1) return -1; }
    if (c < 0) return -1;
    if (s->size == 0 && look(s) == -1) return -1;       /* the buffers must exist before a byte can be parked in them */
    unsigned char *const out = s->out.data();
    const size_t cap = (size_t)s->size << 1;
    if (s->x.have && s->ra_cur >= 0) {                  /* the data lies in a read-ahead buffer: there is room in front of it */
        if (s->x.next == s->ra_buf[s->ra_cur].data()) { set_error(s, Z_DATA_ERROR, "out of room to push characters"); return -1; }
        s->x.have++; s->x.next--; s->x.next[0] = (unsigned char)c;
        s->x.pos--; s->past = 0;
        return c;
    }
    if (s->x.have == 0) {                               /* empty: the byte goes to the very end, room for more in front */
        s->x.have = 1; s->x.next = out + cap - 1; s->x.next[0] = (unsigned char)c; s->ra_cur = -1;
        s->x.pos--; s->past = 0;
        return c;
    }
    if (s->x.have == cap) { set_error(s, Z_DATA_ERROR, "out of room to push characters"); return -1; }
    if (s->x.next == out) {                             /* slide the data to the end of the buffer */
        memmove(out + cap - s->x.have, out, s->x.have);
        s->x.next = out + cap - s->x.have;
    }
    s->x.have++; s->x.next--; s->x.next[0] = (unsigned char)c;
    s->x.pos--; s->past = 0;
    return c;
}
int gzungetc(int c, gzFile file)
{
    try { return gzungetc_body(c, file); }
    catch (...) { return -1; }      /* an allocation failed: the call fails, nothing is thrown at a C caller */
}

static char *gzgets_body(gzFile file, char *buf, int len)                                                   /* gzread.c:483-544 */
{
    GzFile *s = gz(file);
    if (s == NULL || buf == NULL || len < 1) return NULL;
    if (s->mode != GZ_READ || fatal(s)) return NULL;
    if (s->seek) { s->seek = 0; if (skip_out(s, s->skip) == -1) return NULL; }
    char *str = buf;
    unsigned left = (unsigned)len - 1;
    const unsigned char *eol = NULL;
    while (left && eol == NULL) {
        if (s->x.have == 0 && fetch(s) == -1) return NULL;
        if (s->x.have == 0) { s->past = 1; break; }
        unsigned n = s->x.have > left ? left : s->x.have;
        eol = (const unsigned char *)memchr(s->x.next, '\n', n);
        if (eol != NULL) n = (unsigned)(eol - s->x.next) + 1;
        memcpy(buf, s->x.next, n);
        s->x.have -= n; s->x.next += n; s->x.pos += n; left -= n; buf += n;
    }
    if (buf == str) return NULL;
    buf[0] = 0;
    return str;
}
char *gzgets(gzFile file, char *buf, int len)
{
    try { return gzgets_body(file, buf, len); }
    catch (...) { return NULL; }      /* an allocation failed: the call fails, nothing is thrown at a C caller */
}

int gzdirect(gzFile file)                                                                       /* gzread.c:547-564 */
{
    GzFile *s = gz(file);
    if (s == NULL) return 0;
    if (s->mode == GZ_READ) wait_idle(s);
    if (s->mode == GZ_READ && s->how == LOOK && s->x.have == 0 && !s->ra_out) (void)look(s);
    return s->direct;
}

static int gzclose_r_body(gzFile file)                                                                      /* gzread.c:567-594 */
{
    GzFile *s = gz(file);
    if (s == NULL) return Z_STREAM_ERROR;
    if (s->mode != GZ_READ) return Z_STREAM_ERROR;
    stop_worker(s);
    if (s->size) inflateEnd(&s->strm);
    int err = s->err == Z_BUF_ERROR ? Z_BUF_ERROR : Z_OK;
    int ret = close(s->fd);
    delete s;
    return ret ? Z_ERRNO : err;
}
int gzclose_r(gzFile file)
{
    try { return gzclose_r_body(file); }
    catch (...) { return Z_MEM_ERROR; }      /* an allocation failed: the call fails, nothing is thrown at a C caller */
}

/* ------------------------------------------------------------------ gzwrite.c */
static int gzwrite_body(gzFile file, const void *buf, unsigned len)                                         /* gzwrite.c:165-243 */
{
    GzFile *s = gz(file);
    if (s == NULL || s->mode != GZ_WRITE) return 0;
    if (idle(s) && s->err != Z_OK) return 0;            /* (an error of a job that is still out shows once it is done) */
    if ((int)len < 0) { (void)settle(s); set_error(s, Z_DATA_ERROR, "requested length does not fit in int"); return 0; }
    if (len == 0) return 0;
    if (s->size == 0 && winit(s) == -1) return 0;
    if (s->seek) { s->seek = 0; if (settle(s) == -1 || zeros(s, s->skip) == -1) return 0; }
    z_stream &z = s->strm;
    const unsigned put = len;
    if (s->carry_cap && !s->direct && len < s->carry_cap && (behind(s) || len >= (s->carry_cap >> 2 < (1u << 20) ? s->carry_cap >> 2 : (1u << 20)))) {
        /* collected for the thread behind: what the small buffer holds is older and goes in first */
        if (!s->carry[s->cur]) s->carry[s->cur] = (unsigned char *)malloc(s->carry_cap);
        if (s->carry[s->cur]) {
            if (!behind(s) && z.avail_in) {
                memcpy(s->carry[s->cur], z.next_in, z.avail_in);
                s->carry_n[s->cur] = z.avail_in; z.avail_in = 0;
            }
            while (len) {
                const int k = s->cur;
                size_t copy = s->carry_cap - s->carry_n[k];
                if (copy > len) copy = len;
                memcpy(s->carry[k] + s->carry_n[k], buf, copy);
                s->carry_n[k] += copy; s->x.pos += copy;
                buf = (const char *)buf + copy; len -= (unsigned)copy;
                if (s->carry_n[k] == s->carry_cap && submit(s) == -1) return 0;
            }
            return (int)put;
        }
    }
    if (behind(s) && settle(s) == -1) return 0;
    if (len < s->size) {                                /* collect small writes in the input buffer */
        do {
            if (z.avail_in == 0) z.next_in = s->in.data();
            unsigned have = (unsigned)((z.next_in + z.avail_in) - s->in.data());
            unsigned copy = s->size - have;
            if (copy > len) copy = len;
            memcpy(s->in.data() + have, buf, copy);
            z.avail_in += copy; s->x.pos += copy;
            buf = (const char *)buf + copy; len -= copy;
            if (len && comp(s, Z_NO_FLUSH) == -1) return 0;
        } while (len);
    } else {                                            /* large writes are compressed from the caller's memory */
        if (z.avail_in && comp(s, Z_NO_FLUSH) == -1) return 0;
        z.avail_in = len; z.next_in = (const Bytef *)buf;
        s->x.pos += len;
        if (comp(s, Z_NO_FLUSH) == -1) return 0;
    }
    return (int)put;
}
int gzwrite(gzFile file, const void *buf, unsigned len)
{
    try { return gzwrite_body(file, buf, len); }
    catch (...) { return 0; }      /* an allocation failed: the call fails, nothing is thrown at a C caller */
}

int gzputc(gzFile file, int c)                                                                  /* gzwrite.c:246-291 */
{
    GzFile *s = gz(file);
    if (s == NULL || s->mode != GZ_WRITE) return -1;
    unsigned char b = (unsigned char)c;
    return gzwrite(file, &b, 1) != 1 ? -1 : (c & 0xff);
}

int gzputs(gzFile file, const char *str)                                                        /* gzwrite.c:294-305 */
{
    unsigned len = (unsigned)strlen(str);
    int ret = gzwrite(file, str, len);
    return ret == 0 && len != 0 ? -1 : ret;
}

static int gzvprintf_body(gzFile file, const char *format, va_list va)                                      /* gzwrite.c:311-373 */
{
    GzFile *s = gz(file);
    if (s == NULL || s->mode != GZ_WRITE) return 0;
    (void)settle(s);
    if (s->err != Z_OK) return 0;
    if (s->size == 0 && winit(s) == -1) return 0;
    if (s->seek) { s->seek = 0; if (zeros(s, s->skip) == -1) return 0; }
    z_stream &z = s->strm;
    if (z.avail_in && comp(s, Z_NO_FLUSH) == -1) return 0;
    int size = (int)(s->size > (unsigned)INT_MAX ? INT_MAX : s->size);
    s->in[size - 1] = 0;
    int len = vsnprintf((char *)s->in.data(), (size_t)size, format, va);
    if (len <= 0 || len >= size || s->in[size - 1] != 0) return 0;      /* what does not fit the buffer is not written */
    z.avail_in = (unsigned)len; z.next_in = s->in.data();
    s->x.pos += len;
    return len;
}
int gzvprintf(gzFile file, const char *format, va_list va)
{
    try { return gzvprintf_body(file, format, va); }
    catch (...) { return 0; }      /* an allocation failed: the call fails, nothing is thrown at a C caller */
}

int gzprintf(gzFile file, const char *format, ...)                                              /* gzwrite.c:375-385 */
{
    va_list va;
    va_start(va, format);
    int ret = gzvprintf(file, format, va);
    va_end(va);
    return ret;
}

static int gzflush_body(gzFile file, int flush)                                                             /* gzwrite.c:478-510 */
{
    GzFile *s = gz(file);
    if (s == NULL) return -1;
    if (s->mode != GZ_WRITE) return Z_STREAM_ERROR;
    if (idle(s) && s->err != Z_OK) return Z_STREAM_ERROR;
    if (settle(s) == -1) return s->err;                 /* the failure belongs to data that this call was to flush */
    if (flush < 0 || flush > Z_FINISH) return Z_STREAM_ERROR;
    if (s->seek) { s->seek = 0; if (zeros(s, s->skip) == -1) return -1; }
    comp(s, flush);
    return s->err;
}
int gzflush(gzFile file, int flush)
{
    try { return gzflush_body(file, flush); }
    catch (...) { return Z_MEM_ERROR; }      /* an allocation failed: the call fails, nothing is thrown at a C caller */
}

static int gzsetparams_body(gzFile file, int level, int strategy)                                           /* gzwrite.c:513-552 */
{
    GzFile *s = gz(file);
    if (s == NULL) return Z_STREAM_ERROR;
    if (s->mode != GZ_WRITE) return Z_STREAM_ERROR;
    if (idle(s) && s->err != Z_OK) return Z_STREAM_ERROR;
    if (settle(s) == -1) return s->err;
    if (level == s->level && strategy == s->strategy) return Z_OK;
    if (s->seek) { s->seek = 0; if (zeros(s, s->skip) == -1) return -1; }
    if (s->size) {
        if (s->strm.avail_in && comp(s, Z_PARTIAL_FLUSH) == -1) return s->err;
        deflateParams(&s->strm, level, strategy);
    }
    s->level = level; s->strategy = strategy;
    return Z_OK;
}
int gzsetparams(gzFile file, int level, int strategy)
{
    try { return gzsetparams_body(file, level, strategy); }
    catch (...) { return Z_MEM_ERROR; }      /* an allocation failed: the call fails, nothing is thrown at a C caller */
}

static int gzclose_w_body(gzFile file)                                                                      /* gzwrite.c:555-577 */
{
    GzFile *s = gz(file);
    if (s == NULL) return Z_STREAM_ERROR;
    if (s->mode != GZ_WRITE) return Z_STREAM_ERROR;
    int ret = Z_OK;
    if (settle(s) == -1) ret = s->err;
    stop_worker(s);
    if (s->seek) { s->seek = 0; if (zeros(s, s->skip) == -1) ret = s->err; }
    if (comp(s, Z_FINISH) == -1) ret = s->err;
    if (ret == Z_OK && s->want_index && !s->direct && s->members == 1 && !s->index.empty() && write_index(s) == -1) ret = s->err;
    if (s->size && !s->direct) (void)deflateEnd(&s->strm);
    if (close(s->fd) == -1) ret = Z_ERRNO;
    delete s;
    return ret;
}
int gzclose_w(gzFile file)
{
    try { return gzclose_w_body(file); }
    catch (...) { return Z_MEM_ERROR; }      /* an allocation failed: the call fails, nothing is thrown at a C caller */
}

int gzclose(gzFile file)                                                                        /* gzclose.c:11-25 */
{
    GzFile *s = gz(file);
    if (s == NULL) return Z_STREAM_ERROR;
    return s->mode == GZ_READ ? gzclose_r(file) : gzclose_w(file);
}

}  // extern "C"
