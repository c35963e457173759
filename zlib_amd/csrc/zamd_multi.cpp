/*
 * zamd_multi.cpp — one deflate call spread over several GPUs of one node, inside the library.
 *
 * The input is cut into contiguous ranges of chunks, one range per device; a device's first chunk is primed with the
 * 32 KiB of input in front of its range, as every chunk inside a range is primed with the chunk before it (the
 * dictionary rule of deflate.c:323-389 applied at every cut), so the ranges compress independently and their outputs,
 * laid end to end, are byte for byte what one device makes of the whole input.  What the devices have to exchange is the
 * table of compressed chunk lengths, 8 bytes per chunk: every device contributes the lengths of its range, an
 * all-gather hands every device the whole table, and a scan of the table on each device gives the place of every chunk
 * in the stream.  The all-gather is RCCL's (ncclAllGather, one communicator per device in this process,
 * ncclCommInitAll), loaded from librccl.so at run time; when that library is not there, or a device is listed more than
 * once (RCCL refuses duplicates), the lengths travel through host memory instead and the scan runs on the host.
 *
 * One host thread per device: upload, deflate, lengths; all meet; exchange; scan; download to the range's place.
 * The reference has no counterpart: its deflate() is one thread on one core (deflate.c:665-976).
 */
#include "../../include/zamd.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>       /* types and prototypes only: the library itself is opened with dlopen */

#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool load()
    {
        if (getenv("ZAMD_NO_RCCL")) return false;
        lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) return false;
        CommInitAll = (decltype(CommInitAll))dlsym(lib, "ncclCommInitAll");
        CommDestroy = (decltype(CommDestroy))dlsym(lib, "ncclCommDestroy");
        AllGather = (decltype(AllGather))dlsym(lib, "ncclAllGather");
        GroupStart = (decltype(GroupStart))dlsym(lib, "ncclGroupStart");
        GroupEnd = (decltype(GroupEnd))dlsym(lib, "ncclGroupEnd");
        GetErrorString = (decltype(GetErrorString))dlsym(lib, "ncclGetErrorString");
        return CommInitAll && CommDestroy && AllGather && GroupStart && GroupEnd;
    }
};

struct Dev {
    int device = 0;
    zamd_engine *eng = nullptr;
    hipStream_t stream = nullptr;                  /* the exchange's stream on this device */
    uint8_t *d_in = nullptr; uint64_t in_cap = 0;  /* [32 KiB window][range] */
    uint8_t *d_out = nullptr; uint64_t out_cap = 0;
    uint64_t *d_tab = nullptr; uint64_t tab_cap = 0;   /* local offsets | lengths to send | gathered lengths | global offsets */
    ncclComm_t comm = nullptr;
};

}  // namespace

struct zamd_multi {
    std::vector<Dev> devs;
    Rccl rccl;
    bool use_rccl = false;
};

/* Chunks [first, first + count) of total_chunks belong to device `index` of n_devices: ranges of
 * per = ceil(total / n) chunks in device order, so that every device in front of the last one that has any holds
 * exactly `per` — a table of n * per entries, device after device, then has its padding at the very end and chunk c at
 * entry c. */
extern "C" int zamd_multi_partition(uint64_t total_chunks, int n_devices, int index, uint64_t *first, uint64_t *count, uint64_t *per_device)
{
    if (n_devices < 1 || index < 0 || index >= n_devices || !first || !count) return ZAMD_STREAM_ERROR;
    const uint64_t per = (total_chunks + (uint64_t)n_devices - 1) / (uint64_t)n_devices;
    const uint64_t f = per * (uint64_t)index;
    *first = f < total_chunks ? f : total_chunks;
    *count = f >= total_chunks ? 0 : (total_chunks - f < per ? total_chunks - f : per);
    if (per_device) *per_device = per;
    return ZAMD_OK;
}

/* The host's form of the scan (used when the lengths travel through host memory): offsets[i] = base + sum of
 * lengths[0, i), n + 1 entries. */
extern "C" int zamd_multi_offsets_host(const uint64_t *lengths, uint64_t n, uint64_t base, uint64_t *offsets)
{
    if ((!lengths && n) || !offsets) return ZAMD_STREAM_ERROR;
    uint64_t run = base;
    for (uint64_t i = 0; i < n; i++) { offsets[i] = run; run += lengths[i]; }
    offsets[n] = run;
    return ZAMD_OK;
}

extern "C" void zamd_multi_destroy(zamd_multi *m)
{
    if (!m) return;
    for (Dev &d : m->devs) {
        if (d.comm && m->rccl.CommDestroy) { hipSetDevice(d.device); m->rccl.CommDestroy(d.comm); }
        if (d.eng) {
            zamd_dev_free(d.eng, d.d_in); zamd_dev_free(d.eng, d.d_out); zamd_dev_free(d.eng, d.d_tab);
            hipSetDevice(d.device);
            if (d.stream) hipStreamDestroy(d.stream);
            zamd_engine_destroy(d.eng);
        }
    }
    if (m->rccl.lib) dlclose(m->rccl.lib);
    delete m;
}

extern "C" int zamd_multi_create(zamd_multi **out, const int *devices, int n_devices, uint64_t workspace_bytes)
{
    if (!out || !devices || n_devices < 1 || n_devices > 64) return ZAMD_STREAM_ERROR;
    zamd_multi *m = new (std::nothrow) zamd_multi();
    if (!m) return ZAMD_MEM_ERROR;
    m->devs.resize((size_t)n_devices);
    bool distinct = true;
    for (int i = 0; i < n_devices; i++) {
        m->devs[(size_t)i].device = devices[i];
        for (int j = 0; j < i; j++) if (devices[j] == devices[i]) distinct = false;
    }
    for (Dev &d : m->devs) {
        int rc = zamd_engine_create(&d.eng, d.device, workspace_bytes);
        if (rc != ZAMD_OK) { zamd_multi_destroy(m); return rc; }
        if (hipSetDevice(d.device) != hipSuccess || hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking) != hipSuccess) { zamd_multi_destroy(m); return ZAMD_HIP_ERROR(1); }
    }
    /* RCCL wants every rank of a communicator on a GPU of its own */
    if (distinct && m->rccl.load()) {
        std::vector<ncclComm_t> comms((size_t)n_devices, nullptr);
        const ncclResult_t r = m->rccl.CommInitAll(comms.data(), n_devices, devices);
        if (r == ncclSuccess) {
            for (int i = 0; i < n_devices; i++) m->devs[(size_t)i].comm = comms[(size_t)i];
            m->use_rccl = true;
        } else if (getenv("ZAMD_MULTI_VERBOSE"))
            fprintf(stderr, "zamd: ncclCommInitAll failed (%s): chunk lengths will travel through host memory\n", m->rccl.GetErrorString ? m->rccl.GetErrorString(r) : "?");
    }
    *out = m;
    return ZAMD_OK;
}

extern "C" int zamd_multi_devices(zamd_multi *m) { return m ? (int)m->devs.size() : 0; }
extern "C" zamd_engine *zamd_multi_engine(zamd_multi *m, int i) { return (m && i >= 0 && (size_t)i < m->devs.size()) ? m->devs[(size_t)i].eng : nullptr; }
extern "C" int zamd_multi_uses_rccl(zamd_multi *m) { return m && m->use_rccl ? 1 : 0; }

namespace {

int grow(Dev &d, uint8_t **p, uint64_t *cap, uint64_t need)
{
    if (*cap >= need) return ZAMD_OK;
    if (*p) zamd_dev_free(d.eng, *p);
    *p = nullptr; *cap = 0;
    void *q = nullptr;
    const uint64_t want = need + (need >> 3) + 4096;
    if (zamd_dev_alloc(d.eng, want, &q) != ZAMD_OK) return ZAMD_MEM_ERROR;
    *p = (uint8_t *)q; *cap = want;
    return ZAMD_OK;
}

/* what the threads of one call share */
struct Call {
    std::mutex mu; std::condition_variable cv;
    int arrived = 0, exchanged = 0, err = 0;
    std::vector<uint64_t> h_len;        /* host exchange: the table of lengths, n * per entries */
    std::vector<uint64_t> h_off;        /* and its scan */
};

}  // namespace

/*
 * Ranges already on their devices.  Device i holds n[i] bytes at d_in[i] with head_len[i] bytes of the input in front of
 * them (its first chunk's dictionary: 32768 for every range but the first, which has what the caller's stream has);
 * n[i] must be the partition's (zamd_multi_partition of the total chunk count: `per` whole chunks on every device in
 * front of the last one that has any).  Each range's stream goes to d_out[i] (out_cap[i] bytes of room).  d_table[i]
 * (device i, n_devices * per + 1 entries) receives the offsets of ALL chunks in the whole stream, entry c for chunk c,
 * the entries behind the last chunk repeating the total; h_table (host, same size, may be NULL) the same.  Device i's
 * bytes belong at offset table[i * per] of the stream.
 */
extern "C" int zamd_deflate_multi_dev(zamd_multi *m, const void *const *d_in, const uint64_t *n, const uint32_t *head_len, uint32_t chunk_size,
                                      int level, int strategy, int mem_level, int finish, void *const *d_out, const uint64_t *out_cap,
                                      uint64_t *const *d_table, uint64_t *h_table, uint64_t *total_out)
{
    if (!m || !d_in || !n || !head_len || !d_out || !out_cap || !d_table || chunk_size == 0) return ZAMD_STREAM_ERROR;
    const int nd = (int)m->devs.size();
    uint64_t total_in = 0;
    for (int i = 0; i < nd; i++) total_in += n[i];
    const uint64_t total_chunks = total_in ? (total_in + chunk_size - 1) / chunk_size : 1;
    uint64_t per = 0;
    int last_dev = 0;                  /* the device that holds the end of the input */
    for (int i = 0; i < nd; i++) {
        uint64_t first, count;
        zamd_multi_partition(total_chunks, nd, i, &first, &count, &per);
        const uint64_t lo = first * chunk_size < total_in ? first * chunk_size : total_in;
        const uint64_t hi = (first + count) * chunk_size < total_in ? (first + count) * chunk_size : total_in;
        if (n[i] != hi - lo) return ZAMD_STREAM_ERROR;
        if (count) last_dev = i;
    }
    const uint64_t entries = (uint64_t)nd * per;
    Call call;
    if (!m->use_rccl) { call.h_len.assign(entries, 0); call.h_off.assign(entries + 1, 0); }
    std::vector<uint64_t> totals((size_t)nd, 0);

    auto worker = [&](int i) {
        Dev &d = m->devs[(size_t)i];
        int rc = ZAMD_OK;
        uint64_t first, count, p2;
        zamd_multi_partition(total_chunks, nd, i, &first, &count, &p2);
        /* scratch of the exchange on this device: [local offsets: per + 1][send: per][gathered: entries] */
        const uint64_t tab_words = (per + 1) + per + entries;
        if (hipSetDevice(d.device) != hipSuccess) rc = ZAMD_HIP_ERROR(1);
        if (rc == ZAMD_OK) rc = grow(d, (uint8_t **)&d.d_tab, &d.tab_cap, tab_words * 8);
        uint64_t *d_off = d.d_tab, *d_send = d.d_tab + per + 1, *d_all = d_send + per;
        if (rc == ZAMD_OK && count) {
            /* the empty input is one (final) chunk on device 0 */
            rc = zamd_deflate_shard_dev(d.eng, d_in[i], n[i], head_len[i], chunk_size, level, strategy, mem_level, finish && i == last_dev,
                                        d_out[i], out_cap[i], d_off, &totals[(size_t)i]);
        }
        if (rc == ZAMD_OK) rc = zamd_chunk_lengths_dev(d.eng, count ? d_off : nullptr, count, d_send, per);
        /* all ranges are compressed and every device holds its lengths: exchange them */
        {
            std::unique_lock<std::mutex> g(call.mu);
            if (rc != ZAMD_OK && !call.err) call.err = rc;
            call.arrived++;
            call.cv.notify_all();
            call.cv.wait(g, [&] { return call.arrived >= nd; });
            if (call.err) return;
        }
        if (m->use_rccl) {
            if (i == 0) {
                /* one thread issues the collective for every rank of this process, as a group */
                ncclResult_t r = m->rccl.GroupStart();
                for (int k = 0; k < nd && r == ncclSuccess; k++) {
                    Dev &dk = m->devs[(size_t)k];
                    uint64_t *tab = dk.d_tab;
                    if (hipSetDevice(dk.device) != hipSuccess) { r = ncclUnhandledCudaError; break; }
                    r = m->rccl.AllGather(tab + per + 1, tab + per + 1 + per, (size_t)per, ncclUint64, dk.comm, dk.stream);
                }
                const ncclResult_t r2 = m->rccl.GroupEnd();
                if (r == ncclSuccess) r = r2;
                std::lock_guard<std::mutex> g(call.mu);
                if (r != ncclSuccess && !call.err) { call.err = ZAMD_HIP_ERROR(1); fprintf(stderr, "zamd: ncclAllGather failed: %s\n", m->rccl.GetErrorString ? m->rccl.GetErrorString(r) : "?"); }
                call.exchanged = 1;
                call.cv.notify_all();
            }
            { std::unique_lock<std::mutex> g(call.mu); call.cv.wait(g, [&] { return call.exchanged == 1; }); if (call.err) return; }
            if (hipSetDevice(d.device) != hipSuccess || hipStreamSynchronize(d.stream) != hipSuccess) rc = ZAMD_HIP_ERROR(1);
            /* the place of every chunk: a scan of the gathered lengths, on every device */
            if (rc == ZAMD_OK) rc = zamd_chunk_offsets_dev(d.eng, d_all, entries, 0, d_table[i]);
            if (rc == ZAMD_OK && i == 0 && h_table) rc = zamd_dev_download(d.eng, h_table, d_table[i], (entries + 1) * 8);
            if (rc == ZAMD_OK && i == 0 && total_out) rc = zamd_dev_download(d.eng, total_out, d_table[i] + entries, 8);
        } else {
            /* through the host: every device's lengths into the shared table, one scan, the result back to every device */
            if (per) rc = zamd_dev_download(d.eng, call.h_len.data() + (uint64_t)i * per, d_send, per * 8);
            {
                std::unique_lock<std::mutex> g(call.mu);
                if (rc != ZAMD_OK && !call.err) call.err = rc;
                call.arrived++;
                if (call.arrived == 2 * nd) { zamd_multi_offsets_host(call.h_len.data(), entries, 0, call.h_off.data()); call.exchanged = 1; }
                call.cv.notify_all();
                call.cv.wait(g, [&] { return call.exchanged == 1; });
                if (call.err) return;
            }
            rc = zamd_dev_upload(d.eng, d_table[i], call.h_off.data(), (entries + 1) * 8);
            if (rc == ZAMD_OK && i == 0) {
                if (h_table) memcpy(h_table, call.h_off.data(), (entries + 1) * 8);
                if (total_out) *total_out = call.h_off[entries];
            }
        }
        if (rc != ZAMD_OK) { std::lock_guard<std::mutex> g(call.mu); if (!call.err) call.err = rc; }
    };
    std::vector<std::thread> th;
    try {
        for (int i = 1; i < nd; i++) th.emplace_back(worker, i);
    } catch (...) {
        /* a thread that did not start would leave the others waiting for it */
        { std::lock_guard<std::mutex> g(call.mu); call.err = ZAMD_MEM_ERROR; call.arrived = 2 * nd; call.exchanged = 1; }
        call.cv.notify_all();
        for (std::thread &t : th) t.join();
        return ZAMD_MEM_ERROR;
    }
    worker(0);
    for (std::thread &t : th) t.join();
    return call.err;
}

/*
 * Host buffers in, host buffer out: what deflate() of include/zamd_zlib.h does with an input of several batches when
 * more than one device is configured.  h_dict / dict_len: the input in front of h_in (the stream's history or a preset
 * dictionary, at most 32768 bytes).  h_chunk_off (may be NULL) receives total_chunks + 1 offsets.  adler / crc (may be
 * NULL) are continued over the input as adler32() / crc32() would (the ranges' values joined with the combine rules,
 * adler32.c:136-162, crc32.c:327-408).
 */
extern "C" int zamd_deflate_multi(zamd_multi *m, const void *h_in, uint64_t n, const void *h_dict, uint32_t dict_len, uint32_t chunk_size,
                                  int level, int strategy, int mem_level, int finish, void *h_out, uint64_t out_cap,
                                  uint64_t *h_chunk_off, uint64_t *total_out, uint32_t *adler, uint32_t *crc)
{
    if (!m || (!h_in && n) || !h_out || chunk_size == 0 || dict_len > 32768 || (dict_len && !h_dict)) return ZAMD_STREAM_ERROR;
    const int nd = (int)m->devs.size();
    const uint64_t total_chunks = n ? (n + chunk_size - 1) / chunk_size : 1;
    uint64_t per = 0;
    std::vector<uint64_t> first((size_t)nd), count((size_t)nd), bytes((size_t)nd), cap((size_t)nd);
    std::vector<uint32_t> head((size_t)nd);
    std::vector<const void *> d_in((size_t)nd);
    std::vector<void *> d_out((size_t)nd);
    std::vector<uint64_t *> d_table((size_t)nd);
    for (int i = 0; i < nd; i++) {
        zamd_multi_partition(total_chunks, nd, i, &first[(size_t)i], &count[(size_t)i], &per);
        const uint64_t lo = first[(size_t)i] * chunk_size < n ? first[(size_t)i] * chunk_size : n;
        const uint64_t hi = (first[(size_t)i] + count[(size_t)i]) * chunk_size < n ? (first[(size_t)i] + count[(size_t)i]) * chunk_size : n;
        bytes[(size_t)i] = hi - lo;
        head[(size_t)i] = i == 0 ? dict_len : (uint32_t)(lo < 32768 ? lo + (dict_len < 32768 - lo ? dict_len : 32768 - lo) : 32768);
        cap[(size_t)i] = zamd_deflate_bound(bytes[(size_t)i], chunk_size, mem_level);
    }
    const uint64_t entries = (uint64_t)nd * per;
    std::vector<uint64_t> table(entries + 1, 0);
    std::vector<uint32_t> part_a((size_t)nd, 1), part_c((size_t)nd, 0);
    std::mutex mu; int err = 0;
    auto note = [&](int rc) { if (rc != ZAMD_OK) { std::lock_guard<std::mutex> g(mu); if (!err) err = rc; } };
    /* uploads, one thread per device */
    {
        std::vector<std::thread> th;
        auto up = [&](int i) {
            Dev &d = m->devs[(size_t)i];
            int rc = grow(d, &d.d_in, &d.in_cap, 32768 + bytes[(size_t)i] + 64);
            if (rc == ZAMD_OK) rc = grow(d, &d.d_out, &d.out_cap, cap[(size_t)i] + 64);
            if (rc == ZAMD_OK) rc = grow(d, (uint8_t **)&d.d_tab, &d.tab_cap, ((per + 1) + per + entries + (entries + 1)) * 8);
            const uint64_t lo = first[(size_t)i] * chunk_size < n ? first[(size_t)i] * chunk_size : n;
            const uint32_t h = head[(size_t)i];
            if (rc == ZAMD_OK && h) {
                /* the window in front of the range: the end of the dictionary, then the input in front of the range */
                const uint32_t from_in = (uint32_t)(lo < h ? lo : h), from_dict = h - from_in;
                if (from_dict) rc = zamd_dev_upload(d.eng, d.d_in + 32768 - h, (const uint8_t *)h_dict + dict_len - from_dict, from_dict);
                if (rc == ZAMD_OK && from_in) rc = zamd_dev_upload(d.eng, d.d_in + 32768 - from_in, (const uint8_t *)h_in + lo - from_in, from_in);
            }
            if (rc == ZAMD_OK && bytes[(size_t)i]) rc = zamd_dev_upload(d.eng, d.d_in + 32768, (const uint8_t *)h_in + lo, bytes[(size_t)i]);
            note(rc);
        };
        try { for (int i = 1; i < nd; i++) th.emplace_back(up, i); } catch (...) { note(ZAMD_MEM_ERROR); }
        up(0);
        for (std::thread &t : th) t.join();
        if (err) return err;
    }
    for (int i = 0; i < nd; i++) {
        Dev &d = m->devs[(size_t)i];
        d_in[(size_t)i] = d.d_in + 32768; d_out[(size_t)i] = d.d_out;
        d_table[(size_t)i] = d.d_tab + (per + 1) + per + entries;
    }
    uint64_t total = 0;
    int rc = zamd_deflate_multi_dev(m, d_in.data(), bytes.data(), head.data(), chunk_size, level, strategy, mem_level, finish, d_out.data(), cap.data(),
                                    d_table.data(), table.data(), &total);
    if (rc != ZAMD_OK) return rc;
    if (total > out_cap) return ZAMD_BUF_ERROR;
    /* downloads to each range's place, and the ranges' check values */
    {
        std::vector<std::thread> th;
        auto down = [&](int i) {
            Dev &d = m->devs[(size_t)i];
            int r = ZAMD_OK;
            const uint64_t at = table[(uint64_t)i * per], end = table[(uint64_t)i * per + count[(size_t)i]];
            if (end > at) r = zamd_dev_download(d.eng, (uint8_t *)h_out + at, d.d_out, end - at);
            if (r == ZAMD_OK && adler && bytes[(size_t)i]) r = zamd_adler32_dev(d.eng, 1, d.d_in + 32768, bytes[(size_t)i], &part_a[(size_t)i]);
            if (r == ZAMD_OK && crc && bytes[(size_t)i]) r = zamd_crc32_dev(d.eng, 0, d.d_in + 32768, bytes[(size_t)i], &part_c[(size_t)i]);
            note(r);
        };
        try { for (int i = 1; i < nd; i++) th.emplace_back(down, i); } catch (...) { note(ZAMD_MEM_ERROR); }
        down(0);
        for (std::thread &t : th) t.join();
        if (err) return err;
    }
    for (int i = 0; i < nd; i++) {
        if (!bytes[(size_t)i]) continue;
        if (adler) *adler = zamd_adler32_combine(*adler, part_a[(size_t)i], (int64_t)bytes[(size_t)i]);
        if (crc) *crc = zamd_crc32_combine(*crc, part_c[(size_t)i], (int64_t)bytes[(size_t)i]);
    }
    if (h_chunk_off) memcpy(h_chunk_off, table.data(), (total_chunks + 1) * 8);
    if (total_out) *total_out = total;
    return ZAMD_OK;
}
