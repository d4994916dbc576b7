// nns_multi.hip — multi-GPU search behind the C ABI: the V8/V9 analogue (one process, one host
// thread per GPU) and the exchange primitive of the one-process-per-GPU form (nns_comm_*).
//
// Reference (core.cu:761-853, 965-1057): one OpenMP thread per GPU, contiguous
// ceil(n / G) ref shards (core.cu:781-791), all queries uploaded to every GPU, per-GPU
// INDICES gathered into a host vector and re-ranked on the host (wrong for m > 1: F4,
// arrival order nondeterministic: F5).
//
// Here: every GPU leaves packed (V0 distance, global index) keys in its own memory and the
// exchange is ONE ncclAllReduce(ncclUint64, ncclMin) of m keys over xGMI — a MINLOC in one
// integer min, independent of arrival order, and V0's answer for every m.  Both forms go
// through the same call site (allreduce_min below):
//   * nns_search_{f32,bf16}_multi: one host thread per GPU does upload + index build + search;
//     the communicators of a device set are created once (ncclCommInitAll) and CACHED for the
//     life of the process (a communicator costs hundreds of ms to set up, a C3-sized shard
//     120 ms to search); nns_shutdown() destroys them.
//   * nns_comm_*: one rank per process (bench.py --gpus N, torch.distributed.run): rank 0 draws
//     an id (nns_comm_unique_id), the caller carries its 128 bytes to the other ranks by any
//     means, every rank joins (nns_comm_create) and calls nns_comm_allreduce_min on its stream.
// RCCL is dlopen()ed on first use (a process that already has torch's librccl.so.1 loaded
// shares it).  If it cannot be loaded the thread-per-GPU form merges the keys through the host
// with the same operator; nns_comm_* then fail with NNS_ERR_UNSUPPORTED.
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <atomic>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "nns_internal.h"

namespace nns {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

// loaded exactly once, also under concurrent first calls (C++11 static initialisation)
static const RcclApi &rccl()
{
    static const RcclApi api = [] {
        RcclApi a;
        // A process that already has an RCCL mapped (torch's bundled librccl.so) must share it: two RCCL
        // instances driving the same GPUs is asking for trouble.  RTLD_NOLOAD finds a mapped one; only
        // then load one ourselves (RTLD_LOCAL: no symbol interposition with anybody).
        const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            a.handle = dlopen(nm, RTLD_NOW | RTLD_NOLOAD);
            if (a.handle) break;
        }
        for (int i = 1; i < 3 && !a.handle; ++i) a.handle = dlopen(names[i], RTLD_NOW | RTLD_LOCAL);
        if (!a.handle) a.handle = dlopen(names[0], RTLD_NOW | RTLD_LOCAL);
        if (!a.handle) return a;
        a.CommInitAll = (decltype(a.CommInitAll))dlsym(a.handle, "ncclCommInitAll");
        a.CommInitRank = (decltype(a.CommInitRank))dlsym(a.handle, "ncclCommInitRank");
        a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(a.handle, "ncclGetUniqueId");
        a.CommCount = (decltype(a.CommCount))dlsym(a.handle, "ncclCommCount");
        a.CommDestroy = (decltype(a.CommDestroy))dlsym(a.handle, "ncclCommDestroy");
        a.AllReduce = (decltype(a.AllReduce))dlsym(a.handle, "ncclAllReduce");
        a.GroupStart = (decltype(a.GroupStart))dlsym(a.handle, "ncclGroupStart");
        a.GroupEnd = (decltype(a.GroupEnd))dlsym(a.handle, "ncclGroupEnd");
        a.GetErrorString = (decltype(a.GetErrorString))dlsym(a.handle, "ncclGetErrorString");
        a.ok = a.CommInitAll && a.CommInitRank && a.GetUniqueId && a.CommCount && a.CommDestroy && a.AllReduce &&
               a.GroupStart && a.GroupEnd;
        return a;
    }();
    return api;
}

static const char *rccl_err(ncclResult_t r)
{
    const RcclApi &api = rccl();
    return api.GetErrorString ? api.GetErrorString(r) : "rccl error";
}

// THE exchange: keys[i] = min over ranks of keys[i] (in place), both multi-GPU forms end here
static ncclResult_t allreduce_min(nns_key *keys_dev, int m, ncclComm_t comm, hipStream_t st)
{
    return rccl().AllReduce(keys_dev, keys_dev, (size_t)m, ncclUint64, ncclMin, comm, st);
}

// ---- thread-per-GPU form: communicators cached per device list ---------------------------
struct CommSet {
    std::vector<ncclComm_t> comms;
};
struct CommCache {
    std::mutex mu;   // held over create AND over the collective: one multi call at a time per process
    std::map<std::vector<int>, CommSet> sets;
};
// ranks of the last grouped all-reduce that completed (0: none yet, or it failed): lets a test see that the
// collective branch really ran (nns_multi_last_exchange_ranks)
static std::atomic<int> g_last_exchange_ranks{0};

static CommCache &comm_cache()
{
    static CommCache *c = new CommCache();   // leaked on purpose (no destructor order games at exit)
    return *c;
}

struct ShardJob {
    int device = 0;
    int beg = 0, cnt = 0;
    char *q_d = nullptr, *r_d = nullptr;
    nns_key *keys = nullptr;
    hipStream_t st = nullptr;   // the shard's non-blocking stream: search, exchange and unpack are enqueued here
    int rc = NNS_OK;
    char err[256] = "";
};

// one GPU's share: upload, build, search -> keys on that device (core.cu:793-819 per thread)
static void run_shard(ShardJob *job, int k, int m, int n, const void *q, const void *r, int bf16, unsigned flags)
{
    auto fail = [&](int rc, const char *what) {
        job->rc = rc;
        snprintf(job->err, sizeof(job->err), "device %d: %s (%s)", job->device, what, nns_last_error());
    };
    if (hipSetDevice(job->device) != hipSuccess) return fail(NNS_ERR_HIP, "hipSetDevice");   // (this thread's device)
    job->st = lib_stream_acquire();   // (nullptr: the default stream)
    const size_t esz = bf16 ? sizeof(uint16_t) : sizeof(float);
    const size_t qb = (size_t)m * k * esz, rb = (size_t)job->cnt * k * esz;
    if (pool_alloc(&job->q_d, qb) != hipSuccess || pool_alloc(&job->r_d, rb) != hipSuccess ||
        pool_alloc(&job->keys, (size_t)m * sizeof(nns_key)) != hipSuccess)
        return fail(NNS_ERR_NOMEM, "device allocation");
    bool up = hipMemcpy(job->q_d, q, qb, hipMemcpyHostToDevice) == hipSuccess;
    if (flags & NNS_REFS_SOA) {
        // dimension-major host array [k][n]: this shard is columns beg .. beg + cnt of every row ->
        // a dense [k][cnt] device array, which the index transposes (NNS_REFS_SOA)
        up = up && hipMemcpy2D(job->r_d, (size_t)job->cnt * esz, (const char *)r + (size_t)job->beg * esz,
                               (size_t)n * esz, (size_t)job->cnt * esz, (size_t)k, hipMemcpyHostToDevice) == hipSuccess;
    } else if (up && upload_overlap_pays(k, m, job->cnt, bf16, flags, rb)) {
        // a shard worth it goes up in chunks that are searched while the next one is copied (nns_api.hip)
        nns_key *tmp = nullptr;
        if (pool_alloc(&tmp, (size_t)m * sizeof(nns_key)) != hipSuccess) return fail(NNS_ERR_NOMEM, "device allocation");
        const int orc = search_range_overlapped(job->device, k, m, job->cnt, job->q_d, (const char *)r + (size_t)job->beg * k * esz,
                                                job->r_d, bf16, job->beg, flags, job->keys, tmp);
        pool_free(tmp);   // (the overlapped core returns with its stream waited for)
        if (orc == NNS_OK) return;
        // (NOMEM: four chunk indexes carry four query-side workspaces; the plain single-index path may still fit)
        if (orc != NNS_ERR_UNSUPPORTED && orc != NNS_ERR_NOMEM) return fail(orc, "search (overlapped upload)");
        up = hipMemcpy(job->r_d, (const char *)r + (size_t)job->beg * k * esz, rb, hipMemcpyHostToDevice) == hipSuccess;
    } else {
        up = up && hipMemcpy(job->r_d, (const char *)r + (size_t)job->beg * k * esz, rb, hipMemcpyHostToDevice) == hipSuccess;
    }
    if (!up) return fail(NNS_ERR_HIP, "H2D copy");
    if (order_after_default_stream(job->st) != NNS_OK) return fail(NNS_ERR_HIP, "event edge upload -> search");
    nns_index *ix = nullptr;
    int rc = bf16 ? nns_index_create_bf16(&ix, job->device, k, job->cnt, (const uint16_t *)job->r_d, job->beg, flags, job->st)
                  : nns_index_create(&ix, job->device, k, job->cnt, (const float *)job->r_d, job->beg, flags, job->st);
    if (rc == NNS_OK)
        rc = bf16 ? nns_index_search_bf16(ix, m, (const uint16_t *)job->q_d, job->keys, job->st)
                  : nns_index_search(ix, m, (const float *)job->q_d, job->keys, job->st);
    if (rc == NNS_OK && hipStreamSynchronize(job->st) != hipSuccess) rc = NNS_ERR_HIP;
    nns_index_destroy(ix);
    if (rc != NNS_OK) fail(rc, "search");
}

static int search_multi_impl(int k, int m, int n, const void *s_points, const void *r_points, int bf16, int *idx_out,
                             float *dist_out, int num_devices, unsigned flags)
{
    if (k <= 0 || m <= 0 || n <= 0 || !s_points || !r_points || !idx_out) {
        set_error("nns_search_multi: k, m, n must be > 0 and pointers non-null");
        return NNS_ERR_INVALID;
    }
    // the same limits as the single-device entry points (nns.h), before any thread or allocation
    if (m > NNS_MAX_POINTS || n > NNS_MAX_POINTS) {
        set_error("nns_search_multi: m = %d / n = %d exceeds NNS_MAX_POINTS (%d)", m, n, NNS_MAX_POINTS);
        return NNS_ERR_INVALID;
    }
    if ((int64_t)k * m > 0x7FFFFFFFll * 4 || (int64_t)k * n > 0x7FFFFFFFll * 4) {
        set_error("nns_search_multi: point set too large for one call");
        return NNS_ERR_INVALID;
    }
    DeviceScope keep_device;   // the caller's current device is restored on every return path
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) {
        set_error("no HIP device visible (the HIP path has no CPU fallback)");
        return NNS_ERR_NODEVICE;
    }
    int G = num_devices <= 0 ? visible : num_devices;   // <= 0: all visible GPUs (core.cu:769-770)
    // NNS_MULTI_VIRTUAL: rehearsal on fewer GPUs than shards — shard g runs on device
    // g % visible (several host threads per device) and the keys merge through the host
    const bool virt = (flags & NNS_MULTI_VIRTUAL) != 0;
    // NNS_MULTI_FORCE_COLLECTIVE (tests): no single-GPU shortcut — also ONE shard goes through the thread-per-GPU
    // body, ncclCommInitAll and the grouped all-reduce, so that branch runs on a one-GPU box too
    const bool force = (flags & NNS_MULTI_FORCE_COLLECTIVE) != 0;
    flags &= ~(unsigned)(NNS_MULTI_VIRTUAL | NNS_MULTI_FORCE_COLLECTIVE);
    if (G > visible && !virt) G = visible;
    if (G > n) G = n;                                   // core.cu:771-772
    // the reference keeps small problems on one GPU (core.cu:775-777)
    const int64_t small_n = ((int64_t)m << 10) < (1 << 18) ? ((int64_t)m << 10) : (1 << 18);
    if (!force && (G == 1 || (!virt && n <= small_n)))
        return bf16 ? nns_search_bf16_ex(k, m, n, (const uint16_t *)s_points, (const uint16_t *)r_points, idx_out, dist_out, 1, flags, 0)
                    : nns_search_f32_ex(k, m, n, (const float *)s_points, (const float *)r_points, idx_out, dist_out, 1, flags, 0);

    const int per = divup(n, G);                        // contiguous shards (core.cu:781-791)
    std::vector<ShardJob> jobs;
    for (int g = 0; g < G; ++g) {
        const int beg = g * per;
        const int cnt = ((int64_t)beg + per <= n) ? per : n - beg;
        if (cnt <= 0) break;
        ShardJob j;
        j.device = g % visible;
        j.beg = beg;
        j.cnt = cnt;
        jobs.push_back(j);
    }
    G = (int)jobs.size();
    {
        std::vector<std::thread> th;
        for (int g = 0; g < G; ++g) th.emplace_back(run_shard, &jobs[g], k, m, n, s_points, r_points, bf16, flags);
        for (auto &t : th) t.join();
    }
    int rc = NNS_OK;
    for (int g = 0; g < G; ++g)
        if (jobs[g].rc != NNS_OK) {
            rc = jobs[g].rc;
            set_error("nns_search_multi: %s", jobs[g].err);
            break;
        }

    // ---- the exchange: one min all-reduce of the packed keys ----------------------------
    bool reduced = false;
    if (rc == NNS_OK && (G > 1 || force) && G <= visible) {   // distinct devices only: RCCL rejects duplicates
        const RcclApi &api = rccl();
        if (api.ok) {
            std::vector<int> devs(G);
            for (int g = 0; g < G; ++g) devs[g] = jobs[g].device;
            CommCache &cc = comm_cache();
            std::lock_guard<std::mutex> lk(cc.mu);
            auto it = cc.sets.find(devs);
            if (it == cc.sets.end()) {
                CommSet cs;
                cs.comms.resize(G);
                if (api.CommInitAll(cs.comms.data(), G, devs.data()) == ncclSuccess) it = cc.sets.emplace(devs, std::move(cs)).first;
            }
            if (it != cc.sets.end()) {
                const std::vector<ncclComm_t> &comms = it->second.comms;
                bool ok = api.GroupStart() == ncclSuccess;
                for (int g = 0; g < G && ok; ++g)
                    ok = hipSetDevice(jobs[g].device) == hipSuccess && allreduce_min(jobs[g].keys, m, comms[g], jobs[g].st) == ncclSuccess;
                ok = (api.GroupEnd() == ncclSuccess) && ok;
                for (int g = 0; g < G; ++g) {
                    (void)hipSetDevice(jobs[g].device);
                    if (hipStreamSynchronize(jobs[g].st) != hipSuccess) ok = false;
                }
                reduced = ok;
                g_last_exchange_ranks.store(ok ? G : 0, std::memory_order_relaxed);
                if (!ok) {   // a failed collective leaves the communicators in an unknown state: drop them
                    for (ncclComm_t c : comms) (void)api.CommDestroy(c);
                    cc.sets.erase(it);
                    (void)hipGetLastError();
                }
            }
        }
    }
    if (rc == NNS_OK) {
        (void)hipSetDevice(jobs[0].device);
        if (!reduced && G > 1) {
            // no RCCL (or virtual shards sharing a device): same operator through the host (8 B per query)
            std::vector<nns_key> acc((size_t)m), tmp((size_t)m);
            if (hipMemcpy(acc.data(), jobs[0].keys, (size_t)m * sizeof(nns_key), hipMemcpyDeviceToHost) != hipSuccess)
                rc = NNS_ERR_HIP;
            for (int g = 1; g < G && rc == NNS_OK; ++g) {
                (void)hipSetDevice(jobs[g].device);
                if (hipMemcpy(tmp.data(), jobs[g].keys, (size_t)m * sizeof(nns_key), hipMemcpyDeviceToHost) != hipSuccess)
                    rc = NNS_ERR_HIP;
                for (int i = 0; i < m; ++i) acc[i] = tmp[i] < acc[i] ? tmp[i] : acc[i];
            }
            (void)hipSetDevice(jobs[0].device);
            if (rc == NNS_OK &&
                hipMemcpy(jobs[0].keys, acc.data(), (size_t)m * sizeof(nns_key), hipMemcpyHostToDevice) != hipSuccess)
                rc = NNS_ERR_HIP;
        }
        // unpack on device 0 (every device holds the reduced keys after the all-reduce)
        int *idx_d = nullptr;
        float *dist_d = nullptr;
        if (rc == NNS_OK && (pool_alloc(&idx_d, (size_t)m * sizeof(int)) != hipSuccess ||
                             pool_alloc(&dist_d, (size_t)m * sizeof(float)) != hipSuccess))
            rc = NNS_ERR_NOMEM;
        if (rc == NNS_OK) rc = order_after_default_stream(jobs[0].st);   // (the host-merge upload above, if any)
        if (rc == NNS_OK) rc = nns_keys_unpack(jobs[0].keys, m, idx_d, dist_d, jobs[0].st);
        if (rc == NNS_OK && hipStreamSynchronize(jobs[0].st) != hipSuccess) rc = NNS_ERR_HIP;
        if (rc == NNS_OK &&
            (hipMemcpy(idx_out, idx_d, (size_t)m * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
             (dist_out && hipMemcpy(dist_out, dist_d, (size_t)m * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)))
            rc = NNS_ERR_HIP;
        void *const outs[] = {idx_d, dist_d};
        pool_free_after(outs, 2, jobs[0].st);
        if (rc == NNS_ERR_HIP) set_error("nns_search_multi: merge/unpack failed: %s", hipGetErrorString(hipGetLastError()));
    }
    for (int g = 0; g < G; ++g) {
        (void)hipSetDevice(jobs[g].device);
        // the blocks go back to the pool behind an event on the shard's stream (no device-wide wait)
        void *const blocks[] = {jobs[g].q_d, jobs[g].r_d, jobs[g].keys};
        pool_free_after(blocks, 3, jobs[g].st);
        lib_stream_release(jobs[g].st);
    }
    return rc;   // (keep_device restores the caller's device: round 2 left device 0 selected)
}

}  // namespace nns

using namespace nns;

// one rank of a one-process-per-GPU job
struct nns_comm {
    ncclComm_t comm = nullptr;
    int device = 0, nranks = 0, rank = 0;
    hipStream_t last_stream = nullptr;   // of the last all-reduce: destroy waits for that stream, not for the device
    bool used = false;
};

extern "C" {

int nns_search_f32_multi(int k, int m, int n, const float *s_points, const float *r_points, int *idx_out,
                         float *dist_out, int num_devices, unsigned flags)
{
    return search_multi_impl(k, m, n, s_points, r_points, 0, idx_out, dist_out, num_devices, flags);
}

int nns_search_bf16_multi(int k, int m, int n, const uint16_t *s_points, const uint16_t *r_points, int *idx_out,
                          float *dist_out, int num_devices, unsigned flags)
{
    return search_multi_impl(k, m, n, s_points, r_points, 1, idx_out, dist_out, num_devices, flags);
}

int nns_comm_unique_id(void *id_out, size_t id_bytes)
{
    static_assert(NNS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "the id is RCCL's ncclUniqueId");
    if (!id_out || id_bytes < NNS_COMM_ID_BYTES) {
        set_error("nns_comm_unique_id: needs a buffer of NNS_COMM_ID_BYTES (%d) bytes", NNS_COMM_ID_BYTES);
        return NNS_ERR_INVALID;
    }
    const RcclApi &api = rccl();
    if (!api.ok) {
        set_error("nns_comm_unique_id: librccl could not be loaded");
        return NNS_ERR_UNSUPPORTED;
    }
    ncclUniqueId id;
    const ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) {
        set_error("ncclGetUniqueId: %s", rccl_err(r));
        return NNS_ERR_HIP;
    }
    memcpy(id_out, &id, NNS_COMM_ID_BYTES);
    return NNS_OK;
}

int nns_comm_create(nns_comm **out, const void *id, size_t id_bytes, int nranks, int rank, int device)
{
    if (!out || !id || id_bytes < NNS_COMM_ID_BYTES || nranks < 1 || rank < 0 || rank >= nranks) {
        set_error("nns_comm_create: bad arguments (nranks=%d rank=%d id_bytes=%zu)", nranks, rank, id_bytes);
        return NNS_ERR_INVALID;
    }
    *out = nullptr;
    const RcclApi &api = rccl();
    if (!api.ok) {
        set_error("nns_comm_create: librccl could not be loaded");
        return NNS_ERR_UNSUPPORTED;
    }
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt < 1) {
        set_error("no HIP device visible (the HIP path has no CPU fallback)");
        return NNS_ERR_NODEVICE;
    }
    if (device < 0 || device >= cnt) {
        set_error("nns_comm_create: device %d out of range (%d visible)", device, cnt);
        return NNS_ERR_INVALID;
    }
    DeviceScope keep_device;
    NNS_HIP(hipSetDevice(device));
    ncclUniqueId uid;
    memcpy(&uid, id, NNS_COMM_ID_BYTES);
    nns_comm *c = new (std::nothrow) nns_comm();
    if (!c) return NNS_ERR_NOMEM;
    const ncclResult_t r = api.CommInitRank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        set_error("ncclCommInitRank(nranks=%d, rank=%d): %s", nranks, rank, rccl_err(r));
        delete c;
        return NNS_ERR_HIP;
    }
    c->device = device;
    c->nranks = nranks;
    c->rank = rank;
    *out = c;
    return NNS_OK;
}

int nns_comm_size(nns_comm *c)
{
    if (!c) return 0;
    int n = 0;
    if (rccl().CommCount(c->comm, &n) != ncclSuccess) return 0;
    return n;
}

int nns_comm_allreduce_min(nns_comm *c, nns_key *keys_dev, int m, void *stream)
{
    if (!c || !keys_dev || m <= 0) {
        set_error("nns_comm_allreduce_min: bad arguments");
        return NNS_ERR_INVALID;
    }
    DeviceScope keep_device;
    NNS_HIP(hipSetDevice(c->device));
    c->last_stream = (hipStream_t)stream;
    c->used = true;
    const ncclResult_t r = allreduce_min(keys_dev, m, c->comm, (hipStream_t)stream);
    if (r != ncclSuccess) {
        set_error("ncclAllReduce(uint64, min, %d keys): %s", m, rccl_err(r));
        return NNS_ERR_HIP;
    }
    return NNS_OK;
}

int nns_comm_destroy(nns_comm *c)
{
    if (!c) return NNS_OK;
    DeviceScope keep_device;
    (void)hipSetDevice(c->device);
    if (c->used && hipStreamSynchronize(c->last_stream) != hipSuccess) {   // its last collective, not the device
        (void)hipGetLastError();
        (void)hipDeviceSynchronize();   // (the stream is gone: the blunt way)
    }
    if (c->comm) (void)rccl().CommDestroy(c->comm);
    delete c;
    return NNS_OK;
}

int nns_multi_last_exchange_ranks(void) { return g_last_exchange_ranks.load(std::memory_order_relaxed); }

int nns_shutdown(void)
{
    DeviceScope keep_device;
    {
        CommCache &cc = comm_cache();
        std::lock_guard<std::mutex> lk(cc.mu);
        for (auto &s : cc.sets)
            for (ncclComm_t c : s.second.comms) (void)rccl().CommDestroy(c);
        cc.sets.clear();
    }
    stager_release();
    (void)nns_trim();
    return NNS_OK;
}

}  // extern "C"
