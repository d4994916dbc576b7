// nns_multi.hip — the in-library multi-GPU search: the V8/V9 analogue behind the C ABI.
//
// Reference (core.cu:761-853, 965-1057): one OpenMP thread per GPU, contiguous
// ceil(n / G) ref shards (core.cu:781-791), all queries uploaded to every GPU, per-GPU
// INDICES gathered into a host vector and re-ranked on the host (wrong for m > 1: F4,
// arrival order nondeterministic: F5).
//
// Here: one host thread per GPU does upload + index build + search and leaves packed
// (V0 distance, global index) keys on its device; the exchange is ONE min all-reduce of
// m uint64 keys over RCCL (ncclCommInitAll + grouped ncclAllReduce, xGMI) — a MINLOC in
// one integer min, independent of arrival order, and V0's answer for every m.  RCCL is
// dlopen()ed on first use (a process that already has torch's librccl.so.1 loaded shares
// it); if it cannot be loaded or initialised the keys are merged through the host with
// the same operator.
//
// (bench.py's N > 1 mode runs one PROCESS per GPU with torch.distributed instead, as the
// round's bench contract requires; both share the per-shard path and the key algebra.)
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "nns_internal.h"

namespace nns {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

static RcclApi &rccl()
{
    static RcclApi api;
    static bool tried = false;
    if (tried) return api;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names) {
        api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
        if (api.handle) break;
    }
    if (!api.handle) return api;
    api.CommInitAll = (decltype(api.CommInitAll))dlsym(api.handle, "ncclCommInitAll");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.handle, "ncclAllReduce");
    api.GroupStart = (decltype(api.GroupStart))dlsym(api.handle, "ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.handle, "ncclGroupEnd");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
    api.ok = api.CommInitAll && api.CommDestroy && api.AllReduce && api.GroupStart && api.GroupEnd;
    return api;
}

struct ShardJob {
    int device = 0;
    int beg = 0, cnt = 0;
    float *q_d = nullptr, *r_d = nullptr;
    nns_key *keys = nullptr;
    int rc = NNS_OK;
    char err[256] = "";
};

// one GPU's share: upload, build, search -> keys on that device (core.cu:793-819 per thread)
static void run_shard(ShardJob *job, int k, int m, const float *q, const float *r, unsigned flags)
{
    auto fail = [&](int rc, const char *what) {
        job->rc = rc;
        snprintf(job->err, sizeof(job->err), "device %d: %s (%s)", job->device, what, nns_last_error());
    };
    if (hipSetDevice(job->device) != hipSuccess) return fail(NNS_ERR_HIP, "hipSetDevice");
    const size_t qb = (size_t)m * k * sizeof(float), rb = (size_t)job->cnt * k * sizeof(float);
    if (pool_alloc(&job->q_d, qb) != hipSuccess || pool_alloc(&job->r_d, rb) != hipSuccess ||
        pool_alloc(&job->keys, (size_t)m * sizeof(nns_key)) != hipSuccess)
        return fail(NNS_ERR_NOMEM, "device allocation");
    if (hipMemcpy(job->q_d, q, qb, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(job->r_d, r + (size_t)job->beg * k, rb, hipMemcpyHostToDevice) != hipSuccess)
        return fail(NNS_ERR_HIP, "H2D copy");
    nns_index *ix = nullptr;
    int rc = nns_index_create(&ix, job->device, k, job->cnt, job->r_d, job->beg, flags, nullptr);
    if (rc == NNS_OK) rc = nns_index_search(ix, m, job->q_d, job->keys, nullptr);
    if (rc == NNS_OK && hipDeviceSynchronize() != hipSuccess) rc = NNS_ERR_HIP;
    nns_index_destroy(ix);
    if (rc != NNS_OK) fail(rc, "search");
}

}  // namespace nns

using namespace nns;

extern "C" int nns_search_f32_multi(int k, int m, int n, const float *s_points, const float *r_points,
                                    int *idx_out, float *dist_out, int num_devices, unsigned flags)
{
    if (k <= 0 || m <= 0 || n <= 0 || !s_points || !r_points || !idx_out) {
        set_error("nns_search_f32_multi: k, m, n must be > 0 and pointers non-null");
        return NNS_ERR_INVALID;
    }
    if (flags & NNS_REFS_SOA) {
        set_error("nns_search_f32_multi: NNS_REFS_SOA is not supported (shards are ranges of point-major refs)");
        return NNS_ERR_UNSUPPORTED;
    }
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible < 1) {
        set_error("no HIP device visible (the HIP path has no CPU fallback)");
        return NNS_ERR_NODEVICE;
    }
    int G = num_devices <= 0 ? visible : num_devices;   // <= 0: all visible GPUs (core.cu:769-770)
    // NNS_MULTI_VIRTUAL: rehearsal on fewer GPUs than shards — shard g runs on device
    // g % visible (several host threads per device) and the keys merge through the host
    const bool virt = (flags & NNS_MULTI_VIRTUAL) != 0;
    flags &= ~(unsigned)NNS_MULTI_VIRTUAL;
    if (G > visible && !virt) G = visible;
    if (G > n) G = n;                                   // core.cu:771-772
    // the reference keeps small problems on one GPU (core.cu:775-777)
    const int64_t small_n = ((int64_t)m << 10) < (1 << 18) ? ((int64_t)m << 10) : (1 << 18);
    if (G == 1 || (!virt && n <= small_n))
        return nns_search_f32_ex(k, m, n, s_points, r_points, idx_out, dist_out, 1, flags, 0);

    const int per = divup(n, G);                        // contiguous shards (core.cu:781-791)
    std::vector<ShardJob> jobs;
    for (int g = 0; g < G; ++g) {
        const int beg = g * per;
        const int cnt = (beg + per <= n) ? per : n - beg;
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
        for (int g = 0; g < G; ++g) th.emplace_back(run_shard, &jobs[g], k, m, s_points, r_points, flags);
        for (auto &t : th) t.join();
    }
    int rc = NNS_OK;
    for (int g = 0; g < G; ++g)
        if (jobs[g].rc != NNS_OK) {
            rc = jobs[g].rc;
            set_error("nns_search_f32_multi: %s", jobs[g].err);
            break;
        }

    // ---- the exchange: one min all-reduce of the packed keys ----------------------------
    bool reduced = false;
    if (rc == NNS_OK && G > 1 && G <= visible) {   // distinct devices only: RCCL rejects duplicates
        RcclApi &api = rccl();
        if (api.ok) {
            std::vector<ncclComm_t> comms(G);
            std::vector<int> devs(G);
            for (int g = 0; g < G; ++g) devs[g] = jobs[g].device;
            if (api.CommInitAll(comms.data(), G, devs.data()) == ncclSuccess) {
                bool ok = api.GroupStart() == ncclSuccess;
                for (int g = 0; g < G && ok; ++g) {
                    ok = hipSetDevice(jobs[g].device) == hipSuccess &&
                         api.AllReduce(jobs[g].keys, jobs[g].keys, (size_t)m, ncclUint64, ncclMin, comms[g], nullptr) ==
                             ncclSuccess;
                }
                ok = (api.GroupEnd() == ncclSuccess) && ok;
                for (int g = 0; g < G; ++g) {
                    (void)hipSetDevice(jobs[g].device);
                    if (hipDeviceSynchronize() != hipSuccess) ok = false;
                }
                for (int g = 0; g < G; ++g) (void)api.CommDestroy(comms[g]);
                reduced = ok;
            }
        }
    }
    if (rc == NNS_OK) {
        (void)hipSetDevice(jobs[0].device);
        if (!reduced && G > 1) {
            // no RCCL: same operator through the host (keys are 8 B per query)
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
        if (rc == NNS_OK) rc = nns_keys_unpack(jobs[0].keys, m, idx_d, dist_d, nullptr);
        if (rc == NNS_OK &&
            (hipMemcpy(idx_out, idx_d, (size_t)m * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
             (dist_out && hipMemcpy(dist_out, dist_d, (size_t)m * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)))
            rc = NNS_ERR_HIP;
        pool_free(idx_d);
        pool_free(dist_d);
        if (rc == NNS_ERR_HIP) set_error("nns_search_f32_multi: merge/unpack failed: %s", hipGetErrorString(hipGetLastError()));
    }
    for (int g = 0; g < G; ++g) {
        (void)hipSetDevice(jobs[g].device);
        (void)hipDeviceSynchronize();   // the blocks go back to the pool: nothing may still use them
        pool_free(jobs[g].q_d);
        pool_free(jobs[g].r_d);
        pool_free(jobs[g].keys);
    }
    (void)hipSetDevice(0);
    return rc;
}
