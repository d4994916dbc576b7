// dev_pool.hip — a small caching allocator for the library's device workspaces, and the
// library's own (non-blocking) streams.
//
// The reference's vN::cudaCall allocates and frees every buffer on every call
// (thrust::device_vector / cudaMalloc in core.cu:123-151, 634-697, 793-802).  On this runtime a
// hipMalloc + hipFree pair costs tens of microseconds and a whole-call search makes ~16 of them,
// which is most of the wall clock of the reference driver's small samples (main.cu:38-51).
// Freed blocks are therefore parked here per device and handed back to the next request of a
// similar size; nns_trim() (or NNS_POOL_BYTES=0) returns everything to the runtime.
//
// Blocks are rounded up to a size class (1/8-octave steps above 64 KiB) so that a call sequence
// with slowly varying sizes still hits.  A block is reused only for requests of at least half its
// size.  The pool never holds more than NNS_POOL_BYTES (default 16 GiB of the 288 GB) per process.
//
// Two ways to free:
//   pool_free(ptr)            the caller has already waited for the work that used the block;
//   pool_free_after(ptr, st)  the block may still be in use by work enqueued on stream `st`: an event is
//                             recorded there and the block becomes reusable when it has fired.  No host
//                             wait, and in particular no hipDeviceSynchronize(): an application's kernels
//                             on other streams are never waited for (round 2 synchronised the device at
//                             every destroy / workspace regrow / whole-call exit).
#include <stdlib.h>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "nns_internal.h"

namespace nns {

namespace {

// blocks freed together behind ONE event (an index's workspaces at destroy)
struct Pending {
    std::vector<std::pair<void *, size_t>> blocks;   // (ptr, class size)
    hipEvent_t ev;
};

struct Pool {
    std::mutex mu;
    // device -> (class size -> blocks)
    std::map<int, std::multimap<size_t, void *>> free_blocks;
    std::unordered_map<void *, std::pair<int, size_t>> live;   // ptr -> (device, class size)
    std::map<int, std::vector<Pending>> pending;               // freed behind an event that has not fired yet
    std::map<int, std::vector<hipEvent_t>> spare_events;
    std::map<int, std::vector<hipStream_t>> spare_streams;     // the library's non-blocking streams, idle
    size_t cached_bytes = 0;
    size_t pending_bytes = 0;
    size_t limit = (size_t)16 << 30;
    bool limit_read = false;
};

Pool &pool()
{
    static Pool *p = new Pool();   // leaked on purpose: no destructor order games at exit
    return *p;
}

size_t size_class(size_t bytes)
{
    if (bytes <= 4096) return 4096;
    if (bytes <= (64u << 10)) {
        size_t c = 4096;
        while (c < bytes) c <<= 1;
        return c;
    }
    // 1/8-octave steps
    int top = 63 - __builtin_clzll((unsigned long long)bytes);
    const size_t step = (size_t)1 << (top - 3);
    return (bytes + step - 1) & ~(step - 1);
}

void read_limit(Pool &p)
{
    if (p.limit_read) return;
    p.limit_read = true;
    if (const char *e = getenv("NNS_POOL_BYTES")) p.limit = (size_t)strtoull(e, nullptr, 10);
}

// hand a block to the runtime (hipFree waits for the device by itself)
void release_block(int dev, void *ptr)
{
    int cur = 0;
    const bool sw = dev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != dev;
    if (sw) (void)hipSetDevice(dev);
    (void)hipFree(ptr);
    if (sw) (void)hipSetDevice(cur);
}

// (lock held) move the pending blocks of `dev` whose event has fired — all of them after waiting, with `wait` —
// to the free lists; blocks the pool has no room for are collected in `drop` and freed by the caller outside the lock
void reap(Pool &p, int dev, bool wait, std::vector<void *> *drop)
{
    auto it = p.pending.find(dev);
    if (it == p.pending.end()) return;
    std::vector<Pending> &v = it->second;
    size_t keep = 0;
    for (size_t i = 0; i < v.size(); ++i) {
        bool fired = hipEventQuery(v[i].ev) == hipSuccess;
        if (!fired && wait) fired = hipEventSynchronize(v[i].ev) == hipSuccess;
        if (!fired) {
            (void)hipGetLastError();   // hipErrorNotReady is not an error here
            if (keep != i) v[keep] = std::move(v[i]);
            ++keep;
            continue;
        }
        p.spare_events[dev].push_back(v[i].ev);
        for (auto &b : v[i].blocks) {
            p.pending_bytes -= b.second;
            if (p.cached_bytes + b.second <= p.limit) {
                p.free_blocks[dev].emplace(b.second, b.first);
                p.cached_bytes += b.second;
            } else {
                drop->push_back(b.first);
            }
        }
    }
    v.resize(keep);
}

}  // namespace

hipError_t pool_alloc(void **out, size_t bytes)
{
    *out = nullptr;
    if (bytes == 0) bytes = 1;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const size_t cls = size_class(bytes);
    Pool &p = pool();
    std::vector<void *> drop;
    {
        std::lock_guard<std::mutex> lk(p.mu);
        read_limit(p);
        reap(p, dev, false, &drop);
        auto &fb = p.free_blocks[dev];
        auto it = fb.lower_bound(cls);
        if (it != fb.end() && it->first <= 2 * cls) {
            void *ptr = it->second;
            const size_t got = it->first;
            fb.erase(it);
            p.cached_bytes -= got;
            p.live[ptr] = {dev, got};
            *out = ptr;
        }
    }
    for (void *d : drop) release_block(dev, d);
    if (*out) return hipSuccess;
    void *ptr = nullptr;
    e = hipMalloc(&ptr, cls);
    if (e != hipSuccess) {
        // out of memory with blocks parked or pending: wait for the pending ones, give everything back, retry once
        (void)hipGetLastError();
        pool_trim();
        e = hipMalloc(&ptr, cls);
        if (e != hipSuccess) return e;
    }
    {
        std::lock_guard<std::mutex> lk(p.mu);
        p.live[ptr] = {dev, cls};
    }
    *out = ptr;
    return hipSuccess;
}

// The caller has waited for the work that used the block: it is reusable at once.
void pool_free(void *ptr)
{
    if (!ptr) return;
    Pool &p = pool();
    int dev = -1;
    size_t cls = 0;
    bool park = false;
    {
        std::lock_guard<std::mutex> lk(p.mu);
        auto it = p.live.find(ptr);
        if (it == p.live.end()) {
            // not ours (should not happen): hand it to the runtime
            park = false;
        } else {
            dev = it->second.first;
            cls = it->second.second;
            p.live.erase(it);
            if (p.cached_bytes + cls <= p.limit) {
                p.free_blocks[dev].emplace(cls, ptr);
                p.cached_bytes += cls;
                park = true;
            }
        }
    }
    if (!park) release_block(dev, ptr);
}

// The blocks may still be read or written by work already enqueued on `st` (a stream of the blocks' device, which
// is the calling thread's current device): reusable once that work has completed — ONE event for all of them.
// Never waits on the host unless the event machinery itself fails (then: hipStreamSynchronize(st), or the device as
// the last resort).  Null pointers are skipped.
void pool_free_after(void *const *ptrs, int count, hipStream_t st)
{
    Pool &p = pool();
    Pending pe;
    pe.ev = nullptr;
    int dev = -1;
    std::vector<void *> foreign;
    {
        std::lock_guard<std::mutex> lk(p.mu);
        for (int i = 0; i < count; ++i) {
            if (!ptrs[i]) continue;
            auto it = p.live.find(ptrs[i]);
            if (it == p.live.end()) {   // not ours (should not happen)
                foreign.push_back(ptrs[i]);
                continue;
            }
            dev = it->second.first;
            pe.blocks.emplace_back(ptrs[i], it->second.second);
        }
        if (dev >= 0) {
            auto &sp = p.spare_events[dev];
            if (!sp.empty()) {
                pe.ev = sp.back();
                sp.pop_back();
            }
        }
    }
    if (pe.blocks.empty() && foreign.empty()) return;
    if (!pe.ev && hipEventCreateWithFlags(&pe.ev, hipEventDisableTiming) != hipSuccess) pe.ev = nullptr;
    if (!foreign.empty() || !pe.ev || hipEventRecord(pe.ev, st) != hipSuccess) {
        // no event, or `st` is no longer a stream (destroyed by its owner before the index): wait the blunt way
        (void)hipGetLastError();
        if (hipStreamSynchronize(st) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipDeviceSynchronize();
        }
        if (pe.ev) {
            std::lock_guard<std::mutex> lk(p.mu);
            p.spare_events[dev].push_back(pe.ev);
        }
        for (auto &b : pe.blocks) pool_free(b.first);
        for (void *f : foreign) release_block(-1, f);
        return;
    }
    std::lock_guard<std::mutex> lk(p.mu);
    for (auto &b : pe.blocks) {
        p.live.erase(b.first);
        p.pending_bytes += b.second;
    }
    p.pending[dev].push_back(std::move(pe));
}

void pool_free_after(void *ptr, hipStream_t st)
{
    if (ptr) pool_free_after(&ptr, 1, st);
}

size_t pool_trim()
{
    Pool &p = pool();
    std::map<int, std::multimap<size_t, void *>> take;
    std::map<int, std::vector<void *>> drop;
    size_t bytes = 0;
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    {
        std::lock_guard<std::mutex> lk(p.mu);
        // pending blocks first: wait for their events (stream-scoped waits), then they are parked like the rest
        std::vector<int> devs;
        for (auto &d : p.pending)
            if (!d.second.empty()) devs.push_back(d.first);
        for (int d : devs) {
            (void)hipSetDevice(d);
            reap(p, d, true, &drop[d]);
        }
        take.swap(p.free_blocks);
        bytes = p.cached_bytes;
        p.cached_bytes = 0;
    }
    for (auto &d : take) {
        if (d.second.empty()) continue;
        (void)hipSetDevice(d.first);
        for (auto &b : d.second) (void)hipFree(b.second);
    }
    for (auto &d : drop)
        for (void *b : d.second) release_block(d.first, b);
    if (have) (void)hipSetDevice(cur);
    return bytes;
}

// ---- the library's own streams --------------------------------------------------------------------------------
// Whole-call entry points enqueue their kernels on a NON-BLOCKING stream of the library instead of the legacy
// default stream (which serialises with every blocking stream of the application); a few are kept per device.
hipStream_t lib_stream_acquire()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    Pool &p = pool();
    {
        std::lock_guard<std::mutex> lk(p.mu);
        auto &sp = p.spare_streams[dev];
        if (!sp.empty()) {
            hipStream_t s = sp.back();
            sp.pop_back();
            return s;
        }
    }
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;   // (callers fall back to the default stream)
    }
    return s;
}

void lib_stream_release(hipStream_t s)
{
    if (!s) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return;
    Pool &p = pool();
    std::lock_guard<std::mutex> lk(p.mu);
    auto &sp = p.spare_streams[dev];
    if (sp.size() < 8) sp.push_back(s);
    else (void)hipStreamDestroy(s);
}

}  // namespace nns

extern "C" size_t nns_trim(void) { return nns::pool_trim(); }
