// dev_pool.hip — a small caching allocator for the library's device workspaces.
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
#include <stdlib.h>
#include <map>
#include <mutex>
#include <unordered_map>

#include "nns_internal.h"

namespace nns {

namespace {

struct Pool {
    std::mutex mu;
    // device -> (class size -> blocks)
    std::map<int, std::multimap<size_t, void *>> free_blocks;
    std::unordered_map<void *, std::pair<int, size_t>> live;   // ptr -> (device, class size)
    size_t cached_bytes = 0;
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
    {
        std::lock_guard<std::mutex> lk(p.mu);
        read_limit(p);
        auto &fb = p.free_blocks[dev];
        auto it = fb.lower_bound(cls);
        if (it != fb.end() && it->first <= 2 * cls) {
            void *ptr = it->second;
            const size_t got = it->first;
            fb.erase(it);
            p.cached_bytes -= got;
            p.live[ptr] = {dev, got};
            *out = ptr;
            return hipSuccess;
        }
    }
    void *ptr = nullptr;
    e = hipMalloc(&ptr, cls);
    if (e != hipSuccess) {
        // out of memory with blocks parked: give them back and retry once
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

// Callers free only after the stream that used the block has been synchronised (every entry
// point that frees does), so a parked block has no work in flight.
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
    if (!park) {
        int cur = 0;
        const bool sw = dev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != dev;
        if (sw) (void)hipSetDevice(dev);
        (void)hipFree(ptr);
        if (sw) (void)hipSetDevice(cur);
    }
}

size_t pool_trim()
{
    Pool &p = pool();
    std::map<int, std::multimap<size_t, void *>> take;
    size_t bytes = 0;
    {
        std::lock_guard<std::mutex> lk(p.mu);
        take.swap(p.free_blocks);
        bytes = p.cached_bytes;
        p.cached_bytes = 0;
    }
    int cur = 0;
    const bool have = hipGetDevice(&cur) == hipSuccess;
    for (auto &d : take) {
        if (d.second.empty()) continue;
        (void)hipSetDevice(d.first);
        for (auto &b : d.second) (void)hipFree(b.second);
    }
    if (have) (void)hipSetDevice(cur);
    return bytes;
}

}  // namespace nns

extern "C" size_t nns_trim(void) { return nns::pool_trim(); }
