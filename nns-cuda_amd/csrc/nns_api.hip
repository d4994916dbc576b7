// nns_api.hip — the C ABI of include/nns.h: host orchestration of K1..K5.
//
// Mirrors the host half of the reference's vN::cudaCall (alloc -> H2D -> layout
// prep -> kernels -> D2H -> free; core.cu:123-151 for V1, 634-697 for V7,
// 761-853 for V8), split so that the device-resident part (index create +
// search on the caller's stream) can be timed and sharded on its own.
#include <stdarg.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <mutex>
#include <vector>

#include "nns_internal.h"

namespace nns {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace nns

using namespace nns;

// one point of a tile image
static size_t img_row_bytes(const FilterGeom &g) { return (size_t)g.kt * (g.bf16 ? 2 : 4); }
// The filter's ring DMA runs up to three slots (32 KiB of image, slot_pts norms each) past the last one without a
// bounds branch: images and norm arrays carry that much padding (+ the over-read of a 128-norm piece)
static size_t img_bytes(const FilterGeom &g, int pts_pad) { return (size_t)pts_pad * img_row_bytes(g) + 3 * 32768 + 4096; }
static size_t norm_bytes(const FilterGeom &g, int pts_pad) { return ((size_t)pts_pad + 3 * g.slot_pts + 256) * sizeof(float); }
// below this many queries the AUTO path skips the MFMA filter (and, in the whole-call
// entry points, its ref pre-pass too)
static const int kTinyM = 64;
// most points a set may hold (nns.h NNS_MAX_POINTS): a margin of 2^20 below 2^31 for padded images and range ends
static const int kMaxPoints = NNS_MAX_POINTS;
// internal create flag (never set by callers: masked off at the C ABI): build the index without the
// synchronising read-back of K2's max-|value| word; K5's device-side check then covers NaN / INF / huge refs
static const unsigned kCreateNoSync = 1u << 30;
// ... and a small problem in a dimensionality the lane-per-query exact kernel is instantiated for (8, 16)
// is faster there than through the filter's fixed costs (K2 on both clouds, filter ramp, K5: ~35 us in round 3, ~55 in
// round 2): the reference driver's 16-D 1024 x 1024 sample (main.cu:44) takes 9 us instead of 40.  Crossovers re-measured
// with round 3's record forms and K5 (tools/probe_crossover.py, profiles/r03_crossover.txt): 16-D 2^25 pairs 41 us either
// way, 2^26 50 (filter) vs 66 us — unless the refs are few (65536 x 1024: 110 vs 62 us: a stream of 1024 refs is all ramp);
// 8-D 2^26 pairs 35 (exact) vs 48 us
static bool small_exact(int k, int64_t m, int64_t n)
{
    if (m < kTinyM) return false;
    if (k == 8) return m * n <= ((int64_t)1 << 27);
    if (k == 16) return m * n <= ((int64_t)1 << 25) || (n < 8192 && m * n <= ((int64_t)1 << 26));
    return false;
}
// deepest dimensionality the MFMA filter tiles (bf16 operands; fp32 operands: 256)
static const int kMaxFilterK = 1024;

enum { EV_BEGIN = 0, EV_QPREP, EV_FILTER, EV_FINAL, EV_RERANK, EV_END, EV_R0, EV_R1, EV_COUNT };
static const int kEvRing = 32;

struct nns_index {
    int device = 0;
    int k = 0, n = 0;
    int64_t base = 0;
    unsigned flags = 0;
    int path = NNS_PATH_EXACT;
    const void *r_dev = nullptr;   // fp32 [n][k] or bf16 bits [n][k]
    const void *r_soa = nullptr;   // NNS_REFS_SOA: the caller's [k][n] array; r_dev is then r_own
    void *r_own = nullptr;         // owned point-major copy of r_soa
    int bf16 = 0;
    bool profile = false;
    bool refs_bad = false;
    bool mixed = false;            // NNS_FILTER_BF16: fp32 points, bf16 filter operands

    // MFMA path, ref side
    FilterGeom geom{};
    void *rimg = nullptr;
    float *rnorm = nullptr, *mean = nullptr;
    double *mean_ws = nullptr;
    DevScalars *scal = nullptr;

    // MFMA path, query side (grown on demand)
    int m_cap = 0;
    void *qimg = nullptr;
    float *qnorm = nullptr;
    CandEntry *lists = nullptr;   // [splits][m_pad/32][kCandCap][64]
    int *counts = nullptr;        // [splits][m_pad/32][64]
    size_t lists_cap = 0;         // in lane-lists
    int *amb_list = nullptr;
    int *multi_list = nullptr;    // queries K5 decided among several candidates (nns_index_near_ties)

    // exact path: per-split partial keys of K1a
    nns_key *exact_ws = nullptr;
    size_t exact_ws_keys = 0;
    int exact_ws_m = 0;
    bool exact_ws_fresh = false;   // (re)allocated since the last K1a launch: its arrival counters need zeroing once

    // NNS_PROFILE: a ring of event sets, one per search (refresh + search = one step), so that a
    // caller can time many steps back to back and read the averages once, without a device
    // synchronisation inside every step
    hipEvent_t evr[kEvRing][EV_COUNT] = {};
    bool ev_refreshed[kEvRing] = {};   // set holds a K2-on-refs interval
    int ev_path[kEvRing] = {};         // path of the set's search
    int ev_slot = 0;                   // set the next refresh / search records into
    int ev_count = 0;                  // searches recorded since the last nns_index_stats
    bool ev_valid = false;
    bool searched = false;
    int last_m = 0;
    int last_path = NNS_PATH_EXACT;
    // the stream the index last enqueued work on: destroy / workspace regrow / read-outs wait for (or free behind)
    // THAT stream, never for the whole device
    hipStream_t last_stream = nullptr;
};

static int ensure_device_ok(int device)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess || cnt < 1) {
        set_error("no HIP device visible (the HIP path has no CPU fallback)");
        return NNS_ERR_NODEVICE;
    }
    if (device < 0 || device >= cnt) {
        set_error("device %d out of range (%d visible)", device, cnt);
        return NNS_ERR_INVALID;
    }
    NNS_HIP(hipSetDevice(device));
    return NNS_OK;
}

extern "C++" {
namespace nns {
int order_after_default_stream(hipStream_t st)
{
    if (!st) return NNS_OK;   // same stream: already ordered
    int dev = 0;
    NNS_HIP(hipGetDevice(&dev));
    static thread_local hipEvent_t evs[64] = {};   // one per device and host thread, never destroyed
    if (dev < 0 || dev >= 64) return NNS_OK;
    if (!evs[dev]) NNS_HIP(hipEventCreateWithFlags(&evs[dev], hipEventDisableTiming));
    NNS_HIP(hipEventRecord(evs[dev], nullptr));
    NNS_HIP(hipStreamWaitEvent(st, evs[dev], 0));
    return NNS_OK;
}
}  // namespace nns
}  // extern "C++"

static int prep_refs(nns_index *ix, hipStream_t st)
{
    const FilterGeom &g = ix->geom;
    NNS_HIP(hipMemsetAsync(ix->scal, 0, sizeof(DevScalars), st));
    if (ix->bf16) {
        NNS_TRY(launch_prep_image_bf16(g.lpq == 4 ? 1 : 0, g.kt, ix->k, ix->n, g.n_pad, (const uint16_t *)ix->r_dev, -2.0f, INFINITY,
                                       ix->rimg, ix->rnorm, &ix->scal->ymax2_bits,
                                       &ix->scal->r_maxabs_bits, st));
        return NNS_OK;
    }
    NNS_TRY(launch_prep_mean(ix->k, g.kt, ix->n, (const float *)ix->r_dev, ix->mean_ws, ix->mean,
                             &ix->scal->r_maxabs_bits, st));
    NNS_TRY(launch_prep_image(ix->k, g.kt, ix->n, g.n_pad, (const float *)ix->r_dev, ix->mean, -2.0f,
                              INFINITY, (float *)ix->rimg, ix->rnorm, &ix->scal->ymax2_bits, nullptr, st,
                              ix->mixed));
    return NNS_OK;
}

extern "C" {

int nns_version(void) { return NNS_VERSION_MAJOR * 1000 + NNS_VERSION_MINOR; }

const char *nns_last_error(void) { return g_err; }

const char *nns_strerror(int status)
{
    switch (status) {
    case NNS_OK: return "ok";
    case NNS_ERR_INVALID: return "invalid argument";
    case NNS_ERR_HIP: return "HIP runtime error";
    case NNS_ERR_NOMEM: return "out of memory";
    case NNS_ERR_NODEVICE: return "no gfx950 device";
    case NNS_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown status";
    }
}

int nns_device_count(void)
{
    int cnt = 0;
    if (hipGetDeviceCount(&cnt) != hipSuccess) return 0;
    return cnt;
}

// stream_idle: the caller has just waited for the stream the index worked on (the whole-call paths): the workspaces are
// reusable at once, no event needed
static int index_destroy_impl(nns_index *ix, bool stream_idle);

int nns_index_destroy(nns_index *ix) { return index_destroy_impl(ix, false); }

static int index_destroy_impl(nns_index *ix, bool stream_idle)
{
    if (!ix) return NNS_OK;
    DeviceScope keep_device;
    (void)hipSetDevice(ix->device);
    // Workspaces go back to the pool behind ONE event on the stream the index last worked on: they become
    // reusable when that work has completed.  No host wait — an application's kernels on other streams (or this
    // index's own, still running) are not waited for.  (Round 2: hipDeviceSynchronize() here.)
    void *const blocks[] = {ix->r_own, ix->rimg, ix->rnorm, ix->mean, ix->mean_ws, ix->scal, ix->qimg, ix->qnorm,
                            ix->lists, ix->counts, ix->amb_list, ix->multi_list, ix->exact_ws};
    if (stream_idle)
        for (void *b : blocks) pool_free(b);
    else
        pool_free_after(blocks, (int)(sizeof(blocks) / sizeof(blocks[0])), ix->last_stream);
    if (ix->ev_valid)
        for (int r = 0; r < kEvRing; ++r)
            for (int i = 0; i < EV_COUNT; ++i) (void)hipEventDestroy(ix->evr[r][i]);
    delete ix;
    return NNS_OK;
}

static int index_create_impl(nns_index **out, int device, int k, int n, const void *r_dev, int bf16,
                             int64_t index_base, unsigned flags, void *stream)
{
    if (!out || !r_dev || k <= 0 || n <= 0) {
        set_error("nns_index_create: k, n must be > 0 and pointers non-null (k=%d n=%d)", k, n);
        return NNS_ERR_INVALID;
    }
    if (n > kMaxPoints) {
        set_error("nns_index_create: n = %d exceeds NNS_MAX_POINTS (%d): padded tile images and range ends must stay below 2^31", n, kMaxPoints);
        return NNS_ERR_INVALID;
    }
    if (index_base < 0 || index_base + (int64_t)n > 0x7FFFFFFFll) {
        set_error("nns_index_create: index_base + n exceeds int32 (the reference's index type)");
        return NNS_ERR_INVALID;
    }
    *out = nullptr;
    NNS_TRY(ensure_device_ok(device));
    hipStream_t st = (hipStream_t)stream;

    nns_index *ix = new (std::nothrow) nns_index();
    if (!ix) return NNS_ERR_NOMEM;
    ix->device = device;
    ix->k = k;
    ix->n = n;
    ix->base = index_base;
    ix->flags = flags;
    ix->r_dev = r_dev;
    ix->bf16 = bf16;
    ix->profile = (flags & NNS_PROFILE) != 0;
    ix->last_stream = st;
    if (flags & NNS_FILTER_BF16) {
        if (bf16) {
            set_error("NNS_FILTER_BF16 applies to fp32 points (bf16 points already use the bf16 filter)");
            delete ix;   // nothing allocated or enqueued yet
            return NNS_ERR_INVALID;
        }
        ix->mixed = true;
    }

    // fp32 points beyond the deepest fp32 tile (256): AUTO takes the bf16-operand filter — the re-rank makes
    // the result bits identical, and the alternative is the VALU scan
    if ((flags & NNS_PATH_MASK) == NNS_PATH_AUTO && !bf16 && k > 256 && k <= kMaxFilterK) ix->mixed = true;
    const int kmax = (bf16 || ix->mixed) ? kMaxFilterK : 256;   // deepest tile of the MFMA filter
    int path = flags & NNS_PATH_MASK;
    // crossover: from k = 8 the MFMA filter (KT = 16 / 32 tile) beats 3k VALU ops per pair; bf16 tiles
    // are at least 128 deep, so they only pay from k = 32
    const int kmin = bf16 ? 32 : 8;
    if (path == NNS_PATH_AUTO) path = (k >= (ix->mixed ? 32 : kmin) && k <= kmax) ? NNS_PATH_MFMA : NNS_PATH_EXACT;
    if (path == NNS_PATH_MFMA && k > kmax) {
        set_error("NNS_PATH_MFMA: k = %d > %d is not tiled (use NNS_PATH_AUTO/EXACT)", k, kmax);
        delete ix;   // nothing allocated or enqueued yet
        return NNS_ERR_UNSUPPORTED;
    }
    ix->path = path;
    if (flags & NNS_REFS_SOA) {
        // dimension-major refs: one transpose into a point-major copy the index owns
        const size_t esz = bf16 ? sizeof(uint16_t) : sizeof(float);
        ix->r_soa = r_dev;
        if (pool_alloc(&ix->r_own, (size_t)n * k * esz) != hipSuccess) {
            set_error("nns_index_create: device allocation failed (point-major copy of %d x %d refs)", n, k);
            delete ix;
            return NNS_ERR_NOMEM;
        }
        const int trc = launch_soa_to_aos(k, n, r_dev, ix->r_own, (int)esz, st);
        if (trc != NNS_OK) {
            nns_index_destroy(ix);
            return trc;
        }
        ix->r_dev = ix->r_own;
    }

    int rc = NNS_OK;
    do {
        if (ix->profile) {
            bool ok = true;
            for (int r = 0; r < kEvRing; ++r)
                for (int i = 0; i < EV_COUNT; ++i) ok = ok && hipEventCreate(&ix->evr[r][i]) == hipSuccess;
            if (!ok) {
                set_error("hipEventCreate failed");
                rc = NNS_ERR_HIP;
                break;
            }
            ix->ev_valid = true;
        }
        if (path == NNS_PATH_MFMA) {
            if ((rc = filter_plan(k, 1, n, bf16 != 0, &ix->geom, ix->mixed, (flags & NNS_RECORDS_PER_REF) != 0)) != NNS_OK) break;
            const FilterGeom &g = ix->geom;
            size_t ws = 0;
            prep_workspace_bytes(g.kt, &ws);
            if (pool_alloc(&ix->rimg, img_bytes(g, g.n_pad)) != hipSuccess ||
                pool_alloc(&ix->rnorm, norm_bytes(g, g.n_pad)) != hipSuccess ||
                pool_alloc(&ix->mean, (size_t)g.kt * sizeof(float)) != hipSuccess ||
                pool_alloc(&ix->mean_ws, ws) != hipSuccess ||
                pool_alloc(&ix->scal, sizeof(DevScalars)) != hipSuccess) {
                set_error("nns_index_create: device allocation failed (n_pad=%d kt=%d)", g.n_pad, g.kt);
                rc = NNS_ERR_NOMEM;
                break;
            }
            if (ix->profile) (void)hipEventRecord(ix->evr[ix->ev_slot][EV_R0], st);
            if ((rc = prep_refs(ix, st)) != NNS_OK) break;
            if (ix->profile) {
                (void)hipEventRecord(ix->evr[ix->ev_slot][EV_R1], st);
                ix->ev_refreshed[ix->ev_slot] = true;
            }
            if (flags & kCreateNoSync) break;   // (pipelined whole call: stay asynchronous, K5 checks on the device)
            // index build is synchronous: learn whether the refs void the error bound
            DevScalars h{};
            if (hipMemcpyAsync(&h, ix->scal, sizeof(h), hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) {
                set_error("nns_index_create: readback failed: %s", hipGetErrorString(hipGetLastError()));
                rc = NNS_ERR_HIP;
                break;
            }
            ix->refs_bad = h.r_maxabs_bits >= 0x5BB1A2BCu;   // NaN / INF / |v| >= 1e17
        }
    } while (0);
    if (rc != NNS_OK) {
        nns_index_destroy(ix);
        return rc;
    }
    *out = ix;
    return NNS_OK;
}

int nns_index_create(nns_index **out, int device, int k, int n, const float *r_dev, int64_t index_base,
                     unsigned flags, void *stream)
{
    DeviceScope keep_device;
    return index_create_impl(out, device, k, n, r_dev, 0, index_base, flags & ~kCreateNoSync, stream);
}

int nns_index_create_bf16(nns_index **out, int device, int k, int n, const uint16_t *r_dev,
                          int64_t index_base, unsigned flags, void *stream)
{
    DeviceScope keep_device;
    return index_create_impl(out, device, k, n, r_dev, 1, index_base, flags & ~kCreateNoSync, stream);
}

int nns_index_refresh(nns_index *ix, void *stream)
{
    if (!ix) return NNS_ERR_INVALID;
    DeviceScope keep_device;
    NNS_TRY(ensure_device_ok(ix->device));
    hipStream_t st = (hipStream_t)stream;
    ix->last_stream = st;
    if (ix->r_soa)   // the caller's dimension-major array may have changed
        NNS_TRY(launch_soa_to_aos(ix->k, ix->n, ix->r_soa, ix->r_own, ix->bf16 ? 2 : 4, st));
    if (ix->path != NNS_PATH_MFMA) return NNS_OK;
    if (ix->profile) (void)hipEventRecord(ix->evr[ix->ev_slot][EV_R0], st);
    NNS_TRY(prep_refs(ix, st));
    // The new values may or may not void the error bound (NaN / INF / huge); refresh stays asynchronous,
    // so the next search runs the filter and K5 decides on the device (finalize.hip re-checks K2's max-|v|
    // word and sends every query to the exact scan if it must).  nns_index_stats() re-latches the flag.
    ix->refs_bad = false;
    if (ix->profile) {
        (void)hipEventRecord(ix->evr[ix->ev_slot][EV_R1], st);
        ix->ev_refreshed[ix->ev_slot] = true;
    }
    return NNS_OK;
}

// (a block that is replaced goes back to the pool behind an event on `st`, the stream of the search that is about
//  to be enqueued: the index's earlier searches were ordered before it by the caller — searches of one index do
//  not overlap — so once that event has fired nothing reads the old block any more.  No host wait.)
static int ensure_query_ws(nns_index *ix, int m, hipStream_t st)
{
    FilterGeom g = ix->geom;
    FilterGeom gq{};
    NNS_TRY(filter_plan(ix->k, m, ix->n, ix->bf16 != 0, &gq, ix->mixed, (ix->flags & NNS_RECORDS_PER_REF) != 0));
    ix->geom = gq;   // same kt / n_pad / total_slots; m-dependent grid now filled in
    (void)g;
    if (gq.m_pad > ix->m_cap) {
        void *const old[] = {ix->qimg, ix->qnorm, ix->amb_list, ix->multi_list};
        pool_free_after(old, 4, st);   // an earlier search may still read them
        ix->qimg = nullptr;
        ix->qnorm = nullptr;
        ix->amb_list = nullptr;
        ix->multi_list = nullptr;
        ix->m_cap = 0;
        if (pool_alloc(&ix->qimg, img_bytes(gq, gq.m_pad)) != hipSuccess ||
            pool_alloc(&ix->qnorm, (size_t)gq.m_pad * sizeof(float)) != hipSuccess ||
            pool_alloc(&ix->amb_list, (size_t)gq.m_pad * sizeof(int)) != hipSuccess ||
            pool_alloc(&ix->multi_list, (size_t)gq.m_pad * sizeof(int)) != hipSuccess) {
            set_error("query workspace allocation failed (m_pad=%d)", gq.m_pad);
            return NNS_ERR_NOMEM;
        }
        ix->m_cap = gq.m_pad;
    }
    const size_t need = (size_t)gq.splits * gq.m_pad * gq.lpq;   // lane-lists
    if (need > ix->lists_cap) {
        void *const old[] = {ix->lists, ix->counts};
        pool_free_after(old, 2, st);
        ix->lists = nullptr;
        ix->counts = nullptr;
        ix->lists_cap = 0;
        if (pool_alloc(&ix->lists, need * kCandCap * sizeof(CandEntry)) != hipSuccess ||
            pool_alloc(&ix->counts, need * sizeof(int)) != hipSuccess) {
            set_error("candidate list allocation failed (%zu lists)", need);
            return NNS_ERR_NOMEM;
        }
        ix->lists_cap = need;
    }
    return NNS_OK;
}

// a search has recorded its events into the current set: close it and move on to the next one
static void profile_advance(nns_index *ix, int path)
{
    if (!ix->profile) return;
    ix->ev_path[ix->ev_slot] = path;
    ix->ev_slot = (ix->ev_slot + 1) % kEvRing;
    ix->ev_refreshed[ix->ev_slot] = false;
    if (ix->ev_count < kEvRing) ++ix->ev_count;
}

// idx_dev / dist_dev (optional): also leave the unpacked indices / distances (K1a writes them in its one
// launch; every other path appends the unpack kernel)
static int index_search_impl(nns_index *ix, int m, const void *q_dev, int bf16, nns_key *keys_dev, void *stream,
                             int *idx_dev = nullptr, float *dist_dev = nullptr)
{
    bool unpacked = false;
    if (ix && ix->bf16 != bf16) {
        set_error("nns_index_search: query dtype does not match the index (%s index)", ix->bf16 ? "bf16" : "fp32");
        return NNS_ERR_INVALID;
    }
    if (!ix || !q_dev || !keys_dev || m <= 0) {
        set_error("nns_index_search: m must be > 0 and pointers non-null (m=%d)", m);
        return NNS_ERR_INVALID;
    }
    if (m > kMaxPoints) {
        set_error("nns_index_search: m = %d exceeds NNS_MAX_POINTS (%d)", m, kMaxPoints);
        return NNS_ERR_INVALID;
    }
    NNS_TRY(ensure_device_ok(ix->device));
    hipStream_t st = (hipStream_t)stream;
    const bool prof = ix->profile;
    ix->last_m = m;
    ix->last_stream = st;

    // A handful of queries cannot fill MFMA tiles (they are padded to 256): the ref stream
    // is then HBM-bound and the exact lane-per-ref kernel is the faster path (AUTO only).
    const bool tiny = (ix->flags & NNS_PATH_MASK) == NNS_PATH_AUTO && (m < kTinyM || (!bf16 && small_exact(ix->k, m, ix->n)));
    if (ix->path != NNS_PATH_MFMA || ix->refs_bad || tiny) {
        if (prof) (void)hipEventRecord(ix->evr[ix->ev_slot][EV_BEGIN], st);
        if (bf16)
            NNS_TRY(launch_exact_search_bf16(ix->k, m, ix->n, (const uint16_t *)q_dev, (const uint16_t *)ix->r_dev,
                                             ix->base, keys_dev, st));
        else
        {
            const size_t need = exact_workspace_keys(ix->k, m, ix->n);
            if (need > ix->exact_ws_keys) {
                pool_free_after(ix->exact_ws, st);
                ix->exact_ws = nullptr;
                ix->exact_ws_keys = 0;
                if (pool_alloc(&ix->exact_ws, need * sizeof(nns_key)) == hipSuccess) {
                    ix->exact_ws_keys = need;
                    ix->exact_ws_fresh = true;
                } else {
                    (void)hipGetLastError();   // no workspace: K1a runs one ref range per query tile
                }
            }
            // (the workspace layout depends on m: a different m re-lays the counters out -> zero them again)
            if (m != ix->exact_ws_m) ix->exact_ws_fresh = true;
            ix->exact_ws_m = m;
            NNS_TRY(launch_exact_search(ix->k, m, ix->n, (const float *)q_dev, (const float *)ix->r_dev, ix->base,
                                        keys_dev, ix->exact_ws, ix->exact_ws_keys, ix->exact_ws_fresh, idx_dev, dist_dev, st));
            ix->exact_ws_fresh = false;
            unpacked = idx_dev != nullptr;
        }
        if (idx_dev && !unpacked) NNS_TRY(launch_keys_unpack(keys_dev, m, idx_dev, dist_dev, st));
        if (prof) (void)hipEventRecord(ix->evr[ix->ev_slot][EV_END], st);
        ix->last_path = NNS_PATH_EXACT;
        ix->searched = true;
        profile_advance(ix, NNS_PATH_EXACT);
        return NNS_OK;
    }

    NNS_TRY(ensure_query_ws(ix, m, st));
    const FilterGeom &g = ix->geom;
    if (prof) (void)hipEventRecord(ix->evr[ix->ev_slot][EV_BEGIN], st);
    // reset the per-search scalars (q max-abs, ambiguous count); keep the ref-side ones
    static_assert(offsetof(DevScalars, amb_count) == offsetof(DevScalars, q_maxabs_bits) + sizeof(unsigned) &&
                      offsetof(DevScalars, multi_count) == offsetof(DevScalars, amb_count) + sizeof(int),
                  "per-search scalars must be adjacent");
    NNS_HIP(hipMemsetAsync(&ix->scal->q_maxabs_bits, 0, sizeof(unsigned) + 2 * sizeof(int), st));
    if (bf16)
        NNS_TRY(launch_prep_image_bf16(g.lpq == 4 ? 1 : 0, g.kt, ix->k, m, g.m_pad, (const uint16_t *)q_dev, 1.0f, 0.0f, ix->qimg, ix->qnorm,
                                       nullptr, &ix->scal->q_maxabs_bits, st));
    else
        NNS_TRY(launch_prep_image(ix->k, g.kt, m, g.m_pad, (const float *)q_dev, ix->mean, 1.0f, 0.0f,
                                  (float *)ix->qimg, ix->qnorm, nullptr, &ix->scal->q_maxabs_bits, st, ix->mixed));
    if (prof) (void)hipEventRecord(ix->evr[ix->ev_slot][EV_QPREP], st);
    NNS_TRY(launch_filter(g, ix->qimg, ix->rimg, ix->rnorm, ix->qnorm, ix->scal, ix->lists, ix->counts, st));
    if (prof) (void)hipEventRecord(ix->evr[ix->ev_slot][EV_FILTER], st);
#ifdef NNS_DIAG
    // tuning diagnostic for ablated filter builds (-DNNS_FILTER_ABLATE: empty candidate lists would
    // send every query to the exact scan): stop after the filter; the keys are NOT results
    static const bool filter_only = getenv("NNS_DIAG_FILTER_ONLY") != nullptr;
    if (filter_only) {
        NNS_TRY(launch_keys_fill(keys_dev, m, NNS_KEY_NONE, st));
        if (prof) {
            (void)hipEventRecord(ix->evr[ix->ev_slot][EV_FINAL], st);
            (void)hipEventRecord(ix->evr[ix->ev_slot][EV_RERANK], st);
            (void)hipEventRecord(ix->evr[ix->ev_slot][EV_END], st);
        }
        ix->last_path = NNS_PATH_MFMA;
        ix->searched = true;
        profile_advance(ix, NNS_PATH_MFMA);
        return NNS_OK;
    }
#endif
    NNS_TRY(launch_finalize(g, ix->k, m, ix->n, q_dev, ix->r_dev, ix->lists, ix->counts, ix->qnorm,
                            ix->scal, ix->base, keys_dev, ix->amb_list, ix->multi_list, st));
    if (prof) (void)hipEventRecord(ix->evr[ix->ev_slot][EV_FINAL], st);
    if (bf16)
        NNS_TRY(launch_exact_listed_bf16(ix->k, ix->n, (const uint16_t *)q_dev, (const uint16_t *)ix->r_dev,
                                         ix->amb_list, &ix->scal->amb_count, m, ix->base, keys_dev, st));
    else
        NNS_TRY(launch_exact_listed(ix->k, ix->n, (const float *)q_dev, (const float *)ix->r_dev, ix->amb_list,
                                    &ix->scal->amb_count, m, ix->base, keys_dev, st));
    if (prof) (void)hipEventRecord(ix->evr[ix->ev_slot][EV_RERANK], st);
    if (idx_dev) NNS_TRY(launch_keys_unpack(keys_dev, m, idx_dev, dist_dev, st));
    if (prof) (void)hipEventRecord(ix->evr[ix->ev_slot][EV_END], st);
    ix->last_path = NNS_PATH_MFMA;
    ix->searched = true;
    profile_advance(ix, NNS_PATH_MFMA);
    return NNS_OK;
}

int nns_index_search(nns_index *ix, int m, const float *q_dev, nns_key *keys_dev, void *stream)
{
    DeviceScope keep_device;
    return index_search_impl(ix, m, q_dev, 0, keys_dev, stream);
}

int nns_index_search_bf16(nns_index *ix, int m, const uint16_t *q_dev, nns_key *keys_dev, void *stream)
{
    DeviceScope keep_device;
    return index_search_impl(ix, m, q_dev, 1, keys_dev, stream);
}

int nns_index_search_indices(nns_index *ix, int m, const void *q_dev, nns_key *keys_dev, int *idx_dev,
                             float *dist_dev, void *stream)
{
    if (!idx_dev) {
        set_error("nns_index_search_indices: idx_dev is null");
        return NNS_ERR_INVALID;
    }
    DeviceScope keep_device;
    return index_search_impl(ix, m, q_dev, ix ? ix->bf16 : 0, keys_dev, stream, idx_dev, dist_dev);
}

int nns_index_stats(nns_index *ix, nns_stats *out)
{
    if (!ix || !out) return NNS_ERR_INVALID;
    DeviceScope keep_device;
    NNS_TRY(ensure_device_ok(ix->device));
    memset(out, 0, sizeof(*out));
    out->path = ix->searched ? ix->last_path : ix->path;
    out->nonfinite = ix->refs_bad ? 1 : 0;
    if (ix->path == NNS_PATH_MFMA) {
        out->k_tile = ix->geom.kt;
        out->splits = ix->geom.splits;
        // (the index's own stream, not the device: the read-out waits for this index's work only)
        DevScalars h{};
        NNS_HIP(hipMemcpyAsync(&h, ix->scal, sizeof(h), hipMemcpyDeviceToHost, ix->last_stream));
        NNS_HIP(hipStreamSynchronize(ix->last_stream));
        out->ambiguous = ix->searched && ix->last_path == NNS_PATH_MFMA ? h.amb_count : 0;
        out->multi_candidate = ix->searched && ix->last_path == NNS_PATH_MFMA ? h.multi_count : 0;
        ix->refs_bad = h.r_maxabs_bits >= 0x5BB1A2BCu;   // (re-)latch: refs that void the bound go straight to K1
        if (ix->refs_bad || h.q_maxabs_bits >= 0x5BB1A2BCu) out->nonfinite = 1;
    }
    if (ix->profile && ix->ev_valid) {
        NNS_HIP(hipStreamSynchronize(ix->last_stream));
        // averages over the searches recorded since the previous call (the last kEvRing at most)
        const int cnt = ix->ev_count;
        double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // refs, qprep, filter, final, rerank, exact, total
        int n_refs = 0, n_mfma = 0, n_exact = 0;
        auto span = [&](int set, int a, int b, double *dst) {
            float ms = 0;
            if (hipEventElapsedTime(&ms, ix->evr[set][a], ix->evr[set][b]) == hipSuccess) *dst += ms;
        };
        if (cnt == 0) {
            // nothing searched since the last call: the index build / a refresh alone
            if (ix->ev_refreshed[ix->ev_slot]) {
                span(ix->ev_slot, EV_R0, EV_R1, &acc[0]);
                n_refs = 1;
            }
        }
        for (int j = 0; j < cnt; ++j) {
            const int set = (ix->ev_slot - 1 - j + 2 * kEvRing) % kEvRing;
            if (ix->ev_refreshed[set]) {
                span(set, EV_R0, EV_R1, &acc[0]);
                ++n_refs;
            }
            if (ix->ev_path[set] == NNS_PATH_MFMA) {
                span(set, EV_BEGIN, EV_QPREP, &acc[1]);
                span(set, EV_QPREP, EV_FILTER, &acc[2]);
                span(set, EV_FILTER, EV_FINAL, &acc[3]);
                span(set, EV_FINAL, EV_RERANK, &acc[4]);
                ++n_mfma;
            } else {
                span(set, EV_BEGIN, EV_END, &acc[5]);
                ++n_exact;
            }
            span(set, EV_BEGIN, EV_END, &acc[6]);
        }
        if (n_refs) out->prep_refs_ms = (float)(acc[0] / n_refs);
        if (n_mfma) {
            out->prep_queries_ms = (float)(acc[1] / n_mfma);
            out->filter_ms = (float)(acc[2] / n_mfma);
            out->finalize_ms = (float)(acc[3] / n_mfma);
            out->rerank_ms = (float)(acc[4] / n_mfma);
        }
        if (n_exact) out->exact_ms = (float)(acc[5] / n_exact);
        if (cnt) out->total_ms = (float)(acc[6] / cnt);
        ix->ev_count = 0;
        (void)hipGetLastError();
    }
    return NNS_OK;
}

int nns_index_near_ties(nns_index *ix, int *ids_out, int cap, int *count_out)
{
    if (!ix || !count_out || cap < 0 || (cap > 0 && !ids_out)) {
        set_error("nns_index_near_ties: bad arguments");
        return NNS_ERR_INVALID;
    }
    *count_out = 0;
    DeviceScope keep_device;
    NNS_TRY(ensure_device_ok(ix->device));
    if (ix->path != NNS_PATH_MFMA || !ix->searched || ix->last_path != NNS_PATH_MFMA) return NNS_OK;
    DevScalars h{};
    NNS_HIP(hipMemcpyAsync(&h, ix->scal, sizeof(h), hipMemcpyDeviceToHost, ix->last_stream));
    NNS_HIP(hipStreamSynchronize(ix->last_stream));
    *count_out = h.multi_count;
    const int take = h.multi_count < cap ? h.multi_count : cap;
    if (take > 0) {
        NNS_HIP(hipMemcpyAsync(ids_out, ix->multi_list, (size_t)take * sizeof(int), hipMemcpyDeviceToHost, ix->last_stream));
        NNS_HIP(hipStreamSynchronize(ix->last_stream));
    }
    return NNS_OK;
}

int nns_tau_consts(int kt, float qnorm2, float ymax2, int mode, float *out3)
{
    if (kt <= 0 || mode < 0 || mode > 2 || !out3) return NNS_ERR_INVALID;
    const TauConsts t = tau_consts(kt, qnorm2, ymax2, mode);
    out3[0] = t.c0;
    out3[1] = t.c1;
    out3[2] = t.x2;
    return NNS_OK;
}

int nns_keys_min(nns_key *inout_dev, const nns_key *other_dev, int m, void *stream)
{
    if (!inout_dev || !other_dev || m <= 0) return NNS_ERR_INVALID;
    return launch_keys_min(inout_dev, other_dev, m, (hipStream_t)stream);
}

int nns_keys_unpack(const nns_key *keys_dev, int m, int *idx_dev, float *dist_dev, void *stream)
{
    if (!keys_dev || !idx_dev || m <= 0) return NNS_ERR_INVALID;
    return launch_keys_unpack(keys_dev, m, idx_dev, dist_dev, (hipStream_t)stream);
}

int nns_fill_uniform(float *dev, size_t count, uint64_t seed, uint64_t offset, void *stream)
{
    if (!dev && count) return NNS_ERR_INVALID;
    return launch_fill_uniform(dev, count, seed, offset, (hipStream_t)stream);
}

int nns_selftest_mfma(int kt, int bf16, const float *a, const float *b, const float *c0, float *out)
{
    if (kt <= 0 || (kt & 15) || (bf16 == 2 && (kt & 31)) || bf16 < 0 || bf16 > 2 || !a || !b || !c0 || !out)
        return NNS_ERR_INVALID;
    DeviceScope keep_device;
    NNS_TRY(ensure_device_ok(0));
    float *d = nullptr;
    const size_t na = (size_t)32 * kt, total = 2 * na + 32 + 1024;
    NNS_HIP(pool_alloc(&d, total * sizeof(float)));
    int rc = NNS_OK;
    if (hipMemcpy(d, a, na * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + na, b, na * 4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(d + 2 * na, c0, 32 * 4, hipMemcpyHostToDevice) != hipSuccess)
        rc = NNS_ERR_HIP;
    if (rc == NNS_OK) rc = launch_mfma_selftest(kt, bf16, d, d + na, d + 2 * na, d + 2 * na + 32, nullptr);
    if (rc == NNS_OK && hipMemcpy(out, d + 2 * na + 32, 1024 * 4, hipMemcpyDeviceToHost) != hipSuccess)
        rc = NNS_ERR_HIP;
    if (rc == NNS_ERR_HIP) set_error("nns_selftest_mfma: %s", hipGetErrorString(hipGetLastError()));
    pool_free(d);
    return rc;
}

// ---- host-side state of the whole-call entry points ---------------------------------------------------------
namespace {
// small whole calls: one pinned scratch for inputs and outputs (search_host_small)
constexpr size_t kSmallStageBytes = (size_t)2 << 20;
void *g_small_pinned = nullptr;
int g_small_device = -1;
std::mutex g_small_mu;
}  // namespace

extern "C++" {
namespace nns {
// nns_shutdown: give the pinned scratch back
void stager_release()
{
    std::lock_guard<std::mutex> lk2(g_small_mu);
    if (g_small_pinned) {
        (void)hipHostFree(g_small_pinned);
        g_small_pinned = nullptr;
        g_small_device = -1;
    }
}
}  // namespace nns
}  // extern "C++"

int nns_plan_filter(int k, int m, int n, int bf16_points, unsigned flags, int *out, int out_len)
{
    if (!out || out_len < 12 || k <= 0 || m <= 0 || n <= 0) return NNS_ERR_INVALID;
    const bool mixed = !bf16_points && ((flags & NNS_FILTER_BF16) || (k > 256 && (flags & NNS_PATH_MASK) == NNS_PATH_AUTO));
    FilterGeom g{};
    NNS_TRY(filter_plan(k, m, n, bf16_points != 0, &g, mixed, (flags & NNS_RECORDS_PER_REF) != 0));
    const int v[14] = {g.kt, g.bf16, g.mixed, g.lpq, g.m_pad, g.n_pad, g.total_slots, g.splits, g.slots_per_split,
                       g.qgroups, g.slot_pts, g.m_pad / g.qgroups, g.share_thr, g.tile_rec};
    memcpy(out, v, (out_len >= 14 ? 14 : 12) * sizeof(int));
    return NNS_OK;
}

int nns_plan_exact(int k, int m, int n, int refs_aligned, int have_workspace, int *out, int out_len)
{
    if (!out || out_len < 6 || k <= 0 || m <= 0 || n <= 0) return NNS_ERR_INVALID;
    if ((int64_t)m > kMaxPoints || (int64_t)n > kMaxPoints) return NNS_ERR_INVALID;
    int v[6] = {0, 0, 0, 0, 0, 0};
    NNS_TRY(exact_plan(k, m, n, refs_aligned != 0, have_workspace != 0, v));
    memcpy(out, v, sizeof(v));
    return NNS_OK;
}

int nns_selftest_lane_share(int tile16, const float *in64, float *out64)
{
    if (!in64 || !out64) return NNS_ERR_INVALID;
    DeviceScope keep_device;
    NNS_TRY(ensure_device_ok(0));
    float *d = nullptr;
    NNS_HIP(pool_alloc(&d, 128 * sizeof(float)));
    int rc = NNS_OK;
    if (hipMemcpy(d, in64, 64 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) rc = NNS_ERR_HIP;
    if (rc == NNS_OK) rc = launch_lane_share_selftest(tile16 ? 1 : 0, d, d + 64, nullptr);
    if (rc == NNS_OK && hipMemcpy(out64, d + 64, 64 * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) rc = NNS_ERR_HIP;
    if (rc == NNS_ERR_HIP) set_error("nns_selftest_lane_share: %s", hipGetErrorString(hipGetLastError()));
    pool_free(d);
    return rc;
}

// Small whole calls (the reference driver's m = 1 and 1024 x 1024 samples, main.cu:38-51: tens of KiB): what
// main.cu:73-75 times is then the host round trips, not the search.  Inputs go through ONE pinned scratch and
// ONE asynchronous upload into ONE pooled device block, the outputs come back through the same scratch, and the
// host waits once (the plain path: two synchronous pageable uploads, a stream sync, one or two synchronous
// downloads).  NNS_ERR_UNSUPPORTED: not applicable (too big, scratch busy or unavailable) -> the plain path.
static int search_host_small(int k, int m, int n, const void *s_points, const void *r_points, int bf16,
                             int *idx_out, float *dist_out, unsigned flags, int device)
{
    const size_t esz = bf16 ? sizeof(uint16_t) : sizeof(float);
    auto up = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t qb = (size_t)m * k * esz, rb = (size_t)n * k * esz;
    const size_t off_r = up(qb), in_bytes = off_r + rb;
    const size_t off_keys = up(in_bytes), off_idx = off_keys + up((size_t)m * sizeof(nns_key));
    const size_t off_dist = off_idx + up((size_t)m * sizeof(int)), total = off_dist + up((size_t)m * sizeof(float));
    const size_t out_bytes = (off_dist - off_idx) + (size_t)m * sizeof(float);
    if (in_bytes + out_bytes + 512 > kSmallStageBytes || (flags & (NNS_REFS_SOA | NNS_PROFILE))) return NNS_ERR_UNSUPPORTED;
    std::unique_lock<std::mutex> own(g_small_mu, std::try_to_lock);
    if (!own.owns_lock()) return NNS_ERR_UNSUPPORTED;
    if (g_small_pinned && g_small_device != device) {
        (void)hipHostFree(g_small_pinned);
        g_small_pinned = nullptr;
    }
    if (!g_small_pinned) {
        if (hipHostMalloc(&g_small_pinned, kSmallStageBytes, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            g_small_pinned = nullptr;
            return NNS_ERR_UNSUPPORTED;
        }
        g_small_device = device;
    }
    char *host = (char *)g_small_pinned, *host_out = host + up(in_bytes);
    char *blk = nullptr;
    if (pool_alloc(&blk, total) != hipSuccess) {
        set_error("nns_search_f32: device allocation failed");
        return NNS_ERR_NOMEM;
    }
    // a non-blocking stream of the library: the legacy default stream would serialise with every blocking stream
    // of the application (nullptr if none can be made: the default stream then)
    hipStream_t st = lib_stream_acquire();
    nns_index *ix = nullptr;
    int rc = NNS_OK;
    do {
        memcpy(host, s_points, qb);
        memcpy(host + off_r, r_points, rb);
        if (hipMemcpyAsync(blk, host, in_bytes, hipMemcpyHostToDevice, st) != hipSuccess) {
            set_error("nns_search_f32: H2D copy failed: %s", hipGetErrorString(hipGetLastError()));
            rc = NNS_ERR_HIP;
            break;
        }
        // (no synchronising read-back at create: K5 checks NaN / INF / huge refs on the device)
        if ((rc = index_create_impl(&ix, device, k, n, blk + off_r, bf16, 0, flags | kCreateNoSync, st)) != NNS_OK) break;
        if ((rc = index_search_impl(ix, m, blk, bf16, (nns_key *)(blk + off_keys), st, (int *)(blk + off_idx),
                                    (float *)(blk + off_dist))) != NNS_OK)
            break;
        if (hipMemcpyAsync(host_out, blk + off_idx, out_bytes, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) {
            set_error("nns_search_f32: kernel execution or D2H copy failed: %s", hipGetErrorString(hipGetLastError()));
            rc = NNS_ERR_HIP;
            break;
        }
        memcpy(idx_out, host_out, (size_t)m * sizeof(int));
        if (dist_out) memcpy(dist_out, host_out + (off_dist - off_idx), (size_t)m * sizeof(float));
    } while (0);
    // (rc == NNS_OK: the stream was synchronised above and everything is reusable at once; on an error path the
    //  blocks wait behind an event on the stream)
    if (ix) index_destroy_impl(ix, rc == NNS_OK);
    if (rc == NNS_OK) pool_free(blk);
    else pool_free_after(blk, st);
    lib_stream_release(st);
    return rc;
}

// ---- overlapped upload of the whole-call entry points ----------------------------------------------------
// The reference times alloc + H2D + kernels + D2H (main.cu:73-75).  Whole calls whose search is worth hiding an
// upload behind send the refs up in four growing chunks (1/8, 1/4, 5/16, 5/16) with the runtime's own synchronous
// pageable copy, and search each chunk on a NON-BLOCKING stream as soon as its copy has returned: the next chunk's
// copy runs beside that search (on the legacy default stream it does not — round 1 read that as "pageable copies
// do not overlap resident workgroups" and built a ring of pinned buffers filled by host threads; on a non-blocking
// stream the plain copy overlaps just as well and moves 43 GB/s against the ring's 27: C3 122.3 ms either way,
// 4096 x 524288 x 128 9.4 -> 6.7 ms, profiles/r02_midcall.txt; the ring is gone).  Chunk results merge with
// nns_keys_min: the answer is the unsharded one bit for bit.  No threads, no pinned memory.
//
// When it pays: four shards cost ~0.25 ms of launches, so the search must be worth >= 0.5 ms (estimated from the
// measured rates: fp32 MFMA 140 TF, bf16-operand tiles 1.4 PF, exact VALU kernels 45 Tflop/s at 3k flop per pair).
#ifndef NNS_CHUNK_MIN_MB
#define NNS_CHUNK_MIN_MB 8
#endif
static bool chunked_pays(int k, int64_t m, int64_t n, int bf16, unsigned flags, size_t rbytes)
{
    if (flags & (NNS_REFS_SOA | NNS_PROFILE)) return false;
    if (rbytes < ((size_t)NNS_CHUNK_MIN_MB << 20) || n < 4096 || m < kTinyM) return false;
    const unsigned path = flags & NNS_PATH_MASK;
    const bool exact = path == NNS_PATH_EXACT || k < (bf16 ? 32 : 8) || k > kMaxFilterK;
    double est_s;
    if (exact) {
        est_s = 3.0 * k * (double)m * (double)n / 45e12;
    } else {
        int kt = bf16 ? 128 : 16;
        while (kt < k) kt *= 2;
        const bool bf16_ops = bf16 || k > 256 || (flags & NNS_FILTER_BF16);
        est_s = 2.0 * kt * (double)m * (double)n / (bf16_ops ? 1.4e15 : 140e12);
    }
    return est_s >= 0.5e-3;
}

// the core: refs of the contiguous range [base, base + n) go up chunk by chunk into r_d and are searched (queries
// resident in q_d) as they land; returns when the range's packed keys are complete in `keys` (device memory)
static int search_range_overlapped_impl(int device, int k, int m, int n, const void *q_d, const void *r_host, char *r_d,
                                        int bf16, int64_t base, unsigned flags, nns_key *keys, nns_key *keys_tmp)
{
    const size_t esz = bf16 ? sizeof(uint16_t) : sizeof(float);
    hipStream_t st = lib_stream_acquire();
    if (!st) return NNS_ERR_UNSUPPORTED;
    // the caller's query upload (synchronous copy) happens before the first search
    if (order_after_default_stream(st) != NNS_OK) {
        lib_stream_release(st);
        return NNS_ERR_HIP;
    }
    std::vector<nns_index *> shards;
    int rc = NNS_OK;
    // chunks of 1/8, 1/4, 5/16, 5/16 of the refs (multiples of 512: whole ring slots at every tile depth)
    int bounds[5] = {0, 0, 0, 0, n};
    {
        const int64_t unit = 512;
        bounds[1] = (int)(((int64_t)n * 2 / 16 + unit - 1) / unit * unit);
        bounds[2] = (int)(((int64_t)n * 6 / 16 + unit - 1) / unit * unit);
        bounds[3] = (int)(((int64_t)n * 11 / 16 + unit - 1) / unit * unit);
        for (int c = 1; c < 4; ++c)
            if (bounds[c] > n) bounds[c] = n;
    }
    bool first = true;
    for (int c = 0; c < 4 && rc == NNS_OK; ++c) {
        const int beg = bounds[c], cnt = bounds[c + 1] - bounds[c];
        if (cnt <= 0) continue;
        // synchronous: the chunk is on the device when this returns; the previous chunk's search keeps running
        if (hipMemcpy(r_d + (size_t)beg * k * esz, (const char *)r_host + (size_t)beg * k * esz, (size_t)cnt * k * esz,
                      hipMemcpyHostToDevice) != hipSuccess) {
            rc = NNS_ERR_HIP;
            break;
        }
        // explicit edge copy -> this chunk's kernels (an event on the default stream: it does not wait for the
        // previous chunk's search, which runs on the non-blocking stream)
        if ((rc = order_after_default_stream(st)) != NNS_OK) break;
        nns_index *ix = nullptr;
        rc = index_create_impl(&ix, device, k, cnt, r_d + (size_t)beg * k * esz, bf16, base + beg, flags | kCreateNoSync, st);
        if (rc != NNS_OK) break;
        shards.push_back(ix);
        rc = index_search_impl(ix, m, q_d, bf16, first ? keys : keys_tmp, st);
        if (rc == NNS_OK && !first) rc = nns_keys_min(keys, keys_tmp, m, st);
        first = false;
    }
    if (rc == NNS_OK && hipStreamSynchronize(st) != hipSuccess) rc = NNS_ERR_HIP;
    if (rc == NNS_ERR_HIP) set_error("nns_search (chunked upload): %s", hipGetErrorString(hipGetLastError()));
    if (rc != NNS_OK) (void)hipStreamSynchronize(st);   // error paths: the caller frees r_d / keys right away
    for (nns_index *ix : shards) index_destroy_impl(ix, true);   // (the stream has been waited for on every path)
    lib_stream_release(st);
    return rc;
}

extern "C++" {
namespace nns {
// for nns_multi.hip: one GPU's shard of nns_search_*_multi takes the same overlapped upload
bool upload_overlap_pays(int k, int64_t m, int64_t n, int bf16, unsigned flags, size_t rbytes)
{
    if ((flags & NNS_PATH_MASK) == NNS_PATH_AUTO && (m < kTinyM || (!bf16 && small_exact(k, m, n)))) flags |= NNS_PATH_EXACT;
    return chunked_pays(k, m, n, bf16, flags, rbytes);
}
int search_range_overlapped(int device, int k, int m, int n, const void *q_d, const void *r_host, char *r_d, int bf16,
                            int64_t base, unsigned flags, nns_key *keys, nns_key *keys_tmp)
{
    return search_range_overlapped_impl(device, k, m, n, q_d, r_host, r_d, bf16, base, flags, keys, keys_tmp);
}
}  // namespace nns
}  // extern "C++"

static int search_host_chunked(int k, int m, int n, const void *s_points, const void *r_points, int bf16,
                               int *idx_out, float *dist_out, unsigned flags, int device)
{
    const size_t esz = bf16 ? sizeof(uint16_t) : sizeof(float);
    const size_t qb = (size_t)m * k * esz, rb = (size_t)n * k * esz;
    char *q_d = nullptr, *r_d = nullptr;
    float *dist_d = nullptr;
    nns_key *keys = nullptr, *keys_tmp = nullptr;
    int *idx_d = nullptr;
    int rc = NNS_OK;
    do {
        if (pool_alloc(&q_d, qb) != hipSuccess || pool_alloc(&r_d, rb) != hipSuccess ||
            pool_alloc(&keys, (size_t)m * sizeof(nns_key)) != hipSuccess ||
            pool_alloc(&keys_tmp, (size_t)m * sizeof(nns_key)) != hipSuccess ||
            pool_alloc(&idx_d, (size_t)m * sizeof(int)) != hipSuccess ||
            pool_alloc(&dist_d, (size_t)m * sizeof(float)) != hipSuccess) {
            set_error("nns_search: device allocation failed");
            rc = NNS_ERR_NOMEM;
            break;
        }
        if (hipMemcpy(q_d, s_points, qb, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("nns_search (chunked upload): %s", hipGetErrorString(hipGetLastError()));
            rc = NNS_ERR_HIP;
            break;
        }
        rc = search_range_overlapped_impl(device, k, m, n, q_d, r_points, r_d, bf16, 0, flags, keys, keys_tmp);
        // (UNSUPPORTED: no stream — the caller takes the plain path; NOMEM likewise: the four chunk indexes carry
        //  four query-side workspaces, the plain single-index path may still fit)
        if (rc == NNS_ERR_NOMEM) rc = NNS_ERR_UNSUPPORTED;
        if (rc != NNS_OK) break;
        rc = nns_keys_unpack(keys, m, idx_d, dist_d, nullptr);
        if (rc != NNS_OK) break;
        if (hipMemcpy(idx_out, idx_d, (size_t)m * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
            (dist_out && hipMemcpy(dist_out, dist_d, (size_t)m * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)) {
            set_error("nns_search (chunked upload): D2H copy failed: %s", hipGetErrorString(hipGetLastError()));
            rc = NNS_ERR_HIP;
        }
    } while (0);
    // (the overlapped core has waited for its stream; unpack + the synchronous downloads ran on the default stream)
    if (rc != NNS_OK) (void)hipStreamSynchronize(nullptr);
    pool_free(q_d);
    pool_free(r_d);
    pool_free(keys);
    pool_free(keys_tmp);
    pool_free(idx_d);
    pool_free(dist_d);
    return rc;
}

static int search_host_impl(int k, int m, int n, const void *s_points, const void *r_points, int bf16,
                            int *idx_out, float *dist_out, int num_shards, unsigned flags, int device)
{
    const size_t esz = bf16 ? sizeof(uint16_t) : sizeof(float);
    if (k <= 0 || m <= 0 || n <= 0 || !s_points || !r_points || !idx_out) {
        set_error("nns_search_f32: k, m, n must be > 0 and pointers non-null (k=%d m=%d n=%d)", k, m, n);
        return NNS_ERR_INVALID;
    }
    if (m > kMaxPoints || n > kMaxPoints) {
        set_error("nns_search_f32: m = %d / n = %d exceeds NNS_MAX_POINTS (%d)", m, n, kMaxPoints);
        return NNS_ERR_INVALID;
    }
    if ((int64_t)k * m > 0x7FFFFFFFll * 4 || (int64_t)k * n > 0x7FFFFFFFll * 4) {
        set_error("nns_search_f32: point set too large for one call");
        return NNS_ERR_INVALID;
    }
    DeviceScope keep_device;
    NNS_TRY(ensure_device_ok(device));
    if (num_shards < 1) num_shards = 1;
    if (num_shards > n) num_shards = n;   // the reference clamps GPUs to n (core.cu:771-772)
    if ((flags & NNS_PATH_MASK) == NNS_PATH_AUTO && (m < kTinyM || (!bf16 && small_exact(k, m, n)))) flags |= NNS_PATH_EXACT;
    if (num_shards == 1) {
        const int src = search_host_small(k, m, n, s_points, r_points, bf16, idx_out, dist_out, flags, device);
        if (src != NNS_ERR_UNSUPPORTED) return src;
    }
    if (num_shards == 1 && chunked_pays(k, m, n, bf16, flags, (size_t)n * k * esz)) {
        const int crc = search_host_chunked(k, m, n, s_points, r_points, bf16, idx_out, dist_out, flags, device);
        if (crc != NNS_ERR_UNSUPPORTED) return crc;   // (UNSUPPORTED: no stream -> plain path)
    }

    char *q_d = nullptr, *r_d = nullptr, *r_t = nullptr;
    float *dist_d = nullptr;
    nns_key *keys = nullptr, *keys_tmp = nullptr;
    int *idx_d = nullptr;
    // kernels on a non-blocking stream of the library (nullptr if none can be made: the default stream then);
    // the synchronous copies stay on the default stream, with explicit event edges in between
    hipStream_t st = lib_stream_acquire();
    int rc = NNS_OK;
    nns_index *ix = nullptr;
    do {
        const size_t qb = (size_t)m * k * esz, rb = (size_t)n * k * esz;
        if (pool_alloc(&q_d, qb) != hipSuccess || pool_alloc(&r_d, rb) != hipSuccess ||
            pool_alloc(&keys, (size_t)m * sizeof(nns_key)) != hipSuccess ||
            pool_alloc(&keys_tmp, (size_t)m * sizeof(nns_key)) != hipSuccess ||
            pool_alloc(&idx_d, (size_t)m * sizeof(int)) != hipSuccess ||
            pool_alloc(&dist_d, (size_t)m * sizeof(float)) != hipSuccess) {
            set_error("nns_search_f32: device allocation failed");
            rc = NNS_ERR_NOMEM;
            break;
        }
        // Everything of this call on the ONE library stream — uploads, kernels, downloads (the caller's pageable
        // buffers stay valid until the wait below): copies on the default stream + kernels on the library's stream
        // meant an event edge between two hardware queues on either side of the search, 45 us of a 60 us call.
#ifdef NNS_PLAIN_SYNC_COPIES   // (A/B builds: synchronous copies on the default stream, event edge to the kernels)
        if (hipMemcpy(q_d, s_points, qb, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(r_d, r_points, rb, hipMemcpyHostToDevice) != hipSuccess) {
            set_error("nns_search_f32: H2D copy failed: %s", hipGetErrorString(hipGetLastError()));
            rc = NNS_ERR_HIP;
            break;
        }
        if ((rc = order_after_default_stream(st)) != NNS_OK) break;   // uploads -> kernels
#else
        if (hipMemcpyAsync(q_d, s_points, qb, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(r_d, r_points, rb, hipMemcpyHostToDevice, st) != hipSuccess) {
            set_error("nns_search_f32: H2D copy failed: %s", hipGetErrorString(hipGetLastError()));
            rc = NNS_ERR_HIP;
            break;
        }
#endif
        if (flags & NNS_REFS_SOA) {
            // dimension-major refs: transpose once on the device, then shard the point-major copy
            if (pool_alloc(&r_t, rb) != hipSuccess) {
                set_error("nns_search_f32: device allocation failed (point-major copy)");
                rc = NNS_ERR_NOMEM;
                break;
            }
            if ((rc = launch_soa_to_aos(k, n, r_d, r_t, (int)esz, st)) != NNS_OK) break;
            char *tmp = r_d;
            r_d = r_t;
            r_t = tmp;
            flags &= ~(unsigned)NNS_REFS_SOA;
        }
        // contiguous ceil(n / shards) ranges (reference split rule core.cu:781-791)
        const int per = divup(n, num_shards);
        bool first = true;
        for (int s = 0; s < num_shards && rc == NNS_OK; ++s) {
            const int beg = s * per;
            const int cnt = ((int64_t)beg + per <= n) ? per : n - beg;
            if (cnt <= 0) break;
            rc = index_create_impl(&ix, device, k, cnt, r_d + (size_t)beg * k * esz, bf16, beg, flags, st);
            if (rc != NNS_OK) break;
            // one shard: indices (and distances) straight out of the search (K1a: the same launch)
            rc = num_shards == 1 ? index_search_impl(ix, m, q_d, bf16, keys, st, idx_d, dist_d)
                                 : index_search_impl(ix, m, q_d, bf16, first ? keys : keys_tmp, st);
            if (rc == NNS_OK && !first) rc = nns_keys_min(keys, keys_tmp, m, st);
            // (ONE shard: the index lives until the wait that comes with the download, below)
            if (num_shards == 1) break;
            // several shards: each is waited for before its index goes
            bool idle = rc == NNS_OK;
            if (rc == NNS_OK && hipStreamSynchronize(st) != hipSuccess) {
                set_error("nns_search_f32: kernel execution failed: %s", hipGetErrorString(hipGetLastError()));
                rc = NNS_ERR_HIP;
                idle = false;
            }
            index_destroy_impl(ix, idle);
            ix = nullptr;
            first = false;
        }
        if (rc != NNS_OK) break;
        if (num_shards > 1) rc = nns_keys_unpack(keys, m, idx_d, dist_d, st);
        if (rc != NNS_OK) break;
        if (hipMemcpyAsync(idx_out, idx_d, (size_t)m * sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess ||
            (dist_out && hipMemcpyAsync(dist_out, dist_d, (size_t)m * sizeof(float), hipMemcpyDeviceToHost, st) != hipSuccess) ||
            hipStreamSynchronize(st) != hipSuccess) {
            set_error("nns_search_f32: kernel execution or D2H copy failed: %s", hipGetErrorString(hipGetLastError()));
            rc = NNS_ERR_HIP;
        }
    } while (0);
    // success: the stream was waited for with the download; error paths: behind an event on the stream
    if (ix) index_destroy_impl(ix, rc == NNS_OK);
    void *const blocks[] = {q_d, r_d, r_t, keys, keys_tmp, idx_d, dist_d};
    if (rc == NNS_OK) for (void *b : blocks) pool_free(b);
    else pool_free_after(blocks, 7, st);
    lib_stream_release(st);
    return rc;
}

int nns_search_f32_ex(int k, int m, int n, const float *s_points, const float *r_points, int *idx_out,
                      float *dist_out, int num_shards, unsigned flags, int device)
{
    return search_host_impl(k, m, n, s_points, r_points, 0, idx_out, dist_out, num_shards, flags, device);
}

int nns_search_bf16_ex(int k, int m, int n, const uint16_t *s_points, const uint16_t *r_points, int *idx_out,
                       float *dist_out, int num_shards, unsigned flags, int device)
{
    return search_host_impl(k, m, n, s_points, r_points, 1, idx_out, dist_out, num_shards, flags, device);
}

int nns_warmup(int device)
{
    DeviceScope keep_device;
    NNS_TRY(ensure_device_ok(device));
    // (k, m, n, bf16, path): K1a (3-D and 16-D), K1b, the fp32 tile depths 16 / 32 / 64 / 128 / 256 (forced onto the
    // filter: AUTO keeps a 64 x 512 x 16 problem on the exact kernel), the bf16-operand tiles for fp32 points (512 /
    // 384 / 512 / 640 / 768 / 1024: AUTO), the bf16 tiles 128 / 256 / 384 / 512 / 640 / 768 / 1024
    static const int shapes[][5] = {{3, 64, 512, 0, NNS_PATH_AUTO},    {16, 64, 512, 0, NNS_PATH_AUTO},  {16, 1, 512, 0, NNS_PATH_AUTO},
                                    {16, 64, 512, 0, NNS_PATH_MFMA},   {24, 64, 512, 0, NNS_PATH_MFMA},  {40, 64, 512, 0, NNS_PATH_MFMA},
                                    {100, 64, 512, 0, NNS_PATH_MFMA},  {200, 64, 512, 0, NNS_PATH_MFMA}, {300, 64, 512, 0, NNS_PATH_AUTO},
                                    {400, 64, 512, 0, NNS_PATH_AUTO},  {400, 64, 512, 1, NNS_PATH_MFMA},
                                    {600, 64, 512, 0, NNS_PATH_AUTO},  {700, 64, 512, 0, NNS_PATH_AUTO}, {800, 64, 512, 0, NNS_PATH_AUTO},
                                    {64, 64, 512, 1, NNS_PATH_MFMA},   {200, 64, 512, 1, NNS_PATH_MFMA}, {300, 64, 512, 1, NNS_PATH_MFMA},
                                    {600, 64, 512, 1, NNS_PATH_MFMA},  {700, 64, 512, 1, NNS_PATH_MFMA}, {800, 64, 512, 1, NNS_PATH_MFMA}};
    const int kmax = 800, mmax = 64, nmax = 512;
    float *q = (float *)malloc(sizeof(float) * kmax * mmax), *r = (float *)malloc(sizeof(float) * kmax * nmax);
    int *idx = (int *)malloc(sizeof(int) * mmax);
    if (!q || !r || !idx) {
        free(q);
        free(r);
        free(idx);
        return NNS_ERR_NOMEM;
    }
    unsigned x = 12345u;   // any finite values do
    for (int i = 0; i < kmax * mmax; ++i) q[i] = (float)((x = x * 1664525u + 1013904223u) >> 8) * (1.0f / 16777216.0f);
    for (int i = 0; i < kmax * nmax; ++i) r[i] = (float)((x = x * 1664525u + 1013904223u) >> 8) * (1.0f / 16777216.0f);
    int rc = NNS_OK;
    for (const auto &sh : shapes) {
        if (sh[3]) {
            // the same numbers as bf16 bit patterns (truncated: this is a warm-up, not a result)
            uint16_t *qb = (uint16_t *)malloc(sizeof(uint16_t) * sh[0] * sh[1]);
            uint16_t *rb = (uint16_t *)malloc(sizeof(uint16_t) * sh[0] * sh[2]);
            if (qb && rb) {
                for (int i = 0; i < sh[0] * sh[1]; ++i) {
                    unsigned u;
                    memcpy(&u, &q[i], 4);
                    qb[i] = (uint16_t)(u >> 16);
                }
                for (int i = 0; i < sh[0] * sh[2]; ++i) {
                    unsigned u;
                    memcpy(&u, &r[i], 4);
                    rb[i] = (uint16_t)(u >> 16);
                }
                rc = search_host_impl(sh[0], sh[1], sh[2], qb, rb, 1, idx, nullptr, 1, (unsigned)sh[4], device);
            } else {
                rc = NNS_ERR_NOMEM;
            }
            free(qb);
            free(rb);
        } else {
            rc = search_host_impl(sh[0], sh[1], sh[2], q, r, 0, idx, nullptr, 1, (unsigned)sh[4], device);
        }
        if (rc != NNS_OK) break;
    }
    free(q);
    free(r);
    free(idx);
    return rc;
}

int nns_search_f32(int k, int m, int n, const float *s_points, const float *r_points, int **results)
{
    if (!results) {
        set_error("nns_search_f32: results is null");
        return NNS_ERR_INVALID;
    }
    *results = nullptr;
    if (m <= 0) {
        set_error("nns_search_f32: m must be > 0");
        return NNS_ERR_INVALID;
    }
    int *tmp = (int *)malloc(sizeof(int) * (size_t)m);   // caller free()s, as core.cu:31,52
    if (!tmp) return NNS_ERR_NOMEM;
    const int rc = nns_search_f32_ex(k, m, n, s_points, r_points, tmp, nullptr, 1, NNS_PATH_AUTO, 0);
    if (rc != NNS_OK) {
        free(tmp);
        return rc;
    }
    *results = tmp;
    return NNS_OK;
}

}  // extern "C"
