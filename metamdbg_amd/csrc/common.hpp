// common.hpp -- context, error plumbing, device buffers and wave64 helpers shared by the
// HIP translation units of libmdbg_hip.so.  gfx950 (MI355X) only: 64-lane wavefronts.
#pragma once
#include <ctime>
#include <cstdio>
#include <cstdlib>

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/mdbg_hip.h"

namespace mdbg {

constexpr int WAVE = 64;

// Size-class cache of device allocations.  hipMalloc / hipFree of GB-sized buffers cost milliseconds and
// hipFree synchronises the device; a step of the hot path allocates the same few dozen buffers every
// time, so freed blocks are kept and handed out again.  Everything a context launches goes to ONE
// stream, so reusing a block is ordered after its previous users by stream order.  The pool is shared
// (shared_ptr) by the context and every buffer it handed out, so handles may outlive the context.
// One lock for everything that changes the process's device address space: hipMalloc / hipFree of the pools and the peer-copy staging
// buffers, hipIpcGetMemHandle / hipIpcOpenMemHandle / hipIpcCloseMemHandle.  Several contexts are driven from several host threads; while one
// thread's exchange opened a peer's buffer and the others' pools were still growing (the warm-up steps of an N > 1 job), one run in ten of
// tests/test_gpu_multirank.py ended in a GPU memory access fault (round 5).  These calls are rare in steady state; taking turns costs nothing.
inline std::mutex &hip_mem_mutex() { static std::mutex mu; return mu; }

struct DevPool {
    std::mutex mu;
    std::multimap<size_t, void *> free_list;        // capacity -> block
    std::unordered_map<void *, size_t> capacity;    // every block this pool owns (free or handed out)
    size_t cached_bytes = 0;
    size_t cache_limit = (size_t)96 << 30;
    int device = 0;
    bool closed = false;

    static size_t size_class(size_t bytes) {
        if (bytes < 256) return 256;
        size_t p2 = 256;
        while (p2 * 2 <= bytes) p2 *= 2;            // largest power of two <= bytes
        size_t step = p2 / 8;                       // 12.5 % granularity
        return (bytes + step - 1) / step * step;
    }
    hipError_t get(size_t bytes, void **out) {
        const size_t cap = size_class(bytes);
        {
            std::lock_guard<std::mutex> g(mu);
            auto it = free_list.lower_bound(cap);
            if (it != free_list.end() && it->first <= cap + cap / 4) {
                *out = it->second;
                cached_bytes -= it->first;
                free_list.erase(it);
                return hipSuccess;
            }
        }
        hipError_t e;
        { std::lock_guard<std::mutex> g(hip_mem_mutex()); e = hipMalloc(out, cap); }
        if (e != hipSuccess) {                      // out of memory: drop the cache and retry once
            trim();
            (void)hipGetLastError();
            std::lock_guard<std::mutex> g(hip_mem_mutex());
            e = hipMalloc(out, cap);
        }
        if (e == hipSuccess) {
            std::lock_guard<std::mutex> g(mu);
            capacity[*out] = cap;
        }
        return e;
    }
    void put(void *p) {
        if (!p) return;
        std::unique_lock<std::mutex> g(mu);
        auto it = capacity.find(p);
        size_t cap = it == capacity.end() ? 0 : it->second;
        if (closed || cap == 0 || cached_bytes + cap > cache_limit) {
            if (it != capacity.end()) capacity.erase(it);
            g.unlock();
            std::lock_guard<std::mutex> gm(hip_mem_mutex());
            (void)hipFree(p);
            return;
        }
        free_list.emplace(cap, p);
        cached_bytes += cap;
    }
    void trim() {
        std::vector<void *> victims;
        {
            std::lock_guard<std::mutex> g(mu);
            for (auto &kv : free_list) { victims.push_back(kv.second); capacity.erase(kv.second); }
            free_list.clear();
            cached_bytes = 0;
        }
        std::lock_guard<std::mutex> gm(hip_mem_mutex());
        for (void *p : victims) (void)hipFree(p);
    }
    void close() { { std::lock_guard<std::mutex> g(mu); closed = true; } trim(); }
    ~DevPool() { trim(); }
};

struct TimedLaunch {
    const char *name;
    hipEvent_t start, stop;
};

}  // namespace mdbg

constexpr uint32_t INDEX_TUNING_DEFAULT = 3u | 16u;

struct mdbg_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t scan_stream = nullptr;      // "table_cu_count": the block-structured scan kernel runs here (every CU) while `stream`, which
                                            // carries everything else, is confined to a few CUs (mdbg_set_option)
    unsigned table_cu_count = 0;
    hipStream_t upload_stream = nullptr;    // mdbg_reads_from_packed_async: host-to-device copies that run beside the kernels of `stream` (created on first use)
    hipStream_t side_stream = nullptr;      // copies that must not queue behind the main stream's kernel (quality sums during the scan)
    void *pinned = nullptr;                 // grow-only pinned host buffer for them
    size_t pinned_bytes = 0;
    std::string err;
    std::string arch;
    int n_cu = 0;
    uint64_t hbm_bytes = 0;
    int clock_khz = 0;                      // hipDeviceProp_t::clockRate (peak engine clock)
    bool timing = false;
    std::vector<mdbg::TimedLaunch> launches;               // pending (not yet folded) timed launches
    std::map<std::string, std::pair<double, uint64_t>> timers;  // name -> (ms, launches)
    unsigned table_grid_blocks = 0;                        // > 0: the absolute grid of those kernels (fewer blocks than CUs: beside a scan, mdbg_set_option)
    unsigned table_blocks_per_cu = 1024;                   // resident blocks per CU of the kernels that walk every k-min-mer instance (mdbg_set_option)
    unsigned scan_reads_per_wave = 2;                      // reads a scan wave processes before it retires (mdbg_set_option)
    uint32_t scan_wave_priority = 0;        // s_setprio level of the block-structured scan's waves (mdbg_set_option)
    int test_exchange_fail_phase = 0;       // tests: this rank fails in phase 1 (before the counts) / 2 (buffers) / 3 (reduction) of the next exchange
    bool test_corrupt_replies = false;      // tests: the next exchange hands back one reply with a wrong count (the job's self-check must see it)
    uint32_t scan_cand_slack = 0;           // tests: widens the candidate test of the block-structured scan (see span_step)
    // distinct keys per k-min-mer instance seen by the last call OF THE SAME KIND (table sizing): the first pass keeps every
    // key, refined / index only those above abundance 1 -- one shared hint made every first pass after an index pass rebuild its table
    double key_ratio_hint[4] = {0.0625, 0.0625, 0.0625, 0.0625};   // [0] first pass, [1] refined, [2] index, [3] sharded first pass
    bool key_ratio_known[4] = {false, false, false, false};        // the hint was measured by a call of that kind (a table is sized densely only then)
    // the first pass (k = firstK): instances partitioned by key and counted in LDS (partition.hip) or one global table (kminmer.hip)
    int first_pass_mode = 0;                // 0: by size (partitioned from part_auto_min instances up), 1: one table, 2: partitioned
    uint64_t part_auto_min = 1ull << 17;    // mode 0: fewer minimizers than this take the one-table path (50 000 reads: 0.18 against 0.25 ms)
    uint32_t part_bits = 0;                 // > 0: bucket bits of the first attempt (tests; default: from the key hint)
    uint32_t part_lds_slots = 0;            // 0: 1024 or 2048 by the key hint; 256 / 1024 / 2048: forced (tests)
    uint64_t part_max_records = 0;          // > 0: instances per group of keys (tests; default: a quarter of the HBM)
    uint32_t part_tile = 0;                 // 2048: the scatter regroups tiles of 2048 records (24 KB of LDS instead of 43), else 4096
    uint32_t part_slot_list = 1;            // 0: bucket_count keeps no slot list (24 KB of LDS per 1024-slot bucket instead of 40)
    uint32_t scan_lds_pad = 0;              // bytes of unused dynamic LDS per block of the block-structured scan: caps its blocks per CU and so
                                            // leaves LDS, registers and wave slots to other contexts' kernels (mdbg_set_option)
    uint32_t scan_lds_reserve = 0;          // the same as a goal: bytes of a CU's LDS the scan leaves free, the padding worked out per kernel variant
    size_t lds_per_cu = 0;                  // hipDeviceProp_t::maxSharedMemoryPerMultiProcessor
    // the passes above firstK (round 5; measured side by side in profiles/round5_b_index_table_forms_timed.json):
    uint32_t index_table_form = 1;          // 1 = one 32-byte slot per key (default); 0 = bucket tables, three keys per 64-byte sector (table.hpp): a third of
                                            // the bytes and no faster -- these passes run at the rate of random sectors whatever the table's size
    uint32_t refined_form = 1;              // k = firstK + 1: 1 = every distinct key first, then two look-ups per key (default); 0 = like an index pass (a
                                            // look-up per (k-1)-window, only kept keys inserted: 344 M look-ups instead of 78 M -- slower)
    uint32_t scan_quality_beside = 1;       // FASTQ: the per-read quality sums run beside the scan kernel on the side stream (0: in front of it, rounds 1 - 4)
    uint32_t keep_index_table = 1;          // an index pass's hash table stays with its result as the look-up structure of the next pass (round 6)
    double index_last_miss_fraction = -1.0; // the last index pass's sample: k-windows with min(prev[i], prev[i+1]) <= 1 (what chose its form)
    uint32_t index_tuning = INDEX_TUNING_DEFAULT;   // one-slot index passes: bit 0 a slot's key and value in one trip, bit 1 the insert's plain-load first look (both on:
                                            // 20.5 -> 18.5 ms a pass), bit 2 two windows of a lane in flight (measured, no gain: off), bit 3 look-up and insert in one kernel (measured, slower: off); 0 = the kernels of rounds 1 - 4;
                                            // round 6: bit 4 the look-up fetches both slots of a key's home sector at once (on; with bit 1 and bit 5 off the insert keeps round 5's first look), bit 5 32 lanes a sequence (measured, slower in the library: off)
    uint64_t part_info[8] = {0};            // last first pass: [0] path (1 one table, 2 partitioned), [1] groups, [2] bucket bits, [3] levels,
                                            // [4] attempts, [5] LDS slots per bucket, [6] buckets, [7] instances
    std::shared_ptr<mdbg::DevPool> pool;                   // device memory cache shared with every buffer handed out
};

namespace mdbg {

int set_error(mdbg_ctx *ctx, int code, const char *fmt, ...);
extern thread_local std::string g_last_error;  // for failures before a context exists (per calling thread)

#define MDBG_HIP_CHECK(ctx, expr)                                                              \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return mdbg::set_error((ctx), e_ == hipErrorOutOfMemory ? MDBG_ENOMEM : MDBG_EHIP, \
                                   "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// No C++ exception may cross the C ABI: every entry point that can allocate on the host is a function-try-block ending in this.
#define MDBG_API_CATCH(ctx)                                                                                     \
    catch (const std::bad_alloc &) { return mdbg::set_error((ctx), MDBG_ENOMEM, "host memory allocation failed"); } \
    catch (const std::exception &e_) { return mdbg::set_error((ctx), MDBG_EHIP, "internal error: %s", e_.what()); }

#define MDBG_TRY(expr)            \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != MDBG_OK) return rc_; \
    } while (0)

// Blocking copy ORDERED ON THE CONTEXT STREAM.  The stream is created non-blocking, so the legacy
// null-stream hipMemcpy would not wait for kernels queued on it.
inline hipError_t memcpy_sync(mdbg_ctx *ctx, void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, ctx->stream);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(ctx->stream);
}

// RAII device allocation (synchronous hipMalloc; sizes here are tens of MB to GB, allocated
// a handful of times per call, never per read).
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    std::shared_ptr<DevPool> pool;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), pool(std::move(o.pool)) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; pool = std::move(o.pool); o.p = nullptr; o.n = 0; }
        return *this;
    }
    ~DevBuf() { release(); }
    void release() {
        if (p) { if (pool) pool->put(p); else { std::lock_guard<std::mutex> g(hip_mem_mutex()); (void)hipFree(p); } }
        p = nullptr; n = 0;
    }
    int alloc(mdbg_ctx *ctx, size_t count) {
        release();
        if (count == 0) count = 1;
        pool = ctx->pool;
        hipError_t e = pool->get(count * sizeof(T), (void **)&p);
        if (e != hipSuccess) {
            p = nullptr;
            return set_error(ctx, MDBG_ENOMEM, "device allocation of %zu bytes failed: %s", count * sizeof(T), hipGetErrorString(e));
        }
        n = count;
        return MDBG_OK;
    }
};

// Scoped kernel timer: records HIP events on ctx->stream around a launch when timing is on.
// MDBG_DEBUG=1: wall-clock stamped progress lines on stderr from the long calls (which stage a call that does not come
// back is in)
inline bool debug_on() { static const bool on = getenv("MDBG_DEBUG") != nullptr; return on; }
#define MDBG_DBG(ctx, ...) do { if (mdbg::debug_on()) { struct timespec ts_; clock_gettime(CLOCK_MONOTONIC, &ts_); \
    fprintf(stderr, "[mdbg %p %ld.%03ld] ", (void *)(ctx), (long)ts_.tv_sec, ts_.tv_nsec / 1000000); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

struct LaunchTimer {
    mdbg_ctx *ctx;
    TimedLaunch t{};
    bool on;
    hipStream_t stream;
    LaunchTimer(mdbg_ctx *c, const char *name, hipStream_t s = nullptr) : ctx(c), on(c->timing), stream(s ? s : c->stream) {
        if (!on) return;
        t.name = name;
        if (hipEventCreate(&t.start) != hipSuccess || hipEventCreate(&t.stop) != hipSuccess) { on = false; return; }
        (void)hipEventRecord(t.start, stream);
    }
    ~LaunchTimer() {
        if (!on) return;
        (void)hipEventRecord(t.stop, stream);
        ctx->launches.push_back(t);
    }
};

inline unsigned grid_for(uint64_t items, unsigned per_block, unsigned max_blocks = 1u << 30) {
    uint64_t b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    // (a launch of 2^32 threads or more does not fail, it wraps: kernels over that many items are grid-stride and name their cap)
    // (thrown, not aborted: every entry point of the C ABI turns it into an error code -- MDBG_API_CATCH)
    if (b * per_block >= (1ull << 32) && max_blocks == 1u << 30)
        throw std::length_error("a launch over " + std::to_string(items) + " items would be 2^32 threads or more (the kernel needs a capped grid)");
    return (unsigned)b;
}

// ---- device-side wave64 helpers -----------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ unsigned lane_id() { return __lane_id(); }

// Order LDS traffic between the lanes of ONE wavefront (waves of a block run independent reads,
// so __syncthreads() is not usable inside the per-read loops).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Inclusive prefix sum across the 64 lanes (shuffle ladder).
__device__ __forceinline__ unsigned wave_inclusive_sum(unsigned v) {
    unsigned lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        unsigned t = __shfl_up(v, d, 64);
        if (lane >= (unsigned)d) v += t;
    }
    return v;
}

// The same on the data-parallel-primitive path of the VALU (no LDS crossbar: __shfl_up is a ds_bpermute, 24 cycles per wave
// instruction on gfx950 against 4.4 for a DPP move -- tools/ubench/op_rates.hip): shifts inside the rows of 16 lanes, then the
// last lane of a row broadcast to the following rows.
__device__ __forceinline__ unsigned wave_inclusive_sum_dpp(unsigned v) {
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, false);   // row_shr:1
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, false);   // row_shr:2
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, false);   // row_shr:4
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, false);   // row_shr:8
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
    v += (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
    return v;
}

// value of the lane below (lane 0 receives `first`): wave_shr:1
__device__ __forceinline__ unsigned lane_below(unsigned v, unsigned first) {
    return (unsigned)__builtin_amdgcn_update_dpp((int)first, (int)v, 0x138, 0xf, 0xf, false);
}

__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}
#endif

// device-wide exclusive scan (prims.hip): out[i] = sum(in[0..i)), out[n] = total; out has n+1 entries.
int exclusive_scan_u32(mdbg_ctx *ctx, const uint32_t *d_in, uint64_t *d_out, uint64_t n);

// density (f32) -> integer threshold: hash < T  <=>  (double)hash < (double)density * 2^64, the compare of
// MinimizerParser (utils/kmer/Kmer.hpp:1421-1430) and Utils::applyDensityThreshold (Commons.hpp:2524-2533; its
// float product density * 2^64 is exact).
inline uint64_t density_threshold(float density) {
    const double bound = (double)density * 18446744073709551616.0;
    if (!((double)UINT64_MAX >= bound)) return UINT64_MAX;
    uint64_t lo = 0, hi = UINT64_MAX;
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if ((double)mid >= bound) hi = mid; else lo = mid + 1;
    }
    return lo;
}

}  // namespace mdbg
