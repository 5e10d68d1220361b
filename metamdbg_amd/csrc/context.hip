// context.hip -- context lifetime, error reporting, timing of libmdbg_hip.so.
#include "common.hpp"

#include <algorithm>
#include <cstdlib>

namespace mdbg {

thread_local std::string g_last_error;

int set_error(mdbg_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_last_error = buf;
    return code;
}

static void fold_timers(mdbg_ctx *ctx) {
    if (ctx->launches.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->side_stream) (void)hipStreamSynchronize(ctx->side_stream);
    for (auto &t : ctx->launches) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, t.start, t.stop) == hipSuccess) {
            auto &slot = ctx->timers[t.name];
            slot.first += ms;
            slot.second += 1;
        }
        (void)hipEventDestroy(t.start);
        (void)hipEventDestroy(t.stop);
    }
    ctx->launches.clear();
}

}  // namespace mdbg

using namespace mdbg;

extern "C" int mdbg_create(int device, mdbg_ctx **out) try {
    if (!out) return set_error(nullptr, MDBG_EINVAL, "mdbg_create: null out pointer");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        return set_error(nullptr, MDBG_ENODEV, "mdbg_create: no HIP device (%s); this library has no CPU path",
                         e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
    if (device < 0 || device >= count)
        return set_error(nullptr, MDBG_EINVAL, "mdbg_create: device %d out of range (%d devices)", device, count);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess)
        return set_error(nullptr, MDBG_EHIP, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_error(nullptr, MDBG_ENODEV, "mdbg_create: device %d is %s; kernels are built for gfx950 only",
                         device, prop.gcnArchName);
    if ((e = hipSetDevice(device)) != hipSuccess)
        return set_error(nullptr, MDBG_EHIP, "hipSetDevice: %s", hipGetErrorString(e));
    mdbg_ctx *ctx = new mdbg_ctx();
    ctx->device = device;
    ctx->arch = prop.gcnArchName;
    ctx->n_cu = prop.multiProcessorCount;
    // defaults of the per-context options from the environment (mdbg_set_option changes them later)
    if (const char *e = getenv("MDBG_TABLE_BLOCKS_PER_CU")) if (atoi(e) > 0) ctx->table_blocks_per_cu = (unsigned)atoi(e);
    if (const char *e = getenv("MDBG_SCAN_WAVE_PRIORITY")) ctx->scan_wave_priority = (uint32_t)std::max(0, std::min(3, atoi(e)));
    if (const char *e = getenv("MDBG_SCAN_READS_PER_WAVE")) if (atoi(e) > 0) ctx->scan_reads_per_wave = (unsigned)atoi(e);
    if (const char *e = getenv("MDBG_SCAN_LDS_RESERVE")) ctx->scan_lds_reserve = (uint32_t)std::max(0, std::min(131072, atoi(e)));
    if (const char *e = getenv("MDBG_SCAN_LDS_PAD")) ctx->scan_lds_pad = (uint32_t)std::max(0, std::min(32768, atoi(e)));
    if (const char *e = getenv("MDBG_PARTITION_TILE")) ctx->part_tile = atoi(e) == 2048 ? 2048u : 0u;
    if (const char *e = getenv("MDBG_PARTITION_SLOT_LIST")) ctx->part_slot_list = atoi(e) != 0;
    if (const char *e = getenv("MDBG_FIRST_PASS_MODE")) ctx->first_pass_mode = std::max(0, std::min(2, atoi(e)));
    ctx->hbm_bytes = prop.totalGlobalMem;
    ctx->lds_per_cu = prop.maxSharedMemoryPerMultiProcessor;
    ctx->clock_khz = prop.clockRate;
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking)) != hipSuccess) {
        if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
        delete ctx;
        return set_error(nullptr, MDBG_EHIP, "hipStreamCreate: %s", hipGetErrorString(e));
    }
    ctx->pool = std::make_shared<DevPool>();
    ctx->pool->device = device;
    // freed blocks are kept up to a little over half of the HBM (an allocation that fails drops the cache and retries): a
    // batch of 100 Gbp with qualities turns over tens of GB of scratch per call, and hipFree / hipMalloc of such blocks
    // cost hundreds of milliseconds
    if (ctx->hbm_bytes) ctx->pool->cache_limit = std::max<size_t>(ctx->pool->cache_limit, (size_t)((double)ctx->hbm_bytes * 0.55));
    *out = ctx;
    return MDBG_OK;
} MDBG_API_CATCH((mdbg_ctx *)nullptr)

extern "C" void mdbg_destroy(mdbg_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    fold_timers(ctx);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->pool) ctx->pool->close();      // cached blocks are freed now; blocks still handed out free themselves
    if (ctx->scan_stream) { (void)hipStreamSynchronize(ctx->scan_stream); (void)hipStreamDestroy(ctx->scan_stream); }
    if (ctx->upload_stream) { (void)hipStreamSynchronize(ctx->upload_stream); (void)hipStreamDestroy(ctx->upload_stream); }
    if (ctx->side_stream) { (void)hipStreamSynchronize(ctx->side_stream); (void)hipStreamDestroy(ctx->side_stream); }
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" const char *mdbg_last_error(const mdbg_ctx *ctx) {
    return ctx ? ctx->err.c_str() : g_last_error.c_str();
}

extern "C" int mdbg_synchronize(mdbg_ctx *ctx) try {
    if (!ctx) return MDBG_EINVAL;
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

// one wave that does nothing for `ticks` of the constant-rate wall clock
__global__ void spin_kernel(long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" int mdbg_stream_spin(mdbg_ctx *ctx, uint32_t microseconds) try {
    if (!ctx) return MDBG_EINVAL;
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    int khz = 0;
    MDBG_HIP_CHECK(ctx, hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, ctx->device));
    if (khz <= 0) khz = 100000;
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, ctx->stream, (long long)microseconds * khz / 1000);
    MDBG_HIP_CHECK(ctx, hipGetLastError());
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" void *mdbg_stream(mdbg_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }

// "table_cu_count" = c > 0: the context's stream is re-created confined to c compute units (hipExtStreamCreateWithCUMask; mask
// bits are dealt round-robin over the XCDs, so the low c bits are c / 8 CUs of every XCD) and the block-structured scan kernel
// gets a stream of its own over every CU.  For several contexts in flight on one device: the table kernels of one batch -- bound
// by the atomic rate and by latency, not by the number of CUs -- then take CUs from another batch's scan only where they are
// confined, instead of a share of every CU.  0 restores one unconfined stream.
static int set_table_cu_count(mdbg_ctx *ctx, unsigned c) {
    if (c > (unsigned)ctx->n_cu) c = (unsigned)ctx->n_cu;
    if (c == ctx->table_cu_count) return MDBG_OK;
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    fold_timers(ctx);                                    // pending events belong to the old streams
    if (c > (unsigned)ctx->n_cu) c = (unsigned)ctx->n_cu;
    hipStream_t fresh = nullptr, scan = nullptr;
    if (c) {
        std::vector<uint32_t> mask(((unsigned)ctx->n_cu + 31u) / 32u, 0u);
        for (unsigned i = 0; i < c; i++) mask[i >> 5] |= 1u << (i & 31u);
        MDBG_HIP_CHECK(ctx, hipExtStreamCreateWithCUMask(&fresh, (uint32_t)mask.size(), mask.data()));
        hipError_t e = hipStreamCreateWithFlags(&scan, hipStreamNonBlocking);
        if (e != hipSuccess) { (void)hipStreamDestroy(fresh); return set_error(ctx, MDBG_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    } else {
        MDBG_HIP_CHECK(ctx, hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
    }
    if (ctx->scan_stream) { (void)hipStreamSynchronize(ctx->scan_stream); (void)hipStreamDestroy(ctx->scan_stream); }
    (void)hipStreamDestroy(ctx->stream);
    ctx->stream = fresh;
    ctx->scan_stream = scan;
    ctx->table_cu_count = c;
    return MDBG_OK;
}

extern "C" int mdbg_set_option(mdbg_ctx *ctx, const char *name, int64_t value) {
    if (!ctx || !name) return set_error(ctx, MDBG_EINVAL, "mdbg_set_option: null argument");
    const std::string n(name);
    if (n == "pool_cache_percent") {
        const int64_t pc = value > 0 ? std::min<int64_t>(value, 95) : 55;
        std::lock_guard<std::mutex> g(ctx->pool->mu);
        ctx->pool->cache_limit = std::max<size_t>((size_t)96 << 30, (size_t)((double)ctx->hbm_bytes * (double)pc / 100.0));
        return MDBG_OK;
    }
    if (n == "pool_trim") {
        MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
        MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        ctx->pool->trim();
        return MDBG_OK;
    }
    if (n == "table_cu_count") return set_table_cu_count(ctx, value > 0 ? (unsigned)std::min<int64_t>(value, 4096) : 0u);
    if (n == "table_grid_blocks") { ctx->table_grid_blocks = value > 0 ? (unsigned)std::min<int64_t>(value, 1 << 20) : 0u; return MDBG_OK; }
    if (n == "table_blocks_per_cu") { ctx->table_blocks_per_cu = value > 0 ? (unsigned)std::min<int64_t>(value, 1024) : 1024u; return MDBG_OK; }
    if (n == "scan_wave_priority") { ctx->scan_wave_priority = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(3, value)); return MDBG_OK; }
    if (n == "scan_candidate_slack") { ctx->scan_cand_slack = value > 0 ? (uint32_t)std::min<int64_t>(value, 1 << 24) : 0u; return MDBG_OK; }
    if (n == "scan_reads_per_wave") { ctx->scan_reads_per_wave = value > 0 ? (unsigned)std::min<int64_t>(value, 1 << 20) : 2u; return MDBG_OK; }
    if (n == "first_pass_mode") { ctx->first_pass_mode = (int)std::max<int64_t>(0, std::min<int64_t>(2, value)); return MDBG_OK; }
    if (n == "partition_auto_min") { ctx->part_auto_min = value > 0 ? (uint64_t)value : (1ull << 17); return MDBG_OK; }
    if (n == "partition_bits") { ctx->part_bits = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(24, value)); return MDBG_OK; }
    if (n == "partition_lds_slots") {
        if (value != 0 && value != 256 && value != 1024 && value != 2048) return set_error(ctx, MDBG_EINVAL, "partition_lds_slots: 0, 256, 1024 or 2048");
        ctx->part_lds_slots = (uint32_t)value; return MDBG_OK;
    }
    if (n == "partition_tile") { ctx->part_tile = value == 2048 ? 2048u : 0u; return MDBG_OK; }
    if (n == "partition_slot_list") { ctx->part_slot_list = value != 0; return MDBG_OK; }
    if (n == "scan_lds_reserve") { ctx->scan_lds_reserve = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(131072, value)); return MDBG_OK; }
    if (n == "scan_lds_pad") { ctx->scan_lds_pad = (uint32_t)std::max<int64_t>(0, std::min<int64_t>(32768, value)); return MDBG_OK; }
    if (n == "partition_max_records") { ctx->part_max_records = value > 0 ? (uint64_t)value : 0; return MDBG_OK; }
    if (n == "test_exchange_fail_phase") { ctx->test_exchange_fail_phase = (int)std::max<int64_t>(0, std::min<int64_t>(3, value)); return MDBG_OK; }
    if (n == "index_table_form") { ctx->index_table_form = value == 0 ? 0u : 1u; return MDBG_OK; }       // (value < 0: the default, 1)
    if (n == "refined_form") { ctx->refined_form = value == 0 ? 0u : 1u; return MDBG_OK; }
    if (n == "scan_quality_stream") { ctx->scan_quality_beside = value == 0 ? 0u : 1u; return MDBG_OK; }
    if (n == "keep_index_table") { ctx->keep_index_table = value != 0; return MDBG_OK; }
    if (n == "index_tuning") { ctx->index_tuning = value < 0 ? INDEX_TUNING_DEFAULT : (uint32_t)(value & 511); return MDBG_OK; }
    if (n == "test_corrupt_replies") { ctx->test_corrupt_replies = value > 0; return MDBG_OK; }
    return set_error(ctx, MDBG_EINVAL, "mdbg_set_option: unknown option '%s'", name);
}

extern "C" int mdbg_device_info(mdbg_ctx *ctx, char *arch, size_t arch_len, int *n_cu, uint64_t *hbm_bytes) try {
    if (!ctx) return MDBG_EINVAL;
    if (arch && arch_len) {
        strncpy(arch, ctx->arch.c_str(), arch_len - 1);
        arch[arch_len - 1] = 0;
    }
    if (n_cu) *n_cu = ctx->n_cu;
    if (hbm_bytes) *hbm_bytes = ctx->hbm_bytes;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_device_clock_khz(mdbg_ctx *ctx, int *clock_khz) {
    if (!ctx || !clock_khz) return MDBG_EINVAL;
    *clock_khz = ctx->clock_khz;
    return MDBG_OK;
}

extern "C" int mdbg_timing_enable(mdbg_ctx *ctx, int on) try {
    if (!ctx) return MDBG_EINVAL;
    ctx->timing = on != 0;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_timing_reset(mdbg_ctx *ctx) try {
    if (!ctx) return MDBG_EINVAL;
    fold_timers(ctx);
    ctx->timers.clear();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_timing_get(mdbg_ctx *ctx, const char *kernel, double *ms_total, uint64_t *launches) try {
    if (!ctx || !kernel) return MDBG_EINVAL;
    fold_timers(ctx);
    auto it = ctx->timers.find(kernel);
    if (ms_total) *ms_total = it == ctx->timers.end() ? 0.0 : it->second.first;
    if (launches) *launches = it == ctx->timers.end() ? 0 : it->second.second;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_host_alloc(mdbg_ctx *ctx, size_t bytes, void **out) try {
    if (!ctx || !out) return set_error(ctx, MDBG_EINVAL, "mdbg_host_alloc: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipError_t e = hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault);
    if (e != hipSuccess) return set_error(ctx, MDBG_ENOMEM, "hipHostMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" void mdbg_host_free(mdbg_ctx *ctx, void *p) {
    (void)ctx;
    if (p) (void)hipHostFree(p);
}
