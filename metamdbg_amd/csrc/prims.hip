// prims.hip -- small device-wide primitives: exclusive scan (reduce / scan / apply).
// Used for CSR offsets (minimizers per read, instances per read, flags -> output rows).
#include "common.hpp"

namespace mdbg {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

__device__ __forceinline__ uint64_t block_exclusive_sum(uint64_t v, uint64_t *total, uint64_t *lds /* 4 */) {
    // wave inclusive scan of 64-bit values, then combine the 4 waves through LDS
    unsigned lane = lane_id(), wave = threadIdx.x >> 6;
    uint64_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint64_t t = __shfl_up(inc, d, 64);
        if (lane >= (unsigned)d) inc += t;
    }
    if (lane == 63) lds[wave] = inc;
    __syncthreads();
    uint64_t w0 = lds[0], w1 = lds[1], w2 = lds[2], w3 = lds[3];
    uint64_t base = (wave > 0 ? w0 : 0) + (wave > 1 ? w1 : 0) + (wave > 2 ? w2 : 0);
    *total = w0 + w1 + w2 + w3;
    __syncthreads();
    return base + inc - v;
}

template <typename Tin>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const Tin *in, uint64_t n, uint64_t *block_sums) {
    __shared__ uint64_t lds[4];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        uint64_t idx = base + (uint64_t)i * SCAN_THREADS + threadIdx.x;
        if (idx < n) s += in[idx];
    }
    uint64_t total;
    block_exclusive_sum(s, &total, lds);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// single block: exclusive scan of m <= arbitrary values, sequential over tiles
__global__ __launch_bounds__(SCAN_THREADS) void scan_small_kernel(uint64_t *vals, uint64_t m) {
    __shared__ uint64_t lds[4];
    uint64_t carry = 0;
    for (uint64_t base = 0; base < m; base += SCAN_THREADS) {
        uint64_t idx = base + threadIdx.x;
        uint64_t v = idx < m ? vals[idx] : 0;
        uint64_t total;
        uint64_t ex = block_exclusive_sum(v, &total, lds);
        if (idx < m) vals[idx] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) vals[m] = carry;
}

template <typename Tin>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply_kernel(const Tin *in, uint64_t n, const uint64_t *block_offsets,
                                                                  uint64_t *out) {
    __shared__ uint64_t lds[4];
    uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * SCAN_ITEMS;
    Tin v[SCAN_ITEMS];
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        uint64_t idx = base + i;
        v[i] = idx < n ? in[idx] : (Tin)0;
        s += v[i];
    }
    uint64_t total;
    uint64_t ex = block_exclusive_sum(s, &total, lds) + block_offsets[blockIdx.x];
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS; i++) {
        uint64_t idx = base + i;
        if (idx < n) out[idx] = ex;
        ex += v[i];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == SCAN_THREADS - 1) out[n] = ex;
}

template <typename Tin>
static int exclusive_scan_impl(mdbg_ctx *ctx, const Tin *d_in, uint64_t *d_out, uint64_t n) {
    if (n == 0) {
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(d_out, 0, sizeof(uint64_t), ctx->stream));
        return MDBG_OK;
    }
    uint64_t nblocks = (n + SCAN_TILE - 1) / SCAN_TILE;
    DevBuf<uint64_t> sums;
    MDBG_TRY(sums.alloc(ctx, nblocks + 1));
    hipLaunchKernelGGL(scan_reduce_kernel<Tin>, dim3((unsigned)nblocks), dim3(SCAN_THREADS), 0, ctx->stream, d_in, n, sums.p);
    if (nblocks <= 64 * 1024) {
        hipLaunchKernelGGL(scan_small_kernel, dim3(1), dim3(SCAN_THREADS), 0, ctx->stream, sums.p, nblocks);
    } else {
        // two-level: scan the block sums recursively (u64 input)
        DevBuf<uint64_t> tmp;
        MDBG_TRY(tmp.alloc(ctx, nblocks + 1));
        MDBG_TRY(exclusive_scan_impl<uint64_t>(ctx, sums.p, tmp.p, nblocks));
        MDBG_HIP_CHECK(ctx, hipMemcpyAsync(sums.p, tmp.p, (nblocks + 1) * sizeof(uint64_t), hipMemcpyDeviceToDevice, ctx->stream));
        MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    hipLaunchKernelGGL(scan_apply_kernel<Tin>, dim3((unsigned)nblocks), dim3(SCAN_THREADS), 0, ctx->stream, d_in, n, sums.p, d_out);
    MDBG_HIP_CHECK(ctx, hipGetLastError());
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // sums freed on return
    return MDBG_OK;
}

int exclusive_scan_u32(mdbg_ctx *ctx, const uint32_t *d_in, uint64_t *d_out, uint64_t n) {
    return exclusive_scan_impl<uint32_t>(ctx, d_in, d_out, n);
}

}  // namespace mdbg
