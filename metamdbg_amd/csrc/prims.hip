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

// A block owns a tile of SCAN_TILE items and walks it in SCAN_SUB sub-tiles of 4 items per thread, so that every
// load is one 16-byte (u32 input) access per lane and every store two 16-byte accesses per lane: fully coalesced.
constexpr int SCAN_VEC = 4;
constexpr int SCAN_SUB = SCAN_ITEMS / SCAN_VEC;

template <typename Tin>
__device__ __forceinline__ void load_vec(const Tin *in, uint64_t idx, uint64_t n, uint64_t v[SCAN_VEC]) {
    if (idx + SCAN_VEC <= n && (reinterpret_cast<uintptr_t>(in + idx) & (sizeof(Tin) * SCAN_VEC - 1)) == 0) {
        if (sizeof(Tin) == 4) {
            const uint4 q = *reinterpret_cast<const uint4 *>(in + idx);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        } else {
            const ulonglong2 q0 = *reinterpret_cast<const ulonglong2 *>(in + idx), q1 = *reinterpret_cast<const ulonglong2 *>(in + idx + 2);
            v[0] = q0.x; v[1] = q0.y; v[2] = q1.x; v[3] = q1.y;
        }
    } else {
#pragma unroll
        for (int i = 0; i < SCAN_VEC; i++) v[i] = idx + i < n ? (uint64_t)in[idx + i] : 0ull;
    }
}

template <typename Tin>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const Tin *in, uint64_t n, uint64_t *block_sums) {
    __shared__ uint64_t lds[4];
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    uint64_t s = 0;
#pragma unroll
    for (int t = 0; t < SCAN_SUB; t++) {
        uint64_t v[SCAN_VEC];
        load_vec(in, base + ((uint64_t)t * SCAN_THREADS + threadIdx.x) * SCAN_VEC, n, v);
        s += v[0] + v[1] + v[2] + v[3];
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
    const uint64_t base = (uint64_t)blockIdx.x * SCAN_TILE;
    uint64_t carry = block_offsets[blockIdx.x];
#pragma unroll
    for (int t = 0; t < SCAN_SUB; t++) {
        const uint64_t idx = base + ((uint64_t)t * SCAN_THREADS + threadIdx.x) * SCAN_VEC;
        uint64_t v[SCAN_VEC];
        load_vec(in, idx, n, v);
        uint64_t total;
        uint64_t ex = block_exclusive_sum(v[0] + v[1] + v[2] + v[3], &total, lds) + carry;
        if (idx + SCAN_VEC <= n) {                       // out is 8-byte aligned and idx a multiple of 4: 32-byte aligned stores
            ulonglong2 o0, o1;
            o0.x = ex; o0.y = ex + v[0]; o1.x = ex + v[0] + v[1]; o1.y = ex + v[0] + v[1] + v[2];
            if ((reinterpret_cast<uintptr_t>(out) & 15) == 0) {
                *reinterpret_cast<ulonglong2 *>(out + idx) = o0;
                *reinterpret_cast<ulonglong2 *>(out + idx + 2) = o1;
            } else { out[idx] = o0.x; out[idx + 1] = o0.y; out[idx + 2] = o1.x; out[idx + 3] = o1.y; }
        } else {
            uint64_t e = ex;
#pragma unroll
            for (int i = 0; i < SCAN_VEC; i++) { if (idx + i < n) out[idx + i] = e; e += v[i]; }
        }
        carry += total;
    }
    // the total: written by the block that owns item n-1, once its last sub-tile is done
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = carry;
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
    LaunchTimer timer(ctx, "prefix_scan");
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
    // no synchronisation: `sums` returns to the context's pool, whose blocks are only ever handed to later work on
    // the same stream, i.e. after the kernels above
    return MDBG_OK;
}

int exclusive_scan_u32(mdbg_ctx *ctx, const uint32_t *d_in, uint64_t *d_out, uint64_t n) {
    return exclusive_scan_impl<uint32_t>(ctx, d_in, d_out, n);
}

}  // namespace mdbg
