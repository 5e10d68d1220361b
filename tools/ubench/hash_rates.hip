// hash_rates.hip -- throughput of candidate implementations of the minimizer hash
// (MurmurHash3_x64_128(&v, 8, 42) -> h1, v < 2^32) on gfx950: cycles per 64 hashes per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/hash_rates.hip -o /tmp/hash_rates && /tmp/hash_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include "../../metamdbg_amd/csrc/murmur.hpp"

using namespace mdbg;

// V1: explicit 32-bit limbs, every 64x64->64 product as 3 x v_mad_u64_u32 (no separate adds)
__device__ __forceinline__ uint64_t mul64(uint64_t a, uint32_t clo, uint32_t chi) {
    uint32_t alo = (uint32_t)a, ahi = (uint32_t)(a >> 32);
    uint64_t p = (uint64_t)alo * clo;                               // mad_u64_u32
    uint64_t t = (uint64_t)alo * chi + (uint32_t)(p >> 32);         // mad_u64_u32 (only low word used)
    uint64_t u = (uint64_t)ahi * clo + (uint32_t)t;                 // mad_u64_u32
    return (uint64_t)(uint32_t)p | ((uint64_t)(uint32_t)u << 32);
}
__device__ __forceinline__ uint64_t fmix_v1(uint64_t k) {
    k ^= k >> 33; k = mul64(k, 0xed558ccdu, 0xff51afd7u);
    k ^= k >> 33; k = mul64(k, 0x1a85ec53u, 0xc4ceb9feu);
    k ^= k >> 33; return k;
}
__device__ __forceinline__ uint64_t hash_v1(uint32_t v) {
    uint64_t p = (uint64_t)v * 0x114253d5u;
    uint64_t t = (uint64_t)v * 0x87c37b91u + (uint32_t)(p >> 32);
    uint64_t k1 = (uint64_t)(uint32_t)p | ((uint64_t)(uint32_t)t << 32);
    k1 = rotl64(k1, 31);
    k1 = mul64(k1, 0x2745937fu, 0x4cf5ad43u);
    uint64_t h1 = (k1 ^ 34ull) + 34ull, h2 = h1 + 34ull;
    return fmix_v1(h1) + fmix_v1(h2);
}

// V2: mul_lo / mul_hi only
__device__ __forceinline__ uint64_t mul64_v2(uint64_t a, uint32_t clo, uint32_t chi) {
    uint32_t alo = (uint32_t)a, ahi = (uint32_t)(a >> 32);
    uint32_t lo = alo * clo;
    uint32_t hi = __umulhi(alo, clo) + alo * chi + ahi * clo;
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
__device__ __forceinline__ uint64_t fmix_v2(uint64_t k) {
    k ^= k >> 33; k = mul64_v2(k, 0xed558ccdu, 0xff51afd7u);
    k ^= k >> 33; k = mul64_v2(k, 0x1a85ec53u, 0xc4ceb9feu);
    k ^= k >> 33; return k;
}
__device__ __forceinline__ uint64_t hash_v2(uint32_t v) {
    uint32_t lo = v * 0x114253d5u, hi = __umulhi(v, 0x114253d5u) + v * 0x87c37b91u;
    uint64_t k1 = rotl64((uint64_t)lo | ((uint64_t)hi << 32), 31);
    k1 = mul64_v2(k1, 0x2745937fu, 0x4cf5ad43u);
    uint64_t h1 = (k1 ^ 34ull) + 34ull, h2 = h1 + 34ull;
    return fmix_v2(h1) + fmix_v2(h2);
}

template <int V, int CHAINS>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, int iters) {
    uint32_t v[CHAINS];
    for (int c = 0; c < CHAINS; c++) v[c] = seed * (c + 1) + threadIdx.x * 977u + blockIdx.x;
    uint32_t acc = 0;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int c = 0; c < CHAINS; c++) {
            if (V == 3) {         // the candidate test of scan_fast_kernel<.., APPROX>: upper half without the carry (murmur.hpp)
                const uint32_t s1 = kmer_hash32_hi_nocarry(v[c]) + 1u;
                acc += (s1 < (uint32_t)(92233718306963448ull >> 32) + 2u) ? 1u : 0u;
                v[c] = v[c] * 1664525u + 1013904223u + (s1 >> 28);
                continue;
            }
            if (V == 4) {         // round 6: the finalisers' last multiplications merged into one (kmer_hash32_hi_merged): what the kernel runs now
                const uint32_t s1 = kmer_hash32_hi_merged(v[c]);
                acc += (s1 < (uint32_t)(92233718306963448ull >> 32) + 3u) ? 1u : 0u;
                v[c] = v[c] * 1664525u + 1013904223u + (s1 >> 28);
                continue;
            }
            uint64_t h = V == 0 ? kmer_hash32(v[c]) : (V == 1 ? hash_v1(v[c]) : hash_v2(v[c]));
            acc += (h < 92233718306963448ull) ? 1u : 0u;
            v[c] = v[c] * 1664525u + 1013904223u + (uint32_t)(h >> 60);   // next input (cheap, dependent)
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc + v[0];
}

template <int V, int CHAINS>
void run(const char *name, uint32_t *d, int waves) {
    const int iters = 2048;
    int blocks = 256 * waves;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<V, CHAINS><<<blocks, 256>>>(d, 12345, iters);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<V, CHAINS><<<blocks, 256>>>(d, 12345, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double hashes_per_simd = (double)iters * CHAINS * waves;    // wave-level hashes (64 lanes each) per SIMD
    printf("%-10s chains=%d waves/SIMD=%d  %.3f ms  %.1f ns per 64 hashes per SIMD (= %.0f cycles @2.36GHz)\n", name, CHAINS, waves, ms,
           ms * 1e6 / hashes_per_simd, ms * 1e6 / hashes_per_simd * 2.36);
}

int main() {
    // check the variants agree
    uint32_t *d; (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {4, 5, 8}) {
        run<0, 1>("compiler", d, w); run<0, 2>("compiler", d, w);
        run<1, 1>("mad_u64", d, w);  run<1, 2>("mad_u64", d, w);
        run<2, 1>("mul_lo_hi", d, w); run<2, 2>("mul_lo_hi", d, w);
        run<3, 1>("hi_nocarry", d, w); run<3, 2>("hi_nocarry", d, w);
        run<4, 1>("hi_merged", d, w); run<4, 2>("hi_merged", d, w);
    }
    return 0;
}
