// murmur.hpp -- MurmurHash3 x64-128 on the device, as the reference uses it:
//   * kmer_hash32():  MurmurHash3_x64_128(&kmer, 8, 42) -> h1, the minimizer selection hash
//                     (utils/kmer/Kmer.hpp:1421; utils/MurmurHash3.cpp:246-325), specialised to an
//                     8-byte key whose upper 32 bits are zero (l <= 16 => value < 2^32).
//   * Murmur128Stream: MurmurHash3_x64_128_original(vec, 4k, 0) -> (h1, h2), the k-min-mer identity
//                     (Commons.hpp:941-969; utils/MurmurHash3.cpp:328-405), fed one u32 at a time.
#pragma once
#include <cstdint>

namespace mdbg {

#define MDBG_C1 0x87c37b91114253d5ull
#define MDBG_C2 0x4cf5ad432745937full

__host__ __device__ __forceinline__ uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

__host__ __device__ __forceinline__ uint64_t fmix64(uint64_t k) {
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdull;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ull;
    k ^= k >> 33;
    return k;
}

// 64 x 64 -> low 64 product by a constant, as three v_mad_u64_u32 (the adds ride along): on gfx950
// this sequence issues ~5 % faster at 5 waves/SIMD than the mul_lo / mad / add3 mix the compiler
// picks for a plain u64 multiply (tools/ubench/hash_rates.hip).
__host__ __device__ __forceinline__ uint64_t mul64_const(uint64_t a, uint32_t clo, uint32_t chi) {
    const uint32_t alo = (uint32_t)a, ahi = (uint32_t)(a >> 32);
    const uint64_t p = (uint64_t)alo * clo;
    const uint64_t t = (uint64_t)alo * chi + (uint32_t)(p >> 32);
    const uint64_t u = (uint64_t)ahi * clo + (uint32_t)t;
    return (uint64_t)(uint32_t)p | ((uint64_t)(uint32_t)u << 32);
}

__host__ __device__ __forceinline__ uint64_t fmix64_const(uint64_t k) {
    k ^= k >> 33;
    k = mul64_const(k, 0xed558ccdu, 0xff51afd7u);
    k ^= k >> 33;
    k = mul64_const(k, 0x1a85ec53u, 0xc4ceb9feu);
    k ^= k >> 33;
    return k;
}

// MurmurHash3_x64_128(&v, 8, 42) -> h1 for v < 2^32.  len = 8, seed = 42: no body block; tail k1 = key;
// h1 = 42 ^ mix(k1); h2 = 42; both ^= 8  =>  h1 = (k1 ^ 34) + 34, h2 = h1 + 34 before the finalisers.
__host__ __device__ __forceinline__ uint64_t kmer_hash32(uint32_t v) {
    const uint64_t p = (uint64_t)v * 0x114253d5u;                        // v * c1, 32 x 64
    const uint64_t t = (uint64_t)v * 0x87c37b91u + (uint32_t)(p >> 32);
    uint64_t k1 = (uint64_t)(uint32_t)p | ((uint64_t)(uint32_t)t << 32);
    k1 = rotl64(k1, 31);
    k1 = mul64_const(k1, 0x2745937fu, 0x4cf5ad43u);                      // * c2
    const uint64_t h1 = (k1 ^ 34ull) + 34ull;
    const uint64_t h2 = h1 + 34ull;
    return fmix64_const(h1) + fmix64_const(h2);
}

// Streaming Murmur3 x64-128 over a sequence of u32 words (little-endian), seed 0.
struct Murmur128Stream {
    uint64_t h1 = 0, h2 = 0;
    uint64_t k1 = 0, k2 = 0;
    uint32_t n = 0;  // u32 words consumed

    __host__ __device__ __forceinline__ void push(uint32_t w) {
        switch (n & 3u) {
            case 0: k1 = w; break;
            case 1: k1 |= (uint64_t)w << 32; break;
            case 2: k2 = w; break;
            default:
                k2 |= (uint64_t)w << 32;
                k1 *= MDBG_C1; k1 = rotl64(k1, 31); k1 *= MDBG_C2; h1 ^= k1;
                h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
                k2 *= MDBG_C2; k2 = rotl64(k2, 33); k2 *= MDBG_C1; h2 ^= k2;
                h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
                k1 = 0; k2 = 0;
                break;
        }
        n++;
    }

    // out_hi = out[0] (h1), out_lo = out[1] (h2): u128 = (h1 << 64) + h2 (Commons.hpp:958-960)
    __host__ __device__ __forceinline__ void finish(uint64_t &out_hi, uint64_t &out_lo) {
        uint32_t rem = n & 3u;  // whole u32 words in the tail: 0..3 -> 0,4,8,12 bytes
        if (rem == 3) { k2 *= MDBG_C2; k2 = rotl64(k2, 33); k2 *= MDBG_C1; h2 ^= k2; }
        if (rem >= 1) { k1 *= MDBG_C1; k1 = rotl64(k1, 31); k1 *= MDBG_C2; h1 ^= k1; }
        uint64_t len = (uint64_t)n * 4;
        h1 ^= len; h2 ^= len;
        h1 += h2; h2 += h1;
        h1 = fmix64(h1); h2 = fmix64(h2);
        h1 += h2; h2 += h1;
        out_hi = h1; out_lo = h2;
    }
};

}  // namespace mdbg
