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

// 64 x 64 -> low 64 product by a constant on explicit halves: one 32 x 32 -> 64 product and two low products
// (v_mad_u64_u32 + 2 v_mul_lo_u32 + v_add3_u32): 3 multiplier operations, the minimum, and no register-pair shuffling
// (chaining three v_mad_u64_u32 needs the 32-bit carry in an aligned 64-bit pair: 8 v_mov per hash).  Either form
// issues at the same rate on gfx950 -- every VOP3 integer op costs 4.2 cycles (tools/ubench/valu_rates.hip).
__host__ __device__ __forceinline__ void mul64_halves(uint32_t &l, uint32_t &h, uint32_t clo, uint32_t chi) {
    const uint64_t p = (uint64_t)l * clo;
    const uint32_t nh = (uint32_t)(p >> 32) + l * chi + h * clo;
    l = (uint32_t)p; h = nh;
}

__host__ __device__ __forceinline__ void fmix64_halves(uint32_t &l, uint32_t &h) {
    l ^= h >> 1;                                   // k ^= k >> 33
    mul64_halves(l, h, 0xed558ccdu, 0xff51afd7u);
    l ^= h >> 1;
    mul64_halves(l, h, 0x1a85ec53u, 0xc4ceb9feu);
    l ^= h >> 1;
}

// MurmurHash3_x64_128(&v, 8, 42) -> h1 for v < 2^32.  len = 8, seed = 42: no body block; tail k1 = key;
// h1 = 42 ^ mix(k1); h2 = 42; both ^= 8  =>  h1 = (k1 ^ 34) + 34, h2 = h1 + 34 before the finalisers.
// 17 multiplier operations per hash (2 + 3 + 4 x 3).
__host__ __device__ __forceinline__ uint64_t kmer_hash32(uint32_t v) {
    const uint64_t p = (uint64_t)v * 0x114253d5u;                        // v * c1, 32 x 64
    uint32_t l = (uint32_t)p, h = (uint32_t)(p >> 32) + v * 0x87c37b91u;
    const uint32_t rl = (l << 31) | (h >> 1), rh = (h << 31) | (l >> 1);   // rotl64(k1, 31)
    l = rl; h = rh;
    mul64_halves(l, h, 0x2745937fu, 0x4cf5ad43u);                        // * c2
    const uint64_t h1 = ((((uint64_t)h << 32) | l) ^ 34ull) + 34ull, h2 = h1 + 34ull;
    uint32_t al = (uint32_t)h1, ah = (uint32_t)(h1 >> 32), bl = (uint32_t)h2, bh = (uint32_t)(h2 >> 32);
    fmix64_halves(al, ah);
    fmix64_halves(bl, bh);
    return (((uint64_t)ah << 32) | al) + (((uint64_t)bh << 32) | bl);
}

// The upper 32 bits of kmer_hash32(v) without the carry out of the lower halves: the final k ^= k >> 33 of either finaliser
// touches only the lower half, so hi(h1) = hi(mix(a)) + hi(mix(b)) + carry, carry in {0, 1}.  Four operations per hash less
// than the full value (two shift/xor pairs, the 64-bit add); hi(kmer_hash32(v)) is the result or the result + 1 (mod 2^32).
__host__ __device__ __forceinline__ uint32_t kmer_hash32_hi_nocarry(uint32_t v) {
    const uint64_t p = (uint64_t)v * 0x114253d5u;
    uint32_t l = (uint32_t)p, h = (uint32_t)(p >> 32) + v * 0x87c37b91u;
    const uint32_t rl = (l << 31) | (h >> 1), rh = (h << 31) | (l >> 1);
    l = rl; h = rh;
    mul64_halves(l, h, 0x2745937fu, 0x4cf5ad43u);
    const uint64_t h1 = ((((uint64_t)h << 32) | l) ^ 34ull) + 34ull, h2 = h1 + 34ull;
    uint32_t al = (uint32_t)h1, ah = (uint32_t)(h1 >> 32), bl = (uint32_t)h2, bh = (uint32_t)(h2 >> 32);
    al ^= ah >> 1; mul64_halves(al, ah, 0xed558ccdu, 0xff51afd7u); al ^= ah >> 1; mul64_halves(al, ah, 0x1a85ec53u, 0xc4ceb9feu);
    bl ^= bh >> 1; mul64_halves(bl, bh, 0xed558ccdu, 0xff51afd7u); bl ^= bh >> 1; mul64_halves(bl, bh, 0x1a85ec53u, 0xc4ceb9feu);
    return ah + bh;
}

// The candidate test's value (round 6): the LAST multiplication of either finaliser is linear, so the two can be done as one.
// With a' and b' the finalisers' states before it,  a'*C + b'*C = (a' + b')*C  (mod 2^64), whose upper half is
// hi(mix a) + hi(mix b) + c0, c0 = the carry of the two products' LOWER halves -- not the carry the hash itself has (that one
// comes after each lower half's last xor-shift), but like it 0 or 1.  So with r = upper32((a' + b')*C) + 1:
//     r - hi(kmer_hash32(v)) is 0, 1 or 2 (mod 2^32)   =>   hash < T  implies  r < hi(T) + 3.
// One 64-bit add and three multiplier operations where kmer_hash32_hi_nocarry has six and four adds; the + 1 (which keeps a
// hash whose upper half is 0 from wrapping below zero) rides in the addend of the 32 x 32 -> 64 product.
__host__ __device__ __forceinline__ uint32_t kmer_hash32_hi_merged(uint32_t v) {
    const uint64_t p = (uint64_t)v * 0x114253d5u;
    uint32_t l = (uint32_t)p, h = (uint32_t)(p >> 32) + v * 0x87c37b91u;
    const uint32_t rl = (l << 31) | (h >> 1), rh = (h << 31) | (l >> 1);
    l = rl; h = rh;
    mul64_halves(l, h, 0x2745937fu, 0x4cf5ad43u);
    const uint64_t h1 = ((((uint64_t)h << 32) | l) ^ 34ull) + 34ull, h2 = h1 + 34ull;
    uint32_t al = (uint32_t)h1, ah = (uint32_t)(h1 >> 32), bl = (uint32_t)h2, bh = (uint32_t)(h2 >> 32);
    al ^= ah >> 1; mul64_halves(al, ah, 0xed558ccdu, 0xff51afd7u); al ^= ah >> 1;
    bl ^= bh >> 1; mul64_halves(bl, bh, 0xed558ccdu, 0xff51afd7u); bl ^= bh >> 1;
    const uint64_t s = (((uint64_t)ah << 32) | al) + (((uint64_t)bh << 32) | bl);
    const uint32_t sl = (uint32_t)s, sh = (uint32_t)(s >> 32);
    const uint64_t q = (uint64_t)sl * 0x1a85ec53u + (1ull << 32);
    return (uint32_t)(q >> 32) + sl * 0xc4ceb9feu + sh * 0x1a85ec53u;
}

// Two of them side by side, statement by statement: the compiler keeps the order it is given, and a SIMD that holds 4 - 5 waves
// has an independent instruction to issue while a multiply's result is on its way (tools/ubench/hash_rates.hip: chains=2).
__host__ __device__ __forceinline__ void kmer_hash32_hi_merged_x2(uint32_t va, uint32_t vb, uint32_t &ra, uint32_t &rb) {
    const uint64_t pa = (uint64_t)va * 0x114253d5u, pb = (uint64_t)vb * 0x114253d5u;
    uint32_t la = (uint32_t)pa, lb = (uint32_t)pb;
    uint32_t ha = (uint32_t)(pa >> 32) + va * 0x87c37b91u, hb = (uint32_t)(pb >> 32) + vb * 0x87c37b91u;
    const uint32_t rla = (la << 31) | (ha >> 1), rlb = (lb << 31) | (hb >> 1);
    const uint32_t rha = (ha << 31) | (la >> 1), rhb = (hb << 31) | (lb >> 1);
    la = rla; lb = rlb; ha = rha; hb = rhb;
    const uint64_t qa = (uint64_t)la * 0x2745937fu, qb = (uint64_t)lb * 0x2745937fu;
    const uint32_t nha = (uint32_t)(qa >> 32) + la * 0x4cf5ad43u + ha * 0x2745937fu, nhb = (uint32_t)(qb >> 32) + lb * 0x4cf5ad43u + hb * 0x2745937fu;
    const uint64_t h1a = ((((uint64_t)nha << 32) | (uint32_t)qa) ^ 34ull) + 34ull, h1b = ((((uint64_t)nhb << 32) | (uint32_t)qb) ^ 34ull) + 34ull;
    const uint64_t h2a = h1a + 34ull, h2b = h1b + 34ull;
    uint32_t ala = (uint32_t)h1a, aha = (uint32_t)(h1a >> 32), alb = (uint32_t)h1b, ahb = (uint32_t)(h1b >> 32);
    uint32_t bla = (uint32_t)h2a, bha = (uint32_t)(h2a >> 32), blb = (uint32_t)h2b, bhb = (uint32_t)(h2b >> 32);
    ala ^= aha >> 1; alb ^= ahb >> 1;
    mul64_halves(ala, aha, 0xed558ccdu, 0xff51afd7u); mul64_halves(alb, ahb, 0xed558ccdu, 0xff51afd7u);
    ala ^= aha >> 1; alb ^= ahb >> 1;
    bla ^= bha >> 1; blb ^= bhb >> 1;
    mul64_halves(bla, bha, 0xed558ccdu, 0xff51afd7u); mul64_halves(blb, bhb, 0xed558ccdu, 0xff51afd7u);
    bla ^= bha >> 1; blb ^= bhb >> 1;
    const uint64_t sa = (((uint64_t)aha << 32) | ala) + (((uint64_t)bha << 32) | bla), sb = (((uint64_t)ahb << 32) | alb) + (((uint64_t)bhb << 32) | blb);
    const uint32_t sla = (uint32_t)sa, sha = (uint32_t)(sa >> 32), slb = (uint32_t)sb, shb = (uint32_t)(sb >> 32);
    const uint64_t ta = (uint64_t)sla * 0x1a85ec53u + (1ull << 32), tb = (uint64_t)slb * 0x1a85ec53u + (1ull << 32);
    ra = (uint32_t)(ta >> 32) + sla * 0xc4ceb9feu + sha * 0x1a85ec53u;
    rb = (uint32_t)(tb >> 32) + slb * 0xc4ceb9feu + shb * 0x1a85ec53u;
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
