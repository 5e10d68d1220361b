// partition.hip -- the first pass (k = firstK) with the k-min-mer instances partitioned by key and counted in LDS.
//
// The reference counts k-min-mers by writing every instance to one of P partition files chosen by `vecHash % P`, then sorting
// each partition and run-length counting it (KminmerCounter::partitionKminmer / dereplicatePartition,
// graph/CreateMdbg.hpp:3714-3851; P from graph/CreateMdbg.cpp:222-225).  kminmer.hip replaced that by ONE open-addressing table
// in HBM: a device-scope atomic and a random 64-byte sector per instance, 6.3 x the algorithmic traffic at 10 M reads
//  and a table that must fit 2^31 slots.  This file is the partitioned design on the device:
//
//   1. mark_starts_kernel      one bit per minimizer: "a sequence starts here" -- a window of k minimizers is an instance iff no
//                              start lies inside it, so the passes below walk the FLAT minimizer array, fully coalesced
//   2. split (hist + scatter)  a radix multisplit of instance records {hash_lo, hash_hi, rep} (rep = flat index of the window's
//                              first minimizer), <= 256 ways per level, one to three levels.  Level 1 makes the records from the
//                              minimizers and splits them by a cheap symmetric mix of the window (window_mix: the histogram pass
//                              need not compute the identity); deeper levels split every bucket of the level above by bits of hash_lo.
//                              Per-block LDS histograms + one device-wide exclusive scan give every (block, digit) its place:
//                              no global atomic, and a block writes one growing run per digit (tiles are regrouped by digit in
//                              LDS first, so the stores are runs of TILE / ways records)
//   3. bucket_count_kernel     one workgroup per final bucket: its records stream through an LDS hash table (LDS atomics: claim
//                              by CAS on the low word, publish the high word, count), the bucket's solid keys are written once,
//                              compacted, and the count of every instance whose key is rare (<= 2 m* + 1) is scattered to a byte
//                              per instance (cnt8[rep]) -- all the rescue pass (graph/CreateMdbg.hpp:4514-4640) needs
//   4. emit / rescue           rows of the solid keys bucket by bucket (vector read back through rep); the rescue decision per
//                              read from cnt8 (the median test only needs exact counts up to 2 m* + 1, see rescue_count_p_kernel)
//
// Keys are processed in G sequential groups (the top bits of hash_lo) when the records of all of them would not fit the budget:
// the reference's _nbPartitions.  Nothing here is limited by a table size; what remains is the 32-bit `rep` (2^32 minimizers
// per call, checked by the caller).  A bucket that holds more distinct keys than its LDS table makes the pass repeat with four
// times the buckets; if that is not enough the caller's one-table path takes over.
#include "common.hpp"
#include "kminmer_dev.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

namespace mdbg {

#ifndef MDBG_PART_NT
#define MDBG_PART_NT 512
#endif
#ifndef MDBG_PART_E
#define MDBG_PART_E 8
#endif
constexpr uint32_t PART_NT = MDBG_PART_NT;             // threads per block of the split kernels
constexpr uint32_t PART_E = MDBG_PART_E;               // records per thread and tile
constexpr uint32_t PART_TILE = PART_NT * PART_E;       // 4096 records regrouped in LDS at a time
constexpr uint32_t PART_MAXW = 256;                    // ways per level
// bucket_count remembers the LDS slot of a bucket's first records between its two passes: 8000 of them beside 1024 slots (not 8192:
// four such buckets then fit a CU's 160 KB), 4096 beside 2048 (the 64 KB of static LDS), none when the kernel is to fit beside a scan
constexpr uint32_t PART_MAX_K = 32;                    // window validity is one 64-bit extract of the start bits

struct RecView { unsigned long long *lo, *hi; uint32_t *rep; };

struct SplitArgs {
    // level 1: records are computed from the flat minimizer array
    const uint32_t *mins; const uint32_t *start_bits; uint64_t n_min; uint32_t k;
    uint32_t group_bits; uint64_t group;               // only keys with hash_lo >> (64 - group_bits) == group (0 bits: all)
    // deeper levels: records are read back
    RecView in;
    // segments of the input that are split independently: segment s = [seg_pos[s * stride], seg_pos[(s + 1) * stride])
    // (the scanned histogram of the level above); null: ONE segment, the n_min flat positions
    const uint64_t *seg_pos; uint32_t seg_stride;
    uint32_t blocks_per_seg;
    uint32_t tile;                                     // a block's share of a segment is a multiple of this (the scatter's tile)
    uint32_t shift, ways;                              // digit = (hash_lo >> shift) & (ways - 1); level 1: the top bits of window_mix
    uint32_t mix_bits;                                 // level 1: log2(ways) (digit = window_mix >> (64 - mix_bits); 0 bits: one way)
    uint8_t *hll;                                      // level-1 histogram: HLL_M registers per block, see hll_estimate (null: not wanted)
};

// How many distinct keys are there?  The plan (buckets so that a bucket's keys fit its LDS table) needs to know before the first
// record is written, and a first pass is usually the only one its process ever runs: nothing to learn from.  The level-1 histogram
// mixes every instance anyway (window_mix), so it keeps a HyperLogLog sketch on the side (Flajolet et al. 2007: register = 11 bits of
// the mix, value = position of the first set bit of 56 others; one LDS atomic max per instance): 2048 registers, standard error 2.3 %.
constexpr uint32_t HLL_P = 11, HLL_M = 1u << HLL_P;

__global__ __launch_bounds__(256) void mark_starts_kernel(const uint64_t *off, uint32_t n_reads, uint32_t *bits) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_reads) return;                           // off[n_reads] = n_min: the end is a start, too
    const uint64_t p = off[r];
    if (r < n_reads && off[r + 1] == p) return;        // an empty sequence adds nothing its successor does not
    atomicOr(&bits[p >> 5], 1u << (p & 31u));
}

// is there an instance at flat position p?
__device__ __forceinline__ bool window_valid(const SplitArgs &a, uint64_t p) {
    if (p + a.k > a.n_min) return false;
    if (a.k > 1) {
        const uint64_t q = p + 1;
        const uint32_t w = (uint32_t)(q >> 5), sh = (uint32_t)(q & 31u);
        const uint64_t x = (((uint64_t)a.start_bits[w + 1] << 32) | a.start_bits[w]) >> sh;   // bits q .. q + 32 at least
        if (x & ((1ull << (a.k - 1)) - 1ull)) return false;
    }
    return true;
}

// ... if so its identity.
__device__ __forceinline__ bool window_at(const SplitArgs &a, uint64_t p, uint64_t &hi, uint64_t &lo) {
    if (!window_valid(a, p)) return false;
    window_hash_uniform(a.mins + p, a.k, hi, lo);
    return true;
}

__device__ __forceinline__ bool in_group(const SplitArgs &a, uint64_t lo) {
    return !a.group_bits || (lo >> (64u - a.group_bits)) == a.group;
}

// Level 1 does not split by bits of the identity: all it needs is SOME function of the key, the same in the histogram and in the
// scatter, and the identity costs a hundred vector instructions per window (Murmur3 x64-128) that the histogram pass would spend a
// second time.  This one is a fifth of that: the window read from both ends, every pair (m[i], m[k-1-i]) folded in through its sum
// and its xor -- both symmetric, so a window and its reverse (one key: KmerVec::normalize, Commons.hpp:886-916) mix alike without
// being oriented first -- into one 64-bit multiply-xorshift chain.  Its top bits are the level-1 digit, the sketch (hll_estimate)
// takes register and rank from it as well.  Deeper levels and the buckets' tables use the identity's own bits.
__device__ __forceinline__ uint64_t window_mix(const uint32_t *m, uint32_t k) {
    uint64_t acc = 0x9E3779B97F4A7C15ull ^ k;
    for (uint32_t i = 0; i < k / 2; i++) {
        const uint32_t x = m[i], y = m[k - 1 - i];
        acc = (acc ^ ((uint64_t)(x + y) | ((uint64_t)(x ^ y) << 32))) * 0xff51afd7ed558ccdull;
        acc ^= acc >> 29;
    }
    if (k & 1u) { acc = (acc ^ m[k / 2]) * 0xc4ceb9fe1a85ec53ull; acc ^= acc >> 29; }
    acc *= 0xc4ceb9fe1a85ec53ull;
    return acc ^ (acc >> 32);
}

__device__ __forceinline__ void split_range(const SplitArgs &a, uint32_t &seg, uint32_t &j, uint64_t &b, uint64_t &e) {
    seg = blockIdx.x / a.blocks_per_seg; j = blockIdx.x % a.blocks_per_seg;
    uint64_t s0 = 0, s1 = a.n_min;
    if (a.seg_pos) { s0 = a.seg_pos[(uint64_t)seg * a.seg_stride]; s1 = a.seg_pos[(uint64_t)(seg + 1) * a.seg_stride]; }
    const uint64_t len = s1 - s0;
    uint64_t per = (len + a.blocks_per_seg - 1) / a.blocks_per_seg;
    per = (per + a.tile - 1) / a.tile * a.tile;
    b = s0 + (uint64_t)j * per; if (b > s1) b = s1;
    e = b + per; if (e > s1) e = s1;
}

// hist[(seg * ways + digit) * blocks_per_seg + j]: scanned in this order the table is every (block, digit)'s first output place,
// and entry (seg * ways + digit) * blocks_per_seg the start of bucket seg * ways + digit
template <bool FROM_MINS>
__global__ __launch_bounds__(PART_NT) void split_hist_kernel(SplitArgs a, uint32_t *hist) {
    __shared__ uint32_t h[PART_MAXW];
    __shared__ uint32_t sketch[FROM_MINS ? HLL_M : 1];
    const bool sketching = FROM_MINS && a.hll != nullptr;
    for (uint32_t t = threadIdx.x; t < a.ways; t += PART_NT) h[t] = 0;
    if (sketching) for (uint32_t t = threadIdx.x; t < HLL_M; t += PART_NT) sketch[t] = 0;
    __syncthreads();
    uint32_t seg, j; uint64_t b, e;
    split_range(a, seg, j, b, e);
    // PART_E records of a thread in flight at a time: one position per trip left the loads of its window (or of its record) waiting
    // for each other, 1.7 ms for a pass whose hashing is 0.7 ms of vector instructions
    for (uint64_t t0 = b; t0 < e; t0 += PART_TILE) {
        uint32_t digit[PART_E];
#pragma unroll
        for (uint32_t q = 0; q < PART_E; q++) {
            const uint64_t i = t0 + (uint64_t)q * PART_NT + threadIdx.x;
            digit[q] = 0xFFFFFFFFu;
            if (i < e) {
                uint64_t lo, hi;
                bool valid = true;
                if (FROM_MINS) {
                    valid = window_valid(a, i);
                    if (valid) {
                        const uint64_t x = window_mix(a.mins + i, a.k);
                        // (register from the low half, rank from the high one: the digit takes the top bits)
                        if (sketching) atomicMax(&sketch[(uint32_t)x & (HLL_M - 1u)], (uint32_t)__clzll((long long)((x << 8) | 1ull)) + 1u);
                        if (a.group_bits) { window_hash_uniform(a.mins + i, a.k, hi, lo); valid = in_group(a, lo); }   // key groups: by the identity
                        if (valid) digit[q] = a.mix_bits ? (uint32_t)(x >> (64u - a.mix_bits)) : 0u;
                    }
                } else {
                    lo = a.in.lo[i];
                    digit[q] = (uint32_t)(lo >> a.shift) & (a.ways - 1u);
                }
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < PART_E; q++) if (digit[q] != 0xFFFFFFFFu) atomicAdd(&h[digit[q]], 1u);
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < a.ways; t += PART_NT) hist[((uint64_t)seg * a.ways + t) * a.blocks_per_seg + j] = h[t];
    if (sketching) for (uint32_t t = threadIdx.x; t < HLL_M; t += PART_NT) a.hll[(uint64_t)blockIdx.x * HLL_M + t] = (uint8_t)sketch[t];
}

// (64 slices of the blocks per register, merged by atomic max: one thread per register walking all 2048 blocks took 0.8 ms)
__global__ __launch_bounds__(256) void hll_merge_kernel(const uint8_t *per_block, uint32_t n_blocks, uint32_t *merged) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= HLL_M) return;
    uint32_t m = 0;
    for (uint32_t b = blockIdx.y; b < n_blocks; b += gridDim.y) { const uint32_t v = per_block[(uint64_t)b * HLL_M + r]; m = v > m ? v : m; }
    if (m) atomicMax(&merged[r], m);
}

// E records per thread and tile: 8 (tiles of 4096: runs of 16 records per digit and tile at 256 ways) when the kernel has the device to
// itself, 4 (24 KB of LDS instead of 43) when it is to fit beside another context's scan blocks (mdbg_set_option "partition_tile")
template <bool FROM_MINS, uint32_t E>
__global__ __launch_bounds__(PART_NT) void split_scatter_kernel(SplitArgs a, const uint64_t *place, RecView out) {
    constexpr uint32_t TILE = PART_NT * E;
    __shared__ unsigned long long stage[TILE];         // one field of the tile's records at a time, grouped by digit
    __shared__ uint8_t stage_digit[TILE];
    __shared__ uint32_t h[PART_MAXW], wsum[4];
    __shared__ uint16_t loff[PART_MAXW];
    __shared__ unsigned long long cur[PART_MAXW], gbase[PART_MAXW];
    uint32_t seg, j; uint64_t b, e;
    split_range(a, seg, j, b, e);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    for (uint32_t t = tid; t < PART_MAXW; t += PART_NT) {
        h[t] = 0;
        cur[t] = t < a.ways ? place[((uint64_t)seg * a.ways + t) * a.blocks_per_seg + j] : 0ull;
    }
    __syncthreads();
    for (uint64_t t0 = b; t0 < e; t0 += TILE) {
        uint64_t lo[E], hi[E];
        uint32_t rep[E], meta[E];            // digit | rank << 8 | valid << 31
#pragma unroll
        for (uint32_t q = 0; q < E; q++) {
            const uint64_t i = t0 + (uint64_t)q * PART_NT + tid;
            bool valid = i < e;
            if (valid) {
                if (FROM_MINS) { valid = window_at(a, i, hi[q], lo[q]) && in_group(a, lo[q]); rep[q] = (uint32_t)i; }
                else { lo[q] = a.in.lo[i]; hi[q] = a.in.hi[i]; rep[q] = a.in.rep[i]; }
            }
            meta[q] = 0;
            if (valid) {
                const uint32_t d = FROM_MINS ? (a.mix_bits ? (uint32_t)(window_mix(a.mins + i, a.k) >> (64u - a.mix_bits)) : 0u)
                                             : ((uint32_t)(lo[q] >> a.shift) & (a.ways - 1u));
                meta[q] = d | (atomicAdd(&h[d], 1u) << 8) | 0x80000000u;
            }
        }
        __syncthreads();
        // exclusive scan of the tile's histogram (ways <= 256: the first four waves; the others carry zeros)
        const uint32_t v = tid < a.ways ? h[tid] : 0u;
        const uint32_t inc = wave_inclusive_sum_dpp(v);
        if (lane == 63u && wave < 4u) wsum[wave] = inc;
        __syncthreads();
        const uint32_t w0 = wsum[0], w1 = wsum[1], w2 = wsum[2], w3 = wsum[3];
        const uint32_t n_tile = w0 + w1 + w2 + w3;
        if (tid < a.ways) {
            const uint32_t excl = (wave > 0 ? w0 : 0u) + (wave > 1 ? w1 : 0u) + (wave > 2 ? w2 : 0u) + inc - v;
            loff[tid] = (uint16_t)excl;
            gbase[tid] = cur[tid] - excl;              // record at staged place j of this digit goes to gbase + j
            cur[tid] += v;
            h[tid] = 0;
        }
        __syncthreads();
        uint32_t dst[E];
#pragma unroll
        for (uint32_t q = 0; q < E; q++) {
            dst[q] = 0xFFFFFFFFu;
            if (meta[q] & 0x80000000u) {
                const uint32_t d = meta[q] & 0xFFu;
                dst[q] = (uint32_t)loff[d] + ((meta[q] >> 8) & 0x7FFFFFu);
                stage[dst[q]] = lo[q];
                stage_digit[dst[q]] = (uint8_t)d;
            }
        }
        __syncthreads();
#if !defined(MDBG_PART_ABLATE)
        for (uint32_t s = tid; s < n_tile; s += PART_NT) out.lo[gbase[stage_digit[s]] + s] = stage[s];
#elif MDBG_PART_ABLATE == 1
        for (uint32_t s = tid; s < n_tile; s += PART_NT) if (stage[s] == 0x123456789ull) out.lo[gbase[stage_digit[s]] + s] = stage[s];   // ablation: LDS work, no stores
#elif MDBG_PART_ABLATE == 3
        for (uint32_t s = tid; s < n_tile; s += PART_NT) out.lo[t0 - b + (place[0] & 1) + s] = stage[s];                                // ablation: stores in input order
#endif
        __syncthreads();
#pragma unroll
        for (uint32_t q = 0; q < E; q++) if (dst[q] != 0xFFFFFFFFu) stage[dst[q]] = hi[q];
        __syncthreads();
#if !defined(MDBG_PART_ABLATE)
        for (uint32_t s = tid; s < n_tile; s += PART_NT) out.hi[gbase[stage_digit[s]] + s] = stage[s];
#elif MDBG_PART_ABLATE == 1
        for (uint32_t s = tid; s < n_tile; s += PART_NT) if (stage[s] == 0x123456789ull) out.hi[gbase[stage_digit[s]] + s] = stage[s];
#elif MDBG_PART_ABLATE == 3
        for (uint32_t s = tid; s < n_tile; s += PART_NT) out.hi[t0 - b + (place[0] & 1) + s] = stage[s];
#endif
        __syncthreads();
        uint32_t *stage32 = reinterpret_cast<uint32_t *>(stage);
#pragma unroll
        for (uint32_t q = 0; q < E; q++) if (dst[q] != 0xFFFFFFFFu) stage32[dst[q]] = rep[q];
        __syncthreads();
#if !defined(MDBG_PART_ABLATE)
        for (uint32_t s = tid; s < n_tile; s += PART_NT) out.rep[gbase[stage_digit[s]] + s] = stage32[s];
#elif MDBG_PART_ABLATE == 1
        for (uint32_t s = tid; s < n_tile; s += PART_NT) if (stage32[s] == 0x12345678u) out.rep[gbase[stage_digit[s]] + s] = stage32[s];
#elif MDBG_PART_ABLATE == 3
        for (uint32_t s = tid; s < n_tile; s += PART_NT) out.rep[t0 - b + (place[0] & 1) + s] = stage32[s];
#endif
        __syncthreads();
    }
}

// ---- counting one bucket in LDS ---------------------------------------------------------------------------------------------
constexpr uint32_t LDS_NONE = 0xFFFFu;
constexpr uint32_t PART_OVERFLOW_TABLE = 1u, PART_OVERFLOW_ZERO_KEY = 2u;      // what a void counting pass says went wrong

template <uint32_t C>
struct LdsTable {
    unsigned long long lo[C], hi[C];                   // 0 = empty / not yet published (a key with a zero word: the pass gives up)
    uint32_t cnt[C], rep[C];
};

// find or create the slot of (lo, hi) and count one occurrence; LDS_NONE when the table is full
template <uint32_t C>
__device__ __forceinline__ uint32_t lds_upsert(LdsTable<C> &t, uint64_t lo, uint64_t hi, uint32_t rep) {
    uint32_t s = (uint32_t)lo & (C - 1u);
    for (uint32_t probes = 0; probes < C; probes++, s = (s + 1u) & (C - 1u)) {
        unsigned long long cur = __hip_atomic_load(&t.lo[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == 0ull) {
            cur = atomicCAS(&t.lo[s], 0ull, (unsigned long long)lo);
            if (cur == 0ull) cur = lo;
        }
        if (cur != lo) continue;
        unsigned long long h = __hip_atomic_load(&t.hi[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (h == 0ull) {
            h = atomicCAS(&t.hi[s], 0ull, (unsigned long long)hi);
            if (h == 0ull) { h = hi; t.rep[s] = rep; }  // whoever publishes the key represents it
        }
        if (h == hi) { atomicAdd(&t.cnt[s], 1u); return s; }
    }
    return LDS_NONE;
}

template <uint32_t C>
__device__ __forceinline__ uint32_t lds_find(const LdsTable<C> &t, uint64_t lo, uint64_t hi) {
    uint32_t s = (uint32_t)lo & (C - 1u);
    for (uint32_t probes = 0; probes < C; probes++, s = (s + 1u) & (C - 1u)) {
        const unsigned long long cur = t.lo[s];
        if (cur == 0ull) return LDS_NONE;
        if (cur == lo && t.hi[s] == hi) return s;
    }
    return LDS_NONE;
}

// One workgroup per bucket.  keys: the bucket's kept keys (count > 1 and >= min_abundance; keep_all: every key), written to the
// first places of the bucket's own range in the record buffer that is free by now, with their counts in kcnt; n_keys / n_kept
// per bucket; cnt8[rep] = count for every instance of a key counted at most `clip` times, 0 for the others (cnt8 null: not wanted;
// the array arrives filled with cnt8_default -- 0 where most instances belong to frequent keys, 1 where most keys are singletons --
// and only the instances that differ are stored: random one-byte stores are what this costs).
constexpr uint32_t BC_NT = 512;                        // threads per bucket: 4 blocks of 40 KB fill a CU's 32 wave slots
constexpr uint32_t BC_U = 4;                           // records of a thread in flight

template <uint32_t C, uint32_t LIST>
__global__ __launch_bounds__(BC_NT) void bucket_count_kernel(RecView in, const uint64_t *pos, uint32_t stride, uint32_t min_abundance, uint32_t clip,
                                                           int keep_all, RecView keys, uint32_t *kcnt, uint32_t *n_keys, uint32_t *n_kept,
                                                           uint8_t *cnt8, uint32_t cnt8_default, uint32_t *overflow) {
    __shared__ LdsTable<C> t;
    constexpr uint32_t PART_SLOTLIST = LIST;           // 0: none (24 bytes of LDS per slot and nothing else: fits beside a scan)
    __shared__ uint16_t slot_of[LIST ? LIST : 1];
    __shared__ uint32_t kept, occupied;
    const uint32_t tid = threadIdx.x;
    const uint64_t s0 = pos[(uint64_t)blockIdx.x * stride], s1 = pos[(uint64_t)(blockIdx.x + 1) * stride];
    const uint64_t n = s1 - s0;
    for (uint32_t s = tid; s < C; s += BC_NT) { t.lo[s] = 0ull; t.hi[s] = 0ull; t.cnt[s] = 0u; }
    if (tid == 0) { kept = 0; occupied = 0; }
    __syncthreads();
    // a pass in which some bucket outgrew its table is void and will be repeated: the rest of it is not worth running (a full table
    // costs every further record a walk over all its slots: 115 ms once for a pass of 4)
    __shared__ uint32_t give_up;
    if (tid == 0) give_up = __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (give_up) return;
    bool failed = false, zero_key = false;
    for (uint64_t i0 = 0; i0 < n; i0 += BC_NT * BC_U) {
        if (__hip_atomic_load(&give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) break;
        uint64_t lo[BC_U], hi[BC_U]; uint32_t rep[BC_U];
#pragma unroll
        for (uint32_t q = 0; q < BC_U; q++) {
            const uint64_t i = i0 + (uint64_t)q * BC_NT + tid;
            if (i < n) { lo[q] = in.lo[s0 + i]; hi[q] = in.hi[s0 + i]; rep[q] = in.rep[s0 + i]; }
        }
#pragma unroll
        for (uint32_t q = 0; q < BC_U; q++) {
            const uint64_t i = i0 + (uint64_t)q * BC_NT + tid;
            if (i >= n) continue;
            uint32_t s = LDS_NONE;
            const bool zero_word = lo[q] == 0ull || hi[q] == 0ull;       // 0 marks an empty slot: such a key (2^-63) is not for this pass at all
            if (!zero_word) s = lds_upsert<C>(t, lo[q], hi[q], rep[q]);
            if (s == LDS_NONE) { failed = true; zero_key = zero_key || zero_word; __hip_atomic_store(&give_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
            if (i < PART_SLOTLIST) slot_of[i] = (uint16_t)s;
        }
    }
    // void pass.  PART_OVERFLOW_TABLE: a bucket outgrew its table -- the host repeats the pass with more buckets; PART_OVERFLOW_ZERO_KEY: no
    // number of buckets helps -- the host hands the input to the one-table pass at once (round-4 ADVICE: it used to repeat four times first)
    if (failed) atomicMax(overflow, zero_key ? PART_OVERFLOW_ZERO_KEY : PART_OVERFLOW_TABLE);
    __syncthreads();
    const bool stop = give_up != 0u;                   // (read once, behind the barrier: every wave takes the same way out)
    if (stop) return;
    uint32_t my_occ = 0;
    for (uint32_t s = tid; s < C; s += BC_NT) {
        const unsigned long long lo = t.lo[s];
        if (lo == 0ull) continue;
        my_occ++;
        const uint32_t c = t.cnt[s];
        if (keep_all || (c > 1u && !(c < min_abundance))) {
            const uint64_t d = s0 + atomicAdd(&kept, 1u);
            keys.lo[d] = lo; keys.hi[d] = t.hi[s]; keys.rep[d] = t.rep[s]; kcnt[d] = c;
        }
    }
    if (my_occ) atomicAdd(&occupied, my_occ);
    if (cnt8) {
        for (uint64_t i0 = 0; i0 < n; i0 += BC_NT * BC_U) {
            uint32_t rep[BC_U], sl[BC_U];
#pragma unroll
            for (uint32_t q = 0; q < BC_U; q++) {
                const uint64_t i = i0 + (uint64_t)q * BC_NT + tid;
                sl[q] = LDS_NONE;
                if (i < n) {
                    rep[q] = in.rep[s0 + i];
                    sl[q] = i < PART_SLOTLIST ? (uint32_t)slot_of[i] : lds_find<C>(t, in.lo[s0 + i], in.hi[s0 + i]);
                }
            }
#pragma unroll
            for (uint32_t q = 0; q < BC_U; q++) {
                if (sl[q] == LDS_NONE) continue;
                const uint32_t c = t.cnt[sl[q]], v = c <= clip ? c : 0u;
                if (v != cnt8_default) cnt8[rep[q]] = (uint8_t)v;     // the array was filled with the value most instances have
            }
        }
    }
    __syncthreads();
    if (tid == 0) { n_keys[blockIdx.x] = occupied; n_kept[blockIdx.x] = kept; }
}

// the canonical vector of the window at m into row `row` (k = 4: one 16-byte store instead of four strided ones)
__device__ __forceinline__ void store_vector(const uint32_t *m, uint32_t k, bool reversed, uint32_t *vec, uint64_t row) {
    if (k == 4u) {
        uint4 v;
        v.x = reversed ? m[3] : m[0]; v.y = reversed ? m[2] : m[1]; v.z = reversed ? m[1] : m[2]; v.w = reversed ? m[0] : m[3];
        *reinterpret_cast<uint4 *>(vec + row * 4u) = v;          // the row arrays come from the pool: 256-byte aligned
        return;
    }
    for (uint32_t q = 0; q < k; q++) vec[row * k + q] = reversed ? m[k - 1 - q] : m[q];
}

// rows of the kept keys, bucket by bucket
__global__ __launch_bounds__(256) void emit_bucket_rows_kernel(RecView keys, const uint32_t *kcnt, const uint64_t *pos, uint32_t stride,
                                                               const uint32_t *n_kept, const uint64_t *row_of, const uint32_t *mins, RowOut o, uint64_t row_base) {
    const uint64_t s0 = pos[(uint64_t)blockIdx.x * stride];
    const uint32_t n = n_kept[blockIdx.x];
    const uint64_t r0 = row_base + row_of[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint64_t row = r0 + i;
        o.lo[row] = keys.lo[s0 + i]; o.hi[row] = keys.hi[s0 + i]; o.ab[row] = kcnt[s0 + i];
        if (!o.vec) continue;
        const uint32_t *m = mins + keys.rep[s0 + i];           // the canonical vector of the instance that represents the key
        bool reversed = true;
        for (uint32_t q = 0; q < o.k; q++) {
            const uint32_t x = m[q], y = m[o.k - 1 - q];
            if (x == y) continue;
            reversed = !(x < y);
            break;
        }
        store_vector(m, o.k, reversed, o.vec, row);
    }
}

// ---- rescue from the per-instance counts (graph/CreateMdbg.hpp:4514-4640) -----------------------------------------------------
// cnt8[p] of the instance that starts at flat position p: its key's count when that is <= clip = 2 m* + 1, else 0 ("many").
// The decision per read (see rescue_count_kernel in kminmer.hip for the derivation): "median * 0.1f > 1" is false iff at least
// n/2 + 1 counts are <= m*, or -- n even -- exactly n/2 are and (max{<= m*} + min{> m*}) / 2 <= m*.  A count above 2 m* + 1 in
// the second term gives a median above m* whatever the first is, so "many" needs no value.
// One lane per read, its windows' bytes four at a time: the sixteen-lanes-a-read form of the one-table pass spent 270 M vector
// instructions on cross-lane sums for 10 M reads (rocprofv3, round 4) -- instructions another batch's scan would have used.
__global__ __launch_bounds__(256) void rescue_count_p_kernel(const uint64_t *off, uint32_t n_reads, uint32_t k, const uint8_t *cnt8,
                                                             uint32_t m_star, uint32_t *resc_cnt) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_reads) return;
    const uint64_t f = off[r];
    const uint64_t len = off[r + 1] - f;
    const uint32_t n = len >= k ? (uint32_t)(len - k + 1) : 0u;
    uint32_t n_weak = 0, n_small = 0, n_big = 0, mx_le = 1u, mn_gt = 0xFFFFu;
    auto take = [&](uint32_t c) {
        if (c == 0u || c > m_star) { n_big++; if (c != 0u && c < mn_gt) mn_gt = c; }
        else { if (c <= 1u) n_weak++; else n_small++; if (c > mx_le) mx_le = c; }
    };
    uint32_t i = 0;
    const uint8_t *p = cnt8 + f;
    for (; i < n && ((uintptr_t)(p + i) & 3u); i++) take(p[i]);
    for (; i + 4 <= n; i += 4) {
        const uint32_t w = *reinterpret_cast<const uint32_t *>(p + i);
        take(w & 0xFFu); take((w >> 8) & 0xFFu); take((w >> 16) & 0xFFu); take(w >> 24);
    }
    for (; i < n; i++) take(p[i]);
    const uint32_t c_le = n_weak + n_small, half = n / 2;
    const bool any_solid = (n_small + n_big) != 0u;            // all-ones reads are skipped (:4612)
    bool rescue = n && any_solid && c_le >= half + 1;
    if (n && any_solid && (n & 1u) == 0u && c_le == half) {
        const uint32_t median = (mx_le + mn_gt) / 2u;          // Utils::compute_median on u32 (Commons.hpp:2972-2988)
        rescue = !((float)median * 0.1f > 1.0f);               // :4610
    }
    resc_cnt[r] = rescue ? n_weak : 0u;
}

// rows of the rescued reads' count-1 instances, in read order then window order (abundance 1, :4630-4636)
__global__ __launch_bounds__(256) void emit_rescued_p_kernel(const uint64_t *off, const uint32_t *mins, uint32_t n_reads, uint32_t k, const uint8_t *cnt8,
                                                             const uint32_t *resc_cnt, const uint64_t *resc_pos, RowOut o, uint64_t row_base) {
    const unsigned sub = threadIdx.x & 15u, gshift = (threadIdx.x & 63u) & ~15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (uint64_t r = group; r < n_reads; r += ngroups) {
        if (resc_cnt[r] == 0u) continue;
        const uint64_t f = off[r];
        const uint32_t n = (uint32_t)(off[r + 1] - f - k + 1);  // rescued reads have instances
        const uint32_t *m0 = mins + f;
        uint64_t row = row_base + resc_pos[r];
        for (uint32_t i0 = 0; i0 < n; i0 += 16) {              // the 16 lanes of a group stay converged: ballot sees all of them
            const uint32_t i = i0 + sub;
            const bool weak = i < n && cnt8[f + i] == 1u;
            const uint32_t bal = (uint32_t)(__ballot(weak) >> gshift) & 0xFFFFu;
            if (weak) {
                const uint64_t dst = row + (uint32_t)__popc(bal & ((1u << sub) - 1u));
                const uint32_t *m = m0 + i;
                uint64_t hi, lo;
                const bool reversed = window_hash_uniform(m, k, hi, lo);
                o.lo[dst] = lo; o.hi[dst] = hi; o.ab[dst] = 1u;
                store_vector(m, k, reversed, o.vec, dst);
            }
            row += (uint32_t)__popc(bal);
        }
    }
}

// ---- the sharded first pass on the same buckets (mdbg_shard_begin / _finish) ------------------------------------------------------
// every distinct local key of every bucket, bucket after bucket, as contiguous arrays (what goes to the owners as rows)
__global__ __launch_bounds__(256) void gather_bucket_keys_kernel(RecView keys, const uint32_t *kcnt, const uint64_t *pos, uint32_t stride, const uint32_t *n_kept,
                                                                 const uint64_t *row_of, unsigned long long *lo, unsigned long long *hi, uint32_t *cnt, uint32_t *rep) {
    const uint64_t s0 = pos[(uint64_t)blockIdx.x * stride], d0 = row_of[blockIdx.x];
    const uint32_t n = n_kept[blockIdx.x];
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        lo[d0 + i] = keys.lo[s0 + i]; hi[d0 + i] = keys.hi[s0 + i]; cnt[d0 + i] = kcnt[s0 + i]; rep[d0 + i] = keys.rep[s0 + i];
    }
}

// The counts of the keys summed over all ranks are back: one workgroup per bucket puts its keys with their GLOBAL counts into an LDS table
// and walks the bucket's records once more -- cnt8[rep] = the global count of every instance whose key is rare (the rescue pass of the
// local reads against the global counts, graph/CreateMdbg.hpp:4514-4640).
template <uint32_t C>
__global__ __launch_bounds__(BC_NT) void bucket_apply_kernel(RecView in, const uint64_t *pos, uint32_t stride, const uint64_t *row_of, const unsigned long long *c_lo,
                                                             const unsigned long long *c_hi, const uint32_t *gcount, uint32_t clip, uint8_t *cnt8) {
    __shared__ LdsTable<C> t;
    const uint32_t tid = threadIdx.x;
    const uint64_t s0 = pos[(uint64_t)blockIdx.x * stride], s1 = pos[(uint64_t)(blockIdx.x + 1) * stride];
    const uint64_t k0 = row_of[blockIdx.x], k1 = row_of[blockIdx.x + 1];
    for (uint32_t s = tid; s < C; s += BC_NT) { t.lo[s] = 0ull; t.hi[s] = 0ull; t.cnt[s] = 0u; }
    __syncthreads();
    for (uint64_t i = k0 + tid; i < k1; i += BC_NT) {
        const uint32_t s = lds_upsert<C>(t, c_lo[i], c_hi[i], 0u);      // the keys are distinct and fitted this table once already
        if (s != LDS_NONE) t.cnt[s] = gcount[i];                        // (one thread per key: nobody else touches its slot)
    }
    __syncthreads();
    const uint64_t n = s1 - s0;
    for (uint64_t i0 = 0; i0 < n; i0 += BC_NT * BC_U) {
        uint64_t lo[BC_U], hi[BC_U]; uint32_t rep[BC_U];
#pragma unroll
        for (uint32_t q = 0; q < BC_U; q++) {
            const uint64_t i = i0 + (uint64_t)q * BC_NT + tid;
            if (i < n) { lo[q] = in.lo[s0 + i]; hi[q] = in.hi[s0 + i]; rep[q] = in.rep[s0 + i]; }
        }
#pragma unroll
        for (uint32_t q = 0; q < BC_U; q++) {
            const uint64_t i = i0 + (uint64_t)q * BC_NT + tid;
            if (i >= n) continue;
            const uint32_t s = lds_find<C>(t, lo[q], hi[q]);
            if (s == LDS_NONE) continue;
            const uint32_t c = t.cnt[s];
            if (c <= clip && c != 0u) cnt8[rep[q]] = (uint8_t)c;          // the array was filled with 0 = "many"
        }
    }
}

// rows of the keys this rank lists (flag), global counts, the vector through rep
__global__ __launch_bounds__(256) void emit_listed_rows_kernel(const unsigned long long *c_lo, const unsigned long long *c_hi, const uint32_t *gcount, const uint32_t *c_rep,
                                                               const uint32_t *flag, const uint64_t *row, uint64_t n, const uint32_t *mins, RowOut o) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const uint64_t r = row[i];
    o.lo[r] = c_lo[i]; o.hi[r] = c_hi[i]; o.ab[r] = gcount[i];
    const uint32_t *m = mins + c_rep[i];
    bool reversed = true;
    for (uint32_t q = 0; q < o.k; q++) {
        const uint32_t x = m[q], y = m[o.k - 1 - q];
        if (x == y) continue;
        reversed = !(x < y);
        break;
    }
    store_vector(m, o.k, reversed, o.vec, r);
}

// ---- the owner's side of a sharded pass: the rows it received, summed by key ---------------------------------------------------------
// rows [lo, hi, count] as they arrived -> records {lo, hi, index of the row}
__global__ __launch_bounds__(256) void rows_to_records_kernel(const uint64_t *rows, uint64_t n, RecView out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out.lo[i] = rows[3 * i]; out.hi[i] = rows[3 * i + 1]; out.rep[i] = (uint32_t)i;
}

template <uint32_t C>
__device__ __forceinline__ uint32_t lds_slot(LdsTable<C> &t, uint64_t lo, uint64_t hi) {      // find or create, nothing counted
    uint32_t s = (uint32_t)lo & (C - 1u);
    for (uint32_t probes = 0; probes < C; probes++, s = (s + 1u) & (C - 1u)) {
        unsigned long long cur = __hip_atomic_load(&t.lo[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (cur == 0ull) {
            cur = atomicCAS(&t.lo[s], 0ull, (unsigned long long)lo);
            if (cur == 0ull) cur = lo;
        }
        if (cur != lo) continue;
        unsigned long long h = __hip_atomic_load(&t.hi[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (h == 0ull) {
            h = atomicCAS(&t.hi[s], 0ull, (unsigned long long)hi);
            if (h == 0ull) h = hi;
        }
        if (h == hi) return s;
    }
    return LDS_NONE;
}

// One workgroup per bucket of received rows: the counts of equal keys summed in LDS, the row with the smallest index named the key's
// lister, and every row answered in place: reply[row] = sum | bit 63 for the lister (what rows_add_kernel / rows_reply_kernel do with one
// table in HBM and two device-scope atomics per row).
template <uint32_t C>
__global__ __launch_bounds__(BC_NT) void owner_bucket_kernel(RecView in, const uint64_t *pos, uint32_t stride, const uint64_t *rows, uint64_t *reply, uint32_t *overflow) {
    __shared__ LdsTable<C> t;
    __shared__ uint32_t give_up;
    const uint32_t tid = threadIdx.x;
    const uint64_t s0 = pos[(uint64_t)blockIdx.x * stride], n = pos[(uint64_t)(blockIdx.x + 1) * stride] - s0;
    for (uint32_t s = tid; s < C; s += BC_NT) { t.lo[s] = 0ull; t.hi[s] = 0ull; t.cnt[s] = 0u; t.rep[s] = 0xFFFFFFFFu; }
    if (tid == 0) give_up = __hip_atomic_load(overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (give_up) return;
    bool failed = false, zero_key = false;
    for (uint64_t i = tid; i < n; i += BC_NT) {
        const uint64_t lo = in.lo[s0 + i], hi = in.hi[s0 + i];
        const uint32_t row = in.rep[s0 + i];
        uint32_t s = LDS_NONE;
        if (lo != 0ull && hi != 0ull) s = lds_slot<C>(t, lo, hi);
        if (s == LDS_NONE) { failed = true; zero_key = zero_key || lo == 0ull || hi == 0ull; continue; }
        atomicAdd(&t.cnt[s], (uint32_t)rows[3ull * row + 2]);
        atomicMin(&t.rep[s], row);
    }
    if (failed) { atomicMax(overflow, zero_key ? PART_OVERFLOW_ZERO_KEY : PART_OVERFLOW_TABLE); __hip_atomic_store(&give_up, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    __syncthreads();
    const bool stop = give_up != 0u;
    if (stop) return;
    for (uint64_t i = tid; i < n; i += BC_NT) {
        const uint32_t row = in.rep[s0 + i];
        const uint32_t s = lds_find<C>(t, in.lo[s0 + i], in.hi[s0 + i]);
        reply[row] = (uint64_t)t.cnt[s] | (t.rep[s] == row ? (1ull << 63) : 0ull);
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------------
struct RecBufs {
    DevBuf<unsigned long long> lo, hi;
    DevBuf<uint32_t> rep;
    uint64_t cap = 0;
    int ensure(mdbg_ctx *ctx, uint64_t n) {
        if (n <= cap) return MDBG_OK;
        MDBG_TRY(lo.alloc(ctx, n)); MDBG_TRY(hi.alloc(ctx, n)); MDBG_TRY(rep.alloc(ctx, n));
        cap = n;
        return MDBG_OK;
    }
    RecView view() { return RecView{lo.p, hi.p, rep.p}; }
};

struct LevelPlan { uint32_t bits, shift, ways, blocks_per_seg; uint64_t n_seg; };

template <uint32_t C, uint32_t LIST>
static void launch_bucket_count(mdbg_ctx *ctx, uint64_t n_buckets, RecView in, const uint64_t *pos, uint32_t stride, uint32_t min_abundance, uint32_t clip, int keep_all,
                                RecView keys, uint32_t *kcnt, uint32_t *n_keys, uint32_t *n_kept, uint8_t *cnt8, uint32_t cnt8_default, uint32_t *overflow) {
    hipLaunchKernelGGL((bucket_count_kernel<C, LIST>), dim3((unsigned)n_buckets), dim3(BC_NT), 0, ctx->stream, in, pos, stride, min_abundance, clip, keep_all, keys, kcnt,
                       n_keys, n_kept, cnt8, cnt8_default, overflow);
}

// the rows of one group of keys, kept aside while the next group reuses the record buffers (several groups only)
struct GroupRows {
    DevBuf<uint64_t> lo, hi;
    DevBuf<uint32_t> ab, vec;
    uint64_t n = 0;
};

// HyperLogLog estimate from the merged registers, with the small-range (linear counting) correction
static double hll_estimate(const uint32_t *regs) {
    double sum = 0;
    uint32_t zeros = 0;
    for (uint32_t j = 0; j < HLL_M; j++) { sum += std::ldexp(1.0, -(int)regs[j]); zeros += regs[j] == 0; }
    const double m = (double)HLL_M, alpha = 0.7213 / (1.0 + 1.079 / m);
    double e = alpha * m * m / sum;
    if (e <= 2.5 * m && zeros) e = m * std::log(m / (double)zeros);
    return e;
}

// One partitioned pass over a read set: the plan, the buffers, and the two halves every user runs per group of keys -- split_group (the
// radix levels) and count_group (one workgroup per bucket) -- which mdbg_kminmer_count_first and the sharded first pass put together
// differently (the latter keeps every distinct key and walks the records a second time once the global counts are back).
constexpr uint32_t PART_MAX_BITS = 22;

struct PartRun {
    mdbg_ctx *ctx = nullptr;
    const mdbg_minimizers *reads = nullptr;
    uint32_t k = 0, n_reads = 0;
    uint64_t M = 0;
    // the plan: groups x buckets so that a bucket's distinct keys fit its LDS table at a load of at most 0.78 and the records of a
    // group fit the budget.  The number of distinct keys comes from the sketch the first level-1 histogram keeps (hll_estimate);
    // before that the guess is the last first pass's keys per instance (key_ratio_hint), which only decides the ways of level 1
    // for small inputs: from 2^8 buckets up level 1 always takes 8 bits, so the histogram is not run twice.
    uint64_t I_est = 0, n_groups = 1;
    uint32_t group_bits = 0, lds_slots = 0, bucket_bits = 0, n_levels = 1, extra_bits = 0, tile = PART_TILE;
    double keys_est = 1.0;
    bool sketched = false;
    LevelPlan lv[3];
    // buffers
    DevBuf<uint32_t> start_bits, kcnt, hist, n_keys, n_kept, overflow, hll_merged;
    DevBuf<uint64_t> place[3], key_pos, row_of;
    DevBuf<uint8_t> hll_blocks;
    RecBufs buf[2];
    // the group last split / counted
    uint64_t I = 0, n_buckets = 0;
    uint32_t cur = 0, stride = 1;
    const uint64_t *pos = nullptr;

    uint32_t bits_for(uint32_t c) const {
        // (a mean load of up to 0.78: the fullest of 65 536 buckets is then near 0.9, where an LDS probe sequence is still a dozen slots -- and
        // one radix level less is worth more than short probes: at 6 x coverage, 46 M keys in 172 M instances, 0.70 asked for a third level)
        const double b = keys_est / (double)n_groups / (0.78 * c);
        uint32_t n = 0;
        while ((double)(1ull << n) < b && n < 40) n++;
        return n;
    }
    static uint32_t levels_for(uint32_t bits) { return bits <= 8 ? 1u : (bits + 7u) / 8u; }
    void plan() {
        lds_slots = ctx->part_lds_slots;
        // (1024 slots where they need no level more than 2048 would: four buckets instead of two per CU count a fifth faster)
        if (!lds_slots) lds_slots = levels_for(bits_for(1024) + extra_bits) <= levels_for(bits_for(2048) + extra_bits) && bits_for(1024) + extra_bits <= PART_MAX_BITS ? 1024u : 2048u;
        bucket_bits = (ctx->part_bits ? ctx->part_bits : bits_for(lds_slots)) + extra_bits;
        n_levels = levels_for(bucket_bits);
        // level 1 takes 8 bits whenever there are that many, the deeper levels share the rest
        uint32_t left = bucket_bits, used = group_bits;
        uint64_t segs = 1;
        for (uint32_t l = 0; l < n_levels && l < 3; l++) {
            const uint32_t b = l == 0 ? std::min(left, 8u) : (left + (n_levels - l) - 1) / (n_levels - l);
            lv[l].bits = b; lv[l].ways = 1u << b; lv[l].n_seg = segs;
            // level 1 splits by window_mix, the deeper levels by the bits of hash_lo below the key group's
            lv[l].shift = (l == 0 || used + b == 0) ? 0u : 64u - used - b;      // (one way: the digit is masked to 0 whatever the shift)
            if (l > 0) used += b;
            left -= b; segs <<= b;
        }
    }
    // more distinct keys than the buckets of one launch hold (one 512-thread workgroup per bucket, and a launch of 2^32 threads wraps): not
    // for this path -- 2^22 buckets of 2048 slots are 6.7 G keys per key group, more than the 2^32 minimizers a call takes
    bool too_many_buckets() const { return bucket_bits > PART_MAX_BITS; }

    // sequence starts (one bit per minimizer), the key groups, the first plan
    int init(mdbg_ctx *c, const mdbg_minimizers *r, uint32_t k_) {
        ctx = c; reads = r; k = k_; M = r->n_min; n_reads = r->n_reads;
        const uint64_t n_words = (M >> 5) + 3;
        MDBG_TRY(start_bits.alloc(ctx, n_words));
        {
            LaunchTimer timer(ctx, "kminmer_split");
            MDBG_HIP_CHECK(ctx, hipMemsetAsync(start_bits.p, 0, n_words * 4, ctx->stream));
            hipLaunchKernelGGL(mark_starts_kernel, dim3(grid_for((uint64_t)n_reads + 1, 256)), dim3(256), 0, ctx->stream, reads->d_off.p, n_reads, start_bits.p);
        }
        I_est = std::max<uint64_t>(M / 8 + 1, M > (uint64_t)(k - 1) * n_reads ? M - (uint64_t)(k - 1) * n_reads : 0);
        const uint64_t max_records = ctx->part_max_records ? ctx->part_max_records : std::max<uint64_t>(1ull << 24, (uint64_t)((double)ctx->hbm_bytes * 0.25 / 44.0));
        group_bits = 0;
        while (((I_est + ((1ull << group_bits) - 1)) >> group_bits) > max_records && group_bits < 16) group_bits++;
        n_groups = 1ull << group_bits;
        keys_est = std::max(1.0, (double)I_est * ctx->key_ratio_hint[0]);
        sketched = ctx->part_bits != 0;                 // a forced plan needs no estimate
        tile = ctx->part_tile == 2048 ? PART_TILE / 2 : PART_TILE;
        MDBG_TRY(overflow.alloc(ctx, 1));
        plan();
        return MDBG_OK;
    }
    int begin_attempt() {
        plan();
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(overflow.p, 0, 4, ctx->stream));
        return MDBG_OK;
    }

    // the records of key group g through every radix level; leaves I, cur, pos, stride, n_buckets.  *give_up: the sketch asks for more
    // buckets than this path has (the caller hands the input back)
    int split_group(uint64_t g, bool *give_up) {
        *give_up = false;
        SplitArgs a{};
        a.mins = reads->d_min.p; a.start_bits = start_bits.p; a.n_min = M; a.k = k;
        a.group_bits = group_bits; a.group = g;
        a.tile = tile;
        // level 1: histogram, scan, the group's instance count, scatter
        const uint64_t tiles = (M + tile - 1) / tile;
        uint64_t entries = 0;
        for (;;) {
            lv[0].blocks_per_seg = (uint32_t)std::min<uint64_t>(tiles, 2048);
            a.seg_pos = nullptr; a.seg_stride = 1; a.blocks_per_seg = lv[0].blocks_per_seg; a.shift = lv[0].shift; a.ways = lv[0].ways; a.mix_bits = lv[0].bits;
            entries = (uint64_t)lv[0].ways * lv[0].blocks_per_seg;
            MDBG_TRY(hist.alloc(ctx, entries));
            MDBG_TRY(place[0].alloc(ctx, entries + 1));
            a.hll = nullptr;
            if (!sketched) {
                MDBG_TRY(hll_blocks.alloc(ctx, (uint64_t)lv[0].blocks_per_seg * HLL_M));
                MDBG_TRY(hll_merged.alloc(ctx, HLL_M));
                MDBG_HIP_CHECK(ctx, hipMemsetAsync(hll_merged.p, 0, HLL_M * 4, ctx->stream));
                a.hll = hll_blocks.p;
            }
            {
                LaunchTimer timer(ctx, "kminmer_split");
                hipLaunchKernelGGL(split_hist_kernel<true>, dim3(lv[0].blocks_per_seg), dim3(PART_NT), 0, ctx->stream, a, hist.p);
                if (a.hll) hipLaunchKernelGGL(hll_merge_kernel, dim3(HLL_M / 256, 64), dim3(256), 0, ctx->stream, hll_blocks.p, lv[0].blocks_per_seg, hll_merged.p);
            }
            MDBG_TRY(exclusive_scan_u32(ctx, hist.p, place[0].p, entries));
            uint32_t regs[HLL_M];
            if (a.hll) MDBG_HIP_CHECK(ctx, hipMemcpyAsync(regs, hll_merged.p, HLL_M * 4, hipMemcpyDeviceToHost, ctx->stream));
            MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &I, place[0].p + entries, 8, hipMemcpyDeviceToHost));
            if (!a.hll) break;
            // the sketch saw every key of every group: plan on it (+8 %: three and a half standard errors); the histogram is
            // only repeated when that changes level 1, i.e. for inputs of fewer keys than 2^8 buckets hold
            sketched = true;
            const uint32_t old_bits = lv[0].bits;
            keys_est = std::max(1.0, 1.08 * hll_estimate(regs));
            plan();
            MDBG_DBG(ctx, "partitioned first pass: about %.0f distinct keys", keys_est / 1.08);
            if (too_many_buckets()) { *give_up = true; return MDBG_OK; }
            if (lv[0].bits == old_bits) break;
        }
        n_buckets = 1ull << bucket_bits;
        MDBG_DBG(ctx, "partitioned first pass: group %llu of %llu, %llu instances, %u + %u + %u bits, %u LDS slots", (unsigned long long)g,
                 (unsigned long long)n_groups, (unsigned long long)I, lv[0].bits, n_levels > 1 ? lv[1].bits : 0, n_levels > 2 ? lv[2].bits : 0, lds_slots);
        MDBG_TRY(buf[0].ensure(ctx, I));
        MDBG_TRY(buf[1].ensure(ctx, I));
        if (kcnt.n < I || !kcnt.p) MDBG_TRY(kcnt.alloc(ctx, I));
        {
            LaunchTimer timer(ctx, "kminmer_split");
            if (tile != PART_TILE) hipLaunchKernelGGL((split_scatter_kernel<true, PART_E / 2>), dim3(lv[0].blocks_per_seg), dim3(PART_NT), 0, ctx->stream, a, place[0].p, buf[0].view());
            else hipLaunchKernelGGL((split_scatter_kernel<true, PART_E>), dim3(lv[0].blocks_per_seg), dim3(PART_NT), 0, ctx->stream, a, place[0].p, buf[0].view());
        }
        // deeper levels
        cur = 0;
        for (uint32_t l = 1; l < n_levels; l++) {
            const uint64_t n_seg = lv[l].n_seg;
            const uint64_t avg_tiles = (I / n_seg + tile - 1) / tile;
            lv[l].blocks_per_seg = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::max<uint64_t>(1, 4096 / n_seg), avg_tiles));
            SplitArgs d{};
            d.in = buf[cur].view();
            d.seg_pos = place[l - 1].p; d.seg_stride = lv[l - 1].blocks_per_seg; d.blocks_per_seg = lv[l].blocks_per_seg;
            d.shift = lv[l].shift; d.ways = lv[l].ways; d.n_min = 0; d.tile = tile;
            entries = n_seg * lv[l].ways * lv[l].blocks_per_seg;
            MDBG_TRY(hist.alloc(ctx, entries));
            MDBG_TRY(place[l].alloc(ctx, entries + 1));
            const unsigned grid = (unsigned)(n_seg * lv[l].blocks_per_seg);
            {
                LaunchTimer timer(ctx, "kminmer_split");
                hipLaunchKernelGGL(split_hist_kernel<false>, dim3(grid), dim3(PART_NT), 0, ctx->stream, d, hist.p);
            }
            MDBG_TRY(exclusive_scan_u32(ctx, hist.p, place[l].p, entries));
            {
                LaunchTimer timer(ctx, "kminmer_split");
                if (tile != PART_TILE) hipLaunchKernelGGL((split_scatter_kernel<false, PART_E / 2>), dim3(grid), dim3(PART_NT), 0, ctx->stream, d, place[l].p, buf[cur ^ 1].view());
                else hipLaunchKernelGGL((split_scatter_kernel<false, PART_E>), dim3(grid), dim3(PART_NT), 0, ctx->stream, d, place[l].p, buf[cur ^ 1].view());
            }
            cur ^= 1;
        }
        pos = place[n_levels - 1].p;
        stride = lv[n_levels - 1].blocks_per_seg;
        return MDBG_OK;
    }
    RecView records() { return buf[cur].view(); }
    RecView keys() { return buf[cur ^ 1].view(); }      // the free record buffer: every bucket's kept keys, compacted, at its start

    // one workgroup per bucket: counts, the kept keys, (cnt8) the rare instances' counts; then the buckets' key / kept-key offsets
    int count_group(uint32_t min_abundance, uint32_t clip, int keep_all, uint8_t *cnt8, uint32_t cnt8_default) {
        if (n_keys.n < n_buckets || !n_keys.p) {
            MDBG_TRY(n_keys.alloc(ctx, n_buckets));
            MDBG_TRY(n_kept.alloc(ctx, n_buckets));
            MDBG_TRY(key_pos.alloc(ctx, n_buckets + 1));
            MDBG_TRY(row_of.alloc(ctx, n_buckets + 1));
        }
        {
            LaunchTimer timer(ctx, "kminmer_insert");
#define MDBG_BC(C, L) launch_bucket_count<C, L>(ctx, n_buckets, records(), pos, stride, min_abundance, clip, keep_all, keys(), kcnt.p, n_keys.p, n_kept.p, cnt8, cnt8_default, overflow.p)
            const bool list = ctx->part_slot_list != 0;
            if (lds_slots == 256) { if (list) MDBG_BC(256, 8000); else MDBG_BC(256, 0); }
            else if (lds_slots == 1024) { if (list) MDBG_BC(1024, 8000); else MDBG_BC(1024, 0); }
            else { if (list) MDBG_BC(2048, 4096); else MDBG_BC(2048, 0); }
#undef MDBG_BC
        }
        MDBG_HIP_CHECK(ctx, hipGetLastError());
        MDBG_TRY(exclusive_scan_u32(ctx, n_keys.p, key_pos.p, n_buckets));
        MDBG_TRY(exclusive_scan_u32(ctx, n_kept.p, row_of.p, n_buckets));
        return MDBG_OK;
    }
    void record_info(uint32_t attempt, uint64_t total_inst) const {
        ctx->part_info[0] = 2; ctx->part_info[1] = n_groups; ctx->part_info[2] = bucket_bits; ctx->part_info[3] = n_levels;
        ctx->part_info[4] = attempt; ctx->part_info[5] = lds_slots; ctx->part_info[6] = n_buckets; ctx->part_info[7] = total_inst;
    }
};

static bool part_applicable(const mdbg_minimizers *reads, uint32_t k) {
    return k <= PART_MAX_K && reads->n_min >= k && reads->n_min < (1ull << 32);
}

int count_first_partitioned(mdbg_ctx *ctx, const mdbg_minimizers *reads, uint32_t k, uint32_t min_abundance, mdbg_table **out, bool *done) {
    *done = false;
    if (!part_applicable(reads, k)) return MDBG_OK;
    const uint64_t M = reads->n_min;
    const uint32_t n_reads = reads->n_reads;
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t m_star = rescue_m_star();
    const uint32_t clip = 2u * m_star + 1u;
    const bool do_rescue = min_abundance <= 1;
    if (do_rescue && clip > 254u) return MDBG_OK;

    PartRun run;
    MDBG_TRY(run.init(ctx, reads, k));
    DevBuf<uint8_t> cnt8;
    if (do_rescue) MDBG_TRY(cnt8.alloc(ctx, M));
    uint32_t cnt8_default = 0;
    DevBuf<uint32_t> resc_cnt;
    DevBuf<uint64_t> resc_pos;
    std::vector<GroupRows> group_rows;
    const uint64_t n_groups = run.n_groups;

    for (uint32_t attempt = 1;; attempt++) {
        MDBG_TRY(run.begin_attempt());
        if (run.too_many_buckets() || attempt > 4) {
            MDBG_DBG(ctx, "partitioned first pass gives up at %u bucket bits", run.bucket_bits);
            return MDBG_OK;
        }
        bool cnt8_filled = false;
        group_rows.clear();
        uint64_t total_inst = 0, total_keys = 0, total_solid = 0;
        bool overflowed = false;

        for (uint64_t g = 0; g < n_groups && !overflowed; g++) {
            bool give_up = false;
            MDBG_TRY(run.split_group(g, &give_up));
            if (give_up) return MDBG_OK;
            total_inst += run.I;
            if (do_rescue && !cnt8_filled) {            // (the sketch's estimate: more keys than half the instances = mostly singletons)
                cnt8_default = run.keys_est > 0.5 * (double)run.I_est ? 1u : 0u;
                MDBG_HIP_CHECK(ctx, hipMemsetAsync(cnt8.p, (int)cnt8_default, M, ctx->stream));
                cnt8_filled = true;
            }
            MDBG_TRY(run.count_group(min_abundance, clip, 0, do_rescue ? cnt8.p : nullptr, cnt8_default));
            if (n_groups == 1) break;                   // totals are read with the rescue count's, below
            uint64_t nk = 0, ns = 0; uint32_t ov = 0;
            MDBG_HIP_CHECK(ctx, hipMemcpyAsync(&nk, run.key_pos.p + run.n_buckets, 8, hipMemcpyDeviceToHost, ctx->stream));
            MDBG_HIP_CHECK(ctx, hipMemcpyAsync(&ns, run.row_of.p + run.n_buckets, 8, hipMemcpyDeviceToHost, ctx->stream));
            MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &ov, run.overflow.p, 4, hipMemcpyDeviceToHost));
            if (ov == PART_OVERFLOW_ZERO_KEY) { MDBG_DBG(ctx, "partitioned first pass: a key with a zero word: the one-table pass takes the input"); return MDBG_OK; }
            if (ov) { overflowed = true; break; }
            total_keys += nk; total_solid += ns;
            group_rows.emplace_back();
            GroupRows &gr = group_rows.back();
            gr.n = ns;
            MDBG_TRY(gr.lo.alloc(ctx, ns)); MDBG_TRY(gr.hi.alloc(ctx, ns)); MDBG_TRY(gr.ab.alloc(ctx, ns)); MDBG_TRY(gr.vec.alloc(ctx, ns * k));
            RowOut ro{gr.lo.p, gr.hi.p, gr.ab.p, gr.vec.p, k};
            LaunchTimer timer(ctx, "kminmer_emit");
            hipLaunchKernelGGL(emit_bucket_rows_kernel, dim3((unsigned)run.n_buckets), dim3(256), 0, ctx->stream, run.keys(), run.kcnt.p, run.pos, run.stride, run.n_kept.p,
                               run.row_of.p, reads->d_min.p, ro, (uint64_t)0);
        }

        // rescue decision per read, row positions
        uint64_t n_resc = 0;
        if (!overflowed && do_rescue) {
            MDBG_TRY(resc_cnt.alloc(ctx, n_reads));
            MDBG_TRY(resc_pos.alloc(ctx, (size_t)n_reads + 1));
            {
                LaunchTimer timer(ctx, "kminmer_rescue");
                hipLaunchKernelGGL(rescue_count_p_kernel, dim3(grid_for(n_reads, 256)), dim3(256), 0, ctx->stream, reads->d_off.p, n_reads, k, cnt8.p, m_star, resc_cnt.p);
            }
            MDBG_TRY(exclusive_scan_u32(ctx, resc_cnt.p, resc_pos.p, n_reads));
            MDBG_HIP_CHECK(ctx, hipMemcpyAsync(&n_resc, resc_pos.p + n_reads, 8, hipMemcpyDeviceToHost, ctx->stream));
        }
        if (!overflowed && n_groups == 1) {
            uint32_t ov = 0;
            MDBG_HIP_CHECK(ctx, hipMemcpyAsync(&total_keys, run.key_pos.p + run.n_buckets, 8, hipMemcpyDeviceToHost, ctx->stream));
            MDBG_HIP_CHECK(ctx, hipMemcpyAsync(&total_solid, run.row_of.p + run.n_buckets, 8, hipMemcpyDeviceToHost, ctx->stream));
            MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &ov, run.overflow.p, 4, hipMemcpyDeviceToHost));
            if (ov == PART_OVERFLOW_ZERO_KEY) { MDBG_DBG(ctx, "partitioned first pass: a key with a zero word: the one-table pass takes the input"); return MDBG_OK; }
            overflowed = ov != 0;
        } else {
            MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        }
        if (overflowed) {
            MDBG_DBG(ctx, "partitioned first pass: a bucket outgrew its %u LDS slots at %u bucket bits, again with more", run.lds_slots, run.bucket_bits);
            run.extra_bits += 2;
            continue;
        }

        update_key_hint(ctx, 0, total_keys, total_inst);
        std::unique_ptr<mdbg_table> t(new mdbg_table());
        t->k = k;
        t->n_solid = total_solid;
        t->st_minimizers = M; t->st_instances = total_inst; t->st_keys = total_keys; t->st_slots = n_groups * run.n_buckets * run.lds_slots;
        MDBG_TRY(alloc_rows(ctx, t.get(), total_solid + n_resc, true));
        RowOut ro{t->d_lo.p, t->d_hi.p, t->d_ab.p, t->d_vec.p, k};
        {
            LaunchTimer timer(ctx, "kminmer_emit");
            if (n_groups == 1) {
                hipLaunchKernelGGL(emit_bucket_rows_kernel, dim3((unsigned)run.n_buckets), dim3(256), 0, ctx->stream, run.keys(), run.kcnt.p, run.pos, run.stride,
                                   run.n_kept.p, run.row_of.p, reads->d_min.p, ro, (uint64_t)0);
            } else {
                uint64_t at = 0;
                for (GroupRows &gr : group_rows) {
                    if (gr.n) {
                        MDBG_HIP_CHECK(ctx, hipMemcpyAsync(t->d_lo.p + at, gr.lo.p, gr.n * 8, hipMemcpyDeviceToDevice, ctx->stream));
                        MDBG_HIP_CHECK(ctx, hipMemcpyAsync(t->d_hi.p + at, gr.hi.p, gr.n * 8, hipMemcpyDeviceToDevice, ctx->stream));
                        MDBG_HIP_CHECK(ctx, hipMemcpyAsync(t->d_ab.p + at, gr.ab.p, gr.n * 4, hipMemcpyDeviceToDevice, ctx->stream));
                        MDBG_HIP_CHECK(ctx, hipMemcpyAsync(t->d_vec.p + at * k, gr.vec.p, gr.n * k * 4, hipMemcpyDeviceToDevice, ctx->stream));
                    }
                    at += gr.n;
                }
            }
            if (n_resc) {
                const unsigned blocks = grid_for((uint64_t)n_reads * 16, 256, (unsigned)ctx->n_cu * 16u);
                hipLaunchKernelGGL(emit_rescued_p_kernel, dim3(blocks), dim3(256), 0, ctx->stream, reads->d_off.p, reads->d_min.p, n_reads, k, cnt8.p, resc_cnt.p,
                                   resc_pos.p, ro, total_solid);
            }
        }
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return set_error(ctx, MDBG_EHIP, "partitioned first pass failed: %s", hipGetErrorString(e));
        run.record_info(attempt, total_inst);
        *out = t.release();
        *done = true;
        return MDBG_OK;
    }
}

// ---- the sharded first pass: distinct local keys out, global counts back in ---------------------------------------------------------
struct PartLocal {
    PartRun run;
    DevBuf<unsigned long long> lo, hi;       // the distinct local keys, bucket after bucket
    DevBuf<uint32_t> cnt, rep;
    uint64_t n_keys = 0, n_inst = 0;
    uint32_t attempt = 1;
};

void part_local_free(PartLocal *p) { delete p; }

void part_local_arrays(const PartLocal *p, const uint64_t **lo, const uint64_t **hi, const uint32_t **cnt, uint64_t *n, uint64_t *n_inst) {
    *lo = reinterpret_cast<const uint64_t *>(p->lo.p); *hi = reinterpret_cast<const uint64_t *>(p->hi.p); *cnt = p->cnt.p; *n = p->n_keys; *n_inst = p->n_inst;
}

int part_local_keys(mdbg_ctx *ctx, const mdbg_minimizers *reads, uint32_t k, PartLocal **out, bool *done) {
    *done = false;
    if (!part_applicable(reads, k)) return MDBG_OK;
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<PartLocal> pl(new PartLocal());
    PartRun &run = pl->run;
    MDBG_TRY(run.init(ctx, reads, k));
    if (run.n_groups != 1) return MDBG_OK;            // (a rank's share that needs key groups: the one-table pass takes it)
    for (uint32_t attempt = 1;; attempt++) {
        MDBG_TRY(run.begin_attempt());
        if (run.too_many_buckets() || attempt > 4) return MDBG_OK;
        bool give_up = false;
        MDBG_TRY(run.split_group(0, &give_up));
        if (give_up) return MDBG_OK;
        MDBG_TRY(run.count_group(0u, 0u, 1, nullptr, 0u));       // keep every key; the instances' counts wait for the global ones
        uint32_t ov = 0;
        MDBG_HIP_CHECK(ctx, hipMemcpyAsync(&pl->n_keys, run.key_pos.p + run.n_buckets, 8, hipMemcpyDeviceToHost, ctx->stream));
        MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &ov, run.overflow.p, 4, hipMemcpyDeviceToHost));
        if (ov == PART_OVERFLOW_ZERO_KEY) return MDBG_OK;         // (no number of buckets helps: the one-table path)
        if (ov) { run.extra_bits += 2; continue; }
        pl->attempt = attempt;
        break;
    }
    pl->n_inst = run.I;
    const uint64_t D = pl->n_keys;
    MDBG_TRY(pl->lo.alloc(ctx, D)); MDBG_TRY(pl->hi.alloc(ctx, D)); MDBG_TRY(pl->cnt.alloc(ctx, D)); MDBG_TRY(pl->rep.alloc(ctx, D));
    {
        LaunchTimer timer(ctx, "shard_rows");
        hipLaunchKernelGGL(gather_bucket_keys_kernel, dim3((unsigned)run.n_buckets), dim3(256), 0, ctx->stream, run.keys(), run.kcnt.p, run.pos, run.stride, run.n_kept.p,
                           run.row_of.p, pl->lo.p, pl->hi.p, pl->cnt.p, pl->rep.p);
    }
    MDBG_HIP_CHECK(ctx, hipGetLastError());
    run.record_info(pl->attempt, run.I);
    // the free record buffer and the per-instance count array are not needed any more; the records, their bucket offsets and row_of are
    run.buf[run.cur ^ 1] = RecBufs();
    run.kcnt.release();
    *out = pl.release();
    *done = true;
    return MDBG_OK;
}

template <uint32_t C>
static void launch_bucket_apply(mdbg_ctx *ctx, PartLocal *pl, const uint32_t *gcount, uint32_t clip, uint8_t *cnt8) {
    PartRun &run = pl->run;
    hipLaunchKernelGGL(bucket_apply_kernel<C>, dim3((unsigned)run.n_buckets), dim3(BC_NT), 0, ctx->stream, run.records(), run.pos, run.stride, run.row_of.p, pl->lo.p, pl->hi.p,
                       gcount, clip, cnt8);
}

// gcount[i]: the global count of local key i; listed[i]: this rank lists the key (and it is solid)
int part_local_finish(mdbg_ctx *ctx, PartLocal *pl, const uint32_t *gcount, const uint32_t *listed, uint32_t min_abundance, mdbg_table **out) {
    PartRun &run = pl->run;
    const mdbg_minimizers *reads = run.reads;
    const uint32_t k = run.k, n_reads = run.n_reads;
    const uint64_t M = run.M, D = pl->n_keys;
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t m_star = rescue_m_star();
    const uint32_t clip = 2u * m_star + 1u;
    const bool do_rescue = min_abundance <= 1 && clip <= 254u;
    DevBuf<uint64_t> lpos, resc_pos;
    DevBuf<uint8_t> cnt8;
    DevBuf<uint32_t> resc_cnt;
    MDBG_TRY(lpos.alloc(ctx, D + 1));
    MDBG_TRY(exclusive_scan_u32(ctx, listed, lpos.p, D));
    uint64_t n_solid = 0, n_resc = 0;
    MDBG_HIP_CHECK(ctx, hipMemcpyAsync(&n_solid, lpos.p + D, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (do_rescue) {
        MDBG_TRY(cnt8.alloc(ctx, M));
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(cnt8.p, 0, M, ctx->stream));
        {
            LaunchTimer timer(ctx, "kminmer_insert");
            if (run.lds_slots == 256) launch_bucket_apply<256>(ctx, pl, gcount, clip, cnt8.p);
            else if (run.lds_slots == 1024) launch_bucket_apply<1024>(ctx, pl, gcount, clip, cnt8.p);
            else launch_bucket_apply<2048>(ctx, pl, gcount, clip, cnt8.p);
        }
        MDBG_TRY(resc_cnt.alloc(ctx, n_reads));
        MDBG_TRY(resc_pos.alloc(ctx, (size_t)n_reads + 1));
        {
            LaunchTimer timer(ctx, "kminmer_rescue");
            hipLaunchKernelGGL(rescue_count_p_kernel, dim3(grid_for(n_reads, 256)), dim3(256), 0, ctx->stream, reads->d_off.p, n_reads, k, cnt8.p, m_star, resc_cnt.p);
        }
        MDBG_TRY(exclusive_scan_u32(ctx, resc_cnt.p, resc_pos.p, n_reads));
        MDBG_HIP_CHECK(ctx, hipMemcpyAsync(&n_resc, resc_pos.p + n_reads, 8, hipMemcpyDeviceToHost, ctx->stream));
    }
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::unique_ptr<mdbg_table> t(new mdbg_table());
    t->k = k;
    t->n_solid = n_solid;
    t->st_minimizers = M; t->st_instances = pl->n_inst; t->st_keys = D; t->st_slots = run.n_buckets * run.lds_slots;
    MDBG_TRY(alloc_rows(ctx, t.get(), n_solid + n_resc, true));
    RowOut ro{t->d_lo.p, t->d_hi.p, t->d_ab.p, t->d_vec.p, k};
    {
        LaunchTimer timer(ctx, "kminmer_emit");
        if (D) hipLaunchKernelGGL(emit_listed_rows_kernel, dim3(grid_for(D, 256)), dim3(256), 0, ctx->stream, pl->lo.p, pl->hi.p, gcount, pl->rep.p, listed, lpos.p, D,
                                  reads->d_min.p, ro);
        if (n_resc) {
            const unsigned blocks = grid_for((uint64_t)n_reads * 16, 256, (unsigned)ctx->n_cu * 16u);
            hipLaunchKernelGGL(emit_rescued_p_kernel, dim3(blocks), dim3(256), 0, ctx->stream, reads->d_off.p, reads->d_min.p, n_reads, k, cnt8.p, resc_cnt.p, resc_pos.p, ro, n_solid);
        }
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return set_error(ctx, MDBG_EHIP, "sharded first pass (partitioned) failed: %s", hipGetErrorString(e));
    *out = t.release();
    return MDBG_OK;
}

// ---- the owner's side: n_recv rows [lo, hi, count] -> one reply per row (global count | bit 63 = this row's sender lists the key) -------
template <uint32_t C>
static void launch_owner_buckets(mdbg_ctx *ctx, uint64_t n_buckets, RecView in, const uint64_t *pos, uint32_t stride, const uint64_t *rows, uint64_t *reply, uint32_t *overflow) {
    hipLaunchKernelGGL(owner_bucket_kernel<C>, dim3((unsigned)n_buckets), dim3(BC_NT), 0, ctx->stream, in, pos, stride, rows, reply, overflow);
}

int part_owner_reduce(mdbg_ctx *ctx, const uint64_t *d_rows, uint64_t n_recv, uint64_t *d_reply, bool *done) {
    *done = false;
    if (n_recv < (1ull << 17) || n_recv >= (1ull << 32)) return MDBG_OK;
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t lds_slots = ctx->part_lds_slots == 256 || ctx->part_lds_slots == 2048 ? ctx->part_lds_slots : 1024u;
    // every bucket must hold its rows' distinct keys: sized for the ROWS (a key arrives once from every rank that saw it, so this is generous)
    uint32_t bits = 0;
    while ((double)(1ull << bits) < (double)n_recv / (0.78 * lds_slots) && bits <= PART_MAX_BITS) bits++;
    if (bits > PART_MAX_BITS) return MDBG_OK;
    const uint32_t n_levels = bits <= 8 ? 1u : (bits + 7u) / 8u;
    const uint32_t tile = ctx->part_tile == 2048 ? PART_TILE / 2 : PART_TILE;
    RecBufs buf[2];
    MDBG_TRY(buf[0].ensure(ctx, n_recv));
    MDBG_TRY(buf[1].ensure(ctx, n_recv));
    DevBuf<uint32_t> hist, overflow;
    DevBuf<uint64_t> place[3];
    MDBG_TRY(overflow.alloc(ctx, 1));
    MDBG_HIP_CHECK(ctx, hipMemsetAsync(overflow.p, 0, 4, ctx->stream));
    LaunchTimer timer(ctx, "shard_reduce");
    hipLaunchKernelGGL(rows_to_records_kernel, dim3(grid_for(n_recv, 256)), dim3(256), 0, ctx->stream, d_rows, n_recv, buf[0].view());
    uint32_t cur = 0, left = bits, used = 0, prev_bps = 1;
    uint64_t n_seg = 1;
    for (uint32_t l = 0; l < n_levels; l++) {
        const uint32_t b = (left + (n_levels - l) - 1) / (n_levels - l);
        SplitArgs d{};
        d.in = buf[cur].view();
        d.n_min = n_recv;                                          // level 1: one segment, the rows as they arrived
        d.seg_pos = l ? place[l - 1].p : nullptr; d.seg_stride = prev_bps;
        const uint64_t avg_tiles = (n_recv / n_seg + tile - 1) / tile;
        d.blocks_per_seg = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(std::max<uint64_t>(1, 4096 / n_seg), avg_tiles));
        d.ways = 1u << b; d.shift = b ? 64u - used - b : 0u; d.tile = tile;
        const uint64_t entries = n_seg * d.ways * d.blocks_per_seg;
        MDBG_TRY(hist.alloc(ctx, entries));
        MDBG_TRY(place[l].alloc(ctx, entries + 1));
        const unsigned grid = (unsigned)(n_seg * d.blocks_per_seg);
        hipLaunchKernelGGL(split_hist_kernel<false>, dim3(grid), dim3(PART_NT), 0, ctx->stream, d, hist.p);
        MDBG_TRY(exclusive_scan_u32(ctx, hist.p, place[l].p, entries));
        if (tile != PART_TILE) hipLaunchKernelGGL((split_scatter_kernel<false, PART_E / 2>), dim3(grid), dim3(PART_NT), 0, ctx->stream, d, place[l].p, buf[cur ^ 1].view());
        else hipLaunchKernelGGL((split_scatter_kernel<false, PART_E>), dim3(grid), dim3(PART_NT), 0, ctx->stream, d, place[l].p, buf[cur ^ 1].view());
        cur ^= 1; used += b; left -= b; n_seg <<= b; prev_bps = d.blocks_per_seg;
    }
    const uint64_t n_buckets = 1ull << bits;
    if (lds_slots == 256) launch_owner_buckets<256>(ctx, n_buckets, buf[cur].view(), place[n_levels - 1].p, prev_bps, d_rows, d_reply, overflow.p);
    else if (lds_slots == 1024) launch_owner_buckets<1024>(ctx, n_buckets, buf[cur].view(), place[n_levels - 1].p, prev_bps, d_rows, d_reply, overflow.p);
    else launch_owner_buckets<2048>(ctx, n_buckets, buf[cur].view(), place[n_levels - 1].p, prev_bps, d_rows, d_reply, overflow.p);
    MDBG_HIP_CHECK(ctx, hipGetLastError());
    uint32_t ov = 0;
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &ov, overflow.p, 4, hipMemcpyDeviceToHost));
    *done = ov == 0;                                               // a key with a zero word, or a bucket of more keys than slots: the one-table pass takes it
    return MDBG_OK;
}

}  // namespace mdbg
