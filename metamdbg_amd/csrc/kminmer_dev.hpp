// kminmer_dev.hpp -- device helpers shared by the k-min-mer translation units (kminmer.hip: one global hash table per pass;
// partition.hip: the key-partitioned first pass): canonical orientation + identity of a window (KmerVec::normalize,
// Commons.hpp:886-916; KmerVec::hash128, :941-969), the views of a sequence set and of the output rows.
#pragma once
#include "common.hpp"
#include "murmur.hpp"
#include "objects.hpp"

namespace mdbg {

// canonical orientation + hash128 of the window m[0..k), any k.  Returns isReversed.
// Round 6: the reference's default loop runs k = 4 .. N50 x density x 2 (100 for 10 kb HiFi reads, Commons.hpp:1726-1741), and every
// k >= 12 comes here.  The word-at-a-time form of rounds 1 - 5 (Murmur128Stream::push: a four-way switch per minimizer, the window read
// through m[k - 1 - i] twice) made a pass at k = 13 .. 26 cost 30 - 32 ms against 17 at k = 11 (profiles/round6_*_index_by_k_deep.json);
// here the orientation is settled from both ends inwards (the first differing pair decides, as KmerVec::normalize's comparison of the
// vector with its reverse does, Commons.hpp:886-916; equal = palindrome => reversed), then the window is walked ONCE in 16-byte blocks
// along a per-lane stride of +1 or -1 word -- a block's two key words are mixed independently of the running state, so only the
// h1 / h2 chain is serial -- and the 0 .. 3 words of the tail are mixed as MurmurHash3_x64_128's tail switch does (MurmurHash3.cpp:369-395).
__device__ __forceinline__ bool window_hash(const uint32_t *m, uint32_t k, uint64_t &hi, uint64_t &lo) {
    bool reversed = true;  // palindrome => reversed (Commons.hpp:912-913)
    for (uint32_t i = 0, j = k - 1; i < j; i++, j--) {
        const uint32_t a = m[i], b = m[j];
        if (a == b) continue;
        reversed = a > b;
        break;
    }
    const int step = reversed ? -1 : 1;
    const uint32_t *p = reversed ? m + (k - 1) : m;
    uint64_t h1 = 0, h2 = 0;
    for (uint32_t b = k >> 2; b; b--, p += 4 * step) {
        uint64_t k1 = (uint64_t)p[0] | ((uint64_t)p[step] << 32), k2 = (uint64_t)p[2 * step] | ((uint64_t)p[3 * step] << 32);
        k1 *= MDBG_C1; k1 = rotl64(k1, 31); k1 *= MDBG_C2;
        k2 *= MDBG_C2; k2 = rotl64(k2, 33); k2 *= MDBG_C1;
        h1 ^= k1; h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        h2 ^= k2; h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }
    const uint32_t rem = k & 3u;
    if (rem == 3u) { uint64_t k2 = p[2 * step]; k2 *= MDBG_C2; k2 = rotl64(k2, 33); k2 *= MDBG_C1; h2 ^= k2; }
    if (rem) {
        uint64_t k1 = (uint64_t)p[0] | (rem >= 2u ? (uint64_t)p[step] << 32 : 0ull);
        k1 *= MDBG_C1; k1 = rotl64(k1, 31); k1 *= MDBG_C2; h1 ^= k1;
    }
    const uint64_t len = (uint64_t)k * 4;
    h1 ^= len; h2 ^= len;
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    hi = h1; lo = h2;
    return reversed;
}

// The same for a window length known at compile time: the comparison and the four-words-a-block hashing unroll, the
// stream's state machine folds away (about a third of the instructions of the general form).  k is a kernel argument
// (wave-uniform), so the dispatch is a scalar branch.
template <uint32_t KK>
__device__ __forceinline__ bool window_hash_fixed(const uint32_t *m, uint64_t &hi, uint64_t &lo) {
    uint32_t v[KK];
#pragma unroll
    for (uint32_t i = 0; i < KK; i++) v[i] = m[i];
    bool reversed = true, decided = false;
#pragma unroll
    for (uint32_t i = 0; i < KK / 2; i++) {
        const bool differ = v[i] != v[KK - 1 - i];
        if (!decided && differ) { reversed = !(v[i] < v[KK - 1 - i]); decided = true; }
    }
    Murmur128Stream h;
#pragma unroll
    for (uint32_t i = 0; i < KK; i++) h.push(reversed ? v[KK - 1 - i] : v[i]);
    h.finish(hi, lo);
    return reversed;
}

__device__ __forceinline__ bool window_hash_uniform(const uint32_t *m, uint32_t k, uint64_t &hi, uint64_t &lo) {
    switch (k) {
        case 3: return window_hash_fixed<3>(m, hi, lo);
        case 4: return window_hash_fixed<4>(m, hi, lo);
        case 5: return window_hash_fixed<5>(m, hi, lo);
        case 6: return window_hash_fixed<6>(m, hi, lo);
        case 7: return window_hash_fixed<7>(m, hi, lo);
        case 8: return window_hash_fixed<8>(m, hi, lo);
        case 9: return window_hash_fixed<9>(m, hi, lo);
        case 10: return window_hash_fixed<10>(m, hi, lo);
        case 11: return window_hash_fixed<11>(m, hi, lo);
        default: return window_hash(m, k, hi, lo);
    }
}

struct SeqView {
    const uint32_t *mins;
    const uint64_t *off;       // n_reads + 1
    const uint64_t *inst_off;  // n_reads + 1
    uint32_t n_reads;
    uint64_t n_inst;
    uint64_t n_min;            // minimizers in `mins`
};

// A table slot's `rep` names one instance of its key by the FLAT index of the window's first minimizer
// (set a first, then set b): reading the window back needs no search over the offsets.
__device__ __forceinline__ const uint32_t *rep_window(const SeqView &a, const SeqView &b, uint32_t rep) {
    return rep < a.n_min ? a.mins + rep : b.mins + (rep - a.n_min);
}

struct RowOut {
    uint64_t *lo, *hi;
    uint32_t *ab;
    uint32_t *vec;   // may be nullptr
    uint32_t k;
};

// write the canonical vector of the instance `rep` names (over one or two sequence sets)
__device__ __forceinline__ void write_instance_vector(const SeqView &a, const SeqView &b, uint32_t rep, uint32_t k, uint32_t *dst) {
    const uint32_t *m = rep_window(a, b, rep);
    bool reversed = true;
    for (uint32_t i = 0; i < k; i++) {
        uint32_t x = m[i], y = m[k - 1 - i];
        if (x == y) continue;
        reversed = !(x < y);
        break;
    }
    for (uint32_t i = 0; i < k; i++) dst[i] = reversed ? m[k - 1 - i] : m[i];
}

// largest median for which the reference's `double cutoff = median * 0.1f; if (cutoff > 1) return;` does not skip
// (graph/CreateMdbg.hpp:4610)
inline uint32_t rescue_m_star() {
    uint32_t m = 0;
    for (;;) {
        volatile float c = (float)(m + 1) * 0.1f;
        if (c > 1.0f) break;
        m++;
    }
    return m;
}

// host helpers of kminmer.hip used by partition.hip
int alloc_rows(mdbg_ctx *ctx, mdbg_table *t, uint64_t n, bool vec);
void update_key_hint(mdbg_ctx *ctx, int kind, uint64_t distinct, uint64_t instances);

// partition.hip: the first pass with the instances partitioned by key and counted in LDS.  *done = false: this
// input is not for it (the caller takes the one-table path); otherwise *out holds the table.
int count_first_partitioned(mdbg_ctx *ctx, const mdbg_minimizers *reads, uint32_t k, uint32_t min_abundance, mdbg_table **out, bool *done);

// partition.hip: the same for a rank's share of a sharded first pass.  part_local_keys: every distinct local key with its local count
// (device arrays lo / hi / cnt, bucket after bucket) -- what mdbg_shard_begin groups by owner; part_local_finish: the table of this rank's
// share from the keys' global counts and the keys it was told to list (listed[i] != 0), with its own reads rescued against the global counts.
struct PartLocal;
int part_local_keys(mdbg_ctx *ctx, const mdbg_minimizers *reads, uint32_t k, PartLocal **out, bool *done);
void part_local_arrays(const PartLocal *p, const uint64_t **lo, const uint64_t **hi, const uint32_t **cnt, uint64_t *n, uint64_t *n_inst);
int part_local_finish(mdbg_ctx *ctx, PartLocal *p, const uint32_t *gcount, const uint32_t *listed, uint32_t min_abundance, mdbg_table **out);
void part_local_free(PartLocal *p);
// the owner's side of a sharded pass: n_recv rows [lo, hi, count] summed by key in per-bucket LDS tables; one reply per row, in place
// (sum | bit 63 for the row that lists the key).  *done = false: not for this path (few rows, a key with a zero word): the one-table pass
int part_owner_reduce(mdbg_ctx *ctx, const uint64_t *d_rows, uint64_t n_recv, uint64_t *d_reply, bool *done);

}  // namespace mdbg
