// kminmer_dev.hpp -- device helpers shared by the k-min-mer translation units (kminmer.hip: one global hash table per pass;
// partition.hip: the key-partitioned first pass): canonical orientation + identity of a window (KmerVec::normalize,
// Commons.hpp:886-916; KmerVec::hash128, :941-969), the views of a sequence set and of the output rows.
#pragma once
#include "common.hpp"
#include "murmur.hpp"
#include "objects.hpp"

namespace mdbg {

// canonical orientation + hash128 of the window m[0..k).  Returns isReversed.
__device__ __forceinline__ bool window_hash(const uint32_t *m, uint32_t k, uint64_t &hi, uint64_t &lo) {
    bool reversed = true;  // palindrome => reversed (Commons.hpp:912-913)
    for (uint32_t i = 0; i < k; i++) {
        uint32_t a = m[i], b = m[k - 1 - i];
        if (a == b) continue;
        reversed = !(a < b);
        break;
    }
    Murmur128Stream h;
    if (reversed) for (uint32_t i = 0; i < k; i++) h.push(m[k - 1 - i]);
    else          for (uint32_t i = 0; i < k; i++) h.push(m[i]);
    h.finish(hi, lo);
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
