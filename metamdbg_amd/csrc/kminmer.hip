// kminmer.hip -- minimizer-space reads -> k-min-mer tables on the device.
//
// Every window of k consecutive minimizers of every sequence is one *instance*
// (MDBG::getKminmers_complete, Commons.hpp:5282-5361).  One lane per instance: orient the window
// (KmerVec::normalize, Commons.hpp:886-916: lexicographic compare with its reverse, tie => reversed),
// stream it through Murmur3 x64-128 seed 0 (KmerVec::hash128, Commons.hpp:941-969) and meet equal
// keys in a device hash table (table.hpp).
//   first pass (k = firstK)  : slot value = occurrence count; solid = count > 1 (and >= min abundance);
//                              rescue pass appends count-1 instances of reads whose median abundance
//                              is <= 10 (graph/CreateMdbg.hpp:3591-3883, :4514-4640)
//   k = firstK+1             : distinct keys; abundance = min over the two (k-1)-sub-min-mers of the
//                              previous table, missing/0 => 1; keep > 1 (graph/CreateMdbg.hpp:3933-4005)
//   k >= firstK+2            : abundance = min(prev[i], prev[i+1]) along the sequence; insert-if-absent
//                              when > 1 (graph/CreateMdbg.hpp:1240-1265, :1450-1459)
#include "common.hpp"
#include "murmur.hpp"
#include "objects.hpp"
#include "table.hpp"
#include "kminmer_dev.hpp"

#include <vector>

namespace mdbg {

// ---- instance indexing -------------------------------------------------------------------------
__global__ void inst_count_kernel(const uint64_t *off, uint32_t n_reads, uint32_t k, uint32_t *cnt) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads) {
        uint64_t n = off[r + 1] - off[r];
        cnt[r] = n >= k ? (uint32_t)(n - k + 1) : 0u;
    }
}

// read owning global instance g: largest r with inst_off[r] <= g
__device__ __forceinline__ uint32_t find_read(const uint64_t *inst_off, uint32_t n_reads, uint64_t g) {
    uint32_t lo = 0, hi = n_reads;  // invariant: inst_off[lo] <= g < inst_off[hi]
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (inst_off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ uint32_t table_upsert_count(const TableView &t, uint64_t lo, uint64_t hi, uint32_t add, uint32_t rep) {
    if (lo == 0ull || hi == 0ull) return table_exc_upsert(t, lo, hi, add, 0, false, rep, true);
    bool created = false;
    uint32_t s = table_find_or_insert(t, lo, hi, true, &created);
    if (s != SLOT_NONE) {
        if (add) atomicAdd(&t.slots[s].val, add);
        if (created) t.slots[s].rep = rep;   // the instance that published the key represents it: one store per key, not per instance
    }
    return s;
}

__device__ __forceinline__ uint32_t table_upsert_set(const TableView &t, uint64_t lo, uint64_t hi, uint32_t v, uint32_t rep) {
    if (lo == 0ull || hi == 0ull) return table_exc_upsert(t, lo, hi, 0, v, true, rep, true);
    bool created = false;
    uint32_t s = table_find_or_insert(t, lo, hi, true, &created);
    if (s != SLOT_NONE) {
        __hip_atomic_store(&t.slots[s].val, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (created) t.slots[s].rep = rep;
    }
    return s;
}

// insert-if-absent with a value that is a function of the key (the index passes: min of the previous abundances of the window's two
// sub-windows, the same for every instance of the key and in either orientation): only the instance that publishes the key stores it.
// The other instances -- nineteen in twenty at 50x -- stop at the probe instead of each sending its own copy of the value to memory.
__device__ __forceinline__ uint32_t table_insert_once(const TableView &t, uint64_t lo, uint64_t hi, uint32_t v) {
    if (lo == 0ull || hi == 0ull) return table_exc_upsert(t, lo, hi, 0, v, true, 0u, true);
    bool created = false;
    uint32_t s = table_find_or_insert(t, lo, hi, true, &created);
    if (s != SLOT_NONE && created) __hip_atomic_store(&t.slots[s].val, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return s;
}

// the same when the rows of the table carry their vectors (k = firstK + 1): the publishing instance also names itself
__device__ __forceinline__ uint32_t table_insert_once_rep(const TableView &t, uint64_t lo, uint64_t hi, uint32_t v, uint32_t rep) {
    if (lo == 0ull || hi == 0ull) return table_exc_upsert(t, lo, hi, 0, v, true, rep, true);
    bool created = false;
    uint32_t s = table_find_or_insert(t, lo, hi, true, &created);
    if (s != SLOT_NONE && created) { __hip_atomic_store(&t.slots[s].val, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); t.slots[s].rep = rep; }
    return s;
}

// Visit every instance with 16 lanes per sequence (4 sequences per wave): sequences hold a few dozen
// windows, the minimizers of neighbouring windows are loaded coalesced, and no per-instance binary search
// over the offsets is needed.  f(read, global instance id, pointer to the window's first minimizer).
// `give_up` (a table's overflow flag, may be null): once set the pass is void -- the host grows the table and repeats it --
// so the remaining sequences are skipped.  (Without this, a table sized for another kind of data -- HiFi's one key per ten
// instances, then an ONT batch with nine per ten -- was filled to the brim at 96 probes per insert: 3.9 s for a pass that
// was going to be thrown away.)
template <typename F>
__device__ __forceinline__ void for_each_instance(const SeqView &s, F f, const uint32_t *give_up = nullptr) {
    const unsigned sub = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    uint32_t trip = 0;
    for (uint64_t r = group; r < s.n_reads; r += ngroups) {
        // (looked at every 32nd sequence: thousands of groups reading one address at every step is a hot spot of its own)
        if (give_up && (trip++ & 31u) == 0u && __hip_atomic_load(give_up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        const uint64_t base = s.inst_off[r];
        const uint32_t n = (uint32_t)(s.inst_off[r + 1] - base);
        const uint32_t *m0 = s.mins + s.off[r];
        for (uint32_t i = sub; i < n; i += 16) f((uint32_t)r, base + i, m0 + i);
    }
}

// Grid of the kernels that walk every k-min-mer instance.  Alone they want every wave slot they can get (the insert is
// bound by memory-side atomics); beside another context's scan a few resident blocks per CU (mdbg_set_option "table_blocks_per_cu" / MDBG_TABLE_BLOCKS_PER_CU,
// grid-stride over the reads) keep them from displacing the scan's waves: 3 per CU costs the insert 2.9 -> 3.6 ms and
// gives the scan back 0.5 ms, which is what the step then runs at.
static unsigned instance_grid(const mdbg_ctx *ctx, uint32_t n_reads, unsigned lanes = 16) {
    if (ctx->table_grid_blocks) return grid_for((uint64_t)n_reads * lanes, 256, ctx->table_grid_blocks);
    return grid_for((uint64_t)n_reads * lanes, 256, (unsigned)ctx->n_cu * ctx->table_blocks_per_cu);
}

// ---- first pass -----------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void count_insert_kernel(SeqView s, uint32_t k, TableView t, uint32_t *inst_slot,
                                                           uint64_t rep_base) {
    for_each_instance(s, [&](uint32_t, uint64_t g, const uint32_t *m) {
        uint64_t hi, lo;
        window_hash_uniform(m, k, hi, lo);
#if defined(INSERT_ABLATE) && INSERT_ABLATE == 3
        if (inst_slot) inst_slot[g] = (uint32_t)(lo ^ hi);                                          // ablation: hash only
#elif defined(INSERT_ABLATE) && INSERT_ABLATE == 2
        bool created = false;
        uint32_t slot = table_find_or_insert(t, lo, hi, true, &created);                            // ablation: no count
        if (inst_slot) inst_slot[g] = slot;
#else
        uint32_t slot = table_upsert_count(t, lo, hi, 1u, (uint32_t)(rep_base + (uint64_t)(m - s.mins)));
#if !defined(INSERT_ABLATE) || INSERT_ABLATE != 1
        if (inst_slot) inst_slot[g] = slot;
#else
        if (inst_slot && slot == 0x7FFFFFFEu) inst_slot[g] = slot;                                  // ablation: no slot store
#endif
#endif
    }, t.poll_overflow ? t.overflow : nullptr);
}

// ---- rescue (graph/CreateMdbg.hpp:4514-4640) --------------------------------------------------------
// The rescue pass needs the count of every instance's key.  Reading it from the 32-byte table slots is one
// random 64-byte sector per instance (0.65 ms per 34 M instances); instead every slot's count is first
// reduced to a 2-bit class -- all the decision below needs except in one rare tie case -- packed 4 slots per
// byte: 1 MB for a 4 M-slot table, L2-resident, so the per-instance reads are cache hits.
//   class 0: count <= 1   not solid, counts as 1 (CreateMdbg.hpp:4598)
//   class 1: 2 .. m*      solid, <= m*
//   class 2: > m*         solid
// m* is the largest u32 whose float product m * 0.1f is not > 1 (computed on the host: 10).
// class of instance g's key (failed inserts count as 1)
__device__ __forceinline__ uint32_t count_class(const uint8_t *cls, uint64_t cap, uint32_t slot) {
    if (slot == SLOT_NONE) return 0u;
    const uint64_t idx = (slot & 0x80000000u) ? cap + (slot & 0x7FFFFFFFu) : (uint64_t)slot;
    return (cls[idx >> 2] >> (2u * (uint32_t)(idx & 3u))) & 3u;
}

// Rescue decision per read without sorting.  A k-min-mer is solid iff its count is > 1 (the rescue pass only
// runs when min_abundance <= 1, graph/CreateMdbg.cpp:317-319).  "median * 0.1f > 1" (:4610) is false
//   odd n : iff at least n/2+1 abundances are <= m*
//   even n: iff at least n/2+1 are <= m*, or exactly n/2 are and (max{<= m*} + min{> m*}) / 2 passes the
//           same float test (Utils::compute_median on u32, Commons.hpp:2972-2988) -- the one case that needs
//           exact counts, which are then read from the table
// 16 lanes per read (4 reads per wave): reads hold a few dozen k-min-mers.  Output: per read the number of
// non-solid instances to append (0 when the read is not rescued).
__global__ __launch_bounds__(256) void rescue_count_kernel(const uint64_t *inst_off, uint32_t n_reads, const uint32_t *inst_slot,
                                                           const uint8_t *cls, TableView t, uint64_t cap, uint32_t m_star, uint32_t *resc_cnt) {
    const unsigned sub = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (uint64_t r0 = 0; r0 < n_reads; r0 += ngroups) {     // uniform trip count: shuffles need all lanes
        const uint64_t r = r0 + group;
        const bool live = r < n_reads;
        const uint64_t f = live ? inst_off[r] : 0;
        const uint32_t n = live ? (uint32_t)(inst_off[r + 1] - f) : 0u;
        uint32_t n_weak = 0, n_small = 0, n_big = 0;
        for (uint32_t i = sub; i < n; i += 16) {
            const uint32_t c = count_class(cls, cap, inst_slot[f + i]);
            n_weak += c == 0u; n_small += c == 1u; n_big += c == 2u;
        }
#pragma unroll
        for (int d = 8; d >= 1; d >>= 1) {                     // reduce within the 16-lane group
            n_weak += __shfl_xor(n_weak, d, 64);
            n_small += __shfl_xor(n_small, d, 64);
            n_big += __shfl_xor(n_big, d, 64);
        }
        const uint32_t c_le = n_weak + n_small, half = n / 2;
        const bool any_solid = (n_small + n_big) != 0u;        // all-ones reads are skipped (:4612)
        bool rescue = n && any_solid && c_le >= half + 1;
        const bool tie = n && any_solid && (n & 1u) == 0u && c_le == half;
        // the branch below is per read, i.e. uniform over the 16 lanes that shuffle with each other
        uint32_t mx_le = 1u, mn_gt = 0xFFFFFFFFu;
        if (__any(tie)) {
            if (tie) {
                for (uint32_t i = sub; i < n; i += 16) {
                    const uint32_t slot = inst_slot[f + i];
                    uint32_t a = slot == SLOT_NONE ? 1u : table_slot_val(t, slot);
                    if (a <= 1u) a = 1u;
                    if (a <= m_star) mx_le = a > mx_le ? a : mx_le; else mn_gt = a < mn_gt ? a : mn_gt;
                }
            }
#pragma unroll
            for (int d = 8; d >= 1; d >>= 1) {                 // all lanes of the wave take part (tie or not)
                uint32_t x = __shfl_xor(mx_le, d, 64); mx_le = x > mx_le ? x : mx_le;
                x = __shfl_xor(mn_gt, d, 64); mn_gt = x < mn_gt ? x : mn_gt;
            }
            if (tie) {
                const uint32_t median = (uint32_t)(mx_le + mn_gt) / 2u;
                rescue = !((float)median * 0.1f > 1.0f);       // :4610
            }
        }
        if (live && sub == 0) resc_cnt[r] = rescue ? n_weak : 0u;
    }
}

// One pass over the table: which slots become output rows, and (optionally) the 2-bit count class of every slot
// for the rescue pass.  A lane reads its 32-byte slot as two 16-byte loads; 4 neighbouring lanes pack their
// classes into one byte.
__global__ __launch_bounds__(256) void slot_flag_kernel(TableView t, uint64_t cap, uint32_t min_abundance, int mode,
                                                        uint32_t *flag, uint8_t *cls = nullptr, uint32_t m_star = 0, uint32_t *occ_count = nullptr) {
    // mode 0: solid (count > 1 and >= min_abundance); mode 1: any occupied slot with val > 1; mode 2: occupied
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in_range = s < cap + TABLE_EXC_CAP;   // cap is a multiple of 256: whole waves are in or out of the main table
    bool occ = false; uint32_t v = 0;
    if (s < cap) {
        const uint4 *q = reinterpret_cast<const uint4 *>(t.slots + s);
        const uint4 key = q[0], rest = q[1];
        occ = (key.x | key.y) != 0u;
        v = rest.x;
    } else if (in_range) {
        uint32_t i = (uint32_t)(s - cap); occ = i < *t.exc_n; v = occ ? t.exc_val[i] : 0u;
    }
    bool keep = false;
    if (occ) {
        if (mode == 0) keep = v > 1u && !(v < min_abundance);
        else if (mode == 1) keep = v > 1u;
        else keep = true;
    }
    if (in_range && flag) flag[s] = keep ? 1u : 0u;
    if (cls) {
        uint32_t c = (!occ || v <= 1u) ? 0u : (v <= m_star ? 1u : 2u);
        c |= __shfl_down(c, 1, 64) << 2;             // lanes 4j..4j+3 -> lane 4j
        c |= __shfl_down(c, 2, 64) << 4;
        if (in_range && (threadIdx.x & 3u) == 0u) cls[s >> 2] = (uint8_t)c;
    }
    if (occ_count) {   // uniform branch: distinct keys, for sizing the next table of this kind
        const int n = __syncthreads_count(occ);
        if (threadIdx.x == 0 && n) atomicAdd(&occ_count[blockIdx.x % TABLE_OCC_WAYS], (uint32_t)n);
    }
}

__global__ __launch_bounds__(256) void emit_slots_kernel(TableView t, uint64_t cap, const uint32_t *flag, const uint64_t *pos,
                                                         SeqView a, SeqView b, RowOut o, uint64_t row_base) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap + TABLE_EXC_CAP || !flag[s]) return;
    uint64_t row = row_base + pos[s];
    uint32_t rep;
    if (s < cap) { const TableSlot &sl = t.slots[s]; o.lo[row] = sl.lo; o.hi[row] = sl.hi; o.ab[row] = sl.val; rep = sl.rep; }
    else { uint32_t i = (uint32_t)(s - cap); o.lo[row] = t.exc_lo[i]; o.hi[row] = t.exc_hi[i]; o.ab[row] = t.exc_val[i]; rep = t.exc_rep[i]; }
    if (o.vec) write_instance_vector(a, b, rep, o.k, o.vec + row * o.k);
}

// rows of the rescued reads' non-solid instances, in read order then window order (abundance 1, :4630-4636)
__global__ __launch_bounds__(256) void emit_rescued_kernel(SeqView s, uint32_t k, const uint32_t *inst_slot, const uint8_t *cls, uint64_t cap,
                                                           const uint32_t *resc_cnt, const uint64_t *resc_pos, RowOut o, uint64_t row_base) {
    const unsigned sub = threadIdx.x & 15u, gshift = (threadIdx.x & 63u) & ~15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (uint64_t r = group; r < s.n_reads; r += ngroups) {
        if (resc_cnt[r] == 0u) continue;
        const uint64_t f = s.inst_off[r];
        const uint32_t n = (uint32_t)(s.inst_off[r + 1] - f);
        const uint32_t *m0 = s.mins + s.off[r];
        uint64_t row = row_base + resc_pos[r];
        for (uint32_t i0 = 0; i0 < n; i0 += 16) {              // the 16 lanes of a group stay converged: ballot sees all of them
            const uint32_t i = i0 + sub;
            const bool weak = i < n && count_class(cls, cap, inst_slot[f + i]) == 0u;
            const uint32_t bal = (uint32_t)(__ballot(weak) >> gshift) & 0xFFFFu;
            if (weak) {
                const uint64_t dst = row + (uint32_t)__popc(bal & ((1u << sub) - 1u));
                const uint32_t *m = m0 + i;
                uint64_t hi, lo;
                bool reversed = window_hash_uniform(m, k, hi, lo);
                o.lo[dst] = lo; o.hi[dst] = hi; o.ab[dst] = 1u;
                for (uint32_t j = 0; j < k; j++) o.vec[dst * k + j] = reversed ? m[k - 1 - j] : m[j];
            }
            row += (uint32_t)__popc(bal);
        }
    }
}

// The rescue pass over one read set against the counts in `t`: per-read decision, row positions, total.
struct RescuePlan {
    DevBuf<uint8_t> cls;      // 2-bit count class per table slot
    DevBuf<uint32_t> cnt;
    DevBuf<uint64_t> pos;
    uint64_t cap = 0, total = 0;
};

// ---- k > firstK --------------------------------------------------------------------------------------
// distinct keys of all k-windows (k = firstK+1)
__global__ __launch_bounds__(256) void distinct_insert_kernel(SeqView s, uint32_t k, TableView t, uint64_t rep_base) {
    for_each_instance(s, [&](uint32_t, uint64_t g, const uint32_t *m) {
        uint64_t hi, lo;
        window_hash_uniform(m, k, hi, lo);
        table_upsert_count(t, lo, hi, 0u, (uint32_t)(rep_base + (uint64_t)(m - s.mins)));
    }, t.poll_overflow ? t.overflow : nullptr);
}

// The same with U windows of a lane in flight and a first look at every window's home slot by plain loads (see index_insert_u_kernel:
// nine instances in ten at 50 x meet a key that is already there, and those are done without an atomic).
template <int U, bool FAST>
__global__ __launch_bounds__(256) void distinct_insert_u_kernel(SeqView s, uint32_t k, TableView t, uint64_t rep_base) {
    const unsigned sub = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    uint32_t trip = 0;
    for (uint64_t r = group; r < s.n_reads; r += ngroups) {
        if (t.poll_overflow && (trip++ & 31u) == 0u && __hip_atomic_load(t.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        const uint32_t n = (uint32_t)(s.inst_off[r + 1] - s.inst_off[r]);
        const uint32_t *m0 = s.mins + s.off[r];
        for (uint32_t i0 = sub; i0 < n; i0 += 16u * U) {
            uint64_t hi[U], lo[U], home[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = i0 + 16u * (uint32_t)u;
                hi[u] = lo[u] = 0; home[u] = 0;
                if (i < n) { window_hash_uniform(m0 + i, k, hi[u], lo[u]); home[u] = table_home(lo[u], hi[u], t.mask); }
            }
            SlotWords w[U];
            if (FAST) {
#pragma unroll
                for (int u = 0; u < U; u++) w[u] = slot_load(&t.slots[home[u]]);
#pragma unroll
                for (int u = 0; u < U; u++) asm volatile("" : "+v"(w[u].hi));
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = i0 + 16u * (uint32_t)u;
                if (i >= n) continue;
                if (FAST && lo[u] != 0ull && hi[u] != 0ull && w[u].lo == lo[u] && w[u].hi == hi[u]) continue;
                table_upsert_count(t, lo[u], hi[u], 0u, (uint32_t)(rep_base + (uint64_t)(m0 + i - s.mins)));
            }
        }
    }
}

// refined abundance of every distinct key (graph/CreateMdbg.hpp:3933-3970): min over the two
// (k-1)-sub-min-mers of the canonical vector; missing or 0 => 1
__global__ __launch_bounds__(256) void refine_slots_kernel(TableView t, uint64_t cap, SeqView a, SeqView b, uint32_t k,
                                                           TableView prev) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap + TABLE_EXC_CAP) return;
    bool occ; uint32_t rep;
    if (s < cap) { occ = t.slots[s].lo != 0ull; rep = occ ? t.slots[s].rep : 0; }
    else { uint32_t i = (uint32_t)(s - cap); occ = i < *t.exc_n; rep = occ ? t.exc_rep[i] : 0; }
    if (!occ) return;
    const uint32_t *m = rep_window(a, b, rep);
    // sub-windows of the CANONICAL vector; min over both is orientation independent, so use m directly
    uint32_t min_ab = 0xFFFFFFFFu;
    for (uint32_t i = 0; i < 2; i++) {
        uint64_t hi, lo;
        window_hash_uniform(m + i, k - 1, hi, lo);
        uint32_t v;
        if (table_lookup(prev, lo, hi, v)) {
            if (v == 0u) { min_ab = 1u; break; }
            if (v < min_ab) min_ab = v;
        } else { min_ab = 1u; break; }
    }
    if (s < cap) t.slots[s].val = min_ab; else t.exc_val[s - cap] = min_ab;
}

// abundance of every (k-1)-window along the sequences (getPrevAbundances, graph/CreateMdbg.hpp:1240-1265); the previous table in
// either form (TableView: one 32-byte slot per key; BucketView: three keys per 64-byte sector, table.hpp)
template <typename View>
__global__ __launch_bounds__(256) void prev_abundance_kernel(SeqView s /* instances of size k-1 */, uint32_t km1, View prev, uint32_t *out) {
    for_each_instance(s, [&](uint32_t, uint64_t g, const uint32_t *m) {
        uint64_t hi, lo;
        window_hash_uniform(m, km1, hi, lo);
        uint32_t v;
        out[g] = key_lookup(prev, lo, hi, v) ? v : 1u;
    });
}

// The same with U windows of a lane in flight at once (one-slot table).  These passes are random 64-byte sectors, one per window, and a
// lane that hashes a window, waits for its slot, stores and only then turns to its next window has one request in flight at a time; a
// sequence holds two or three windows per lane (16 lanes, some 34 windows).  Here a lane hashes U of them, sends U slot loads off
// together -- each a slot's key and value in one trip (slot_load) -- and then takes the answers.  WIDE = false: the field-by-field
// look-up of rounds 1 - 4 (A/B: mdbg_set_option "index_tuning").
template <int U, bool WIDE>
__global__ __launch_bounds__(256) void prev_abundance_u_kernel(SeqView s /* instances of size k-1 */, uint32_t km1, TableView prev, uint32_t *out) {
    const unsigned sub = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (uint64_t r = group; r < s.n_reads; r += ngroups) {
        const uint64_t base = s.inst_off[r];
        const uint32_t n = (uint32_t)(s.inst_off[r + 1] - base);
        const uint32_t *m0 = s.mins + s.off[r];
        for (uint32_t i0 = sub; i0 < n; i0 += 16u * U) {
            uint64_t hi[U], lo[U], home[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = i0 + 16u * (uint32_t)u;
                hi[u] = lo[u] = 0; home[u] = 0;
                if (i < n) { window_hash_uniform(m0 + i, km1, hi[u], lo[u]); home[u] = table_home(lo[u], hi[u], prev.mask); }
            }
            uint32_t v[U];
            if (WIDE) {
                SlotWords w[U];
#pragma unroll
                for (int u = 0; u < U; u++) w[u] = slot_load(&prev.slots[home[u]]);        // (a lane without a window reads slot 0 and ignores it)
#pragma unroll
                for (int u = 0; u < U; u++) asm volatile("" : "+v"(w[u].val));             // all loads are out before the first answer is looked at (the
                                                                                           // compiler otherwise sinks a value's load behind its key's compare: one more trip)
#pragma unroll
                for (int u = 0; u < U; u++) {
                    v[u] = 1u;
                    if (lo[u] == 0ull || hi[u] == 0ull) { uint32_t x; if (i0 + 16u * (uint32_t)u < n && table_lookup_side(prev, lo[u], hi[u], x)) v[u] = x; }
                    else if (w[u].lo == lo[u] && w[u].hi == hi[u]) v[u] = w[u].val;
                    else if (w[u].lo != 0ull) { uint32_t x; if (table_lookup_from(prev, table_next(home[u], prev.mask), 1, lo[u], hi[u], x)) v[u] = x; }
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; u++) { uint32_t x; v[u] = (i0 + 16u * (uint32_t)u < n && table_lookup_narrow(prev, lo[u], hi[u], x)) ? x : 1u; }
            }
#pragma unroll
            for (int u = 0; u < U; u++) { const uint32_t i = i0 + 16u * (uint32_t)u; if (i < n) out[base + i] = v[u]; }
        }
    }
}

// Round 6: the look-up with BOTH slots of a window's home sector fetched at once (a table's probe sequences start at the even slot of a
// 64-byte sector, table.hpp) and LANES lanes a sequence.  What tools/ubench/lookup_ablate.hip measured on 320 M look-ups into 14 M keys:
// the first slot alone and the second by a dependent trip 38.5 G/s at load 0.39, both at once 42.1, with 32 lanes 43.2, the same at load
// 0.2: 47.0 -- against 47 - 50 for the same kernel WITHOUT its hash or without its store: the rate at which this part serves random sectors.
template <int LANES>
__global__ __launch_bounds__(256) void prev_abundance_pair_kernel(SeqView s /* instances of size k-1 */, uint32_t km1, TableView prev, uint32_t *out) {
    const unsigned sub = threadIdx.x & (LANES - 1);
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / LANES;
    for (uint64_t r = group; r < s.n_reads; r += ngroups) {
        const uint64_t base = s.inst_off[r];
        const uint32_t n = (uint32_t)(s.inst_off[r + 1] - base);
        const uint32_t *m0 = s.mins + s.off[r];
        for (uint32_t i = sub; i < n; i += LANES) {
            uint64_t hi, lo;
            window_hash_uniform(m0 + i, km1, hi, lo);
            uint32_t v = 1u, x;
            if (lo == 0ull || hi == 0ull) { if (table_lookup_side(prev, lo, hi, x)) v = x; }
            else {
                const uint64_t home = table_home(lo, hi, prev.mask);
                SlotPair p = pair_load(&prev.slots[home]);
                asm volatile("" : "+v"(p.a.val), "+v"(p.b.val));     // all four loads are out before the first compare (see prev_abundance_u_kernel)
                const int f = pair_verdict(p, lo, hi, x);
                if (f > 0) v = x;
                else if (f < 0 && table_lookup_from(prev, table_next(home + 1, prev.mask), 2, lo, hi, x)) v = x;
            }
            out[base + i] = v;
        }
    }
}

// k-window i of read r gets min(prev[i], prev[i+1]); insert-if-absent when > 1 (graph/CreateMdbg.hpp:1440-1459)
__global__ __launch_bounds__(256) void index_insert_kernel(SeqView s /* k */, const uint64_t *inst_off_km1, const uint32_t *prev_ab,
                                                           uint32_t k, TableView t) {
    for_each_instance(s, [&](uint32_t r, uint64_t g, const uint32_t *m) {
        uint64_t j = inst_off_km1[r] + (g - s.inst_off[r]);
        uint32_t a0 = prev_ab[j], a1 = prev_ab[j + 1];
        uint32_t a = a0 < a1 ? a0 : a1;
        if (a <= 1u) return;
        uint64_t hi, lo;
        window_hash_uniform(m, k, hi, lo);
        table_insert_once(t, lo, hi, a);
    }, t.poll_overflow ? t.overflow : nullptr);
}

// The same with U windows of a lane in flight and, with FAST, a first look at every window's home slot by PLAIN loads: nineteen in
// twenty of these inserts meet a key that is already there, and a key that a plain load shows in its slot IS there (a slot's words
// are written once and never change; what a stale cache line can show is an empty or half-published slot, never a wrong key) -- those
// instances are done without an atomic.  The others -- absent, displaced from its home slot, or not yet visible -- take the claim-and-
// publish path as before (table_insert_once), whose device-scope loads go to the memory side every time.
template <int U, bool FAST>
__global__ __launch_bounds__(256) void index_insert_u_kernel(SeqView s /* k */, const uint64_t *inst_off_km1, const uint32_t *prev_ab,
                                                             uint32_t k, TableView t) {
    const unsigned sub = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    uint32_t trip = 0;
    for (uint64_t r = group; r < s.n_reads; r += ngroups) {
        if (t.poll_overflow && (trip++ & 31u) == 0u && __hip_atomic_load(t.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        const uint64_t base = s.inst_off[r];
        const uint32_t n = (uint32_t)(s.inst_off[r + 1] - base);
        const uint32_t *m0 = s.mins + s.off[r];
        const uint64_t j0 = inst_off_km1[r];
        for (uint32_t i0 = sub; i0 < n; i0 += 16u * U) {
            uint64_t hi[U], lo[U], home[U];
            uint32_t a[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = i0 + 16u * (uint32_t)u;
                a[u] = 0; hi[u] = lo[u] = 0; home[u] = 0;
                if (i < n) {
                    const uint32_t a0 = prev_ab[j0 + i], a1 = prev_ab[j0 + i + 1];
                    a[u] = a0 < a1 ? a0 : a1;
                    if (a[u] > 1u) { window_hash_uniform(m0 + i, k, hi[u], lo[u]); home[u] = table_home(lo[u], hi[u], t.mask); }
                }
            }
            if (FAST) {
                SlotWords w[U];
#pragma unroll
                for (int u = 0; u < U; u++) w[u] = slot_load(&t.slots[home[u]]);
#pragma unroll
                for (int u = 0; u < U; u++) asm volatile("" : "+v"(w[u].hi));
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (a[u] > 1u && !(lo[u] != 0ull && hi[u] != 0ull && w[u].lo == lo[u] && w[u].hi == hi[u])) table_insert_once(t, lo[u], hi[u], a[u]);
            } else {
#pragma unroll
                for (int u = 0; u < U; u++) if (a[u] > 1u) table_insert_once(t, lo[u], hi[u], a[u]);
            }
        }
    }
}

// Round 6: the insert with the plain-load first look at BOTH slots of the home sector and LANES lanes a sequence.
template <int LANES>
__global__ __launch_bounds__(256) void index_insert_pair_kernel(SeqView s /* k */, const uint64_t *inst_off_km1, const uint32_t *prev_ab, uint32_t k, TableView t) {
    const unsigned sub = threadIdx.x & (LANES - 1);
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / LANES;
    uint32_t trip = 0;
    for (uint64_t r = group; r < s.n_reads; r += ngroups) {
        if (t.poll_overflow && (trip++ & 31u) == 0u && __hip_atomic_load(t.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        const uint32_t n = (uint32_t)(s.inst_off[r + 1] - s.inst_off[r]);
        const uint32_t *m0 = s.mins + s.off[r];
        const uint64_t j0 = inst_off_km1[r];
        for (uint32_t i = sub; i < n; i += LANES) {
            const uint32_t a0 = prev_ab[j0 + i], a1 = prev_ab[j0 + i + 1];
            const uint32_t a = a0 < a1 ? a0 : a1;
            if (a <= 1u) continue;
            uint64_t hi, lo;
            window_hash_uniform(m0 + i, k, hi, lo);
            if (lo != 0ull && hi != 0ull) {
                const TableSlot *home = &t.slots[table_home(lo, hi, t.mask)];
                const uint4 ka = *reinterpret_cast<const uint4 *>(home), kb = *reinterpret_cast<const uint4 *>(home + 1);
                const uint32_t l0 = (uint32_t)lo, l1 = (uint32_t)(lo >> 32), h0 = (uint32_t)hi, h1 = (uint32_t)(hi >> 32);
                if ((ka.x == l0 && ka.y == l1 && ka.z == h0 && ka.w == h1) || (kb.x == l0 && kb.y == l1 && kb.z == h0 && kb.w == h1)) continue;
            }
            table_insert_once(t, lo, hi, a);
        }
    }
}

// Look-up and insert in ONE kernel (round 5, "index_tuning" bit 3).  The two-kernel form writes the abundance of every (k-1)-window to HBM and
// reads it back twice (12 bytes per instance, 4 GB a pass at 10 M reads) only to hand a value to the lane next door.  Here the 16 lanes of a
// sequence take 16 consecutive (k-1)-windows, each looks its own up, a lane gets its right neighbour's answer by a DPP move inside the row of
// 16 (row_shl:1), and lanes 0 .. 14 insert the k-window that starts where their (k-1)-window does; the group advances by 15, so one look-up
// in sixteen is made twice.  No instance index at k - 1, no intermediate array.  (The 16 lanes of a group run the same trip count -- one
// sequence -- so every lane a DPP move reads from is active.)
template <bool FAST>
__global__ __launch_bounds__(256) void index_fused_kernel(SeqView s /* k */, uint32_t k, TableView prev, TableView t) {
    const unsigned sub = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    uint32_t trip = 0;
    for (uint64_t r = group; r < s.n_reads; r += ngroups) {
        if (t.poll_overflow && (trip++ & 31u) == 0u && __hip_atomic_load(t.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
        const uint32_t n = (uint32_t)(s.inst_off[r + 1] - s.inst_off[r]);       // k-windows; the sequence has n + 1 (k-1)-windows when n > 0
        const uint32_t *m0 = s.mins + s.off[r];
        for (uint32_t i0 = 0; i0 < n; i0 += 15u) {
            const uint32_t i = i0 + sub;
            uint32_t v = 1u;
            if (i <= n) {                                                         // getPrevAbundances: missing => 1 (graph/CreateMdbg.hpp:1240-1265)
                uint64_t hi, lo;
                window_hash_uniform(m0 + i, k - 1u, hi, lo);
                uint32_t x;
                if (table_lookup(prev, lo, hi, x)) v = x;
            }
            const uint32_t vn = (uint32_t)__builtin_amdgcn_update_dpp(1, (int)v, 0x101, 0xf, 0xf, false);      // row_shl:1: the lane above, inside the row of 16
            if (sub < 15u && i < n) {
                const uint32_t a = v < vn ? v : vn;
                if (a > 1u) {
                    uint64_t hi, lo;
                    window_hash_uniform(m0 + i, k, hi, lo);
                    bool there = false;
                    if (FAST && lo != 0ull && hi != 0ull) {
                        const SlotWords w = slot_load(&t.slots[table_home(lo, hi, t.mask)]);
                        there = w.lo == lo && w.hi == hi;
                    }
                    if (!there) table_insert_once(t, lo, hi, a);
                }
            }
        }
    }
}

// Round 6: INSERT FIRST, ask the previous table only where the answer is needed ("index_tuning" bit 7; chosen per pass by a sample, see
// index_one_set).  Most k-windows of a read set are instances of a key that is already in the table (coverage), and for those nothing of the
// previous table is wanted: the abundance min(prev[i], prev[i+1]) only decides whether an ABSENT key is inserted, and with what value.  So a
// lane hashes its k-window and looks at its home slot of the NEW table first (plain loads, as index_insert_u_kernel<.., FAST>); the (k-1)-window
// i is looked up only when k-window i or k-window i - 1 was not seen there -- the first instances of a key, and the windows with an error in
// them (a <= 1: never inserted, so never found: 12 % of the instances of HiFi reads at k = 6, 22 % at k = 11, 42 % at k = 24;
// tools/index_miss_fraction.py).  Random sectors per instance: 1 + f (1 + 1 / k) instead of 2 - f.  The layout of index_fused_kernel:
// 16 lanes take 16 consecutive (k-1)-windows, lanes 0 .. 14 own the k-window that starts there, neighbours talk over DPP inside the row.
// stats[0] += k-windows looked at, stats[1] += those not found at the first look (what the next choice could go by).
// U chunks of 15 windows of a sequence in flight per group (A/B, MDBG_INDEX_LAZY_U): 11.9 / 12.2 / 13.2 ms a pass at k = 6 for U = 1 / 2 / 3 --
// the pass is not waiting for the round trips of its few look-ups; U = 1 runs.
// REP (the refined pass, k = firstK + 1, whose rows carry their vectors): the instance that publishes a key names itself, rep_base + the flat
// index of its first minimizer.
template <int U, bool REP = false>
__global__ __launch_bounds__(256) void index_lazy_kernel(SeqView s /* k */, uint32_t k, TableView prev, TableView t, unsigned long long *stats, uint64_t rep_base = 0) {
    const unsigned sub = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    uint32_t trip = 0, n_seen = 0, n_need = 0;
    for (uint64_t r = group; r < s.n_reads; r += ngroups) {
        if (t.poll_overflow && (trip++ & 31u) == 0u && __hip_atomic_load(t.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
        const uint32_t n = (uint32_t)(s.inst_off[r + 1] - s.inst_off[r]);       // k-windows; the sequence has n + 1 (k-1)-windows when n > 0
        const uint32_t *m0 = s.mins + s.off[r];
        for (uint32_t i0 = 0; i0 < n; i0 += 15u * U) {
            // ---- every lane's k-windows: hash, first look at the home slot of the NEW table (all loads out before the first compare)
            uint64_t khi[U], klo[U], home[U];
            bool need[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = i0 + 15u * (uint32_t)u + sub;
                khi[u] = klo[u] = 0; home[u] = 0;
                need[u] = sub < 15u && i < n;
                if (need[u]) {
                    window_hash_uniform(m0 + i, k, khi[u], klo[u]);
                    if (klo[u] != 0ull && khi[u] != 0ull) home[u] = table_home(klo[u], khi[u], t.mask);
                }
            }
            {
                // BOTH slots of the home sector (a key pushed off its home slot sits in the second nine times in ten: with the home slot alone
                // a tenth of the keys were "not seen" every time they came -- 21.7 % of the instances at k = 6 where 16 % are absent)
                uint4 wa[U], wb[U];
#pragma unroll
                for (int u = 0; u < U; u++) {                                               // (a lane without a window reads slot 0 and ignores it)
                    wa[u] = *reinterpret_cast<const uint4 *>(&t.slots[home[u]]);
                    wb[u] = *reinterpret_cast<const uint4 *>(&t.slots[home[u]] + 1);
                }
#pragma unroll
                for (int u = 0; u < U; u++) asm volatile("" : "+v"(wa[u].w), "+v"(wb[u].w));
#pragma unroll
                for (int u = 0; u < U; u++)
                    if (need[u]) {
                        n_seen++;
                        const uint32_t l0 = (uint32_t)klo[u], l1 = (uint32_t)(klo[u] >> 32), h0 = (uint32_t)khi[u], h1 = (uint32_t)(khi[u] >> 32);
                        if (klo[u] != 0ull && khi[u] != 0ull &&
                            ((wa[u].x == l0 && wa[u].y == l1 && wa[u].z == h0 && wa[u].w == h1) || (wb[u].x == l0 && wb[u].y == l1 && wb[u].z == h0 && wb[u].w == h1)))
                            need[u] = false;
                        n_need += need[u] ? 1u : 0u;
                    }
            }
            // ---- (k-1)-window i is asked for when k-window i or k-window i - 1 was not seen: both slots of its home sector at once
            bool ask[U];
            uint64_t phi[U], plo[U], phome[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = i0 + 15u * (uint32_t)u + sub;
                const int need_left = __builtin_amdgcn_update_dpp(0, need[u] ? 1 : 0, 0x111, 0xf, 0xf, false);     // row_shr:1: the lane below, inside the row of 16
                ask[u] = i <= n && (need[u] || need_left);
                phi[u] = plo[u] = 0; phome[u] = 0;
                if (ask[u]) {
                    window_hash_uniform(m0 + i, k - 1u, phi[u], plo[u]);
                    if (plo[u] != 0ull && phi[u] != 0ull) phome[u] = table_home(plo[u], phi[u], prev.mask);
                }
            }
            uint32_t v[U];
            {
                SlotPair p[U];
#pragma unroll
                for (int u = 0; u < U; u++) p[u] = pair_load(&prev.slots[phome[u]]);
#pragma unroll
                for (int u = 0; u < U; u++) asm volatile("" : "+v"(p[u].a.val), "+v"(p[u].b.val));
#pragma unroll
                for (int u = 0; u < U; u++) {
                    v[u] = 1u;                                                    // getPrevAbundances: missing => 1 (graph/CreateMdbg.hpp:1240-1265)
                    if (ask[u]) {
                        uint32_t x;
                        if (plo[u] == 0ull || phi[u] == 0ull) { if (table_lookup_side(prev, plo[u], phi[u], x)) v[u] = x; }
                        else {
                            const int f = pair_verdict(p[u], plo[u], phi[u], x);
                            if (f > 0) v[u] = x;
                            else if (f < 0 && table_lookup_from(prev, table_next(phome[u] + 1, prev.mask), 2, plo[u], phi[u], x)) v[u] = x;
                        }
                    }
                }
            }
            // ---- the k-windows that were not seen: inserted when their abundance says so
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t vn = (uint32_t)__builtin_amdgcn_update_dpp(1, (int)v[u], 0x101, 0xf, 0xf, false);      // row_shl:1: the lane above
                if (need[u]) {
                    const uint32_t a = v[u] < vn ? v[u] : vn;
                    if (a > 1u) {
                        if (REP) table_insert_once_rep(t, klo[u], khi[u], a, (uint32_t)(rep_base + (uint64_t)(m0 + (i0 + 15u * (uint32_t)u + sub) - s.mins)));
                        else table_insert_once(t, klo[u], khi[u], a);
                    }
                }
            }
        }
    }
    if (stats) {
        unsigned long long a = n_seen, b = n_need;
        for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
        if ((threadIdx.x & 63u) == 0u && a) { atomicAdd(&stats[0], a); atomicAdd(&stats[1], b); }
    }
}

// What the choice between the two forms goes by: over every `step`-th sequence, how many k-windows there are and how many of them have
// min(prev[i], prev[i+1]) <= 1 -- the windows that are never inserted, which the insert-first form asks about every time they come.
__global__ __launch_bounds__(256) void index_miss_sample_kernel(SeqView s /* k */, uint32_t k, TableView prev, uint32_t step, unsigned long long *stats) {
    const unsigned sub = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    uint32_t n_seen = 0, n_low = 0;
    for (uint64_t r = group * step; r < s.n_reads; r += ngroups * step) {
        const uint32_t n = (uint32_t)(s.inst_off[r + 1] - s.inst_off[r]);
        const uint32_t *m0 = s.mins + s.off[r];
        for (uint32_t i0 = 0; i0 < n; i0 += 15u) {
            const uint32_t i = i0 + sub;
            uint32_t v = 1u;
            if (i <= n) {
                uint64_t hi, lo;
                window_hash_uniform(m0 + i, k - 1u, hi, lo);
                uint32_t x;
                if (table_lookup(prev, lo, hi, x)) v = x;
            }
            const uint32_t vn = (uint32_t)__builtin_amdgcn_update_dpp(1, (int)v, 0x101, 0xf, 0xf, false);
            if (sub < 15u && i < n) { n_seen++; n_low += (v < vn ? v : vn) <= 1u ? 1u : 0u; }
        }
    }
    unsigned long long a = n_seen, b = n_low;
    for (int d = 32; d >= 1; d >>= 1) { a += __shfl_xor(a, d, 64); b += __shfl_xor(b, d, 64); }
    if ((threadIdx.x & 63u) == 0u && a) { atomicAdd(&stats[0], a); atomicAdd(&stats[1], b); }
}

// the same into a bucket table; rep_base + the flat index of the window's first minimizer names the instance that published the key
// (kept when the table keeps representatives: k = firstK + 1 writes the vectors of its rows)
__global__ __launch_bounds__(256) void index_insert_b_kernel(SeqView s /* k */, const uint64_t *inst_off_km1, const uint32_t *prev_ab,
                                                             uint32_t k, BucketView t, uint64_t rep_base) {
    for_each_instance(s, [&](uint32_t r, uint64_t g, const uint32_t *m) {
        uint64_t j = inst_off_km1[r] + (g - s.inst_off[r]);
        uint32_t a0 = prev_ab[j], a1 = prev_ab[j + 1];
        uint32_t a = a0 < a1 ? a0 : a1;
        if (a <= 1u) return;
        uint64_t hi, lo;
        window_hash_uniform(m, k, hi, lo);
        bucket_insert_once(t, lo, hi, a, (uint32_t)(rep_base + (uint64_t)(m - s.mins)));
    }, t.side.poll_overflow ? t.side.overflow : nullptr);
}

// which entries of a bucket table are rows (every occupied one: only abundances above 1 were inserted), and how many there are
__global__ __launch_bounds__(256) void bucket_flag_kernel(BucketView t, uint32_t *flag) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = t.nb * BUCKET_WAYS;
    if (e >= n + TABLE_EXC_CAP) return;
    bool occ;
    if (e < n) occ = t.b[e / BUCKET_WAYS].lo[e % BUCKET_WAYS] != 0ull;
    else occ = (uint32_t)(e - n) < *t.side.exc_n;
    flag[e] = occ ? 1u : 0u;
}

__global__ __launch_bounds__(256) void bucket_emit_kernel(BucketView t, const uint32_t *flag, const uint64_t *pos, SeqView a, SeqView b, RowOut o) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n = t.nb * BUCKET_WAYS;
    if (e >= n + TABLE_EXC_CAP || !flag[e]) return;
    const uint64_t row = pos[e];
    uint32_t rep = 0;
    if (e < n) {
        const KeyBucket &B = t.b[e / BUCKET_WAYS];
        const uint32_t j = (uint32_t)(e % BUCKET_WAYS);
        o.lo[row] = B.lo[j]; o.hi[row] = B.hi[j]; o.ab[row] = B.val[j];
        if (o.vec) rep = t.rep[e];
    } else {
        const uint32_t i = (uint32_t)(e - n);
        o.lo[row] = t.side.exc_lo[i]; o.hi[row] = t.side.exc_hi[i]; o.ab[row] = t.side.exc_val[i]; rep = t.side.exc_rep[i];
    }
    if (o.vec) write_instance_vector(a, b, rep, o.k, o.vec + row * o.k);
}

// the rows of a finished table into a bucket table (abundance 1 skipped as loadRefinedAbundances does, graph/CreateMdbg.cpp:3445)
__global__ __launch_bounds__(256) void bucket_rows_insert_kernel(const uint64_t *lo, const uint64_t *hi, const uint32_t *ab, uint64_t n,
                                                                 int skip_one, BucketView t) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (skip_one && ab[i] == 1u) return;
    bucket_upsert_set(t, lo[i], hi[i], ab[i]);
}

// every key of a one-slot table (a previous table loaded from records, perhaps overlaid with unitig abundances) with its value
__global__ __launch_bounds__(256) void bucket_from_table_kernel(TableView src, uint64_t cap, BucketView t) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < cap) {
        const TableSlot &sl = src.slots[s];
        if (sl.lo != 0ull) bucket_upsert_set(t, sl.lo, sl.hi, sl.val);
    } else if (s < cap + TABLE_EXC_CAP) {
        const uint32_t i = (uint32_t)(s - cap);
        if (i < *src.exc_n) bucket_upsert_set(t, src.exc_lo[i], src.exc_hi[i], src.exc_val[i]);
    }
}

__global__ __launch_bounds__(256) void table_occupied_kernel(TableView t, uint64_t cap, uint32_t *occ_count) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool occ = s < cap ? t.slots[s].lo != 0ull : (s < cap + TABLE_EXC_CAP && (uint32_t)(s - cap) < *t.exc_n);
    const int n = __syncthreads_count(occ);
    if (threadIdx.x == 0 && n) atomicAdd(&occ_count[blockIdx.x % TABLE_OCC_WAYS], (uint32_t)n);
}

// ---- prev tables ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_insert_kernel(const uint64_t *lo, const uint64_t *hi, const uint32_t *ab, uint64_t n,
                                                          int skip_one, TableView t) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (skip_one && ab[i] == 1u) return;   // graph/CreateMdbg.cpp:3445
    table_upsert_set(t, lo[i], hi[i], ab[i], 0u);
}

// (graph/CreateMdbg.cpp:3466-3507)
__global__ __launch_bounds__(256) void overlay_kernel(SeqView s, uint32_t kprev, const uint32_t *unitig_ab, TableView t) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= s.n_inst) return;
    uint32_t r = find_read(s.inst_off, s.n_reads, g);
    uint32_t a = unitig_ab[r];
    if (a == 0xFFFFFFFFu) return;   // unitig without refined abundance
    const uint32_t *m = s.mins + s.off[r] + (g - s.inst_off[r]);
    uint64_t hi, lo;
    window_hash(m, kprev, hi, lo);
    if (a == 1u) {
        // modify_if: set to 0 only when present
        if (lo == 0ull || hi == 0ull) { table_exc_upsert(t, lo, hi, 0, 0, true, 0, false); return; }
        uint32_t slot = table_find_or_insert(t, lo, hi, false);
        if (slot != SLOT_NONE) __hip_atomic_store(&t.slots[slot].val, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        table_upsert_set(t, lo, hi, a, 0u);
    }
}

__global__ __launch_bounds__(256) void lookup_kernel(TableView t, const uint64_t *lo, const uint64_t *hi, uint64_t n, uint32_t *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t v;
    out[i] = table_lookup(t, lo[i], hi[i], v) ? v : 0u;
}

// small-contig test of IndexKminmerFunctor (graph/CreateMdbg.hpp:1330; getAbundance(0, .) :988-1010): one thread per unitig
__global__ __launch_bounds__(256) void small_contig_kernel(const uint64_t *off, uint32_t n_unitigs, const uint32_t *mins, uint32_t k,
                                                           uint32_t k_prev, TableView prev, uint8_t *flags) {
    uint32_t u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u >= n_unitigs) return;
    const uint64_t b = off[u], n = off[u + 1] - b;
    uint8_t f = 0;
    if (n < k && n >= k_prev) {
        const uint32_t windows = (n - k_prev + 1) < 2 ? 1u : 2u;     // prev[0], or min(prev[0], prev[1])
        uint32_t a = 0xFFFFFFFFu;
        for (uint32_t i = 0; i < windows; i++) {
            uint64_t hi, lo;
            window_hash(mins + b + i, k_prev, hi, lo);
            uint32_t v;
            if (!table_lookup(prev, lo, hi, v)) v = 1u;               // getPrevAbundances: missing => 1 (:1240-1265)
            a = v < a ? v : a;
        }
        f = a > 1u;
    }
    flags[u] = f;
}

// EdgeIndexer::partitionNode (graph/CreateMdbg.hpp:4083-4100): prefix and suffix identities of one node
__global__ __launch_bounds__(256) void edge_insert_kernel(const uint32_t *vecs, uint64_t n, uint32_t k, TableView t) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *v = vecs + i * k;
    uint64_t hi, lo;
    window_hash(v, k - 1, hi, lo);
    table_upsert_count(t, lo, hi, 0u, 0u);
    window_hash(v + 1, k - 1, hi, lo);
    table_upsert_count(t, lo, hi, 0u, 0u);
}

// UnitigEdgeIndexer::partitionUnitig (graph/CreateMdbg.hpp:4352-4389): the same two identities for the first and
// the last k-min-mer of every unitig (the last one skipped when it is the first).  The reference normalises the node
// before taking prefix and suffix; the pair {prefix, suffix} of a vector and of its reverse are each other's reverses,
// and every identity is taken on the normalised (k-1)-vector, so the orientation of the node does not matter.
__global__ __launch_bounds__(256) void unitig_edge_insert_kernel(const uint64_t *off, uint32_t n_unitigs, const uint32_t *mins, uint32_t k, TableView t) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_unitigs) return;
    const uint64_t f = off[i], n = off[i + 1] - f;
    if (n < k) return;
    uint64_t hi, lo;
    const uint32_t *v = mins + f;
    window_hash(v, k - 1, hi, lo);
    table_upsert_count(t, lo, hi, 0u, 0u);
    window_hash(v + 1, k - 1, hi, lo);
    table_upsert_count(t, lo, hi, 0u, 0u);
    if (n == k) return;
    v = mins + f + (n - k);
    window_hash(v, k - 1, hi, lo);
    table_upsert_count(t, lo, hi, 0u, 0u);
    window_hash(v + 1, k - 1, hi, lo);
    table_upsert_count(t, lo, hi, 0u, 0u);
}

__global__ __launch_bounds__(256) void sum_u64_kernel(const uint64_t *v, uint64_t n, unsigned long long *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t x = i < n ? v[i] : 0;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) x += __shfl_xor(x, d, 64);
    if ((threadIdx.x & 63u) == 0 && x) atomicAdd(out, (unsigned long long)x);
}

// out[0] += sum abundance * lo (the reference's "Checksum kminmer abundance"), out[1] += sum abundance, out[2] += sum hi,
// out[3] += sum over rows with a vector of (v[0] + 3 v[1] + 5 v[2] ...) * (lo | 1): ties the vectors to their keys
__global__ __launch_bounds__(256) void table_checksum_kernel(const uint64_t *lo, const uint64_t *hi, const uint32_t *ab, const uint32_t *vec,
                                                             uint32_t k, uint64_t n, unsigned long long *out) {
    uint64_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t l = lo[i], a = ab[i];
        s0 += a * l; s1 += a; s2 += hi[i];
        if (vec) {
            uint64_t w = 0;
            for (uint32_t j = 0; j < k; j++) w += (uint64_t)vec[i * k + j] * (2u * j + 1u);
            s3 += w * (l | 1ull);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        s0 += __shfl_xor(s0, d, 64); s1 += __shfl_xor(s1, d, 64); s2 += __shfl_xor(s2, d, 64); s3 += __shfl_xor(s3, d, 64);
    }
    if ((threadIdx.x & 63u) == 0) {
        atomicAdd(out + 0, (unsigned long long)s0); atomicAdd(out + 1, (unsigned long long)s1);
        atomicAdd(out + 2, (unsigned long long)s2); atomicAdd(out + 3, (unsigned long long)s3);
    }
}

__global__ void interleave_keys_kernel(const uint64_t *lo, const uint64_t *hi, uint64_t n, uint64_t *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out[2 * i] = lo[i]; out[2 * i + 1] = hi[i]; }
}

__global__ void unpack_records_kernel(const uint8_t *rec, uint64_t n, uint64_t *lo, uint64_t *hi, uint32_t *ab) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(rec) + 5 * i;      // (records are 20 bytes from a 256-byte aligned start)
    lo[i] = (uint64_t)p[0] | ((uint64_t)p[1] << 32); hi[i] = (uint64_t)p[2] | ((uint64_t)p[3] << 32); ab[i] = p[4];
}

__global__ void pack_records_kernel(const uint64_t *lo, const uint64_t *hi, const uint32_t *ab, uint64_t n, uint8_t *rec) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint8_t *p = rec + 20 * i;
    uint64_t l = lo[i], h = hi[i]; uint32_t a = ab[i];
    for (int b = 0; b < 8; b++) { p[b] = (uint8_t)(l >> (8 * b)); p[8 + b] = (uint8_t)(h >> (8 * b)); }
    for (int b = 0; b < 4; b++) p[16 + b] = (uint8_t)(a >> (8 * b));
}

// ---- host helpers ------------------------------------------------------------------------------------------
// Size the next table of this kind for the distinct keys just seen (+12.5 %): build_table_adaptive doubles that and
// rounds up to a power of two, i.e. 25-45 % load.
void update_key_hint(mdbg_ctx *ctx, int kind, uint64_t distinct, uint64_t instances) {
    if (instances) { ctx->key_ratio_hint[kind] = 1.125 * (double)distinct / (double)instances; ctx->key_ratio_known[kind] = true; }
}

struct InstIndex {
    DevBuf<uint64_t> off;   // n_reads + 1
    uint64_t total = 0;
};

static int build_inst_index(mdbg_ctx *ctx, const mdbg_minimizers *m, uint32_t k, InstIndex &ix) {
    DevBuf<uint32_t> cnt;
    MDBG_TRY(cnt.alloc(ctx, m->n_reads));
    MDBG_TRY(ix.off.alloc(ctx, (size_t)m->n_reads + 1));
    if (m->n_reads)
        hipLaunchKernelGGL(inst_count_kernel, dim3(grid_for(m->n_reads, 256)), dim3(256), 0, ctx->stream, m->d_off.p, m->n_reads, k, cnt.p);
    MDBG_TRY(exclusive_scan_u32(ctx, cnt.p, ix.off.p, m->n_reads));
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &ix.total, ix.off.p + m->n_reads, 8, hipMemcpyDeviceToHost));
    return MDBG_OK;
}

// dense counts + per-read rescue decision + row positions (only meaningful when min_abundance <= 1)
static int plan_rescue(mdbg_ctx *ctx, const DeviceTable &tab, const InstIndex &ix, uint32_t n_reads, const uint32_t *inst_slot,
                       RescuePlan &p) {
    TableView tv = tab.view();
    const uint64_t nslots = tab.cap + TABLE_EXC_CAP;
    const uint32_t m_star = rescue_m_star();
    MDBG_TRY(p.cnt.alloc(ctx, n_reads));
    MDBG_TRY(p.pos.alloc(ctx, (size_t)n_reads + 1));
    {
        LaunchTimer timer(ctx, "kminmer_rescue");
        if (!p.cls.p) {   // classes not produced by the caller's own pass over the table
            p.cap = tab.cap;
            MDBG_TRY(p.cls.alloc(ctx, (nslots + 3) / 4));
            hipLaunchKernelGGL(slot_flag_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, 0u, 0, (uint32_t *)nullptr,
                               p.cls.p, m_star);
        }
        unsigned blocks = grid_for((uint64_t)n_reads * 16, 256, (unsigned)ctx->n_cu * 16u);
        hipLaunchKernelGGL(rescue_count_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ix.off.p, n_reads, inst_slot, p.cls.p, tv, tab.cap,
                           m_star, p.cnt.p);
    }
    MDBG_TRY(exclusive_scan_u32(ctx, p.cnt.p, p.pos.p, n_reads));
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &p.total, p.pos.p + n_reads, 8, hipMemcpyDeviceToHost));
    return MDBG_OK;
}

static void launch_emit_rescued(mdbg_ctx *ctx, const SeqView &sv, uint32_t k, const uint32_t *inst_slot, const RescuePlan &p, const RowOut &ro,
                                uint64_t row_base) {
    unsigned blocks = grid_for((uint64_t)sv.n_reads * 16, 256, (unsigned)ctx->n_cu * 16u);
    hipLaunchKernelGGL(emit_rescued_kernel, dim3(blocks), dim3(256), 0, ctx->stream, sv, k, inst_slot, p.cls.p, p.cap, p.cnt.p, p.pos.p, ro, row_base);
}

static SeqView make_view(const mdbg_minimizers *m, const InstIndex &ix) {
    SeqView v;
    v.mins = m ? m->d_min.p : nullptr;
    v.off = m ? m->d_off.p : nullptr;
    v.inst_off = ix.off.p;
    v.n_reads = m ? m->n_reads : 0;
    v.n_inst = ix.total;
    v.n_min = m ? m->n_min : 0;
    return v;
}

int alloc_rows(mdbg_ctx *ctx, mdbg_table *t, uint64_t n, bool vec) {
    MDBG_TRY(t->d_lo.alloc(ctx, n));
    MDBG_TRY(t->d_hi.alloc(ctx, n));
    MDBG_TRY(t->d_ab.alloc(ctx, n));
    if (vec) MDBG_TRY(t->d_vec.alloc(ctx, n * t->k));
    t->n_records = n;
    t->has_vectors = vec;
    return MDBG_OK;
}

}  // namespace mdbg

using namespace mdbg;

static int check_seq(mdbg_ctx *ctx, const mdbg_minimizers *m, const char *who) {
    if (!m) return set_error(ctx, MDBG_EINVAL, "%s: null sequence set", who);
    if (m->n_min >= (1ull << 32)) return set_error(ctx, MDBG_ERANGE, "%s: more than 2^32 minimizers in one batch", who);
    return ensure_canonical(ctx, m);      // a scan output handed over without the purge in between
}

extern "C" int mdbg_kminmer_count_first(mdbg_ctx *ctx, const mdbg_minimizers *reads, uint32_t k, uint32_t min_abundance,
                                        mdbg_table **out) try {
    if (!ctx || !out || k < 2) return set_error(ctx, MDBG_EINVAL, "mdbg_kminmer_count_first: bad argument");
    MDBG_TRY(check_seq(ctx, reads, "mdbg_kminmer_count_first"));
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // instances partitioned by key and counted in LDS (partition.hip) -- by default from a few million minimizers up, where the one
    // table below no longer sits in a cache; it hands the input back when it is not for it (k > 32, more keys than its buckets hold)
    if (ctx->first_pass_mode == 2 || (ctx->first_pass_mode == 0 && reads->n_min >= ctx->part_auto_min)) {
        bool done = false;
        MDBG_TRY(count_first_partitioned(ctx, reads, k, min_abundance, out, &done));
        if (done) return MDBG_OK;
    }
    InstIndex ix;
    MDBG_TRY(build_inst_index(ctx, reads, k, ix));
    const uint64_t I = ix.total;
    SeqView sv = make_view(reads, ix), none{};
    DeviceTable tab;
    DevBuf<uint32_t> inst_slot, sflag;
    MDBG_TRY(inst_slot.alloc(ctx, I));
    for (uint64_t &v : ctx->part_info) v = 0;
    ctx->part_info[0] = 1; ctx->part_info[7] = I;
    // distinct keys are usually a small fraction of the instances (coverage): start small so the table
    // stays cache-resident, grow and rebuild when a probe sequence gets long
    MDBG_TRY(build_table_adaptive(ctx, tab, (uint64_t)((double)I * ctx->key_ratio_hint[0]), I, [&](TableView v) {
        if (I) {
            LaunchTimer timer(ctx, "kminmer_insert");
            hipLaunchKernelGGL(count_insert_kernel, dim3(instance_grid(ctx, sv.n_reads)), dim3(256), 0, ctx->stream, sv, k, v, inst_slot.p, (uint64_t)0);
        }
        return MDBG_OK;
    }, ctx->key_ratio_known[0]));
    TableView tv = tab.view();

    // solid rows
    const uint64_t nslots = tab.cap + TABLE_EXC_CAP;
    DevBuf<uint64_t> spos;
    MDBG_TRY(sflag.alloc(ctx, nslots));
    MDBG_TRY(spos.alloc(ctx, nslots + 1));
    // rescue (graph/CreateMdbg.cpp:317-319: only when min_abundance <= 1) wants the count class of every slot:
    // produced by the same pass over the table that flags the solid slots
    RescuePlan plan;
    const bool do_rescue = min_abundance <= 1 && I;
    if (do_rescue) { plan.cap = tab.cap; MDBG_TRY(plan.cls.alloc(ctx, (nslots + 3) / 4)); }
    {
        LaunchTimer timer(ctx, "kminmer_emit");
        hipLaunchKernelGGL(slot_flag_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, min_abundance, 0, sflag.p,
                           plan.cls.p, rescue_m_star(), tv.occ);
    }
    MDBG_TRY(exclusive_scan_u32(ctx, sflag.p, spos.p, nslots));
    uint64_t n_solid = 0, n_resc = 0, n_keys = 0;
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &n_solid, spos.p + nslots, 8, hipMemcpyDeviceToHost));
    MDBG_TRY(tab.occupied(ctx, &n_keys));
    update_key_hint(ctx, 0, n_keys, I);

    if (do_rescue) {
        MDBG_TRY(plan_rescue(ctx, tab, ix, reads->n_reads, inst_slot.p, plan));
        n_resc = plan.total;
    }

    mdbg_table *t = new mdbg_table();
    t->k = k;
    t->n_solid = n_solid;
    t->st_minimizers = reads->n_min; t->st_instances = I; t->st_keys = n_keys; t->st_slots = nslots;
    int rc = alloc_rows(ctx, t, n_solid + n_resc, true);
    if (rc) { delete t; return rc; }
    RowOut ro{t->d_lo.p, t->d_hi.p, t->d_ab.p, t->d_vec.p, k};
    {
        LaunchTimer timer(ctx, "kminmer_emit");
        hipLaunchKernelGGL(emit_slots_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, sflag.p, spos.p, sv, none, ro, (uint64_t)0);
        if (n_resc) launch_emit_rescued(ctx, sv, k, inst_slot.p, plan, ro, n_solid);
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { delete t; return set_error(ctx, MDBG_EHIP, "kminmer_count_first failed: %s", hipGetErrorString(e)); }
    *out = t;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

// build a lookup DeviceTable from the rows of `t` (abundance == 1 skipped as loadRefinedAbundances does)
static int ensure_lookup(mdbg_ctx *ctx, mdbg_table *t, bool skip_one) {
    if (t->lookup) return MDBG_OK;
    std::unique_ptr<DeviceTable> tab(new DeviceTable());
    MDBG_TRY(tab->init(ctx, table_slots_for(t->n_records)));
    if (t->n_records)
        hipLaunchKernelGGL(rows_insert_kernel, dim3(grid_for(t->n_records, 256)), dim3(256), 0, ctx->stream,
                           t->d_lo.p, t->d_hi.p, t->d_ab.p, t->n_records, skip_one ? 1 : 0, tab->view());
    MDBG_HIP_CHECK(ctx, hipGetLastError());
    MDBG_TRY(tab->check_overflow(ctx));
    t->lookup = std::move(tab);
    return MDBG_OK;
}

extern "C" int mdbg_prev_from_records(mdbg_ctx *ctx, const uint8_t *records20, uint64_t n_records, mdbg_table **out) try {
    if (!ctx || !out || (n_records && !records20)) return set_error(ctx, MDBG_EINVAL, "mdbg_prev_from_records: bad argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    mdbg_table *t = new mdbg_table();
    auto fail = [&](int rc) { delete t; return rc; };
    int rc;
    DevBuf<uint8_t> d_rec;
    if ((rc = d_rec.alloc(ctx, n_records * 20)) || (rc = alloc_rows(ctx, t, n_records, false))) return fail(rc);
    if (n_records) {
        hipError_t e = hipMemcpyAsync(d_rec.p, records20, n_records * 20, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "record upload failed: %s", hipGetErrorString(e)));
        hipLaunchKernelGGL(unpack_records_kernel, dim3(grid_for(n_records, 256)), dim3(256), 0, ctx->stream, d_rec.p, n_records,
                           t->d_lo.p, t->d_hi.p, t->d_ab.p);
    }
    // headroom: the unitig overlay may add keys that are not in the record file
    std::unique_ptr<DeviceTable> tab(new DeviceTable());
    if ((rc = tab->init(ctx, table_slots_for(n_records + n_records / 4) + 4096))) return fail(rc);
    if (n_records)
        hipLaunchKernelGGL(rows_insert_kernel, dim3(grid_for(n_records, 256)), dim3(256), 0, ctx->stream,
                           t->d_lo.p, t->d_hi.p, t->d_ab.p, n_records, 1, tab->view());
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "prev table build failed: %s", hipGetErrorString(e)));
    if ((rc = tab->check_overflow(ctx))) return fail(rc);
    t->lookup = std::move(tab);
    *out = t;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

namespace mdbg { int bytes_ready_on(mdbg_ctx *ctx, const mdbg_bytes *b); }

extern "C" int mdbg_prev_from_record_bytes(mdbg_ctx *ctx, const mdbg_bytes *records, uint64_t n_records, mdbg_table **out) try {
    if (!ctx || !out || !records) return set_error(ctx, MDBG_EINVAL, "mdbg_prev_from_record_bytes: bad argument");
    if (n_records * 20 != records->n) return set_error(ctx, MDBG_EINVAL, "mdbg_prev_from_record_bytes: %llu records are not the buffer's %llu bytes",
                                                       (unsigned long long)n_records, (unsigned long long)records->n);
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<mdbg_table> t(new mdbg_table());
    MDBG_TRY(alloc_rows(ctx, t.get(), n_records, false));
    MDBG_TRY(bytes_ready_on(ctx, records));
    if (n_records)
        hipLaunchKernelGGL(unpack_records_kernel, dim3(grid_for(n_records, 256)), dim3(256), 0, ctx->stream, records->d.p, n_records,
                           t->d_lo.p, t->d_hi.p, t->d_ab.p);
    std::unique_ptr<DeviceTable> tab(new DeviceTable());
    MDBG_TRY(tab->init(ctx, table_slots_for(n_records + n_records / 4) + 4096));      // headroom: the unitig overlay may add keys
    if (n_records)
        hipLaunchKernelGGL(rows_insert_kernel, dim3(grid_for(n_records, 256)), dim3(256), 0, ctx->stream,
                           t->d_lo.p, t->d_hi.p, t->d_ab.p, n_records, 1, tab->view());
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    MDBG_TRY(tab->check_overflow(ctx));
    t->lookup = std::move(tab);
    *out = t.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_prev_overlay_unitigs(mdbg_ctx *ctx, mdbg_table *prev, const mdbg_minimizers *unitigs,
                                         const uint32_t *abundance, uint32_t k_prev) try {
    if (!ctx || !prev || !prev->lookup || !abundance || k_prev < 2) return set_error(ctx, MDBG_EINVAL, "mdbg_prev_overlay_unitigs: bad argument");
    MDBG_TRY(check_seq(ctx, unitigs, "mdbg_prev_overlay_unitigs"));
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    InstIndex ix;
    MDBG_TRY(build_inst_index(ctx, unitigs, k_prev, ix));
    if (prev->lookup->cap < (prev->n_records + ix.total) * 3 / 2)
        return set_error(ctx, MDBG_ERANGE, "prev table too small for the unitig overlay (%llu windows)", (unsigned long long)ix.total);
    DevBuf<uint32_t> d_ab;
    MDBG_TRY(d_ab.alloc(ctx, unitigs->n_reads));
    if (unitigs->n_reads) MDBG_HIP_CHECK(ctx, hipMemcpyAsync(d_ab.p, abundance, (size_t)unitigs->n_reads * 4, hipMemcpyHostToDevice, ctx->stream));
    SeqView sv = make_view(unitigs, ix);
    if (ix.total)
        hipLaunchKernelGGL(overlay_kernel, dim3(grid_for(ix.total, 256)), dim3(256), 0, ctx->stream, sv, k_prev, d_ab.p, prev->lookup->view());
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    prev->image.reset();                 // (made again, from the overlaid table, by the pass that reads it)
    return prev->lookup->check_overflow(ctx);
} MDBG_API_CATCH(ctx)

static int prev_view(mdbg_ctx *ctx, const mdbg_table *prev, TableView &pv) {
    if (!prev) return set_error(ctx, MDBG_EINVAL, "previous table is null");
    MDBG_TRY(ensure_lookup(ctx, const_cast<mdbg_table *>(prev), true));
    pv = prev->lookup->view();
    return MDBG_OK;
}

// the previous table in its compact form: what the index pass that built it left behind, or made here -- from the one-slot table
// when there is one (mdbg_prev_from_records, perhaps overlaid: its values, zeros included), else from the rows (abundance 1 skipped)
static int prev_image(mdbg_ctx *ctx, const mdbg_table *prev_c, BucketView &pv) {
    if (!prev_c) return set_error(ctx, MDBG_EINVAL, "previous table is null");
    mdbg_table *prev = const_cast<mdbg_table *>(prev_c);
    if (!prev->image) {
        std::unique_ptr<BucketTable> img(new BucketTable());
        if (prev->lookup) {
            DeviceTable &src = *prev->lookup;
            const uint64_t nslots = src.cap + TABLE_EXC_CAP;
            MDBG_HIP_CHECK(ctx, hipMemsetAsync(src.ctl.p + 4, 0, TABLE_OCC_WAYS * 4, ctx->stream));
            hipLaunchKernelGGL(table_occupied_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, src.view(), src.cap, src.view().occ);
            uint64_t n_keys = 0;
            MDBG_TRY(src.occupied(ctx, &n_keys));
            for (double load = 0.75;; load *= 0.5) {
                MDBG_TRY(img->init(ctx, BucketTable::buckets_for(n_keys + 64, load), false));
                LaunchTimer timer(ctx, "kminmer_prev_image");
                hipLaunchKernelGGL(bucket_from_table_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, src.view(), src.cap, img->view());
                const int o = img->overflowed(ctx);
                if (o < 0) return o;
                if (!o) break;
                if (load < 0.05) return set_error(ctx, MDBG_ERANGE, "previous table: bucket table overflow");
            }
        } else {
            for (double load = 0.75;; load *= 0.5) {
                MDBG_TRY(img->init(ctx, BucketTable::buckets_for(prev->n_records + 64, load), false));
                if (prev->n_records) {
                    LaunchTimer timer(ctx, "kminmer_prev_image");
                    hipLaunchKernelGGL(bucket_rows_insert_kernel, dim3(grid_for(prev->n_records, 256)), dim3(256), 0, ctx->stream,
                                       prev->d_lo.p, prev->d_hi.p, prev->d_ab.p, prev->n_records, 1, img->view());
                }
                const int o = img->overflowed(ctx);
                if (o < 0) return o;
                if (!o) break;
                if (load < 0.05) return set_error(ctx, MDBG_ERANGE, "previous table: bucket table overflow");
            }
        }
        prev->image = std::move(img);
    }
    pv = prev->image->view();
    return MDBG_OK;
}

// rows of a bucket table into a new mdbg_table (every occupied entry is a row); with vectors when the table kept representatives
static int rows_from_buckets(mdbg_ctx *ctx, BucketTable &tab, uint32_t k, const SeqView &a, const SeqView &b, bool vectors, mdbg_table **out) {
    BucketView bv = tab.view();
    const uint64_t n_ent = tab.entries() + TABLE_EXC_CAP;
    DevBuf<uint32_t> flag;
    DevBuf<uint64_t> pos;
    MDBG_TRY(flag.alloc(ctx, n_ent));
    MDBG_TRY(pos.alloc(ctx, n_ent + 1));
    {
        LaunchTimer timer(ctx, "kminmer_emit");
        hipLaunchKernelGGL(bucket_flag_kernel, dim3(grid_for(n_ent, 256)), dim3(256), 0, ctx->stream, bv, flag.p);
    }
    MDBG_TRY(exclusive_scan_u32(ctx, flag.p, pos.p, n_ent));
    uint64_t n_rows = 0;
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &n_rows, pos.p + n_ent, 8, hipMemcpyDeviceToHost));
    std::unique_ptr<mdbg_table> t(new mdbg_table());
    t->k = k;
    t->n_solid = n_rows;
    MDBG_TRY(alloc_rows(ctx, t.get(), n_rows, vectors));
    RowOut ro{t->d_lo.p, t->d_hi.p, t->d_ab.p, vectors ? t->d_vec.p : nullptr, k};
    {
        LaunchTimer timer(ctx, "kminmer_emit");
        hipLaunchKernelGGL(bucket_emit_kernel, dim3(grid_for(n_ent, 256)), dim3(256), 0, ctx->stream, bv, flag.p, pos.p, a, b, ro);
    }
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *out = t.release();
    return MDBG_OK;
}

static int index_pass_buckets(mdbg_ctx *ctx, const mdbg_minimizers *reads, const mdbg_minimizers *unitigs, uint32_t k, const mdbg_table *prev,
                              bool vectors, int hint_kind, mdbg_table **out);

// Which form a pass above firstK takes ("index_tuning" bit 7: insert first, look up only where needed -- index_lazy_kernel; bit 8: never;
// neither: a sample of the sequences decides).  The insert-first form pays 1 + f (1 + 1 / k) random sectors per instance where the two
// kernels pay 2 - f, f = the fraction of windows that are never inserted: better below f = 0.46 -- HiFi reads stay below to k = 26 and
// beyond (profiles/round6_w_*, round6_y_*), ONT reads at 2 % errors do not.
static int choose_insert_first(mdbg_ctx *ctx, const SeqView &vk, uint32_t k, const TableView &pv, bool &lazy) {
    lazy = (ctx->index_tuning & 128u) != 0;
    if (lazy || (ctx->index_tuning & 256u) || (ctx->index_tuning & 8u) || (ctx->index_tuning & 15u) != 3u || !vk.n_inst) return MDBG_OK;
    DevBuf<unsigned long long> st;
    MDBG_TRY(st.alloc(ctx, 2));
    MDBG_HIP_CHECK(ctx, hipMemsetAsync(st.p, 0, 16, ctx->stream));
    const uint32_t step = vk.n_reads > 32768u ? vk.n_reads / 16384u : 1u;
    {
        LaunchTimer timer(ctx, "kminmer_prev_lookup");
        hipLaunchKernelGGL(index_miss_sample_kernel, dim3(instance_grid(ctx, (vk.n_reads + step - 1) / step)), dim3(256), 0, ctx->stream, vk, k, pv, step, st.p);
    }
    unsigned long long h[2] = {0, 0};
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, h, st.p, 16, hipMemcpyDeviceToHost));
    lazy = h[0] > 0 && (double)h[1] < 0.45 * (double)h[0];
    ctx->index_last_miss_fraction = h[0] ? (double)h[1] / (double)h[0] : -1.0;
    return MDBG_OK;
}

extern "C" int mdbg_kminmer_count_refined(mdbg_ctx *ctx, const mdbg_minimizers *reads, const mdbg_minimizers *unitigs,
                                          uint32_t k, const mdbg_table *prev, mdbg_table **out) try {
    if (!ctx || !out || k < 3) return set_error(ctx, MDBG_EINVAL, "mdbg_kminmer_count_refined: bad argument");
    MDBG_TRY(check_seq(ctx, reads, "mdbg_kminmer_count_refined"));
    if (unitigs) MDBG_TRY(check_seq(ctx, unitigs, "mdbg_kminmer_count_refined"));
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (ctx->refined_form == 0 && ctx->index_table_form == 0) return index_pass_buckets(ctx, reads, unitigs, k, prev, true, 1, out);
    TableView pv;
    MDBG_TRY(prev_view(ctx, prev, pv));
    InstIndex ia, ib;
    MDBG_TRY(build_inst_index(ctx, reads, k, ia));
    if (unitigs) MDBG_TRY(build_inst_index(ctx, unitigs, k, ib));
    SeqView a = make_view(reads, ia), b = unitigs ? make_view(unitigs, ib) : SeqView{};
    const uint64_t I = ia.total + ib.total;
    if (I >= (1ull << 32) || a.n_min + b.n_min >= (1ull << 32))
        return set_error(ctx, MDBG_ERANGE, "more than 2^32 minimizers / k-min-mer instances in one call");
    // round 6: the refined pass insert-first like the index passes when the sample says so -- only the keys that are KEPT enter the table
    // (a third of the distinct ones at 50 x HiFi), with their final value and the instance that names their vector; no walk over the
    // slots to refine them afterwards
    bool lazy = false;
    MDBG_TRY(choose_insert_first(ctx, a, k, pv, lazy));
    DeviceTable tab;
    MDBG_TRY(build_table_adaptive(ctx, tab, (uint64_t)((double)I * ctx->key_ratio_hint[1]), I, [&](TableView v) {
        LaunchTimer timer(ctx, "kminmer_insert");
        if (lazy) {
            if (a.n_inst) hipLaunchKernelGGL((index_lazy_kernel<1, true>), dim3(instance_grid(ctx, a.n_reads)), dim3(256), 0, ctx->stream, a, k, pv, v, (unsigned long long *)nullptr, (uint64_t)0);
            if (b.n_inst) hipLaunchKernelGGL((index_lazy_kernel<1, true>), dim3(instance_grid(ctx, b.n_reads)), dim3(256), 0, ctx->stream, b, k, pv, v, (unsigned long long *)nullptr, (uint64_t)a.n_min);
            return MDBG_OK;
        }
        // ("index_tuning" bits 1 and 2, as in the index passes: the plain-load first look, two windows of a lane in flight)
        const bool fast = ctx->index_tuning & 2u, two = ctx->index_tuning & 4u;
        auto launch = [&](const SeqView &sv, uint64_t rep_base) {
            const dim3 grid(instance_grid(ctx, sv.n_reads)), block(256);
            if (!(fast || two)) hipLaunchKernelGGL(distinct_insert_kernel, grid, block, 0, ctx->stream, sv, k, v, rep_base);
            else if (two && fast) hipLaunchKernelGGL((distinct_insert_u_kernel<2, true>), grid, block, 0, ctx->stream, sv, k, v, rep_base);
            else if (two) hipLaunchKernelGGL((distinct_insert_u_kernel<2, false>), grid, block, 0, ctx->stream, sv, k, v, rep_base);
            else hipLaunchKernelGGL((distinct_insert_u_kernel<1, true>), grid, block, 0, ctx->stream, sv, k, v, rep_base);
        };
        if (a.n_inst) launch(a, 0);
        if (b.n_inst) launch(b, a.n_min);
        return MDBG_OK;
    }, ctx->key_ratio_known[1]));
    TableView tv = tab.view();
    const uint64_t nslots = tab.cap + TABLE_EXC_CAP;
    DevBuf<uint32_t> sflag;
    DevBuf<uint64_t> spos;
    MDBG_TRY(sflag.alloc(ctx, nslots));
    MDBG_TRY(spos.alloc(ctx, nslots + 1));
    {
        LaunchTimer timer(ctx, "kminmer_emit");
        if (!lazy) hipLaunchKernelGGL(refine_slots_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, a, b, k, pv);
        hipLaunchKernelGGL(slot_flag_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, 0u, 1, sflag.p,
                           (uint8_t *)nullptr, 0u, tv.occ);
    }
    MDBG_TRY(exclusive_scan_u32(ctx, sflag.p, spos.p, nslots));
    uint64_t n_keys = 0;
    MDBG_TRY(tab.occupied(ctx, &n_keys));
    update_key_hint(ctx, 1, n_keys, I);
    uint64_t n_rows = 0;
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &n_rows, spos.p + nslots, 8, hipMemcpyDeviceToHost));
    mdbg_table *t = new mdbg_table();
    t->k = k;
    t->n_solid = n_rows;
    t->st_minimizers = a.n_min + b.n_min; t->st_instances = I; t->st_keys = n_keys; t->st_slots = nslots;
    int rc = alloc_rows(ctx, t, n_rows, true);
    if (rc) { delete t; return rc; }
    RowOut ro{t->d_lo.p, t->d_hi.p, t->d_ab.p, t->d_vec.p, k};
    {
        LaunchTimer timer(ctx, "kminmer_emit");
        hipLaunchKernelGGL(emit_slots_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, sflag.p, spos.p, a, b, ro, (uint64_t)0);
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { delete t; return set_error(ctx, MDBG_EHIP, "kminmer_count_refined failed: %s", hipGetErrorString(e)); }
    *out = t;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

static int index_one_set(mdbg_ctx *ctx, const mdbg_minimizers *s, uint32_t k, const TableView &pv, const TableView &tv, uint64_t &n_inst) {
    InstIndex ik, ikm1;
    MDBG_TRY(build_inst_index(ctx, s, k, ik));
    n_inst += ik.total;
    if (!ik.total) return MDBG_OK;
    // "index_tuning" (A/B): bit 0 a slot's key and value in one trip, bit 1 the plain-load first look of the insert, bit 2 two windows of
    // a lane in flight, bit 3 look-up and insert in one kernel
    const bool wide = ctx->index_tuning & 1u, fast = ctx->index_tuning & 2u, two = ctx->index_tuning & 4u, fused = ctx->index_tuning & 8u;
    bool lazy = false;
    if (!fused) MDBG_TRY(choose_insert_first(ctx, make_view(s, ik), k, pv, lazy));
    if (lazy) {
        SeqView vk = make_view(s, ik);
        LaunchTimer timer(ctx, "kminmer_insert");
        const dim3 grid(instance_grid(ctx, vk.n_reads)), block(256);
        static const int lazy_u = getenv("MDBG_INDEX_LAZY_U") ? atoi(getenv("MDBG_INDEX_LAZY_U")) : 1;      // A/B: chunks of a sequence in flight per group
        static const bool lazy_stats = getenv("MDBG_INDEX_LAZY_STATS") != nullptr;                             // prints k-windows looked at / not seen at the first look
        DevBuf<unsigned long long> st;
        if (lazy_stats) { MDBG_TRY(st.alloc(ctx, 2)); MDBG_HIP_CHECK(ctx, hipMemsetAsync(st.p, 0, 16, ctx->stream)); }
        if (lazy_u >= 3) hipLaunchKernelGGL(index_lazy_kernel<3>, grid, block, 0, ctx->stream, vk, k, pv, tv, st.p);
        else if (lazy_u == 2) hipLaunchKernelGGL(index_lazy_kernel<2>, grid, block, 0, ctx->stream, vk, k, pv, tv, st.p);
        else hipLaunchKernelGGL(index_lazy_kernel<1>, grid, block, 0, ctx->stream, vk, k, pv, tv, st.p);
        MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (lazy_stats) {
            unsigned long long h[2] = {0, 0};
            MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, h, st.p, 16, hipMemcpyDeviceToHost));
            fprintf(stderr, "[mdbg] index pass k = %u, insert first: %llu k-windows, %llu not seen at the first look (%.3f); sample said %.3f never inserted\n", k, h[0], h[1],
                    h[0] ? (double)h[1] / (double)h[0] : 0.0, ctx->index_last_miss_fraction);
        }
        return MDBG_OK;
    }
    if (fused) {
        SeqView vk = make_view(s, ik);
        LaunchTimer timer(ctx, "kminmer_insert");
        const dim3 grid(instance_grid(ctx, vk.n_reads)), block(256);
        if (fast) hipLaunchKernelGGL(index_fused_kernel<true>, grid, block, 0, ctx->stream, vk, k, pv, tv);
        else hipLaunchKernelGGL(index_fused_kernel<false>, grid, block, 0, ctx->stream, vk, k, pv, tv);
        MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        return MDBG_OK;
    }
    MDBG_TRY(build_inst_index(ctx, s, k - 1, ikm1));
    DevBuf<uint32_t> prev_ab;
    MDBG_TRY(prev_ab.alloc(ctx, ikm1.total));
    SeqView vk = make_view(s, ik), vkm1 = make_view(s, ikm1);
    // round 6 -- bit 4: the look-up fetches both slots of the home sector at once, bit 5: 32 lanes a sequence instead of 16, bit 6: the
    // insert's first look at both slots (measured: the look-up gains, the insert does not -- default 16 | 3)
    const bool pair = ctx->index_tuning & 16u, l32 = ctx->index_tuning & 32u, pair_insert = ctx->index_tuning & 64u;
    {
        LaunchTimer timer(ctx, "kminmer_prev_lookup");
        const dim3 grid(instance_grid(ctx, vkm1.n_reads, pair && l32 ? 32 : 16)), block(256);
        if (pair && l32) hipLaunchKernelGGL(prev_abundance_pair_kernel<32>, grid, block, 0, ctx->stream, vkm1, k - 1, pv, prev_ab.p);
        else if (pair) hipLaunchKernelGGL(prev_abundance_pair_kernel<16>, grid, block, 0, ctx->stream, vkm1, k - 1, pv, prev_ab.p);
        else if (!(ctx->index_tuning & 15u)) hipLaunchKernelGGL(prev_abundance_kernel<TableView>, grid, block, 0, ctx->stream, vkm1, k - 1, pv, prev_ab.p);
        else if (two && wide) hipLaunchKernelGGL((prev_abundance_u_kernel<2, true>), grid, block, 0, ctx->stream, vkm1, k - 1, pv, prev_ab.p);
        else if (two) hipLaunchKernelGGL((prev_abundance_u_kernel<2, false>), grid, block, 0, ctx->stream, vkm1, k - 1, pv, prev_ab.p);
        else if (wide) hipLaunchKernelGGL((prev_abundance_u_kernel<1, true>), grid, block, 0, ctx->stream, vkm1, k - 1, pv, prev_ab.p);
        else hipLaunchKernelGGL((prev_abundance_u_kernel<1, false>), grid, block, 0, ctx->stream, vkm1, k - 1, pv, prev_ab.p);
    }
    {
        LaunchTimer timer(ctx, "kminmer_insert");
        const dim3 grid(instance_grid(ctx, vk.n_reads, pair_insert && l32 ? 32 : 16)), block(256);
        if (pair_insert && l32) hipLaunchKernelGGL(index_insert_pair_kernel<32>, grid, block, 0, ctx->stream, vk, ikm1.off.p, prev_ab.p, k, tv);
        else if (pair_insert) hipLaunchKernelGGL(index_insert_pair_kernel<16>, grid, block, 0, ctx->stream, vk, ikm1.off.p, prev_ab.p, k, tv);
        else if (!(fast || two)) hipLaunchKernelGGL(index_insert_kernel, grid, block, 0, ctx->stream, vk, ikm1.off.p, prev_ab.p, k, tv);
        else if (two && fast) hipLaunchKernelGGL((index_insert_u_kernel<2, true>), grid, block, 0, ctx->stream, vk, ikm1.off.p, prev_ab.p, k, tv);
        else if (two) hipLaunchKernelGGL((index_insert_u_kernel<2, false>), grid, block, 0, ctx->stream, vk, ikm1.off.p, prev_ab.p, k, tv);
        else hipLaunchKernelGGL((index_insert_u_kernel<1, true>), grid, block, 0, ctx->stream, vk, ikm1.off.p, prev_ab.p, k, tv);
    }
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MDBG_OK;
}

// the same with both tables in the compact form; `inst` (optional) receives the k-instance index of the set (the refined pass needs the view again)
static int index_one_set_b(mdbg_ctx *ctx, const mdbg_minimizers *s, uint32_t k, const BucketView &pv, const BucketView &tv, uint64_t rep_base, uint64_t &n_inst) {
    InstIndex ik, ikm1;
    MDBG_TRY(build_inst_index(ctx, s, k, ik));
    n_inst += ik.total;
    if (!ik.total) return MDBG_OK;
    MDBG_TRY(build_inst_index(ctx, s, k - 1, ikm1));
    DevBuf<uint32_t> prev_ab;
    MDBG_TRY(prev_ab.alloc(ctx, ikm1.total));
    SeqView vk = make_view(s, ik), vkm1 = make_view(s, ikm1);
    {
        LaunchTimer timer(ctx, "kminmer_prev_lookup");
        hipLaunchKernelGGL(prev_abundance_kernel<BucketView>, dim3(instance_grid(ctx, vkm1.n_reads)), dim3(256), 0, ctx->stream, vkm1, k - 1, pv, prev_ab.p);
    }
    {
        LaunchTimer timer(ctx, "kminmer_insert");
        hipLaunchKernelGGL(index_insert_b_kernel, dim3(instance_grid(ctx, vk.n_reads)), dim3(256), 0, ctx->stream, vk, ikm1.off.p, prev_ab.p, k, tv, rep_base);
    }
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MDBG_OK;
}

// A pass above firstK over bucket tables (mdbg_kminmer_index; mdbg_kminmer_count_refined with vectors): per (k-1)-window one look-up of
// the previous table, per k-window whose abundance min(prev[i], prev[i+1]) is above 1 an insert-if-absent (graph/CreateMdbg.hpp:1240-1265,
// :1440-1459; for k = firstK + 1 the same numbers come out of KminmerCounter::getRefinedAbundance, :3933-4005: the minimum over the
// vector's two (k-1)-sub-min-mers, missing or 0 => 1, kept when above 1).  The table it fills stays with the result as its look-up image.
static int index_pass_buckets(mdbg_ctx *ctx, const mdbg_minimizers *reads, const mdbg_minimizers *unitigs, uint32_t k, const mdbg_table *prev,
                              bool vectors, int hint_kind, mdbg_table **out) {
    BucketView pv;
    MDBG_TRY(prev_image(ctx, prev, pv));
    const uint64_t bound = reads->n_min + (unitigs ? unitigs->n_min : 0);          // upper bound on distinct keys: total k-windows
    if (vectors && bound >= (1ull << 32)) return set_error(ctx, MDBG_ERANGE, "more than 2^32 minimizers in one call");
    std::unique_ptr<BucketTable> tab(new BucketTable());
    uint64_t n_inst = 0;
    MDBG_TRY(build_buckets_adaptive(ctx, *tab, (uint64_t)((double)bound * ctx->key_ratio_hint[hint_kind]), bound, vectors, [&](BucketView v) {
        n_inst = 0;
        MDBG_TRY(index_one_set_b(ctx, reads, k, pv, v, 0, n_inst));
        if (unitigs) MDBG_TRY(index_one_set_b(ctx, unitigs, k, pv, v, reads->n_min, n_inst));
        return MDBG_OK;
    }));
    SeqView a{}, b{};                    // (write_instance_vector only needs the minimizers of the views)
    if (vectors) {
        a.mins = reads->d_min.p; a.n_min = reads->n_min;
        if (unitigs) { b.mins = unitigs->d_min.p; b.n_min = unitigs->n_min; }
    }
    mdbg_table *t = nullptr;
    MDBG_TRY(rows_from_buckets(ctx, *tab, k, a, b, vectors, &t));
    update_key_hint(ctx, hint_kind, t->n_solid, bound);
    t->st_minimizers = bound; t->st_instances = n_inst; t->st_keys = t->n_solid; t->st_slots = tab->entries() + TABLE_EXC_CAP;
    tab->rep.release(); tab->with_rep = false;           // the image of the next pass needs keys and values only
    t->image = std::move(tab);
    *out = t;
    return MDBG_OK;
}

extern "C" int mdbg_kminmer_index(mdbg_ctx *ctx, const mdbg_minimizers *reads, const mdbg_minimizers *unitigs,
                                  uint32_t k, const mdbg_table *prev, mdbg_table **out) try {
    if (!ctx || !out || k < 3) return set_error(ctx, MDBG_EINVAL, "mdbg_kminmer_index: bad argument");
    MDBG_TRY(check_seq(ctx, reads, "mdbg_kminmer_index"));
    if (unitigs) MDBG_TRY(check_seq(ctx, unitigs, "mdbg_kminmer_index"));
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (ctx->index_table_form == 0) return index_pass_buckets(ctx, reads, unitigs, k, prev, false, 2, out);
    TableView pv;
    MDBG_TRY(prev_view(ctx, prev, pv));
    // upper bound on distinct keys: total k-windows
    uint64_t bound = reads->n_min + (unitigs ? unitigs->n_min : 0);
    std::unique_ptr<DeviceTable> tabp(new DeviceTable());
    DeviceTable &tab = *tabp;
    uint64_t n_inst = 0;
    MDBG_TRY(build_table_adaptive(ctx, tab, (uint64_t)((double)bound * ctx->key_ratio_hint[2]), bound, [&](TableView v) {
        n_inst = 0;
        MDBG_TRY(index_one_set(ctx, reads, k, pv, v, n_inst));
        if (unitigs) MDBG_TRY(index_one_set(ctx, unitigs, k, pv, v, n_inst));
        return MDBG_OK;
    }));
    TableView tv = tab.view();
    const uint64_t nslots = tab.cap + TABLE_EXC_CAP;
    DevBuf<uint32_t> sflag;
    DevBuf<uint64_t> spos;
    MDBG_TRY(sflag.alloc(ctx, nslots));
    MDBG_TRY(spos.alloc(ctx, nslots + 1));
    hipLaunchKernelGGL(slot_flag_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, 0u, 2, sflag.p);
    MDBG_TRY(exclusive_scan_u32(ctx, sflag.p, spos.p, nslots));
    uint64_t n_rows = 0;
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &n_rows, spos.p + nslots, 8, hipMemcpyDeviceToHost));
    update_key_hint(ctx, 2, n_rows, bound);   // every occupied slot is a row
    mdbg_table *t = new mdbg_table();
    t->k = k;
    t->n_solid = n_rows;
    t->st_minimizers = bound; t->st_instances = n_inst; t->st_keys = n_rows; t->st_slots = nslots;
    int rc = alloc_rows(ctx, t, n_rows, false);
    if (rc) { delete t; return rc; }
    RowOut ro{t->d_lo.p, t->d_hi.p, t->d_ab.p, nullptr, k};
    {
        LaunchTimer timer(ctx, "kminmer_emit");
        hipLaunchKernelGGL(emit_slots_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, sflag.p, spos.p, SeqView{}, SeqView{}, ro, (uint64_t)0);
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { delete t; return set_error(ctx, MDBG_EHIP, "kminmer_index failed: %s", hipGetErrorString(e)); }
    // The table the pass filled IS the key -> abundance map of its rows (every slot published, every value above 1): it stays with the
    // result as its look-up structure, so the next pass of the loop -- or mdbg_table_lookup -- does not build one from the rows again
    // (a clear of the slots and an insert per row: 1 ms of every pass at 10 M reads).  "keep_index_table" = 0: drop it (rounds 1 - 5).
    if (ctx->keep_index_table) t->lookup = std::move(tabp);
    *out = t;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_table_info(const mdbg_table *t, uint32_t *k, uint64_t *n_records, uint64_t *n_solid, int *has_vectors) {
    if (!t) return MDBG_EINVAL;
    if (k) *k = t->k;
    if (n_records) *n_records = t->n_records;
    if (n_solid) *n_solid = t->n_solid;
    if (has_vectors) *has_vectors = t->has_vectors ? 1 : 0;
    return MDBG_OK;
}

extern "C" int mdbg_table_stats(const mdbg_table *t, uint64_t stats[4]) {
    if (!t || !stats) return MDBG_EINVAL;
    stats[0] = t->st_minimizers; stats[1] = t->st_instances; stats[2] = t->st_keys; stats[3] = t->st_slots;
    return MDBG_OK;
}

extern "C" int mdbg_first_pass_info(const mdbg_ctx *ctx, uint64_t info[8]) {
    if (!ctx || !info) return MDBG_EINVAL;
    for (int i = 0; i < 8; i++) info[i] = ctx->part_info[i];
    return MDBG_OK;
}

extern "C" int mdbg_table_checksum(mdbg_ctx *ctx, const mdbg_table *t, uint64_t sums[4]) try {
    if (!ctx || !t || !sums) return set_error(ctx, MDBG_EINVAL, "mdbg_table_checksum: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    DevBuf<unsigned long long> acc;
    MDBG_TRY(acc.alloc(ctx, 4));
    MDBG_HIP_CHECK(ctx, hipMemsetAsync(acc.p, 0, 32, ctx->stream));
    if (t->n_records)
        hipLaunchKernelGGL(table_checksum_kernel, dim3(grid_for(t->n_records, 256 * 8, 4096)), dim3(256), 0, ctx->stream, t->d_lo.p, t->d_hi.p,
                           t->d_ab.p, t->has_vectors ? t->d_vec.p : nullptr, t->k, t->n_records, acc.p);
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, sums, acc.p, 32, hipMemcpyDeviceToHost));
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_table_to_host(mdbg_ctx *ctx, const mdbg_table *t, uint8_t *records20, uint32_t *vectors) try {
    if (!ctx || !t) return set_error(ctx, MDBG_EINVAL, "mdbg_table_to_host: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (records20 && t->n_records) {
        DevBuf<uint8_t> d_rec;
        MDBG_TRY(d_rec.alloc(ctx, t->n_records * 20));
        hipLaunchKernelGGL(pack_records_kernel, dim3(grid_for(t->n_records, 256)), dim3(256), 0, ctx->stream,
                           t->d_lo.p, t->d_hi.p, t->d_ab.p, t->n_records, d_rec.p);
        MDBG_HIP_CHECK(ctx, hipMemcpyAsync(records20, d_rec.p, t->n_records * 20, hipMemcpyDeviceToHost, ctx->stream));
        MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (vectors) {
        if (!t->has_vectors) return set_error(ctx, MDBG_EINVAL, "mdbg_table_to_host: table has no vectors (k >= firstK+2)");
        if (t->n_records) MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, vectors, t->d_vec.p, t->n_records * t->k * 4, hipMemcpyDeviceToHost));
    }
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_table_to_host_range(mdbg_ctx *ctx, const mdbg_table *t, uint64_t first, uint64_t count, uint8_t *records20, uint32_t *vectors) try {
    if (!ctx || !t) return set_error(ctx, MDBG_EINVAL, "mdbg_table_to_host_range: null argument");
    if (first > t->n_records || count > t->n_records - first) return set_error(ctx, MDBG_EINVAL, "mdbg_table_to_host_range: rows [%llu, +%llu) of %llu",
                                                                               (unsigned long long)first, (unsigned long long)count, (unsigned long long)t->n_records);
    if (vectors && !t->has_vectors) return set_error(ctx, MDBG_EINVAL, "mdbg_table_to_host_range: table has no vectors (k >= firstK+2)");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (!count) return MDBG_OK;
    DevBuf<uint8_t> d_rec;
    if (records20) {
        MDBG_TRY(d_rec.alloc(ctx, count * 20));
        hipLaunchKernelGGL(pack_records_kernel, dim3(grid_for(count, 256)), dim3(256), 0, ctx->stream, t->d_lo.p + first, t->d_hi.p + first, t->d_ab.p + first,
                           count, d_rec.p);
        MDBG_HIP_CHECK(ctx, hipMemcpyAsync(records20, d_rec.p, count * 20, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (vectors) MDBG_HIP_CHECK(ctx, hipMemcpyAsync(vectors, t->d_vec.p + first * t->k, count * t->k * 4, hipMemcpyDeviceToHost, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));      // one wait for both
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_table_lookup(mdbg_ctx *ctx, const mdbg_table *t, const uint64_t *hash_lo, const uint64_t *hash_hi,
                                 uint64_t n, uint32_t *abundance) try {
    if (!ctx || !t || (n && (!hash_lo || !hash_hi || !abundance))) return set_error(ctx, MDBG_EINVAL, "mdbg_table_lookup: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    MDBG_TRY(ensure_lookup(ctx, const_cast<mdbg_table *>(t), false));
    if (!n) return MDBG_OK;
    DevBuf<uint64_t> dl, dh;
    DevBuf<uint32_t> dv;
    MDBG_TRY(dl.alloc(ctx, n));
    MDBG_TRY(dh.alloc(ctx, n));
    MDBG_TRY(dv.alloc(ctx, n));
    MDBG_HIP_CHECK(ctx, hipMemcpyAsync(dl.p, hash_lo, n * 8, hipMemcpyHostToDevice, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipMemcpyAsync(dh.p, hash_hi, n * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(lookup_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, t->lookup->view(), dl.p, dh.p, n, dv.p);
    MDBG_HIP_CHECK(ctx, hipMemcpyAsync(abundance, dv.p, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_small_contigs(mdbg_ctx *ctx, const mdbg_minimizers *unitigs, uint32_t k, uint32_t k_prev, const mdbg_table *prev,
                                  uint8_t *flags) try {
    if (!ctx || !unitigs || k_prev < 1 || k <= k_prev) return set_error(ctx, MDBG_EINVAL, "mdbg_small_contigs: bad argument");
    MDBG_TRY(check_seq(ctx, unitigs, "mdbg_small_contigs"));
    if (unitigs->n_reads && !flags) return set_error(ctx, MDBG_EINVAL, "mdbg_small_contigs: flags is null");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    TableView pv;
    MDBG_TRY(prev_view(ctx, prev, pv));
    if (!unitigs->n_reads) return MDBG_OK;
    DevBuf<uint8_t> d_flags;
    MDBG_TRY(d_flags.alloc(ctx, unitigs->n_reads));
    hipLaunchKernelGGL(small_contig_kernel, dim3(grid_for(unitigs->n_reads, 256)), dim3(256), 0, ctx->stream, unitigs->d_off.p,
                       unitigs->n_reads, unitigs->d_min.p, k, k_prev, pv, d_flags.p);
    MDBG_HIP_CHECK(ctx, hipMemcpyAsync(flags, d_flags.p, unitigs->n_reads, hipMemcpyDeviceToHost, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

// every occupied slot of `tab` becomes a key-only row of a new table of (k-1)-identities
static int edges_from_table(mdbg_ctx *ctx, DeviceTable &tab, uint32_t k_edge, mdbg_table **edges, uint64_t *checksum, const char *who) {
    TableView tv = tab.view();
    const uint64_t nslots = tab.cap + TABLE_EXC_CAP;
    DevBuf<uint32_t> sflag;
    DevBuf<uint64_t> spos;
    MDBG_TRY(sflag.alloc(ctx, nslots));
    MDBG_TRY(spos.alloc(ctx, nslots + 1));
    hipLaunchKernelGGL(slot_flag_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, 0u, 2, sflag.p);
    MDBG_TRY(exclusive_scan_u32(ctx, sflag.p, spos.p, nslots));
    uint64_t n_rows = 0;
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &n_rows, spos.p + nslots, 8, hipMemcpyDeviceToHost));
    mdbg_table *t = new mdbg_table();
    t->k = k_edge;
    t->n_solid = n_rows;
    int rc = alloc_rows(ctx, t, n_rows, false);
    if (rc) { delete t; return rc; }
    RowOut ro{t->d_lo.p, t->d_hi.p, t->d_ab.p, nullptr, t->k};
    hipLaunchKernelGGL(emit_slots_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, sflag.p, spos.p, SeqView{}, SeqView{}, ro, (uint64_t)0);
    if (checksum) {
        DevBuf<unsigned long long> acc;
        rc = acc.alloc(ctx, 1);
        if (rc) { delete t; return rc; }
        (void)hipMemsetAsync(acc.p, 0, 8, ctx->stream);
        if (n_rows) hipLaunchKernelGGL(sum_u64_kernel, dim3(grid_for(n_rows, 256)), dim3(256), 0, ctx->stream, t->d_lo.p, n_rows, acc.p);
        hipError_t e = memcpy_sync(ctx, checksum, acc.p, 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { delete t; return set_error(ctx, MDBG_EHIP, "edge checksum copy failed: %s", hipGetErrorString(e)); }
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { delete t; return set_error(ctx, MDBG_EHIP, "%s failed: %s", who, hipGetErrorString(e)); }
    *edges = t;
    return MDBG_OK;
}

extern "C" int mdbg_edge_index(mdbg_ctx *ctx, const mdbg_table *nodes, mdbg_table **edges, uint64_t *checksum) try {
    if (!ctx || !nodes || !edges) return set_error(ctx, MDBG_EINVAL, "mdbg_edge_index: null argument");
    if (!nodes->has_vectors || nodes->k < 3) return set_error(ctx, MDBG_EINVAL, "mdbg_edge_index: needs a table with vectors (k <= firstK+1)");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint64_t n = nodes->n_records;
    DeviceTable tab;
    MDBG_TRY(build_table_adaptive(ctx, tab, n, 2 * n, [&](TableView v) {
        if (n) {
            LaunchTimer timer(ctx, "edge_index");
            hipLaunchKernelGGL(edge_insert_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, nodes->d_vec.p, n, nodes->k, v);
        }
        return MDBG_OK;
    }));
    return edges_from_table(ctx, tab, nodes->k - 1, edges, checksum, "mdbg_edge_index");
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_unitig_edge_index(mdbg_ctx *ctx, const mdbg_minimizers *unitigs, uint32_t k, mdbg_table **edges, uint64_t *checksum) try {
    if (!ctx || !edges || k < 3) return set_error(ctx, MDBG_EINVAL, "mdbg_unitig_edge_index: bad argument");
    MDBG_TRY(check_seq(ctx, unitigs, "mdbg_unitig_edge_index"));
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint64_t n = unitigs->n_reads;
    DeviceTable tab;
    MDBG_TRY(build_table_adaptive(ctx, tab, 2 * n, 4 * n, [&](TableView v) {
        if (n) {
            LaunchTimer timer(ctx, "edge_index");
            hipLaunchKernelGGL(unitig_edge_insert_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, unitigs->d_off.p, (uint32_t)n,
                               unitigs->d_min.p, k, v);
        }
        return MDBG_OK;
    }));
    return edges_from_table(ctx, tab, k - 1, edges, checksum, "mdbg_unitig_edge_index");
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_table_keys_to_host(mdbg_ctx *ctx, const mdbg_table *t, uint64_t *keys_lo_hi) try {
    if (!ctx || !t || (t->n_records && !keys_lo_hi)) return set_error(ctx, MDBG_EINVAL, "mdbg_table_keys_to_host: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (!t->n_records) return MDBG_OK;
    DevBuf<uint64_t> tmp;
    MDBG_TRY(tmp.alloc(ctx, t->n_records * 2));
    hipLaunchKernelGGL(interleave_keys_kernel, dim3(grid_for(t->n_records, 256)), dim3(256), 0, ctx->stream, t->d_lo.p, t->d_hi.p, t->n_records, tmp.p);
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, keys_lo_hi, tmp.p, t->n_records * 16, hipMemcpyDeviceToHost));
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" void mdbg_table_free(mdbg_table *t) { delete t; }


// =====================================================================================================
// Sharded first pass (reads split over several GPUs, one process each).  See include/mdbg_hip.h.
// =====================================================================================================
namespace mdbg {

constexpr uint32_t SHARD_ROW_WORDS = 3;                 // [hash_lo, hash_hi, count]
constexpr uint64_t SHARD_EMIT_BIT = 1ull << 63;         // in a reply: "you list this key in your table"

__device__ __forceinline__ uint32_t owner_of(uint64_t hi, uint32_t n_ranks) {
    return (uint32_t)(((hi >> 32) * (uint64_t)n_ranks) >> 32);
}

__device__ __forceinline__ bool slot_read(const TableView &t, uint64_t cap, uint64_t s, uint64_t &lo, uint64_t &hi, uint32_t &v) {
    if (s < cap) {
        const uint4 *q = reinterpret_cast<const uint4 *>(t.slots + s);
        const uint4 key = q[0];
        lo = (uint64_t)key.x | ((uint64_t)key.y << 32);
        if (lo == 0ull) return false;
        hi = (uint64_t)key.z | ((uint64_t)key.w << 32);
        v = q[1].x;
        return true;
    }
    uint32_t i = (uint32_t)(s - cap);
    if (i >= *t.exc_n) return false;
    lo = t.exc_lo[i]; hi = t.exc_hi[i]; v = t.exc_val[i];
    return true;
}

// Rows are grouped by owner without a single global atomic (same-address device-scope atomics cost
// ~0.2 us each on this part, serialised): pass 1 writes one LDS histogram per block of SHARD_SPB slots
// to block_hist[owner][block]; an exclusive scan over that owner-major array gives every (owner, block)
// its first row; pass 2 hands out the places inside the block's range through LDS.
constexpr uint32_t SHARD_SPB = 2048;   // slots per block

__global__ __launch_bounds__(256) void owner_hist_kernel(TableView t, uint64_t cap, uint32_t n_ranks, uint32_t *block_hist) {
    __shared__ uint32_t h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t s0 = (uint64_t)blockIdx.x * SHARD_SPB;
    for (uint32_t j = threadIdx.x; j < SHARD_SPB; j += 256) {
        uint64_t lo, hi; uint32_t v;
        if (s0 + j < cap + TABLE_EXC_CAP && slot_read(t, cap, s0 + j, lo, hi, v)) atomicAdd(&h[owner_of(hi, n_ranks)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < n_ranks) block_hist[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

// write every occupied slot as a row [lo, hi, count] into its owner's range and remember its slot
__global__ __launch_bounds__(256) void owner_scatter_kernel(TableView t, uint64_t cap, uint32_t n_ranks, const uint64_t *block_base,
                                                            uint64_t *rows, uint32_t *row_slot) {
    __shared__ uint32_t h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t s0 = (uint64_t)blockIdx.x * SHARD_SPB;
    for (uint32_t j = threadIdx.x; j < SHARD_SPB; j += 256) {
        const uint64_t s = s0 + j;
        uint64_t lo, hi; uint32_t v;
        if (!(s < cap + TABLE_EXC_CAP && slot_read(t, cap, s, lo, hi, v))) continue;
        const uint32_t own = owner_of(hi, n_ranks);
        const uint64_t row = block_base[(uint64_t)own * gridDim.x + blockIdx.x] + atomicAdd(&h[own], 1u);
        uint64_t *o = rows + row * SHARD_ROW_WORDS;
        o[0] = lo; o[1] = hi; o[2] = v;
        row_slot[row] = s < cap ? (uint32_t)s : (0x80000000u | (uint32_t)(s - cap));
    }
}

// owner: sum the received rows by key; rep = index of the received row that created the key -- its sender is the
// one rank that will list the key
__global__ __launch_bounds__(256) void rows_add_kernel(const uint64_t *rows, uint64_t n, TableView t) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t *r = rows + i * SHARD_ROW_WORDS;
    table_upsert_count(t, r[0], r[1], (uint32_t)r[2], (uint32_t)i);
}

__global__ __launch_bounds__(256) void rows_reply_kernel(const uint64_t *rows, uint64_t n, TableView t, uint64_t *reply) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t *r = rows + i * SHARD_ROW_WORDS;
    const uint32_t slot = table_lookup_slot(t, r[0], r[1]);
    uint64_t out = 0;
    if (slot != SLOT_NONE) out = (uint64_t)table_slot_val(t, slot) | (table_slot_rep(t, slot) == (uint32_t)i ? SHARD_EMIT_BIT : 0ull);
    reply[i] = out;
}

// global counts back into the local table (the slot of every sent row was remembered); the rows this rank
// was told to list and that are solid get their flag
__global__ __launch_bounds__(256) void apply_global_counts_kernel(const uint64_t *reply, const uint32_t *row_slot, uint64_t n, TableView t,
                                                                  uint64_t cap, uint32_t min_abundance, uint32_t *flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = row_slot[i], v = (uint32_t)reply[i];
    const bool emit = (reply[i] & SHARD_EMIT_BIT) != 0ull && v > 1u && !(v < min_abundance);
    if (s & 0x80000000u) { t.exc_val[s & 0x7FFFFFFFu] = v; flag[cap + (s & 0x7FFFFFFFu)] = emit ? 1u : 0u; }
    else { t.slots[s].val = v; flag[s] = emit ? 1u : 0u; }
}

// ---- the same grouping by owner for the rows of a finished local table (sharded k > firstK) ----
constexpr uint32_t SHARD_RPB = 2048;   // rows per block

__global__ __launch_bounds__(256) void table_owner_hist_kernel(const uint64_t *hi, uint64_t n, uint32_t n_ranks, uint32_t *block_hist) {
    __shared__ uint32_t h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t r0 = (uint64_t)blockIdx.x * SHARD_RPB;
    for (uint32_t j = threadIdx.x; j < SHARD_RPB; j += 256)
        if (r0 + j < n) atomicAdd(&h[owner_of(hi[r0 + j], n_ranks)], 1u);
    __syncthreads();
    if (threadIdx.x < n_ranks) block_hist[(uint64_t)threadIdx.x * gridDim.x + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(256) void table_owner_scatter_kernel(const uint64_t *lo, const uint64_t *hi, const uint32_t *ab, uint64_t n,
                                                                  uint32_t n_ranks, const uint64_t *block_base, uint64_t *rows, uint32_t *row_index) {
    __shared__ uint32_t h[64];
    if (threadIdx.x < 64) h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t r0 = (uint64_t)blockIdx.x * SHARD_RPB;
    for (uint32_t j = threadIdx.x; j < SHARD_RPB; j += 256) {
        const uint64_t i = r0 + j;
        if (i >= n) continue;
        const uint32_t own = owner_of(hi[i], n_ranks);
        const uint64_t row = block_base[(uint64_t)own * gridDim.x + blockIdx.x] + atomicAdd(&h[own], 1u);
        uint64_t *o = rows + row * SHARD_ROW_WORDS;
        o[0] = lo[i]; o[1] = hi[i]; o[2] = ab[i];
        row_index[row] = (uint32_t)i;
    }
}

// the sharded first pass on the partitioned machinery: reply i belongs to local key row_index[i]
__global__ void replies_to_keys_kernel(const uint64_t *reply, const uint32_t *row_index, uint64_t n, uint32_t min_abundance, uint32_t *gcount, uint32_t *listed) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t v = (uint32_t)reply[i], key = row_index[i];
    gcount[key] = v;
    listed[key] = ((reply[i] & SHARD_EMIT_BIT) != 0ull && v > 1u && !(v < min_abundance)) ? 1u : 0u;
}

__global__ void keep_flag_kernel(const uint64_t *reply, const uint32_t *row_index, uint64_t n, uint32_t *flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[row_index[i]] = (reply[i] & SHARD_EMIT_BIT) ? 1u : 0u;
}

__global__ void keep_rows_kernel(const uint32_t *flag, const uint64_t *pos, uint64_t n, uint32_t k, const uint64_t *lo, const uint64_t *hi,
                                 const uint32_t *ab, const uint32_t *vec, uint64_t *olo, uint64_t *ohi, uint32_t *oab, uint32_t *ovec) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const uint64_t d = pos[i];
    olo[d] = lo[i]; ohi[d] = hi[i]; oab[d] = ab[i];
    if (vec) for (uint32_t j = 0; j < k; j++) ovec[d * k + j] = vec[i * k + j];
}

}  // namespace mdbg

struct mdbg_shard {
    uint32_t k = 0, n_ranks = 1;
    const mdbg_minimizers *reads = nullptr;
    const mdbg_table *table = nullptr;       // mdbg_shard_from_table: the local table whose rows are being deduplicated
    mdbg::DevBuf<uint32_t> row_index;        // ... and the table row every sent row came from
    mdbg::InstIndex ix;
    mdbg::DeviceTable local, owner;
    mdbg::DevBuf<uint32_t> inst_slot, row_slot;
    mdbg::DevBuf<uint64_t> rows, reply;
    mdbg::DevBuf<uint64_t> local_replies;    // mdbg_shard_exchange_local: the replies to this shard's rows
    uint64_t n_rows = 0;
    bool reduced = false;
    // the rank's share counted by the partitioned pass (csrc/partition.hip) instead of a local table: its distinct keys are the rows
    mdbg::PartLocal *part = nullptr;
    ~mdbg_shard() { if (part) mdbg::part_local_free(part); }
};

extern "C" uint32_t mdbg_row_words(uint32_t) { return mdbg::SHARD_ROW_WORDS; }

extern "C" int mdbg_shard_begin(mdbg_ctx *ctx, const mdbg_minimizers *reads, uint32_t k, uint32_t n_ranks,
                                mdbg_shard **out, const uint64_t **d_rows, uint64_t *counts) try {
    if (!ctx || !reads || !out || !d_rows || !counts || k < 2 || n_ranks < 1 || n_ranks > 64)
        return set_error(ctx, MDBG_EINVAL, "mdbg_shard_begin: bad argument");
    MDBG_TRY(check_seq(ctx, reads, "mdbg_shard_begin"));
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<mdbg_shard> sh(new mdbg_shard());
    sh->k = k; sh->n_ranks = n_ranks; sh->reads = reads;
    // the rank's share through the partitioned pass (as mdbg_kminmer_count_first chooses): its distinct keys with their local counts come out
    // bucket after bucket and are grouped by owner like the rows of a finished table (mdbg_shard_from_table)
    if (ctx->first_pass_mode == 2 || (ctx->first_pass_mode == 0 && reads->n_min >= ctx->part_auto_min)) {
        bool done = false;
        MDBG_TRY(part_local_keys(ctx, reads, k, &sh->part, &done));
        if (done) {
            const uint64_t *klo, *khi; const uint32_t *kcnt; uint64_t n = 0, n_inst = 0;
            part_local_arrays(sh->part, &klo, &khi, &kcnt, &n, &n_inst);
            if (n >= (1ull << 32)) return set_error(ctx, MDBG_ERANGE, "more than 2^32 distinct local keys");
            const unsigned nb = grid_for(n, SHARD_RPB);
            const uint64_t nh = (uint64_t)nb * n_ranks;
            DevBuf<uint32_t> block_hist;
            DevBuf<uint64_t> block_base;
            MDBG_TRY(block_hist.alloc(ctx, nh));
            MDBG_TRY(block_base.alloc(ctx, nh + 1));
            {
                LaunchTimer timer(ctx, "shard_rows");
                hipLaunchKernelGGL(table_owner_hist_kernel, dim3(nb), dim3(256), 0, ctx->stream, khi, n, n_ranks, block_hist.p);
            }
            MDBG_TRY(exclusive_scan_u32(ctx, block_hist.p, block_base.p, nh));
            std::vector<uint64_t> base(nh + 1);
            MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, base.data(), block_base.p, (nh + 1) * 8, hipMemcpyDeviceToHost));
            for (uint32_t r = 0; r < n_ranks; r++) counts[r] = base[(uint64_t)(r + 1) * nb] - base[(uint64_t)r * nb];
            update_key_hint(ctx, 3, n, n_inst);
            MDBG_TRY(sh->rows.alloc(ctx, n * SHARD_ROW_WORDS));
            MDBG_TRY(sh->row_index.alloc(ctx, n));
            sh->n_rows = n;
            {
                LaunchTimer timer(ctx, "shard_rows");
                hipLaunchKernelGGL(table_owner_scatter_kernel, dim3(nb), dim3(256), 0, ctx->stream, klo, khi, kcnt, n, n_ranks, block_base.p, sh->rows.p, sh->row_index.p);
            }
            MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
            MDBG_DBG(ctx, "shard_begin (partitioned): %llu rows", (unsigned long long)n);
            *d_rows = sh->rows.p;
            *out = sh.release();
            return MDBG_OK;
        }
    }
    MDBG_TRY(build_inst_index(ctx, reads, k, sh->ix));
    const uint64_t I = sh->ix.total;
    MDBG_DBG(ctx, "shard_begin: %llu instances", (unsigned long long)I);
    SeqView sv = make_view(reads, sh->ix);
    MDBG_TRY(sh->inst_slot.alloc(ctx, I));
    MDBG_TRY(build_table_adaptive(ctx, sh->local, (uint64_t)((double)I * ctx->key_ratio_hint[3]), I, [&](TableView v) {
        if (I) {
            LaunchTimer timer(ctx, "kminmer_insert");
            hipLaunchKernelGGL(count_insert_kernel, dim3(instance_grid(ctx, sv.n_reads)), dim3(256), 0, ctx->stream, sv, k, v, sh->inst_slot.p, (uint64_t)0);
        }
        return MDBG_OK;
    }, ctx->key_ratio_known[3]));
    TableView tv = sh->local.view();
    const uint64_t nslots = sh->local.cap + TABLE_EXC_CAP;
    const unsigned nb = grid_for(nslots, SHARD_SPB);
    const uint64_t nh = (uint64_t)nb * n_ranks;
    DevBuf<uint32_t> block_hist;
    DevBuf<uint64_t> block_base;
    MDBG_TRY(block_hist.alloc(ctx, nh));
    MDBG_TRY(block_base.alloc(ctx, nh + 1));
    {
        LaunchTimer timer(ctx, "shard_rows");
        hipLaunchKernelGGL(owner_hist_kernel, dim3(nb), dim3(256), 0, ctx->stream, tv, sh->local.cap, n_ranks, block_hist.p);
    }
    MDBG_TRY(exclusive_scan_u32(ctx, block_hist.p, block_base.p, nh));
    std::vector<uint64_t> base(nh + 1);
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, base.data(), block_base.p, (nh + 1) * 8, hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < n_ranks; r++) counts[r] = base[(uint64_t)(r + 1) * nb] - base[(uint64_t)r * nb];
    const uint64_t total = base[nh];
    if (total >= (1ull << 32)) return set_error(ctx, MDBG_ERANGE, "more than 2^32 distinct local keys");
    update_key_hint(ctx, 3, total, I);        // one row per distinct local key
    MDBG_TRY(sh->rows.alloc(ctx, total * SHARD_ROW_WORDS));
    MDBG_TRY(sh->row_slot.alloc(ctx, total));
    sh->n_rows = total;
    {
        LaunchTimer timer(ctx, "shard_rows");
        hipLaunchKernelGGL(owner_scatter_kernel, dim3(nb), dim3(256), 0, ctx->stream, tv, sh->local.cap, n_ranks, block_base.p,
                           sh->rows.p, sh->row_slot.p);
    }
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    MDBG_DBG(ctx, "shard_begin: %llu rows", (unsigned long long)total);
    *d_rows = sh->rows.p;
    *out = sh.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

__global__ void rows_zero_word_kernel(const uint64_t *rows, uint64_t n, unsigned long long *out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && (rows[i * SHARD_ROW_WORDS] == 0ull || rows[i * SHARD_ROW_WORDS + 1] == 0ull)) atomicAdd(out, 1ull);
}

extern "C" int mdbg_shard_reduce(mdbg_ctx *ctx, mdbg_shard *sh, const uint64_t *d_recv, uint64_t n_recv, const uint64_t **d_reply) try {
    if (!ctx || !sh || !d_reply || (n_recv && !d_recv)) return set_error(ctx, MDBG_EINVAL, "mdbg_shard_reduce: bad argument");
    if (n_recv >= (1ull << 32)) return set_error(ctx, MDBG_ERANGE, "more than 2^32 received rows");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // every key this rank owns arrives once from each rank that saw it: about n_recv / n_ranks distinct keys.  Sized for
    // that (+50 %), the table is n_ranks times smaller than one sized for the rows; it grows and refills if that was short.
    const uint64_t expected = n_recv / sh->n_ranks + n_recv / (2 * sh->n_ranks) + 1024;
    MDBG_DBG(ctx, "shard_reduce: %llu rows", (unsigned long long)n_recv);
    if (debug_on() && n_recv) {
        DevBuf<unsigned long long> z;
        MDBG_TRY(z.alloc(ctx, 1));
        (void)hipMemsetAsync(z.p, 0, 8, ctx->stream);
        hipLaunchKernelGGL(rows_zero_word_kernel, dim3(grid_for(n_recv, 256)), dim3(256), 0, ctx->stream, d_recv, n_recv, z.p);
        unsigned long long hz = 0;
        MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &hz, z.p, 8, hipMemcpyDeviceToHost));
        MDBG_DBG(ctx, "shard_reduce: %llu rows with a zero key word", hz);
    }
    if (ctx->first_pass_mode != 1) {          // the rows bucketed by key, every bucket summed in LDS (csrc/partition.hip)
        bool done = false;
        MDBG_TRY(sh->reply.alloc(ctx, n_recv));
        MDBG_TRY(part_owner_reduce(ctx, d_recv, n_recv, sh->reply.p, &done));
        if (done) {
            sh->reduced = true;
            *d_reply = sh->reply.p;
            return MDBG_OK;
        }
    }
    MDBG_TRY(build_table_adaptive(ctx, sh->owner, expected < n_recv ? expected : n_recv, n_recv, [&](TableView v) {
        if (n_recv) {
            LaunchTimer timer(ctx, "shard_reduce");
            hipLaunchKernelGGL(rows_add_kernel, dim3(grid_for(n_recv, 256)), dim3(256), 0, ctx->stream, d_recv, n_recv, v);
        }
        return MDBG_OK;
    }));
    TableView tv = sh->owner.view();
    MDBG_TRY(sh->reply.alloc(ctx, n_recv));
    if (n_recv) {
        LaunchTimer timer(ctx, "shard_reduce");
        hipLaunchKernelGGL(rows_reply_kernel, dim3(grid_for(n_recv, 256)), dim3(256), 0, ctx->stream, d_recv, n_recv, tv, sh->reply.p);
    }
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    MDBG_TRY(sh->owner.check_overflow(ctx));
    sh->owner = DeviceTable();      // only the replies are needed from here on
    sh->reduced = true;
    *d_reply = sh->reply.p;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_shard_finish(mdbg_ctx *ctx, mdbg_shard *sh, const uint64_t *d_replies, uint32_t min_abundance, mdbg_table **out) try {
    if (!ctx || !sh || !out || (sh->n_rows && !d_replies)) return set_error(ctx, MDBG_EINVAL, "mdbg_shard_finish: bad argument");
    if (!sh->reduced) return set_error(ctx, MDBG_EINVAL, "mdbg_shard_finish: mdbg_shard_reduce has not run");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (sh->part) {
        // the replies, row by row, back to the order of the local keys: global count, and whether this rank lists the key
        DevBuf<uint32_t> gcount, listed;
        MDBG_TRY(gcount.alloc(ctx, sh->n_rows));
        MDBG_TRY(listed.alloc(ctx, sh->n_rows));
        if (sh->n_rows)
            hipLaunchKernelGGL(replies_to_keys_kernel, dim3(grid_for(sh->n_rows, 256)), dim3(256), 0, ctx->stream, d_replies, sh->row_index.p, sh->n_rows, min_abundance,
                               gcount.p, listed.p);
        return part_local_finish(ctx, sh->part, gcount.p, listed.p, min_abundance, out);
    }
    const uint32_t k = sh->k;
    const uint64_t I = sh->ix.total;
    TableView lv = sh->local.view();
    const uint64_t nslots = sh->local.cap + TABLE_EXC_CAP;
    DevBuf<uint32_t> sflag;
    DevBuf<uint64_t> spos;
    MDBG_TRY(sflag.alloc(ctx, nslots));
    MDBG_TRY(spos.alloc(ctx, nslots + 1));
    MDBG_HIP_CHECK(ctx, hipMemsetAsync(sflag.p, 0, nslots * 4, ctx->stream));
    if (sh->n_rows)
        hipLaunchKernelGGL(apply_global_counts_kernel, dim3(grid_for(sh->n_rows, 256)), dim3(256), 0, ctx->stream, d_replies, sh->row_slot.p,
                           sh->n_rows, lv, sh->local.cap, min_abundance, sflag.p);
    MDBG_TRY(exclusive_scan_u32(ctx, sflag.p, spos.p, nslots));
    uint64_t n_solid = 0, n_resc = 0;
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &n_solid, spos.p + nslots, 8, hipMemcpyDeviceToHost));
    // rescue over the local reads against the global counts now sitting in the local table
    SeqView sv = make_view(sh->reads, sh->ix), none{};
    RescuePlan plan;
    if (min_abundance <= 1 && I) {
        MDBG_TRY(plan_rescue(ctx, sh->local, sh->ix, sv.n_reads, sh->inst_slot.p, plan));
        n_resc = plan.total;
    }
    mdbg_table *t = new mdbg_table();
    t->k = k;
    t->n_solid = n_solid;
    t->st_minimizers = sv.n_min; t->st_instances = I; t->st_keys = sh->n_rows; t->st_slots = nslots;
    int rc = alloc_rows(ctx, t, n_solid + n_resc, true);
    if (rc) { delete t; return rc; }
    RowOut ro{t->d_lo.p, t->d_hi.p, t->d_ab.p, t->d_vec.p, k};
    {
        LaunchTimer timer(ctx, "kminmer_emit");
        hipLaunchKernelGGL(emit_slots_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, lv, sh->local.cap, sflag.p, spos.p, sv, none, ro, (uint64_t)0);
        if (n_resc) launch_emit_rescued(ctx, sv, k, sh->inst_slot.p, plan, ro, n_solid);
    }
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { delete t; return set_error(ctx, MDBG_EHIP, "mdbg_shard_finish failed: %s", hipGetErrorString(e)); }
    *out = t;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

// ---- sharded k > firstK -------------------------------------------------------------------------------------------
extern "C" int mdbg_shard_from_table(mdbg_ctx *ctx, const mdbg_table *local, uint32_t n_ranks, mdbg_shard **out, const uint64_t **d_rows,
                                     uint64_t *counts) try {
    if (!ctx || !local || !out || !d_rows || !counts || n_ranks < 1 || n_ranks > 64)
        return set_error(ctx, MDBG_EINVAL, "mdbg_shard_from_table: bad argument");
    if (local->n_records >= (1ull << 32)) return set_error(ctx, MDBG_ERANGE, "mdbg_shard_from_table: more than 2^32 rows");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<mdbg_shard> sh(new mdbg_shard());
    sh->k = local->k; sh->n_ranks = n_ranks; sh->table = local;
    const uint64_t n = local->n_records;
    const unsigned nb = grid_for(n, SHARD_RPB);
    const uint64_t nh = (uint64_t)nb * n_ranks;
    DevBuf<uint32_t> block_hist;
    DevBuf<uint64_t> block_base;
    MDBG_TRY(block_hist.alloc(ctx, nh));
    MDBG_TRY(block_base.alloc(ctx, nh + 1));
    {
        LaunchTimer timer(ctx, "shard_rows");
        hipLaunchKernelGGL(table_owner_hist_kernel, dim3(nb), dim3(256), 0, ctx->stream, local->d_hi.p, n, n_ranks, block_hist.p);
    }
    MDBG_TRY(exclusive_scan_u32(ctx, block_hist.p, block_base.p, nh));
    std::vector<uint64_t> base(nh + 1);
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, base.data(), block_base.p, (nh + 1) * 8, hipMemcpyDeviceToHost));
    for (uint32_t r = 0; r < n_ranks; r++) counts[r] = base[(uint64_t)(r + 1) * nb] - base[(uint64_t)r * nb];
    MDBG_TRY(sh->rows.alloc(ctx, n * SHARD_ROW_WORDS));
    MDBG_TRY(sh->row_index.alloc(ctx, n));
    sh->n_rows = n;
    {
        LaunchTimer timer(ctx, "shard_rows");
        hipLaunchKernelGGL(table_owner_scatter_kernel, dim3(nb), dim3(256), 0, ctx->stream, local->d_lo.p, local->d_hi.p, local->d_ab.p, n, n_ranks,
                           block_base.p, sh->rows.p, sh->row_index.p);
    }
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *d_rows = sh->rows.p;
    *out = sh.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_shard_keep(mdbg_ctx *ctx, mdbg_shard *sh, const uint64_t *d_replies, mdbg_table **out) try {
    if (!ctx || !sh || !out || !sh->table || (sh->n_rows && !d_replies)) return set_error(ctx, MDBG_EINVAL, "mdbg_shard_keep: bad argument");
    if (!sh->reduced) return set_error(ctx, MDBG_EINVAL, "mdbg_shard_keep: mdbg_shard_reduce has not run");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const mdbg_table *src = sh->table;
    const uint64_t n = sh->n_rows;
    DevBuf<uint32_t> flag;
    DevBuf<uint64_t> pos;
    MDBG_TRY(flag.alloc(ctx, n));
    MDBG_TRY(pos.alloc(ctx, n + 1));
    if (n) hipLaunchKernelGGL(keep_flag_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, d_replies, sh->row_index.p, n, flag.p);
    MDBG_TRY(exclusive_scan_u32(ctx, flag.p, pos.p, n));
    uint64_t kept = 0;
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &kept, pos.p + n, 8, hipMemcpyDeviceToHost));
    std::unique_ptr<mdbg_table> t(new mdbg_table());
    t->k = src->k;
    MDBG_TRY(alloc_rows(ctx, t.get(), kept, src->has_vectors));
    t->n_solid = kept;
    if (n) hipLaunchKernelGGL(keep_rows_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, flag.p, pos.p, n, src->k, src->d_lo.p, src->d_hi.p,
                              src->d_ab.p, src->has_vectors ? src->d_vec.p : nullptr, t->d_lo.p, t->d_hi.p, t->d_ab.p,
                              src->has_vectors ? t->d_vec.p : nullptr);
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *out = t.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

// Both all-to-alls of a sharded pass among shards that live on ONE device (a job checking itself, tests): device-to-device copies
// stand in for the wire, everything else -- mdbg_shard_reduce on every owner, the replies back in the order the rows were sent --
// is what mdbg_shard_exchange does between GPUs.
extern "C" int mdbg_shard_exchange_local(mdbg_ctx *ctx, mdbg_shard *const *shards, uint32_t n, const uint64_t *const *d_rows, const uint64_t *counts,
                                         const uint64_t **d_replies) try {
    if (!ctx || !shards || !d_rows || !counts || !d_replies || n < 1 || n > 64) return set_error(ctx, MDBG_EINVAL, "mdbg_shard_exchange_local: bad argument");
    for (uint32_t r = 0; r < n; r++)
        if (!shards[r] || shards[r]->n_ranks != n) return set_error(ctx, MDBG_EINVAL, "mdbg_shard_exchange_local: shard %u was not begun for %u ranks", r, n);
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t rw = SHARD_ROW_WORDS;
    // soff[src][dst]: first row of src's rows for dst; roff[dst][src]: first received row on dst that came from src
    std::vector<uint64_t> soff((size_t)n * (n + 1), 0), roff((size_t)n * (n + 1), 0);
    for (uint32_t a = 0; a < n; a++)
        for (uint32_t b = 0; b < n; b++) {
            soff[(size_t)a * (n + 1) + b + 1] = soff[(size_t)a * (n + 1) + b] + counts[(size_t)a * n + b];
            roff[(size_t)a * (n + 1) + b + 1] = roff[(size_t)a * (n + 1) + b] + counts[(size_t)b * n + a];
        }
    std::vector<const uint64_t *> reply(n, nullptr);
    for (uint32_t dst = 0; dst < n; dst++) {
        const uint64_t n_recv = roff[(size_t)dst * (n + 1) + n];
        DevBuf<uint64_t> recv;
        MDBG_TRY(recv.alloc(ctx, n_recv * rw));
        for (uint32_t src = 0; src < n; src++) {
            const uint64_t c = counts[(size_t)src * n + dst];
            if (!c) continue;
            if (!d_rows[src]) return set_error(ctx, MDBG_EINVAL, "mdbg_shard_exchange_local: shard %u has rows but no row pointer", src);
            MDBG_HIP_CHECK(ctx, hipMemcpyAsync(recv.p + roff[(size_t)dst * (n + 1) + src] * rw, d_rows[src] + soff[(size_t)src * (n + 1) + dst] * rw, c * rw * 8,
                                               hipMemcpyDeviceToDevice, ctx->stream));
        }
        MDBG_TRY(mdbg_shard_reduce(ctx, shards[dst], recv.p, n_recv, &reply[dst]));   // (synchronises: recv may go)
    }
    for (uint32_t src = 0; src < n; src++) {
        MDBG_TRY(shards[src]->local_replies.alloc(ctx, soff[(size_t)src * (n + 1) + n]));
        for (uint32_t dst = 0; dst < n; dst++) {
            const uint64_t c = counts[(size_t)src * n + dst];
            if (c) MDBG_HIP_CHECK(ctx, hipMemcpyAsync(shards[src]->local_replies.p + soff[(size_t)src * (n + 1) + dst], reply[dst] + roff[(size_t)dst * (n + 1) + src],
                                                      c * 8, hipMemcpyDeviceToDevice, ctx->stream));
        }
        d_replies[src] = shards[src]->local_replies.p;
    }
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" void mdbg_shard_free(mdbg_shard *shard) { delete shard; }
