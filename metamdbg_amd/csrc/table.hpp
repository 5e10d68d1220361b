// table.hpp -- device open-addressing hash table keyed by the 128-bit k-min-mer identity
// (KmerVec::hash128, Commons.hpp:941-969).  It plays the role of the reference's
// phmap::parallel_flat_hash_map<u128,u32> (_kminmerAbundances / _mdbgNodesLight,
// graph/CreateMdbg.hpp:34-35) and of KminmerCounter's partition + sort + run-length
// (graph/CreateMdbg.hpp:3714-3851): equal keys meet in one slot and are counted with atomics.
//
// Insertion is wait-free and uses device-scope atomics only (per-XCD L2s are not coherent for
// plain accesses inside one launch): a slot is claimed by CAS on its low word, then the high
// word is published by a second CAS; whoever publishes first defines the slot's key, a loser
// with a different high word (a 64-bit collision) simply keeps probing.  0 is the empty marker
// of both words, so keys with a zero word (probability 2^-63) go to a 64-entry side list that is
// searched linearly.  Lookups after the building kernel has finished use plain loads.
#pragma once
#include "common.hpp"

namespace mdbg {

constexpr uint32_t TABLE_EXC_CAP = 64;
constexpr uint32_t SLOT_NONE = 0xFFFFFFFFu;

struct TableView {
    unsigned long long *lo;   // cap
    unsigned long long *hi;   // cap
    uint32_t *val;            // cap: count / abundance
    uint32_t *rep;            // cap: a representative instance id (may be nullptr)
    uint64_t mask;            // cap - 1
    // side list for keys with a zero word
    unsigned long long *exc_lo, *exc_hi;
    uint32_t *exc_val, *exc_rep;
    uint32_t *exc_n;          // entries used
    uint32_t *exc_lock;
    uint32_t *overflow;       // set when the table or the side list is full
};

#ifdef __HIPCC__

__device__ __forceinline__ uint64_t table_home(uint64_t lo, uint64_t hi, uint64_t mask) {
    // the key is already a Murmur3 output; fold both words so owner-rank partitioning by the top
    // bits of `hi` (multi-GPU) does not correlate with the slot
    return (lo ^ (hi >> 17)) & mask;
}

// Side-list insert (serialised by a spin lock taken one lane at a time).  Returns the entry index.
__device__ inline uint32_t table_exc_upsert(const TableView &t, uint64_t lo, uint64_t hi, uint32_t add, uint32_t set_val,
                                            bool do_set, uint32_t rep, bool insert_if_absent) {
    uint32_t result = SLOT_NONE;
    bool done = false;
    while (!done) {
        if (atomicCAS(t.exc_lock, 0u, 1u) == 0u) {
            __threadfence();
            uint32_t n = __hip_atomic_load(t.exc_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            uint32_t i = 0;
            for (; i < n; i++)
                if (__hip_atomic_load(&t.exc_lo[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == lo &&
                    __hip_atomic_load(&t.exc_hi[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == hi) break;
            if (i == n) {
                if (!insert_if_absent) {
                    i = SLOT_NONE;
                } else if (n >= TABLE_EXC_CAP) {
                    atomicExch(t.overflow, 1u);
                    i = SLOT_NONE;
                } else {
                    __hip_atomic_store(&t.exc_lo[n], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&t.exc_hi[n], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&t.exc_val[n], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (t.exc_rep) __hip_atomic_store(&t.exc_rep[n], rep, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(t.exc_n, n + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (i != SLOT_NONE) {
                if (do_set) __hip_atomic_store(&t.exc_val[i], set_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (add) atomicAdd(&t.exc_val[i], add);
            }
            result = i;
            __threadfence();
            atomicExch(t.exc_lock, 0u);
            done = true;
        }
    }
    return result == SLOT_NONE ? SLOT_NONE : (0x80000000u | result);
}

// Find or create the slot of (lo,hi).  Returns the slot index (bit 31 set = side list entry),
// or SLOT_NONE when create == false and the key is absent / the table is full.
__device__ __forceinline__ uint32_t table_find_or_insert(const TableView &t, uint64_t lo, uint64_t hi, bool create) {
    uint64_t s = table_home(lo, hi, t.mask);
    for (uint64_t probes = 0; probes <= t.mask; probes++, s = (s + 1) & t.mask) {
        unsigned long long cur = __hip_atomic_load(&t.lo[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0ull) {
            if (!create) return SLOT_NONE;
            cur = atomicCAS(&t.lo[s], 0ull, (unsigned long long)lo);
            if (cur == 0ull) cur = lo;
        }
        if (cur != lo) continue;
        unsigned long long h = __hip_atomic_load(&t.hi[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (h == 0ull) {
            if (!create) {
                // slot claimed by a concurrent inserter of some key with the same low word: during a
                // build this cannot be told apart from "absent"; pure lookups never race with builds
                return SLOT_NONE;
            }
            h = atomicCAS(&t.hi[s], 0ull, (unsigned long long)hi);
            if (h == 0ull) h = hi;
        }
        if (h == hi) return (uint32_t)s;
    }
    if (create) atomicExch(t.overflow, 1u);
    return SLOT_NONE;
}

// Read-only lookup after the build kernel completed (plain loads are safe across a kernel boundary).
__device__ __forceinline__ bool table_lookup(const TableView &t, uint64_t lo, uint64_t hi, uint32_t &val) {
    if (lo == 0ull || hi == 0ull) {
        uint32_t n = *t.exc_n;
        for (uint32_t i = 0; i < n; i++)
            if (t.exc_lo[i] == lo && t.exc_hi[i] == hi) { val = t.exc_val[i]; return true; }
        return false;
    }
    uint64_t s = table_home(lo, hi, t.mask);
    for (uint64_t probes = 0; probes <= t.mask; probes++, s = (s + 1) & t.mask) {
        unsigned long long cur = t.lo[s];
        if (cur == 0ull) return false;
        if (cur == lo && t.hi[s] == hi) { val = t.val[s]; return true; }
    }
    return false;
}

__device__ __forceinline__ uint32_t table_slot_val(const TableView &t, uint32_t slot) {
    return (slot & 0x80000000u) ? t.exc_val[slot & 0x7FFFFFFFu] : t.val[slot];
}

#endif  // __HIPCC__

// Host-side owner of the table storage.
struct DeviceTable {
    uint64_t cap = 0;
    DevBuf<unsigned long long> lo, hi, exc_lo, exc_hi;
    DevBuf<uint32_t> val, rep, exc_val, exc_rep, ctl;  // ctl: [0]=exc_n [1]=exc_lock [2]=overflow

    int init(mdbg_ctx *ctx, uint64_t min_slots, bool with_rep) {
        cap = 1024;
        while (cap < min_slots) cap <<= 1;
        if (cap > (1ull << 31)) return set_error(ctx, MDBG_ERANGE, "hash table of %llu slots exceeds 2^31", (unsigned long long)cap);
        MDBG_TRY(lo.alloc(ctx, cap));
        MDBG_TRY(hi.alloc(ctx, cap));
        MDBG_TRY(val.alloc(ctx, cap));
        if (with_rep) MDBG_TRY(rep.alloc(ctx, cap));
        MDBG_TRY(exc_lo.alloc(ctx, TABLE_EXC_CAP));
        MDBG_TRY(exc_hi.alloc(ctx, TABLE_EXC_CAP));
        MDBG_TRY(exc_val.alloc(ctx, TABLE_EXC_CAP));
        MDBG_TRY(exc_rep.alloc(ctx, TABLE_EXC_CAP));
        MDBG_TRY(ctl.alloc(ctx, 4));
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(lo.p, 0, cap * 8, ctx->stream));
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(hi.p, 0, cap * 8, ctx->stream));
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(val.p, 0, cap * 4, ctx->stream));
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(exc_val.p, 0, TABLE_EXC_CAP * 4, ctx->stream));
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(ctl.p, 0, 16, ctx->stream));
        return MDBG_OK;
    }
    TableView view() const {
        TableView v;
        v.lo = lo.p; v.hi = hi.p; v.val = val.p; v.rep = rep.p; v.mask = cap - 1;
        v.exc_lo = exc_lo.p; v.exc_hi = exc_hi.p; v.exc_val = exc_val.p; v.exc_rep = exc_rep.p;
        v.exc_n = ctl.p; v.exc_lock = ctl.p + 1; v.overflow = ctl.p + 2;
        return v;
    }
    int check_overflow(mdbg_ctx *ctx) {
        uint32_t c[4];
        MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, c, ctl.p, 16, hipMemcpyDeviceToHost));
        if (c[2]) return set_error(ctx, MDBG_ERANGE, "k-min-mer hash table overflow (cap %llu)", (unsigned long long)cap);
        return MDBG_OK;
    }
};

}  // namespace mdbg
