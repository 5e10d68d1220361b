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
constexpr uint32_t TABLE_OCC_WAYS = 256;    // same-address atomics serialise at ~0.2 us each: spread the block sums

// One slot = 32 bytes, half a 64-byte sector: key, value and representative arrive with ONE memory
// transaction per probe (the SoA layout of round 1 touched four sectors per insert).
struct alignas(32) TableSlot {
    unsigned long long lo;    // 0 = empty
    unsigned long long hi;    // 0 = not yet published
    uint32_t val;             // count / abundance
    uint32_t rep;             // a representative instance id
    uint32_t pad[2];
};

constexpr uint32_t TABLE_MAX_PROBES = 96;   // longer probe sequences mean the table is too full: grow and rebuild

struct TableView {
    TableSlot *slots;         // cap
    uint64_t mask;            // cap - 1
    // side list for keys with a zero word
    unsigned long long *exc_lo, *exc_hi;
    uint32_t *exc_val, *exc_rep;
    uint32_t *exc_n;          // entries used
    uint32_t *exc_lock;
    uint32_t *overflow;       // set when the table or the side list is full
    uint32_t poll_overflow;   // passes over the sequences look at `overflow` now and then and stop once it is set (MDBG_NO_GIVE_UP=1: never)
    uint32_t *occ;            // TABLE_OCC_WAYS partial counts of occupied slots, filled by a pass that asks for them
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
    // a full list is final: look the key up without the lock (rows of zeros -- a transfer that did not arrive -- otherwise
    // queue millions of threads on one spin lock for minutes before the overflow is reported)
    if (__hip_atomic_load(t.exc_n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= TABLE_EXC_CAP) {
        __threadfence();
        uint32_t i = 0;
        for (; i < TABLE_EXC_CAP; i++)
            if (__hip_atomic_load(&t.exc_lo[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == lo &&
                __hip_atomic_load(&t.exc_hi[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == hi) break;
        if (i == TABLE_EXC_CAP) {
            if (insert_if_absent) atomicExch(t.overflow, 1u);
            return SLOT_NONE;
        }
        if (do_set) __hip_atomic_store(&t.exc_val[i], set_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (add) atomicAdd(&t.exc_val[i], add);
        return 0x80000000u | i;
    }
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
                    __threadfence();
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
// *created (optional) is set when THIS call published the key: exactly one caller per key sees it.
__device__ __forceinline__ uint32_t table_find_or_insert(const TableView &t, uint64_t lo, uint64_t hi, bool create, bool *created = nullptr) {
    uint64_t s = table_home(lo, hi, t.mask);
    const uint64_t limit = t.mask < TABLE_MAX_PROBES ? t.mask : (uint64_t)TABLE_MAX_PROBES;
    for (uint64_t probes = 0; probes <= limit; probes++, s = (s + 1) & t.mask) {
        unsigned long long cur = __hip_atomic_load(&t.slots[s].lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (cur == 0ull) {
            if (!create) return SLOT_NONE;
            cur = atomicCAS(&t.slots[s].lo, 0ull, (unsigned long long)lo);
            if (cur == 0ull) cur = lo;
        }
        if (cur != lo) continue;
        unsigned long long h = __hip_atomic_load(&t.slots[s].hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (h == 0ull) {
            if (!create) {
                // slot claimed by a concurrent inserter of some key with the same low word: during a
                // build this cannot be told apart from "absent"; pure lookups never race with builds
                return SLOT_NONE;
            }
            h = atomicCAS(&t.slots[s].hi, 0ull, (unsigned long long)hi);
            if (h == 0ull) { h = hi; if (created) *created = true; }
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
    const uint64_t limit = t.mask < TABLE_MAX_PROBES ? t.mask : (uint64_t)TABLE_MAX_PROBES;
    for (uint64_t probes = 0; probes <= limit; probes++, s = (s + 1) & t.mask) {
        const TableSlot &sl = t.slots[s];
        unsigned long long cur = sl.lo;
        if (cur == 0ull) return false;
        if (cur == lo && sl.hi == hi) { val = sl.val; return true; }
    }
    return false;   // inserts never place a key beyond the probe limit
}

// Same, returning the slot (bit 31 set = side-list entry) or SLOT_NONE.
__device__ __forceinline__ uint32_t table_lookup_slot(const TableView &t, uint64_t lo, uint64_t hi) {
    if (lo == 0ull || hi == 0ull) {
        uint32_t n = *t.exc_n;
        for (uint32_t i = 0; i < n; i++)
            if (t.exc_lo[i] == lo && t.exc_hi[i] == hi) return 0x80000000u | i;
        return SLOT_NONE;
    }
    uint64_t s = table_home(lo, hi, t.mask);
    const uint64_t limit = t.mask < TABLE_MAX_PROBES ? t.mask : (uint64_t)TABLE_MAX_PROBES;
    for (uint64_t probes = 0; probes <= limit; probes++, s = (s + 1) & t.mask) {
        const TableSlot &sl = t.slots[s];
        unsigned long long cur = sl.lo;
        if (cur == 0ull) return SLOT_NONE;
        if (cur == lo && sl.hi == hi) return (uint32_t)s;
    }
    return SLOT_NONE;
}

__device__ __forceinline__ uint32_t table_slot_val(const TableView &t, uint32_t slot) {
    return (slot & 0x80000000u) ? t.exc_val[slot & 0x7FFFFFFFu] : t.slots[slot].val;
}

__device__ __forceinline__ uint32_t table_slot_rep(const TableView &t, uint32_t slot) {
    return (slot & 0x80000000u) ? t.exc_rep[slot & 0x7FFFFFFFu] : t.slots[slot].rep;
}

#endif  // __HIPCC__

// Host-side owner of the table storage.
struct DeviceTable {
    uint64_t cap = 0;
    DevBuf<TableSlot> slots;
    DevBuf<unsigned long long> exc_lo, exc_hi;
    DevBuf<uint32_t> exc_val, exc_rep, ctl;  // ctl: [0]=exc_n [1]=exc_lock [2]=overflow [4..4+TABLE_OCC_WAYS)=occupied counts

    int init(mdbg_ctx *ctx, uint64_t min_slots) {
        cap = 1024;
        while (cap < min_slots) cap <<= 1;
        if (cap > (1ull << 31)) return set_error(ctx, MDBG_ERANGE, "hash table of %llu slots exceeds 2^31", (unsigned long long)cap);
        MDBG_TRY(slots.alloc(ctx, cap));
        MDBG_TRY(exc_lo.alloc(ctx, TABLE_EXC_CAP));
        MDBG_TRY(exc_hi.alloc(ctx, TABLE_EXC_CAP));
        MDBG_TRY(exc_val.alloc(ctx, TABLE_EXC_CAP));
        MDBG_TRY(exc_rep.alloc(ctx, TABLE_EXC_CAP));
        MDBG_TRY(ctl.alloc(ctx, 4 + TABLE_OCC_WAYS));
        return clear(ctx);
    }
    int clear(mdbg_ctx *ctx) {
        LaunchTimer timer(ctx, "table_clear");
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(slots.p, 0, cap * sizeof(TableSlot), ctx->stream));
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(exc_val.p, 0, TABLE_EXC_CAP * 4, ctx->stream));
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(ctl.p, 0, (4 + TABLE_OCC_WAYS) * 4, ctx->stream));
        return MDBG_OK;
    }
    TableView view() const {
        TableView v;
        v.slots = slots.p; v.mask = cap - 1;
        v.exc_lo = exc_lo.p; v.exc_hi = exc_hi.p; v.exc_val = exc_val.p; v.exc_rep = exc_rep.p;
        v.exc_n = ctl.p; v.exc_lock = ctl.p + 1; v.overflow = ctl.p + 2; v.occ = ctl.p + 4;
        static const bool no_give_up = getenv("MDBG_NO_GIVE_UP") != nullptr;
        v.poll_overflow = no_give_up ? 0u : 1u;
        return v;
    }
    // 0 = fine, 1 = too full (caller grows and rebuilds), negative = error
    int overflowed(mdbg_ctx *ctx) {
        uint32_t c[4];
        MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, c, ctl.p, 16, hipMemcpyDeviceToHost));
        return c[2] ? 1 : 0;
    }
    // distinct keys in the table, after a pass that filled view().occ (slot_flag_kernel)
    int occupied(mdbg_ctx *ctx, uint64_t *n) {
        uint32_t c[TABLE_OCC_WAYS];
        MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, c, ctl.p + 4, sizeof(c), hipMemcpyDeviceToHost));
        uint64_t s = 0;
        for (uint32_t i = 0; i < TABLE_OCC_WAYS; i++) s += c[i];
        *n = s;
        return MDBG_OK;
    }
    int check_overflow(mdbg_ctx *ctx) {
        int o = overflowed(ctx);
        if (o < 0) return o;
        if (o) return set_error(ctx, MDBG_ERANGE, "k-min-mer hash table overflow (cap %llu)", (unsigned long long)cap);
        return MDBG_OK;
    }
};

// Build a table with a capacity guessed from `expected` keys and grow x4 until `fill` (which launches
// the insert kernels) completes without a probe sequence exceeding TABLE_MAX_PROBES.
template <typename Fill>
int build_table_adaptive(mdbg_ctx *ctx, DeviceTable &tab, uint64_t expected, uint64_t upper_bound, Fill fill) {
    uint64_t want = expected * 2 + 1024;
    const uint64_t most = upper_bound + upper_bound / 2 + 1024;   // load <= 2/3 even if every key is distinct
    if (want > most) want = most;
    for (;;) {
        MDBG_DBG(ctx, "build_table_adaptive: %llu slots (expected %llu keys, at most %llu)", (unsigned long long)want, (unsigned long long)expected, (unsigned long long)upper_bound);
        MDBG_TRY(tab.init(ctx, want));
        MDBG_DBG(ctx, "build_table_adaptive: table allocated");
        MDBG_TRY(fill(tab.view()));
        MDBG_HIP_CHECK(ctx, hipGetLastError());
        MDBG_DBG(ctx, "build_table_adaptive: fill launched");
        int o = tab.overflowed(ctx);
        MDBG_DBG(ctx, "build_table_adaptive: filled, overflow %d", o);
        if (o < 0) return o;
        if (!o) return MDBG_OK;
        // At load 2/3 a linear-probing run can still exceed TABLE_MAX_PROBES (nearly all keys distinct: ONT, the edge index),
        // so `most` is not an error yet: grow twice more (load <= 1/3, then <= 1/6) before giving up.
        if (tab.cap >= 4 * most) return set_error(ctx, MDBG_ERANGE, "k-min-mer hash table overflow at %llu slots", (unsigned long long)tab.cap);
        want = tab.cap >= most ? tab.cap * 2 : (tab.cap * 4 > most ? most : tab.cap * 4);
    }
}

}  // namespace mdbg
