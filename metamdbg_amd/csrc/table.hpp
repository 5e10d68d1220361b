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
    uint64_t mask;            // cap - 1: the last slot (cap is any even number, see table_home)
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

// A table has ANY even number of slots (round 6; it was a power of two): the home slot is a multiply-shift of the folded key, not a mask,
// so a table is sized for the load the passes run best at whatever the key count (tools/ubench/lookup_ablate.hip: a pass's look-ups run
// 12 - 18 % faster at load 0.2 than at 0.4, and a power of two leaves the load anywhere in a factor of two).  A probe sequence starts
// at the EVEN slot of a 64-byte sector: a key's first two probes are one sector, and the look-up kernels fetch both slots at once.
// `mask` keeps its name: it is the index of the LAST slot (cap - 1).
__device__ __forceinline__ uint64_t table_home(uint64_t lo, uint64_t hi, uint64_t mask) {
    // the key is already a Murmur3 output; fold both words so owner-rank partitioning by the top
    // bits of `hi` (multi-GPU) does not correlate with the slot
    return __umul64hi(lo ^ (hi >> 17), mask + 1) & ~1ull;
}
__device__ __forceinline__ uint64_t table_next(uint64_t s, uint64_t mask) { return s == mask ? 0 : s + 1; }

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
    for (uint64_t probes = 0; probes <= limit; probes++, s = table_next(s, t.mask)) {
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

// A slot's key and value in ONE round trip: two loads issued together (the field-by-field form -- low word, then the high word if it
// matched, then the value -- was three dependent trips to the same sector: the ISA of round 4's prev_abundance_kernel).
struct SlotWords { unsigned long long lo, hi; uint32_t val; };
__device__ __forceinline__ SlotWords slot_load(const TableSlot *p) {
    const uint4 a = *reinterpret_cast<const uint4 *>(p);
    const uint32_t v = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const unsigned char *>(p) + 16);
    SlotWords w;
    w.lo = (unsigned long long)a.x | ((unsigned long long)a.y << 32);
    w.hi = (unsigned long long)a.z | ((unsigned long long)a.w << 32);
    w.val = v;
    return w;
}

__device__ __forceinline__ bool table_lookup_side(const TableView &t, uint64_t lo, uint64_t hi, uint32_t &val) {
    const uint32_t n = *t.exc_n;
    for (uint32_t i = 0; i < n; i++)
        if (t.exc_lo[i] == lo && t.exc_hi[i] == hi) { val = t.exc_val[i]; return true; }
    return false;
}

// the probe sequence from slot s on (the key has no zero word)
__device__ __forceinline__ bool table_lookup_from(const TableView &t, uint64_t s, uint64_t probes, uint64_t lo, uint64_t hi, uint32_t &val) {
    const uint64_t limit = t.mask < TABLE_MAX_PROBES ? t.mask : (uint64_t)TABLE_MAX_PROBES;
    for (; probes <= limit; probes++, s = table_next(s, t.mask)) {
        const SlotWords w = slot_load(&t.slots[s]);
        if (w.lo == 0ull) return false;
        if (w.lo == lo && w.hi == hi) { val = w.val; return true; }
    }
    return false;   // inserts never place a key beyond the probe limit
}

// Both slots of a key's home sector in one round trip, and the verdict they allow: 1 found (val set), 0 absent, -1 look further (from
// the next sector on, two probes done).  A key sits in the second slot only if the first was taken when it arrived, and slots are never
// emptied: an empty first slot ends the search.
struct SlotPair { SlotWords a, b; };
__device__ __forceinline__ SlotPair pair_load(const TableSlot *home) {
    SlotPair p;
    p.a = slot_load(home);
    p.b = slot_load(home + 1);
    return p;
}
__device__ __forceinline__ int pair_verdict(const SlotPair &p, uint64_t lo, uint64_t hi, uint32_t &val) {
    if (p.a.lo == lo && p.a.hi == hi) { val = p.a.val; return 1; }
    if (p.a.lo == 0ull) return 0;
    if (p.b.lo == lo && p.b.hi == hi) { val = p.b.val; return 1; }
    if (p.b.lo == 0ull) return 0;
    return -1;
}

// Read-only lookup after the build kernel completed (plain loads are safe across a kernel boundary).
__device__ __forceinline__ bool table_lookup(const TableView &t, uint64_t lo, uint64_t hi, uint32_t &val) {
    if (lo == 0ull || hi == 0ull) return table_lookup_side(t, lo, hi, val);
    const uint64_t s = table_home(lo, hi, t.mask);
    const int v = pair_verdict(pair_load(&t.slots[s]), lo, hi, val);
    if (v >= 0) return v != 0;
    return table_lookup_from(t, table_next(s + 1, t.mask), 2, lo, hi, val);
}

// (the field-by-field form, kept for A/B timing: mdbg_set_option "index_tuning")
__device__ __forceinline__ bool table_lookup_narrow(const TableView &t, uint64_t lo, uint64_t hi, uint32_t &val) {
    if (lo == 0ull || hi == 0ull) return table_lookup_side(t, lo, hi, val);
    uint64_t s = table_home(lo, hi, t.mask);
    const uint64_t limit = t.mask < TABLE_MAX_PROBES ? t.mask : (uint64_t)TABLE_MAX_PROBES;
    for (uint64_t probes = 0; probes <= limit; probes++, s = table_next(s, t.mask)) {
        const TableSlot &sl = t.slots[s];
        unsigned long long cur = sl.lo;
        if (cur == 0ull) return false;
        if (cur == lo && sl.hi == hi) { val = sl.val; return true; }
    }
    return false;
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
    for (uint64_t probes = 0; probes <= limit; probes++, s = table_next(s, t.mask)) {
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

// ---- the compact form: buckets of three keys in one 64-byte sector -------------------------------------------------
// The passes of the multi-k loop above firstK are look-ups: one per (k-1)-window into the previous table (getPrevAbundances,
// graph/CreateMdbg.hpp:1240-1265) and one insert-if-absent per k-window (:1450-1459), 344 M of each at 10 M reads, nineteen in
// twenty of the inserts meeting a key that is already there.  In the one-slot table above each is a random 64-byte sector of a
// 1 - 2 GB array (12 - 16 M keys in 33 - 67 M 32-byte slots: load 0.18 - 0.48, the power-of-two capacity and the `rep` word
// nobody reads at k >= firstK + 2).  Here a sector holds THREE keys -- low words, high words, values, side by side, every field
// aligned for the same claim-then-publish atomics -- the bucket count is any number (multiply-shift, not a mask), and the load is
// two thirds: the same keys in a third of the bytes, 0.3 - 0.45 GB, most of which the 256 MB memory-side cache keeps.  One probe
// is still one sector; a bucket whose last entry is empty ends the sequence (entries fill in order).
constexpr uint32_t BUCKET_WAYS = 3;
constexpr uint32_t BUCKET_MAX_PROBES = 256;     // buckets: longer sequences mean the table is too full (grow and rebuild).  (24 -- "as many entries as
                                                // TABLE_MAX_PROBES slots" -- overflowed at two thirds full: a run of 25 full buckets is 75 keys where 47 are
                                                // expected, one bucket in a thousand starts one; 257 full buckets at that load do not happen)

struct alignas(64) KeyBucket {
    unsigned long long lo[BUCKET_WAYS];         // 0 = empty
    unsigned long long hi[BUCKET_WAYS];         // 0 = not yet published
    uint32_t val[BUCKET_WAYS];
    uint32_t spare;
};
static_assert(sizeof(KeyBucket) == 64, "a bucket is one 64-byte sector");

struct BucketView {
    KeyBucket *b;
    uint64_t nb;              // buckets
    uint32_t *rep;            // nb * BUCKET_WAYS representatives (the refined pass writes vectors), or null
    TableView side;           // the side list for keys with a zero word, the overflow flag, the occupancy counters (slots == null)
};

#ifdef __HIPCC__

__device__ __forceinline__ uint64_t bucket_home(uint64_t lo, uint64_t hi, uint64_t nb) {
    return __umul64hi(lo ^ (hi >> 17), nb);     // the key is a Murmur3 output: its top bits are as good as any
}

// Find or create the entry of (lo, hi): its index bucket * 3 + way (bit 31 set = side list entry), or SLOT_NONE when create ==
// false and the key is absent / the probe limit was reached.  *created: THIS call published the key (exactly one caller per key).
__device__ __forceinline__ uint32_t bucket_find_or_insert(const BucketView &t, uint64_t lo, uint64_t hi, bool create, bool *created = nullptr) {
    uint64_t b = bucket_home(lo, hi, t.nb);
    const uint64_t limit = t.nb - 1 < BUCKET_MAX_PROBES ? t.nb - 1 : (uint64_t)BUCKET_MAX_PROBES;
    for (uint64_t probes = 0; probes <= limit; probes++, b = b + 1 == t.nb ? 0 : b + 1) {
        KeyBucket &B = t.b[b];
#pragma unroll
        for (uint32_t j = 0; j < BUCKET_WAYS; j++) {
            unsigned long long cur = __hip_atomic_load(&B.lo[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == 0ull) {
                if (!create) return SLOT_NONE;
                cur = atomicCAS(&B.lo[j], 0ull, (unsigned long long)lo);
                if (cur == 0ull) cur = lo;
            }
            if (cur != lo) continue;
            unsigned long long h = __hip_atomic_load(&B.hi[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (h == 0ull) {
                if (!create) return SLOT_NONE;      // (as in table_find_or_insert: pure look-ups never race with builds)
                h = atomicCAS(&B.hi[j], 0ull, (unsigned long long)hi);
                if (h == 0ull) { h = hi; if (created) *created = true; }
            }
            if (h == hi) return (uint32_t)(b * BUCKET_WAYS + j);
        }
    }
    if (create) atomicExch(t.side.overflow, 1u);
    return SLOT_NONE;
}

// insert-if-absent with a value that is a function of the key (see table_insert_once); `rep` is stored when the view keeps them
__device__ __forceinline__ uint32_t bucket_insert_once(const BucketView &t, uint64_t lo, uint64_t hi, uint32_t v, uint32_t rep) {
    if (lo == 0ull || hi == 0ull) return table_exc_upsert(t.side, lo, hi, 0, v, true, rep, true);
    bool created = false;
    const uint32_t e = bucket_find_or_insert(t, lo, hi, true, &created);
    if (e != SLOT_NONE && created) {
        __hip_atomic_store(&t.b[e / BUCKET_WAYS].val[e % BUCKET_WAYS], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t.rep) t.rep[e] = rep;
    }
    return e;
}

// insert or overwrite (rows of a finished table, each key once)
__device__ __forceinline__ uint32_t bucket_upsert_set(const BucketView &t, uint64_t lo, uint64_t hi, uint32_t v) {
    if (lo == 0ull || hi == 0ull) return table_exc_upsert(t.side, lo, hi, 0, v, true, 0u, true);
    const uint32_t e = bucket_find_or_insert(t, lo, hi, true);
    if (e != SLOT_NONE) __hip_atomic_store(&t.b[e / BUCKET_WAYS].val[e % BUCKET_WAYS], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return e;
}

// Read-only look-up after the building kernel has finished: the three low words of a bucket in two loads, the matching way's
// high word and value from the same sector.
__device__ __forceinline__ bool bucket_lookup(const BucketView &t, uint64_t lo, uint64_t hi, uint32_t &val) {
    if (lo == 0ull || hi == 0ull) {
        const uint32_t n = *t.side.exc_n;
        for (uint32_t i = 0; i < n; i++)
            if (t.side.exc_lo[i] == lo && t.side.exc_hi[i] == hi) { val = t.side.exc_val[i]; return true; }
        return false;
    }
    uint64_t b = bucket_home(lo, hi, t.nb);
    const uint64_t limit = t.nb - 1 < BUCKET_MAX_PROBES ? t.nb - 1 : (uint64_t)BUCKET_MAX_PROBES;
    for (uint64_t probes = 0; probes <= limit; probes++, b = b + 1 == t.nb ? 0 : b + 1) {
        const KeyBucket &B = t.b[b];
        const ulonglong2 l01 = *reinterpret_cast<const ulonglong2 *>(&B.lo[0]);
        const unsigned long long l2 = B.lo[2];
        if (l01.x == lo && B.hi[0] == hi) { val = B.val[0]; return true; }
        if (l01.y == lo && B.hi[1] == hi) { val = B.val[1]; return true; }
        if (l2 == lo && B.hi[2] == hi) { val = B.val[2]; return true; }
        if (l2 == 0ull) return false;           // not full: nothing of this home bucket went further
    }
    return false;
}

// one look-up, whatever the table's form
__device__ __forceinline__ bool key_lookup(const TableView &t, uint64_t lo, uint64_t hi, uint32_t &val) { return table_lookup(t, lo, hi, val); }
__device__ __forceinline__ bool key_lookup(const BucketView &t, uint64_t lo, uint64_t hi, uint32_t &val) { return bucket_lookup(t, lo, hi, val); }

#endif  // __HIPCC__

// Host-side owner of the table storage.
struct DeviceTable {
    uint64_t cap = 0;
    DevBuf<TableSlot> slots;
    DevBuf<unsigned long long> exc_lo, exc_hi;
    DevBuf<uint32_t> exc_val, exc_rep, ctl;  // ctl: [0]=exc_n [1]=exc_lock [2]=overflow [4..4+TABLE_OCC_WAYS)=occupied counts

    int init(mdbg_ctx *ctx, uint64_t min_slots) {
        cap = min_slots < 1024 ? 1024 : (min_slots + 255) / 256 * 256;     // (a multiple of 256: whole blocks walk whole sectors)
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
// slots for `keys` keys at the load the passes run best at (MDBG_TABLE_LOAD_PCT, default 22: DESIGN.md 4.2), never more than the 2^31
// a table may have (then the load is what it is; inserts report an overflow when a probe sequence gets too long)
// (dense = true: a table whose every slot is walked afterwards more than once -- the refined pass looks two keys up per slot, flags and
// emits; the counting passes likewise -- is sized for load 0.35: at 0.22 the refined pass of configs[2] took 22.9 ms instead of 20.7, the
// walks over twice the slots costing more than the inserts gained, profiles/round6_a_index_passes_table_load_*.json.  Not 0.45: with
// 52 M keys the longest probe sequence of a table that full comes within reach of TABLE_MAX_PROBES -- one build in three then overflowed,
// grew fourfold and ran again, 165 ms for a 20 ms pass: profiles/round6_i_bench_detail.json `legs.multik.ms.k5`)
inline uint64_t table_slots_for(uint64_t keys, bool dense = false) {
    static const unsigned pct = [] { const char *e = getenv("MDBG_TABLE_LOAD_PCT"); const int v = e ? atoi(e) : 0; return v >= 5 && v <= 60 ? (unsigned)v : 22u; }();
    const uint64_t want = keys * 100 / (dense ? 35u : pct) + 1024;
    return want > (1ull << 31) ? (1ull << 31) : want;
}

template <typename Fill>
int build_table_adaptive(mdbg_ctx *ctx, DeviceTable &tab, uint64_t expected, uint64_t upper_bound, Fill fill, bool dense = false) {
    uint64_t want = table_slots_for(expected, dense);
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

// Host-side owner of a bucket table.
struct BucketTable {
    uint64_t nb = 0;
    DevBuf<KeyBucket> buckets;
    DevBuf<uint32_t> rep;
    DevBuf<unsigned long long> exc_lo, exc_hi;
    DevBuf<uint32_t> exc_val, exc_rep, ctl;  // ctl as DeviceTable's
    bool with_rep = false;

    static uint64_t buckets_for(uint64_t keys, double load) {
        const uint64_t n = (uint64_t)((double)keys / (load * BUCKET_WAYS)) + 1;
        return n < 1024 ? 1024 : (n + 255) / 256 * 256;      // (a multiple of 256: whole blocks walk whole buckets)
    }
    uint64_t entries() const { return nb * BUCKET_WAYS; }
    int init(mdbg_ctx *ctx, uint64_t n_buckets, bool keep_rep) {
        nb = n_buckets < 1024 ? 1024 : (n_buckets + 255) / 256 * 256;
        if (nb * BUCKET_WAYS >= (1ull << 31)) return set_error(ctx, MDBG_ERANGE, "bucket table of %llu entries exceeds 2^31", (unsigned long long)(nb * BUCKET_WAYS));
        with_rep = keep_rep;
        MDBG_TRY(buckets.alloc(ctx, nb));
        if (keep_rep) MDBG_TRY(rep.alloc(ctx, nb * BUCKET_WAYS));
        MDBG_TRY(exc_lo.alloc(ctx, TABLE_EXC_CAP));
        MDBG_TRY(exc_hi.alloc(ctx, TABLE_EXC_CAP));
        MDBG_TRY(exc_val.alloc(ctx, TABLE_EXC_CAP));
        MDBG_TRY(exc_rep.alloc(ctx, TABLE_EXC_CAP));
        MDBG_TRY(ctl.alloc(ctx, 4 + TABLE_OCC_WAYS));
        LaunchTimer timer(ctx, "table_clear");
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(buckets.p, 0, nb * sizeof(KeyBucket), ctx->stream));
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(exc_val.p, 0, TABLE_EXC_CAP * 4, ctx->stream));
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(ctl.p, 0, (4 + TABLE_OCC_WAYS) * 4, ctx->stream));
        return MDBG_OK;
    }
    BucketView view() const {
        BucketView v;
        v.b = buckets.p; v.nb = nb; v.rep = with_rep ? rep.p : nullptr;
        v.side.slots = nullptr; v.side.mask = 0;
        v.side.exc_lo = exc_lo.p; v.side.exc_hi = exc_hi.p; v.side.exc_val = exc_val.p; v.side.exc_rep = exc_rep.p;
        v.side.exc_n = ctl.p; v.side.exc_lock = ctl.p + 1; v.side.overflow = ctl.p + 2; v.side.occ = ctl.p + 4;
        static const bool no_give_up = getenv("MDBG_NO_GIVE_UP") != nullptr;
        v.side.poll_overflow = no_give_up ? 0u : 1u;
        return v;
    }
    int overflowed(mdbg_ctx *ctx) {
        uint32_t c[4];
        MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, c, ctl.p, 16, hipMemcpyDeviceToHost));
        return c[2] ? 1 : 0;
    }
    int check_overflow(mdbg_ctx *ctx) {
        int o = overflowed(ctx);
        if (o < 0) return o;
        if (o) return set_error(ctx, MDBG_ERANGE, "k-min-mer bucket table overflow (%llu buckets)", (unsigned long long)nb);
        return MDBG_OK;
    }
};

// The same growth rule as build_table_adaptive for a bucket table: sized for `expected` keys at two thirds full, doubled until `fill`
// completes without a probe sequence beyond BUCKET_MAX_PROBES; at most `upper_bound` keys can arrive.
template <typename Fill>
int build_buckets_adaptive(mdbg_ctx *ctx, BucketTable &tab, uint64_t expected, uint64_t upper_bound, bool keep_rep, Fill fill) {
    uint64_t want = BucketTable::buckets_for(expected + 1024, 0.66);
    const uint64_t most = BucketTable::buckets_for(upper_bound + 1024, 0.66);
    if (want > most) want = most;
    for (;;) {
        MDBG_DBG(ctx, "build_buckets_adaptive: %llu buckets (expected %llu keys, at most %llu)", (unsigned long long)want, (unsigned long long)expected, (unsigned long long)upper_bound);
        MDBG_TRY(tab.init(ctx, want, keep_rep));
        MDBG_TRY(fill(tab.view()));
        MDBG_HIP_CHECK(ctx, hipGetLastError());
        int o = tab.overflowed(ctx);
        if (o < 0) return o;
        if (!o) return MDBG_OK;
        if (tab.nb >= 4 * most) return set_error(ctx, MDBG_ERANGE, "k-min-mer bucket table overflow at %llu buckets", (unsigned long long)tab.nb);
        want = tab.nb >= most ? tab.nb * 2 : (tab.nb * 2 > most ? most : tab.nb * 2);
    }
}

}  // namespace mdbg
