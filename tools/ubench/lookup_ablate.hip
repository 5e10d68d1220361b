// lookup_ablate.hip -- what an index pass's look-up kernel (prev_abundance_u_kernel, csrc/kminmer.hip) is bound by, taken apart.
//
// Round-5 VERDICT item 4: the passes above firstK serve 37 - 40 G random 64-byte sectors a second where atomic_rates.hip reaches 55 - 58.
// This program builds a table of the library's own form (csrc/table.hpp: 32-byte slots, the real claim-and-publish insert) over the
// (k-1)-windows of a synthetic genome of minimizers, then runs the look-up over reads cut from that genome in variants that each take
// ONE thing away or change ONE thing, with the library's own window hash (csrc/kminmer_dev.hpp):
//   full        16 lanes a read, window hash, slot_load (16 + 4 bytes), one store per window            = the library's kernel
//   no_store    ... the answers summed into a register instead of stored
//   no_hash     ... the identity read from a precomputed array (16 B per window, coalesced) instead of hashed: memory only
//   flat        one lane per window over the precomputed identities: every lane of every wave busy
//   nt_streams  full, with non-temporal loads of minimizers / offsets and a non-temporal store of the answers
//   nt_table    full, with the slot itself loaded non-temporally
//   lanes8      8 lanes a read
//   lanes32     32 lanes a read
//   u2 / u4     two / four windows of a lane in flight
// and each of them at several grid sizes (blocks per CU).  The insert side: see insert_ablate below (same idea).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I metamdbg_amd/csrc tools/ubench/lookup_ablate.hip -o /tmp/lookup_ablate && /tmp/lookup_ablate [n_reads] [k-1] [genome] [load] [brief]
#include "kminmer_dev.hpp"
#include "table.hpp"

#include <algorithm>
#include <vector>

using namespace mdbg;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}

__global__ void make_genome(uint32_t *g, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) g[i] = (uint32_t)mix64(i * 2 + 12345);
}

// reads: len minimizers from a random start, half of them reversed
__global__ void make_reads(const uint32_t *g, uint64_t glen, uint32_t *mins, uint32_t n_reads, uint32_t len) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n_reads * len) return;
    const uint32_t r = (uint32_t)(i / len), j = (uint32_t)(i % len);
    const uint64_t h = mix64(r * 7919ull + 17);
    const uint64_t a = h % (glen - len);
    mins[i] = (h >> 63) ? g[a + len - 1 - j] : g[a + j];
}

__global__ __launch_bounds__(256) void fill_table(const uint32_t *g, uint64_t glen, uint32_t k, TableView t, uint32_t keep_mod) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i + k > glen) return;
    if (keep_mod && (mix64(i) % 16) >= keep_mod) return;           // some windows absent (look-ups that end at an empty slot)
    uint64_t hi, lo;
    window_hash_uniform(g + i, k, hi, lo);
    bool created = false;
    const uint32_t s = table_find_or_insert(t, lo, hi, true, &created);
    if (s != SLOT_NONE && created) t.slots[s].val = (uint32_t)(lo % 50) + 2;
}

__global__ __launch_bounds__(256) void precompute_ids(const uint32_t *mins, uint32_t n_reads, uint32_t len, uint32_t k, ulonglong2 *ids) {
    const uint32_t nw = len - k + 1;
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (uint64_t)n_reads * nw) return;
    const uint32_t r = (uint32_t)(i / nw), j = (uint32_t)(i % nw);
    uint64_t hi, lo;
    window_hash_uniform(mins + (uint64_t)r * len + j, k, hi, lo);
    ids[i] = make_ulonglong2(lo, hi);
}

enum { F_NO_STORE = 1, F_NO_HASH = 2, F_NT_STREAMS = 4, F_NT_TABLE = 8, F_PAIR = 16, F_STORE8 = 32, F_STORE16 = 64 };

template <typename T> __device__ __forceinline__ T ld(const T *p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }

template <int LANES, int U, int FLAGS>
__global__ __launch_bounds__(256) void lookup(const uint32_t *mins, uint32_t n_reads, uint32_t len, uint32_t k, const ulonglong2 *ids, TableView prev, uint32_t *out, uint32_t *sink) {
    const unsigned sub = threadIdx.x & (LANES - 1);
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / LANES;
    const uint32_t n = len - k + 1;
    uint32_t acc = 0;
    for (uint64_t r = group; r < n_reads; r += ngroups) {
        const uint64_t base = r * n;
        const uint32_t *m0 = mins + r * len;
        for (uint32_t i0 = sub; i0 < n; i0 += LANES * U) {
            uint64_t hi[U], lo[U], home[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = i0 + LANES * (uint32_t)u;
                hi[u] = lo[u] = 0; home[u] = 0;
                if (i < n) {
                    if (FLAGS & F_NO_HASH) { const ulonglong2 id = ids[base + i]; lo[u] = id.x; hi[u] = id.y; }
                    else if (FLAGS & F_NT_STREAMS) {
                        uint32_t w[16];
                        for (uint32_t j = 0; j < k; j++) w[j] = __builtin_nontemporal_load(m0 + i + j);
                        window_hash_uniform(w, k, hi[u], lo[u]);
                    } else window_hash_uniform(m0 + i, k, hi[u], lo[u]);
                    home[u] = table_home(lo[u], hi[u], prev.mask);
                }
            }
            SlotWords w[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (FLAGS & F_NT_TABLE) {
                    const TableSlot *p = &prev.slots[home[u]];
                    const unsigned long long a = __builtin_nontemporal_load(&p->lo), b = __builtin_nontemporal_load(&p->hi);
                    w[u].lo = a; w[u].hi = b; w[u].val = __builtin_nontemporal_load(&p->val);
                } else w[u] = slot_load(&prev.slots[home[u]]);
            }
            SlotWords w1[U];
            if (FLAGS & F_PAIR) {            // (homes are even: the second slot of the sector comes with the first)
#pragma unroll
                for (int u = 0; u < U; u++) w1[u] = slot_load(&prev.slots[home[u] + 1]);
#pragma unroll
                for (int u = 0; u < U; u++) asm volatile("" : "+v"(w1[u].val));
            }
#pragma unroll
            for (int u = 0; u < U; u++) asm volatile("" : "+v"(w[u].val));
#pragma unroll
            for (int u = 0; u < U; u++) {
                const uint32_t i = i0 + LANES * (uint32_t)u;
                uint32_t v = 1u;
                if (w[u].lo == lo[u] && w[u].hi == hi[u]) v = w[u].val;
                else if (FLAGS & F_PAIR) {
                    if (w[u].lo != 0ull) {
                        if (w1[u].lo == lo[u] && w1[u].hi == hi[u]) v = w1[u].val;
                        else if (w1[u].lo != 0ull) { uint32_t x; if (table_lookup_from(prev, table_next(home[u] + 1, prev.mask), 2, lo[u], hi[u], x)) v = x; }
                    }
                }
                else if (w[u].lo != 0ull) { uint32_t x; if (table_lookup_from(prev, table_next(home[u], prev.mask), 1, lo[u], hi[u], x)) v = x; }
                if (i < n) {
                    if (FLAGS & F_NO_STORE) acc += v;
                    else if (FLAGS & F_NT_STREAMS) __builtin_nontemporal_store(v, out + base + i);
                    else if (FLAGS & F_STORE8) reinterpret_cast<uint8_t *>(out)[base + i] = (uint8_t)(v < 255u ? v : 255u);       // a byte per window, 255 = look again
                    else if (FLAGS & F_STORE16) reinterpret_cast<uint16_t *>(out)[base + i] = (uint16_t)(v < 65535u ? v : 65535u);
                    else out[base + i] = v;
                }
            }
        }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// one lane per window over precomputed identities
template <int FLAGS>
__global__ __launch_bounds__(256) void lookup_flat(const ulonglong2 *ids, uint64_t n_inst, TableView prev, uint32_t *out) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = tid; i < n_inst; i += nth) {
        const ulonglong2 id = ids[i];
        const uint64_t home = table_home(id.x, id.y, prev.mask);
        const SlotWords w = slot_load(&prev.slots[home]);
        uint32_t v = 1u;
        if (w.lo == id.x && w.hi == id.y) v = w.val;
        else if (w.lo != 0ull) { uint32_t x; if (table_lookup_from(prev, table_next(home, prev.mask), 1, id.x, id.y, x)) v = x; }
        out[i] = v;
    }
}

// the index pass's insert (index_insert_u_kernel<1, true>): a first look at the home slot with plain loads, then claim-and-publish
template <int LANES, bool PAIR>
__global__ __launch_bounds__(256) void insert(const uint32_t *mins, uint32_t n_reads, uint32_t len, uint32_t k, const uint32_t *prev_ab, TableView t) {
    const unsigned sub = threadIdx.x & (LANES - 1);
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LANES;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) / LANES;
    const uint32_t n = len - k + 1;
    for (uint64_t r = group; r < n_reads; r += ngroups) {
        const uint32_t *m0 = mins + r * len;
        const uint64_t j0 = r * n;
        for (uint32_t i = sub; i < n; i += LANES) {
            const uint32_t a0 = prev_ab[j0 + i], a1 = prev_ab[j0 + (i + 1 < n ? i + 1 : i)];
            const uint32_t a = a0 < a1 ? a0 : a1;
            if (a <= 1u) continue;
            uint64_t hi, lo;
            window_hash_uniform(m0 + i, k, hi, lo);
            const uint64_t home = table_home(lo, hi, t.mask);
            if (PAIR) {
                const SlotPair p = pair_load(&t.slots[home]);
                if ((p.a.lo == lo && p.a.hi == hi) || (p.b.lo == lo && p.b.hi == hi)) continue;
            } else {
                const SlotWords w = slot_load(&t.slots[home]);
                if (w.lo == lo && w.hi == hi) continue;
            }
            bool created = false;
            const uint32_t s = table_find_or_insert(t, lo, hi, true, &created);
            if (s != SLOT_NONE && created) __hip_atomic_store(&t.slots[s].val, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms; }

int main(int argc, char **argv) {
    const uint32_t n_reads = argc > 1 ? (uint32_t)atoll(argv[1]) : 10000000u;
    const uint32_t k = argc > 2 ? (uint32_t)atoi(argv[2]) : 6u;          // the (k-1)-windows of pass k = 7
    const uint64_t glen = argc > 3 ? (uint64_t)atoll(argv[3]) : 14000000ull;
    const double load = argc > 4 ? atof(argv[4]) : 0.22;                 // keys per slot
    const bool brief = argc > 5;
    const uint32_t len = 37;
    const uint32_t nw = len - k + 1;
    const uint64_t n_inst = (uint64_t)n_reads * nw;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const unsigned n_cu = (unsigned)prop.multiProcessorCount;
    uint32_t *g, *mins, *out, *sink;
    ulonglong2 *ids;
    CK(hipMalloc(&g, glen * 4)); CK(hipMalloc(&mins, (uint64_t)n_reads * len * 4)); CK(hipMalloc(&out, n_inst * 4)); CK(hipMalloc(&sink, 64));
    CK(hipMalloc(&ids, n_inst * 16));
    make_genome<<<(unsigned)((glen + 255) / 256), 256>>>(g, glen);
    make_reads<<<(unsigned)(((uint64_t)n_reads * len + 255) / 256), 256>>>(g, glen, mins, n_reads, len);
    TableView t{};
    const uint64_t cap = ((uint64_t)((double)glen * 15 / 16 / load) + 255) / 256 * 256;
    CK(hipMalloc(&t.slots, cap * sizeof(TableSlot))); CK(hipMemset(t.slots, 0, cap * sizeof(TableSlot)));
    t.mask = cap - 1;
    uint32_t *ctl; CK(hipMalloc(&ctl, 4096)); CK(hipMemset(ctl, 0, 4096));
    unsigned long long *exc; CK(hipMalloc(&exc, 2 * 64 * 8));
    t.exc_lo = exc; t.exc_hi = exc + 64; t.exc_val = ctl + 512; t.exc_rep = ctl + 600; t.exc_n = ctl; t.exc_lock = ctl + 1; t.overflow = ctl + 2; t.occ = ctl + 4; t.poll_overflow = 0;
    fill_table<<<(unsigned)((glen + 255) / 256), 256>>>(g, glen, k, t, 15);
    precompute_ids<<<(unsigned)((n_inst + 255) / 256), 256>>>(mins, n_reads, len, k, ids);
    CK(hipDeviceSynchronize());
    uint32_t h_ctl[4]; CK(hipMemcpy(h_ctl, ctl, 16, hipMemcpyDeviceToHost));
    printf("# %u reads x %u minimizers, %u-windows: %llu look-ups; genome %llu, table %llu slots = %.2f GB, load %.2f, overflow %u; %u CUs\n", n_reads, len, k,
           (unsigned long long)n_inst, (unsigned long long)glen, (unsigned long long)cap, cap * 32 / 1e9, (double)glen * 15 / 16 / cap, h_ctl[2], n_cu);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<uint32_t> ref(1 << 20), got(1 << 20);
    bool have_ref = false;
    auto run = [&](const char *name, unsigned bpc, auto launch) {
        const unsigned grid = n_cu * bpc;
        float best = 1e9f;
        for (int it = 0; it < 3; it++) {
            CK(hipEventRecord(e0, 0));
            launch(grid);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            best = std::min(best, time_ms(e0, e1));
        }
        CK(hipGetLastError());
        printf("%-12s %3u blocks/CU  %7.3f ms  %6.2f G look-ups/s\n", name, bpc, best, n_inst / best / 1e6);
        fflush(stdout);
    };
    auto check = [&](const char *name) {
        CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
        if (!have_ref) { ref = got; have_ref = true; uint64_t hit = 0; for (auto v : ref) hit += v != 1; printf("# %.1f %% of the look-ups find their key\n", 100.0 * hit / ref.size()); }
        else if (ref != got) printf("!! %s: answers differ\n", name);
        CK(hipMemset(out, 0, got.size() * 4));
    };
#define RUN(name, L, U, F, bpc) run(name, bpc, [&](unsigned grid) { lookup<L, U, F><<<grid, 256>>>(mins, n_reads, len, k, ids, t, out, sink); })
    for (unsigned bpc : {4u, 8u, 16u, 32u}) { if (brief && bpc != 8u) continue; RUN("full", 16, 1, 0, bpc); }
    check("full");
    if (brief) { RUN("pair_eager", 16, 1, F_PAIR, 8); check("pair_eager"); RUN("pair_e_l32", 32, 1, F_PAIR, 8); check("pair_e_l32");
                 RUN("pair_store8", 16, 1, F_PAIR | F_STORE8, 8); RUN("pair_store16", 16, 1, F_PAIR | F_STORE16, 8); RUN("pair_nostore", 16, 1, F_PAIR | F_NO_STORE, 8); RUN("pair_eager", 16, 1, F_PAIR, 8); }
    for (unsigned bpc : {8u, 32u}) {
        if (brief) break;
        RUN("no_store", 16, 1, F_NO_STORE, bpc);
        RUN("no_hash", 16, 1, F_NO_HASH, bpc); check("no_hash");
        RUN("no_hash_ns", 16, 1, F_NO_HASH | F_NO_STORE, bpc);
        run("flat", bpc, [&](unsigned grid) { lookup_flat<0><<<grid, 256>>>(ids, n_inst, t, out); }); check("flat");
        RUN("nt_streams", 16, 1, F_NT_STREAMS, bpc); check("nt_streams");
        RUN("nt_table", 16, 1, F_NT_TABLE, bpc); check("nt_table");
        RUN("nt_both", 16, 1, F_NT_TABLE | F_NT_STREAMS, bpc); check("nt_both");
        RUN("lanes8", 8, 1, 0, bpc); check("lanes8");
        RUN("lanes32", 32, 1, 0, bpc); check("lanes32");
        RUN("u2", 16, 2, 0, bpc); check("u2");
        RUN("lanes8_u2", 8, 2, 0, bpc); check("lanes8_u2");
        RUN("lanes8_u4", 8, 4, 0, bpc); check("lanes8_u4");
        RUN("pair_eager", 16, 1, F_PAIR, bpc); check("pair_eager"); RUN("pair_e_l32", 32, 1, F_PAIR, bpc); check("pair_e_l32");
    }
    // the insert side: the same windows into a fresh table of the same size (the answers above as the previous abundances)
    RUN("full", 16, 1, 0, 8);
    TableView t2 = t;
    CK(hipMalloc(&t2.slots, cap * sizeof(TableSlot)));
    {
        float best = 1e9f;
        for (int it = 0; it < 3; it++) {
            CK(hipEventRecord(e0, 0));
            CK(hipMemsetAsync(t2.slots, 0, cap * sizeof(TableSlot), 0));
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            best = std::min(best, time_ms(e0, e1));
        }
        printf("%-12s                %7.3f ms\n", "clear", best);
    }
    for (unsigned bpc : {8u, 32u}) {
        if (brief && bpc != 8u) continue;
        for (int variant = 0; variant < 3; variant++) {
        float best = 1e9f;
        for (int it = 0; it < 3; it++) {
            CK(hipMemset(t2.slots, 0, cap * sizeof(TableSlot)));
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            if (variant == 0) insert<16, false><<<n_cu * bpc, 256>>>(mins, n_reads, len, k, out, t2);
            else if (variant == 1) insert<16, true><<<n_cu * bpc, 256>>>(mins, n_reads, len, k, out, t2);
            else insert<32, true><<<n_cu * bpc, 256>>>(mins, n_reads, len, k, out, t2);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            best = std::min(best, time_ms(e0, e1));
        }
        printf("%-12s %3u blocks/CU  %7.3f ms  %6.2f G inserts/s\n", variant == 0 ? "insert" : variant == 1 ? "insert_pair" : "insert_p_l32", bpc, best, n_inst / best / 1e6);
        }
    }
    return 0;
}
