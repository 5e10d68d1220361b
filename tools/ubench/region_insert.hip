// region_insert.hip -- does the first-pass insert get faster when the slots it touches are confined to a region of the table at a time?
//
// count_insert_kernel runs at the rate at which HBM serves random 64-byte sectors (45.7 GB of traffic for
// 7.2 GB of algorithmic bytes over a 1 GB table).  The proposed redesign writes the instances' identities out grouped by table region and
// inserts region by region, each region small enough for the 256 MB memory-side cache.  This program measures the two halves of that
// idea BEFORE anything is built on it:
//   insert   the current insert's memory pattern (two agent-scope 8-byte loads of the slot, then a device-scope atomicAdd: mode 3 of
//            atomic_rates.hip) over n_ops operations on n_keys keys, with the slots confined to one region of R bytes at a time, for
//            R = the whole table down to 16 MB -- the knee, if there is one, says what region size pays;
//   group    the cost of writing n_ops 24-byte records (128-bit identity + instance index) to P output streams chosen by the slot's
//            region (per-wave aggregated cursors), and of reading them back: what the grouping adds.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/region_insert.hip -o /tmp/region_insert && /tmp/region_insert [keys] [operations] [log2 slots]
//   (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes gives the traffic of each launch)
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

struct alignas(32) Slot { unsigned long long lo, hi; uint32_t val, rep, pad[2]; };
struct Rec { unsigned long long lo, hi; uint32_t inst, pad; };          // what a grouped instance would carry

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}

// keys [key0, key0 + n_keys) live in slots [slot0, slot0 + region_slots): one region's share of the table
__global__ void fill(Slot *t, uint64_t slot0, uint64_t region_slots, uint64_t key0, uint64_t n_keys) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_keys) return;
    const uint64_t key = key0 + i;
    t[slot0 + (mix(key * 0x9E3779B97F4A7C15ull + 1) & (region_slots - 1))].lo = key;
}

// n_ops inserts on the keys of one region, in random order (the order instances arrive in: by read, not by key)
__global__ __launch_bounds__(256) void insert(Slot *t, uint64_t slot0, uint64_t region_slots, uint64_t key0, uint64_t n_keys, uint64_t n_ops, uint32_t *sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t i = tid; i < n_ops; i += nth) {
        const uint64_t key = key0 + mix(i + slot0) % n_keys;
        Slot *p = t + slot0 + (mix(key * 0x9E3779B97F4A7C15ull + 1) & (region_slots - 1));
        const unsigned long long a = __hip_atomic_load(&p->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(&p->hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (a == key && b == 0) atomicAdd(&p->val, 1u); else acc++;     // (collisions of the toy table: counted, not probed)
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

// every operation's record to the stream of its region: a wave's lanes that go to the same region take their places with one atomic
__global__ __launch_bounds__(256) void group(Rec *out, unsigned long long *cursor, uint64_t stream_cap, uint32_t region_shift, uint64_t table_slots, uint64_t n_keys, uint64_t n_ops) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = tid - (tid & 63); i0 < n_ops; i0 += nth) {
        const uint64_t i = i0 + (tid & 63);
        const bool live = i < n_ops;
        const uint64_t key = mix(i) % n_keys, slot = mix(key * 0x9E3779B97F4A7C15ull + 1) & (table_slots - 1);
        const uint32_t region = (uint32_t)(slot >> region_shift);
        // lanes of the wave with the same region: the lowest takes room for all of them
        uint64_t todo = __ballot(live);
        uint64_t place = 0;
        while (todo) {
            const int leader = __ffsll((long long)todo) - 1;
            const uint32_t r = __shfl(region, leader, 64);
            const uint64_t same = __ballot(live && region == r) & todo;
            unsigned long long base = 0;
            if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(&cursor[r], (unsigned long long)__popcll(same));
            base = __shfl(base, leader, 64);
            if (live && region == r) place = base + __popcll(same & ((1ull << (threadIdx.x & 63)) - 1));
            todo &= ~same;
        }
        if (live && place < stream_cap) out[(uint64_t)region * stream_cap + place] = Rec{key, 0, (uint32_t)i, 0};
    }
}

__global__ __launch_bounds__(256) void read_back(const Rec *in, uint64_t n, uint32_t *sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (uint64_t i = tid; i < n; i += nth) acc += (uint32_t)in[i].lo + in[i].inst;
    if (acc == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv) {
    // defaults = the first pass of the bench batch (10 M x 10 kb reads): 134 M slots of 32 bytes = 4.3 GB, 38.8 M distinct
    // keys, 344 M instances.  (The one run there has been so far -- profiles/round3_final_region_insert_ubench.txt -- was on 2^25 slots
    // and 7 M keys: region_insert 7000000 344000000 25.)
    const uint64_t n_keys = argc > 1 ? strtoull(argv[1], nullptr, 10) : 38800000ull;
    const uint64_t n_ops = argc > 2 ? strtoull(argv[2], nullptr, 10) : 344000000ull;
    const uint32_t log_slots = argc > 3 ? (uint32_t)atoi(argv[3]) : 27;
    const uint64_t table_slots = 1ull << log_slots;
    Slot *t = nullptr;
    uint32_t *sink = nullptr;
    CK(hipMalloc(&t, table_slots * sizeof(Slot)));
    CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int n_cu = 0;
    CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    const unsigned grid = (unsigned)n_cu * 8;
    printf("table %.0f MB, %llu keys, %llu operations, grid %u x 256\n", table_slots * 32 / 1048576.0, (unsigned long long)n_keys, (unsigned long long)n_ops, grid);
    for (uint64_t region_bytes = table_slots * 32; region_bytes >= (1ull << 24); region_bytes >>= 1) {
        const uint64_t region_slots = region_bytes / 32, n_regions = table_slots / region_slots;
        const uint64_t keys_per = n_keys / n_regions, ops_per = n_ops / n_regions;
        CK(hipMemset(t, 0, table_slots * sizeof(Slot)));
        for (uint64_t r = 0; r < n_regions; r++)
            hipLaunchKernelGGL(fill, dim3((unsigned)((keys_per + 255) / 256)), dim3(256), 0, 0, t, r * region_slots, region_slots, r * keys_per, keys_per);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0));
            for (uint64_t r = 0; r < n_regions; r++)
                hipLaunchKernelGGL(insert, dim3(grid), dim3(256), 0, 0, t, r * region_slots, region_slots, r * keys_per, keys_per, ops_per, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("insert, regions of %5llu MB (%3llu launches): %7.2f ms = %5.1f G operations/s\n", (unsigned long long)(region_bytes >> 20), (unsigned long long)n_regions, best,
               n_ops / best / 1e6);
    }
    // the grouping: 8 regions of 128 MB
    {
        const uint32_t P = 8, region_shift = log_slots - 3;
        const uint64_t stream_cap = n_ops / P + n_ops / (P * 4);
        Rec *out = nullptr;
        unsigned long long *cursor = nullptr;
        CK(hipMalloc(&out, (uint64_t)P * stream_cap * sizeof(Rec)));
        CK(hipMalloc(&cursor, P * 8));
        float best = 1e30f, best_r = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipMemset(cursor, 0, P * 8));
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(group, dim3(grid), dim3(256), 0, 0, out, cursor, stream_cap, region_shift, table_slots, n_keys, n_ops);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(read_back, dim3(grid), dim3(256), 0, 0, out, (uint64_t)P * stream_cap, sink);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best_r) best_r = ms;
        }
        std::vector<unsigned long long> h(P);
        CK(hipMemcpy(h.data(), cursor, P * 8, hipMemcpyDeviceToHost));
        unsigned long long total = 0;
        for (auto v : h) total += v;
        printf("group into %u streams of 24-byte records: %7.2f ms (%.1f GB written, %llu records placed); reading them back: %7.2f ms\n", P, best,
               total * 24 / 1e9, total, best_r);
    }
    return 0;
}
