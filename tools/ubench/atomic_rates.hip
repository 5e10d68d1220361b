// atomic_rates.hip -- what a random-access hash-table update costs on gfx950 (8 XCDs, one L2 each):
// device-scope atomics / loads (coherent across XCDs, resolved beyond the L2) against XCD-local ones
// (workgroup-scope RMW + plain loads on an address range only one XCD touches).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/atomic_rates.hip -o /tmp/atomic_rates && /tmp/atomic_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

struct alignas(32) Slot { unsigned long long lo, hi; uint32_t val, rep, pad[2]; };

__device__ __forceinline__ uint64_t mix(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33; return x;
}

__device__ __forceinline__ uint32_t xcc_id() {
    uint32_t v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xFu;
}

// mode 0: device-scope atomicAdd        1: workgroup-scope RMW (any address)   2: workgroup-scope RMW, XCD-local 1/8 of the table
// mode 3: 2 agent-scope loads + device atomicAdd (the current insert)          4: plain 16 B load + workgroup RMW, XCD-local
// mode 5: agent-scope 8 B load only     6: plain 8 B load only                 7: plain load, XCD-local
template <int MODE>
__global__ __launch_bounds__(256) void k(Slot *t, uint64_t mask, uint64_t n_ops, uint32_t n_keys, uint32_t *sink) {
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t part = (mask + 1) >> 3;
    const uint32_t x = (MODE == 2 || MODE == 4 || MODE == 7) ? xcc_id() : 0u;
    uint32_t acc = 0;
    for (uint64_t i = tid; i < n_ops; i += nth) {
        uint64_t key = mix(i) % n_keys;              // ~n_ops / n_keys hits per key, like 30x coverage
        uint64_t s = mix(key * 0x9E3779B97F4A7C15ull + 1) & mask;
        if (MODE == 2 || MODE == 4 || MODE == 7) s = (s & (part - 1)) + x * part;
        Slot *p = t + s;
        if (MODE == 0) atomicAdd(&p->val, 1u);
        if (MODE == 8) acc += atomicAdd(&p->val, 1u);
        if (MODE == 1 || MODE == 2) (void)__hip_atomic_fetch_add(&p->val, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (MODE == 3) {
            unsigned long long a = __hip_atomic_load(&p->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned long long b = __hip_atomic_load(&p->hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a == key && b == 0) atomicAdd(&p->val, 1u); else acc++;
        }
        if (MODE == 4) {
            ulonglong2 ab = *reinterpret_cast<const ulonglong2 *>(p);
            if (ab.x == key && ab.y == 0) (void)__hip_atomic_fetch_add(&p->val, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); else acc++;
        }
        if (MODE == 5) acc += (uint32_t)__hip_atomic_load(&p->lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (MODE == 6 || MODE == 7) acc += (uint32_t)p->lo;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

__global__ void fill(Slot *t, uint64_t cap, uint32_t n_keys, bool partitioned) {
    uint64_t key = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (key >= n_keys) return;
    uint64_t s = mix(key * 0x9E3779B97F4A7C15ull + 1) & (cap - 1);
    const uint64_t part = cap >> 3;
    if (!partitioned) { t[s].lo = key; return; }      // the slot modes 0/1/3/5/6/8 map this key to
    for (int x = 0; x < 8; x++) t[(s & (part - 1)) + x * part].lo = key;   // modes 2/4/7: one copy per XCD partition
}

template <int MODE>
void run(const char *name, Slot *t, uint64_t cap, uint64_t n_ops, uint32_t n_keys, uint32_t *sink) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE><<<256 * 8, 256>>>(t, cap - 1, n_ops, n_keys, sink);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE><<<256 * 8, 256>>>(t, cap - 1, n_ops, n_keys, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s cap=%6lluK slots keys=%5.2fM  %7.3f ms  %6.2f G ops/s\n", name, (unsigned long long)(cap >> 10), n_keys / 1e6, ms, n_ops / ms / 1e6);
}

int main() {
    const uint64_t n_ops = 34400000;
    uint32_t *sink;
    (void)hipMalloc(&sink, 4);
    const uint64_t caps[4] = {32ull << 10, 256ull << 10, 4ull << 20, 32ull << 20};
    const uint32_t keys[4] = {16000u, 128000u, 1080000u, 8600000u};
    for (int cfg = 0; cfg < 4; cfg++) {
        const uint64_t cap = caps[cfg];
        const uint32_t n_keys = keys[cfg];
        Slot *t;
        (void)hipMalloc(&t, cap * sizeof(Slot));
        (void)hipMemset(t, 0, cap * sizeof(Slot));
        fill<<<(n_keys + 255) / 256, 256>>>(t, cap, n_keys, false);
        (void)hipDeviceSynchronize();
        run<0>("device-scope atomicAdd, no return", t, cap, n_ops, n_keys, sink);
        run<8>("device-scope atomicAdd, returning", t, cap, n_ops, n_keys, sink);
        run<1>("workgroup-scope RMW, any address (NOT coherent)", t, cap, n_ops, n_keys, sink);
        run<3>("2 agent-scope loads + device atomicAdd (insert today)", t, cap, n_ops, n_keys, sink);
        run<5>("agent-scope 8 B load", t, cap, n_ops, n_keys, sink);
        run<6>("plain 8 B load", t, cap, n_ops, n_keys, sink);
        (void)hipMemset(t, 0, cap * sizeof(Slot));
        fill<<<(n_keys + 255) / 256, 256>>>(t, cap, n_keys, true);
        (void)hipDeviceSynchronize();
        run<2>("workgroup-scope RMW, XCD-local partition", t, cap, n_ops, n_keys, sink);
        run<4>("plain 16 B load + workgroup RMW, XCD-local partition", t, cap, n_ops, n_keys, sink);
        run<7>("plain 8 B load, XCD-local partition", t, cap, n_ops, n_keys, sink);
        (void)hipFree(t);
    }
    return 0;
}
