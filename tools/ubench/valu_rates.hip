// valu_rates.hip -- throughput of the integer instructions the Murmur3 minimizer hash is made of,
// measured on gfx950: cycles per wave64 instruction per SIMD.  Build & run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITERS 4096

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    uint32_t b = seed | 1;
    uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3;
    for (int i = 0; i < ITERS; i++) {
        if (OP == 0) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 1) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 2) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(b), "v"(a0) : "vcc");) }
        if (OP == 3) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 4) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 5) { REP16(asm volatile("v_alignbit_b32 %0, %0, %4, 7\n v_alignbit_b32 %1, %1, %4, 7\n v_alignbit_b32 %2, %2, %4, 7\n v_alignbit_b32 %3, %3, %4, 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 6) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %0\n v_mad_u32_u24 %1, %1, %4, %1\n v_mad_u32_u24 %2, %2, %4, %2\n v_mad_u32_u24 %3, %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 7) { REP16(asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
        if (OP == 8) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4" : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(w0));) }
        if (OP == 9) { REP16(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)(w0 ^ w1 ^ w2 ^ w3);
}

template <int OP>
void run(const char *name, uint32_t *d, int waves_per_simd) {
    int blocks = 256 * waves_per_simd;   // 256 CUs x (4 SIMDs = one 256-thread block) x waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 12345);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, 12345);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts_per_simd = (double)ITERS * 64 * waves_per_simd;   // wave-instructions issued per SIMD
    double ns_per_inst = ms * 1e6 / insts_per_simd;
    printf("%-18s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD  (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
           ns_per_inst, ns_per_inst * 2.4);
}

int main() {
    uint32_t *d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    for (int w : {1, 4, 8}) {
        run<0>("v_mul_lo_u32", d, w);
        run<1>("v_mul_hi_u32", d, w);
        run<2>("v_mad_u64_u32", d, w);
        run<3>("v_mul_u32_u24", d, w);
        run<6>("v_mad_u32_u24", d, w);
        run<7>("v_mul_hi_u32_u24", d, w);
        run<4>("v_add_u32", d, w);
        run<5>("v_alignbit_b32", d, w);
        run<8>("v_lshl_add_u64", d, w);
        run<9>("v_xor_b32", d, w);
    }
    return 0;
}
