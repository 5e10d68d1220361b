// op_rates.hip -- issue cost of the gfx950 instructions the scan kernel is (or could be) built from: cycles per wave64
// instruction per SIMD with 8 waves per SIMD and four independent chains per wave (throughput, not latency).
// Round 1 measured ten opcodes (valu_rates.hip): v_add/v_xor at 2.4 cycles, every multiply and v_alignbit at 4.2.  Which
// of the OTHER opcodes run at the fast rate decides how the non-hash part of the kernel should be written.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/op_rates.hip -o /tmp/op_rates && /tmp/op_rates
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define REP16(x) x x x x x x x x x x x x x x x x
#define ITERS 2048
#define C4(F) F(0) F(1) F(2) F(3)

// 32-bit chains %0..%3 (a0..a3), operands %4 (b), %5 (c); 64-bit chains use w0..w3 the same way with %4 = b, %5 = w-sized d
#define X32(ID, NAME, F)                                                                                                  \
    if (OP == ID) { REP16(asm volatile(C4(F) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b), "v"(c) : "vcc", "scc", "s20", "s21");) }
#define X64(ID, NAME, F)                                                                                                  \
    if (OP == ID) { REP16(asm volatile(C4(F) : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(b), "v"(d64) : "vcc", "scc", "s20", "s21");) }
#define XLDS(ID, NAME, F)                                                                                                 \
    if (OP == ID) { REP16(asm volatile(C4(F) "s_waitcnt lgkmcnt(0)\n" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(laddr), "v"(c) : "memory");) }

#define F_AND(n) "v_and_b32 %" #n ", %" #n ", %4\n"
#define F_OR(n) "v_or_b32 %" #n ", %" #n ", %4\n"
#define F_XOR(n) "v_xor_b32 %" #n ", %" #n ", %4\n"
#define F_ADD(n) "v_add_u32 %" #n ", %" #n ", %4\n"
#define F_SUB(n) "v_sub_u32 %" #n ", %" #n ", %4\n"
#define F_ADDLIT(n) "v_add_u32 %" #n ", 0x12345, %" #n "\n"
#define F_ADDE64(n) "v_add_u32_e64 %" #n ", %" #n ", %4\n"
#define F_XORE64(n) "v_xor_b32_e64 %" #n ", %" #n ", %4\n"
#define F_SHL(n) "v_lshlrev_b32 %" #n ", 7, %" #n "\n"
#define F_SHR(n) "v_lshrrev_b32 %" #n ", 1, %" #n "\n"
#define F_SHLV(n) "v_lshlrev_b32 %" #n ", %4, %" #n "\n"
#define F_MIN(n) "v_min_u32 %" #n ", %" #n ", %4\n"
#define F_MOV(n) "v_mov_b32 %" #n ", %4\n"
#define F_NOT(n) "v_not_b32 %" #n ", %" #n "\n"
#define F_BFREV(n) "v_bfrev_b32 %" #n ", %" #n "\n"
#define F_FFBL(n) "v_ffbl_b32 %" #n ", %" #n "\n"
#define F_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %4, vcc\n"
#define F_SHL1(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define F_SHR7(n) "v_lshrrev_b32 %" #n ", 7, %" #n "\n"
#define F_ASHR(n) "v_ashrrev_i32 %" #n ", 3, %" #n "\n"
#define F_MAX(n) "v_max_u32 %" #n ", %" #n ", %4\n"
#define F_BITOP3(n) "v_bitop3_b32 %" #n ", %" #n ", %4, %5 bitop3:0x6c\n"
#define F_SUBREV(n) "v_subrev_u32 %" #n ", %" #n ", %4\n"
#define F_ADDCO(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %4\n"
#define F_ADDC(n) "v_addc_co_u32 %" #n ", vcc, %" #n ", %4, vcc\n"
#define F_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 2, %4\n"
#define F_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %4, %5\n"
#define F_OR3(n) "v_or3_b32 %" #n ", %" #n ", %4, %5\n"
#define F_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %4, %5\n"
#define F_XAD(n) "v_xad_u32 %" #n ", %" #n ", %4, %5\n"
#define F_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 2, %4\n"
#define F_ADDLSHL(n) "v_add_lshl_u32 %" #n ", %" #n ", %4, 2\n"
#define F_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 3, 15\n"
#define F_BFI(n) "v_bfi_b32 %" #n ", %4, %" #n ", %5\n"
#define F_PERM(n) "v_perm_b32 %" #n ", %" #n ", %4, %5\n"
#define F_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %4, 7\n"
#define F_ALIGNBYTE(n) "v_alignbyte_b32 %" #n ", %" #n ", %4, 1\n"
#define F_MIN3(n) "v_min3_u32 %" #n ", %" #n ", %4, %5\n"
#define F_BCNT(n) "v_bcnt_u32_b32 %" #n ", %" #n ", %4\n"
#define F_MBCNT(n) "v_mbcnt_lo_u32_b32 %" #n ", %" #n ", %4\n"
#define F_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %4\n"
#define F_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %4\n"
#define F_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %4, %5\n"
#define F_CMP32(n) "v_cmp_lt_u32_e32 vcc, %" #n ", %4\n"
#define F_CMP32S(n) "v_cmp_lt_u32_e64 s[20:21], %" #n ", %4\n"
#define F_MOVDPP(n) "v_mov_b32_dpp %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_ADDDPP(n) "v_add_u32_dpp %" #n ", %" #n ", %4 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define F_MOVDPPB(n) "v_mov_b32_dpp %" #n ", %" #n " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define F_ANDSDWA(n) "v_and_b32_sdwa %" #n ", %" #n ", %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n"
#define F_PKADD(n) "v_pk_add_u16 %" #n ", %" #n ", %4\n"
#define F_PKMUL(n) "v_pk_mul_lo_u16 %" #n ", %" #n ", %4\n"
#define F_PKSHL(n) "v_pk_lshlrev_b16 %" #n ", %4, %" #n "\n"
#define F_PKMAD(n) "v_pk_mad_u16 %" #n ", %" #n ", %4, %5\n"
#define F_READFL(n) "v_readfirstlane_b32 s20, %" #n "\n"
#define F_READLN(n) "v_readlane_b32 s20, %" #n ", 17\n"
#define F_MUL_SALU(n) "v_mul_lo_u32 %" #n ", %" #n ", %4\n s_add_u32 s20, s20, 1\n s_lshl_b32 s21, s20, 2\n"
#define F_XOR_SALU(n) "v_xor_b32 %" #n ", %" #n ", %4\n s_add_u32 s20, s20, 1\n"
// 64-bit chains
#define F_SHR64(n) "v_lshrrev_b64 %" #n ", 33, %" #n "\n"
#define F_SHL64(n) "v_lshlrev_b64 %" #n ", 31, %" #n "\n"
#define F_MAD64(n) "v_mad_u64_u32 %" #n ", vcc, %4, %4, %" #n "\n"
#define F_LSHLADD64(n) "v_lshl_add_u64 %" #n ", %" #n ", 0, %5\n"
#define F_CMP64(n) "v_cmp_lt_u64_e32 vcc, %" #n ", %5\n"
// LDS (address %4 = per-lane, conflict-free)
#define F_DSR32(n) "ds_read_b32 %" #n ", %4\n"
#define F_DSR64(n) "ds_read_b64 %" #n ", %4\n"
#define F_DSRU16(n) "ds_read_u16 %" #n ", %4\n"
#define F_DSW32(n) "ds_write_b32 %4, %" #n "\n"
#define F_DSOR32(n) "ds_or_b32 %4, %" #n "\n"
#define F_DSBPERM(n) "ds_bpermute_b32 %" #n ", %4, %" #n "\n"
#define F_DSSWZ(n) "ds_swizzle_b32 %" #n ", %" #n " offset:0x041F\n"

#define OPS32(X)                                                                                                          \
    X(0, "v_xor_b32", F_XOR) X(1, "v_and_b32", F_AND) X(2, "v_or_b32", F_OR) X(3, "v_add_u32", F_ADD) X(4, "v_sub_u32", F_SUB)      \
    X(5, "v_add_u32 literal", F_ADDLIT) X(6, "v_add_u32_e64", F_ADDE64) X(7, "v_xor_b32_e64", F_XORE64)                             \
    X(8, "v_lshlrev_b32 const", F_SHL) X(9, "v_lshrrev_b32 const", F_SHR) X(10, "v_lshlrev_b32 vgpr", F_SHLV)                        \
    X(11, "v_min_u32", F_MIN) X(12, "v_mov_b32", F_MOV) X(13, "v_not_b32", F_NOT) X(14, "v_bfrev_b32", F_BFREV)                       \
    X(15, "v_ffbl_b32", F_FFBL) X(16, "v_cndmask_b32", F_CNDMASK) X(17, "v_add_co_u32", F_ADDCO) X(18, "v_addc_co_u32", F_ADDC)       \
    X(19, "v_lshl_or_b32", F_LSHLOR) X(20, "v_and_or_b32", F_ANDOR) X(21, "v_or3_b32", F_OR3) X(22, "v_add3_u32", F_ADD3)             \
    X(23, "v_xad_u32", F_XAD) X(24, "v_lshl_add_u32", F_LSHLADD) X(25, "v_add_lshl_u32", F_ADDLSHL) X(26, "v_bfe_u32", F_BFE)         \
    X(27, "v_bfi_b32", F_BFI) X(28, "v_perm_b32", F_PERM) X(29, "v_alignbit_b32", F_ALIGNBIT) X(30, "v_alignbyte_b32", F_ALIGNBYTE)   \
    X(31, "v_min3_u32", F_MIN3) X(32, "v_bcnt_u32_b32", F_BCNT) X(33, "v_mbcnt_lo_u32_b32", F_MBCNT) X(34, "v_mul_lo_u32", F_MULLO)   \
    X(35, "v_mul_hi_u32", F_MULHI) X(36, "v_mad_u32_u24", F_MAD24) X(37, "v_cmp_lt_u32 vcc", F_CMP32) X(38, "v_cmp_lt_u32 sgpr", F_CMP32S) \
    X(39, "v_mov_b32 dpp row_shr", F_MOVDPP) X(40, "v_add_u32 dpp row_shr", F_ADDDPP) X(41, "v_mov_b32 dpp row_bcast", F_MOVDPPB)    \
    X(42, "v_and_b32 sdwa", F_ANDSDWA) X(43, "v_pk_add_u16", F_PKADD) X(44, "v_pk_mul_lo_u16", F_PKMUL) X(45, "v_pk_lshlrev_b16", F_PKSHL) \
    X(46, "v_pk_mad_u16", F_PKMAD) X(47, "v_readfirstlane_b32", F_READFL) X(48, "v_readlane_b32", F_READLN)                          \
    X(49, "v_mul_lo_u32 + 2 salu", F_MUL_SALU) X(50, "v_xor_b32 + 1 salu", F_XOR_SALU)                                                   \
    X(51, "v_lshlrev_b32 by 1", F_SHL1) X(52, "v_lshrrev_b32 by 7", F_SHR7) X(53, "v_ashrrev_i32", F_ASHR) X(54, "v_max_u32", F_MAX)      \
    X(55, "v_bitop3_b32", F_BITOP3) X(56, "v_subrev_u32", F_SUBREV)
#define OPS64(X)                                                                                                          \
    X(60, "v_lshrrev_b64", F_SHR64) X(61, "v_lshlrev_b64", F_SHL64) X(62, "v_mad_u64_u32", F_MAD64) X(63, "v_lshl_add_u64", F_LSHLADD64) \
    X(64, "v_cmp_lt_u64 vcc", F_CMP64)
#define OPSLDS(X)                                                                                                         \
    X(70, "ds_read_b32", F_DSR32) X(71, "ds_read_b64 (pair dest)", F_DSR64) X(72, "ds_read_u16", F_DSRU16) X(73, "ds_write_b32", F_DSW32) \
    X(74, "ds_or_b32", F_DSOR32) X(75, "ds_bpermute_b32", F_DSBPERM) X(76, "ds_swizzle_b32", F_DSSWZ)

template <int OP>
__global__ __launch_bounds__(256) void k(uint32_t *out, uint32_t seed, unsigned long long *clk) {
    __shared__ uint32_t lds[2048];
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    uint32_t b = seed | 1, c = seed * 77u + 5u;
    uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3, d64 = ((uint64_t)seed << 20) | 12345u;
    uint32_t laddr = threadIdx.x * 4u;
    lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a1;
    __syncthreads();
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    for (int i = 0; i < ITERS; i++) {
        OPS32(X32)
        OPS64(X64)
        if (OP == 71) {   // 64-bit destination pairs
            REP16(asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4\n ds_read_b64 %2, %4\n ds_read_b64 %3, %4\n s_waitcnt lgkmcnt(0)\n"
                               : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(laddr) : "memory");)
        } else {
            OPSLDS(XLDS)
        }
    }
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0 && clk) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (uint32_t)(w0 ^ w1 ^ w2 ^ w3) ^ lds[(threadIdx.x * 7) & 2047];
}

static double g_mhz = 0;

template <int OP>
void run(const char *name, uint32_t *d, unsigned long long *dclk, int waves_per_simd, int per_chain = 1) {
    int blocks = 256 * waves_per_simd;   // 256 CUs x (4 SIMDs = one 256-thread block) x waves per SIMD
    printf("%-26s ", name);              // before the launch: a kernel that never returns is named in the output
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 12345, dclk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, 12345, dclk);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[2];
    (void)hipMemcpy(clk, dclk, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)clk[0] / ((double)clk[1] / 100.0);   // s_memtime ticks per microsecond of the 100 MHz real-time counter
    if (g_mhz == 0) g_mhz = mhz;
    double insts_per_simd = (double)ITERS * 64 * waves_per_simd * per_chain;   // wave-instructions (of the op under test) issued per SIMD
    double ns_per_inst = ms * 1e6 / insts_per_simd;
    printf("waves/SIMD=%d  %7.3f ms  %6.3f ns per wave-instr per SIMD = %5.2f cycles @2.4GHz  (s_memtime/realtime: %.0f MHz)\n",
           waves_per_simd, ms, ns_per_inst, ns_per_inst * 2.4, mhz);
}

#define RUN32(ID, NAME, F) if (only < 0 || only == ID) run<ID>(NAME, d, dclk, w);

// op_rates [id]: one opcode only (tools/ubench/op_rates_all.sh runs every id in its own process under a timeout, so that
// an opcode this part does not execute cannot take the others with it)
int main(int argc, char **argv) {
    setvbuf(stdout, nullptr, _IONBF, 0);
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    uint32_t *d;
    unsigned long long *dclk;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4);
    (void)hipMalloc(&dclk, 16);
    for (int w : {8, 5}) {
        OPS32(RUN32)
        OPS64(RUN32)
        OPSLDS(RUN32)
    }
    return 0;
}
