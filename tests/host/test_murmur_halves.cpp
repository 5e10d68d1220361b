// The device header csrc/murmur.hpp compiled for the host: the closed form of the minimizer hash against the oracle's
// (the reference's) MurmurHash3_x64_128, and the carry-less upper half the block kernel takes its candidates by:
//     hi(kmer_hash32(v)) - kmer_hash32_hi_nocarry(v)  is 0 or 1 (mod 2^32)
// so that  kmer_hash32(v) < T  implies  kmer_hash32_hi_nocarry(v) + 1 < hi(T) + 2, and the form the kernel runs since round 6, the
// finalisers' last multiplications merged into one:
//     kmer_hash32_hi_merged(v) - hi(kmer_hash32(v))  is 0, 1 or 2 (mod 2^32)
// so that  kmer_hash32(v) < T  implies  kmer_hash32_hi_merged(v) < hi(T) + 3  (scan.hip, span_step<APPROX>).
#define __host__
#define __device__
#define __forceinline__ inline
#include "../../metamdbg_amd/csrc/murmur.hpp"
#include <cstdio>
#include <cstdlib>
#include <random>

extern "C" uint64_t orc_kmer_hash(uint64_t v);      // oracle/mdbg_oracle.c

int main(int argc, char **argv) {
    const long n = argc > 1 ? atol(argv[1]) : 20000000;
    std::mt19937_64 g(12345);
    const uint64_t T = 92233718306963448ull;       // density 0.005f
    const uint32_t limit = (uint32_t)(T >> 32) + 2u;
    const uint32_t limit_m = (uint32_t)(T >> 32) + 3u;
    unsigned long carries = 0, selected = 0, candidates = 0, candidates_m = 0, off_m[3] = {0, 0, 0};
    for (long i = 0; i < n; i++) {
        uint32_t v = i < 70000 ? (uint32_t)i : (uint32_t)g();
        if (i >= 70000 && i < 140000) v = 0xFFFFFFFFu - (uint32_t)(i - 70000);
        const uint64_t h = mdbg::kmer_hash32(v);
        if (i % 997 == 0 && h != orc_kmer_hash(v)) { printf("closed form differs from the oracle at %u\n", v); return 1; }
        const uint32_t u = mdbg::kmer_hash32_hi_nocarry(v);
        const uint32_t d = (uint32_t)(h >> 32) - u;
        if (d > 1u) { printf("upper half off by %u at %u\n", d, v); return 1; }
        carries += d;
        const bool sel = h < T, cand = (uint32_t)(u + 1u) < limit;
        if (sel && !cand) { printf("selected but not a candidate: %u\n", v); return 1; }
        selected += sel; candidates += cand;
        const uint32_t m = mdbg::kmer_hash32_hi_merged(v);
        const uint32_t dm = m - (uint32_t)(h >> 32);
        if (dm > 2u) { printf("merged upper half off by %u at %u\n", dm, v); return 1; }
        off_m[dm]++;
        {   // the two-at-a-time form the unrolled block runs is the same function
            uint32_t ra, rb;
            mdbg::kmer_hash32_hi_merged_x2(v, v ^ 0x5A5A5A5Au, ra, rb);
            if (ra != m || rb != mdbg::kmer_hash32_hi_merged(v ^ 0x5A5A5A5Au)) { printf("the paired form differs at %u\n", v); return 1; }
        }
        const bool cand_m = m < limit_m;
        if (sel && !cand_m) { printf("selected but not a candidate of the merged form: %u\n", v); return 1; }
        candidates_m += cand_m;
    }
    if (!off_m[0] || !off_m[1] || !off_m[2]) { printf("the merged form never met one of its three cases\n"); return 1; }
    printf("ok: %ld values, %lu carries, %lu selected, %lu candidates, %lu candidates of the merged form (off by 0/1/2: %lu/%lu/%lu)\n", n, carries, selected,
           candidates, candidates_m, off_m[0], off_m[1], off_m[2]);
    return 0;
}
