// scan.hip -- reads -> minimizers on the device.
//
// One wavefront (64 lanes) owns one read at a time and walks it in tiles of 64 packed words
// (2048 bases).  Per tile:
//   1. every lane holds one u64 word (32 bases); run-start flags for homopolymer compression come
//      from a bit trick on the packed word (x ^ (x<<2 | prev base)), a wave prefix sum of their
//      popcounts gives each lane its offset in the compressed stream        [EncoderRLE, Commons.hpp:4163-4203]
//   2. lanes squeeze their kept bases to contiguous 2-bit fields (a 1024-entry LDS table: 5 bases per
//      look-up) and OR them into a per-wave LDS bit stream (carry of the last K bases from the
//      previous tile in front)
//   3. lanes take 4 adjacent k-mers per trip: one 3-word LDS read + v_alignbit extracts the bases
//      LSB-first (E); revcomp = E ^ 0xAAAA.., forward = digit-reverse(E) once, then one shift-in per
//      position; canonical = min; closed-form 8-byte Murmur3 (seed 42) on explicit 32-bit halves and
//      an integer threshold replace the u64-vs-double compare
//                                                           [KmerModel::iterate + MinimizerParser::parse,
//                                                            utils/kmer/Kmer.hpp:531-611, :1373-1456]
//   4. selected lanes are compacted in position order with ballot + popcount.
// A k-mer is evaluated only once the base after it is known, which drops the last k-mer of the
// read exactly as the reference's loop bound (pos < nK-1) does, for any homopolymer tail.
// Low-complexity reads (computeSequenceComplexity, ReadSelection.hpp:1171-1228) are detected with a
// 2-mer upper bound per 64-position window and confirmed by an exact 3-mer pass only when needed.
//
// Template variants: HAS_N adds a 1-bit "invalid" stream (characters with bit 3 set, Kmer.hpp:462;
// k-mers touching one are never selected, :574-580); HAS_QUAL records, for every selected minimizer,
// the ORIGINAL coordinates [rle[pos], rle[pos+K]) its per-minimizer minimum quality spans
// (ReadSelection.hpp:1047-1142) -- the quality bytes themselves are read by the gather kernel.
#include "common.hpp"
#include "murmur.hpp"
#include "objects.hpp"

#include <fcntl.h>
#include <sys/file.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>

namespace mdbg {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_WAVES = SCAN_BLOCK / 64;
constexpr int TILE_WORDS = 64;                 // u64 words per tile (one per lane)
constexpr int STREAM_WORDS = 136;              // u32: (16 carry + 2048 new bases) / 16 = 129, + slack for the w+1 read
constexpr int ISTREAM_WORDS = 68;              // u32: 1 bit per compressed base (HAS_N)
constexpr uint64_t M5 = 0x5555555555555555ull;
constexpr uint8_t READ_SUSPECT = 0x80;         // internal: 2-mer complexity bound exceeded, exact pass pending

struct ScanArgs {
    const uint64_t *words;
    const uint64_t *word_off;
    const uint32_t *len;
    const uint32_t *invalid;      // per word mask (HAS_N)
    const uint32_t *brk;          // per word mask of character changes the 2-bit codes do not show (HAS_N; may be null)
    const uint8_t *qual;          // phred+33 bytes (inline min-quality in the overflow re-run)
    const uint64_t *qual_off;
    uint32_t n_reads;
    uint32_t K;
    uint64_t threshold;           // hash < threshold  <=>  (double)hash < density * 2^64
    uint32_t trim;                // MinimizerParser::_trimBps: 1 (default) skips the first and last l-mer, 0 keeps them
    uint32_t q_last;              // 1: quality span ends ON the last base of the l-mer (correction scan), 0: at the next run
    const uint32_t *rep;          // sorted repetitive minimizers
    uint32_t n_rep;
    int apply_filters;
    const uint32_t *subset;       // optional list of read indices to process (overflow re-run)
    // outputs (padded per read: slots [cap_off[r], cap_off[r+1]))
    const uint64_t *cap_off;
    uint32_t *out_min;
    uint32_t *out_pos;
    uint8_t *out_dir;
    uint32_t *out_os;             // HAS_QUAL: original start of the minimizer's span
    uint32_t *out_oe;             // HAS_QUAL: original end (exclusive)
    uint8_t *out_mqual;           // HAS_QUAL && inline_minq: min quality written directly
    int inline_minq;
    uint32_t *out_count;          // per read: number selected (may exceed capacity -> overflow)
    uint8_t *out_flags;           // per read: MDBG_READ_* | READ_SUSPECT
    // scan_fast_kernel, bump mode (cursor != null; cap_off unused): a wave takes room for a finished read with one atomic
    // add and writes its rows there; reads that outgrow the LDS stage and suspects of the complexity bound are listed
    unsigned long long *cursor;   // [g]: rows handed out so far in region g of the output arrays (one address takes about 80 M
                                  // returning atomics per second on this part; a million 10 kb reads finish in 11 ms)
    uint32_t n_regions;           // power of two
    uint64_t out_capacity;        // rows one region holds
    uint64_t *out_begin;          // per read: first row
    uint32_t *over_list, *suspect_list;
    uint32_t *list_counters;      // [0] = reads in over_list, [1] = reads in suspect_list
    const uint8_t *skip;          // scan_fast_kernel: reads flagged here carry side-mask bits (N, mixed case) and are left to the general kernel
    uint32_t cand_slack;          // scan_fast_kernel<.., APPROX>: extra width of the candidate test (0 but in tests)
    uint32_t wave_priority;       // scan_fast_kernel: s_setprio level of its waves (0 = leave alone)
};

// squeeze the 2-bit fields of x whose flag bit (bit 2i of d) is set down to the low end
__device__ __forceinline__ uint64_t compress_pairs(uint64_t x, uint64_t d) {
    // zero the dropped fields once, then deposit field i at 2 * (number of kept fields below it)
    uint64_t xm = x & (d | (d << 1));
    uint32_t xl = (uint32_t)xm, xh = (uint32_t)(xm >> 32);
    uint32_t dl = (uint32_t)d, dh = (uint32_t)(d >> 32);
    uint32_t lo = 0, hi = 0;
    unsigned n = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        lo |= ((xl >> (2 * i)) & 3u) << n;
        n += (dl >> (2 * i) & 1u) << 1;
    }
    const unsigned nlo = n;
    n = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        hi |= ((xh >> (2 * i)) & 3u) << n;
        n += (dh >> (2 * i) & 1u) << 1;
    }
    return (uint64_t)lo | ((uint64_t)hi << nlo);
}

// Table-driven homopolymer compaction.  For 4 bases (8 bits) and the base before them, which bases
// start a run depends on those 10 bits only, so a 1024-entry table gives the squeezed bits (low byte)
// and twice the number kept (high byte) in one LDS lookup per 4 bases.
constexpr int HPC_LUT_SIZE = 1024;
__device__ __forceinline__ uint16_t hpc_lut_entry(unsigned idx) {
    unsigned prev = idx & 3u, out = 0, n = 0;
    for (int i = 0; i < 4; i++) {
        unsigned b = (idx >> (2 + 2 * i)) & 3u;
        if (b != prev) { out |= b << n; n += 2; }
        prev = b;
    }
    return (uint16_t)(out | (n << 8));
}

// x squeezed to its run starts given the base `pl` preceding the word; *nbits = 2 * kept
// (the two halves of an entry live in two tables: LDS reads are not what this kernel is short of, and the merge below
// then is one shift-or and one add per entry instead of four operations)
struct HpcLut { const uint8_t *bits, *twice_kept; };
__device__ __forceinline__ uint64_t compress_pairs_lut(HpcLut lut, uint64_t x, uint32_t pl, unsigned *nbits) {
    const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
    const uint32_t i0 = ((xl & 0xFFu) << 2) | pl, i1 = (xl >> 6) & 0x3FFu, i2 = (xl >> 14) & 0x3FFu, i3 = xl >> 22;
    const uint32_t i4 = __builtin_amdgcn_alignbit(xh, xl, 30) & 0x3FFu, i5 = (xh >> 6) & 0x3FFu, i6 = (xh >> 14) & 0x3FFu, i7 = xh >> 22;
    uint32_t lo = lut.bits[i0];
    unsigned n = lut.twice_kept[i0];
    lo |= (uint32_t)lut.bits[i1] << n; n += lut.twice_kept[i1];
    lo |= (uint32_t)lut.bits[i2] << n; n += lut.twice_kept[i2];
    lo |= (uint32_t)lut.bits[i3] << n; n += lut.twice_kept[i3];
    uint32_t hi = lut.bits[i4];
    unsigned m = lut.twice_kept[i4];
    hi |= (uint32_t)lut.bits[i5] << m; m += lut.twice_kept[i5];
    hi |= (uint32_t)lut.bits[i6] << m; m += lut.twice_kept[i6];
    hi |= (uint32_t)lut.bits[i7] << m; m += lut.twice_kept[i7];
    *nbits = n + m;
    return (uint64_t)lo | ((uint64_t)hi << n);
}

// same for 1-bit fields: keep bit i of v where bit 2i of d is set
__device__ __forceinline__ uint32_t compress_bits(uint32_t v, uint64_t d) {
    uint32_t out = 0;
    unsigned n = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) {
        unsigned f = (unsigned)(d >> (2 * i)) & 1u;
        out |= ((v >> i) & f) << n;
        n += f;
    }
    return out;
}

// bit i of v -> bit 2i
__device__ __forceinline__ uint64_t spread_bits(uint32_t v) {
    uint64_t x = v;
    x = (x | (x << 16)) & 0x0000FFFF0000FFFFull;
    x = (x | (x << 8)) & 0x00FF00FF00FF00FFull;
    x = (x | (x << 4)) & 0x0F0F0F0F0F0F0F0Full;
    x = (x | (x << 2)) & 0x3333333333333333ull;
    x = (x | (x << 1)) & 0x5555555555555555ull;
    return x;
}

// K bases starting at stream position j, LSB-first
__device__ __forceinline__ uint32_t stream_extract(const uint32_t *S, unsigned j, uint32_t kmask) {
    unsigned b = 2u * j, w = b >> 5, sh = b & 31u;
    uint32_t lo = S[w], hi = S[w + 1];
    return __builtin_amdgcn_alignbit(hi, lo, sh) & kmask;
}

__device__ __forceinline__ uint32_t istream_extract(const uint32_t *SI, unsigned j, uint32_t mask) {
    unsigned w = j >> 5, sh = j & 31u;
    uint32_t lo = SI[w], hi = SI[w + 1];
    return __builtin_amdgcn_alignbit(hi, lo, sh) & mask;
}

// reverse the order of the K 2-bit digits of e
__device__ __forceinline__ uint32_t digit_reverse(uint32_t e, unsigned K) {
    uint32_t r = __builtin_bitreverse32(e);
    r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
    return r >> (32u - 2u * K);
}

constexpr unsigned REP_FILTER_BITS = 4096;            // 512 bytes of LDS a block
__device__ __forceinline__ uint32_t rep_filter_bit(uint32_t v) { return (v * 0x9E3779B1u) >> 20; }      // 12 bits

__device__ __forceinline__ bool rep_contains(const uint32_t *rep, uint32_t n, uint32_t v) {
    uint32_t lo = 0, hi = n;
    while (lo < hi) {
        uint32_t mid = (lo + hi) >> 1;
        uint32_t x = rep[mid];
        if (x < v) lo = mid + 1; else hi = mid;
    }
    return lo < n && rep[lo] == v;
}

// indicator masks in "spread" form: bit 2i set iff base i of the word equals c
__device__ __forceinline__ void base_eq_masks(uint64_t x, uint64_t eq[4]) {
    uint64_t lo = x & M5, hi = (x >> 1) & M5;
    eq[0] = ~(lo | hi) & M5;
    eq[1] = lo & ~hi;
    eq[2] = hi & ~lo;
    eq[3] = lo & hi;
}

// For the window of 64 positions starting at base 0 of x0 (bases from x0, x1 and the first two of
// x2): sum over all 2-mers (ORDER=2) or 3-mers (ORDER=3) v of count(v)^2.  S = (sum - 64) / 2 is
// sum_v c(c-1)/2, the numerator of the reference's window score (ReadSelection.hpp:1206-1216).
template <int ORDER>
__device__ __forceinline__ uint32_t window_sq_sum(uint64_t x0, uint64_t x1, uint64_t x2) {
    uint64_t a0[4], a1[4], a2[4];
    base_eq_masks(x0, a0);
    base_eq_masks(x1, a1);
    base_eq_masks(x2, a2);
    uint32_t sum = 0;
#pragma unroll
    for (int c0 = 0; c0 < 4; c0++) {
#pragma unroll
        for (int c1 = 0; c1 < 4; c1++) {
            // positions i with base(i)==c0 and base(i+1)==c1, for i in word0 / word1
            uint64_t p0 = a0[c0] & ((a0[c1] >> 2) | (a1[c1] << 62));
            uint64_t p1 = a1[c0] & ((a1[c1] >> 2) | (a2[c1] << 62));
            if (ORDER == 2) {
                uint32_t c = __popcll(p0) + __popcll(p1);
                sum += c * c;
            } else {
#pragma unroll
                for (int c2 = 0; c2 < 4; c2++) {
                    uint64_t t0 = p0 & ((a0[c2] >> 4) | (a1[c2] << 60));
                    uint64_t t1 = p1 & ((a1[c2] >> 4) | (a2[c2] << 60));
                    uint32_t c = __popcll(t0) + __popcll(t1);
                    sum += c * c;
                }
            }
        }
    }
    return sum;
}

// sum over the 16 2-mers of (count among the 32 positions of word x)^2; nb = first base of the next word
__device__ __forceinline__ uint32_t word_pair_sq_sum(uint64_t x, uint32_t nb) {
    uint64_t eq[4], sh[4];
    base_eq_masks(x, eq);
#pragma unroll
    for (int c = 0; c < 4; c++) sh[c] = (eq[c] >> 2) | ((uint64_t)(nb == (uint32_t)c) << 62);   // base(i+1) == c
    uint32_t sum = 0;
#pragma unroll
    for (int c0 = 0; c0 < 4; c0++) {
#pragma unroll
        for (int c1 = 0; c1 < 4; c1++) {
            uint32_t c = (uint32_t)__popcll(eq[c0] & sh[c1]);
            sum += c * c;
        }
    }
    return sum;
}

// The same sum on 32-bit bit planes.  The two halves of the word are interleaved so that every mask is one register:
// base i < 16 sits at bit 2i, base 16 + i at bit 2i + 1 ("slot" order); L / H = low / high bit of the 2-bit code.  The
// successor of a slot is the slot two bits up, except slot 30 (base 15 -> base 16 = slot 1) and slot 31 (base 31 -> the first
// base of the next word).  Built from the instructions that issue at the fast rate on gfx950 (and / or / xor / add / right
// shift / v_bitop3, tools/ubench/op_rates.hip) plus two rotates; then 16 x (and, popcount, multiply-add).
__device__ __forceinline__ uint32_t word_pair_sq_sum32(uint64_t x, uint32_t next_lo32) {
    const uint32_t M = 0x55555555u;
    const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
    const uint32_t th = xh & M;
    const uint32_t L = (xl & M) | (th + th);                       // low code bits in slot order
    const uint32_t H = ((xl >> 1) & M) | (xh & ~M);                // high code bits in slot order
    // successor planes: slot s <- slot s + 2; slot 30 <- slot 1; slot 31 <- base 0 of the next word
    const uint32_t nl = __builtin_amdgcn_alignbit(next_lo32, next_lo32, 1);      // bit 31 = low code bit of the next word's base 0
    const uint32_t nh = __builtin_amdgcn_alignbit(next_lo32, next_lo32, 2);      // bit 31 = its high code bit
    const uint32_t NL = (L >> 2) | (__builtin_amdgcn_alignbit(L, L, 3) & 0x40000000u) | (nl & 0x80000000u);
    const uint32_t NH = (H >> 2) | (__builtin_amdgcn_alignbit(H, H, 3) & 0x40000000u) | (nh & 0x80000000u);
    const uint32_t e[4] = {~(L | H), L & ~H, H & ~L, L & H};       // base == A, C, T, G (codes 0..3)
    const uint32_t n[4] = {~(NL | NH), NL & ~NH, NH & ~NL, NL & NH};
    uint32_t sum = 0;
#pragma unroll
    for (int c0 = 0; c0 < 4; c0++) {
#pragma unroll
        for (int c1 = 0; c1 < 4; c1++) {
            const uint32_t c = (uint32_t)__popc(e[c0] & n[c1]);
            sum = __umul24(c, c) + sum;
        }
    }
    return sum;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// Exact complexity decision for one read (second pass, rare): returns 1 when the reference's
// mean window score is > 5.  sum_w S_w / (61 nW) > 5  <=>  sum_w S_w > 305 nW; the equality case
// is resolved by replaying the reference's sequential double arithmetic.
__device__ int exact_low_complexity(const uint64_t *rw, uint32_t L, unsigned lane) {
    if (L < 66) return 0;                       // no full window -> NaN > 5 is false
    uint32_t nW = (L - 66) / 32 + 1;
    uint64_t total = 0;
    for (uint32_t base = 0; base < nW; base += 64) {
        uint32_t w = base + lane;
        uint64_t s = 0;
        if (w < nW) {
            uint32_t nwords = (L + 31) / 32;
            uint64_t x0 = rw[w], x1 = (w + 1 < nwords) ? rw[w + 1] : 0, x2 = (w + 2 < nwords) ? rw[w + 2] : 0;
            s = (window_sq_sum<3>(x0, x1, x2) - 64u) / 2u;
        }
        total += wave_sum_u64(s);
    }
    uint64_t rhs = 305ull * nW;
    if (total != rhs) return total > rhs;
    // tie: replay sequentially in double (all lanes redundantly; practically never taken)
    double acc = 0;
    uint32_t nwords = (L + 31) / 32;
    for (uint32_t w = 0; w < nW; w++) {
        uint64_t x0 = rw[w], x1 = (w + 1 < nwords) ? rw[w + 1] : 0, x2 = (w + 2 < nwords) ? rw[w + 2] : 0;
        double sc = (double)((window_sq_sum<3>(x0, x1, x2) - 64u) / 2u);
        sc /= 61.0;
        acc += sc;
    }
    return (acc / (double)nW) > 5.0;
}

// one wave per suspect read: exact decision; low-complexity reads lose their minimizers but keep
// their (empty) record (ReadSelection.hpp:890-899)
__global__ __launch_bounds__(256) void complexity_exact_kernel(const uint64_t *words, const uint64_t *word_off, const uint32_t *len,
                                                               const uint32_t *list, uint32_t n_list, uint32_t *count, uint8_t *flags,
                                                               unsigned long long *dropped /* may be null: sum of the counts cleared */) {
    const unsigned lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = (gridDim.x * 256u) >> 6;
    for (uint32_t i = wave; i < n_list; i += n_waves) {
        const uint32_t r = list[i];
        int low = exact_low_complexity(words + word_off[r], len[r], lane);
        if (lane == 0) {
            flags[r] = low ? (uint8_t)MDBG_READ_LOW_COMPLEXITY : (uint8_t)0;
            if (low) { if (dropped && count[r]) atomicAdd(dropped, (unsigned long long)count[r]); count[r] = 0; }
        }
    }
}

// per-wave LDS used by the HAS_QUAL && HPC variant: the original coordinate (rlePositions[hpc index], i.e. the
// start of the run) of every position of the compressed stream -- [carry (cb)] [new bases (C)] like the stream
// itself.  Filled by the lanes while they compact their words (one 4-byte LDS store per kept base), read twice
// per selected minimizer.
constexpr int ORIG_WORDS = 16 + 64 * 32;
struct QualMap {
    uint32_t orig[ORIG_WORDS];
};

#ifndef SCAN_PP
#define SCAN_PP 4          // adjacent k-mer positions per lane per trip of the hash loop (1..4)
#endif

// One trip of the hash loop: PP adjacent stream positions per lane, j = j0 + PP * lane + u.
template <bool HPC, bool HAS_QUAL, bool HAS_N>
struct KmerTrip {
    const ScanArgs &a;
    const uint32_t *S, *SI;
    const QualMap *Q;
    unsigned K;
    uint32_t kmask, kbits, comp_mask;
    unsigned lane, nk;
    uint32_t hp_base;
    unsigned cb;
    uint32_t tile_base;
    uint64_t cap0;
    uint32_t cap, r;

    // INTERIOR: every position of the trip is a real k-mer followed by a known base and none is the read's trimmed
    // first position -- the per-position range tests fall away (all full trips but the very first of a read).
    template <int PP, bool INTERIOR>
    __device__ __forceinline__ uint32_t run(unsigned j0, uint32_t nout) const {
        const unsigned jb = j0 + (unsigned)PP * lane;
        // 64 stream bits from position jb (enough for K + PP - 1 <= 19 bases)
        const unsigned b = 2u * jb, w = b >> 5, sh = b & 31u;
        uint32_t a0, a1;
        if (INTERIOR || w + 2 < STREAM_WORDS) {   // interior trips end K + 4 bases before the end of the stream: always in range
            const uint32_t w0 = S[w], w1 = S[w + 1], w2 = S[w + 2];
            a0 = __builtin_amdgcn_alignbit(w1, w0, sh);
            a1 = __builtin_amdgcn_alignbit(w2, w1, sh);
        } else { a0 = 0; a1 = 0; }   // lanes far beyond nk in a short last trip
        bool sel[PP];
        uint32_t val[PP];
        uint32_t fwd = 0;
#pragma unroll
        for (int u = 0; u < PP; u++) {
            const unsigned j = jb + u;
            const uint32_t e = (u == 0 ? a0 : __builtin_amdgcn_alignbit(a1, a0, 2 * u)) & kmask;
            const uint32_t rev = e ^ comp_mask;
            // forward k-mer: full digit reversal once, then shift in the newest base (top digit of e)
            fwd = u == 0 ? digit_reverse(e, K) : (((fwd << 2) | (e >> (2u * K - 2u))) & kmask);
            val[u] = fwd < rev ? fwd : rev;             // canonical; the direction is worked out for the selected few
#if defined(SCAN_ABLATE) && SCAN_ABLATE == 1
            sel[u] = (val[u] == 0x12345u) && (j < nk);                                  // ablation: no hash
#else
            // first k-mer of the read skipped (Kmer.hpp:1395)
            sel[u] = kmer_hash32(val[u]) < a.threshold;
            if (!INTERIOR) sel[u] = sel[u] && (hp_base + j >= a.trim) && (j < nk);
#endif
            if (HAS_N) sel[u] = sel[u] && (istream_extract(SI, j < nk ? j : nk - 1u, kbits) == 0u);   // Kmer.hpp:574-580
        }
        unsigned long long bal[PP];
        unsigned long long any = 0;
#pragma unroll
        for (int u = 0; u < PP; u++) { bal[u] = __ballot(sel[u]); any |= bal[u]; }
        if (any) {
            if (a.n_rep) {   // Kmer.hpp:1437
#pragma unroll
                for (int u = 0; u < PP; u++) {
                    if (sel[u]) sel[u] = !rep_contains(a.rep, a.n_rep, val[u]);
                    bal[u] = __ballot(sel[u]);
                }
            }
            // output order = position order = (lane, u): everything selected by lower lanes comes first
            const unsigned long long lt = lanemask_lt();
            uint32_t idx = nout, total = 0;
#pragma unroll
            for (int u = 0; u < PP; u++) { idx += (uint32_t)__popcll(bal[u] & lt); total += (uint32_t)__popcll(bal[u]); }
#pragma unroll
            for (int u = 0; u < PP; u++) {
                if (sel[u]) {
                    const unsigned j = jb + u;
                    const uint32_t p = hp_base + j;
                    if (idx < cap) {
                        // direction 1 iff the reverse complement is the canonical form, ties included (Kmer.hpp:427)
                        const uint32_t e = (u == 0 ? a0 : __builtin_amdgcn_alignbit(a1, a0, 2 * u)) & kmask;
                        a.out_min[cap0 + idx] = val[u];
                        a.out_pos[cap0 + idx] = p;
                        a.out_dir[cap0 + idx] = (uint8_t)(val[u] == (e ^ comp_mask) ? 1u : 0u);
                        if (HAS_QUAL) {
                            uint32_t os, oe;   // [rle[pos], rle[pos + K]) in original coordinates, or [.., rle[pos + K - 1]]
                            if (HPC) { os = Q->orig[j]; oe = Q->orig[j + K - a.q_last] + a.q_last; }
                            else { os = p; oe = p + K; }
                            if (a.inline_minq) {
                                const uint8_t *qq = a.qual + a.qual_off[r];
                                uint8_t mq = 255;   // getMinQuality (ReadSelection.hpp:1302-1320)
                                for (uint32_t bq = os; bq < oe; bq++) { uint8_t q = (uint8_t)(qq[bq] - 33); if (q < mq) mq = q; }
                                a.out_mqual[cap0 + idx] = mq;
                            } else {
                                a.out_os[cap0 + idx] = os;
                                a.out_oe[cap0 + idx] = oe;
                            }
                        }
                    }
                    idx++;
                }
            }
            nout += total;
        }
        return nout;
    }
};
// Waves per SIMD the register allocator must allow for the plain variant (no qualities, no N).  With the wave-uniform
// values in scalar registers it needs 76 VGPRs (6 waves fit); asking for 7 (72 VGPRs, no spill) makes the kernel no
// faster alone (14.0 vs 13.8 ms) and starves the other batch's table kernels when two batches are in flight; 8 spills.
#ifndef SCAN_MIN_WAVES
#define SCAN_MIN_WAVES 5
#endif

template <bool HPC, bool HAS_QUAL, bool HAS_N>
__global__ __launch_bounds__(SCAN_BLOCK, (HAS_QUAL || HAS_N) ? 1 : SCAN_MIN_WAVES) void scan_kernel(ScanArgs a) {
    __shared__ uint32_t lds_stream[SCAN_WAVES][STREAM_WORDS];
    __shared__ uint32_t lds_istream[HAS_N ? SCAN_WAVES : 1][HAS_N ? ISTREAM_WORDS : 1];
    __shared__ QualMap lds_qmap[(HAS_QUAL && HPC) ? SCAN_WAVES : 1];
    __shared__ uint8_t lds_lut_b[(HPC && !HAS_N) ? HPC_LUT_SIZE : 1], lds_lut_n[(HPC && !HAS_N) ? HPC_LUT_SIZE : 1];
    if (HPC && !HAS_N) {
        for (unsigned i = threadIdx.x; i < HPC_LUT_SIZE; i += SCAN_BLOCK) { const uint16_t e = hpc_lut_entry(i); lds_lut_b[i] = (uint8_t)e; lds_lut_n[i] = (uint8_t)(e >> 8); }
        __syncthreads();
    }
    const unsigned lane = threadIdx.x & 63u;
    // everything that depends only on the wave (its read, the read's length, tile counts, stream lengths) is told to the
    // compiler as wave-uniform, so it lives in scalar registers and the loops around the tiles and trips are scalar branches
    const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *S = lds_stream[wv];
    uint32_t *SI = lds_istream[HAS_N ? wv : 0];
    QualMap *Q = &lds_qmap[(HAS_QUAL && HPC) ? wv : 0];
    const unsigned K = a.K;
    const uint32_t kmask = (K >= 16) ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1u);
    const uint32_t kbits = (1u << K) - 1u;
    const uint32_t comp_mask = 0xAAAAAAAAu & kmask;

    // a wave takes a few reads and retires (launch_variant sizes the grid); written as a grid-stride loop
    const uint32_t wave_global = blockIdx.x * (SCAN_BLOCK / 64) + wv;
    const uint32_t n_waves = (gridDim.x * SCAN_BLOCK) >> 6;
    for (uint32_t slot = wave_global; slot < a.n_reads; slot += n_waves) {
        const uint32_t r = a.subset ? a.subset[slot] : slot;

        const uint32_t L = a.len[r];
        const uint64_t w_base = a.word_off[r];
        const uint64_t *rw = a.words + w_base;
        const uint32_t nwords = (L + 31u) / 32u;
        const uint32_t ntiles = (nwords + TILE_WORDS - 1) / TILE_WORDS;
        const uint64_t cap0 = a.cap_off[r];
        const uint32_t cap = (uint32_t)(a.cap_off[r + 1] - cap0);

        uint32_t carry = 0;        // last cb bases of the compressed stream, LSB-first
        uint32_t icarry = 0;       // their invalid bits (HAS_N)
        unsigned cb = 0;
        uint32_t hp_total = 0;     // compressed bases seen so far
        uint32_t nout = 0;
        uint32_t prev_last = 0;    // last base of the previous tile's last word
        uint32_t prev_inv = 0;     // its invalid bit (HAS_N)
        uint64_t cx_bound = 0;     // sum over windows of the 2-mer bound numerator
        uint64_t prev_word = 0;    // last word of the previous tile (complexity bound of lane 0)

        uint64_t x_next = (lane < nwords) ? rw[lane] : 0;

        for (uint32_t t = 0; t < ntiles; t++) {
            const uint64_t x = x_next;
            const uint32_t wi = t * TILE_WORDS + lane;
            {   // prefetch the next tile
                uint32_t nwi = wi + TILE_WORDS;
                x_next = (nwi < nwords) ? rw[nwi] : 0;
            }
            const int rem = (int)L - (int)(wi * 32u);
            const unsigned nvalid = rem <= 0 ? 0u : (rem >= 32 ? 32u : (unsigned)rem);
            const uint64_t vspread = nvalid == 32 ? M5 : (((1ull << (2 * nvalid)) - 1ull) & M5);
            uint32_t iv = 0;
            if (HAS_N) iv = (wi < nwords) ? a.invalid[w_base + wi] : 0u;

            // ---- complexity: upper bound from per-word 2-mer counts --------------------------------
            // window w = words w, w+1 (64 positions).  With a_v / b_v the 2-mer counts of the two words,
            // sum_v (a_v + b_v)^2 <= 2 (Q_w + Q_{w+1}),  Q = sum_v count^2, and 3-mer collisions <= 2-mer
            // collisions, so  S_w <= Q_w + Q_{w+1} - 32.  Each word's Q enters at most two windows.
            // Lane l evaluates the word BEFORE its own (lane 0: the last word of the previous tile): the base
            // after that word is the lane's own first base, so nothing here waits for the prefetched tile.
#if defined(SCAN_ABLATE) && SCAN_ABLATE == 3
            if (false) {                                                                          // ablation: no complexity bound
#else
            if (a.apply_filters) {
#endif
                uint64_t xp = __shfl_up(x, 1, 64);
                if (lane == 0) xp = prev_word;
                prev_word = __shfl(x, 63, 64);
                if (wi >= 1u && (uint64_t)wi * 32u + 1u <= L) {          // word wi-1 is full and has a successor base
                    const uint32_t wq = wi - 1u;
                    const uint32_t nW = L >= 66u ? (L - 66u) / 32u + 1u : 0u;
                    uint32_t mult = (wq < nW ? 1u : 0u) + ((wq >= 1u && wq - 1u < nW) ? 1u : 0u);
                    if (mult) cx_bound += (uint64_t)mult * word_pair_sq_sum(xp, (uint32_t)x & 3u);
                }
            }

            // ---- 1. run starts / compaction ---------------------------------------------------
            const uint32_t tile_base_now = t * TILE_WORDS * 32u;
            uint64_t y, d;
            unsigned c;
            if (HPC) {
                uint32_t pl = (uint32_t)__shfl_up((uint32_t)(x >> 62), 1, 64);
                if (lane == 0) pl = prev_last;
                if (!HAS_N) {
                    // the first base of the read starts a run whatever precedes it: pretend a different base does
                    if (wi == 0) pl = ((uint32_t)x & 3u) ^ 1u;
                    unsigned nbits;
                    y = compress_pairs_lut(HpcLut{lds_lut_b, lds_lut_n}, x, pl, &nbits);
                    c = nbits >> 1;
                    d = 0;
                    if (HAS_QUAL || nvalid < 32) {
                        // run-start flags themselves: needed for the coordinate map, and to cut a partial word
                        // (zero padding after the last base may look like one more run start)
                        uint64_t diff = x ^ ((x << 2) | (uint64_t)pl);
                        d = (diff | (diff >> 1)) & vspread;
                        c = (unsigned)__popcll(d);
                        y &= c >= 32 ? ~0ull : ((1ull << (2 * c)) - 1ull);
                    }
                } else {
                    uint64_t diff = x ^ ((x << 2) | (uint64_t)pl);
                    d = (diff | (diff >> 1)) & M5;
                    // a change of the invalid bit also starts a run (N vs G share code 3)
                    uint32_t pi = (uint32_t)__shfl_up(iv >> 31, 1, 64);
                    if (lane == 0) pi = prev_inv;
                    d |= spread_bits(iv ^ ((iv << 1) | pi));
                    // ... and so does any other change of character ("aA", IUPAC letters sharing a code): HPC compares
                    // characters (Commons.hpp:4177-4178)
                    if (a.brk) d |= spread_bits((wi < nwords) ? a.brk[w_base + wi] : 0u);
                    prev_inv = (uint32_t)__shfl(iv >> 31, 63, 64);
                    if (wi == 0) d |= 1ull;
                    d &= vspread;
                    c = (unsigned)__popcll(d);
                    y = compress_pairs(x, d);
                }
                prev_last = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 62), 63);
            } else {
                d = vspread;
                c = nvalid;
                y = x & (vspread | (vspread << 1));
            }
            const unsigned inc = wave_inclusive_sum(c);
            const unsigned o = inc - c;
            const unsigned C = (unsigned)__builtin_amdgcn_readlane((int)inc, 63);

            // ---- 2. LDS bit stream: [carry (cb bases)] [new C bases] ---------------------------
            S[lane] = 0;
            S[lane + 64] = 0;
            if (lane < STREAM_WORDS - 128) S[lane + 128] = 0;
            if (HAS_N) { SI[lane] = 0; if (lane < ISTREAM_WORDS - 64) SI[lane + 64] = 0; }
            if (HAS_QUAL && HPC) {
                // d: bit 2i set iff base i of this lane's word starts a run; the r-th set bit is stream position cb + o + r
                uint32_t *dst = Q->orig + cb + o;
                const uint32_t w0 = tile_base_now + lane * 32u;
#pragma unroll
                for (int half = 0; half < 2; half++) {          // 32-bit halves: cheaper bit scans than on the u64
                    uint32_t m = half ? (uint32_t)(d >> 32) : (uint32_t)d;
                    const uint32_t wb = w0 + 16u * half;
                    while (m) {
                        *dst++ = wb + (((uint32_t)__ffs((int)m) - 1u) >> 1);
                        m &= m - 1u;
                    }
                }
            }
            wave_lds_sync();
            if (lane == 0 && cb) { atomicOr(&S[0], carry); if (HAS_N && icarry) atomicOr(&SI[0], icarry); }
            if (c) {
                unsigned dst = 2u * (cb + o), w = dst >> 5, sh = dst & 31u;
                uint32_t yl = (uint32_t)y, yh = (uint32_t)(y >> 32);
                uint32_t p0 = yl << sh;
                uint32_t p1 = sh ? ((yl >> (32u - sh)) | (yh << sh)) : yh;
                uint32_t p2 = sh ? (yh >> (32u - sh)) : 0u;
                atomicOr(&S[w], p0);
                if (p1) atomicOr(&S[w + 1], p1);
                if (p2) atomicOr(&S[w + 2], p2);
                if (HAS_N) {
                    uint32_t ic = HPC ? compress_bits(iv, d) : (iv & (nvalid == 32 ? 0xFFFFFFFFu : ((1u << nvalid) - 1u)));
                    if (ic) {
                        unsigned di = cb + o, wq = di >> 5, sq = di & 31u;
                        atomicOr(&SI[wq], ic << sq);
                        if (sq && (ic >> (32u - sq))) atomicOr(&SI[wq + 1], ic >> (32u - sq));
                    }
                }
            }
            wave_lds_sync();

            // ---- 3./4. k-mers, hash, select, compact -------------------------------------------
            const unsigned tot = cb + C;
            const unsigned nk = tot > K ? tot - K : 0u;
            const uint32_t hp_base = hp_total - cb;    // compressed-stream position of S base 0
            const uint32_t tile_base = t * TILE_WORDS * 32u;
            // Each lane takes PP ADJACENT positions per trip: one LDS read and one digit reversal serve all of
            // them (the forward k-mer of the next position is a shift-in of one base), and the PP hash chains
            // interleave.  The last trip of a tile uses the smallest PP that covers what is left.
            for (unsigned j0 = 0; j0 < nk; j0 += 64 * SCAN_PP) {
                const unsigned left = nk - j0;
                const unsigned pp = left >= 64u * SCAN_PP ? (unsigned)SCAN_PP : (left + 63u) / 64u;
                KmerTrip<HPC, HAS_QUAL, HAS_N> trip{a, S, SI, Q, K, kmask, kbits, comp_mask, lane, nk, hp_base, cb, tile_base, cap0, cap, r};
                // full trips are interior unless this is the trimmed start of the read
                if (pp == (unsigned)SCAN_PP && left >= 64u * SCAN_PP && hp_base + j0 >= a.trim) {
                    nout = trip.template run<SCAN_PP, true>(j0, nout);
                } else {
#if SCAN_PP <= 4
                    switch (pp) {
                        case 1: nout = trip.template run<1, false>(j0, nout); break;
                        case 2: nout = trip.template run<2, false>(j0, nout); break;
                        case 3: nout = trip.template run<3, false>(j0, nout); break;
                        default: nout = trip.template run<4, false>(j0, nout); break;
                    }
#else
                    if (pp <= 2) nout = trip.template run<2, false>(j0, nout);
                    else if (pp <= 4) nout = trip.template run<4, false>(j0, nout);
                    else nout = trip.template run<SCAN_PP, false>(j0, nout);
#endif
                }
            }

            // ---- carry the last min(tot, K) bases ----------------------------------------------
            const unsigned cbn = tot < K ? tot : K;
            const uint32_t cmask = (cbn >= 16) ? 0xFFFFFFFFu : ((1u << (2 * cbn)) - 1u);
            carry = cbn ? stream_extract(S, tot - cbn, cmask) : 0u;
            if (HAS_N) icarry = cbn ? istream_extract(SI, tot - cbn, (1u << cbn) - 1u) : 0u;
            uint32_t my_orig = 0;
            if (HAS_QUAL && HPC) { if (lane < cbn) my_orig = Q->orig[tot - cbn + lane]; }
            cb = cbn;
            hp_total += C;
            wave_lds_sync();   // all reads of S / Q done before the next tile rewrites them
            if (HAS_QUAL && HPC) { if (lane < cbn) Q->orig[lane] = my_orig; }
        }

        // ---- _trimBps == 0 (GenerateGfa.hpp:366): the last l-mer, which the tiles never evaluate because
        // no base follows it.  Its K bases are the carry; one lane does the work.
        if (a.trim == 0u && hp_total >= K) {
            if (HAS_QUAL && HPC) wave_lds_sync();   // carry_orig of the last tile
            const uint32_t e = carry & kmask;
            const uint32_t rev = e ^ comp_mask, fwd = digit_reverse(e, K);
            const uint32_t d = fwd < rev ? 0u : 1u, v = d ? rev : fwd;
            bool s1 = kmer_hash32(v) < a.threshold;
            if (HAS_N) s1 = s1 && icarry == 0u;
            if (s1 && a.n_rep) s1 = !rep_contains(a.rep, a.n_rep, v);
            if (s1) {
                if (lane == 0 && nout < cap) {
                    const uint32_t p = hp_total - K;
                    a.out_min[cap0 + nout] = v;
                    a.out_pos[cap0 + nout] = p;
                    a.out_dir[cap0 + nout] = (uint8_t)d;
                    if (HAS_QUAL) {
                        const uint32_t os = HPC ? Q->orig[0] : p;
                        const uint32_t oe = a.q_last ? (HPC ? Q->orig[K - 1] : p + K - 1u) + 1u : L;
                        if (a.inline_minq) {
                            const uint8_t *qq = a.qual + a.qual_off[r];
                            uint8_t mq = 255;
                            for (uint32_t bq = os; bq < oe; bq++) { uint8_t q = (uint8_t)(qq[bq] - 33); if (q < mq) mq = q; }
                            a.out_mqual[cap0 + nout] = mq;
                        } else {
                            a.out_os[cap0 + nout] = os;
                            a.out_oe[cap0 + nout] = oe;
                        }
                    }
                }
                nout++;
            }
        }

        // ---- per-read epilogue -----------------------------------------------------------------
        uint8_t flags = 0;
        if (a.apply_filters && L >= 66) {
            uint32_t nW = (L - 66u) / 32u + 1u;
            uint64_t bound = wave_sum_u64(cx_bound);    // sum_w (Q_w + Q_{w+1}); subtract 32 per window
            // (bound - 32 nW)/(61 nW) >= true mean score; only reads whose bound exceeds ~4.9 need the exact
            // pass, which runs as its own (rare) kernel so its registers do not cap this kernel's occupancy
            if (bound > (300ull + 32ull) * nW) flags |= READ_SUSPECT;
        }
        if (lane == 0) {
            a.out_count[r] = nout;
            a.out_flags[r] = flags;
        }
    }
}

// ================================================================================================================
// scan_fast_kernel<HPC> -- the variant every upper-case ACGT FASTA batch takes (no qualities, no side masks).
//
// Same arithmetic as scan_kernel, reorganised around what the instructions cost on gfx950 (tools/ubench/op_rates.hip,
// profiles/r02a_op_rates_gfx950.txt: and/or/xor/add/sub/right shifts/v_bitop3 issue in 2.6 cycles per wave64, everything
// else -- left shifts, min, compares, v_alignbit, DPP, every multiply -- in 4.4 to 4.8, a scalar instruction in 2.1, an
// LDS read in 9, an LDS write or atomic in 17):
//   * the compressed bases go to a per-wave LDS RING (8192 bases); whenever 2048 positions are complete (each followed
//     by a known base) the wave hashes them as one BLOCK: lane l owns the 32 CONTIGUOUS positions 32 l .. 32 l + 31, reads
//     its three stream words once and walks them in registers with compile-time shifts -- one v_alignbit per position,
//     the forward k-mer rolled by one base, no address arithmetic, no scalar bookkeeping, no branches;
//   * "selected" is recorded with two instructions per position (v_cmp + v_addc: a per-lane bit vector); the few selected
//     positions of a block (about 10 of 2048) are materialised ONCE PER BLOCK, the lanes that own one recomputing its
//     value from the ring -- instead of ballots, prefix counts and exec-masked stores after every 256 positions;
//   * the minimizers of a read are staged in LDS and leave in coalesced rows;
//   * the tail of a read (fewer than 2048 positions) runs the same code on spans of 4 G <= 32 positions per lane, the
//     stream words fetched from the ring for every group of four.
// ================================================================================================================
constexpr int FAST_BLOCK = 256;                       // threads per workgroup (4 independent waves)
constexpr int FAST_WAVES = FAST_BLOCK / 64;
constexpr unsigned RING_BASES = 16384;                // per wave: DEFER_BLOCKS blocks awaiting materialisation + 2 * 2048 + 16 + 1 live bases
constexpr unsigned DEFER_BLOCKS = 3;                  // blocks whose selected positions may wait in the list before they are materialised
constexpr unsigned RING_WORDS = RING_BASES / 16;      // u32
constexpr unsigned RING_WMASK = RING_WORDS - 1;
constexpr unsigned SPAN = 32;                         // positions per lane in a full block
constexpr unsigned BLOCK_POS = 64 * SPAN;             // 2048
constexpr bool DUAL_CHAIN = true;                      // the unrolled block walks two half spans of a lane side by side (aligned_block)
constexpr int STAGE_CAP = 384;                        // minimizers of one read staged per wave before a flush

// One position of a lane's span: canonical k-mer from the 32 stream bits T that start at the position, Murmur3, compare,
// and the verdict shifted into `bits` (the oldest position ends up in the highest of the bits used).
struct SpanState {
    uint32_t fwd;         // forward k-mer of the previous position
    uint32_t bits;
};

// APPROX: the verdict is taken on the upper half of the hash with the finalisers' last multiplications merged into one
// (kmer_hash32_hi_merged, murmur.hpp): s = that is hi(hash), hi(hash) + 1 or hi(hash) + 2, so  hash < threshold  implies
// s < hi(threshold) + 3.  The positions recorded are a superset of the selected ones, off by about one in 2^31; every one of
// them is hashed in full when it is materialised (emit), and a read with a false one is handed to the general kernel.
// (Rounds 2 - 5 took hi(mix a) + hi(mix b) + 1 by two separate multiplications: 9 cycles per position more.)
// K15 (l = 15, what the reference's presets use): the digit that enters the forward k-mer at this position is the top digit of
// the PREVIOUS position's 32 stream bits, so the roll is one v_alignbit and a mask instead of shift, field extract and merge.
template <bool APPROX, bool K15 = false>
__device__ __forceinline__ void span_step(SpanState &st, uint32_t T, bool first, uint32_t kmask, uint32_t comp_mask, unsigned top_shift,
                                          unsigned K, uint64_t threshold, uint32_t cand_limit, uint32_t Tprev = 0u) {
    const uint32_t rev = (T ^ comp_mask) & kmask;                      // complement of every digit, already in reversed order
    // forward k-mer: digits in reading order; rolled: drop the oldest digit, append the newest (the top digit of T's k-mer)
    if (K15) st.fwd = first ? digit_reverse(T & kmask, K) : (__builtin_amdgcn_alignbit(st.fwd, Tprev, 30) & kmask);
    else st.fwd = first ? digit_reverse(T & kmask, K) : (((st.fwd << 2) | ((T >> top_shift) & 3u)) & kmask);
    const uint32_t val = st.fwd < rev ? st.fwd : rev;
    if (APPROX) {
        const uint32_t s1 = kmer_hash32_hi_merged(val);
        asm("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(st.bits) : "v"(s1), "s"(cand_limit) : "vcc");
    } else {
        const uint64_t h = kmer_hash32(val);
        // bits = 2 * bits + (h < threshold): one compare into vcc, one add-with-carry
        asm("v_cmp_lt_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(st.bits) : "v"(h), "s"(threshold) : "vcc");
    }
}

// the 32 stream bits that start at ring position p (any p), from LDS
__device__ __forceinline__ uint32_t ring_window(const uint32_t *S, unsigned p) {
    const unsigned b = 2u * p, w = (b >> 5) & RING_WMASK, sh = b & 31u;
    return __builtin_amdgcn_alignbit(S[(w + 1) & RING_WMASK], S[w], sh);
}

// QUAL (reads with qualities): the minimum quality over every minimizer's ORIGINAL bases (getMinQuality,
// ReadSelection.hpp:1302-1320; ReadCorrection.hpp:2467-2481) is taken when the minimizer is materialised and travels with its
// row.  Without homopolymer compression original and compressed coordinates coincide.  With it, rlePositions[j] -- the original
// start of run j (Commons.hpp:4188-4190) -- is looked up for the few selected positions only, instead of being stored for all
// of them: per tile every lane keeps its offset in the compressed stream (HIST_TILES tiles back, 2 bytes per word), a selected
// lane finds the tile, then the word (binary search over the 64 offsets), re-reads that word and picks the k-th run start.
constexpr int HIST_TILES = 8;
template <bool HPC, bool QUAL, bool APPROX>
__global__ __launch_bounds__(FAST_BLOCK, 5) void scan_fast_kernel(ScanArgs a) {
    __shared__ uint8_t lds_stage_q[QUAL ? FAST_WAVES : 1][QUAL ? STAGE_CAP : 1];
    __shared__ uint16_t lds_hist_o[(QUAL && HPC) ? FAST_WAVES * HIST_TILES * 64 : 1];
    __shared__ uint32_t lds_hist_c[(QUAL && HPC) ? FAST_WAVES : 1][(QUAL && HPC) ? HIST_TILES : 1];
    __shared__ uint32_t lds_ring[FAST_WAVES][RING_WORDS];
    __shared__ uint2 lds_stage[FAST_WAVES][STAGE_CAP];
    __shared__ uint8_t lds_lut_b[HPC ? HPC_LUT_SIZE : 1], lds_lut_n[HPC ? HPC_LUT_SIZE : 1];
    // the repetitive minimizers (ONT: a hundred or so values, sorted, in global memory) behind a filter of REP_FILTER_BITS bits: a
    // candidate whose bit is clear is not among them, and only the others pay the binary search -- eight dependent loads that a
    // lane of every block used to wait for (9 % of the ONT scan: profiles/round6_p_scan_ablation_ont_before_the_repetitive_filter_in_lds.txt, round6_q_* after)
    __shared__ uint32_t lds_rep_filter[REP_FILTER_BITS / 32];
    if (HPC) {
        for (unsigned i = threadIdx.x; i < HPC_LUT_SIZE; i += FAST_BLOCK) { const uint16_t e = hpc_lut_entry(i); lds_lut_b[i] = (uint8_t)e; lds_lut_n[i] = (uint8_t)(e >> 8); }
    }
    for (unsigned i = threadIdx.x; i < FAST_WAVES * RING_WORDS; i += FAST_BLOCK) (&lds_ring[0][0])[i] = 0;
    for (unsigned i = threadIdx.x; i < REP_FILTER_BITS / 32; i += FAST_BLOCK) lds_rep_filter[i] = 0;
    __syncthreads();
    for (unsigned i = threadIdx.x; i < a.n_rep; i += FAST_BLOCK) {
        const uint32_t b = rep_filter_bit(a.rep[i]);
        atomicOr(&lds_rep_filter[b >> 5], 1u << (b & 31u));
    }
    __syncthreads();
    // the kernel is bound by instruction issue: next to the table kernels of other batches (bound by memory latency) its waves
    // go first on their SIMD
    if (a.wave_priority == 1u) __builtin_amdgcn_s_setprio(1);
    else if (a.wave_priority == 2u) __builtin_amdgcn_s_setprio(2);
    else if (a.wave_priority >= 3u) __builtin_amdgcn_s_setprio(3);
    const unsigned lane = threadIdx.x & 63u;
    const unsigned wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint32_t *S = lds_ring[wv];
    uint2 *stage = lds_stage[wv];
    uint8_t *stage_q = lds_stage_q[QUAL ? wv : 0];
    uint16_t *hist_o = lds_hist_o + ((QUAL && HPC) ? wv * HIST_TILES * 64 : 0);      // [tile % HIST_TILES][lane]
    uint32_t *hist_c = lds_hist_c[(QUAL && HPC) ? wv : 0];
    const unsigned K = a.K;
    const uint32_t kmask = (K >= 16) ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1u);
    const uint32_t comp_mask = 0xAAAAAAAAu & kmask;
    const unsigned top_shift = 2u * K - 2u;
    const uint64_t threshold = a.threshold;
    // APPROX (bump mode only: a read with a false candidate needs the host's re-run): see span_step.  cand_slack widens the
    // superset on purpose (tests of the re-run path)
    // (saturating: a threshold near 2^64 -- densities close to 1 -- must not wrap the limit around to "nothing is a candidate")
    const uint64_t cand_limit64 = (threshold >> 32) + 3ull + (uint64_t)a.cand_slack;
    const uint32_t cand_limit = cand_limit64 > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cand_limit64;

    const uint32_t wave_global = blockIdx.x * FAST_WAVES + wv;
    const uint32_t n_waves = gridDim.x * FAST_WAVES;
    for (uint32_t r = wave_global; r < a.n_reads; r += n_waves) {
        if (a.skip && a.skip[r]) {          // a read with an N or a case flip: the host sends it through the general kernel
            if (a.cursor && lane == 0) { a.out_begin[r] = 0; a.out_count[r] = 0; a.out_flags[r] = 0; }
            continue;
        }
        const uint32_t L = a.len[r];
        const uint64_t w_base = a.word_off[r];
        const uint64_t *rw = a.words + w_base;
        const uint32_t nwords = (L + 31u) / 32u;
        const uint32_t ntiles = (nwords + TILE_WORDS - 1) / TILE_WORDS;
        const bool bump = a.cursor != nullptr;
        const uint64_t cap0 = bump ? 0ull : a.cap_off[r];
        const uint32_t cap = bump ? 0u : (uint32_t)(a.cap_off[r + 1] - cap0);
        uint32_t t_done = 0;       // raw tiles appended to the ring so far (HPC && QUAL: their offsets are in hist_*)
        uint32_t fill = 0;         // compressed bases written to the ring so far (= stream length)
        bool outgrown = false;     // the read selected more than the stage holds, or needs a run start the history no longer has:
                                   // listed, placed and re-run by the host with the general kernel

        // rlePositions[j]: the original coordinate of compressed position j (HPC && QUAL); j == fill at the end of the read is the
        // sentinel `length` (Commons.hpp:4188-4190).  0xFFFFFFFF: the tile is no longer in the history.
        auto orig_of = [&](uint32_t j, bool at_end) -> uint32_t {
            if (at_end && j >= fill) return L;
            uint32_t tt = t_done - 1u;
            const uint32_t t_low = t_done > (uint32_t)HIST_TILES ? t_done - (uint32_t)HIST_TILES : 0u;
            while (tt > t_low && hist_c[tt % HIST_TILES] > j) tt--;
            const uint32_t c0 = hist_c[tt % HIST_TILES];
            if (c0 > j) return 0xFFFFFFFFu;
            const uint32_t rel = j - c0;
            const uint16_t *o = hist_o + (tt % HIST_TILES) * 64u;
            uint32_t lo = 0;           // last word whose first run starts at or before rel: three rounds of three reads each
#pragma unroll
            for (uint32_t st = 16; st >= 1; st >>= 2) {
                const uint32_t o1 = o[lo + st], o2 = o[lo + 2 * st], o3 = o[lo + 3 * st];
                lo += st * ((o1 <= rel ? 1u : 0u) + (o2 <= rel ? 1u : 0u) + (o3 <= rel ? 1u : 0u));
            }
            uint32_t k = rel - o[lo];
            const uint32_t wi = tt * TILE_WORDS + lo;
            const uint64_t x = rw[wi];
            const uint32_t pl = wi ? (uint32_t)(rw[wi - 1] >> 62) : (((uint32_t)x & 3u) ^ 1u);   // the first base of the read starts a run
            const int rem = (int)L - (int)(wi * 32u);
            const unsigned nvalid = rem >= 32 ? 32u : (unsigned)rem;
            const uint64_t vspread = nvalid == 32 ? M5 : (((1ull << (2 * nvalid)) - 1ull) & M5);
            const uint64_t diff = x ^ ((x << 2) | (uint64_t)pl);
            uint64_t m = (diff | (diff >> 1)) & vspread;          // bit 2i: base i starts a run
            uint32_t pos = 0;
#pragma unroll
            for (int sh = 32; sh >= 2; sh >>= 1) {                // the k-th set bit (k counted from 0)
                const uint32_t c = (uint32_t)__popcll(m & ((1ull << sh) - 1ull));
                if (k >= c) { k -= c; m >>= sh; pos += (uint32_t)sh; }
            }
            return wi * 32u + (pos >> 1);
        };
        const uint8_t *qq = QUAL ? a.qual + a.qual_off[r] : nullptr;
        // minimum quality of the minimizer at compressed position j; unknown = false when a run start has left the history
        auto min_quality = [&](uint32_t j, bool &known) -> uint8_t {
            uint32_t os = j, oe = j + K;
            if (HPC) {     // [rle[pos], rle[pos + l]) (ReadSelection.hpp:1135) or [rle[pos], rle[pos + l - 1]] (ReadCorrection.hpp:2340)
                os = orig_of(j, false);
                const uint32_t last = orig_of(j + K - a.q_last, true);
                if (os == 0xFFFFFFFFu || last == 0xFFFFFFFFu) { known = false; return 0; }
                oe = last + a.q_last;
            }
            // 16 bytes per step in two unaligned 8-byte loads (the quality buffer is padded by 32 bytes) instead of a chain of
            // dependent byte loads: a selected lane would otherwise wait longer than its whole block took to hash
            uint8_t mq = 255;
            for (uint32_t b0 = os; b0 < oe; b0 += 16) {
                uint64_t w[2];
                __builtin_memcpy(w, qq + b0, 16);
                const uint32_t nb = oe - b0 < 16u ? oe - b0 : 16u;
#pragma unroll
                for (unsigned b = 0; b < 16; b++) {
                    const uint8_t q = (uint8_t)((uint8_t)(w[b >> 3] >> (8 * (b & 7))) - 33);
                    if (b < nb && q < mq) mq = q;
                }
            }
            return mq;
        };

        bool lost = false;         // per lane: a quality span could not be resolved, or a recorded position turned out not to be
                                   // selected (APPROX): folded into `outgrown` at the end of the read
        uint32_t n_false = 0;      // APPROX: recorded positions the full hash rejected (the read's count is nout - n_false)
        uint32_t n_staged_at_outgrowth = 0;     // APPROX: rows in the stage when the read outgrew it
        uint32_t done = 0;         // positions already evaluated (= ring position of the next block, a multiple of 2048)
        uint32_t nout = 0;         // minimizers of this read so far
        uint32_t flushed = 0;      // ... of which already written to the output slot
        uint32_t n_mat = 0;        // ... of which materialised (rows [n_mat, nout) of the stage hold a position only)
        uint32_t zero_from = 0;    // ring position from which bases are still kept (<= done): the oldest block with listed positions
        uint32_t prev_last = 0;
        uint64_t cx_acc = 0;       // per lane: sum of weight x Q over its words
        uint64_t prev_word = 0;
        const uint32_t cx_nW = L >= 66u ? (L - 66u) / 32u + 1u : 0u;      // number of complexity windows (ReadSelection.hpp:1171-1228)

        // ---- materialisation of the listed positions, one per lane: window, canonical form, direction and, with qualities, the
        // look-ups behind min_quality.  Deferred over up to DEFER_BLOCKS blocks (a block lists about ten positions: materialised
        // there, five lanes in six idled through the longest dependent chain of the kernel) ----
        auto materialise = [&]() {
            const uint32_t pend = nout - n_mat;
            if (pend == 0u) return;
            if (outgrown) { n_mat = nout; return; }       // counted only: the read is re-run by the general kernel
            const uint32_t base = n_mat - flushed;
            for (uint32_t i = lane; i < pend; i += 64) {
                const uint32_t j = stage[base + i].y;
                const uint32_t e = ring_window(S, j) & kmask;
                const uint32_t rev = e ^ comp_mask, fw = digit_reverse(e, K);
                // direction 1 iff the reverse complement is the canonical form, ties included (Kmer.hpp:427)
                const uint32_t d = fw < rev ? 0u : 1u;
                if (QUAL) { bool known = true; stage_q[base + i] = min_quality(j, known); if (!known) lost = true; }
                stage[base + i] = make_uint2(d ? rev : fw, (j << 1) | d);
            }
            n_mat = nout;
            wave_lds_sync();
        };
        // bases of [zero_from, upto) are not needed any more: back to zero for the next lap of the ring
        auto release_ring = [&](uint32_t upto) {
            const unsigned zb = zero_from >> 4, nz = (upto - zero_from) >> 4;
            for (unsigned i = lane; i < nz; i += 64) S[(zb + i) & RING_WMASK] = 0;
            zero_from = upto;
        };

        // ---- emission: list the positions recorded in `bits` (span length P per lane, oldest position in bit P-1) ----
        auto emit = [&](uint32_t bits, unsigned P, uint32_t npos_limit) {
            // positions at or beyond npos_limit (tail lanes past the end) and the trimmed first position of the read
            if (P < 32u) bits &= (1u << P) - 1u;
            {
                const uint32_t first_j = lane * P;                 // span-relative index of this lane's first position
                if (first_j >= npos_limit) bits = 0;
                else if (first_j + P > npos_limit) bits &= ~((1u << (first_j + P - npos_limit)) - 1u);
                if (done == 0u && lane == 0u && a.trim) bits &= ~(1u << (P - 1u));      // Kmer.hpp:1395
            }
            if (__ballot(bits != 0u) == 0ull) return;
            if (a.n_rep) {          // Kmer.hpp:1437: repetitive minimizers are not selected
                uint32_t b2 = bits;
                while (b2) {
                    const unsigned bit = 31u - (unsigned)__clz((int)b2);
                    b2 &= ~(1u << bit);
                    const unsigned u = P - 1u - bit;
                    const uint32_t e = ring_window(S, done + lane * P + u) & kmask;
                    const uint32_t rev = e ^ comp_mask, fw = digit_reverse(e, K);
                    const uint32_t v = fw < rev ? fw : rev, fb = rep_filter_bit(v);
                    if (((lds_rep_filter[fb >> 5] >> (fb & 31u)) & 1u) && rep_contains(a.rep, a.n_rep, v)) bits &= ~(1u << bit);
                }
            }
            const unsigned cnt = (unsigned)__popc(bits);
            const unsigned incl = wave_inclusive_sum_dpp(cnt);
            const unsigned total = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
            if (total == 0u) return;
            if (bump) {
                if (outgrown || nout + total > (unsigned)STAGE_CAP) {
                    if (!outgrown) { materialise(); n_staged_at_outgrowth = nout; }
                    if (APPROX) {          // the count the host places the read by must be exact: hash the candidates in full
                        uint32_t bad = 0;
                        while (bits) {
                            const unsigned bit = 31u - (unsigned)__clz((int)bits);
                            bits &= ~(1u << bit);
                            const uint32_t e = ring_window(S, done + lane * P + (P - 1u - bit)) & kmask;
                            const uint32_t rev = e ^ comp_mask, fw = digit_reverse(e, K);
                            if (!(kmer_hash32(fw < rev ? fw : rev) < threshold)) bad++;
                        }
                        n_false += (uint32_t)wave_sum_u64(bad);
                    }
                    outgrown = true; nout += total; return;
                }
            } else if (nout - flushed + total > (unsigned)STAGE_CAP) {       // make room: the staged rows leave for the output slot
                materialise();
                const uint32_t ns = nout - flushed;
                for (uint32_t i = lane; i < ns; i += 64) {
                    const uint32_t idx = flushed + i;
                    if (idx < cap) {
                        const uint2 e = stage[i];
                        a.out_min[cap0 + idx] = e.x; a.out_pos[cap0 + idx] = e.y >> 1; a.out_dir[cap0 + idx] = (uint8_t)(e.y & 1u);
                        if (QUAL) a.out_mqual[cap0 + idx] = stage_q[i];
                    }
                }
                flushed = nout;
                wave_lds_sync();
            }
            if (total <= (unsigned)STAGE_CAP) {
                // the selected positions are listed in the stage; materialise() takes them one per lane later
                uint32_t at = nout - flushed + incl - cnt;
                while (bits) {
                    const unsigned bit = 31u - (unsigned)__clz((int)bits);
                    bits &= ~(1u << bit);
                    stage[at++].y = done + lane * P + (P - 1u - bit);          // position in the compressed read
                }
                wave_lds_sync();
            } else {
                // more selected positions in one block than the stage holds (densities near 1): straight to the slot
                uint32_t at = nout + incl - cnt;
                while (bits) {
                    const unsigned bit = 31u - (unsigned)__clz((int)bits);
                    bits &= ~(1u << bit);
                    const unsigned u = P - 1u - bit;
                    const uint32_t j = done + lane * P + u;
                    const uint32_t e = ring_window(S, j) & kmask;
                    const uint32_t rev = e ^ comp_mask, fw = digit_reverse(e, K);
                    const uint32_t d = fw < rev ? 0u : 1u;
                    if (at < cap) {
                        a.out_min[cap0 + at] = d ? rev : fw; a.out_pos[cap0 + at] = j; a.out_dir[cap0 + at] = (uint8_t)d;
                        if (QUAL) { bool known = true; a.out_mqual[cap0 + at] = min_quality(j, known); if (!known) lost = true; }
                    }
                    at++;
                }
                flushed = nout + total;
                n_mat = nout + total;
            }
            nout += total;
        };

        // ---- 64 * SP positions from `done` on, each followed by a known base: lane l owns positions SP l .. SP l + SP - 1, i.e.
        // 2 SP stream bits that start on a word boundary of the ring; walked in registers with compile-time shifts ----
        auto aligned_block = [&](auto sp_tag) {
            constexpr int SP = decltype(sp_tag)::value;              // 32 or 16
            constexpr unsigned WPL = (unsigned)SP / 16u;               // stream words per lane
            const unsigned wb = ((done >> 4) + WPL * lane) & RING_WMASK;
            const uint32_t W0 = S[wb], W1 = S[(wb + 1) & RING_WMASK], W2 = SP > 16 ? S[(wb + 2) & RING_WMASK] : 0u;
            SpanState st{0u, 0u};
            auto walk = [&](auto k15_tag) {
                constexpr bool K15 = decltype(k15_tag)::value;
                if (SP == 32 && DUAL_CHAIN) {
                    // two independent chains per lane -- positions 0 .. 15 and 16 .. 31 side by side -- so that a wave has another
                    // instruction to issue while a multiply's result is on its way (at 4 - 5 waves per SIMD one chain leaves gaps:
                    // tools/ubench/hash_rates.hip, chains=1 against chains=2)
                    uint32_t fa = 0u, fb = 0u, ba = 0u, bb = 0u, pa = 0u, pb = 0u;
#pragma unroll
                    for (int u = 0; u < 16; u++) {
                        const uint32_t Ta = u == 0 ? W0 : __builtin_amdgcn_alignbit(W1, W0, 2 * u);
                        const uint32_t Tb = u == 0 ? W1 : __builtin_amdgcn_alignbit(W2, W1, 2 * u);
                        const uint32_t reva = (Ta ^ comp_mask) & kmask, revb = (Tb ^ comp_mask) & kmask;
                        if (u == 0) { fa = digit_reverse(Ta & kmask, K); fb = digit_reverse(Tb & kmask, K); }
                        else if (K15) { fa = __builtin_amdgcn_alignbit(fa, pa, 30) & kmask; fb = __builtin_amdgcn_alignbit(fb, pb, 30) & kmask; }
                        else { fa = ((fa << 2) | ((Ta >> top_shift) & 3u)) & kmask; fb = ((fb << 2) | ((Tb >> top_shift) & 3u)) & kmask; }
                        const uint32_t va = fa < reva ? fa : reva, vb = fb < revb ? fb : revb;
                        if (APPROX) {
                            uint32_t ra, rb;
                            kmer_hash32_hi_merged_x2(va, vb, ra, rb);
                            asm("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(ba) : "v"(ra), "s"(cand_limit) : "vcc");
                            asm("v_cmp_lt_u32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bb) : "v"(rb), "s"(cand_limit) : "vcc");
                        } else {
                            const uint64_t ha = kmer_hash32(va), hb = kmer_hash32(vb);
                            asm("v_cmp_lt_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(ba) : "v"(ha), "s"(threshold) : "vcc");
                            asm("v_cmp_lt_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bb) : "v"(hb), "s"(threshold) : "vcc");
                        }
                        pa = Ta; pb = Tb;
                    }
                    st.bits = (ba << 16) | bb;
                    return;
                }
                uint32_t Tprev = 0u;
#pragma unroll
                for (int u = 0; u < SP; u++) {
                    const uint32_t T = u == 0 ? W0 : (u < 16 ? __builtin_amdgcn_alignbit(W1, W0, 2 * u)
                                                             : (u == 16 ? W1 : __builtin_amdgcn_alignbit(W2, W1, 2 * (u - 16))));
                    span_step<APPROX, K15>(st, T, u == 0, kmask, comp_mask, top_shift, K, threshold, cand_limit, Tprev);
                    Tprev = T;
                }
            };
            if (K == 15u) walk(std::true_type()); else walk(std::false_type());
            emit(st.bits, (unsigned)SP, 64u * (unsigned)SP);
            wave_lds_sync();
            done += 64u * (unsigned)SP;
            // materialise when most of a wave's lanes have a position to take, or when the ring cannot keep the blocks any longer
            // (with qualities under HPC the run starts of a listed position must still be in the tile history: one block less)
            constexpr unsigned defer = (QUAL && HPC) ? DEFER_BLOCKS - 1u : DEFER_BLOCKS;
            if (nout - n_mat >= 48u || nout == n_mat || done - zero_from > defer * BLOCK_POS) {
                materialise();
                release_ring(done);
            }
            wave_lds_sync();
        };

        uint64_t x_next = (lane < nwords) ? rw[lane] : 0;
        for (uint32_t t = 0; t < ntiles; t++) {
            const uint64_t x = x_next;
            const uint32_t wi = t * TILE_WORDS + lane;
            {
                const uint32_t nwi = wi + TILE_WORDS;
                x_next = (nwi < nwords) ? rw[nwi] : 0;
            }
            const int rem = (int)L - (int)(wi * 32u);
            const unsigned nvalid = rem <= 0 ? 0u : (rem >= 32 ? 32u : (unsigned)rem);

            // ---- complexity: upper bound from per-word 2-mer counts (as in scan_kernel; the word below comes over DPP) ----
            if (a.apply_filters) {
                const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
                const uint32_t pl32 = lane_below(xl, (uint32_t)prev_word), ph32 = lane_below(xh, (uint32_t)(prev_word >> 32));
                prev_word = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)xh, 63) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)xl, 63);
                // word wq = wi - 1 enters window wq (if wq < nW) and window wq - 1 (if 1 <= wq <= nW): weight 1 at wq = 0 and nW, 2 between
                const uint32_t q = word_pair_sq_sum32(((uint64_t)ph32 << 32) | pl32, xl);
                const uint32_t wq = wi - 1u;                      // lane 0 of tile 0: wraps around, weight 0
                const uint32_t w2 = wq < cx_nW ? q : 0u, w1 = (wq - 1u) < cx_nW ? q : 0u;
                cx_acc += w2 + w1;
            }

            // ---- run starts / compaction of this lane's word ----
            uint64_t y;
            unsigned c;
            if (HPC) {
                uint32_t pl = lane_below((uint32_t)(x >> 62), prev_last);
                if (wi == 0) pl = ((uint32_t)x & 3u) ^ 1u;      // the first base of the read starts a run whatever precedes it
                unsigned nbits;
                y = compress_pairs_lut(HpcLut{lds_lut_b, lds_lut_n}, x, pl, &nbits);
                c = nbits >> 1;
                if (nvalid < 32) {
                    const uint64_t vspread = ((1ull << (2 * nvalid)) - 1ull) & M5;
                    const uint64_t diff = x ^ ((x << 2) | (uint64_t)pl);
                    const uint64_t d = (diff | (diff >> 1)) & vspread;
                    c = (unsigned)__popcll(d);
                    y &= c >= 32 ? ~0ull : ((1ull << (2 * c)) - 1ull);
                }
                prev_last = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 62), 63);
            } else {
                c = nvalid;
                y = nvalid == 32 ? x : (x & ((1ull << (2 * nvalid)) - 1ull));
            }
            const unsigned inc = wave_inclusive_sum_dpp(c);
            const unsigned o = inc - c;
            const unsigned C = (unsigned)__builtin_amdgcn_readlane((int)inc, 63);

            // ---- append to the ring (the region ahead of `fill` is zero) ----
            if (c) {
                const unsigned dst = 2u * ((fill + o) & (RING_BASES - 1u)), w = dst >> 5, sh = dst & 31u;
                const uint32_t yl = (uint32_t)y, yh = (uint32_t)(y >> 32);
                const uint32_t p0 = yl << sh;
                const uint32_t p1 = sh ? ((yl >> (32u - sh)) | (yh << sh)) : yh;
                const uint32_t p2 = sh ? (yh >> (32u - sh)) : 0u;
                atomicOr(&S[w], p0);
                if (p1) atomicOr(&S[(w + 1) & RING_WMASK], p1);
                if (p2) atomicOr(&S[(w + 2) & RING_WMASK], p2);
            }
            if (HPC && QUAL) {         // where this tile's words start in the compressed stream (for rlePositions look-ups)
                hist_o[(t % HIST_TILES) * 64u + lane] = (uint16_t)o;
                if (lane == 0) hist_c[t % HIST_TILES] = fill;
                t_done = t + 1u;
            }
            fill += C;
            wave_lds_sync();

            // ---- full blocks: 2048 positions, every one followed by a known base ----
            // (with _trimBps == 0 the last l-mer is a position too: a block also runs at exactly 2048 + K bases, or the tail would
            // be left with 2049 positions -- 36 per lane, four more than a lane's verdict vector holds)
            while (fill - done >= BLOCK_POS + K + a.trim) aligned_block(std::integral_constant<int, (int)SPAN>());
        }
        // (a last half block of 1024 positions, 16 per lane, before the tail was measured: 11.73 against 11.76 ms -- the tail's
        // positions cost about what a block's do; not kept)

        // ---- tail: the positions left (each followed by a known base; with _trimBps == 0 also the last l-mer) ----
        {
            const uint32_t live = fill - done;
            uint32_t npos = live > K ? live - K : 0u;
            if (a.trim == 0u && live >= K) npos = live - K + 1u;
            if (npos) {
                // groups of four positions per lane: at most 8, the verdicts of a lane are a 32-bit vector (the block loop above
                // leaves at most 2048 positions whatever _trimBps is)
                if (npos > BLOCK_POS) __builtin_trap();
                const unsigned G = (npos + 255u) / 256u;
                const unsigned P = 4u * G;
                SpanState st{0u, 0u};
                auto tail_walk = [&](auto k15_tag) {
                    constexpr bool K15 = decltype(k15_tag)::value;
                    uint32_t Tprev = 0u;               // a lane's positions are consecutive across its groups: the roll of l = 15 carries over
                    for (unsigned g = 0; g < G; g++) {
                        const unsigned p = done + lane * P + 4u * g;     // ring position of the group's first position
                        const unsigned b = 2u * p, w = (b >> 5) & RING_WMASK, sh = b & 31u;
                        const uint32_t w0 = S[w], w1 = S[(w + 1) & RING_WMASK], w2 = S[(w + 2) & RING_WMASK];
                        const uint32_t a0 = __builtin_amdgcn_alignbit(w1, w0, sh), a1 = __builtin_amdgcn_alignbit(w2, w1, sh);
#pragma unroll
                        for (int u = 0; u < 4; u++) {
                            const uint32_t T = u == 0 ? a0 : __builtin_amdgcn_alignbit(a1, a0, 2 * u);
                            span_step<APPROX, K15>(st, T, g == 0 && u == 0, kmask, comp_mask, top_shift, K, threshold, cand_limit, Tprev);
                            Tprev = T;
                        }
                    }
                };
                if (K == 15u) tail_walk(std::true_type()); else tail_walk(std::false_type());
                emit(st.bits, P, npos);
            }
            wave_lds_sync();
            materialise();
            // leave the ring zero for the next read: everything from `zero_from` to `fill`
            release_ring(((fill + 15u) & ~15u) + 16u);
        }

        if (APPROX) {
            // the full hash of every staged position, once per read (a block stages about ten of them: confirming them there kept
            // five lanes in six idle); a false one loses the read to the re-run.  The positions of a read that outgrew the stage
            // after these were confirmed as they came (emit)
            const uint32_t n_staged = outgrown ? n_staged_at_outgrowth : nout;
            for (uint32_t i = lane; i < ((n_staged + 63u) & ~63u); i += 64) {
                const uint64_t fb = __ballot(i < n_staged && !(kmer_hash32(stage[i].x) < threshold));
                if (fb) { n_false += (uint32_t)__popcll(fb); lost = true; }
            }
        }
        if (((QUAL && HPC) || APPROX) && __ballot(lost) != 0ull) outgrown = true;      // a run start had left the history, or a false candidate:
                                                                                        // the general kernel redoes the read
        uint8_t flags = 0;
        if (a.apply_filters && L >= 66) {
            const uint64_t bound = wave_sum_u64(cx_acc);
            if (bound > (300ull + 32ull) * cx_nW) flags |= READ_SUSPECT;
        }
        // ---- the staged minimizers leave in rows ----
        if (bump) {
            uint64_t start = 0;
            if (!outgrown && nout) {
                uint32_t lo = 0, hi = 0;
                const uint32_t region = wave_global & (a.n_regions - 1u);
                if (lane == 0) { const unsigned long long s0 = atomicAdd(a.cursor + region, (unsigned long long)nout); lo = (uint32_t)s0; hi = (uint32_t)(s0 >> 32); }
                start = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)hi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)lo);
                const bool fits = start + nout <= a.out_capacity;
                start += (uint64_t)region * a.out_capacity;
                if (fits) {
                    for (uint32_t i = lane; i < nout; i += 64) {
                        const uint2 e = stage[i];
                        a.out_min[start + i] = e.x; a.out_pos[start + i] = e.y >> 1; a.out_dir[start + i] = (uint8_t)(e.y & 1u);
                        if (QUAL) a.out_mqual[start + i] = stage_q[i];
                    }
                }
            }
            if (lane == 0) {
                a.out_begin[r] = start;
                a.out_count[r] = nout - n_false;    // also for a read that outgrew the stage: the host places and re-runs it
                a.out_flags[r] = flags;
                if (outgrown) a.over_list[atomicAdd(&a.list_counters[0], 1u)] = r;
                if (flags & READ_SUSPECT) a.suspect_list[atomicAdd(&a.list_counters[1], 1u)] = r;
            }
        } else {
            const uint32_t ns = nout - flushed;
            for (uint32_t i = lane; i < ns; i += 64) {
                const uint32_t idx = flushed + i;
                if (idx < cap) {
                    const uint2 e = stage[i];
                    a.out_min[cap0 + idx] = e.x; a.out_pos[cap0 + idx] = e.y >> 1; a.out_dir[cap0 + idx] = (uint8_t)(e.y & 1u);
                    if (QUAL) a.out_mqual[cap0 + idx] = stage_q[i];
                }
            }
            if (lane == 0) {
                a.out_count[r] = nout;
                a.out_flags[r] = flags;
            }
        }
        wave_lds_sync();       // ring zeroed, stage drained: the next read starts clean
    }
}

// ---- padded -> dense CSR (+ per-minimizer minimum quality) -------------------------------------
// G lanes per read: 16 when reads hold a few dozen minimizers (4 reads in flight per wave: the kernel is a chain of
// dependent loads per read, so reads in flight are what it runs on), 64 for long reads / high densities.
template <bool HAS_QUAL, int G>
__global__ __launch_bounds__(256) void compact_minimizers_kernel(
    const uint64_t *cap_off, const uint64_t *off, uint32_t n_reads,
    const uint32_t *pmin, const uint32_t *ppos, const uint8_t *pdir, const uint32_t *pos_s, const uint32_t *pos_e,
    const uint8_t *qual, const uint64_t *qual_off,
    uint32_t *omin, uint32_t *opos, uint8_t *odir, uint8_t *oqual) {
    const unsigned lane = threadIdx.x & (unsigned)(G - 1);
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) / G;
    for (uint64_t r = wave; r < n_reads; r += nwaves) {
        uint64_t src = cap_off[r], dst = off[r];
        uint32_t n = (uint32_t)(off[r + 1] - dst);
        uint32_t ncopy = (uint32_t)(cap_off[r + 1] - src);   // padded capacity (overflowed reads are re-run)
        if (ncopy > n) ncopy = n;
        for (uint32_t i = lane; i < n; i += G) {
            uint8_t mq = 1;   // no qualities: ReadSelection.hpp:1047-1051
            if (i < ncopy) {
                omin[dst + i] = pmin[src + i];
                opos[dst + i] = ppos[src + i];
                odir[dst + i] = pdir[src + i];
                if (HAS_QUAL) {   // getMinQuality over the original span (ReadSelection.hpp:1302-1320)
                    const uint8_t *qq = qual + qual_off[r];
                    mq = 255;
                    for (uint32_t b = pos_s[src + i], e = pos_e[src + i]; b < e; b++) {
                        uint8_t q = (uint8_t)(qq[b] - 33);
                        if (q < mq) mq = q;
                    }
                }
            }
            oqual[dst + i] = mq;
        }
    }
}

// Exact sum of per-base error probabilities as a 128-bit fixed-point integer (units of 2^-64): sum_q T[q] with
// T[q] = float table entry * 2^64 (exact).  The host turns it into the reference's long double error sum
// (ReadSelection.hpp:870-879).  Every T[q] is a multiple of 2^QSHIFT (24-bit mantissas, smallest entry 10^-9.4;
// checked on the host), so T >> QSHIFT fits 56 bits: a lane adds 16 table entries per 16-byte load into a u64 and
// carries into a u32 above it.  One wave per read, 1 KB per wave per load; the 96-entry table sits in LDS.
constexpr int QBINS = 96;   // bins 0..94 = chars 33..127, bin 95 = everything else (contributes 0)
constexpr int QSHIFT = 9;
__global__ __launch_bounds__(256) void quality_sum_kernel(const uint8_t *qual, const uint64_t *qual_off, const uint32_t *len,
                                                          uint32_t n_reads, const uint64_t *tab_shifted,
                                                          uint64_t *sum_lo, uint64_t *sum_hi) {
    // the table by the quality CHARACTER itself (256 entries, zero outside 33 .. 127): a byte of the load is an address after one shift
    // and one mask -- no subtraction, no clamp.  This kernel runs beside the scan, which is bound by the vector unit: what it issues
    // there the scan pays for (6 ms of a 57 ms ONT scan before, tools/scan_ablate_ont.py)
    __shared__ uint64_t T[256];
    for (unsigned i = threadIdx.x; i < 256u; i += blockDim.x) T[i] = (i >= 33u && i < 33u + (unsigned)(QBINS - 1)) ? tab_shifted[i - 33u] : 0ull;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += nwaves) {
        const uint64_t begin = qual_off[r], end = begin + len[r];
        uint64_t acc_lo = 0;
        uint32_t acc_hi = 0;
        for (uint64_t chunk = (begin & ~15ull) + 16ull * lane; chunk < end; chunk += 16ull * 64ull) {
            const uint4 v = *reinterpret_cast<const uint4 *>(qual + chunk);   // the buffer is 16-byte padded at both ends
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
            if (!(chunk >= begin && chunk + 16 <= end)) {       // a read's first and last chunk: the bytes outside it become character 0
#pragma unroll
                for (int b = 0; b < 16; b++)
                    if (chunk + b < begin || chunk + b >= end) w[b >> 2] &= ~(0xFFu << (8 * (b & 3)));
            }
            uint64_t part = 0, part2 = 0;
#pragma unroll
            for (int b = 0; b < 16; b += 2) {
                part += T[(w[b >> 2] >> (8 * (b & 3))) & 0xFFu];
                part2 += T[(w[(b + 1) >> 2] >> (8 * ((b + 1) & 3))) & 0xFFu];
            }
            part += part2;
            const uint64_t nlo = acc_lo + part;
            acc_hi += nlo < acc_lo ? 1u : 0u;
            acc_lo = nlo;
        }
        for (int dlt = 32; dlt >= 1; dlt >>= 1) {     // 96-bit wave reduction
            const uint64_t olo = __shfl_xor(acc_lo, dlt, 64);
            const uint32_t ohi = __shfl_xor(acc_hi, dlt, 64);
            const uint64_t nlo = acc_lo + olo;
            acc_hi += ohi + (nlo < acc_lo ? 1u : 0u);
            acc_lo = nlo;
        }
        if (lane == 0) {
            sum_lo[r] = acc_lo << QSHIFT;
            sum_hi[r] = ((uint64_t)acc_hi << QSHIFT) | (acc_lo >> (64 - QSHIFT));
        }
    }
}

__global__ void capacity_kernel(const uint32_t *len, uint32_t n_reads, float density, uint32_t *cap) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_reads) {
        // generous first guess: 3x the expected count + slack; overflowing reads are re-run exactly
        float e = (float)len[i] * density * 3.0f + 32.0f;
        cap[i] = (uint32_t)e;
    }
}

// after the main launch: list the reads that overflowed their padded slots and the complexity suspects
__global__ void post_scan_lists_kernel(const uint32_t *count, const uint32_t *cap, const uint8_t *flags, uint32_t n_reads,
                                       uint32_t *over_list, uint32_t *suspect_list, uint32_t *counters /* [0]=over [1]=suspect */) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_reads) return;
    if (count[i] > cap[i]) over_list[atomicAdd(&counters[0], 1u)] = (uint32_t)i;
    if (flags[i] & READ_SUSPECT) suspect_list[atomicAdd(&counters[1], 1u)] = (uint32_t)i;
}

// reads that outgrew the fast kernel's stage: their counts, then their places behind the regions
__global__ void gather_counts_kernel(const uint32_t *list, uint32_t n_list, const uint32_t *count, uint32_t *out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_list) out[i] = count[list[i]];
}

// suspects of the complexity bound among the listed reads (the general kernel only flags them): appended to the suspect list
__global__ void listed_suspects_kernel(const uint32_t *list, uint32_t n_list, const uint8_t *flags, uint32_t *suspect_list, uint32_t *n_suspect) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_list && (flags[list[i]] & READ_SUSPECT)) suspect_list[atomicAdd(n_suspect, 1u)] = list[i];
}

__global__ void place_overflow_kernel(const uint32_t *list, const uint64_t *start, const uint32_t *cnt, uint32_t n_list, uint64_t *cap_off,
                                      uint64_t *begin) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_list) return;
    const uint32_t r = list[i];
    cap_off[r] = start[i];
    cap_off[r + 1] = start[i] + cnt[i];      // consecutive listed reads (the list is sorted) agree on this entry
    begin[r] = start[i];
}

__global__ void apply_low_quality_kernel(const uint8_t *low, uint32_t n_reads, uint32_t *count, uint8_t *flags,
                                         unsigned long long *dropped /* may be null: sum of the counts cleared */) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_reads && low[i]) {
        if (dropped && count[i]) atomicAdd(dropped, (unsigned long long)count[i]);
        count[i] = 0; flags[i] |= (uint8_t)MDBG_READ_LOW_QUALITY;
    }
}

}  // namespace mdbg

using namespace mdbg;

// ReadSelection.hpp:870-879: float meanReadError = errorSum / n; meanReadQuality = -10.0f * log10(meanReadError).
// Evaluated at run time (volatile) so that n == 0 yields the same NaN bits (0xFFC00000 on x86-64)
// the reference writes into read_data_init.txt.
static float mean_quality_from_sum(long double error_sum_in, size_t n_in) {
    volatile long double error_sum = error_sum_in;
    volatile size_t n = n_in;
    volatile float mean_err = (float)(error_sum / n);
    return -10.0f * log10f(mean_err);
}

// grow-only pinned host buffer of the context (device -> host copies that must not wait for the main stream)
static int pinned_reserve(mdbg_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->pinned_bytes) return MDBG_OK;
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    ctx->pinned = nullptr; ctx->pinned_bytes = 0;
    const size_t want = bytes + bytes / 4;
    hipError_t e = hipHostMalloc(&ctx->pinned, want, hipHostMallocDefault);
    if (e != hipSuccess) { ctx->pinned = nullptr; return set_error(ctx, MDBG_ENOMEM, "hipHostMalloc(%zu): %s", want, hipGetErrorString(e)); }
    ctx->pinned_bytes = want;
    return MDBG_OK;
}

// A/B switch for the rule "one scan kernel at a time per device" (profiles/r03b_scan_mutex_ab.txt)
static bool scans_may_interleave() {
    static const bool on = getenv("MDBG_SCAN_NO_MUTEX") != nullptr;
    return on;
}

static std::mutex &device_scan_mutex(int device) {
    static std::mutex m[64];
    return m[(unsigned)device % 64u];
}

// The same rule ACROSS processes (two tools on one GPU, the ranks of a test sharing device 0): an advisory lock on a file named after
// the device's PCI address, taken inside the process-wide mutex.  Best effort: no lock file, no cross-process turn-taking
// (MDBG_SCAN_NO_XPROC_LOCK=1 switches it off).  flock costs about a microsecond; a scan launch is milliseconds.
static int device_scan_lockfile(int device) {
    static int fds[64];
    static std::once_flag once[64];
    const unsigned d = (unsigned)device % 64u;
    std::call_once(once[d], [&] {
        fds[d] = -1;
        if (getenv("MDBG_SCAN_NO_XPROC_LOCK")) return;
        char bus[64] = {0};
        if (hipDeviceGetPCIBusId(bus, (int)sizeof bus, device) != hipSuccess) snprintf(bus, sizeof bus, "device%d", device);
        for (char *c = bus; *c; c++) if (*c == ':' || *c == '.' || *c == '/') *c = '_';
        for (const char *dir : {"/dev/shm", "/tmp"}) {
            std::string path = std::string(dir) + "/mdbg_scan_turn_" + bus + ".lock";
            const int fd = open(path.c_str(), O_RDWR | O_CREAT | O_CLOEXEC, 0666);
            if (fd >= 0) { (void)fchmod(fd, 0666); fds[d] = fd; return; }
        }
    });
    return fds[d];
}

// one scan kernel at a time per device: the turn is held from the launch to the read-back of the counters
struct ScanTurn {
    std::unique_lock<std::mutex> in_process;
    int fd = -1;
    bool held = false;
    explicit ScanTurn(int device) : in_process(device_scan_mutex(device), std::defer_lock), fd(device_scan_lockfile(device)) {}
    void lock() {
        in_process.lock();
        if (fd >= 0) while (flock(fd, LOCK_EX) != 0 && errno == EINTR) {}
        held = true;
    }
    void unlock() {
        if (!held) return;
        if (fd >= 0) (void)flock(fd, LOCK_UN);
        in_process.unlock();
        held = false;
    }
    bool owns_lock() const { return held; }
    ~ScanTurn() { unlock(); }
};

template <bool HPC, bool Q, bool N>
static void launch_variant(mdbg_ctx *ctx, const ScanArgs &a, unsigned max_blocks, uint32_t n_items) {
    // A few reads per wave, then the wave retires.  One resident generation of persistent waves (the first design)
    // took 17.9 ms for 1 M x 10 kb reads although every wave had the same work: CUs do not all run at the same
    // speed, and the kernel ended with the slowest.  Dealt out by the dispatcher as slots free up, the same reads take
    // 15.4 ms; short-lived workgroups also let the kernels of another stream (a second batch in flight, RCCL) in
    // instead of queueing them behind the resident generation.  The per-block set-up (1024-entry LUT) is noise next
    // to a 10 kb read.  2 reads per wave measured best with two batches in flight: a waiting kernel of the other batch gets
    // a slot within ~0.2 ms (MDBG_SCAN_READS_PER_WAVE to tune).
    (void)max_blocks;
    const uint64_t per_wave = ctx->scan_reads_per_wave;
    uint64_t blocks = ((uint64_t)n_items + SCAN_WAVES * per_wave - 1) / (SCAN_WAVES * per_wave);
    if (blocks < 1) blocks = 1;
    if (blocks > 0x7FFFFFFFull) blocks = 0x7FFFFFFFull;
    hipLaunchKernelGGL((scan_kernel<HPC, Q, N>), dim3((unsigned)blocks), dim3(SCAN_BLOCK), 0, ctx->stream, a);
}

static int launch_scan(mdbg_ctx *ctx, ScanArgs &a, bool hpc, bool has_q, bool has_n, uint32_t n_items) {
    a.n_reads = n_items;
    const unsigned max_blocks = (unsigned)ctx->n_cu * 8u;
    static const bool no_fast = getenv("MDBG_SCAN_NO_FAST") != nullptr;      // A/B: the general kernel for everything
    static const bool no_approx = getenv("MDBG_SCAN_NO_APPROX") != nullptr;  // A/B: full 64-bit verdict at every position
    const bool fast = (!has_q || a.cursor) && !has_n && !a.subset && !no_fast;
    // "table_cu_count": the context's own stream is confined to a few CUs; the block-structured kernel goes to a stream over all of
    // them, ordered after what the context has queued so far and before what it queues next
    hipStream_t on = ctx->stream;
    hipEvent_t ordered = nullptr;
    if (fast && ctx->scan_stream) {
        MDBG_HIP_CHECK(ctx, hipEventCreateWithFlags(&ordered, hipEventDisableTiming));
        hipError_t e = hipEventRecord(ordered, ctx->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->scan_stream, ordered, 0);
        if (e != hipSuccess) { (void)hipEventDestroy(ordered); return set_error(ctx, MDBG_EHIP, "scan stream hand-over: %s", hipGetErrorString(e)); }
        on = ctx->scan_stream;
    }
    {
        LaunchTimer timer(ctx, "scan", on);
        if (fast) {
            // plain ACGT without qualities: the block-structured kernel; a few reads per wave, then the wave retires
            const uint64_t per_wave = ctx->scan_reads_per_wave;
            uint64_t blocks = ((uint64_t)n_items + FAST_WAVES * per_wave - 1) / (FAST_WAVES * per_wave);
            if (blocks < 1) blocks = 1;
            if (blocks > 0x7FFFFFFFull) blocks = 0x7FFFFFFFull;
            // bump mode: candidates by the upper half of the hash (span_step<APPROX>); the host's re-run of a read covers a false one
            const dim3 g((unsigned)blocks), b(FAST_BLOCK);
            using FastKernel = void (*)(ScanArgs);
            // (not when the candidate limit would saturate -- a threshold within a few 2^32 of 2^64, density 1.0f: the full verdict then)
            const bool approx = a.cursor && !no_approx && (a.threshold >> 32) + 3ull + (uint64_t)a.cand_slack < 0xFFFFFFFFull;
            const FastKernel fk = approx ? (hpc ? (has_q ? scan_fast_kernel<true, true, true> : scan_fast_kernel<true, false, true>)
                                                : (has_q ? scan_fast_kernel<false, true, true> : scan_fast_kernel<false, false, true>))
                                         : (hpc ? (has_q ? scan_fast_kernel<true, true, false> : scan_fast_kernel<true, false, false>)
                                                : (has_q ? scan_fast_kernel<false, true, false> : scan_fast_kernel<false, false, false>));
            // Unused dynamic LDS caps this kernel's blocks per CU (mdbg_set_option): "scan_lds_pad" gives the bytes as they are;
            // "scan_lds_reserve" says how much of a CU's LDS the kernel is to LEAVE to other contexts' kernels, whatever the variant
            // (30 KB a block for FASTA with homopolymer compression, 36.5 KB for FASTQ): as many blocks as fit beside the reserve, and
            // enough padding that one more does not.
            unsigned lds_pad = ctx->scan_lds_pad;
            if (!lds_pad && ctx->scan_lds_reserve) {
                hipFuncAttributes at;
                if (hipFuncGetAttributes(&at, reinterpret_cast<const void *>(fk)) == hipSuccess && at.sharedSizeBytes > 0) {
                    const size_t total = ctx->lds_per_cu ? ctx->lds_per_cu : 163840u, lds = at.sharedSizeBytes;
                    size_t fit = total > ctx->scan_lds_reserve ? (total - ctx->scan_lds_reserve) / lds : 1;
                    // never fewer than two blocks per CU: one block would have to be padded to more than half a CU's LDS -- beyond the 64 KB
                    // a block may ask for, the launch would fail with hipErrorInvalidValue instead of capping anything (round-4 ADVICE);
                    // a reserve that large is answered with the two-block form (and whatever LDS that leaves)
                    if (fit < 2) fit = 2;
                    const size_t need = total / (fit + 1) + 1;          // a block of this size: fit + 1 of them are more than a CU has
                    if (need > lds) lds_pad = (unsigned)(((need - lds) + 255u) / 256u * 256u);
                    const size_t most = (size_t)65536 > lds ? (size_t)65536 - lds : 0;        // static + dynamic LDS of a block stay within 64 KB
                    if (lds_pad > most) lds_pad = (unsigned)(most / 256u * 256u);
                } else (void)hipGetLastError();
            }
            hipLaunchKernelGGL(fk, g, b, lds_pad, on, a);
        } else if (hpc) {
            if (has_n) { if (has_q) launch_variant<true, true, true>(ctx, a, max_blocks, n_items); else launch_variant<true, false, true>(ctx, a, max_blocks, n_items); }
            else { if (has_q) launch_variant<true, true, false>(ctx, a, max_blocks, n_items); else launch_variant<true, false, false>(ctx, a, max_blocks, n_items); }
        } else {
            if (has_n) { if (has_q) launch_variant<false, true, true>(ctx, a, max_blocks, n_items); else launch_variant<false, false, true>(ctx, a, max_blocks, n_items); }
            else { if (has_q) launch_variant<false, true, false>(ctx, a, max_blocks, n_items); else launch_variant<false, false, false>(ctx, a, max_blocks, n_items); }
        }
    }
    if (ordered) {
        hipError_t e = hipEventRecord(ordered, on);
        if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ordered, 0);
        (void)hipEventDestroy(ordered);         // released by the runtime once the recorded work is done
        if (e != hipSuccess) return set_error(ctx, MDBG_EHIP, "scan stream hand-back: %s", hipGetErrorString(e));
    }
    MDBG_HIP_CHECK(ctx, hipGetLastError());
    return MDBG_OK;
}

extern "C" int mdbg_scan(mdbg_ctx *ctx, const mdbg_reads *reads, const mdbg_scan_params *p, mdbg_minimizers **out) try {
    if (!ctx || !reads || !p || !out) return set_error(ctx, MDBG_EINVAL, "mdbg_scan: null argument");
    if (p->quality_window != 0 && p->quality_window != 1) return set_error(ctx, MDBG_EINVAL, "mdbg_scan: quality_window must be 0 or 1");
    if (p->minimizer_size < 2 || p->minimizer_size > 16)
        return set_error(ctx, MDBG_EINVAL, "mdbg_scan: minimizer_size %u outside [2,16]", p->minimizer_size);
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    MDBG_HIP_CHECK(ctx, reads_ready_on(ctx, reads));        // an upload still in flight (mdbg_reads_from_packed_async): ordered on the device
    const uint32_t n = reads->n_reads;
    const bool hpc = p->hpc != 0, has_q = reads->has_qual && !p->ignore_qualities, has_n = reads->has_invalid;
    mdbg_minimizers *m = new mdbg_minimizers();
    m->n_reads = n;
    m->from_scan = true;
    auto fail = [&](int rc) { delete m; return rc; };
    hipError_t e;

    DevBuf<uint32_t> d_cap, d_count, d_rep;
    DevBuf<uint64_t> d_cap_off;
    int rc;
    if ((rc = d_cap.alloc(ctx, n)) || (rc = d_count.alloc(ctx, n)) || (rc = d_cap_off.alloc(ctx, (size_t)n + 1)) ||
        (rc = m->d_flags.alloc(ctx, n)) || (rc = m->d_len.alloc(ctx, n)) || (rc = m->d_off.alloc(ctx, (size_t)n + 1)))
        return fail(rc);
    if (p->n_repetitive) {
        std::vector<uint32_t> rep(p->repetitive, p->repetitive + p->n_repetitive);
        std::sort(rep.begin(), rep.end());
        if ((rc = d_rep.alloc(ctx, rep.size()))) return fail(rc);
        e = memcpy_sync(ctx, d_rep.p, rep.data(), rep.size() * 4, hipMemcpyHostToDevice);
        if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "copy of repetitive set failed"));
    }
    e = hipMemcpyAsync(m->d_len.p, reads->d_len.p, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "len copy failed: %s", hipGetErrorString(e)));

    // ---- mean read quality (exact 128-bit sums on the device, long double finish on the host) ----
    std::vector<uint8_t> low_quality;
    bool any_low_quality = false;
    DevBuf<uint64_t> d_qtab, d_slo, d_shi;
    struct SideGuard { hipStream_t s = nullptr; ~SideGuard() { if (s) (void)hipStreamSynchronize(s); } } side_guard;   // before the buffers go
    bool quality_pending = false;
    // the host's share of the mean qualities: the reference's long double division and log10 per read, on a few host threads
    auto quality_finish = [&]() -> int {
        if (!quality_pending) return MDBG_OK;
        quality_pending = false;
        hipError_t qe;
        if ((qe = hipStreamSynchronize(ctx->side_stream)) != hipSuccess)
            return set_error(ctx, MDBG_EHIP, "quality sums download failed: %s", hipGetErrorString(qe));
        const uint64_t *slo = (const uint64_t *)ctx->pinned, *shi = slo + n;
        const uint32_t *lens = (const uint32_t *)(shi + n);
        m->h_mean_quality.resize(n);
        low_quality.assign(n, 0);
        const bool filter = p->apply_read_filters != 0;
        const float min_q = p->min_read_quality;
        const unsigned n_thr = n < 65536 ? 1u : std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        std::vector<char> any(n_thr, 0);
        auto work = [&](unsigned t) {
            const uint32_t r0 = (uint32_t)((uint64_t)n * t / n_thr), r1 = (uint32_t)((uint64_t)n * (t + 1) / n_thr);
            for (uint32_t r = r0; r < r1; r++) {
                long double sq = ((long double)shi[r] * 18446744073709551616.0L + (long double)slo[r]) / 18446744073709551616.0L;
                float mq = mean_quality_from_sum(sq, lens[r]);
                m->h_mean_quality[r] = mq;
                if (filter && mq < min_q) { low_quality[r] = 1; any[t] = 1; }   // ReadSelection.hpp:901-909
            }
        };
        if (n_thr == 1) work(0);
        else {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < n_thr; t++) pool.emplace_back(work, t);
            for (auto &th : pool) th.join();
        }
        for (char c : any) any_low_quality |= c != 0;
        return MDBG_OK;
    };
    if (has_q && n) {
        // table exactly as the reference builds it (ReadSelection.hpp:101-104, Commons.hpp:2338-2341)
        std::vector<uint64_t> tsh(QBINS, 0);
        for (int q = 33; q <= 127; q++) {
            float qq = (float)(uint8_t)(q - 33);
            float t = powf(10.0f, -qq / 10.0f);
            long double scaled = (long double)t * 18446744073709551616.0L;   // * 2^64, exact (24-bit mantissa)
            unsigned __int128 v = (unsigned __int128)scaled;
            if ((uint64_t)v & ((1ull << QSHIFT) - 1ull)) return fail(set_error(ctx, MDBG_EINVAL, "quality table entry %d is not a multiple of 2^%d", q, QSHIFT));
            tsh[q - 33] = (uint64_t)(v >> QSHIFT);
        }
        if ((rc = d_qtab.alloc(ctx, QBINS)) || (rc = d_slo.alloc(ctx, n)) || (rc = d_shi.alloc(ctx, n))) return fail(rc);
        if ((e = memcpy_sync(ctx, d_qtab.p, tsh.data(), QBINS * 8, hipMemcpyHostToDevice)) != hipSuccess)
            return fail(set_error(ctx, MDBG_EHIP, "quality table upload failed: %s", hipGetErrorString(e)));
        // The sums run BESIDE the scan, on the side stream (round 5; until then in front of it, on the context's stream: 61 ms of a 0.52 s
        // pass over 200 Gbp of ONT reads): they are a second pass over the quality bytes at 3 TB/s -- memory traffic the VALU-bound scan
        // leaves idle -- in blocks of 768 bytes of LDS that fit beside the scan's.  Their way to the host (pinned memory) and the host's share
        // (quality_finish) follow on the same stream: nothing of the mean quality is needed before the rows are counted.
        // ("scan_quality_stream" 0: in front of the scan as before, A/B.)
        if ((rc = pinned_reserve(ctx, (size_t)n * 20))) return fail(rc);
        {
            const bool beside = ctx->scan_quality_beside != 0;
            hipStream_t qs = beside ? ctx->side_stream : ctx->stream;
            hipEvent_t ev;
            if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "hipEventCreate: %s", hipGetErrorString(e)));
            if (beside) {                                   // behind whatever the context's stream has queued so far (the reads, the table upload)
                (void)hipEventRecord(ev, ctx->stream);
                (void)hipStreamWaitEvent(ctx->side_stream, ev, 0);
            }
            side_guard.s = ctx->side_stream;     // from here on every way out waits for the side stream before d_slo / d_shi / d_qtab go back to the pool (round-5 ADVICE)
            {
                LaunchTimer timer(ctx, "quality_sum", qs);
                unsigned blocks = grid_for((uint64_t)n * 64, 256, (unsigned)ctx->n_cu * 8u);
                hipLaunchKernelGGL(quality_sum_kernel, dim3(blocks), dim3(256), 0, qs, reads->d_qual.p, reads->d_qual_off.p,
                                   reads->d_len.p, n, d_qtab.p, d_slo.p, d_shi.p);
            }
            if (!beside) {
                (void)hipEventRecord(ev, ctx->stream);
                (void)hipStreamWaitEvent(ctx->side_stream, ev, 0);
            }
            (void)hipEventDestroy(ev);
            uint8_t *h = (uint8_t *)ctx->pinned;
            if ((e = hipMemcpyAsync(h, d_slo.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->side_stream)) != hipSuccess ||
                (e = hipMemcpyAsync(h + (size_t)n * 8, d_shi.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->side_stream)) != hipSuccess ||
                (e = hipMemcpyAsync(h + (size_t)n * 16, reads->d_len.p, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->side_stream)) != hipSuccess)
                return fail(set_error(ctx, MDBG_EHIP, "quality sums download failed: %s", hipGetErrorString(e)));
        }
        side_guard.s = ctx->side_stream;
        quality_pending = true;
    } else {
        // no qualities: every read's mean quality is the NaN the reference writes; kept as one value, not n of them (a 40 MB
        // host fill per 10 M reads sat on the critical path in front of the kernel launch)
        m->h_mean_quality.clear();
        m->mean_quality_all = mean_quality_from_sum(0, 0);
    }

    // ---- plain ACGT without qualities: the block-structured kernel writes every read's rows once, where a wave found room
    // (mdbg_minimizers::scattered); one read-back instead of four.  Batches it cannot take -- a read selecting more than the
    // LDS stage holds, more rows than the estimate allowed for -- go through the general path below.
    static const bool no_bump = getenv("MDBG_SCAN_NO_BUMP") != nullptr || getenv("MDBG_SCAN_NO_FAST") != nullptr;
    // (a read may stage STAGE_CAP minimizers: batches whose longest read is expected to select more go straight to the general path)
    // A read may stage STAGE_CAP minimizers; the few that select more are re-run by the general kernel into a reserve behind
    // the regions.  Batches whose AVERAGE read is expected to outgrow the stage go straight to the general path.
    // A batch in which a few reads carry side-mask bits (an N here and there, soft-masked stretches) stays on this path: the block
    // kernel skips those reads (ScanArgs::skip) and they alone are counted, placed and scanned by the general kernel below.
    const uint32_t n_masked = has_n ? reads->n_masked : 0u;
    const bool route_masked = has_n && reads->d_masked.p && (uint64_t)n_masked * 8ull <= (uint64_t)n + 128ull;
    if (n && (!has_n || route_masked) && !no_bump && p->density < 0.2f && reads->max_len < (1u << 31) &&
        (double)reads->n_bases / (double)n * (double)p->density * (hpc ? 0.8 : 1.0) * 1.4 + 24.0 < (double)STAGE_CAP) {
        // the output arrays are cut into regions, each with its own cursor (reads are dealt to the waves round-robin, so the regions
        // fill evenly); small batches use one
        uint32_t n_regions = 1;
        while (n_regions < 64u && (uint64_t)n_regions * 8192ull <= n) n_regions <<= 1;
        const uint64_t region_cap = ((uint64_t)((double)reads->n_bases * (double)p->density * 1.3) + 64ull * n) / n_regions + 4096ull;
        const uint64_t capacity = region_cap * n_regions;
        const uint64_t reserve = capacity / 16 + 65536;       // rows behind the regions for reads that outgrow the stage
        constexpr uint32_t CTL_OVER = 64, CTL_DROPPED = 65, CTL_WORDS = 66;     // u64 words: cursors, {n_over, n_suspect}, rows dropped
        DevBuf<uint32_t> d_over, d_susp;
        DevBuf<unsigned long long> d_ctl;
        if ((rc = m->d_begin.alloc(ctx, n)) || (rc = m->d_cnt.alloc(ctx, n)) || (rc = d_over.alloc(ctx, n)) || (rc = d_susp.alloc(ctx, n)) ||
            (rc = d_ctl.alloc(ctx, CTL_WORDS)) || (rc = m->d_min.alloc(ctx, capacity + reserve)) || (rc = m->d_pos.alloc(ctx, capacity + reserve)) ||
            (rc = m->d_dir.alloc(ctx, capacity + reserve)) || (rc = m->d_mqual.alloc(ctx, capacity + reserve)))
            return fail(rc);
        (void)hipMemsetAsync(d_ctl.p, 0, CTL_WORDS * 8, ctx->stream);
        ScanArgs a{};
        a.words = reads->d_words.p; a.word_off = reads->d_word_off.p; a.len = reads->d_len.p;
        a.qual = has_q ? reads->d_qual.p : nullptr;
        a.qual_off = has_q ? reads->d_qual_off.p : nullptr;
        a.K = p->minimizer_size;
        a.threshold = density_threshold(p->density);
        a.trim = p->no_end_trim ? 0u : 1u;
        a.q_last = p->quality_window == 1 ? 1u : 0u;
        a.rep = d_rep.p; a.n_rep = p->n_repetitive;
        a.apply_filters = p->apply_read_filters;
        a.out_min = m->d_min.p; a.out_pos = m->d_pos.p; a.out_dir = m->d_dir.p; a.out_mqual = m->d_mqual.p;
        a.out_count = m->d_cnt.p; a.out_flags = m->d_flags.p;
        a.cursor = d_ctl.p; a.n_regions = n_regions; a.out_capacity = region_cap; a.out_begin = m->d_begin.p;
        a.over_list = d_over.p; a.suspect_list = d_susp.p; a.list_counters = (uint32_t *)(d_ctl.p + CTL_OVER);
        a.cand_slack = ctx->scan_cand_slack;
        a.wave_priority = ctx->scan_wave_priority;
        a.skip = route_masked ? reads->d_masked.p : nullptr;
        unsigned long long h_ctl[CTL_WORDS];
        {
            ScanTurn scan_turn(ctx->device);      // one scan kernel at a time per device (see below)
            if (!scans_may_interleave()) scan_turn.lock();
            if ((rc = launch_scan(ctx, a, hpc, has_q, false, n))) return fail(rc);
            if ((rc = quality_finish())) return fail(rc);          // host work while the kernel runs
            e = memcpy_sync(ctx, h_ctl, d_ctl.p, CTL_WORDS * 8, hipMemcpyDeviceToHost);
        }
        if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "scan counters copy failed: %s", hipGetErrorString(e)));
        uint64_t rows = 0;
        bool fits = true;
        for (uint32_t g = 0; g < n_regions; g++) { rows += h_ctl[g]; fits = fits && h_ctl[g] <= region_cap; }
        const uint32_t n_over = (uint32_t)h_ctl[CTL_OVER];
        uint32_t n_suspect = (uint32_t)(h_ctl[CTL_OVER] >> 32);
        // ---- reads the block kernel did not finish: those that outgrew its stage (long reads; listed with their exact counts) and
        // those it skipped for their side masks (counted here by the general kernel with no room to write).  Both kinds are placed
        // in read order behind the regions and scanned by the general kernel, each kind with its own variant.
        ScanArgs masked_args = a;
        std::vector<uint32_t> m_cnt;
        bool counted = true;
        if (fits && n_masked && route_masked) {
            if ((e = hipMemsetAsync(d_cap_off.p, 0, ((size_t)n + 1) * 8, ctx->stream)) != hipSuccess)
                return fail(set_error(ctx, MDBG_EHIP, "capacity clear failed: %s", hipGetErrorString(e)));
            masked_args.cursor = nullptr; masked_args.skip = nullptr;
            masked_args.subset = reads->d_masked_list.p;
            masked_args.cap_off = d_cap_off.p;                       // every read: no room -- the pass only counts
            masked_args.invalid = reads->d_invalid.p;
            masked_args.brk = reads->has_break ? reads->d_break.p : nullptr;
            masked_args.q_last = p->quality_window == 1 ? 1u : 0u;
            masked_args.inline_minq = 1;
            masked_args.out_count = m->d_cnt.p; masked_args.out_flags = m->d_flags.p;
            if ((rc = launch_scan(ctx, masked_args, hpc, has_q, true, n_masked))) return fail(rc);
            hipLaunchKernelGGL(listed_suspects_kernel, dim3(grid_for(n_masked, 256)), dim3(256), 0, ctx->stream, reads->d_masked_list.p, n_masked,
                               m->d_flags.p, d_susp.p, (uint32_t *)(d_ctl.p + CTL_OVER) + 1);
            DevBuf<uint32_t> d_mc;
            if ((rc = d_mc.alloc(ctx, n_masked))) return fail(rc);
            hipLaunchKernelGGL(gather_counts_kernel, dim3(grid_for(n_masked, 256)), dim3(256), 0, ctx->stream, reads->d_masked_list.p, n_masked, m->d_cnt.p, d_mc.p);
            m_cnt.resize(n_masked);
            unsigned long long ctl_over = 0;
            if ((e = memcpy_sync(ctx, m_cnt.data(), d_mc.p, (size_t)n_masked * 4, hipMemcpyDeviceToHost)) != hipSuccess ||
                (e = memcpy_sync(ctx, &ctl_over, d_ctl.p + CTL_OVER, 8, hipMemcpyDeviceToHost)) != hipSuccess)
                return fail(set_error(ctx, MDBG_EHIP, "masked counts copy failed: %s", hipGetErrorString(e)));
            n_suspect = (uint32_t)(ctl_over >> 32);
        } else if (n_masked && route_masked) counted = false;
        bool placed = n_over == 0 && (!route_masked || n_masked == 0);
        if (fits && counted && !placed && n_over <= n / 4 + 16) {
            std::vector<uint32_t> list(n_over), cnts(n_over);
            DevBuf<uint32_t> d_oc;
            if ((rc = d_oc.alloc(ctx, (size_t)n_over + 1))) return fail(rc);
            if (n_over) {
                hipLaunchKernelGGL(gather_counts_kernel, dim3(grid_for(n_over, 256)), dim3(256), 0, ctx->stream, d_over.p, n_over, m->d_cnt.p, d_oc.p);
                if ((e = memcpy_sync(ctx, list.data(), d_over.p, (size_t)n_over * 4, hipMemcpyDeviceToHost)) != hipSuccess ||
                    (e = memcpy_sync(ctx, cnts.data(), d_oc.p, (size_t)n_over * 4, hipMemcpyDeviceToHost)) != hipSuccess)
                    return fail(set_error(ctx, MDBG_EHIP, "overflow list copy failed: %s", hipGetErrorString(e)));
            }
            std::vector<uint32_t> m_list(route_masked ? n_masked : 0u);
            if (!m_list.empty() && (e = memcpy_sync(ctx, m_list.data(), reads->d_masked_list.p, m_list.size() * 4, hipMemcpyDeviceToHost)) != hipSuccess)
                return fail(set_error(ctx, MDBG_EHIP, "masked list copy failed: %s", hipGetErrorString(e)));
            // one order over both kinds: (read, count, kind)
            struct Item { uint32_t read, cnt, kind; };
            std::vector<Item> items;
            items.reserve((size_t)n_over + m_list.size());
            for (uint32_t i = 0; i < n_over; i++) items.push_back(Item{list[i], cnts[i], 0u});
            for (size_t i = 0; i < m_list.size(); i++) items.push_back(Item{m_list[i], m_cnt[i], 1u});
            std::sort(items.begin(), items.end(), [](const Item &x, const Item &y) { return x.read < y.read; });
            std::vector<uint32_t> slist[2], scnt[2];
            std::vector<uint64_t> sstart[2];
            uint64_t at = capacity;
            for (const Item &it : items) { slist[it.kind].push_back(it.read); scnt[it.kind].push_back(it.cnt); sstart[it.kind].push_back(at); at += it.cnt; }
            if (at - capacity <= reserve) {
                DevBuf<uint32_t> scratch_count;
                DevBuf<uint8_t> scratch_flags;
                if ((rc = scratch_count.alloc(ctx, n)) || (rc = scratch_flags.alloc(ctx, n))) return fail(rc);
                DevBuf<uint64_t> d_start[2];
                DevBuf<uint32_t> d_list[2], d_cnt2[2];
                for (int kind = 0; kind < 2; kind++) {
                    const uint32_t nk = (uint32_t)slist[kind].size();
                    if (!nk) continue;
                    if ((rc = d_start[kind].alloc(ctx, nk)) || (rc = d_list[kind].alloc(ctx, nk)) || (rc = d_cnt2[kind].alloc(ctx, nk))) return fail(rc);
                    if ((e = memcpy_sync(ctx, d_list[kind].p, slist[kind].data(), (size_t)nk * 4, hipMemcpyHostToDevice)) != hipSuccess ||
                        (e = memcpy_sync(ctx, d_cnt2[kind].p, scnt[kind].data(), (size_t)nk * 4, hipMemcpyHostToDevice)) != hipSuccess ||
                        (e = memcpy_sync(ctx, d_start[kind].p, sstart[kind].data(), (size_t)nk * 8, hipMemcpyHostToDevice)) != hipSuccess)
                        return fail(set_error(ctx, MDBG_EHIP, "overflow plan upload failed: %s", hipGetErrorString(e)));
                    hipLaunchKernelGGL(place_overflow_kernel, dim3(grid_for(nk, 256)), dim3(256), 0, ctx->stream, d_list[kind].p, d_start[kind].p, d_cnt2[kind].p, nk,
                                       d_cap_off.p, m->d_begin.p);
                }
                for (int kind = 0; kind < 2; kind++) {
                    const uint32_t nk = (uint32_t)slist[kind].size();
                    if (!nk) continue;
                    ScanArgs b = kind ? masked_args : a;
                    b.cursor = nullptr; b.skip = nullptr;
                    b.subset = d_list[kind].p;
                    b.cap_off = d_cap_off.p;
                    b.q_last = p->quality_window == 1 ? 1u : 0u;
                    b.inline_minq = 1;
                    b.out_count = scratch_count.p; b.out_flags = scratch_flags.p;
                    if ((rc = launch_scan(ctx, b, hpc, has_q, kind == 1, nk))) return fail(rc);
                }
                if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "overflow re-run failed: %s", hipGetErrorString(e)));
                rows += at - capacity;
                placed = true;
            }
        }
        if (placed && fits) {
            uint64_t dropped = 0;
            if (n_suspect) {
                LaunchTimer timer(ctx, "complexity_exact");
                hipLaunchKernelGGL(complexity_exact_kernel, dim3(grid_for((uint64_t)n_suspect * 64, 256, (unsigned)ctx->n_cu * 8u)), dim3(256), 0,
                                   ctx->stream, reads->d_words.p, reads->d_word_off.p, reads->d_len.p, d_susp.p, n_suspect,
                                   m->d_cnt.p, m->d_flags.p, d_ctl.p + CTL_DROPPED);
                if ((e = memcpy_sync(ctx, &dropped, d_ctl.p + CTL_DROPPED, 8, hipMemcpyDeviceToHost)) != hipSuccess)
                    return fail(set_error(ctx, MDBG_EHIP, "complexity pass failed: %s", hipGetErrorString(e)));
            }
            if (any_low_quality) {          // reads below --min-read-quality keep their (empty) record (ReadSelection.hpp:901-909)
                DevBuf<uint8_t> d_low;
                if ((rc = d_low.alloc(ctx, n))) return fail(rc);
                if ((e = memcpy_sync(ctx, d_low.p, low_quality.data(), n, hipMemcpyHostToDevice)) != hipSuccess)
                    return fail(set_error(ctx, MDBG_EHIP, "low-quality flags upload failed: %s", hipGetErrorString(e)));
                hipLaunchKernelGGL(apply_low_quality_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, d_low.p, n, m->d_cnt.p, m->d_flags.p,
                                   d_ctl.p + CTL_DROPPED);
                if ((e = memcpy_sync(ctx, &dropped, d_ctl.p + CTL_DROPPED, 8, hipMemcpyDeviceToHost)) != hipSuccess)
                    return fail(set_error(ctx, MDBG_EHIP, "low-quality pass failed: %s", hipGetErrorString(e)));
            }
            if (!has_q) (void)hipMemsetAsync(m->d_mqual.p, 1, capacity + reserve, ctx->stream);     // no qualities: ReadSelection.hpp:1047-1051
            m->scattered = true;
            m->owner = ctx;
            m->n_rows = capacity + reserve;        // extent of the row arrays (regions are not filled to the brim)
            m->n_min = rows - dropped;
            if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "scan failed: %s", hipGetErrorString(e)));
            *out = m;
            return MDBG_OK;
        }
        // not this batch: give the buffers back and take the general path
        m->d_begin.release(); m->d_cnt.release(); m->d_min.release(); m->d_pos.release(); m->d_dir.release(); m->d_mqual.release();
    }

    if (n) hipLaunchKernelGGL(capacity_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream,
                              reads->d_len.p, n, p->density, d_cap.p);
    if ((rc = exclusive_scan_u32(ctx, d_cap.p, d_cap_off.p, n))) return fail(rc);
    uint64_t cap_total = 0;
    e = memcpy_sync(ctx, &cap_total, d_cap_off.p + n, 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "cap total copy failed"));

    DevBuf<uint32_t> p_min, p_pos, p_os, p_oe;
    DevBuf<uint8_t> p_dir;
    if ((rc = p_min.alloc(ctx, cap_total)) || (rc = p_pos.alloc(ctx, cap_total)) || (rc = p_dir.alloc(ctx, cap_total)))
        return fail(rc);
    if (has_q && ((rc = p_os.alloc(ctx, cap_total)) || (rc = p_oe.alloc(ctx, cap_total)))) return fail(rc);

    ScanArgs a{};
    a.words = reads->d_words.p; a.word_off = reads->d_word_off.p; a.len = reads->d_len.p;
    a.invalid = has_n ? reads->d_invalid.p : nullptr;
    a.brk = reads->has_break ? reads->d_break.p : nullptr;
    a.qual = has_q ? reads->d_qual.p : nullptr;
    a.qual_off = has_q ? reads->d_qual_off.p : nullptr;
    a.K = p->minimizer_size;
    a.threshold = density_threshold(p->density);
    a.q_last = p->quality_window == 1 ? 1u : 0u;
    a.trim = p->no_end_trim ? 0u : 1u;
    a.rep = d_rep.p; a.n_rep = p->n_repetitive;
    a.apply_filters = p->apply_read_filters;
    a.subset = nullptr;
    a.cap_off = d_cap_off.p;
    a.out_min = p_min.p; a.out_pos = p_pos.p; a.out_dir = p_dir.p;
    a.out_os = p_os.p; a.out_oe = p_oe.p; a.out_mqual = nullptr; a.inline_minq = 0;
    a.out_count = d_count.p; a.out_flags = m->d_flags.p;
    // One scan kernel at a time per device, whatever the number of contexts: two of them interleaved workgroup by
    // workgroup each run at half speed and worse (measured: 2 x 15.5 ms alone, 2 x 30 ms interleaved), while everything
    // else a second context does -- table building, purge, RCCL -- overlaps a running scan nicely.  The lock covers the
    // launch and is released when the kernel has finished (the counter download below waits for it).
    ScanTurn scan_turn(ctx->device);
    if (!scans_may_interleave()) scan_turn.lock();
    if (n && (rc = launch_scan(ctx, a, hpc, has_q, has_n, n))) return fail(rc);
    if ((rc = quality_finish())) return fail(rc);                  // host work while the kernel runs (once: the call above may have done it)

    // overflow handling (reads that selected more than their padded capacity are re-run with exact room)
    // and the exact complexity pass over the few reads the 2-mer bound could not clear
    DevBuf<uint32_t> d_list, d_suspects, d_nlist;
    if ((rc = d_list.alloc(ctx, n)) || (rc = d_suspects.alloc(ctx, n)) || (rc = d_nlist.alloc(ctx, 2))) return fail(rc);
    (void)hipMemsetAsync(d_nlist.p, 0, 8, ctx->stream);
    if (n) hipLaunchKernelGGL(post_scan_lists_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream,
                              d_count.p, d_cap.p, m->d_flags.p, n, d_list.p, d_suspects.p, d_nlist.p);
    uint32_t h_counters[2] = {0, 0};
    e = memcpy_sync(ctx, h_counters, d_nlist.p, 8, hipMemcpyDeviceToHost);
    if (scan_turn.owns_lock()) scan_turn.unlock();
    if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "overflow count copy failed: %s", hipGetErrorString(e)));
    const uint32_t n_over = h_counters[0], n_suspect = h_counters[1];
    if (n_suspect) {
        LaunchTimer timer(ctx, "complexity_exact");
        hipLaunchKernelGGL(complexity_exact_kernel, dim3(grid_for((uint64_t)n_suspect * 64, 256, (unsigned)ctx->n_cu * 8u)), dim3(256), 0,
                           ctx->stream, reads->d_words.p, reads->d_word_off.p, reads->d_len.p, d_suspects.p, n_suspect,
                           d_count.p, m->d_flags.p, (unsigned long long *)nullptr);
    }
    if (any_low_quality) {
        DevBuf<uint8_t> d_low;
        if ((rc = d_low.alloc(ctx, n))) return fail(rc);
        if ((e = memcpy_sync(ctx, d_low.p, low_quality.data(), n, hipMemcpyHostToDevice)) != hipSuccess)
            return fail(set_error(ctx, MDBG_EHIP, "low-quality flags upload failed: %s", hipGetErrorString(e)));
        hipLaunchKernelGGL(apply_low_quality_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, d_low.p, n, d_count.p, m->d_flags.p, (unsigned long long *)nullptr);
        if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "low-quality pass failed"));
    }

    // dense offsets from the true counts
    if ((rc = exclusive_scan_u32(ctx, d_count.p, m->d_off.p, n))) return fail(rc);
    uint64_t total = 0;
    e = memcpy_sync(ctx, &total, m->d_off.p + n, 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "total copy failed"));
    m->n_min = total;
    if ((rc = m->d_min.alloc(ctx, total)) || (rc = m->d_pos.alloc(ctx, total)) || (rc = m->d_dir.alloc(ctx, total)) ||
        (rc = m->d_mqual.alloc(ctx, total)))
        return fail(rc);

    if (n) {
        const bool few = total / (n ? n : 1) < 96;   // minimizers per read
        unsigned blocks = grid_for((uint64_t)n * (few ? 16 : 64), 256, (unsigned)ctx->n_cu * (few ? 32u : 8u));
        LaunchTimer timer(ctx, "scan_compact");
#define MDBG_COMPACT(Q, G, ...) hipLaunchKernelGGL((compact_minimizers_kernel<Q, G>), dim3(blocks), dim3(256), 0, ctx->stream, __VA_ARGS__)
        if (has_q) {
            if (few) MDBG_COMPACT(true, 16, d_cap_off.p, m->d_off.p, n, p_min.p, p_pos.p, p_dir.p, p_os.p, p_oe.p, reads->d_qual.p, reads->d_qual_off.p,
                                  m->d_min.p, m->d_pos.p, m->d_dir.p, m->d_mqual.p);
            else MDBG_COMPACT(true, 64, d_cap_off.p, m->d_off.p, n, p_min.p, p_pos.p, p_dir.p, p_os.p, p_oe.p, reads->d_qual.p, reads->d_qual_off.p,
                              m->d_min.p, m->d_pos.p, m->d_dir.p, m->d_mqual.p);
        } else {
            if (few) MDBG_COMPACT(false, 16, d_cap_off.p, m->d_off.p, n, p_min.p, p_pos.p, p_dir.p, nullptr, nullptr, nullptr, nullptr,
                                  m->d_min.p, m->d_pos.p, m->d_dir.p, m->d_mqual.p);
            else MDBG_COMPACT(false, 64, d_cap_off.p, m->d_off.p, n, p_min.p, p_pos.p, p_dir.p, nullptr, nullptr, nullptr, nullptr,
                              m->d_min.p, m->d_pos.p, m->d_dir.p, m->d_mqual.p);
        }
#undef MDBG_COMPACT
    }
    if (n_over) {
        // reads that overflowed their padded slots are re-run straight into their dense slots
        // (capacity == exact count), after the gather so the exact rows win over the truncated prefix
        ScanArgs b = a;
        b.subset = d_list.p;
        b.cap_off = m->d_off.p;
        b.out_min = m->d_min.p; b.out_pos = m->d_pos.p; b.out_dir = m->d_dir.p;
        b.out_mqual = m->d_mqual.p; b.inline_minq = 1;
        DevBuf<uint32_t> scratch_count;
        DevBuf<uint8_t> scratch_flags;
        if ((rc = scratch_count.alloc(ctx, n)) || (rc = scratch_flags.alloc(ctx, n))) return fail(rc);
        b.out_count = scratch_count.p; b.out_flags = scratch_flags.p;
        if ((rc = launch_scan(ctx, b, hpc, has_q, has_n, n_over))) return fail(rc);
        (void)hipStreamSynchronize(ctx->stream);
    }
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "scan failed: %s", hipGetErrorString(e)));
    *out = m;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)
