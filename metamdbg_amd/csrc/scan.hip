// scan.hip -- reads -> minimizers on the device.
//
// One wavefront (64 lanes) owns one read at a time and walks it in tiles of 64 packed words
// (2048 bases).  Per tile:
//   1. every lane holds one u64 word (32 bases); run-start flags for homopolymer compression come
//      from a bit trick on the packed word (x ^ (x<<2 | prev base)), a wave prefix sum of their
//      popcounts gives each lane its offset in the compressed stream        [EncoderRLE, Commons.hpp:4163-4203]
//   2. lanes squeeze their kept bases to contiguous 2-bit fields and OR them into a per-wave LDS
//      bit stream (carry of the last K bases from the previous tile in front)
//   3. lanes take k-mers j = lane, lane+64, ...: one ds_read2 + v_alignbit extracts the K bases
//      LSB-first (E); revcomp = E ^ 0xAAAA.., forward = digit-reverse(E); canonical = min;
//      closed-form 8-byte Murmur3 (seed 42) and an integer threshold replace the double compare
//                                                           [KmerModel::iterate + MinimizerParser::parse,
//                                                            utils/kmer/Kmer.hpp:531-611, :1373-1456]
//   4. selected lanes are compacted in position order with ballot + popcount.
// A k-mer is evaluated only once the base after it is known, which drops the last k-mer of the
// read exactly as the reference's loop bound (pos < nK-1) does, for any homopolymer tail.
// Low-complexity reads (computeSequenceComplexity, ReadSelection.hpp:1171-1228) are detected with a
// 2-mer upper bound per 64-position window and confirmed by an exact 3-mer pass only when needed.
#include "common.hpp"
#include "murmur.hpp"
#include "objects.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace mdbg {

constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_WAVES = SCAN_BLOCK / 64;
constexpr int TILE_WORDS = 64;                 // u64 words per tile (one per lane)
constexpr int STREAM_WORDS = 136;              // u32: (16 carry + 2048 new bases) / 16 = 129, + slack for the w+1 read
constexpr uint64_t M5 = 0x5555555555555555ull;
constexpr uint8_t READ_SUSPECT = 0x80;         // internal: 2-mer complexity bound exceeded, exact pass pending

struct ScanArgs {
    const uint64_t *words;
    const uint64_t *word_off;
    const uint32_t *len;
    const uint32_t *invalid;      // per word mask or nullptr
    uint32_t n_reads;
    uint32_t K;
    uint64_t threshold;           // hash < threshold  <=>  (double)hash < density * 2^64
    const uint32_t *rep;          // sorted repetitive minimizers
    uint32_t n_rep;
    int apply_filters;
    const uint32_t *subset;       // optional list of read indices to process (overflow re-run)
    // outputs (padded per read: slots [cap_off[r], cap_off[r+1]))
    const uint64_t *cap_off;
    uint32_t *out_min;
    uint32_t *out_pos;
    uint8_t *out_dir;
    uint32_t *out_count;          // per read: number selected (may exceed capacity -> overflow)
    uint8_t *out_flags;           // per read: MDBG_READ_*
    uint32_t *work_counter;
};

// squeeze the 2-bit fields of x whose flag bit (bit 2i of d) is set down to the low end
__device__ __forceinline__ uint64_t compress_pairs(uint64_t x, uint64_t d) {
    uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
    uint32_t dl = (uint32_t)d, dh = (uint32_t)(d >> 32);
    uint64_t out = 0;
    unsigned n = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) {
        unsigned f = (dl >> (2 * i)) & 1u;
        uint64_t b = (uint64_t)(((xl >> (2 * i)) & 3u) & (0u - f));
        out |= b << (2 * n);
        n += f;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        unsigned f = (dh >> (2 * i)) & 1u;
        uint64_t b = (uint64_t)(((xh >> (2 * i)) & 3u) & (0u - f));
        out |= b << (2 * n);
        n += f;
    }
    return out;
}

// K bases starting at stream position j, LSB-first
__device__ __forceinline__ uint32_t stream_extract(const uint32_t *S, unsigned j, uint32_t kmask) {
    unsigned b = 2u * j, w = b >> 5, sh = b & 31u;
    uint32_t lo = S[w], hi = S[w + 1];
    return __builtin_amdgcn_alignbit(hi, lo, sh) & kmask;
}

// reverse the order of the K 2-bit digits of e
__device__ __forceinline__ uint32_t digit_reverse(uint32_t e, unsigned K) {
    uint32_t r = __builtin_bitreverse32(e);
    r = ((r & 0x55555555u) << 1) | ((r >> 1) & 0x55555555u);
    return r >> (32u - 2u * K);
}

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
                                                               const uint32_t *list, uint32_t n_list, uint32_t *count, uint8_t *flags) {
    const unsigned lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 256u + threadIdx.x) >> 6, n_waves = (gridDim.x * 256u) >> 6;
    for (uint32_t i = wave; i < n_list; i += n_waves) {
        const uint32_t r = list[i];
        int low = exact_low_complexity(words + word_off[r], len[r], lane);
        if (lane == 0) {
            flags[r] = low ? (uint8_t)MDBG_READ_LOW_COMPLEXITY : (uint8_t)0;
            if (low) count[r] = 0;
        }
    }
}

#ifndef SCAN_MIN_WAVES
#define SCAN_MIN_WAVES 1   // waves per SIMD the register allocator must allow (tuning knob)
#endif

template <bool HPC>
__global__ __launch_bounds__(SCAN_BLOCK, SCAN_MIN_WAVES) void scan_kernel(ScanArgs a) {
    __shared__ uint32_t lds_stream[SCAN_WAVES][STREAM_WORDS];
    const unsigned lane = threadIdx.x & 63u;
    uint32_t *S = lds_stream[threadIdx.x >> 6];
    const unsigned K = a.K;
    const uint32_t kmask = (K >= 16) ? 0xFFFFFFFFu : ((1u << (2 * K)) - 1u);
    const uint32_t comp_mask = 0xAAAAAAAAu & kmask;

    // reads are dealt round-robin to the resident waves (grid-stride); lengths are similar within a
    // batch, and the grid holds 8 waves per SIMD so a long read only delays its own wave
    const uint32_t wave_global = (blockIdx.x * SCAN_BLOCK + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * SCAN_BLOCK) >> 6;
    for (uint32_t slot = wave_global; slot < a.n_reads; slot += n_waves) {
        const uint32_t r = a.subset ? a.subset[slot] : slot;

        const uint32_t L = a.len[r];
        const uint64_t *rw = a.words + a.word_off[r];
        const uint32_t nwords = (L + 31u) / 32u;
        const uint32_t ntiles = (nwords + TILE_WORDS - 1) / TILE_WORDS;
        const uint64_t cap0 = a.cap_off[r];
        const uint32_t cap = (uint32_t)(a.cap_off[r + 1] - cap0);

        uint32_t carry = 0;        // last cb bases of the compressed stream, LSB-first
        unsigned cb = 0;
        uint32_t hp_total = 0;     // compressed bases seen so far
        uint32_t nout = 0;
        uint32_t prev_last = 0;    // last base of the previous tile's last word
        uint64_t cx_bound = 0;     // sum over windows of the 2-mer bound numerator

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

            // ---- complexity: 2-mer upper bound of the window starting at this word -------------
            if (a.apply_filters) {
                uint64_t x1 = __shfl_down(x, 1, 64), x2 = __shfl_down(x, 2, 64);
                uint64_t n0 = __shfl(x_next, 0, 64), n1 = __shfl(x_next, 1, 64);
                if (lane == 63) { x1 = n0; x2 = n1; }
                if (lane == 62) { x2 = n0; }
                uint64_t s = 0;
                if ((uint64_t)wi * 32u + 66u <= L) s = (window_sq_sum<2>(x, x1, x2) - 64u) / 2u;
                cx_bound += s;   // per lane partial; reduced at the end of the read
            }

            // ---- 1. run starts / compaction ---------------------------------------------------
            uint64_t y;
            unsigned c;
            if (HPC) {
                uint32_t pl = (uint32_t)__shfl_up((uint32_t)(x >> 62), 1, 64);
                if (lane == 0) pl = prev_last;
                uint64_t diff = x ^ ((x << 2) | (uint64_t)pl);
                uint64_t d = (diff | (diff >> 1)) & M5;
                if (wi == 0) d |= 1ull;
                d &= vspread;
                c = (unsigned)__popcll(d);
                y = compress_pairs(x, d);
                prev_last = (uint32_t)__shfl((uint32_t)(x >> 62), 63, 64);
            } else {
                c = nvalid;
                y = x & (vspread | (vspread << 1));
            }
            const unsigned inc = wave_inclusive_sum(c);
            const unsigned o = inc - c;
            const unsigned C = __shfl(inc, 63, 64);

            // ---- 2. LDS bit stream: [carry (cb bases)] [new C bases] ---------------------------
            S[lane] = 0;
            S[lane + 64] = 0;
            if (lane < STREAM_WORDS - 128) S[lane + 128] = 0;
            wave_lds_sync();
            if (lane == 0 && cb) atomicOr(&S[0], carry);
            if (c) {
                unsigned dst = 2u * (cb + o), w = dst >> 5, sh = dst & 31u;
                uint32_t yl = (uint32_t)y, yh = (uint32_t)(y >> 32);
                uint32_t p0 = yl << sh;
                uint32_t p1 = sh ? ((yl >> (32u - sh)) | (yh << sh)) : yh;
                uint32_t p2 = sh ? (yh >> (32u - sh)) : 0u;
                atomicOr(&S[w], p0);
                if (p1) atomicOr(&S[w + 1], p1);
                if (p2) atomicOr(&S[w + 2], p2);
            }
            wave_lds_sync();

            // ---- 3./4. k-mers, hash, select, compact -------------------------------------------
            const unsigned tot = cb + C;
            const unsigned nk = tot > K ? tot - K : 0u;
            const uint32_t hp_base = hp_total - cb;    // compressed-stream position of S base 0
            for (unsigned j0 = 0; j0 < nk; j0 += 64) {
                const unsigned j = j0 + lane;
                const uint32_t p = hp_base + j;
                bool sel = false;
                uint32_t val = 0, dir = 0;
                if (j < nk) {
                    uint32_t e = stream_extract(S, j, kmask);
                    uint32_t rev = e ^ comp_mask;
                    uint32_t fwd = digit_reverse(e, K);
                    dir = fwd < rev ? 0u : 1u;               // tie -> 1 (Kmer.hpp:427)
                    val = dir ? rev : fwd;
                    uint64_t h = kmer_hash32(val);
                    sel = (h < a.threshold) && (p >= 1u);     // first k-mer skipped (Kmer.hpp:1395)
                    if (sel && a.n_rep) sel = !rep_contains(a.rep, a.n_rep, val);   // Kmer.hpp:1437
                }
                unsigned long long bal = __ballot(sel);
                if (bal) {
                    if (sel) {
                        uint32_t idx = nout + (uint32_t)__popcll(bal & lanemask_lt());
                        if (idx < cap) {
                            a.out_min[cap0 + idx] = val;
                            a.out_pos[cap0 + idx] = p;
                            a.out_dir[cap0 + idx] = (uint8_t)dir;
                        }
                    }
                    nout += (uint32_t)__popcll(bal);
                }
            }

            // ---- carry the last min(tot, K) bases ----------------------------------------------
            const unsigned cbn = tot < K ? tot : K;
            const uint32_t cmask = (cbn >= 16) ? 0xFFFFFFFFu : ((1u << (2 * cbn)) - 1u);
            carry = cbn ? stream_extract(S, tot - cbn, cmask) : 0u;
            cb = cbn;
            hp_total += C;
            wave_lds_sync();   // all reads of S done before the next tile zeroes it
        }

        // ---- per-read epilogue -----------------------------------------------------------------
        uint8_t flags = 0;
        if (a.apply_filters && L >= 66) {
            uint32_t nW = (L - 66u) / 32u + 1u;
            uint64_t bound = wave_sum_u64(cx_bound);
            // bound/(61 nW) >= true mean score; only reads whose bound exceeds ~4.9 need the exact pass,
            // which runs as its own (rare) kernel so its registers do not cap this kernel's occupancy
            if (bound > 300ull * nW) flags |= READ_SUSPECT;
        }
        if (lane == 0) {
            a.out_count[r] = nout;
            a.out_flags[r] = flags;
        }
    }
}

// ---- padded -> dense CSR ------------------------------------------------------------------------
__global__ __launch_bounds__(256) void compact_minimizers_kernel(
    const uint64_t *cap_off, const uint64_t *off, uint32_t n_reads,
    const uint32_t *pmin, const uint32_t *ppos, const uint8_t *pdir,
    uint32_t *omin, uint32_t *opos, uint8_t *odir, uint8_t *oqual) {
    // one wave per read (reads hold a few dozen to a few thousand minimizers)
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t r = wave; r < n_reads; r += nwaves) {
        uint64_t src = cap_off[r], dst = off[r];
        uint32_t n = (uint32_t)(off[r + 1] - dst);
        uint32_t ncopy = (uint32_t)(cap_off[r + 1] - src);   // padded capacity (overflowed reads are re-run)
        if (ncopy > n) ncopy = n;
        for (uint32_t i = lane; i < n; i += 64) {
            if (i < ncopy) {
                omin[dst + i] = pmin[src + i];
                opos[dst + i] = ppos[src + i];
                odir[dst + i] = pdir[src + i];
            }
            oqual[dst + i] = 1;   // no qualities: ReadSelection.hpp:1047-1051
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

__global__ void gather_u32_kernel(const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t *dst) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[idx[i]];
}

}  // namespace mdbg

using namespace mdbg;

// host double threshold -> integer threshold (see oracle/mdbg_oracle.c orc_density_threshold)
static uint64_t density_threshold(float density) {
    const double bound = (double)density * 18446744073709551616.0;
    if (!((double)UINT64_MAX >= bound)) return UINT64_MAX;
    uint64_t lo = 0, hi = UINT64_MAX;
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if ((double)mid >= bound) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// ReadSelection.hpp:870-879 with an empty quality string: long double 0 / size_t 0, narrowed to
// float, through log10f -- evaluated at run time so the NaN carries the same sign bit (0xFFC00000
// on x86-64) the reference writes into read_data_init.txt.
static float mean_quality_without_qualities() {
    volatile long double error_sum = 0;
    volatile size_t n = 0;
    volatile float mean_err = (float)(error_sum / n);
    return -10.0f * log10f(mean_err);
}

static int launch_scan(mdbg_ctx *ctx, ScanArgs &a, bool hpc, uint32_t n_items) {
    MDBG_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_work_counter, 0, sizeof(uint32_t), ctx->stream));
    a.work_counter = ctx->d_work_counter;
    a.n_reads = n_items;
    // persistent-style grid: enough waves to fill every SIMD 8 deep, reads handed out dynamically
    unsigned blocks = (unsigned)ctx->n_cu * 8u;
    uint64_t need = ((uint64_t)n_items + SCAN_WAVES - 1) / SCAN_WAVES;
    if (need < blocks) blocks = (unsigned)(need ? need : 1);
    {
        LaunchTimer timer(ctx, "scan");
        if (hpc) hipLaunchKernelGGL(scan_kernel<true>, dim3(blocks), dim3(SCAN_BLOCK), 0, ctx->stream, a);
        else     hipLaunchKernelGGL(scan_kernel<false>, dim3(blocks), dim3(SCAN_BLOCK), 0, ctx->stream, a);
    }
    MDBG_HIP_CHECK(ctx, hipGetLastError());
    return MDBG_OK;
}

#define TRACE(msg) do { if (getenv("MDBG_TRACE")) { fprintf(stderr, "[mdbg_scan] %s\n", msg); fflush(stderr); } } while (0)

extern "C" int mdbg_scan(mdbg_ctx *ctx, const mdbg_reads *reads, const mdbg_scan_params *p, mdbg_minimizers **out) {
    if (!ctx || !reads || !p || !out) return set_error(ctx, MDBG_EINVAL, "mdbg_scan: null argument");
    if (p->minimizer_size < 2 || p->minimizer_size > 16)
        return set_error(ctx, MDBG_EINVAL, "mdbg_scan: minimizer_size %u outside [2,16]", p->minimizer_size);
    if (reads->has_invalid)
        return set_error(ctx, MDBG_ERANGE, "mdbg_scan: reads with N are not supported by this build yet");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t n = reads->n_reads;
    mdbg_minimizers *m = new mdbg_minimizers();
    m->n_reads = n;
    m->from_scan = true;
    auto fail = [&](int rc) { delete m; return rc; };

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
        hipError_t e = hipMemcpyAsync(d_rep.p, rep.data(), rep.size() * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "copy of repetitive set failed"));
        (void)hipStreamSynchronize(ctx->stream);
    }
    hipError_t e = hipMemcpyAsync(m->d_len.p, reads->d_len.p, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream);
    if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "len copy failed: %s", hipGetErrorString(e)));

    if (n) hipLaunchKernelGGL(capacity_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream,
                              reads->d_len.p, n, p->density, d_cap.p);
    TRACE("capacity launched");
    if ((rc = exclusive_scan_u32(ctx, d_cap.p, d_cap_off.p, n))) return fail(rc);
    TRACE("capacity scanned");
    uint64_t cap_total = 0;
    e = memcpy_sync(ctx, &cap_total, d_cap_off.p + n, 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "cap total copy failed"));

    DevBuf<uint32_t> p_min, p_pos;
    DevBuf<uint8_t> p_dir;
    if ((rc = p_min.alloc(ctx, cap_total)) || (rc = p_pos.alloc(ctx, cap_total)) || (rc = p_dir.alloc(ctx, cap_total)))
        return fail(rc);

    ScanArgs a{};
    a.words = reads->d_words.p; a.word_off = reads->d_word_off.p; a.len = reads->d_len.p;
    a.invalid = nullptr;
    a.K = p->minimizer_size;
    a.threshold = density_threshold(p->density);
    a.rep = d_rep.p; a.n_rep = p->n_repetitive;
    a.apply_filters = p->apply_read_filters;
    a.subset = nullptr;
    a.cap_off = d_cap_off.p;
    a.out_min = p_min.p; a.out_pos = p_pos.p; a.out_dir = p_dir.p;
    a.out_count = d_count.p; a.out_flags = m->d_flags.p;
    TRACE("launching scan kernel");
    if (n && (rc = launch_scan(ctx, a, p->hpc != 0, n))) return fail(rc);
    if (getenv("MDBG_TRACE")) { hipError_t se = hipStreamSynchronize(ctx->stream); fprintf(stderr, "[mdbg_scan] scan kernel done: %s\n", hipGetErrorString(se)); }

    // overflow handling (reads that selected more than their padded capacity are re-run with exact room)
    // and the exact complexity pass over the few reads the 2-mer bound could not clear
    DevBuf<uint32_t> d_list, d_suspects, d_nlist;
    if ((rc = d_list.alloc(ctx, n)) || (rc = d_suspects.alloc(ctx, n)) || (rc = d_nlist.alloc(ctx, 2))) return fail(rc);
    (void)hipMemsetAsync(d_nlist.p, 0, 8, ctx->stream);
    if (n) hipLaunchKernelGGL(post_scan_lists_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream,
                              d_count.p, d_cap.p, m->d_flags.p, n, d_list.p, d_suspects.p, d_nlist.p);
    uint32_t h_counters[2] = {0, 0};
    e = memcpy_sync(ctx, h_counters, d_nlist.p, 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "overflow count copy failed: %s", hipGetErrorString(e)));
    const uint32_t n_over = h_counters[0], n_suspect = h_counters[1];
    if (n_suspect) {
        LaunchTimer timer(ctx, "complexity_exact");
        hipLaunchKernelGGL(complexity_exact_kernel, dim3(grid_for((uint64_t)n_suspect * 64, 256, (unsigned)ctx->n_cu * 8u)), dim3(256), 0,
                           ctx->stream, reads->d_words.p, reads->d_word_off.p, reads->d_len.p, d_suspects.p, n_suspect,
                           d_count.p, m->d_flags.p);
    }
    TRACE("overflow list done");
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
        unsigned blocks = (unsigned)ctx->n_cu * 8u;
        LaunchTimer timer(ctx, "scan_compact");
        hipLaunchKernelGGL(compact_minimizers_kernel, dim3(blocks), dim3(256), 0, ctx->stream,
                           d_cap_off.p, m->d_off.p, n, p_min.p, p_pos.p, p_dir.p,
                           m->d_min.p, m->d_pos.p, m->d_dir.p, m->d_mqual.p);
    }
    if (n_over) {
        // reads that overflowed their padded slots are re-run straight into their dense slots
        // (capacity == exact count), after the gather so the exact rows win over the truncated prefix
        ScanArgs b = a;
        b.subset = d_list.p;
        b.cap_off = m->d_off.p;
        b.out_min = m->d_min.p; b.out_pos = m->d_pos.p; b.out_dir = m->d_dir.p;
        DevBuf<uint32_t> scratch_count;
        DevBuf<uint8_t> scratch_flags;
        if ((rc = scratch_count.alloc(ctx, n)) || (rc = scratch_flags.alloc(ctx, n))) return fail(rc);
        b.out_count = scratch_count.p; b.out_flags = scratch_flags.p;
        if ((rc = launch_scan(ctx, b, p->hpc != 0, n_over))) return fail(rc);
        (void)hipStreamSynchronize(ctx->stream);
    }
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "scan failed: %s", hipGetErrorString(e)));
    m->h_mean_quality.assign(n, mean_quality_without_qualities());
    *out = m;
    return MDBG_OK;
}
