// minimizers.hip -- minimizer-space sequence sets: host <-> device, palindrome purging
// (Commons::purgePalindrome, Commons.hpp:1617-1723) and the repetitive-minimizer census
// (ReadSelection::determineRepetitiveMinimizers, readSelection/ReadSelection.hpp:497-625).
#include "common.hpp"
#include "objects.hpp"
#include "murmur.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

namespace mdbg {

// KmerVec::isPalindrome (Commons.hpp:918-921) on a[i..i+k)
__device__ __forceinline__ bool window_is_palindrome(const uint32_t *a, uint32_t k) {
    for (uint32_t t = 0; t < k / 2; t++)
        if (a[t] != a[k - 1 - t]) return false;
    return true;
}

// A palindromic window of length k >= 2 needs m[c]==m[c+1] (even k) or m[c-1]==m[c+1] (odd k) at its
// centre, so reads without such a pair are untouched (the common case).  16 lanes per read look for
// one; suspects are listed for the serial pass.
// Where the minimizers of read r are: CSR (end = begin + 1 of the same array, cnt null) or the scattered form of a fresh scan
// output (begin[r], cnt[r]).
struct ReadSpans {
    const uint64_t *begin;
    const uint64_t *end;
    const uint32_t *cnt;
    __device__ __forceinline__ uint32_t n(uint64_t r) const { return cnt ? cnt[r] : (uint32_t)(end[r] - begin[r]); }
};

__global__ __launch_bounds__(256) void purge_detect_kernel(ReadSpans sp, uint32_t n_reads, const uint32_t *mins,
                                                           uint32_t *new_count, uint32_t *list, uint32_t *n_list) {
    const unsigned sub = threadIdx.x & 15u;
    const unsigned gshift = (threadIdx.x & 63u) & ~15u;           // first lane of this 16-lane group in the wave
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (uint64_t r0 = 0; r0 < n_reads; r0 += ngroups) {
        const uint64_t r = r0 + group;
        const bool live = r < n_reads;
        const uint64_t f = live ? sp.begin[r] : 0;
        const uint32_t n = live ? sp.n(r) : 0u;
        bool suspect = false;
        for (uint32_t i = sub; i + 1 < n; i += 16) {
            uint32_t x = mins[f + i];
            suspect |= (x == mins[f + i + 1]) || (i + 2 < n && x == mins[f + i + 2]);
        }
        unsigned long long bal = __ballot(suspect);
        bool any = ((bal >> gshift) & 0xFFFFull) != 0ull;
        if (live && sub == 0) {
            new_count[r] = n;
            if (any) list[atomicAdd(n_list, 1u)] = (uint32_t)r;
        }
    }
}

// Serial replay of the reference for the listed reads: smallest k, then smallest i whose window of k
// surviving minimizers is a palindrome -> drop its first element -> restart, until none is left.
// Dropping from a compacted copy is equivalent to the reference's banned-position bookkeeping.
// Replay of Commons::purgePalindrome on one read: the reference looks, for k = first_k .. last_k-1 and then for
// i = 0 .. n-k, for the first palindromic window, drops its first element and starts over (Commons.hpp:1617-1723).
// A window is a palindrome iff it is centred on a palindromic centre whose radius covers it, and with a centre of
// radius R every shorter window of the same parity about it is one too.  So the first hit is, over all centres
// (between two equal neighbours, or on an element whose two neighbours are equal), the lexicographically smallest
// (k0, start) with k0 the smallest length >= first_k of the centre's parity that its radius allows: one linear pass
// with a short expansion per centre instead of O(n * k) window tests per drop.
template <typename Ptr>
__device__ __forceinline__ uint32_t purge_replay(Ptr a, uint32_t n, uint32_t first_k, uint32_t last_k) {
    const uint32_t k_even = first_k + (first_k & 1u), k_odd = first_k | 1u;     // smallest allowed length of each parity
    for (;;) {
        uint32_t best_k = 0xFFFFFFFFu, best_i = 0;
        for (uint32_t c = 0; c + 1 < n; c++) {
            // even centre between c and c+1: window of length k_even starts at c + 1 - k_even / 2
            if (a[c] == a[c + 1] && k_even < last_k && k_even <= n && k_even < best_k) {
                const uint32_t half = k_even / 2;
                if (c + 1 >= half && c + half < n) {
                    uint32_t r = 1;
                    while (r < half && a[c - r] == a[c + 1 + r]) r++;
                    if (r == half) { best_k = k_even; best_i = c + 1 - half; }
                }
            }
            // odd centre on c+1 (neighbours c and c+2): window of length k_odd starts at c + 1 - k_odd / 2
            if (c + 2 < n && a[c] == a[c + 2] && k_odd < last_k && k_odd <= n && k_odd < best_k) {
                const uint32_t half = k_odd / 2;              // pairs to match around the centre
                if (c + 1 >= half && c + 1 + half < n) {
                    uint32_t r = 1;
                    while (r < half && a[c - r] == a[c + 2 + r]) r++;
                    if (r == half) { best_k = k_odd; best_i = c + 1 - half; }
                }
            }
        }
        if (best_k == 0xFFFFFFFFu) break;
        for (uint32_t j = best_i; j + 1 < n; j++) a[j] = a[j + 1];
        n--;
    }
    return n;
}

// One thread per suspect read.  The replay is a long chain of dependent reads of the same few dozen minimizers, so
// reads of up to PURGE_LDS_MAX minimizers are staged in LDS (a padded row per thread) and written back once.
// Sixteen threads a block, 6 KB of LDS: a block must fit into what another context's scan leaves of a CU (five scan blocks hold 150
// of the 160 KB; with 64 threads and 24 KB this kernel sat out whole scan launches -- 0.08 ms of work, 103 ms from start to end,
// rocprofv3 trace of round 3 -- and its batch's table pass behind it).
constexpr uint32_t PURGE_LDS_MAX = 96;
constexpr uint32_t PURGE_FIX_THREADS = 16;
// The rows of a suspect read are read from the input and written -- purged -- to the same place of `work`, and the read is marked in
// `fixed`: `work` holds nothing else (round 4: it used to be a copy of the whole input, 1.5 GB and 2.3 ms per 10 M reads, made for the
// hundred thousand reads this kernel looks at).
__global__ __launch_bounds__(PURGE_FIX_THREADS) void purge_fix_kernel(ReadSpans sp, const uint32_t *list, uint32_t n_list, const uint32_t *src, uint32_t *work,
                                                                      uint8_t *fixed, uint32_t first_k, uint32_t last_k, uint32_t *new_count) {
    __shared__ uint32_t stage[PURGE_FIX_THREADS][PURGE_LDS_MAX + 1];
    uint32_t li = blockIdx.x * blockDim.x + threadIdx.x;
    if (li >= n_list) return;
    const uint32_t r = list[li];
    const uint32_t *in = src + sp.begin[r];
    uint32_t *a = work + sp.begin[r];
    uint32_t n = sp.n(r);
    if (n <= PURGE_LDS_MAX) {
        uint32_t *row = stage[threadIdx.x];
        for (uint32_t i = 0; i < n; i++) row[i] = in[i];
        n = purge_replay(row, n, first_k, last_k);
        for (uint32_t i = 0; i < n; i++) a[i] = row[i];
    } else {
        for (uint32_t i = 0; i < n; i++) a[i] = in[i];
        n = purge_replay(a, n, first_k, last_k);
    }
    fixed[r] = 1;
    new_count[r] = n;
}

// 16 lanes per read: a chain of dependent loads per read, so reads in flight are what it runs on
// (fixed, alt: the reads marked in `fixed` come from `alt` -- the purged rows -- instead of `src`; both null: all from `src`)
__global__ __launch_bounds__(256) void gather_prefix_kernel(const uint64_t *src_off, const uint64_t *dst_off, uint32_t n_reads,
                                                            const uint32_t *src, uint32_t *dst, const uint8_t *fixed = nullptr, const uint32_t *alt = nullptr) {
    const unsigned lane = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (uint64_t r = group; r < n_reads; r += ngroups) {
        uint64_t s = src_off[r], d = dst_off[r];
        uint32_t n = (uint32_t)(dst_off[r + 1] - d);
        const uint32_t *from = (fixed && fixed[r]) ? alt : src;
        for (uint32_t i = lane; i < n; i += 16) dst[d + i] = from[s + i];
    }
}

// scattered scan output -> CSR order: values, positions, directions of every read (16 lanes per read)
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint64_t *src_begin, const uint64_t *dst_off, uint32_t n_reads,
                                                          const uint32_t *smin, const uint32_t *spos, const uint8_t *sdir, const uint8_t *sq,
                                                          uint32_t *dmin, uint32_t *dpos, uint8_t *ddir, uint8_t *dq) {
    const unsigned lane = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (uint64_t r = group; r < n_reads; r += ngroups) {
        const uint64_t s = src_begin[r], d = dst_off[r];
        const uint32_t n = (uint32_t)(dst_off[r + 1] - d);
        for (uint32_t i = lane; i < n; i += 16) { dmin[d + i] = smin[s + i]; dpos[d + i] = spos[s + i]; ddir[d + i] = sdir[s + i]; dq[d + i] = sq[s + i]; }
    }
}

int ensure_canonical(mdbg_ctx *ctx, const mdbg_minimizers *cm) {
    if (!cm) return MDBG_OK;
    // the conversion rewrites the object's arrays: one caller does it, the others wait and find it done (a read-only consumer on
    // another thread -- a census on one context while another purges -- must not see the arrays half swapped)
    std::lock_guard<std::mutex> once(cm->canon_mu);
    if (!cm->scattered) return MDBG_OK;
    mdbg_minimizers *m = const_cast<mdbg_minimizers *>(cm);
    mdbg_ctx *c = m->owner ? m->owner : ctx;            // on the stream that produced the rows
    MDBG_HIP_CHECK(ctx, hipSetDevice(c->device));
    const uint32_t n = m->n_reads;
    MDBG_TRY(m->d_off.alloc(c, (size_t)n + 1));
    MDBG_TRY(exclusive_scan_u32(c, m->d_cnt.p, m->d_off.p, n));
    DevBuf<uint32_t> nmin, npos;
    DevBuf<uint8_t> ndir, nq;
    MDBG_TRY(nmin.alloc(c, m->n_min));
    MDBG_TRY(npos.alloc(c, m->n_min));
    MDBG_TRY(ndir.alloc(c, m->n_min));
    MDBG_TRY(nq.alloc(c, m->n_min));
    if (n) {
        LaunchTimer timer(c, "scan_compact");
        unsigned blocks = grid_for((uint64_t)n * 16, 256, (unsigned)c->n_cu * 32u);
        hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, c->stream, m->d_begin.p, m->d_off.p, n, m->d_min.p, m->d_pos.p,
                           m->d_dir.p, m->d_mqual.p, nmin.p, npos.p, ndir.p, nq.p);
    }
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(c->stream));   // the old rows go back to the pool; other contexts may read the object next
    m->d_min = std::move(nmin);
    m->d_pos = std::move(npos);
    m->d_dir = std::move(ndir);
    m->d_mqual = std::move(nq);
    m->d_begin.release();
    m->d_cnt.release();
    m->scattered = false;
    return MDBG_OK;
}

// ---- Utils::applyDensityThreshold (Commons.hpp:2507-2550) ----------------------------------------
__global__ __launch_bounds__(256) void density_flag_kernel(const uint32_t *mins, uint64_t n, uint64_t threshold, uint32_t *flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) flag[i] = kmer_hash32(mins[i]) < threshold ? 1u : 0u;
}

__global__ __launch_bounds__(256) void density_offsets_kernel(const uint64_t *off, uint32_t n_reads, const uint64_t *pos, uint64_t *new_off) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r <= n_reads) new_off[r] = pos[off[r]];
}

__global__ __launch_bounds__(256) void density_compact_kernel(uint64_t n, const uint32_t *flag, const uint64_t *pos, const uint32_t *mins,
                                                              const uint32_t *mpos, const uint8_t *dir, const uint8_t *qual,
                                                              uint32_t *omin, uint32_t *opos, uint8_t *odir, uint8_t *oqual) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n || !flag[i]) return;
    const uint64_t d = pos[i];
    omin[d] = mins[i];
    if (mpos) { opos[d] = mpos[i]; odir[d] = dir[i]; oqual[d] = qual[i]; }
}

// ---- u32 value census: ONE COUNT PER POSSIBLE VALUE.  A minimizer is a canonical l-mer, l <= 16: its value is below 4^l, so the
// table of counts is simply indexed by it -- 4 GiB at l = 15, 16 GiB at l = 16, of 288 -- and a value costs one device-scope add
// without a return.  (Rounds 1 - 5 kept an open-addressing table of 8-byte keys and counts, two random accesses per value and
// 12.9 GB to clear and sweep at the ONT census: 38.8 ms for 5 x 10^8 values; this form: the rate of random adds, 26 G/s.)
__global__ __launch_bounds__(256) void census_max_kernel(const uint32_t *vals, uint64_t n, uint32_t *out) {
    uint32_t mine = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) mine = vals[i] > mine ? vals[i] : mine;
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(mine, d, 64); mine = o > mine ? o : mine; }
    if ((threadIdx.x & 63u) == 0u && mine) atomicMax(out, mine);
}

__global__ __launch_bounds__(256) void census_count_kernel(const uint32_t *vals, uint64_t n, uint32_t *counts) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&counts[vals[i]], 1u);
}

// The same over a FRESH scan output (mdbg_minimizers::scattered: the rows of read r at [begin[r], begin[r] + cnt[r]), slack between
// the regions): a census asks for the values, not their order -- bringing 5 x 10^8 rows into CSR order first was 10 GB of copies.
// 16 lanes per read.
__global__ __launch_bounds__(256) void census_max_scattered_kernel(const uint64_t *begin, const uint32_t *cnt, uint32_t n_reads, const uint32_t *vals, uint32_t *out) {
    const unsigned lane = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    uint32_t mine = 0;
    for (uint64_t r = group; r < n_reads; r += ngroups) {
        const uint64_t s = begin[r];
        for (uint32_t i = lane, n = cnt[r]; i < n; i += 16) mine = vals[s + i] > mine ? vals[s + i] : mine;
    }
    for (int d = 32; d >= 1; d >>= 1) { const uint32_t o = __shfl_xor(mine, d, 64); mine = o > mine ? o : mine; }
    if ((threadIdx.x & 63u) == 0u && mine) atomicMax(out, mine);
}

__global__ __launch_bounds__(256) void census_count_scattered_kernel(const uint64_t *begin, const uint32_t *cnt, uint32_t n_reads, const uint32_t *vals, uint32_t *counts) {
    const unsigned lane = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4, ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    for (uint64_t r = group; r < n_reads; r += ngroups) {
        const uint64_t s = begin[r];
        for (uint32_t i = lane, n = cnt[r]; i < n; i += 16) atomicAdd(&counts[vals[s + i]], 1u);
    }
}

// How many distinct values were seen 1, 2, ... CENSUS_BINS-1 (or more) times: the cut-off count of the top fraction is
// read off this histogram, so only the values at or above it travel to the host (tens to hundreds of 13 M distinct values
// at the ONT census; downloading and partially sorting all of them took longer than counting them).  `cap` is a multiple of 4:
// sixteen bytes a lane, and nearly all of them zero.
constexpr uint32_t CENSUS_BINS = 4096;
__global__ __launch_bounds__(256) void census_hist_kernel(const uint32_t *counts, uint64_t cap, unsigned long long *hist) {
    __shared__ uint32_t h[CENSUS_BINS];
    for (uint32_t i = threadIdx.x; i < CENSUS_BINS; i += 256) h[i] = 0;
    __syncthreads();
    const uint4 *c4 = reinterpret_cast<const uint4 *>(counts);
    for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < cap / 4; s += (uint64_t)gridDim.x * 256) {
        const uint4 v = c4[s];
        if ((v.x | v.y | v.z | v.w) == 0u) continue;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) if (w[j]) atomicAdd(&h[w[j] < CENSUS_BINS - 1u ? w[j] : CENSUS_BINS - 1u], 1u);
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < CENSUS_BINS; i += 256) if (h[i]) atomicAdd(&hist[i], (unsigned long long)h[i]);
}

__global__ __launch_bounds__(256) void census_top_kernel(const uint32_t *counts, uint64_t cap, uint32_t cut,
                                                         uint32_t *out_val, uint32_t *out_cnt, unsigned long long *cursor, uint64_t room) {
    const uint4 *c4 = reinterpret_cast<const uint4 *>(counts);
    for (uint64_t s = (uint64_t)blockIdx.x * 256 + threadIdx.x; s < cap / 4; s += (uint64_t)gridDim.x * 256) {
        const uint4 v = c4[s];
        if ((v.x | v.y | v.z | v.w) == 0u) continue;
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            if (w[j] >= cut && w[j]) {
                const unsigned long long at = atomicAdd(cursor, 1ull);
                if (at < room) { out_val[at] = (uint32_t)(4 * s + (uint64_t)j); out_cnt[at] = w[j]; }
            }
        }
    }
}

}  // namespace mdbg

using namespace mdbg;

extern "C" int mdbg_minimizers_info(const mdbg_minimizers *m, uint32_t *n_reads, uint64_t *n_minimizers) {
    if (!m) return MDBG_EINVAL;
    if (n_reads) *n_reads = m->n_reads;
    if (n_minimizers) *n_minimizers = m->n_min;
    return MDBG_OK;
}

extern "C" int mdbg_minimizers_to_host(mdbg_ctx *ctx, const mdbg_minimizers *m, uint64_t *offsets,
                                       uint32_t *minimizers, uint32_t *positions, uint8_t *directions, uint8_t *qualities,
                                       uint32_t *read_lengths, float *mean_quality, uint8_t *read_flags) try {
    if (!ctx || !m) return set_error(ctx, MDBG_EINVAL, "mdbg_minimizers_to_host: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    MDBG_TRY(ensure_canonical(ctx, m));
    const size_t n = m->n_reads, t = m->n_min;
    if ((positions || directions || qualities || read_lengths || mean_quality || read_flags) && !m->from_scan)
        return set_error(ctx, MDBG_EINVAL, "mdbg_minimizers_to_host: positions/directions/qualities exist only for mdbg_scan output");
    // every array queued, ONE wait: a feeder's batch of a few thousand reads is eight small copies, and eight round trips of the
    // stream used to cost more than the bytes (0.39 s of a 1.9 s pass over 50 Gbp); into page-locked memory they are plain DMA
    auto dl = [&](void *dst, const void *src, size_t bytes) { return bytes ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream) : hipSuccess; };
    if (offsets) MDBG_HIP_CHECK(ctx, dl(offsets, m->d_off.p, (n + 1) * 8));
    if (minimizers) MDBG_HIP_CHECK(ctx, dl(minimizers, m->d_min.p, t * 4));
    if (positions) MDBG_HIP_CHECK(ctx, dl(positions, m->d_pos.p, t * 4));
    if (directions) MDBG_HIP_CHECK(ctx, dl(directions, m->d_dir.p, t));
    if (qualities) MDBG_HIP_CHECK(ctx, dl(qualities, m->d_mqual.p, t));
    if (read_lengths) MDBG_HIP_CHECK(ctx, dl(read_lengths, m->d_len.p, n * 4));
    if (read_flags) MDBG_HIP_CHECK(ctx, dl(read_flags, m->d_flags.p, n));
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (mean_quality && n) {
        if (m->h_mean_quality.size() == n) memcpy(mean_quality, m->h_mean_quality.data(), n * sizeof(float));
        else for (size_t i = 0; i < n; i++) mean_quality[i] = m->mean_quality_all;
    }
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_minimizers_from_host(mdbg_ctx *ctx, const uint32_t *minimizers, const uint64_t *offsets,
                                         uint32_t n_reads, mdbg_minimizers **out) try {
    if (!ctx || !out || !offsets) return set_error(ctx, MDBG_EINVAL, "mdbg_minimizers_from_host: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    mdbg_minimizers *m = new mdbg_minimizers();
    auto fail = [&](int rc) { delete m; return rc; };
    m->n_reads = n_reads;
    std::vector<uint64_t> rel((size_t)n_reads + 1);
    for (size_t i = 0; i <= n_reads; i++) {
        if (i && offsets[i] < offsets[i - 1]) return fail(set_error(ctx, MDBG_EINVAL, "offsets must be non-decreasing"));
        rel[i] = offsets[i] - offsets[0];
    }
    m->n_min = rel[n_reads];
    if (m->n_min && !minimizers) return fail(set_error(ctx, MDBG_EINVAL, "mdbg_minimizers_from_host: null minimizers"));
    int rc;
    if ((rc = m->d_off.alloc(ctx, rel.size())) || (rc = m->d_min.alloc(ctx, m->n_min))) return fail(rc);
    hipError_t e = memcpy_sync(ctx, m->d_off.p, rel.data(), rel.size() * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess && m->n_min) e = memcpy_sync(ctx, m->d_min.p, minimizers + offsets[0], m->n_min * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "upload failed: %s", hipGetErrorString(e)));
    *out = m;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

// ---- a file's bytes on the device (mdbg_bytes_*) and the records taken apart there --------------------------------------------------
extern "C" int mdbg_bytes_create(mdbg_ctx *ctx, uint64_t n_bytes, mdbg_bytes **out) try {
    if (!ctx || !out) return set_error(ctx, MDBG_EINVAL, "mdbg_bytes_create: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<mdbg_bytes> b(new mdbg_bytes());
    b->n = n_bytes;
    b->owner = ctx;
    MDBG_TRY(b->d.alloc(ctx, n_bytes + 16));            // (the gather reads whole aligned words around a record's values)
    if (!ctx->upload_stream) MDBG_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
    // the block comes from the pool, whose reuse is ordered by the context's stream: the first upload starts after what that stream has queued so far
    hipEvent_t ev = nullptr;
    MDBG_HIP_CHECK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->upload_stream, ev, 0);
    (void)hipEventDestroy(ev);
    if (e != hipSuccess) return set_error(ctx, MDBG_EHIP, "mdbg_bytes_create: %s", hipGetErrorString(e));
    *out = b.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_bytes_upload_async(mdbg_ctx *ctx, mdbg_bytes *b, uint64_t at, const void *host, uint64_t n, uint64_t *ticket) try {
    if (!ctx || !b || (n && !host)) return set_error(ctx, MDBG_EINVAL, "mdbg_bytes_upload_async: null argument");
    if (at > b->n || n > b->n - at) return set_error(ctx, MDBG_EINVAL, "mdbg_bytes_upload_async: [%llu, +%llu) lies outside the %llu bytes of the buffer",
                                                     (unsigned long long)at, (unsigned long long)n, (unsigned long long)b->n);
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipEvent_t ev = nullptr;
    MDBG_HIP_CHECK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    std::lock_guard<std::mutex> g(b->mu);               // several feeder threads: copy and event stay together
    hipError_t e = n ? hipMemcpyAsync(b->d.p + at, host, n, hipMemcpyHostToDevice, ctx->upload_stream) : hipSuccess;
    if (e == hipSuccess) e = hipEventRecord(ev, ctx->upload_stream);
    if (e != hipSuccess) { (void)hipEventDestroy(ev); return set_error(ctx, MDBG_EHIP, "mdbg_bytes_upload_async: %s", hipGetErrorString(e)); }
    b->events.push_back(ev);
    if (ticket) *ticket = b->events.size();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_bytes_upload_done(mdbg_ctx *ctx, mdbg_bytes *b, uint64_t ticket, int wait) try {
    if (!ctx || !b) return set_error(ctx, MDBG_EINVAL, "mdbg_bytes_upload_done: null argument");
    hipEvent_t ev = nullptr;
    {
        std::lock_guard<std::mutex> g(b->mu);
        if (ticket == 0 || ticket <= b->done_upto) return 1;
        if (ticket > b->events.size()) return set_error(ctx, MDBG_EINVAL, "mdbg_bytes_upload_done: no such ticket");
        ev = b->events[ticket - 1];
    }
    const hipError_t e = wait ? hipEventSynchronize(ev) : hipEventQuery(ev);
    if (e == hipErrorNotReady) return 0;
    if (e != hipSuccess) return set_error(ctx, MDBG_EHIP, "mdbg_bytes_upload_done: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> g(b->mu);
    if (ticket > b->done_upto) b->done_upto = ticket;
    return 1;
} MDBG_API_CATCH(ctx)

extern "C" void mdbg_bytes_free(mdbg_bytes *b) { delete b; }

namespace mdbg {
// the kernels `ctx` queues next see every piece uploaded so far
int bytes_ready_on(mdbg_ctx *ctx, const mdbg_bytes *b) {
    mdbg_bytes *mb = const_cast<mdbg_bytes *>(b);
    std::lock_guard<std::mutex> g(mb->mu);
    if (!mb->events.empty()) MDBG_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, mb->events.back(), 0));
    return MDBG_OK;
}
}

// Record r of a `u32 n; u8 circular; u32 m[n]` file starts at byte 5 r + 4 off[r]; its values sit 5 bytes further: at a misalignment
// of (r + 1) mod 4 that is the same for the whole record, so a value is two aligned words and one v_alignbyte.  16 lanes a record.
__global__ __launch_bounds__(256) void records_gather_kernel(const uint8_t *raw, uint64_t n_bytes, const uint64_t *off, uint32_t n_reads, uint32_t *mins,
                                                             uint8_t *circ, uint32_t *bad) {
    const unsigned sub = threadIdx.x & 15u;
    const uint64_t group = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const uint64_t ngroups = ((uint64_t)gridDim.x * blockDim.x) >> 4;
    const uint32_t *words = reinterpret_cast<const uint32_t *>(raw);
    for (uint64_t r = group; r < n_reads; r += ngroups) {
        const uint64_t o0 = off[r], n = off[r + 1] - o0;
        const uint64_t at = 5 * r + 4 * o0;                         // the record's first byte
        if (at + 5 + 4 * n > n_bytes) { if (sub == 0) atomicExch(bad, 1u); continue; }
        if (sub == 0) {
            uint32_t cnt = 0;
            for (int j = 0; j < 4; j++) cnt |= (uint32_t)raw[at + j] << (8 * j);
            if (cnt != (uint32_t)n || n >> 32) atomicExch(bad, 1u);
            if (circ) circ[r] = raw[at + 4];
        }
        const uint64_t v0 = at + 5;
        const uint32_t sh = (uint32_t)(v0 & 3u);
        const uint64_t w0 = v0 >> 2;
        for (uint64_t j = sub; j < n; j += 16) {
            const uint32_t a = words[w0 + j];
            const uint32_t b = sh ? words[w0 + j + 1] : 0u;
            mins[o0 + j] = sh ? __builtin_amdgcn_alignbyte(b, a, sh) : a;
        }
    }
}

extern "C" int mdbg_minimizers_from_record_bytes(mdbg_ctx *ctx, const mdbg_bytes *records, const uint64_t *offsets, uint32_t n_reads,
                                                 uint8_t *circular, mdbg_minimizers **out) try {
    if (!ctx || !records || !offsets || !out) return set_error(ctx, MDBG_EINVAL, "mdbg_minimizers_from_record_bytes: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    for (size_t i = 1; i <= n_reads; i++)
        if (offsets[i] < offsets[i - 1]) return set_error(ctx, MDBG_EINVAL, "mdbg_minimizers_from_record_bytes: offsets must be non-decreasing");
    if (offsets[0] != 0) return set_error(ctx, MDBG_EINVAL, "mdbg_minimizers_from_record_bytes: offsets[0] must be 0");
    const uint64_t n_min = offsets[n_reads];
    if (5ull * n_reads + 4ull * n_min != records->n)
        return set_error(ctx, MDBG_EINVAL, "mdbg_minimizers_from_record_bytes: %u records of %llu minimizers are %llu bytes, the buffer holds %llu",
                         n_reads, (unsigned long long)n_min, (unsigned long long)(5ull * n_reads + 4ull * n_min), (unsigned long long)records->n);
    std::unique_ptr<mdbg_minimizers> m(new mdbg_minimizers());
    m->n_reads = n_reads;
    m->n_min = n_min;
    MDBG_TRY(m->d_off.alloc(ctx, (size_t)n_reads + 1));
    MDBG_TRY(m->d_min.alloc(ctx, n_min));
    DevBuf<uint8_t> d_circ;
    DevBuf<uint32_t> d_bad;
    if (circular) MDBG_TRY(d_circ.alloc(ctx, n_reads));
    MDBG_TRY(d_bad.alloc(ctx, 1));
    MDBG_HIP_CHECK(ctx, hipMemsetAsync(d_bad.p, 0, 4, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipMemcpyAsync(m->d_off.p, offsets, ((size_t)n_reads + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
    MDBG_TRY(bytes_ready_on(ctx, records));
    if (n_reads)
        hipLaunchKernelGGL(records_gather_kernel, dim3(grid_for((uint64_t)n_reads * 16, 256, (unsigned)ctx->n_cu * 16)), dim3(256), 0, ctx->stream,
                           records->d.p, records->n, m->d_off.p, n_reads, m->d_min.p, circular ? d_circ.p : nullptr, d_bad.p);
    uint32_t bad = 0;
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &bad, d_bad.p, 4, hipMemcpyDeviceToHost));
    if (circular && n_reads) MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, circular, d_circ.p, n_reads, hipMemcpyDeviceToHost));
    if (bad) return set_error(ctx, MDBG_EINVAL, "mdbg_minimizers_from_record_bytes: a record's count does not agree with the offsets (not this file's bytes?)");
    *out = m.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_minimizers_device_ptrs(const mdbg_minimizers *m, const uint64_t **d_offsets, const uint32_t **d_minimizers) {
    if (!m) return MDBG_EINVAL;
    if (m->scattered) { int rc = ensure_canonical(m->owner, m); if (rc) return rc; }
    if (d_offsets) *d_offsets = m->d_off.p;
    if (d_minimizers) *d_minimizers = m->d_min.p;
    return MDBG_OK;
}

extern "C" void mdbg_minimizers_free(mdbg_minimizers *m) { delete m; }

// dst[i] = src[i] + base for i < n (the offsets of an appended part, rebased)
__global__ __launch_bounds__(256) void rebase_offsets_kernel(const uint64_t *src, uint64_t n, uint64_t base, uint64_t *dst) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] + base;
}

extern "C" int mdbg_minimizers_concat(mdbg_ctx *ctx, const mdbg_minimizers *const *parts, uint32_t n_parts, mdbg_minimizers **out) try {
    if (!ctx || !out || (n_parts && !parts)) return set_error(ctx, MDBG_EINVAL, "mdbg_minimizers_concat: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    uint64_t n_reads = 0, n_min = 0;
    bool side = n_parts > 0, per_read = n_parts > 0, any_mq = false;
    for (uint32_t p = 0; p < n_parts; p++) {
        if (!parts[p]) return set_error(ctx, MDBG_EINVAL, "mdbg_minimizers_concat: part %u is null", p);
        MDBG_TRY(ensure_canonical(ctx, parts[p]));
        n_reads += parts[p]->n_reads; n_min += parts[p]->n_min;
        side = side && parts[p]->from_scan && parts[p]->d_pos.p && parts[p]->d_dir.p && parts[p]->d_mqual.p;
        per_read = per_read && parts[p]->from_scan && parts[p]->d_len.p && parts[p]->d_flags.p;
        any_mq = any_mq || !parts[p]->h_mean_quality.empty();
    }
    if (n_reads >= (1ull << 32)) return set_error(ctx, MDBG_ERANGE, "mdbg_minimizers_concat: more than 2^32 reads");
    std::unique_ptr<mdbg_minimizers> m(new mdbg_minimizers());
    m->n_reads = (uint32_t)n_reads;
    m->n_min = n_min;
    m->from_scan = side && per_read;          // positions, directions, qualities and the per-read fields follow only when every part has them
    m->owner = ctx;
    MDBG_TRY(m->d_off.alloc(ctx, n_reads + 1));
    MDBG_TRY(m->d_min.alloc(ctx, n_min));
    if (m->from_scan) {
        MDBG_TRY(m->d_pos.alloc(ctx, n_min)); MDBG_TRY(m->d_dir.alloc(ctx, n_min)); MDBG_TRY(m->d_mqual.alloc(ctx, n_min));
        MDBG_TRY(m->d_len.alloc(ctx, n_reads)); MDBG_TRY(m->d_flags.alloc(ctx, n_reads));
        if (any_mq) m->h_mean_quality.reserve(n_reads);
    }
    uint64_t r0 = 0, m0 = 0;
    for (uint32_t p = 0; p < n_parts; p++) {
        const mdbg_minimizers *q = parts[p];
        // a part another context produced: its rows are complete once that context's stream is (ensure_canonical and the scans
        // synchronise before they return, so the data is there; the copies below are ordered on THIS context's stream)
        const uint64_t nr = q->n_reads, nm = q->n_min;
        // offsets [0, nr) of the part rebased; the closing offset is written by the next part or below
        if (nr) hipLaunchKernelGGL(rebase_offsets_kernel, dim3(grid_for(nr, 256)), dim3(256), 0, ctx->stream, q->d_off.p, nr, m0, m->d_off.p + r0);
        if (nm) {
            MDBG_HIP_CHECK(ctx, hipMemcpyAsync(m->d_min.p + m0, q->d_min.p, nm * 4, hipMemcpyDeviceToDevice, ctx->stream));
            if (m->from_scan) {
                MDBG_HIP_CHECK(ctx, hipMemcpyAsync(m->d_pos.p + m0, q->d_pos.p, nm * 4, hipMemcpyDeviceToDevice, ctx->stream));
                MDBG_HIP_CHECK(ctx, hipMemcpyAsync(m->d_dir.p + m0, q->d_dir.p, nm, hipMemcpyDeviceToDevice, ctx->stream));
                MDBG_HIP_CHECK(ctx, hipMemcpyAsync(m->d_mqual.p + m0, q->d_mqual.p, nm, hipMemcpyDeviceToDevice, ctx->stream));
            }
        }
        if (m->from_scan && nr) {
            MDBG_HIP_CHECK(ctx, hipMemcpyAsync(m->d_len.p + r0, q->d_len.p, nr * 4, hipMemcpyDeviceToDevice, ctx->stream));
            MDBG_HIP_CHECK(ctx, hipMemcpyAsync(m->d_flags.p + r0, q->d_flags.p, nr, hipMemcpyDeviceToDevice, ctx->stream));
            if (any_mq) {
                if (q->h_mean_quality.size() == nr) m->h_mean_quality.insert(m->h_mean_quality.end(), q->h_mean_quality.begin(), q->h_mean_quality.end());
                else m->h_mean_quality.insert(m->h_mean_quality.end(), nr, q->mean_quality_all);
            }
        }
        if (p == 0) m->mean_quality_all = q->mean_quality_all;
        r0 += nr; m0 += nm;
    }
    MDBG_HIP_CHECK(ctx, hipMemcpyAsync(m->d_off.p + n_reads, &n_min, 8, hipMemcpyHostToDevice, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));      // n_min is a stack word; the parts may be freed by the caller next
    *out = m.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_minimizers_slice(mdbg_ctx *ctx, const mdbg_minimizers *in, uint32_t first_read, uint32_t n_reads, mdbg_minimizers **out) try {
    if (!ctx || !in || !out) return set_error(ctx, MDBG_EINVAL, "mdbg_minimizers_slice: null argument");
    if (first_read > in->n_reads || n_reads > in->n_reads - first_read)
        return set_error(ctx, MDBG_EINVAL, "mdbg_minimizers_slice: reads [%u, +%u) of %u", first_read, n_reads, in->n_reads);
    MDBG_TRY(ensure_canonical(ctx, in));
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    uint64_t ends[2] = {0, 0};
    MDBG_HIP_CHECK(ctx, hipMemcpyAsync(&ends[0], in->d_off.p + first_read, 8, hipMemcpyDeviceToHost, ctx->stream));
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &ends[1], in->d_off.p + first_read + n_reads, 8, hipMemcpyDeviceToHost));
    std::unique_ptr<mdbg_minimizers> m(new mdbg_minimizers());
    m->n_reads = n_reads;
    m->n_min = ends[1] - ends[0];
    m->owner = ctx;
    MDBG_TRY(m->d_off.alloc(ctx, (size_t)n_reads + 1));
    MDBG_TRY(m->d_min.alloc(ctx, m->n_min));
    hipLaunchKernelGGL(rebase_offsets_kernel, dim3(grid_for((uint64_t)n_reads + 1, 256)), dim3(256), 0, ctx->stream, in->d_off.p + first_read, (uint64_t)n_reads + 1,
                       (uint64_t)0 - ends[0], m->d_off.p);
    if (m->n_min) MDBG_HIP_CHECK(ctx, hipMemcpyAsync(m->d_min.p, in->d_min.p + ends[0], m->n_min * 4, hipMemcpyDeviceToDevice, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *out = m.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_apply_density_threshold(mdbg_ctx *ctx, const mdbg_minimizers *in, float density, mdbg_minimizers **out) try {
    if (!ctx || !in || !out) return set_error(ctx, MDBG_EINVAL, "mdbg_apply_density_threshold: null argument");
    MDBG_TRY(ensure_canonical(ctx, in));
    if (!(density > 0.0f)) return set_error(ctx, MDBG_EINVAL, "mdbg_apply_density_threshold: density must be > 0");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t n = in->n_reads;
    const uint64_t total = in->n_min;
    std::unique_ptr<mdbg_minimizers> m(new mdbg_minimizers());
    m->n_reads = n;
    m->from_scan = in->from_scan;
    m->h_mean_quality = in->h_mean_quality;
    m->mean_quality_all = in->mean_quality_all;
    DevBuf<uint32_t> flag;
    DevBuf<uint64_t> pos;
    MDBG_TRY(flag.alloc(ctx, total));
    MDBG_TRY(pos.alloc(ctx, total + 1));
    MDBG_TRY(m->d_off.alloc(ctx, (size_t)n + 1));
    if (total) {
        LaunchTimer timer(ctx, "density_threshold");
        hipLaunchKernelGGL(density_flag_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, in->d_min.p, total,
                           density_threshold(density), flag.p);
    }
    MDBG_TRY(exclusive_scan_u32(ctx, flag.p, pos.p, total));
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &m->n_min, pos.p + total, 8, hipMemcpyDeviceToHost));
    MDBG_TRY(m->d_min.alloc(ctx, m->n_min));
    const bool side = in->from_scan && in->d_pos.p;
    if (side) {
        MDBG_TRY(m->d_pos.alloc(ctx, m->n_min));
        MDBG_TRY(m->d_dir.alloc(ctx, m->n_min));
        MDBG_TRY(m->d_mqual.alloc(ctx, m->n_min));
    }
    if (in->d_len.p) {
        MDBG_TRY(m->d_len.alloc(ctx, n));
        MDBG_HIP_CHECK(ctx, hipMemcpyAsync(m->d_len.p, in->d_len.p, (size_t)n * 4, hipMemcpyDeviceToDevice, ctx->stream));
    }
    if (in->d_flags.p) {
        MDBG_TRY(m->d_flags.alloc(ctx, n));
        MDBG_HIP_CHECK(ctx, hipMemcpyAsync(m->d_flags.p, in->d_flags.p, (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    }
    {
        LaunchTimer timer(ctx, "density_threshold");
        hipLaunchKernelGGL(density_offsets_kernel, dim3(grid_for((uint64_t)n + 1, 256)), dim3(256), 0, ctx->stream, in->d_off.p, n, pos.p, m->d_off.p);
        if (total)
            hipLaunchKernelGGL(density_compact_kernel, dim3(grid_for(total, 256)), dim3(256), 0, ctx->stream, total, flag.p, pos.p, in->d_min.p,
                               side ? in->d_pos.p : nullptr, in->d_dir.p, in->d_mqual.p, m->d_min.p, m->d_pos.p, m->d_dir.p, m->d_mqual.p);
    }
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *out = m.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_purge_palindromes(mdbg_ctx *ctx, const mdbg_minimizers *in, uint32_t first_k, uint32_t last_k,
                                      mdbg_minimizers **out) try {
    if (!ctx || !in || !out || first_k < 2) return set_error(ctx, MDBG_EINVAL, "mdbg_purge_palindromes: bad argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // this call reads the input in whatever form it is in: nobody converts it to CSR order under its kernels (ensure_canonical
    // takes the same lock; every path out of this function synchronises the stream first or has launched nothing)
    std::lock_guard<std::mutex> as_it_is(in->canon_mu);
    const uint32_t n = in->n_reads;
    DevBuf<uint32_t> cnt, list, n_list;
    MDBG_TRY(cnt.alloc(ctx, n));
    MDBG_TRY(list.alloc(ctx, n));
    MDBG_TRY(n_list.alloc(ctx, 1));
    MDBG_HIP_CHECK(ctx, hipMemsetAsync(n_list.p, 0, 4, ctx->stream));
    mdbg_minimizers *m = new mdbg_minimizers();
    auto fail = [&](int rc) { delete m; return rc; };
    m->n_reads = n;
    int rc;
    if ((rc = m->d_off.alloc(ctx, (size_t)n + 1))) return fail(rc);
    // the input as it is: CSR, or the scattered rows of a fresh scan output (which this step then puts in CSR order: the copy it
    // makes anyway is a gather)
    const ReadSpans sp = in->scattered ? ReadSpans{in->d_begin.p, nullptr, in->d_cnt.p} : ReadSpans{in->d_off.p, in->d_off.p + 1, nullptr};
    if (n) {
        LaunchTimer timer(ctx, "purge_palindromes");
        unsigned blocks = grid_for((uint64_t)n * 16, 256, (unsigned)ctx->n_cu * 16u);
        hipLaunchKernelGGL(purge_detect_kernel, dim3(blocks), dim3(256), 0, ctx->stream, sp, n, in->d_min.p, cnt.p, list.p, n_list.p);
    }
    uint32_t n_suspect = 0;
    hipError_t e = memcpy_sync(ctx, &n_suspect, n_list.p, 4, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "purge suspect count copy failed: %s", hipGetErrorString(e)));
    const unsigned gblocks = grid_for((uint64_t)n * 16, 256, (unsigned)ctx->n_cu * 32u);
    if (n_suspect == 0) {
        // nothing to purge: the output is a copy of the input
        m->n_min = in->n_min;
        if ((rc = m->d_min.alloc(ctx, m->n_min))) return fail(rc);
        if (in->scattered) {
            if ((rc = exclusive_scan_u32(ctx, cnt.p, m->d_off.p, n))) return fail(rc);
            LaunchTimer timer(ctx, "purge_palindromes");
            if (n) hipLaunchKernelGGL(gather_prefix_kernel, dim3(gblocks), dim3(256), 0, ctx->stream, in->d_begin.p, m->d_off.p, n, in->d_min.p, m->d_min.p);
        } else {
            LaunchTimer timer(ctx, "purge_palindromes");
            e = hipMemcpyAsync(m->d_off.p, in->d_off.p, ((size_t)n + 1) * 8, hipMemcpyDeviceToDevice, ctx->stream);
            if (e == hipSuccess && m->n_min) e = hipMemcpyAsync(m->d_min.p, in->d_min.p, m->n_min * 4, hipMemcpyDeviceToDevice, ctx->stream);
            if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "purge copy failed: %s", hipGetErrorString(e)));
        }
    } else {
        // room for the purged rows of the suspect reads, in the input's own layout (a scattered input occupies n_rows entries, rows of dropped
        // reads included); nothing else of it is ever written or read
        const size_t work_n = in->scattered ? (size_t)in->n_rows : (size_t)in->n_min;
        DevBuf<uint32_t> work;
        DevBuf<uint8_t> fixed;
        if ((rc = work.alloc(ctx, work_n)) || (rc = fixed.alloc(ctx, n))) return fail(rc);
        {
            LaunchTimer timer(ctx, "purge_palindromes");
            e = hipMemsetAsync(fixed.p, 0, n, ctx->stream);
            if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "purge memset failed: %s", hipGetErrorString(e)));
            hipLaunchKernelGGL(purge_fix_kernel, dim3(grid_for(n_suspect, PURGE_FIX_THREADS)), dim3(PURGE_FIX_THREADS), 0, ctx->stream, sp, list.p, n_suspect,
                               in->d_min.p, work.p, fixed.p, first_k, last_k, cnt.p);
        }
        if ((rc = exclusive_scan_u32(ctx, cnt.p, m->d_off.p, n))) return fail(rc);
        e = memcpy_sync(ctx, &m->n_min, m->d_off.p + n, 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "purge total copy failed: %s", hipGetErrorString(e)));
        if (getenv("MDBG_TRACE")) fprintf(stderr, "[mdbg] purge: %u suspect reads of %u, %llu minimizers dropped\n", n_suspect, n,
                                          (unsigned long long)(in->n_min - m->n_min));
        if ((rc = m->d_min.alloc(ctx, m->n_min))) return fail(rc);
        {
            LaunchTimer timer(ctx, "purge_palindromes");
            hipLaunchKernelGGL(gather_prefix_kernel, dim3(gblocks), dim3(256), 0, ctx->stream, sp.begin, m->d_off.p, n, in->d_min.p, m->d_min.p, fixed.p, work.p);
        }
        e = hipStreamSynchronize(ctx->stream);   // `work` goes back to the pool on return
        if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "purge failed: %s", hipGetErrorString(e)));
    }
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "purge failed: %s", hipGetErrorString(e)));
    *out = m;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

// The census as an object: determineRepetitiveMinimizers counts the minimizers of the first million reads of every input file
// (ReadSelection.hpp:497-561); the reads arrive batch by batch, and every batch's values are counted where they are instead of
// travelling to the host and back.  counts[v] for every v below `cap`, a power of two above the largest value seen so far
// (a batch with a larger one makes the array grow: the counts so far keep their places).
struct mdbg_census {
    DevBuf<uint32_t> counts;
    uint64_t cap = 0;
};

extern "C" int mdbg_census_create(mdbg_ctx *ctx, mdbg_census **out) try {
    if (!ctx || !out) return set_error(ctx, MDBG_EINVAL, "mdbg_census_create: null argument");
    *out = new mdbg_census();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" void mdbg_census_free(mdbg_census *c) { delete c; }

extern "C" int mdbg_census_add(mdbg_ctx *ctx, mdbg_census *c, const mdbg_minimizers *m) try {
    if (!ctx || !c || !m) return set_error(ctx, MDBG_EINVAL, "mdbg_census_add: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint64_t n = m->n_min;
    if (!n) return MDBG_OK;
    // a fresh scan output is counted where it lies (census_count_scattered_kernel); nobody may bring it into CSR order meanwhile
    // (ensure_canonical swaps the arrays under this lock and frees the old rows)
    std::unique_lock<std::mutex> rows(m->canon_mu);
    const bool scattered = m->scattered;
    if (!scattered) rows.unlock();
    const unsigned sweep_max = (unsigned)ctx->n_cu * 8u, per_read_blocks = grid_for((uint64_t)m->n_reads * 16, 256, (unsigned)ctx->n_cu * 32u);
    DevBuf<uint32_t> d_max;
    MDBG_TRY(d_max.alloc(ctx, 1));
    MDBG_HIP_CHECK(ctx, hipMemsetAsync(d_max.p, 0, 4, ctx->stream));
    {
        LaunchTimer timer(ctx, "minimizer_census");
        if (scattered) hipLaunchKernelGGL(census_max_scattered_kernel, dim3(per_read_blocks), dim3(256), 0, ctx->stream, m->d_begin.p, m->d_cnt.p, m->n_reads, m->d_min.p, d_max.p);
        else hipLaunchKernelGGL(census_max_kernel, dim3(grid_for(n, 256, sweep_max)), dim3(256), 0, ctx->stream, m->d_min.p, n, d_max.p);
    }
    uint32_t vmax = 0;
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &vmax, d_max.p, 4, hipMemcpyDeviceToHost));
    uint64_t want = 1024;
    while (want <= (uint64_t)vmax) want <<= 1;
    if (want > c->cap) {
        DevBuf<uint32_t> counts;
        MDBG_TRY(counts.alloc(ctx, want));
        if (c->cap) MDBG_HIP_CHECK(ctx, hipMemcpyAsync(counts.p, c->counts.p, c->cap * 4, hipMemcpyDeviceToDevice, ctx->stream));
        MDBG_HIP_CHECK(ctx, hipMemsetAsync(counts.p + c->cap, 0, (want - c->cap) * 4, ctx->stream));
        if (c->cap) MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));      // the old array goes back to the pool below
        c->counts = std::move(counts);
        c->cap = want;
    }
    {
        LaunchTimer timer(ctx, "minimizer_census");
        if (scattered) hipLaunchKernelGGL(census_count_scattered_kernel, dim3(per_read_blocks), dim3(256), 0, ctx->stream, m->d_begin.p, m->d_cnt.p, m->n_reads, m->d_min.p, c->counts.p);
        else hipLaunchKernelGGL(census_count_kernel, dim3(grid_for(n, 256)), dim3(256), 0, ctx->stream, m->d_min.p, n, c->counts.p);
    }
    MDBG_HIP_CHECK(ctx, hipGetLastError());
    if (scattered) MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));      // the rows were read: the lock goes
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_census_top(mdbg_ctx *ctx, const mdbg_census *c, uint32_t *out, uint32_t *n_out) try {
    if (!ctx || !c || !out || !n_out) return set_error(ctx, MDBG_EINVAL, "mdbg_census_top: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (!c->cap) { *n_out = 0; return MDBG_OK; }
    const uint64_t cap = c->cap;
    DevBuf<unsigned long long> hist;
    DevBuf<uint32_t> oval, ocnt;
    MDBG_TRY(hist.alloc(ctx, CENSUS_BINS + 1));               // + the cursor of the second pass
    MDBG_HIP_CHECK(ctx, hipMemsetAsync(hist.p, 0, (CENSUS_BINS + 1) * 8, ctx->stream));
    const unsigned sweep_blocks = grid_for(cap / 4, 256, (unsigned)ctx->n_cu * 16u);
    std::vector<unsigned long long> h_hist(CENSUS_BINS);
    {
        LaunchTimer timer(ctx, "minimizer_census");
        hipLaunchKernelGGL(census_hist_kernel, dim3(sweep_blocks), dim3(256), 0, ctx->stream, c->counts.p, cap, hist.p);
    }
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, h_hist.data(), hist.p, CENSUS_BINS * 8, hipMemcpyDeviceToHost));
    uint64_t distinct = 0;
    for (unsigned long long v : h_hist) distinct += v;
    // the top max(1, fraction * distinct) values by count (ReadSelection.hpp:515-540)
    float fraction = 0.00001f;
    int keep = (int)(fraction * (float)distinct);
    if (keep < 1) keep = 1;
    if ((uint64_t)keep > distinct) keep = (int)distinct;
    // smallest count such that the values seen at least that often are `keep` or more: everything above it is in, the ties at it
    // are ordered on the host.  (The last bin holds every count >= CENSUS_BINS - 1: a cut there takes them all.)
    uint32_t cut = CENSUS_BINS - 1u;
    uint64_t n_top = h_hist[cut];
    while (cut > 1u && n_top < (uint64_t)keep) n_top += h_hist[--cut];
    std::vector<uint32_t> hv(n_top), hc(n_top);
    if (n_top) {
        MDBG_TRY(oval.alloc(ctx, n_top));
        MDBG_TRY(ocnt.alloc(ctx, n_top));
        {
            LaunchTimer timer(ctx, "minimizer_census");
            hipLaunchKernelGGL(census_top_kernel, dim3(sweep_blocks), dim3(256), 0, ctx->stream, c->counts.p, cap, cut, oval.p, ocnt.p,
                               hist.p + CENSUS_BINS, n_top);
        }
        MDBG_HIP_CHECK(ctx, hipMemcpyAsync(hv.data(), oval.p, n_top * 4, hipMemcpyDeviceToHost, ctx->stream));
        MDBG_HIP_CHECK(ctx, hipMemcpyAsync(hc.data(), ocnt.p, n_top * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::vector<uint64_t> order(n_top);
    for (uint64_t i = 0; i < n_top; i++) order[i] = i;
    std::partial_sort(order.begin(), order.begin() + keep, order.end(), [&](uint64_t a, uint64_t b) {
        if (hc[a] != hc[b]) return hc[a] > hc[b];
        return hv[a] < hv[b];
    });
    if ((uint32_t)keep > *n_out) return set_error(ctx, MDBG_ERANGE, "mdbg_census_top: need room for %d values", keep);
    for (int i = 0; i < keep; i++) out[i] = hv[order[i]];
    *n_out = (uint32_t)keep;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_repetitive_minimizers(mdbg_ctx *ctx, const mdbg_minimizers *m, uint32_t *out, uint32_t *n_out) try {
    if (!ctx || !m || !out || !n_out) return set_error(ctx, MDBG_EINVAL, "mdbg_repetitive_minimizers: null argument");
    mdbg_census c;
    MDBG_TRY(mdbg_census_add(ctx, &c, m));
    return mdbg_census_top(ctx, &c, out, n_out);
} MDBG_API_CATCH(ctx)
