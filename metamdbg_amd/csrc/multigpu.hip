// multigpu.hip -- first-pass k-min-mer counting when reads are sharded over several GPUs
// (one process per GPU).  The library never moves bytes between GPUs: it produces partial-count
// rows grouped by owner rank, the caller exchanges them (RCCL all-to-all over xGMI in bench.py,
// gloo in the CPU tests), and the library reduces / finishes.  Nearest reference analogue: the
// on-disk hash partitioning `vecHash % _nbPartitions` of KminmerCounter (graph/CreateMdbg.hpp:3714-3724).
//
// Row layout (u64 words): [hash_lo, hash_hi, count, vec01, vec23, ...]  -- 3 + ceil(k/2) words;
// the canonical vector rides along because the owner of a key may hold no read containing it and
// kminmerData_min.txt needs the vector of every solid key.
#include "common.hpp"
#include "murmur.hpp"
#include "objects.hpp"
#include "table.hpp"

namespace mdbg {

// shared with kminmer.hip (duplicated small device helpers keep the TUs independent, no RDC)
__device__ __forceinline__ uint32_t mg_find_read(const uint64_t *inst_off, uint32_t n_reads, uint64_t g) {
    uint32_t lo = 0, hi = n_reads;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (inst_off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ bool mg_window_hash(const uint32_t *m, uint32_t k, uint64_t &hi, uint64_t &lo) {
    bool reversed = true;
    for (uint32_t i = 0; i < k; i++) {
        uint32_t a = m[i], b = m[k - 1 - i];
        if (a == b) continue;
        reversed = !(a < b);
        break;
    }
    Murmur128Stream h;
    if (reversed) for (uint32_t i = 0; i < k; i++) h.push(m[k - 1 - i]);
    else          for (uint32_t i = 0; i < k; i++) h.push(m[i]);
    h.finish(hi, lo);
    return reversed;
}

__host__ __device__ __forceinline__ uint32_t row_words_for(uint32_t k) { return 3u + (k + 1u) / 2u; }

__device__ __forceinline__ uint32_t owner_of(uint64_t hi, uint32_t n_ranks) {
    return (uint32_t)(((hi >> 32) * (uint64_t)n_ranks) >> 32);
}

__global__ void mg_inst_count_kernel(const uint64_t *off, uint32_t n_reads, uint32_t k, uint32_t *cnt) {
    uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n_reads) {
        uint64_t n = off[r + 1] - off[r];
        cnt[r] = n >= k ? (uint32_t)(n - k + 1) : 0u;
    }
}

__global__ __launch_bounds__(256) void mg_count_insert_kernel(const uint32_t *mins, const uint64_t *off, const uint64_t *inst_off,
                                                              uint32_t n_reads, uint64_t n_inst, uint32_t k, TableView t) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_inst) return;
    uint32_t r = mg_find_read(inst_off, n_reads, g);
    const uint32_t *m = mins + off[r] + (g - inst_off[r]);
    uint64_t hi, lo;
    mg_window_hash(m, k, hi, lo);
    if (lo == 0ull || hi == 0ull) { table_exc_upsert(t, lo, hi, 1u, 0, false, (uint32_t)g, true); return; }
    uint32_t s = table_find_or_insert(t, lo, hi, true);
    if (s != SLOT_NONE) { atomicAdd(&t.slots[s].val, 1u); t.slots[s].rep = (uint32_t)g; }
}

__device__ __forceinline__ bool slot_read(const TableView &t, uint64_t cap, uint64_t s, uint64_t &lo, uint64_t &hi, uint32_t &v, uint32_t &rep) {
    if (s < cap) {
        const TableSlot &sl = t.slots[s];
        lo = sl.lo;
        if (lo == 0ull) return false;
        hi = sl.hi; v = sl.val; rep = sl.rep;
        return true;
    }
    uint32_t i = (uint32_t)(s - cap);
    if (i >= *t.exc_n) return false;
    lo = t.exc_lo[i]; hi = t.exc_hi[i]; v = t.exc_val[i]; rep = t.exc_rep[i];
    return true;
}

__global__ __launch_bounds__(256) void owner_hist_kernel(TableView t, uint64_t cap, uint32_t n_ranks, unsigned long long *hist) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap + TABLE_EXC_CAP) return;
    uint64_t lo, hi; uint32_t v, rep;
    if (slot_read(t, cap, s, lo, hi, v, rep)) atomicAdd(&hist[owner_of(hi, n_ranks)], 1ull);
}

__global__ __launch_bounds__(256) void owner_scatter_kernel(TableView t, uint64_t cap, uint32_t n_ranks, unsigned long long *cursor,
                                                            const uint32_t *mins, const uint64_t *off, const uint64_t *inst_off,
                                                            uint32_t n_reads, uint32_t k, uint64_t *rows) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap + TABLE_EXC_CAP) return;
    uint64_t lo, hi; uint32_t v, rep;
    if (!slot_read(t, cap, s, lo, hi, v, rep)) return;
    const uint32_t rw = row_words_for(k);
    uint64_t row = atomicAdd(&cursor[owner_of(hi, n_ranks)], 1ull);
    uint64_t *o = rows + row * rw;
    o[0] = lo; o[1] = hi; o[2] = v;
    uint32_t r = mg_find_read(inst_off, n_reads, rep);
    const uint32_t *m = mins + off[r] + (rep - inst_off[r]);
    bool reversed = true;
    for (uint32_t i = 0; i < k; i++) {
        uint32_t a = m[i], b = m[k - 1 - i];
        if (a == b) continue;
        reversed = !(a < b);
        break;
    }
    for (uint32_t w = 0; w < (k + 1) / 2; w++) {
        uint32_t i0 = 2 * w, i1 = 2 * w + 1;
        uint64_t a = reversed ? m[k - 1 - i0] : m[i0];
        uint64_t b = i1 < k ? (reversed ? m[k - 1 - i1] : m[i1]) : 0u;
        o[3 + w] = a | (b << 32);
    }
}

// insert rows into a table: val += count, rep = row index (so the vector can be fetched back)
__global__ __launch_bounds__(256) void rows_add_kernel(const uint64_t *rows, uint64_t n, uint32_t rw, TableView t) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t *r = rows + i * rw;
    uint64_t lo = r[0], hi = r[1];
    uint32_t c = (uint32_t)r[2];
    if (lo == 0ull || hi == 0ull) { table_exc_upsert(t, lo, hi, c, 0, false, (uint32_t)i, true); return; }
    uint32_t s = table_find_or_insert(t, lo, hi, true);
    if (s != SLOT_NONE) { atomicAdd(&t.slots[s].val, c); t.slots[s].rep = (uint32_t)i; }
}

__global__ __launch_bounds__(256) void mg_flag_kernel(TableView t, uint64_t cap, uint32_t min_abundance, int solid_only,
                                                      uint32_t rank, uint32_t n_ranks, uint32_t *flag) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap + TABLE_EXC_CAP) return;
    uint64_t lo, hi; uint32_t v, rep;
    bool keep = false;
    if (slot_read(t, cap, s, lo, hi, v, rep)) {
        keep = true;
        if (solid_only) keep = v > 1u && !(v < min_abundance) && owner_of(hi, n_ranks) == rank;
    }
    flag[s] = keep ? 1u : 0u;
}

// compact table slots back into rows (vector copied from the representative source row)
__global__ __launch_bounds__(256) void mg_emit_rows_kernel(TableView t, uint64_t cap, const uint32_t *flag, const uint64_t *pos,
                                                           const uint64_t *src_rows, uint32_t rw, uint64_t *dst_rows) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap + TABLE_EXC_CAP || !flag[s]) return;
    uint64_t lo, hi; uint32_t v, rep;
    slot_read(t, cap, s, lo, hi, v, rep);
    uint64_t *o = dst_rows + pos[s] * rw;
    const uint64_t *src = src_rows + (uint64_t)rep * rw;
    o[0] = lo; o[1] = hi; o[2] = v;
    for (uint32_t w = 3; w < rw; w++) o[w] = src[w];
}

__global__ __launch_bounds__(256) void mg_emit_solid_kernel(TableView t, uint64_t cap, const uint32_t *flag, const uint64_t *pos,
                                                            const uint64_t *src_rows, uint32_t rw, uint32_t k,
                                                            uint64_t *olo, uint64_t *ohi, uint32_t *oab, uint32_t *ovec) {
    uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= cap + TABLE_EXC_CAP || !flag[s]) return;
    uint64_t lo, hi; uint32_t v, rep;
    slot_read(t, cap, s, lo, hi, v, rep);
    uint64_t row = pos[s];
    olo[row] = lo; ohi[row] = hi; oab[row] = v;
    const uint64_t *src = src_rows + (uint64_t)rep * rw + 3;
    for (uint32_t i = 0; i < k; i++) ovec[row * k + i] = (uint32_t)(src[i / 2] >> (32 * (i & 1)));
}

// per local instance: global abundance if solid else 0
__global__ __launch_bounds__(256) void mg_inst_abundance_kernel(const uint32_t *mins, const uint64_t *off, const uint64_t *inst_off,
                                                                uint32_t n_reads, uint64_t n_inst, uint32_t k, TableView t,
                                                                uint32_t min_abundance, uint32_t *ab) {
    uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_inst) return;
    uint32_t r = mg_find_read(inst_off, n_reads, g);
    const uint32_t *m = mins + off[r] + (g - inst_off[r]);
    uint64_t hi, lo;
    mg_window_hash(m, k, hi, lo);
    uint32_t v = 0;
    bool found = table_lookup(t, lo, hi, v);
    bool solid = found && v > 1u && !(v < min_abundance);
    ab[g] = solid ? v : 0u;
}

}  // namespace mdbg

using namespace mdbg;

// kernels defined in kminmer.hip that this TU reuses through tiny host wrappers
namespace mdbg {
int mg_rescue_rows(mdbg_ctx *ctx, const mdbg_minimizers *reads, const uint64_t *inst_off, uint64_t n_inst, uint32_t k,
                   const uint32_t *ab, mdbg_table *t, uint64_t n_solid);
}

static int mg_inst_index(mdbg_ctx *ctx, const mdbg_minimizers *m, uint32_t k, DevBuf<uint64_t> &off, uint64_t &total) {
    DevBuf<uint32_t> cnt;
    MDBG_TRY(cnt.alloc(ctx, m->n_reads));
    MDBG_TRY(off.alloc(ctx, (size_t)m->n_reads + 1));
    if (m->n_reads)
        hipLaunchKernelGGL(mg_inst_count_kernel, dim3(grid_for(m->n_reads, 256)), dim3(256), 0, ctx->stream, m->d_off.p, m->n_reads, k, cnt.p);
    MDBG_TRY(exclusive_scan_u32(ctx, cnt.p, off.p, m->n_reads));
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &total, off.p + m->n_reads, 8, hipMemcpyDeviceToHost));
    return MDBG_OK;
}

extern "C" uint32_t mdbg_row_words(uint32_t k) { return row_words_for(k); }

extern "C" int mdbg_kminmer_partial_counts(mdbg_ctx *ctx, const mdbg_minimizers *reads, uint32_t k, uint32_t n_ranks,
                                           const uint64_t **d_rows, uint64_t *counts) {
    if (!ctx || !reads || !d_rows || !counts || k < 2 || n_ranks < 1 || n_ranks > 64)
        return set_error(ctx, MDBG_EINVAL, "mdbg_kminmer_partial_counts: bad argument");
    if (reads->n_min >= (1ull << 32)) return set_error(ctx, MDBG_ERANGE, "more than 2^32 minimizers in one batch");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    DevBuf<uint64_t> inst_off;
    uint64_t I = 0;
    MDBG_TRY(mg_inst_index(ctx, reads, k, inst_off, I));
    DeviceTable tab;
    MDBG_TRY(build_table_adaptive(ctx, tab, (uint64_t)((double)I * ctx->key_ratio_hint), I, [&](TableView v) {
        if (I) {
            LaunchTimer timer(ctx, "kminmer_insert");
            hipLaunchKernelGGL(mg_count_insert_kernel, dim3(grid_for(I, 256)), dim3(256), 0, ctx->stream, reads->d_min.p, reads->d_off.p,
                               inst_off.p, reads->n_reads, I, k, v);
        }
        return MDBG_OK;
    }));
    if (I) ctx->key_ratio_hint = (double)tab.cap / 2.0 / (double)I;
    TableView tv = tab.view();
    const uint64_t nslots = tab.cap + TABLE_EXC_CAP;
    DevBuf<unsigned long long> hist;
    MDBG_TRY(hist.alloc(ctx, 64));
    MDBG_HIP_CHECK(ctx, hipMemsetAsync(hist.p, 0, 64 * 8, ctx->stream));
    hipLaunchKernelGGL(owner_hist_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, n_ranks, hist.p);
    unsigned long long h[64];
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, h, hist.p, 64 * 8, hipMemcpyDeviceToHost));
    unsigned long long cur[64];
    uint64_t total = 0;
    for (uint32_t r = 0; r < 64; r++) { cur[r] = total; if (r < n_ranks) { counts[r] = h[r]; total += h[r]; } }
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, hist.p, cur, 64 * 8, hipMemcpyHostToDevice));
    if (ctx->partial_rows) { (void)hipFree(ctx->partial_rows); ctx->partial_rows = nullptr; }
    const uint32_t rw = row_words_for(k);
    hipError_t e = hipMalloc(&ctx->partial_rows, (total ? total : 1) * rw * 8);
    if (e != hipSuccess) return set_error(ctx, MDBG_ENOMEM, "partial rows allocation failed: %s", hipGetErrorString(e));
    {
        LaunchTimer timer(ctx, "kminmer_emit");
        hipLaunchKernelGGL(owner_scatter_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, n_ranks, hist.p,
                           reads->d_min.p, reads->d_off.p, inst_off.p, reads->n_reads, k, (uint64_t *)ctx->partial_rows);
    }
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *d_rows = (const uint64_t *)ctx->partial_rows;
    return MDBG_OK;
}

extern "C" int mdbg_reduce_rows(mdbg_ctx *ctx, uint64_t *d_rows, uint64_t n_rows, uint32_t k, uint64_t *n_out) {
    if (!ctx || !n_out || (n_rows && !d_rows) || k < 2) return set_error(ctx, MDBG_EINVAL, "mdbg_reduce_rows: bad argument");
    if (n_rows >= (1ull << 32)) return set_error(ctx, MDBG_ERANGE, "more than 2^32 rows");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t rw = row_words_for(k);
    DeviceTable tab;
    MDBG_TRY(tab.init(ctx, n_rows * 2 + 1024));
    TableView tv = tab.view();
    if (n_rows) hipLaunchKernelGGL(rows_add_kernel, dim3(grid_for(n_rows, 256)), dim3(256), 0, ctx->stream, d_rows, n_rows, rw, tv);
    MDBG_HIP_CHECK(ctx, hipGetLastError());
    MDBG_TRY(tab.check_overflow(ctx));
    const uint64_t nslots = tab.cap + TABLE_EXC_CAP;
    DevBuf<uint32_t> flag;
    DevBuf<uint64_t> pos, tmp;
    MDBG_TRY(flag.alloc(ctx, nslots));
    MDBG_TRY(pos.alloc(ctx, nslots + 1));
    hipLaunchKernelGGL(mg_flag_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, 0u, 0, 0u, 1u, flag.p);
    MDBG_TRY(exclusive_scan_u32(ctx, flag.p, pos.p, nslots));
    uint64_t n = 0;
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &n, pos.p + nslots, 8, hipMemcpyDeviceToHost));
    MDBG_TRY(tmp.alloc(ctx, n * rw));
    hipLaunchKernelGGL(mg_emit_rows_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, flag.p, pos.p, d_rows, rw, tmp.p);
    if (n) MDBG_HIP_CHECK(ctx, hipMemcpyAsync(d_rows, tmp.p, n * rw * 8, hipMemcpyDeviceToDevice, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = n;
    return MDBG_OK;
}

extern "C" int mdbg_kminmer_count_first_merged(mdbg_ctx *ctx, const mdbg_minimizers *reads, uint32_t k, uint32_t min_abundance,
                                               const uint64_t *d_global_rows, uint64_t n_global_rows,
                                               uint32_t rank, uint32_t n_ranks, mdbg_table **out) {
    if (!ctx || !reads || !out || k < 2 || n_ranks < 1 || rank >= n_ranks || (n_global_rows && !d_global_rows))
        return set_error(ctx, MDBG_EINVAL, "mdbg_kminmer_count_first_merged: bad argument");
    if (n_global_rows >= (1ull << 32)) return set_error(ctx, MDBG_ERANGE, "more than 2^32 global rows");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t rw = row_words_for(k);
    // global key -> count table (every key appears once in the reduced rows; duplicates would be summed)
    DeviceTable tab;
    MDBG_TRY(tab.init(ctx, n_global_rows * 2 + 1024));
    TableView tv = tab.view();
    if (n_global_rows)
        hipLaunchKernelGGL(rows_add_kernel, dim3(grid_for(n_global_rows, 256)), dim3(256), 0, ctx->stream, d_global_rows, n_global_rows, rw, tv);
    MDBG_HIP_CHECK(ctx, hipGetLastError());
    MDBG_TRY(tab.check_overflow(ctx));
    const uint64_t nslots = tab.cap + TABLE_EXC_CAP;
    DevBuf<uint32_t> flag;
    DevBuf<uint64_t> pos;
    MDBG_TRY(flag.alloc(ctx, nslots));
    MDBG_TRY(pos.alloc(ctx, nslots + 1));
    hipLaunchKernelGGL(mg_flag_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, min_abundance, 1, rank, n_ranks, flag.p);
    MDBG_TRY(exclusive_scan_u32(ctx, flag.p, pos.p, nslots));
    uint64_t n_solid = 0;
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &n_solid, pos.p + nslots, 8, hipMemcpyDeviceToHost));

    // rescue over the local reads against GLOBAL abundances
    DevBuf<uint64_t> inst_off;
    uint64_t I = 0;
    MDBG_TRY(mg_inst_index(ctx, reads, k, inst_off, I));
    DevBuf<uint32_t> ab;
    MDBG_TRY(ab.alloc(ctx, I));
    if (I)
        hipLaunchKernelGGL(mg_inst_abundance_kernel, dim3(grid_for(I, 256)), dim3(256), 0, ctx->stream, reads->d_min.p, reads->d_off.p,
                           inst_off.p, reads->n_reads, I, k, tv, min_abundance, ab.p);
    mdbg_table *t = new mdbg_table();
    t->k = k;
    t->n_solid = n_solid;
    // mg_rescue_rows sizes the table (n_solid + rescued), emits rescued rows after the solid block
    int rc = mg_rescue_rows(ctx, reads, inst_off.p, min_abundance <= 1 ? I : 0, k, ab.p, t, n_solid);
    if (rc) { delete t; return rc; }
    hipLaunchKernelGGL(mg_emit_solid_kernel, dim3(grid_for(nslots, 256)), dim3(256), 0, ctx->stream, tv, tab.cap, flag.p, pos.p,
                       d_global_rows, rw, k, t->d_lo.p, t->d_hi.p, t->d_ab.p, t->d_vec.p);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { delete t; return set_error(ctx, MDBG_EHIP, "count_first_merged failed: %s", hipGetErrorString(e)); }
    *out = t;
    return MDBG_OK;
}
