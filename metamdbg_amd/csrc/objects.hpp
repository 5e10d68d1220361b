// objects.hpp -- the opaque handles of mdbg_hip.h: device-resident reads, minimizer CSR, tables.
#pragma once
#include "common.hpp"
#include "table.hpp"

#include <memory>
#include <mutex>

// Base-space reads in HBM.  Layout (DESIGN.md "data layout"):
//   d_words      u64[n_words]   32 bases per word, base i at bits [2i, 2i+2), code (c>>1)&3
//   d_word_off   u64[n+1]       read r occupies words [off[r], off[r+1]); off[r] is even (16-byte aligned reads)
//   d_len        u32[n]         original length in bases
//   d_invalid    u32[n_words]   optional, bit i = base i of the word had bit 3 set (N/n); NULL when no read has any
//   d_break      u32[n_words]   optional, bit i = base i differs as a CHARACTER from the base before it although (code, invalid)
//                               are equal ("aA", "CR", ...): the reference's homopolymer compression compares raw characters
//                               (Commons.hpp:4177-4178), so such a base starts a run; NULL when no read has any
//   d_qual       u8[..]         optional phred+33 bytes, read r at [qual_off[r], qual_off[r+1])
struct mdbg_reads {
    uint32_t n_reads = 0;
    uint64_t n_bases = 0;
    uint64_t n_words = 0;
    uint32_t max_len = 0;
    mdbg::DevBuf<uint64_t> d_words;
    mdbg::DevBuf<uint64_t> d_word_off;
    mdbg::DevBuf<uint32_t> d_len;
    mdbg::DevBuf<uint32_t> d_invalid;
    mdbg::DevBuf<uint32_t> d_break;
    mdbg::DevBuf<uint8_t> d_qual;
    mdbg::DevBuf<uint64_t> d_qual_off;
    // reads with a side-mask bit anywhere (an N, a case flip inside a run ...): one flag per read and their sorted list.  A batch in
    // which they are few is scanned by the block-structured kernel; these reads alone take the general kernel (mdbg_scan)
    mdbg::DevBuf<uint8_t> d_masked;         // n_reads (present when has_invalid)
    mdbg::DevBuf<uint32_t> d_masked_list;   // n_masked, ascending
    uint32_t n_masked = 0;
    bool has_invalid = false;   // d_invalid is present (some base is invalid, or d_break is present)
    bool has_break = false;
    bool has_qual = false;
    // mdbg_reads_from_packed_async: the upload is queued on the context's upload stream; `ready` is recorded behind it and every
    // consumer orders itself after it (reads_ready_on / reads_ready_host).  h_rel keeps the rebased offsets alive until then.
    hipEvent_t ready = nullptr;
    std::vector<uint64_t> h_rel, h_qrel;
    std::vector<uint32_t> h_len;            // the lengths as uploaded (mdbg_reads_attach_qualities_async checks the qualities against them)
    mdbg_reads() = default;
    mdbg_reads(const mdbg_reads &) = delete;
    mdbg_reads &operator=(const mdbg_reads &) = delete;
    ~mdbg_reads() { if (ready) { (void)hipEventSynchronize(ready); (void)hipEventDestroy(ready); } }   // before the buffers go
};

namespace mdbg {
// the kernels `ctx` queues next see the reads complete (no host wait)
inline hipError_t reads_ready_on(mdbg_ctx *ctx, const mdbg_reads *r) {
    return r && r->ready ? hipStreamWaitEvent(ctx->stream, r->ready, 0) : hipSuccess;
}
// the host may read the device buffers / reuse the buffers it uploaded from
inline hipError_t reads_ready_host(const mdbg_reads *r) { return r && r->ready ? hipEventSynchronize(r->ready) : hipSuccess; }
}

// A file's bytes on the device, uploaded in pieces on the context's upload stream (mdbg_bytes_*).  Every piece records an event behind
// itself; a ticket is the piece's ordinal (copies on one stream complete in order, so "ticket t done" covers every earlier one).
struct mdbg_bytes {
    uint64_t n = 0;
    mdbg::DevBuf<uint8_t> d;
    std::mutex mu;
    std::vector<hipEvent_t> events;     // events[t - 1] behind piece t
    uint64_t done_upto = 0;             // tickets <= done_upto are known complete
    mdbg_ctx *owner = nullptr;
    mdbg_bytes() = default;
    mdbg_bytes(const mdbg_bytes &) = delete;
    mdbg_bytes &operator=(const mdbg_bytes &) = delete;
    ~mdbg_bytes() { for (hipEvent_t e : events) if (e) { (void)hipEventSynchronize(e); (void)hipEventDestroy(e); } }   // before the buffer goes
};

// Minimizer-space sequences in HBM as CSR.  Per-minimizer side arrays exist only for scan output.
struct mdbg_minimizers {
    uint32_t n_reads = 0;
    uint64_t n_min = 0;
    mdbg::DevBuf<uint64_t> d_off;     // n_reads + 1
    mdbg::DevBuf<uint32_t> d_min;     // n_min
    mdbg::DevBuf<uint32_t> d_pos;     // n_min (scan output only)
    mdbg::DevBuf<uint8_t> d_dir;      // n_min (scan output only)
    mdbg::DevBuf<uint8_t> d_mqual;    // n_min (scan output only)
    mdbg::DevBuf<uint32_t> d_len;     // n_reads: original read length (scan output only)
    mdbg::DevBuf<uint8_t> d_flags;    // n_reads: MDBG_READ_* (scan output only)
    std::vector<float> h_mean_quality;  // n_reads, finished on the host from per-read quality sums; empty: every read has mean_quality_all
    float mean_quality_all = 0.0f;
    bool from_scan = false;
    // Fresh output of the block-structured scan kernel: the rows of read r sit at [d_begin[r], d_begin[r] + d_cnt[r]) of
    // d_min / d_pos / d_dir in the order the waves finished their reads (each wave takes room with one atomic add when its
    // read is done, and writes coalesced rows once).  mdbg_purge_palindromes -- the next step of the path -- reads this
    // form directly and writes canonical CSR; every other consumer has it brought into CSR order first (ensure_canonical),
    // which is the copy the padded-slot design of round 1 always paid.
    bool scattered = false;
    mdbg::DevBuf<uint64_t> d_begin;   // n_reads (scattered only)
    mdbg::DevBuf<uint32_t> d_cnt;     // n_reads (scattered only)
    mutable std::mutex canon_mu;      // ensure_canonical: several consumers may meet the same fresh scan output (two contexts, two threads)
    uint64_t n_rows = 0;              // scattered only: extent of the row arrays (>= n_min: the regions the waves fill are not full,
                                      // and rows of reads the complexity filter emptied stay behind)
    mdbg_ctx *owner = nullptr;        // the context whose stream produced the object
};

namespace mdbg {
// brings a scattered scan output into CSR order in place (no-op otherwise); the object is logically unchanged
int ensure_canonical(mdbg_ctx *ctx, const mdbg_minimizers *m);
}

// k-min-mer table: output rows (file order) + an open-addressing lookup over the same keys.
struct mdbg_table {
    uint32_t k = 0;
    uint64_t n_records = 0;
    uint64_t n_solid = 0;
    bool has_vectors = false;
    mdbg::DevBuf<uint64_t> d_lo, d_hi;   // n_records
    mdbg::DevBuf<uint32_t> d_ab;         // n_records
    mdbg::DevBuf<uint32_t> d_vec;        // n_records * k (when has_vectors)
    // what the pass that built the table walked (mdbg_table_stats): minimizers read, k-min-mer instances, distinct keys inserted,
    // slots of the hash table it used -- the terms of SURVEY.md 8(d)'s algorithmic bytes 4 M + 16 I + 20 D
    uint64_t st_minimizers = 0, st_instances = 0, st_keys = 0, st_slots = 0;
    // key -> abundance lookup (always present for prev tables; built on demand for output tables)
    std::unique_ptr<mdbg::DeviceTable> lookup;
    // the same map in the compact form the passes above firstK read (table.hpp "buckets of three keys"): left behind by the index
    // pass that built the table, or made on first use from `lookup` (if present: it may carry a unitig overlay) or from the rows
    std::unique_ptr<mdbg::BucketTable> image;
};
