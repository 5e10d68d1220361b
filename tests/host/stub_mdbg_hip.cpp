// stub_mdbg_hip.cpp -- TEST DOUBLE of libmdbg_hip.so for the host pipeline of metamdbg_amd/host/mdbg_tool.cpp (test infrastructure;
// never shipped, never loaded by the product: tests/test_tool_host_pipeline.py builds mdbg_tool against it in a scratch directory).
// No GPU, no minimizer arithmetic: every read of L bases gets max(0, L / 271) fake minimizers whose values are a function of the
// read's length alone (whatever batch it lands in), so that the tool's threads -- feeder, consumers one batch ahead, record builders writing
// in place, the statistics thread, the purge pass -- can be driven through tens of thousands of batches on a CPU and their output
// checked for order and completeness.  What it cannot show is numerics; that is what the -m gpu tests are for.
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mdbg_hip.h"

struct mdbg_ctx { std::string err; };
struct mdbg_reads { std::vector<uint32_t> len; };
struct mdbg_minimizers { std::vector<uint64_t> off; std::vector<uint32_t> m, pos, len; std::vector<uint8_t> dir, qual, flags; };
struct mdbg_table { uint32_t k = 0; std::vector<uint8_t> rec; std::vector<uint32_t> vec; };
struct mdbg_bytes { std::vector<uint8_t> d; std::atomic<uint64_t> tickets{0}; };
struct mdbg_census {};
struct mdbg_comm {};
struct mdbg_shard {};

static void jitter() {           // a little work of varying length, like kernels of varying batches
    static std::atomic<uint32_t> x{12345};
    uint32_t v = x.fetch_add(2654435761u);
    if (const char *e = getenv("MDBG_STUB_JITTER_US")) {
        const int us = atoi(e);
        if (us > 0) std::this_thread::sleep_for(std::chrono::microseconds((v >> 8) % (unsigned)us));
    }
}

extern "C" {
int mdbg_create(int, mdbg_ctx **ctx) { *ctx = new mdbg_ctx(); return MDBG_OK; }
void mdbg_destroy(mdbg_ctx *c) { delete c; }
const char *mdbg_last_error(const mdbg_ctx *c) { return c ? c->err.c_str() : "stub"; }
int mdbg_device_info(mdbg_ctx *, char *arch, size_t n, int *cu, uint64_t *hbm) { if (arch && n) snprintf(arch, n, "stub"); if (cu) *cu = 1; if (hbm) *hbm = 0; return MDBG_OK; }
int mdbg_host_alloc(mdbg_ctx *, size_t bytes, void **out) { *out = malloc(bytes ? bytes : 1); return *out ? MDBG_OK : MDBG_ENOMEM; }
void mdbg_host_free(mdbg_ctx *, void *p) { free(p); }

static mdbg_reads *make_reads(const uint32_t *lengths, uint32_t n) { mdbg_reads *r = new mdbg_reads(); r->len.assign(lengths, lengths + n); return r; }
int mdbg_reads_from_packed(mdbg_ctx *, const uint64_t *, const uint64_t *, const uint32_t *lengths, uint32_t n, mdbg_reads **out) { *out = make_reads(lengths, n); return MDBG_OK; }
int mdbg_reads_from_packed_async(mdbg_ctx *, const uint64_t *, const uint64_t *, const uint32_t *lengths, uint32_t n, mdbg_reads **out) { *out = make_reads(lengths, n); return MDBG_OK; }
int mdbg_reads_from_ascii(mdbg_ctx *, const char *, const char *, const uint64_t *offsets, uint32_t n, mdbg_reads **out) {
    mdbg_reads *r = new mdbg_reads();
    for (uint32_t i = 0; i < n; i++) r->len.push_back((uint32_t)(offsets[i + 1] - offsets[i]));
    *out = r;
    return MDBG_OK;
}
int mdbg_reads_attach_qualities(mdbg_ctx *, mdbg_reads *, const char *, const uint64_t *) { return MDBG_OK; }
int mdbg_reads_attach_qualities_async(mdbg_ctx *, mdbg_reads *, const char *, const uint64_t *) { return MDBG_OK; }
int mdbg_reads_mark_ascii(mdbg_ctx *, mdbg_reads *, const uint32_t *, uint32_t, const char *, const uint64_t *) { return MDBG_OK; }
int mdbg_reads_wait(mdbg_ctx *, const mdbg_reads *) { return MDBG_OK; }
void mdbg_reads_free(mdbg_reads *r) { delete r; }

int mdbg_scan(mdbg_ctx *, const mdbg_reads *reads, const mdbg_scan_params *, mdbg_minimizers **out) {
    jitter();
    mdbg_minimizers *m = new mdbg_minimizers();
    const uint32_t n = (uint32_t)reads->len.size();
    m->off.assign(1, 0);
    for (uint32_t r = 0; r < n; r++) {
        const uint32_t L = reads->len[r], k = L / 271;
        for (uint32_t i = 0; i < k; i++) {
            m->m.push_back(L * 2654435761u + i * 40503u);
            m->pos.push_back(i * 271u);
            m->dir.push_back((uint8_t)((L + i) & 1u));
            m->qual.push_back(1);
        }
        m->off.push_back(m->m.size());
        m->len.push_back(L);
        m->flags.push_back(0);
    }
    *out = m;
    return MDBG_OK;
}
int mdbg_minimizers_info(const mdbg_minimizers *m, uint32_t *n, uint64_t *t) { if (n) *n = (uint32_t)(m->off.size() - 1); if (t) *t = m->m.size(); return MDBG_OK; }
int mdbg_minimizers_to_host(mdbg_ctx *, const mdbg_minimizers *m, uint64_t *off, uint32_t *mins, uint32_t *pos, uint8_t *dir, uint8_t *qual,
                            uint32_t *len, float *meanq, uint8_t *flags) {
    jitter();
    const size_t n = m->off.size() - 1, t = m->m.size();
    if (off) memcpy(off, m->off.data(), (n + 1) * 8);
    if (mins && t) memcpy(mins, m->m.data(), t * 4);
    if (pos && t && !m->pos.empty()) memcpy(pos, m->pos.data(), t * 4);
    if (dir && t && !m->dir.empty()) memcpy(dir, m->dir.data(), t);
    if (qual && t && !m->qual.empty()) memcpy(qual, m->qual.data(), t);
    if (len && n && !m->len.empty()) memcpy(len, m->len.data(), n * 4);
    if (flags && n && !m->flags.empty()) memcpy(flags, m->flags.data(), n);
    if (meanq) for (size_t i = 0; i < n; i++) { const uint32_t nan = 0xFFC00000u; memcpy(&meanq[i], &nan, 4); }
    return MDBG_OK;
}
int mdbg_minimizers_from_host(mdbg_ctx *, const uint32_t *mins, const uint64_t *off, uint32_t n, mdbg_minimizers **out) {
    mdbg_minimizers *m = new mdbg_minimizers();
    m->off.assign(off, off + n + 1);
    m->m.assign(mins + off[0], mins + off[n]);
    *out = m;
    return MDBG_OK;
}
void mdbg_minimizers_free(mdbg_minimizers *m) { delete m; }
// a record file handed over as bytes: the double copies at once and takes the records apart on the host
int mdbg_bytes_create(mdbg_ctx *, uint64_t n, mdbg_bytes **out) { *out = new mdbg_bytes(); (*out)->d.resize(n); return MDBG_OK; }
int mdbg_bytes_upload_async(mdbg_ctx *, mdbg_bytes *b, uint64_t at, const void *host, uint64_t n, uint64_t *ticket) {
    if (at > b->d.size() || n > b->d.size() - at) return MDBG_EINVAL;
    jitter();
    if (n) memcpy(b->d.data() + at, host, n);
    const uint64_t t = b->tickets.fetch_add(1) + 1;
    if (ticket) *ticket = t;
    return MDBG_OK;
}
int mdbg_bytes_upload_done(mdbg_ctx *, mdbg_bytes *, uint64_t, int) { return 1; }
void mdbg_bytes_free(mdbg_bytes *b) { delete b; }
int mdbg_minimizers_from_record_bytes(mdbg_ctx *, const mdbg_bytes *b, const uint64_t *off, uint32_t n, uint8_t *circ, mdbg_minimizers **out) {
    if (5ull * n + 4ull * off[n] != b->d.size()) return MDBG_EINVAL;
    mdbg_minimizers *m = new mdbg_minimizers();
    m->off.assign(off, off + n + 1);
    m->m.resize(off[n]);
    for (uint32_t r = 0; r < n; r++) {
        const uint8_t *p = b->d.data() + 5ull * r + 4ull * off[r];
        uint32_t cnt; memcpy(&cnt, p, 4);
        if (cnt != off[r + 1] - off[r]) { delete m; return MDBG_EINVAL; }
        if (circ) circ[r] = p[4];
        if (cnt) memcpy(m->m.data() + off[r], p + 5, (size_t)cnt * 4);
    }
    *out = m;
    return MDBG_OK;
}
int mdbg_prev_from_record_bytes(mdbg_ctx *, const mdbg_bytes *, uint64_t, mdbg_table **) { return MDBG_ENODEV; }
int mdbg_minimizers_concat(mdbg_ctx *, const mdbg_minimizers *const *parts, uint32_t n_parts, mdbg_minimizers **out) {
    jitter();
    mdbg_minimizers *m = new mdbg_minimizers();
    m->off.assign(1, 0);
    for (uint32_t p = 0; p < n_parts; p++) {
        const mdbg_minimizers *q = parts[p];
        const uint64_t base = m->m.size();
        for (size_t i = 1; i < q->off.size(); i++) m->off.push_back(base + q->off[i]);
        m->m.insert(m->m.end(), q->m.begin(), q->m.end());
    }
    *out = m;
    return MDBG_OK;
}
int mdbg_purge_palindromes(mdbg_ctx *, const mdbg_minimizers *in, uint32_t, uint32_t, mdbg_minimizers **out) {
    jitter();
    mdbg_minimizers *m = new mdbg_minimizers();
    m->off = in->off; m->m = in->m;
    *out = m;
    return MDBG_OK;
}
int mdbg_census_create(mdbg_ctx *, mdbg_census **out) { *out = new mdbg_census(); return MDBG_OK; }
int mdbg_census_add(mdbg_ctx *, mdbg_census *, const mdbg_minimizers *) { return MDBG_OK; }
int mdbg_census_top(mdbg_ctx *, const mdbg_census *, uint32_t *, uint32_t *n) { *n = 0; return MDBG_OK; }
void mdbg_census_free(mdbg_census *c) { delete c; }
// ---- `graph --firstpass` through the stub: one fake row per read that has at least k minimizers (key and abundance a function of the read's
// minimizers alone, the vector its first k), in the order the reads are given -- enough to check that `asmStep` hands the first pass the same
// reads as `graph` finds in the file.  The passes above firstK are not exercised: their entry points exist so that the tool links.
int mdbg_kminmer_count_first(mdbg_ctx *, const mdbg_minimizers *m, uint32_t k, uint32_t, mdbg_table **out) {
    jitter();
    mdbg_table *t = new mdbg_table();
    t->k = k;
    for (size_t r = 0; r + 1 < m->off.size(); r++) {
        const uint64_t a = m->off[r], n = m->off[r + 1] - a;
        if (n < k) continue;
        const uint64_t lo = m->m[a] | ((uint64_t)m->m[a + 1] << 32), hi = (uint64_t)m->m[a + n - 1] * 0x9E3779B97F4A7C15ull + n;
        const uint32_t ab = (uint32_t)n;
        uint8_t rec[20];
        memcpy(rec, &lo, 8); memcpy(rec + 8, &hi, 8); memcpy(rec + 16, &ab, 4);
        t->rec.insert(t->rec.end(), rec, rec + 20);
        t->vec.insert(t->vec.end(), m->m.begin() + (long)a, m->m.begin() + (long)(a + k));
    }
    *out = t;
    return MDBG_OK;
}
int mdbg_kminmer_count_first_sharded(mdbg_ctx *, mdbg_comm *, const mdbg_minimizers *, uint32_t, uint32_t, mdbg_table **) { return MDBG_ENODEV; }
int mdbg_prev_from_records(mdbg_ctx *, const uint8_t *, uint64_t, mdbg_table **) { return MDBG_ENODEV; }
int mdbg_prev_overlay_unitigs(mdbg_ctx *, mdbg_table *, const mdbg_minimizers *, const uint32_t *, uint32_t) { return MDBG_ENODEV; }
int mdbg_kminmer_count_refined(mdbg_ctx *, const mdbg_minimizers *, const mdbg_minimizers *, uint32_t, const mdbg_table *, mdbg_table **) { return MDBG_ENODEV; }
int mdbg_kminmer_index(mdbg_ctx *, const mdbg_minimizers *, const mdbg_minimizers *, uint32_t, const mdbg_table *, mdbg_table **) { return MDBG_ENODEV; }
int mdbg_small_contigs(mdbg_ctx *, const mdbg_minimizers *, uint32_t, uint32_t, const mdbg_table *, uint8_t *) { return MDBG_ENODEV; }
int mdbg_table_info(const mdbg_table *t, uint32_t *k, uint64_t *n, uint64_t *solid, int *has_vec) {
    if (k) *k = t->k;
    if (n) *n = t->rec.size() / 20;
    if (solid) *solid = t->rec.size() / 20;
    if (has_vec) *has_vec = 1;
    return MDBG_OK;
}
int mdbg_table_checksum(mdbg_ctx *, const mdbg_table *t, uint64_t *sums) {
    sums[0] = sums[1] = sums[2] = sums[3] = 0;
    for (size_t i = 0; i < t->rec.size(); i += 20) { uint64_t lo; uint32_t ab; memcpy(&lo, &t->rec[i], 8); memcpy(&ab, &t->rec[i + 16], 4); sums[0] += lo * ab; sums[1] += ab; sums[2] += lo; }
    return MDBG_OK;
}
int mdbg_table_to_host_range(mdbg_ctx *, const mdbg_table *t, uint64_t first, uint64_t count, uint8_t *rec, uint32_t *vec) {
    if (rec && count) memcpy(rec, t->rec.data() + first * 20, count * 20);
    if (vec && count) memcpy(vec, t->vec.data() + first * t->k, count * t->k * 4);
    return MDBG_OK;
}
int mdbg_table_to_host(mdbg_ctx *c, const mdbg_table *t, uint8_t *rec, uint32_t *vec) { return mdbg_table_to_host_range(c, t, 0, t->rec.size() / 20, rec, vec); }
void mdbg_table_free(mdbg_table *t) { delete t; }
int mdbg_edge_index(mdbg_ctx *, const mdbg_table *, mdbg_table **, uint64_t *) { return MDBG_ENODEV; }     // (refdrv_hip edges_hip links them; never called against the double)
int mdbg_unitig_edge_index(mdbg_ctx *, const mdbg_minimizers *, uint32_t, mdbg_table **, uint64_t *) { return MDBG_ENODEV; }
int mdbg_table_keys_to_host(mdbg_ctx *, const mdbg_table *, uint64_t *) { return MDBG_ENODEV; }
int mdbg_shard_from_table(mdbg_ctx *, const mdbg_table *, uint32_t, mdbg_shard **, const uint64_t **, uint64_t *) { return MDBG_ENODEV; }
int mdbg_shard_exchange(mdbg_ctx *, mdbg_comm *, mdbg_shard *, const uint64_t *, const uint64_t *, const uint64_t **) { return MDBG_ENODEV; }
int mdbg_shard_begin(mdbg_ctx *, const mdbg_minimizers *, uint32_t, uint32_t, mdbg_shard **, const uint64_t **, uint64_t *) { return MDBG_ENODEV; }
int mdbg_shard_finish(mdbg_ctx *, mdbg_shard *, const uint64_t *, uint32_t, mdbg_table **) { return MDBG_ENODEV; }
int mdbg_shard_exchange_local(mdbg_ctx *, mdbg_shard *const *, uint32_t, const uint64_t *const *, const uint64_t *, const uint64_t **) { return MDBG_ENODEV; }
int mdbg_shard_keep(mdbg_ctx *, mdbg_shard *, const uint64_t *, mdbg_table **) { return MDBG_ENODEV; }
void mdbg_shard_free(mdbg_shard *s) { delete s; }
int mdbg_comm_unique_id(uint8_t *) { return MDBG_ENODEV; }
int mdbg_comm_create(mdbg_ctx *, const uint8_t *, int, int, mdbg_comm **) { return MDBG_ENODEV; }
int mdbg_comm_create_mode(mdbg_ctx *, const uint8_t *, int, int, int, mdbg_comm **) { return MDBG_ENODEV; }
int mdbg_comm_mode(const mdbg_comm *) { return MDBG_COMM_RCCL; }
const char *mdbg_comm_note(const mdbg_comm *) { return ""; }
int mdbg_comm_times(const mdbg_comm *, double *) { return MDBG_ENODEV; }
void mdbg_comm_destroy(mdbg_comm *c) { delete c; }
}
