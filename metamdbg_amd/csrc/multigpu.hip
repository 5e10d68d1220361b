// multigpu.hip -- the exchange of the sharded first pass inside the library: RCCL point-to-point over xGMI.
//
// Reads shard across GPUs; only the k-min-mer counts are global.  mdbg_kminmer_count_first_sharded runs
// mdbg_shard_begin -> rows to their owner ranks -> mdbg_shard_reduce -> replies back -> mdbg_shard_finish in one call, on the
// context's stream, every transfer an ncclSend / ncclRecv pair inside one group (an all-to-all: all seven xGMI links of a GPU
// carry traffic at once; xGMI is point-to-point, a ring collective would be bound by one link).  Nearest reference analogue:
// KminmerCounter's on-disk partitioning `vecHash % _nbPartitions` + per-partition dereplication
// (graph/CreateMdbg.hpp:3714-3724, :3744-3851).
//
// RCCL is loaded on first use (dlopen of librccl.so.1: the copy already in the process if the caller -- torch, MPI -- brought
// one), so single-GPU users of libmdbg_hip.so do not depend on it.
#include "common.hpp"
#include "objects.hpp"

#include <algorithm>
#include <chrono>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <memory>
#include <mutex>
#include <vector>

namespace mdbg {

struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};

static RcclApi *rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
        }
        if (!api.handle) { api.error = std::string("RCCL not found: ") + dlerror(); return; }
#define MDBG_SYM(field, sym)                                                       \
        api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym)); \
        if (!api.field && api.error.empty()) api.error = std::string("RCCL symbol missing: ") + sym;
        MDBG_SYM(GetUniqueId, "ncclGetUniqueId")
        MDBG_SYM(CommInitRank, "ncclCommInitRank")
        MDBG_SYM(CommDestroy, "ncclCommDestroy")
        MDBG_SYM(GroupStart, "ncclGroupStart")
        MDBG_SYM(GroupEnd, "ncclGroupEnd")
        MDBG_SYM(Send, "ncclSend")
        MDBG_SYM(Recv, "ncclRecv")
        MDBG_SYM(AllGather, "ncclAllGather")
        MDBG_SYM(CommCount, "ncclCommCount")
        MDBG_SYM(CommUserRank, "ncclCommUserRank")
        MDBG_SYM(CommAbort, "ncclCommAbort")
        MDBG_SYM(GetErrorString, "ncclGetErrorString")
#undef MDBG_SYM
    });
    return &api;
}

}  // namespace mdbg

using namespace mdbg;

struct mdbg_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, n_ranks = 1;
    int rccl_count = 0, rccl_rank = -1;   // what the communicator itself reports (ncclCommCount / ncclCommUserRank)
    bool owned = false;          // created by mdbg_comm_create (destroyed with the handle) or adopted from the caller
    bool broken = false;         // an RCCL call failed on it: every later exchange fails at once instead of hanging the peers
    mdbg::DevBuf<uint64_t> replies;   // global counts for the rows of the last exchange (valid until the next one)
    // agreement buffers, allocated with the communicator so that taking part in the first collective of an exchange cannot fail
    // locally: this rank's row of the count matrix + its status word, everybody's, and a status word per later phase
    mdbg::DevBuf<uint64_t> d_mine, d_all;
    // mdbg_comm_stats
    uint64_t n_exchanges = 0, bytes_to_peers = 0, bytes_from_peers = 0, bytes_local = 0;
    double exchange_ms = 0.0;    // host wall time inside mdbg_shard_exchange (includes waiting for the slowest peer)
};

#define MDBG_NCCL_CHECK(ctx, api, expr)                                                                                   \
    do {                                                                                                                  \
        ncclResult_t r_ = (expr);                                                                                         \
        if (r_ != ncclSuccess)                                                                                            \
            return set_error((ctx), MDBG_EHIP, "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

extern "C" int mdbg_comm_unique_id(uint8_t *id128) {
    if (!id128) return MDBG_EINVAL;
    RcclApi *api = rccl_api();
    if (!api->error.empty()) { g_last_error = api->error; return MDBG_ENODEV; }
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) { g_last_error = std::string("ncclGetUniqueId failed: ") + api->GetErrorString(r); return MDBG_EHIP; }
    memcpy(id128, &id, sizeof id);
    return MDBG_OK;
}

// what the communicator says about itself must be what the caller said: a rank / size mix-up shows here, not as a hang
static int comm_finish_setup(mdbg_ctx *ctx, RcclApi *api, mdbg_comm *c, const char *who) {
    MDBG_NCCL_CHECK(ctx, api, api->CommCount(c->comm, &c->rccl_count));
    MDBG_NCCL_CHECK(ctx, api, api->CommUserRank(c->comm, &c->rccl_rank));
    if (c->rccl_count != c->n_ranks || c->rccl_rank != c->rank)
        return set_error(ctx, MDBG_EINVAL, "%s: the communicator is rank %d of %d, the caller said rank %d of %d", who, c->rccl_rank,
                         c->rccl_count, c->rank, c->n_ranks);
    MDBG_TRY(c->d_mine.alloc(ctx, (size_t)c->n_ranks + 1));
    MDBG_TRY(c->d_all.alloc(ctx, (size_t)c->n_ranks * (c->n_ranks + 1)));
    return MDBG_OK;
}

extern "C" int mdbg_comm_create(mdbg_ctx *ctx, const uint8_t *id128, int rank, int n_ranks, mdbg_comm **out) try {
    if (!ctx || !id128 || !out || n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks)
        return set_error(ctx, MDBG_EINVAL, "mdbg_comm_create: bad argument (ranks 1..64)");
    RcclApi *api = rccl_api();
    if (!api->error.empty()) return set_error(ctx, MDBG_ENODEV, "%s", api->error.c_str());
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    std::unique_ptr<mdbg_comm, void (*)(mdbg_comm *)> c(new mdbg_comm(), mdbg_comm_destroy);
    c->rank = rank; c->n_ranks = n_ranks; c->owned = true;
    MDBG_NCCL_CHECK(ctx, api, api->CommInitRank(&c->comm, n_ranks, id, rank));
    MDBG_TRY(comm_finish_setup(ctx, api, c.get(), "mdbg_comm_create"));
    *out = c.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_comm_adopt(mdbg_ctx *ctx, void *nccl_comm, int rank, int n_ranks, mdbg_comm **out) try {
    if (!ctx || !nccl_comm || !out || n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks)
        return set_error(ctx, MDBG_EINVAL, "mdbg_comm_adopt: bad argument (ranks 1..64)");
    RcclApi *api = rccl_api();
    if (!api->error.empty()) return set_error(ctx, MDBG_ENODEV, "%s", api->error.c_str());
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<mdbg_comm, void (*)(mdbg_comm *)> c(new mdbg_comm(), mdbg_comm_destroy);
    c->comm = (ncclComm_t)nccl_comm; c->rank = rank; c->n_ranks = n_ranks; c->owned = false;
    MDBG_TRY(comm_finish_setup(ctx, api, c.get(), "mdbg_comm_adopt"));
    *out = c.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" void mdbg_comm_destroy(mdbg_comm *c) {
    if (!c) return;
    // a communicator an RCCL call failed on is aborted, not destroyed: ncclCommDestroy waits for operations that will never finish
    if (c->owned && c->comm) (void)(c->broken ? rccl_api()->CommAbort(c->comm) : rccl_api()->CommDestroy(c->comm));
    delete c;
}

extern "C" int mdbg_comm_stats(const mdbg_comm *c, uint64_t stats[8], double *exchange_ms) {
    if (!c || !stats) return MDBG_EINVAL;
    stats[0] = (uint64_t)c->rank; stats[1] = (uint64_t)c->n_ranks; stats[2] = (uint64_t)c->rccl_count; stats[3] = c->n_exchanges;
    stats[4] = c->bytes_to_peers; stats[5] = c->bytes_from_peers; stats[6] = c->bytes_local; stats[7] = (uint64_t)(int64_t)c->rccl_rank;
    if (exchange_ms) *exchange_ms = c->exchange_ms;
    return MDBG_OK;
}

namespace {

// ncclGroupStart ... ncclGroupEnd that cannot be left open: the group is closed on every path out of the scope (an open group
// makes every later RCCL call of the thread part of it -- the next collective on any communicator would never start).
struct GroupGuard {
    RcclApi *api;
    bool open = false;
    explicit GroupGuard(RcclApi *a) : api(a) {}
    ncclResult_t start() { ncclResult_t r = api->GroupStart(); open = r == ncclSuccess; return r; }
    ncclResult_t end() { if (!open) return ncclSuccess; open = false; return api->GroupEnd(); }
    ~GroupGuard() { if (open) (void)api->GroupEnd(); }
};

// Every rank says whether it can go on (0) or not (its negative MDBG_E* code) and learns what the others said: a rank that failed
// locally between two transfers (an allocation, the reduction on the owner) still takes part in this small all-gather, so that
// its peers return an error too instead of waiting in the next receive for rows that will never come.
// Returns MDBG_OK when everybody can go on; otherwise the local error (already set) or MDBG_EPEER naming the first failed rank.
int agree(mdbg_ctx *ctx, RcclApi *api, mdbg_comm *comm, int local_rc, const char *phase) {
    const int n = comm->n_ranks;
    const uint64_t st = (uint64_t)(int64_t)local_rc;
    {       // (also with one rank: the communicator is exercised the same way whatever the job's size)
        hipError_t e = hipMemcpyAsync(comm->d_mine.p, &st, 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);      // `st` is a stack word
        if (e != hipSuccess) { comm->broken = true; return set_error(ctx, MDBG_EHIP, "mdbg_shard_exchange (%s): %s", phase, hipGetErrorString(e)); }
        ncclResult_t r = api->AllGather(comm->d_mine.p, comm->d_all.p, 1, ncclUint64, comm->comm, ctx->stream);
        if (r != ncclSuccess) { comm->broken = true; return set_error(ctx, MDBG_EHIP, "ncclAllGather (%s) failed: %s", phase, api->GetErrorString(r)); }
        std::vector<uint64_t> all((size_t)n);
        e = memcpy_sync(ctx, all.data(), comm->d_all.p, (size_t)n * 8, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { comm->broken = true; return set_error(ctx, MDBG_EHIP, "mdbg_shard_exchange (%s): %s", phase, hipGetErrorString(e)); }
        if (local_rc != MDBG_OK) return local_rc;
        for (int r2 = 0; r2 < n; r2++)
            if (all[r2] != 0)
                return set_error(ctx, MDBG_EPEER, "mdbg_shard_exchange: rank %d failed %s (code %lld); nothing was exchanged", r2, phase,
                                 (long long)(int64_t)all[r2]);
    }
    return local_rc;
}

// The first collective of an exchange: every rank's row of the count matrix and its status word to everybody.
// all[r * (n + 1) + j] = rows rank r holds for rank j; all[r * (n + 1) + n] = status of rank r.
int gather_counts(mdbg_ctx *ctx, RcclApi *api, mdbg_comm *comm, const uint64_t *counts, int local_rc, std::vector<uint64_t> &all) {
    const int n = comm->n_ranks;
    std::vector<uint64_t> mine((size_t)n + 1, 0);
    if (local_rc == MDBG_OK) for (int r = 0; r < n; r++) mine[r] = counts[r];
    mine[n] = (uint64_t)(int64_t)local_rc;
    all.assign((size_t)n * (n + 1), 0);
    hipError_t e = hipMemcpyAsync(comm->d_mine.p, mine.data(), mine.size() * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { comm->broken = true; return set_error(ctx, MDBG_EHIP, "mdbg_shard_exchange (counts): %s", hipGetErrorString(e)); }
    ncclResult_t r = api->AllGather(comm->d_mine.p, comm->d_all.p, (size_t)n + 1, ncclUint64, comm->comm, ctx->stream);
    if (r != ncclSuccess) { comm->broken = true; return set_error(ctx, MDBG_EHIP, "ncclAllGather (counts) failed: %s", api->GetErrorString(r)); }
    e = memcpy_sync(ctx, all.data(), comm->d_all.p, all.size() * 8, hipMemcpyDeviceToHost);
    if (e != hipSuccess) { comm->broken = true; return set_error(ctx, MDBG_EHIP, "mdbg_shard_exchange (counts): %s", hipGetErrorString(e)); }
    if (local_rc != MDBG_OK) return local_rc;
    for (int r2 = 0; r2 < n; r2++)
        if (all[(size_t)r2 * (n + 1) + n] != 0)
            return set_error(ctx, MDBG_EPEER, "mdbg_shard_exchange: rank %d failed before the exchange (code %lld); nothing was exchanged", r2,
                             (long long)(int64_t)all[(size_t)r2 * (n + 1) + n]);
    return MDBG_OK;
}

// One all-to-all: for every peer r, n_send[r] u64 from src + s_off[r] and n_get[r] u64 into dst + r_off[r], in pieces of at most
// 256 MiB (the same cut on both sides), all inside ONE group that is closed whatever happens.
int all_to_all(mdbg_ctx *ctx, RcclApi *api, mdbg_comm *comm, const uint64_t *src, const std::vector<uint64_t> &s_off, const std::vector<uint64_t> &n_send,
               uint64_t *dst, const std::vector<uint64_t> &r_off, const std::vector<uint64_t> &n_get) {
    constexpr uint64_t PIECE = 1ull << 25;          // u64 elements
    const int n = comm->n_ranks, me = comm->rank;
    if (n <= 1) return MDBG_OK;
    GroupGuard group(api);
    ncclResult_t first = group.start();
    const char *what = "ncclGroupStart";
    for (int r = 0; r < n && first == ncclSuccess; r++) {
        if (r == me) continue;
        for (uint64_t at = 0; at < n_send[r] && first == ncclSuccess; at += PIECE) {
            first = api->Send(src + s_off[r] + at, std::min(PIECE, n_send[r] - at), ncclUint64, r, comm->comm, ctx->stream);
            what = "ncclSend";
        }
        for (uint64_t at = 0; at < n_get[r] && first == ncclSuccess; at += PIECE) {
            first = api->Recv(dst + r_off[r] + at, std::min(PIECE, n_get[r] - at), ncclUint64, r, comm->comm, ctx->stream);
            what = "ncclRecv";
        }
    }
    const ncclResult_t ended = group.end();          // always: also after a failed send / receive
    if (first == ncclSuccess && ended != ncclSuccess) { first = ended; what = "ncclGroupEnd"; }
    if (first != ncclSuccess) {
        comm->broken = true;
        return set_error(ctx, MDBG_EHIP, "%s failed inside mdbg_shard_exchange: %s", what, api->GetErrorString(first));
    }
    for (int r = 0; r < n; r++)
        if (r != me) { comm->bytes_to_peers += n_send[r] * 8; comm->bytes_from_peers += n_get[r] * 8; }
    return MDBG_OK;
}

int exchange_impl(mdbg_ctx *ctx, mdbg_comm *comm, mdbg_shard *shard, const uint64_t *d_rows, const uint64_t *counts, const uint64_t **d_replies,
                  int local_rc) {
    RcclApi *api = rccl_api();
    if (!api->error.empty()) return set_error(ctx, MDBG_ENODEV, "%s", api->error.c_str());
    if (comm->broken) return set_error(ctx, MDBG_EHIP, "mdbg_shard_exchange: an earlier RCCL call failed on this communicator");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int n = comm->n_ranks, me = comm->rank;
    const uint32_t rw = mdbg_row_words(4);     // the same for every k: vectors never travel
    const auto t_enter = std::chrono::steady_clock::now();
    MDBG_DBG(ctx, "shard_exchange: enter, %d ranks", n);

    const int fail_phase = ctx->test_exchange_fail_phase;       // tests: one-shot local failures
    ctx->test_exchange_fail_phase = 0;
    if (fail_phase == 1 && local_rc == MDBG_OK) local_rc = set_error(ctx, MDBG_EINVAL, "mdbg_shard_exchange: test failure before the counts");
    // ---- who sends how many rows to whom (and whether everybody got this far) ----
    if (local_rc == MDBG_OK && !d_rows) {
        uint64_t n_rows = 0;
        for (int r = 0; r < n; r++) n_rows += counts[r];
        if (n_rows) local_rc = set_error(ctx, MDBG_EINVAL, "mdbg_shard_exchange: null rows");
    }
    std::vector<uint64_t> all;
    MDBG_TRY(gather_counts(ctx, api, comm, counts, local_rc, all));
    std::vector<uint64_t> s_cnt((size_t)n), got((size_t)n), soff((size_t)n + 1, 0), roff((size_t)n + 1, 0);
    for (int r = 0; r < n; r++) {
        s_cnt[r] = counts[r];
        got[r] = all[(size_t)r * (n + 1) + me];          // rows rank r holds for me
        soff[r + 1] = soff[r] + s_cnt[r];
        roff[r + 1] = roff[r] + got[r];
    }
    const uint64_t n_sent = soff[n], n_recv = roff[n];
    MDBG_DBG(ctx, "shard_exchange: counts known, %llu rows out, %llu in", (unsigned long long)n_sent, (unsigned long long)n_recv);

    // ---- rows to their owners ----
    DevBuf<uint64_t> d_recv;
    int rc = d_recv.alloc(ctx, n_recv * rw);
    if (rc == MDBG_OK) rc = comm->replies.alloc(ctx, n_sent);
    if (fail_phase == 2 && rc == MDBG_OK) rc = set_error(ctx, MDBG_ENOMEM, "mdbg_shard_exchange: test failure allocating the receive buffers");
    MDBG_TRY(agree(ctx, api, comm, rc, "allocating the receive buffers"));
    // A rank's own share never goes through RCCL: it is a device-to-device copy on the same stream.  (RCCL 2.26's send/receive
    // to self returned with 531 MiB of a 1.1 GB message in place, the rest still zero when the next kernel on the stream read it;
    // one rank on the per-rank workload of an 8-GPU job, profiles/r02z_*.)
    std::vector<uint64_t> w_soff(n + 1), w_roff(n + 1), w_send(n), w_get(n);
    for (int r = 0; r < n; r++) { w_soff[r] = soff[r] * rw; w_roff[r] = roff[r] * rw; w_send[r] = s_cnt[r] * rw; w_get[r] = got[r] * rw; }
    {
        LaunchTimer timer(ctx, "shard_exchange");
        if (s_cnt[me]) {
            MDBG_HIP_CHECK(ctx, hipMemcpyAsync(d_recv.p + w_roff[me], d_rows + w_soff[me], w_send[me] * 8, hipMemcpyDeviceToDevice, ctx->stream));
            comm->bytes_local += w_send[me] * 8;
        }
        MDBG_TRY(all_to_all(ctx, api, comm, d_rows, w_soff, w_send, d_recv.p, w_roff, w_get));
    }
    // ---- the owner sums and answers ----
    MDBG_DBG(ctx, "shard_exchange: rows queued");
    if (debug_on()) {
        hipError_t de = hipStreamSynchronize(ctx->stream);
        uint64_t a[5] = {0, 0, 0, 0, 0}, b[5] = {0, 0, 0, 0, 0};
        if (n_sent) (void)hipMemcpy(a, d_rows, sizeof a, hipMemcpyDeviceToHost);
        if (n_recv) (void)hipMemcpy(b, d_recv.p, sizeof b, hipMemcpyDeviceToHost);
        MDBG_DBG(ctx, "shard_exchange: rows arrived (%s); first row out %llx %llx %llu, in %llx %llx %llu", hipGetErrorString(de),
                 (unsigned long long)a[0], (unsigned long long)a[1], (unsigned long long)a[2], (unsigned long long)b[0], (unsigned long long)b[1], (unsigned long long)b[2]);
    }
    const uint64_t *d_reply = nullptr;
    rc = mdbg_shard_reduce(ctx, shard, d_recv.p, n_recv, &d_reply);
    if (fail_phase == 3 && rc == MDBG_OK) rc = set_error(ctx, MDBG_EHIP, "mdbg_shard_exchange: test failure in the reduction");
    MDBG_TRY(agree(ctx, api, comm, rc, "summing the rows it owns"));
    MDBG_DBG(ctx, "shard_exchange: reduced");
    // ---- replies back, transposed sizes: what came from rank r returns to rank r, in the order it was sent ----
    {
        LaunchTimer timer(ctx, "shard_exchange");
        if (got[me]) {
            MDBG_HIP_CHECK(ctx, hipMemcpyAsync(comm->replies.p + soff[me], d_reply + roff[me], got[me] * 8, hipMemcpyDeviceToDevice, ctx->stream));
            comm->bytes_local += got[me] * 8;
        }
        MDBG_TRY(all_to_all(ctx, api, comm, d_reply, roff, got, comm->replies.p, soff, s_cnt));
    }
    if (ctx->test_corrupt_replies && n_sent) {      // tests: the global count of one key this rank was told to LIST is off by one
        ctx->test_corrupt_replies = false;          // (a wrong count for a key another rank lists never reaches a table)
        std::vector<uint64_t> h(n_sent);
        MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, h.data(), comm->replies.p, n_sent * 8, hipMemcpyDeviceToHost));
        for (uint64_t i = 0; i < n_sent; i++)
            if (h[i] >> 63) {
                h[i] += 1;
                MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, comm->replies.p + i, &h[i], 8, hipMemcpyHostToDevice));
                break;
            }
    }
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    MDBG_DBG(ctx, "shard_exchange: done");
    comm->n_exchanges++;
    comm->exchange_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
    *d_replies = comm->replies.p;
    return MDBG_OK;
}

}  // namespace

extern "C" int mdbg_shard_exchange(mdbg_ctx *ctx, mdbg_comm *comm, mdbg_shard *shard, const uint64_t *d_rows, const uint64_t *counts,
                                   const uint64_t **d_replies) try {
    if (!ctx || !comm) return set_error(ctx, MDBG_EINVAL, "mdbg_shard_exchange: null context or communicator");
    // a bad argument is this rank's failure, and the peers must hear of it: it takes part in the first collective with its
    // error code instead of returning while they wait
    int local_rc = MDBG_OK;
    if (!shard || !counts || !d_replies) local_rc = set_error(ctx, MDBG_EINVAL, "mdbg_shard_exchange: null argument");
    static const uint64_t no_counts[64] = {0};
    return exchange_impl(ctx, comm, shard, d_rows, counts ? counts : no_counts, d_replies, local_rc);
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_shard_abort(mdbg_ctx *ctx, mdbg_comm *comm, int code) try {
    if (!ctx || !comm) return set_error(ctx, MDBG_EINVAL, "mdbg_shard_abort: null context or communicator");
    static const uint64_t no_counts[64] = {0};
    const std::string keep = ctx->err;              // the message of the failure being reported stays the context's last error
    const int rc = exchange_impl(ctx, comm, nullptr, nullptr, no_counts, nullptr, code < 0 ? code : MDBG_EINVAL);
    const bool told = rc == (code < 0 ? code : MDBG_EINVAL);
    if (told) ctx->err = keep;
    return told ? MDBG_OK : rc;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_kminmer_count_first_sharded(mdbg_ctx *ctx, mdbg_comm *comm, const mdbg_minimizers *reads, uint32_t k,
                                                uint32_t min_abundance, mdbg_table **out) try {
    if (!ctx || !comm) return set_error(ctx, MDBG_EINVAL, "mdbg_kminmer_count_first_sharded: null context or communicator");
    mdbg_shard *sh_raw = nullptr;
    const uint64_t *d_rows = nullptr, *d_replies = nullptr;
    std::vector<uint64_t> sent((size_t)comm->n_ranks, 0);
    int rc = (!reads || !out) ? set_error(ctx, MDBG_EINVAL, "mdbg_kminmer_count_first_sharded: null argument")
                              : mdbg_shard_begin(ctx, reads, k, (uint32_t)comm->n_ranks, &sh_raw, &d_rows, sent.data());
    std::unique_ptr<mdbg_shard, void (*)(mdbg_shard *)> sh(sh_raw, mdbg_shard_free);
    if (rc != MDBG_OK) {            // the local half failed: the peers are about to enter the exchange and must not wait for this rank
        (void)mdbg_shard_abort(ctx, comm, rc);
        return rc;
    }
    MDBG_TRY(mdbg_shard_exchange(ctx, comm, sh.get(), d_rows, sent.data(), &d_replies));
    return mdbg_shard_finish(ctx, sh.get(), d_replies, min_abundance, out);
} MDBG_API_CATCH(ctx)
