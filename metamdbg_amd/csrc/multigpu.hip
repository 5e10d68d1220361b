// multigpu.hip -- the exchange of the sharded passes inside the library, two transports behind one call (mdbg_shard_exchange):
//   * RCCL point-to-point over xGMI (ncclSend / ncclRecv groups), and
//   * PEER COPIES: every rank stages its rows, every owner pulls its slices with device-to-device copies (one stream per peer: the
//     seven links of a GPU carry their slices at once), the hand-shakes are words in host memory the ranks share (peerlink.hpp) --
//     no collective kernel that has to find room on a compute unit beside another batch's scan, no host collective of another
//     library.  Ranks are processes (staging buffers shared by hipIpcGetMemHandle / hipIpcOpenMemHandle) or threads of one
//     process (mdbg_tool graph --gpus G: plain pointers, peer access between the devices).
// mdbg_comm_create_mode selects; "auto" takes peer copies after a self-test every rank passed, RCCL otherwise.
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
#include "peerlink.hpp"

#include <algorithm>
#include <chrono>
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <functional>
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

namespace mdbg {

// ---- peer copies: what a rank keeps besides the shared control block ----
struct PeerOwn {                 // a staging buffer of this rank (plain hipMalloc: it must outlive the peers' mappings, not a pool block)
    void *p = nullptr;
    size_t cap = 0;
    uint64_t generation = 0;
    hipIpcMemHandle_t handle{};
};
struct PeerView {                // a peer's staging buffer as this rank reaches it
    uint64_t generation = 0;
    void *p = nullptr;
    bool opened = false;         // through hipIpcOpenMemHandle (another process): to be closed
    std::vector<void *> stale;   // mappings of buffers the peer has replaced since: closed with the communicator, not in the middle of a job
};
struct PeerLink {
    PeerCtl ctl;
    PeerOwn rows, replies;
    std::vector<PeerView> v_rows, v_replies;
    std::vector<hipStream_t> streams;        // one per peer: pulls from different peers run at once
    std::vector<hipEvent_t> events;
    std::vector<void *> retired;             // staging buffers this rank has replaced by larger ones: freed with the communicator (see exchange_peer, phase 3)
    DevBuf<uint64_t> test_reply;             // the self-test's replies
    uint64_t exchange = 0;
    double timeout_s = 120.0, late_s = 0.0;
};

}  // namespace mdbg

struct mdbg_comm {
    ncclComm_t comm = nullptr;
    int mode = MDBG_COMM_RCCL;   // the transport in use: MDBG_COMM_RCCL or MDBG_COMM_PEER
    std::unique_ptr<mdbg::PeerLink> link;    // MDBG_COMM_PEER
    std::string fallback_note;   // "auto" that ended on RCCL: why the peer copies were not taken
    int device = 0;
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
    // ... of which (peer copies): the owner's reduction (mdbg_shard_reduce: device work of this rank, with its own waits) and the waits
    // for this rank's copies and for its peers' phases; what is left is the transport's own host time (mdbg_comm_times)
    double reduce_ms = 0.0, wait_ms = 0.0;
};

namespace {
struct MsInto {                  // adds the scope's wall time to a counter
    double &to;
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    explicit MsInto(double &d) : to(d) {}
    ~MsInto() { to += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

#define MDBG_NCCL_CHECK(ctx, api, expr)                                                                                   \
    do {                                                                                                                  \
        ncclResult_t r_ = (expr);                                                                                         \
        if (r_ != ncclSuccess)                                                                                            \
            return set_error((ctx), MDBG_EHIP, "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

extern "C" int mdbg_comm_unique_id(uint8_t *id128) {
    if (!id128) return MDBG_EINVAL;
    RcclApi *api = rccl_api();
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    if (api->error.empty()) {
        ncclUniqueId id;
        ncclResult_t r = api->GetUniqueId(&id);
        if (r != ncclSuccess) { g_last_error = std::string("ncclGetUniqueId failed: ") + api->GetErrorString(r); return MDBG_EHIP; }
        memcpy(id128, &id, sizeof id);
        return MDBG_OK;
    }
    // no RCCL in this process: 128 random bytes name a communicator of the peer-copy transport just as well (MDBG_COMM_PEER;
    // mdbg_comm_create_mode with MDBG_COMM_RCCL or a fallback to it then fails with the loader's message)
    FILE *f = fopen("/dev/urandom", "rb");
    const size_t got = f ? fread(id128, 1, 128, f) : 0;
    if (f) fclose(f);
    if (got != 128) { g_last_error = api->error + "; and /dev/urandom gave no id"; return MDBG_ENODEV; }
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

extern "C" int mdbg_comm_adopt(mdbg_ctx *ctx, void *nccl_comm, int rank, int n_ranks, mdbg_comm **out) try {
    if (!ctx || !nccl_comm || !out || n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks)
        return set_error(ctx, MDBG_EINVAL, "mdbg_comm_adopt: bad argument (ranks 1..64)");
    RcclApi *api = rccl_api();
    if (!api->error.empty()) return set_error(ctx, MDBG_ENODEV, "%s", api->error.c_str());
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<mdbg_comm, void (*)(mdbg_comm *)> c(new mdbg_comm(), mdbg_comm_destroy);
    c->comm = (ncclComm_t)nccl_comm; c->rank = rank; c->n_ranks = n_ranks; c->owned = false; c->device = ctx->device;
    MDBG_TRY(comm_finish_setup(ctx, api, c.get(), "mdbg_comm_adopt"));
    *out = c.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

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

int exchange_rccl(mdbg_ctx *ctx, mdbg_comm *comm, mdbg_shard *shard, const uint64_t *d_rows, const uint64_t *counts, const uint64_t **d_replies,
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


// =====================================================================================================================
// The peer-copy transport
// =====================================================================================================================
using ReduceFn = std::function<int(const uint64_t *d_recv, uint64_t n_recv, const uint64_t **d_reply)>;

double peer_timeout_s() {
    if (const char *e = getenv("MDBG_PEER_TIMEOUT_S")) { const double v = atof(e); if (v > 0) return v; }
    return 120.0;
}

// this rank's staging buffer holds at least `bytes`; a replaced buffer stays alive until the communicator goes (L->retired)
int peer_grow(mdbg_ctx *ctx, PeerLink *L, PeerOwn &own, size_t bytes) {
    if (bytes <= own.cap) return MDBG_OK;
    const size_t cap = bytes + bytes / 4 + (1u << 20);
    void *p = nullptr;
    hipError_t e;
    { std::lock_guard<std::mutex> g(hip_mem_mutex()); e = hipMalloc(&p, cap); }
    if (e != hipSuccess) { (void)hipGetLastError(); ctx->pool->trim(); std::lock_guard<std::mutex> g(hip_mem_mutex()); e = hipMalloc(&p, cap); }
    if (e != hipSuccess) { (void)hipGetLastError(); return set_error(ctx, MDBG_ENOMEM, "peer-copy staging buffer of %zu bytes: %s", cap, hipGetErrorString(e)); }
    hipIpcMemHandle_t h{};
    if (L->ctl.n_ranks() > 1) {
        std::lock_guard<std::mutex> g(hip_mem_mutex());
        e = hipIpcGetMemHandle(&h, p);
        if (e != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p); return set_error(ctx, MDBG_EHIP, "hipIpcGetMemHandle: %s", hipGetErrorString(e)); }
    }
    if (own.p) L->retired.push_back(own.p);
    own.p = p; own.cap = cap; own.handle = h; own.generation++;
    return MDBG_OK;
}

void peer_publish(const PeerOwn &own, PeerBufWords &w) {
    static_assert(sizeof(hipIpcMemHandle_t) == sizeof(w.handle), "hipIpcMemHandle_t is 64 bytes");
    w.generation = own.generation; w.pointer = (uint64_t)(uintptr_t)own.p; w.capacity = own.cap;
    memcpy(w.handle, &own.handle, sizeof w.handle);
}

// (a close that fails must not leave its code behind as the thread's "last error": the kernels launched next -- the owner's
// reduction -- are followed by hipGetLastError() checks that would report it as theirs)
void peer_close(void *p) {
    std::lock_guard<std::mutex> g(hip_mem_mutex());
    if (p && hipIpcCloseMemHandle(p) != hipSuccess) (void)hipGetLastError();
}

// the communicator is going: every mapping of a peer's memory this rank still holds
void peer_unmap(PeerView &v) {
    if (v.opened) peer_close(v.p);
    for (void *p : v.stale) peer_close(p);
    v = PeerView();
}

// the peer has replaced the buffer this view shows.  Nothing is unmapped and nothing is freed while a job runs (see exchange_peer):
// the old mapping is set aside until the communicator goes
void peer_set_aside(PeerView &v) {
    if (v.opened && v.p) v.stale.push_back(v.p);
    v.generation = 0; v.p = nullptr; v.opened = false;
}

// rank r's staging buffer as published in `w`, reachable from this rank's device
// drawn once per process (time, pid and the address of a static mixed: two processes do not meet in it)
uint64_t process_token() {
    static const uint64_t token = [] {
        struct timespec ts;
        clock_gettime(CLOCK_REALTIME, &ts);
        uint64_t x = (uint64_t)ts.tv_sec * 1000000007ull ^ (uint64_t)ts.tv_nsec << 20 ^ (uint64_t)getpid() << 44 ^ (uint64_t)(uintptr_t)&ts;
        FILE *f = fopen("/dev/urandom", "rb");
        if (f) { uint64_t r = 0; if (fread(&r, 8, 1, f) == 1) x ^= r; fclose(f); }
        x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33;
        return x ? x : 1;
    }();
    return token;
}

int peer_map(mdbg_ctx *ctx, PeerLink *L, int r, const PeerBufWords &w, PeerView &v, void **out) {
    if (v.generation == w.generation && v.p) { *out = v.p; return MDBG_OK; }
    peer_set_aside(v);
    if (w.generation == 0 || w.pointer == 0) return set_error(ctx, MDBG_EPEER, "rank %d published no staging buffer", r);
    PeerSlot *ps = L->ctl.slot(r);
    if (ps->pid == (int32_t)getpid() && ps->process_token == process_token()) {     // a thread of this process: its pointer is ours (one address space)
        if (ps->device != ctx->device) {
            hipError_t e = hipDeviceEnablePeerAccess(ps->device, 0);     // (without it the copy is staged by the runtime: slower, still right)
            if (e != hipSuccess) (void)hipGetLastError();
        }
        v.p = (void *)(uintptr_t)w.pointer; v.opened = false;
    } else {
        hipIpcMemHandle_t h;
        memcpy(&h, w.handle, sizeof h);
        void *p = nullptr;
        hipError_t e;
        { std::lock_guard<std::mutex> g(hip_mem_mutex()); e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess); }
        if (e != hipSuccess) { (void)hipGetLastError(); return set_error(ctx, MDBG_EHIP, "hipIpcOpenMemHandle (staging buffer of rank %d, pid %d, device %d): %s", r, ps->pid, ps->device, hipGetErrorString(e)); }
        v.p = p; v.opened = true;
    }
    v.generation = w.generation;
    *out = v.p;
    return MDBG_OK;
}

// One phase of an exchange: this rank says `rc` (its words for the phase are written), waits for everybody, reads what they said.
// MDBG_OK: all can go on.  Otherwise this rank's own code (message already set), MDBG_EPEER naming the first rank that failed, or
// MDBG_EPEER for a rank that did not arrive in time / has left (the communicator is then broken: nobody can tell how far it came).
int peer_phase(mdbg_ctx *ctx, mdbg_comm *comm, uint64_t E, int phase, int rc, const char *what) {
    PeerLink *L = comm->link.get();
    L->ctl.words(comm->rank, E)->status[phase] = rc;
    L->ctl.arrive(PeerCtl::tick_of(E, phase));
    int late;
    { MsInto w(comm->wait_ms); late = L->ctl.wait_all(PeerCtl::tick_of(E, phase), L->timeout_s, L->late_s); }
    if (late >= 0) {
        comm->broken = true;
        const std::string own = rc != MDBG_OK ? ctx->err : "";
        set_error(ctx, MDBG_EPEER, "mdbg_shard_exchange (peer copies): rank %d did not arrive %s within %.0f s (MDBG_PEER_TIMEOUT_S)%s%s", late, what, L->timeout_s,
                  own.empty() ? "" : "; this rank had failed: ", own.c_str());
        return rc != MDBG_OK ? rc : MDBG_EPEER;
    }
    if (rc != MDBG_OK) return rc;
    for (int r = 0; r < comm->n_ranks; r++) {
        if (L->ctl.left_before(r, PeerCtl::tick_of(E, phase))) {
            comm->broken = true;
            return set_error(ctx, MDBG_EPEER, "mdbg_shard_exchange (peer copies): rank %d has left the communicator", r);
        }
        const int64_t st = L->ctl.words(r, E)->status[phase];
        if (st != 0)
            return set_error(ctx, MDBG_EPEER, "mdbg_shard_exchange: rank %d failed %s (code %lld); nothing was exchanged", r, what, (long long)st);
    }
    return MDBG_OK;
}

int exchange_peer(mdbg_ctx *ctx, mdbg_comm *comm, const uint64_t *d_rows, const uint64_t *counts, const uint64_t **d_replies, int local_rc,
                  const ReduceFn &reduce) {
    PeerLink *L = comm->link.get();
    if (comm->broken) return set_error(ctx, MDBG_EHIP, "mdbg_shard_exchange: an earlier exchange failed on this communicator (a rank did not arrive or has left)");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int n = comm->n_ranks, me = comm->rank;
    const uint32_t rw = mdbg_row_words(4);
    const auto t_enter = std::chrono::steady_clock::now();
    const uint64_t E = ++L->exchange;
    PeerWords *mine = L->ctl.words(me, E);
    MDBG_DBG(ctx, "shard_exchange (peer copies): enter, %d ranks, exchange %llu", n, (unsigned long long)E);

    const int fail_phase = ctx->test_exchange_fail_phase;       // tests: one-shot local failures
    ctx->test_exchange_fail_phase = 0;
    if (fail_phase == 1 && local_rc == MDBG_OK) local_rc = set_error(ctx, MDBG_EINVAL, "mdbg_shard_exchange: test failure before the counts");
    // ---- phase 0: who sends how many rows to whom, where they will be staged (and whether everybody got this far) ----
    uint64_t n_sent = 0;
    if (local_rc == MDBG_OK) {
        for (int r = 0; r < n; r++) n_sent += counts[r];
        if (n_sent && !d_rows) local_rc = set_error(ctx, MDBG_EINVAL, "mdbg_shard_exchange: null rows");
    }
    if (local_rc == MDBG_OK) local_rc = peer_grow(ctx, L, L->rows, (size_t)n_sent * rw * 8);
    for (int r = 0; r < n; r++) mine->counts[r] = local_rc == MDBG_OK ? counts[r] : 0;
    peer_publish(L->rows, mine->rows);
    MDBG_TRY(peer_phase(ctx, comm, E, 0, local_rc, "before the exchange"));
    // m(r, d) = rows rank r holds for owner d
    auto m = [&](int r, int d) { return L->ctl.words(r, E)->counts[d]; };
    std::vector<uint64_t> s_cnt((size_t)n), got((size_t)n), soff((size_t)n + 1, 0), roff((size_t)n + 1, 0);
    for (int r = 0; r < n; r++) {
        s_cnt[r] = counts[r];
        got[r] = m(r, me);
        soff[r + 1] = soff[r] + s_cnt[r];
        roff[r + 1] = roff[r] + got[r];
    }
    const uint64_t n_recv = roff[n];
    MDBG_DBG(ctx, "shard_exchange: counts known, %llu rows out, %llu in", (unsigned long long)n_sent, (unsigned long long)n_recv);

    // ---- phase 1: my rows are staged, my receive buffers exist ----
    DevBuf<uint64_t> d_recv;
    int rc = d_recv.alloc(ctx, n_recv * rw);
    if (rc == MDBG_OK) rc = comm->replies.alloc(ctx, n_sent);
    if (rc == MDBG_OK) rc = peer_grow(ctx, L, L->replies, (size_t)n_recv * 8);
    if (fail_phase == 2 && rc == MDBG_OK) rc = set_error(ctx, MDBG_ENOMEM, "mdbg_shard_exchange: test failure allocating the receive buffers");
    if (rc == MDBG_OK && n_sent - s_cnt[me] > 0) {
        // (the rows for this rank itself stay where they are: they are copied straight into the receive buffer below)
        hipError_t e = hipSuccess;
        if (soff[me]) e = hipMemcpyAsync(L->rows.p, d_rows, soff[me] * rw * 8, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess && n_sent > soff[me + 1])
            e = hipMemcpyAsync((uint64_t *)L->rows.p + soff[me + 1] * rw, d_rows + soff[me + 1] * rw, (n_sent - soff[me + 1]) * rw * 8, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) { MsInto w(comm->wait_ms); e = hipStreamSynchronize(ctx->stream); }
        if (e != hipSuccess) rc = set_error(ctx, MDBG_EHIP, "staging the rows: %s", hipGetErrorString(e));
    }
    peer_publish(L->replies, mine->replies);
    MDBG_TRY(peer_phase(ctx, comm, E, 1, rc, "allocating the receive buffers"));

    // ---- rows to their owners: every owner pulls its slice from every sender, one stream per peer ----
    rc = MDBG_OK;
    {
        LaunchTimer timer(ctx, "shard_exchange");
        hipError_t e = hipSuccess;
        if (s_cnt[me]) {
            e = hipMemcpyAsync(d_recv.p + roff[me] * rw, d_rows + soff[me] * rw, s_cnt[me] * rw * 8, hipMemcpyDeviceToDevice, ctx->stream);
            comm->bytes_local += s_cnt[me] * rw * 8;
        }
        for (int r = 0; r < n && e == hipSuccess && rc == MDBG_OK; r++) {
            if (r == me || got[r] == 0) continue;
            void *src = nullptr;
            rc = peer_map(ctx, L, r, L->ctl.words(r, E)->rows, L->v_rows[r], &src);
            if (rc != MDBG_OK) break;
            uint64_t first = 0;                                  // my slice in rank r's staging buffer: behind what it holds for lower owners
            for (int d = 0; d < me; d++) first += m(r, d);
            e = hipMemcpyAsync(d_recv.p + roff[r] * rw, (const uint64_t *)src + first * rw, got[r] * rw * 8, hipMemcpyDeviceToDevice, L->streams[r]);
            if (e == hipSuccess) e = hipEventRecord(L->events[r], L->streams[r]);
            if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, L->events[r], 0);
            comm->bytes_from_peers += got[r] * rw * 8;
            comm->bytes_to_peers += s_cnt[r] * rw * 8;           // (what the peers pull from this rank: the same matrix read the other way)
        }
        if (e != hipSuccess && rc == MDBG_OK) rc = set_error(ctx, MDBG_EHIP, "pulling the rows: %s", hipGetErrorString(e));
    }
    for (int r = 0; r < n; r++)                                  // the senders that staged rows for others but none for this rank still count
        if (r != me && got[r] == 0) comm->bytes_to_peers += s_cnt[r] * rw * 8;
    // ---- the owner sums and answers; the replies are staged ----
    const uint64_t *d_reply = nullptr;
    if (rc == MDBG_OK) { MsInto w(comm->reduce_ms); rc = reduce(d_recv.p, n_recv, &d_reply); }
    if (fail_phase == 3 && rc == MDBG_OK) rc = set_error(ctx, MDBG_EHIP, "mdbg_shard_exchange: test failure in the reduction");
    if (rc == MDBG_OK) {
        hipError_t e = hipSuccess;
        if (n_recv - got[me] > 0) e = hipMemcpyAsync(L->replies.p, d_reply, n_recv * 8, hipMemcpyDeviceToDevice, ctx->stream);
        if (e == hipSuccess) { MsInto w(comm->wait_ms); e = hipStreamSynchronize(ctx->stream); }          // (also: every pull of this rank has landed)
        if (e != hipSuccess) rc = set_error(ctx, MDBG_EHIP, "staging the replies: %s", hipGetErrorString(e));
    } else {
        (void)hipStreamSynchronize(ctx->stream);
    }
    MDBG_TRY(peer_phase(ctx, comm, E, 2, rc, "summing the rows it owns"));
    MDBG_DBG(ctx, "shard_exchange: reduced");

    // ---- replies back: every sender pulls, from every owner, the answers to the rows it sent there, in the order it sent them ----
    // (what goes wrong here is told at phase 3 like everything else: nobody is left waiting there for a rank that has returned)
    rc = MDBG_OK;
    {
        LaunchTimer timer(ctx, "shard_exchange");
        hipError_t e = hipSuccess;
        if (got[me]) {
            e = hipMemcpyAsync(comm->replies.p + soff[me], d_reply + roff[me], got[me] * 8, hipMemcpyDeviceToDevice, ctx->stream);
            comm->bytes_local += got[me] * 8;
        }
        for (int d = 0; d < n && e == hipSuccess && rc == MDBG_OK; d++) {
            if (d == me || s_cnt[d] == 0) continue;
            void *src = nullptr;
            rc = peer_map(ctx, L, d, L->ctl.words(d, E)->replies, L->v_replies[d], &src);
            if (rc != MDBG_OK) break;
            uint64_t first = 0;                                  // owner d received the ranks' rows in rank order
            for (int r = 0; r < me; r++) first += m(r, d);
            e = hipMemcpyAsync(comm->replies.p + soff[d], (const uint64_t *)src + first, s_cnt[d] * 8, hipMemcpyDeviceToDevice, L->streams[d]);
            if (e == hipSuccess) e = hipEventRecord(L->events[d], L->streams[d]);
            if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, L->events[d], 0);
            comm->bytes_from_peers += s_cnt[d] * 8;
            comm->bytes_to_peers += got[d] * 8;
        }
        if (e != hipSuccess && rc == MDBG_OK) rc = set_error(ctx, MDBG_EHIP, "pulling the replies: %s", hipGetErrorString(e));
    }
    for (int d = 0; d < n; d++)
        if (d != me && s_cnt[d] == 0) comm->bytes_to_peers += got[d] * 8;
    {
        hipError_t e;
        { MsInto w(comm->wait_ms); e = hipStreamSynchronize(ctx->stream); }          // every pull of this rank has landed
        if (e != hipSuccess && rc == MDBG_OK) rc = set_error(ctx, MDBG_EHIP, "pulling the replies: %s", hipGetErrorString(e));
    }
    if (rc == MDBG_OK && ctx->test_corrupt_replies && n_sent) {      // tests: the global count of one key this rank was told to LIST is off by one
        ctx->test_corrupt_replies = false;
        std::vector<uint64_t> h(n_sent);
        hipError_t e = memcpy_sync(ctx, h.data(), comm->replies.p, n_sent * 8, hipMemcpyDeviceToHost);
        for (uint64_t i = 0; i < n_sent && e == hipSuccess; i++)
            if (h[i] >> 63) {
                h[i] += 1;
                e = memcpy_sync(ctx, comm->replies.p + i, &h[i], 8, hipMemcpyHostToDevice);
                break;
            }
        if (e != hipSuccess) rc = set_error(ctx, MDBG_EHIP, "test_corrupt_replies: %s", hipGetErrorString(e));
    }
    // ---- phase 3: everybody has pulled its replies.  Until here a peer may still be reading this rank's staging buffers; behind it
    // nobody reads anything of this exchange any more, whatever a rank does next (the next exchange, another communicator's, leaving).
    // Replaced staging buffers (L->retired) and the peers' mappings of theirs (PeerView::stale) are NOT released on the way: a staging
    // buffer grows a handful of times in a job's life (by a quarter each time), and unmapping or freeing memory other processes have
    // mapped, in the middle of a run, buys nothing worth the ways it can go wrong -- they go with the communicator (peer_teardown).
    MDBG_TRY(peer_phase(ctx, comm, E, 3, rc, "pulling its replies"));
    MDBG_DBG(ctx, "shard_exchange: done");
    comm->n_exchanges++;
    comm->exchange_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count();
    *d_replies = comm->replies.p;
    return MDBG_OK;
}

void peer_teardown(mdbg_comm *c) {
    PeerLink *L = c->link.get();
    if (!L) return;
    (void)hipSetDevice(c->device);
    for (auto &v : L->v_rows) peer_unmap(v);
    for (auto &v : L->v_replies) peer_unmap(v);
    if (L->ctl.attached()) {
        // nobody frees a buffer a peer may still have mapped: say "closing" (after unmapping theirs), wait for the others to say it too
        L->ctl.arrive(PeerCtl::TICK_CLOSING);
        (void)L->ctl.wait_all(PeerCtl::TICK_CLOSING, c->broken ? 1.0 : 10.0);
    }
    for (hipStream_t s : L->streams) if (s) (void)hipStreamDestroy(s);
    for (hipEvent_t e : L->events) if (e) (void)hipEventDestroy(e);
    {
        std::lock_guard<std::mutex> g(hip_mem_mutex());
        for (void *r : L->retired) (void)hipFree(r);
        if (L->rows.p) (void)hipFree(L->rows.p);
        if (L->replies.p) (void)hipFree(L->replies.p);
    }
    L->test_reply.release();
    L->ctl.detach();
    c->link.reset();
}

// The control block, the identities, the per-peer streams.  Collective.
int peer_setup(mdbg_ctx *ctx, mdbg_comm *c, const uint8_t *id128) {
    c->link.reset(new PeerLink());
    PeerLink *L = c->link.get();
    L->timeout_s = peer_timeout_s();
    // a rank that is late but ALIVE is waited for this long in all (MDBG_PEER_LATE_S, default an hour; 0: the deadline is the deadline)
    L->late_s = 3600.0;
    if (const char *e = getenv("MDBG_PEER_LATE_S")) L->late_s = atof(e);
    double setup_s = 30.0;
    if (const char *e = getenv("MDBG_PEER_SETUP_TIMEOUT_S")) { const double v = atof(e); if (v > 0) setup_s = v; }
    // (tests: MDBG_PEER_TEST_NO_SHM -- rank 1 cannot reach the control block, as when /dev/shm is not shared between the ranks' containers)
    const std::string err = (getenv("MDBG_PEER_TEST_NO_SHM") && c->rank == 1) ? std::string("test: the control block is out of reach")
                                                                             : L->ctl.attach(PeerCtl::name_for(id128), c->rank, c->n_ranks, setup_s);
    if (!err.empty()) return set_error(ctx, MDBG_EHIP, "peer copies: %s", err.c_str());
    PeerSlot *ps = L->ctl.mine();
    ps->pid = (int32_t)getpid();
    ps->device = ctx->device;
    ps->process_token = process_token();
    L->ctl.arrive(PeerCtl::TICK_ATTACHED);
    const int late = L->ctl.wait_all(PeerCtl::TICK_ATTACHED, setup_s);
    L->ctl.unlink_name();
    if (late >= 0) return set_error(ctx, MDBG_EPEER, "peer copies: rank %d did not attach to the control block within %.0f s", late, setup_s);
    L->v_rows.resize((size_t)c->n_ranks); L->v_replies.resize((size_t)c->n_ranks);
    L->streams.assign((size_t)c->n_ranks, nullptr); L->events.assign((size_t)c->n_ranks, nullptr);
    for (int r = 0; r < c->n_ranks; r++) {
        if (r == c->rank) continue;
        MDBG_HIP_CHECK(ctx, hipStreamCreateWithFlags(&L->streams[r], hipStreamNonBlocking));
        MDBG_HIP_CHECK(ctx, hipEventCreateWithFlags(&L->events[r], hipEventDisableTiming));
    }
    return MDBG_OK;
}

__global__ void peer_test_reply_kernel(const uint64_t *rows, uint64_t n, uint32_t rw, uint64_t *reply) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) reply[i] = rows[i * rw] * 1000003ull + rows[i * rw + 2];
}

// A small exchange of known rows through the very buffers, copies and hand-shakes a job will use; then every rank says whether what
// came back was right, so that all of them take the same decision ("auto": peer copies or RCCL).  Collective.
int peer_self_test(mdbg_ctx *ctx, mdbg_comm *c, bool ask_the_runtime_first) {
    PeerLink *L = c->link.get();
    const int n = c->n_ranks, me = c->rank;
    const uint32_t rw = mdbg_row_words(4);
    // "auto" (round-5 ADVICE): a pull from a device this one cannot address ends in a GPU memory fault -- the process is gone, nothing falls
    // back.  So before the first byte is pulled every rank asks the runtime whether its device reaches every other rank's
    // (hipDeviceCanAccessPeer; ranks on one device need not ask); a rank that hears "no" enters the test with that as its failure, and
    // the test's phases carry it to all the others: they fall back to RCCL together.  A forced MDBG_COMM_PEER does not ask.
    int can_rc = MDBG_OK;
    if (ask_the_runtime_first)
        for (int r = 0; r < n && can_rc == MDBG_OK; r++) {
            const int dev = L->ctl.slot(r)->device;
            if (r == me || dev == ctx->device) continue;
            int can = 0;
            const hipError_t e = hipDeviceCanAccessPeer(&can, ctx->device, dev);
            if (e != hipSuccess || !can) {
                (void)hipGetLastError();
                can_rc = set_error(ctx, MDBG_EHIP, "peer copies: device %d (rank %d) cannot address device %d (rank %d) directly (hipDeviceCanAccessPeer)", ctx->device, me, dev, r);
            }
        }
    auto cnt = [&](int r, int d) { return (uint64_t)(3 + ((r + d) % 5)); };
    std::vector<uint64_t> counts((size_t)n), h;
    uint64_t total = 0;
    for (int d = 0; d < n; d++) { counts[d] = cnt(me, d); total += counts[d]; }
    h.assign((size_t)total * rw, 0);
    uint64_t at = 0;
    for (int d = 0; d < n; d++)
        for (uint64_t i = 0; i < counts[d]; i++, at++) { h[at * rw] = (uint64_t)me + 1; h[at * rw + 1] = (uint64_t)d + 1; h[at * rw + 2] = at; }
    DevBuf<uint64_t> d_rows;
    int rc = can_rc;
    if (rc == MDBG_OK) rc = d_rows.alloc(ctx, h.size());
    if (rc == MDBG_OK && memcpy_sync(ctx, d_rows.p, h.data(), h.size() * 8, hipMemcpyHostToDevice) != hipSuccess) rc = set_error(ctx, MDBG_EHIP, "peer-copy self-test: upload failed");
    bool rows_right = true;
    const ReduceFn check = [&](const uint64_t *d_recv, uint64_t n_recv, const uint64_t **d_reply) -> int {
        std::vector<uint64_t> got((size_t)n_recv * rw);
        MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, got.data(), d_recv, got.size() * 8, hipMemcpyDeviceToHost));
        uint64_t i = 0;
        for (int r = 0; r < n; r++) {
            uint64_t first = 0;
            for (int d = 0; d < me; d++) first += cnt(r, d);
            for (uint64_t j = 0; j < cnt(r, me); j++, i++)
                rows_right = rows_right && i < n_recv && got[i * rw] == (uint64_t)r + 1 && got[i * rw + 1] == (uint64_t)me + 1 && got[i * rw + 2] == first + j;
        }
        rows_right = rows_right && i == n_recv;
        MDBG_TRY(L->test_reply.alloc(ctx, n_recv));
        hipLaunchKernelGGL(peer_test_reply_kernel, dim3(grid_for(n_recv, 256)), dim3(256), 0, ctx->stream, d_recv, n_recv, rw, L->test_reply.p);
        *d_reply = L->test_reply.p;
        return MDBG_OK;
    };
    const uint64_t *d_back = nullptr;
    rc = exchange_peer(ctx, c, d_rows.p, counts.data(), &d_back, rc, check);
    if (rc == MDBG_OK) {
        std::vector<uint64_t> back((size_t)total);
        if (memcpy_sync(ctx, back.data(), d_back, total * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = set_error(ctx, MDBG_EHIP, "peer-copy self-test: download failed");
        bool good = rows_right;
        for (uint64_t i = 0; i < total && good; i++) good = back[i] == ((uint64_t)me + 1) * 1000003ull + i;
        if (rc == MDBG_OK && !good) rc = set_error(ctx, MDBG_EHIP, "peer-copy self-test: %s", rows_right ? "a reply came back wrong" : "a slice of rows arrived wrong");
    }
    if (c->broken) return rc != MDBG_OK ? rc : MDBG_EPEER;
    // the verdicts, agreed: one more (empty) exchange whose first phase carries them
    static const uint64_t none[PEER_MAX_RANKS] = {0};
    const std::string keep = ctx->err;
    const int said = rc;
    const uint64_t *unused = nullptr;
    rc = exchange_peer(ctx, c, nullptr, none, &unused, said, [](const uint64_t *, uint64_t, const uint64_t **r) { *r = nullptr; return MDBG_OK; });
    if (said != MDBG_OK) ctx->err = keep;
    c->n_exchanges = 0; c->bytes_to_peers = c->bytes_from_peers = c->bytes_local = 0;      // the job's account starts here
    c->exchange_ms = c->reduce_ms = c->wait_ms = 0.0;
    return rc;
}

}  // namespace

namespace {
int exchange_any(mdbg_ctx *ctx, mdbg_comm *comm, mdbg_shard *shard, const uint64_t *d_rows, const uint64_t *counts, const uint64_t **d_replies, int local_rc) {
    if (comm->mode == MDBG_COMM_PEER)
        return exchange_peer(ctx, comm, d_rows, counts, d_replies, local_rc, [&](const uint64_t *d_recv, uint64_t n_recv, const uint64_t **d_reply) {
            return mdbg_shard_reduce(ctx, shard, d_recv, n_recv, d_reply);
        });
    return exchange_rccl(ctx, comm, shard, d_rows, counts, d_replies, local_rc);
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
    return exchange_any(ctx, comm, shard, d_rows, counts ? counts : no_counts, d_replies, local_rc);
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_shard_abort(mdbg_ctx *ctx, mdbg_comm *comm, int code) try {
    if (!ctx || !comm) return set_error(ctx, MDBG_EINVAL, "mdbg_shard_abort: null context or communicator");
    static const uint64_t no_counts[64] = {0};
    const std::string keep = ctx->err;              // the message of the failure being reported stays the context's last error
    const int rc = exchange_any(ctx, comm, nullptr, nullptr, no_counts, nullptr, code < 0 ? code : MDBG_EINVAL);
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

// ---- communicators ---------------------------------------------------------------------------------------------------
namespace {
int rccl_setup(mdbg_ctx *ctx, mdbg_comm *c, const uint8_t *id128) {
    RcclApi *api = rccl_api();
    if (!api->error.empty()) return set_error(ctx, MDBG_ENODEV, "%s", api->error.c_str());
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    MDBG_NCCL_CHECK(ctx, api, api->CommInitRank(&c->comm, c->n_ranks, id, c->rank));
    return comm_finish_setup(ctx, api, c, "mdbg_comm_create");
}

int default_comm_mode() {
    const char *e = getenv("MDBG_COMM_MODE");
    if (!e || !*e) return MDBG_COMM_AUTO;
    const std::string v(e);
    if (v == "rccl") return MDBG_COMM_RCCL;
    if (v == "peer") return MDBG_COMM_PEER;
    return MDBG_COMM_AUTO;
}
}  // namespace

extern "C" int mdbg_comm_create_mode(mdbg_ctx *ctx, const uint8_t *id128, int rank, int n_ranks, int mode, mdbg_comm **out) try {
    if (!ctx || !id128 || !out || n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks)
        return set_error(ctx, MDBG_EINVAL, "mdbg_comm_create: bad argument (ranks 1..64)");
    if (mode == MDBG_COMM_DEFAULT) mode = default_comm_mode();
    if (mode != MDBG_COMM_RCCL && mode != MDBG_COMM_PEER && mode != MDBG_COMM_AUTO) return set_error(ctx, MDBG_EINVAL, "mdbg_comm_create_mode: unknown mode %d", mode);
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    std::unique_ptr<mdbg_comm, void (*)(mdbg_comm *)> c(new mdbg_comm(), mdbg_comm_destroy);
    c->rank = rank; c->n_ranks = n_ranks; c->owned = true; c->device = ctx->device;
    if (mode == MDBG_COMM_PEER || mode == MDBG_COMM_AUTO) {
        // every step of this is agreed among the ranks through the control block itself (a rank that cannot attach leaves the others
        // waiting for it until the set-up deadline: they all come out of it on the same side), so "auto" needs nothing from RCCL
        // unless the copies are not to be had
        c->mode = MDBG_COMM_PEER;
        int rc = peer_setup(ctx, c.get(), id128);
        if (rc == MDBG_OK) rc = peer_self_test(ctx, c.get(), mode == MDBG_COMM_AUTO);
        if (rc == MDBG_OK) { *out = c.release(); return MDBG_OK; }
        if (mode == MDBG_COMM_PEER) return rc;
        c->fallback_note = ctx->err;
        MDBG_DBG(ctx, "mdbg_comm_create (auto): no peer copies (%s); RCCL", ctx->err.c_str());
        peer_teardown(c.get());
        c->broken = false;
    }
    c->mode = MDBG_COMM_RCCL;
    MDBG_TRY(rccl_setup(ctx, c.get(), id128));
    *out = c.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_comm_create(mdbg_ctx *ctx, const uint8_t *id128, int rank, int n_ranks, mdbg_comm **out) {
    return mdbg_comm_create_mode(ctx, id128, rank, n_ranks, MDBG_COMM_DEFAULT, out);
}

extern "C" int mdbg_comm_mode(const mdbg_comm *c) { return c ? c->mode : MDBG_EINVAL; }

extern "C" int mdbg_comm_times(const mdbg_comm *c, double ms[3]) {
    if (!c || !ms) return MDBG_EINVAL;
    ms[0] = c->exchange_ms; ms[1] = c->reduce_ms; ms[2] = c->wait_ms;
    return MDBG_OK;
}

extern "C" const char *mdbg_comm_note(const mdbg_comm *c) { return c ? c->fallback_note.c_str() : ""; }

extern "C" void mdbg_comm_destroy(mdbg_comm *c) {
    if (!c) return;
    peer_teardown(c);
    // a communicator an RCCL call failed on is aborted, not destroyed: ncclCommDestroy waits for operations that will never finish
    if (c->owned && c->comm) (void)(c->broken ? rccl_api()->CommAbort(c->comm) : rccl_api()->CommDestroy(c->comm));
    delete c;
}
