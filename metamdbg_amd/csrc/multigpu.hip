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
    bool owned = false;          // created by mdbg_comm_create (destroyed with the handle) or adopted from the caller
    mdbg::DevBuf<uint64_t> replies;   // global counts for the rows of the last exchange (valid until the next one)
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

extern "C" int mdbg_comm_create(mdbg_ctx *ctx, const uint8_t *id128, int rank, int n_ranks, mdbg_comm **out) try {
    if (!ctx || !id128 || !out || n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks)
        return set_error(ctx, MDBG_EINVAL, "mdbg_comm_create: bad argument (ranks 1..64)");
    RcclApi *api = rccl_api();
    if (!api->error.empty()) return set_error(ctx, MDBG_ENODEV, "%s", api->error.c_str());
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    std::unique_ptr<mdbg_comm> c(new mdbg_comm());
    c->rank = rank; c->n_ranks = n_ranks; c->owned = true;
    MDBG_NCCL_CHECK(ctx, api, api->CommInitRank(&c->comm, n_ranks, id, rank));
    *out = c.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_comm_adopt(mdbg_ctx *ctx, void *nccl_comm, int rank, int n_ranks, mdbg_comm **out) try {
    if (!ctx || !nccl_comm || !out || n_ranks < 1 || n_ranks > 64 || rank < 0 || rank >= n_ranks)
        return set_error(ctx, MDBG_EINVAL, "mdbg_comm_adopt: bad argument (ranks 1..64)");
    RcclApi *api = rccl_api();
    if (!api->error.empty()) return set_error(ctx, MDBG_ENODEV, "%s", api->error.c_str());
    mdbg_comm *c = new mdbg_comm();
    c->comm = (ncclComm_t)nccl_comm; c->rank = rank; c->n_ranks = n_ranks; c->owned = false;
    *out = c;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" void mdbg_comm_destroy(mdbg_comm *c) {
    if (!c) return;
    if (c->owned && c->comm) (void)rccl_api()->CommDestroy(c->comm);
    delete c;
}

extern "C" int mdbg_shard_exchange(mdbg_ctx *ctx, mdbg_comm *comm, mdbg_shard *shard, const uint64_t *d_rows, const uint64_t *counts,
                                   const uint64_t **d_replies) try {
    if (!ctx || !comm || !shard || !counts || !d_replies) return set_error(ctx, MDBG_EINVAL, "mdbg_shard_exchange: null argument");
    RcclApi *api = rccl_api();
    if (!api->error.empty()) return set_error(ctx, MDBG_ENODEV, "%s", api->error.c_str());
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const int n = comm->n_ranks, me = comm->rank;
    const uint32_t rw = mdbg_row_words(4);     // the same for every k: vectors never travel
    MDBG_DBG(ctx, "shard_exchange: enter, %d ranks", n);

    // ---- who sends how many rows to whom: every rank's row of the count matrix (n u64 each) to every rank ----
    DevBuf<uint64_t> d_cnt, d_all;
    MDBG_TRY(d_cnt.alloc(ctx, (size_t)n));
    MDBG_TRY(d_all.alloc(ctx, (size_t)n * n));
    MDBG_HIP_CHECK(ctx, hipMemcpyAsync(d_cnt.p, counts, (size_t)n * 8, hipMemcpyHostToDevice, ctx->stream));
    MDBG_NCCL_CHECK(ctx, api, api->AllGather(d_cnt.p, d_all.p, (size_t)n, ncclUint64, comm->comm, ctx->stream));
    std::vector<uint64_t> all((size_t)n * n);
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, all.data(), d_all.p, all.size() * 8, hipMemcpyDeviceToHost));
    std::vector<uint64_t> got((size_t)n), soff((size_t)n + 1, 0), roff((size_t)n + 1, 0);
    for (int r = 0; r < n; r++) {
        got[r] = all[(size_t)r * n + me];          // rows rank r holds for me
        soff[r + 1] = soff[r] + counts[r];
        roff[r + 1] = roff[r] + got[r];
    }
    const uint64_t n_sent = soff[n], n_recv = roff[n];
    MDBG_DBG(ctx, "shard_exchange: counts known, %llu rows out, %llu in", (unsigned long long)n_sent, (unsigned long long)n_recv);
    if (n_sent && !d_rows) return set_error(ctx, MDBG_EINVAL, "mdbg_shard_exchange: null rows");

    // ---- rows to their owners ----
    DevBuf<uint64_t> d_recv;
    MDBG_TRY(d_recv.alloc(ctx, n_recv * rw));
    MDBG_TRY(comm->replies.alloc(ctx, n_sent));
    // A rank's own share never goes through RCCL: it is a device-to-device copy on the same stream.  (RCCL 2.26's send/receive
    // to self returned with 531 MiB of a 1.1 GB message in place, the rest still zero when the next kernel on the stream read it;
    // one rank on the per-rank workload of an 8-GPU job, profiles/r02z_*.)  Messages to peers go in pieces of at most 256 MiB,
    // the same cut on both sides.
    constexpr uint64_t PIECE = 1ull << 25;          // u64 elements
    auto move = [&](const uint64_t *src, uint64_t *dst, uint64_t n_send, uint64_t n_get, int r) -> int {
        for (uint64_t at = 0; at < n_send; at += PIECE)
            MDBG_NCCL_CHECK(ctx, api, api->Send(src + at, std::min(PIECE, n_send - at), ncclUint64, r, comm->comm, ctx->stream));
        for (uint64_t at = 0; at < n_get; at += PIECE)
            MDBG_NCCL_CHECK(ctx, api, api->Recv(dst + at, std::min(PIECE, n_get - at), ncclUint64, r, comm->comm, ctx->stream));
        return MDBG_OK;
    };
    {
        LaunchTimer timer(ctx, "shard_exchange");
        if (counts[me]) MDBG_HIP_CHECK(ctx, hipMemcpyAsync(d_recv.p + roff[me] * rw, d_rows + soff[me] * rw, counts[me] * rw * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (n > 1) {
            MDBG_NCCL_CHECK(ctx, api, api->GroupStart());
            for (int r = 0; r < n; r++)
                if (r != me) MDBG_TRY(move(d_rows + soff[r] * rw, d_recv.p + roff[r] * rw, counts[r] * rw, got[r] * rw, r));
            MDBG_NCCL_CHECK(ctx, api, api->GroupEnd());
        }
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
    MDBG_TRY(mdbg_shard_reduce(ctx, shard, d_recv.p, n_recv, &d_reply));
    MDBG_DBG(ctx, "shard_exchange: reduced");
    // ---- replies back, transposed sizes: what came from rank r returns to rank r, in the order it was sent ----
    {
        LaunchTimer timer(ctx, "shard_exchange");
        if (got[me]) MDBG_HIP_CHECK(ctx, hipMemcpyAsync(comm->replies.p + soff[me], d_reply + roff[me], got[me] * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (n > 1) {
            MDBG_NCCL_CHECK(ctx, api, api->GroupStart());
            for (int r = 0; r < n; r++)
                if (r != me) MDBG_TRY(move(d_reply + roff[r], comm->replies.p + soff[r], got[r], counts[r], r));
            MDBG_NCCL_CHECK(ctx, api, api->GroupEnd());
        }
    }
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    MDBG_DBG(ctx, "shard_exchange: done");
    *d_replies = comm->replies.p;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_kminmer_count_first_sharded(mdbg_ctx *ctx, mdbg_comm *comm, const mdbg_minimizers *reads, uint32_t k,
                                                uint32_t min_abundance, mdbg_table **out) try {
    if (!ctx || !comm || !reads || !out) return set_error(ctx, MDBG_EINVAL, "mdbg_kminmer_count_first_sharded: null argument");
    mdbg_shard *sh_raw = nullptr;
    const uint64_t *d_rows = nullptr, *d_replies = nullptr;
    std::vector<uint64_t> sent((size_t)comm->n_ranks, 0);
    MDBG_TRY(mdbg_shard_begin(ctx, reads, k, (uint32_t)comm->n_ranks, &sh_raw, &d_rows, sent.data()));
    std::unique_ptr<mdbg_shard, void (*)(mdbg_shard *)> sh(sh_raw, mdbg_shard_free);
    MDBG_TRY(mdbg_shard_exchange(ctx, comm, sh.get(), d_rows, sent.data(), &d_replies));
    return mdbg_shard_finish(ctx, sh.get(), d_replies, min_abundance, out);
} MDBG_API_CATCH(ctx)
