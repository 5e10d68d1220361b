// exchange_double.cpp -- TEST INFRASTRUCTURE: what csrc/multigpu.hip needs from the other translation units of the library, on a CPU.
// tests/test_exchange_on_cpu.py builds libmdbg_exchange_cpu.so from the SHIPPED context.hip + multigpu.hip (+ peerlink.hpp, common.hpp,
// objects.hpp) compiled by g++ against tests/host/hip_on_host (a stand-in for the HIP runtime over host memory), plus this file: the
// owner-side reduction of a sharded pass (mdbg_shard_reduce, csrc/kminmer.hip on the device) restated over host memory -- for every received
// row [lo, hi, count] the sum of the counts of its key, bit 63 on exactly one row per key -- and a way to make the shard handle it hangs on.
// Nothing of the product links this.
#include <map>
#include <utility>
#include <vector>

#include "../../metamdbg_amd/csrc/common.hpp"
#include "../../metamdbg_amd/csrc/objects.hpp"

struct mdbg_shard {
    mdbg_ctx *ctx = nullptr;
    mdbg::DevBuf<uint64_t> reply;
    uint64_t reduced_rows = 0;
    int fail_next_reduce = 0;
};

extern "C" {
uint32_t mdbg_row_words(uint32_t) { return 3; }

int mdbg_shard_for_test(mdbg_ctx *ctx, mdbg_shard **out) { *out = new mdbg_shard(); (*out)->ctx = ctx; return MDBG_OK; }
int mdbg_shard_test_fail_next_reduce(mdbg_shard *s) { s->fail_next_reduce = 1; return MDBG_OK; }

int mdbg_shard_reduce(mdbg_ctx *ctx, mdbg_shard *s, const uint64_t *d_rows, uint64_t n_recv, const uint64_t **d_reply) {
    if (s->fail_next_reduce) { s->fail_next_reduce = 0; return mdbg::set_error(ctx, MDBG_EHIP, "test: the owner's reduction failed"); }
    MDBG_TRY(s->reply.alloc(ctx, n_recv));
    std::map<std::pair<uint64_t, uint64_t>, std::pair<uint64_t, uint64_t>> keys;       // (hi, lo) -> (sum, first row)
    for (uint64_t i = 0; i < n_recv; i++) {
        auto it = keys.emplace(std::make_pair(d_rows[3 * i + 1], d_rows[3 * i]), std::make_pair(0ull, i)).first;
        it->second.first += d_rows[3 * i + 2];
    }
    for (uint64_t i = 0; i < n_recv; i++) {
        const auto &e = keys[std::make_pair(d_rows[3 * i + 1], d_rows[3 * i])];
        s->reply.p[i] = e.first | (e.second == i ? 1ull << 63 : 0ull);
    }
    s->reduced_rows = n_recv;
    *d_reply = s->reply.p;
    return MDBG_OK;
}
int mdbg_shard_begin(mdbg_ctx *ctx, const mdbg_minimizers *, uint32_t, uint32_t, mdbg_shard **, const uint64_t **, uint64_t *) { return mdbg::set_error(ctx, MDBG_ENODEV, "exchange double: no kernels here"); }
int mdbg_shard_finish(mdbg_ctx *ctx, mdbg_shard *, const uint64_t *, uint32_t, mdbg_table **) { return mdbg::set_error(ctx, MDBG_ENODEV, "exchange double: no kernels here"); }
void mdbg_shard_free(mdbg_shard *s) { delete s; }

// "device" memory of the stand-in is host memory: the test hands rows over and reads replies through these
int mdbg_test_device_alloc(mdbg_ctx *ctx, uint64_t bytes, void **out) { (void)ctx; return hipMalloc(out, bytes ? bytes : 8) == hipSuccess ? MDBG_OK : MDBG_ENOMEM; }
void mdbg_test_device_free(void *p) { (void)hipFree(p); }
}
