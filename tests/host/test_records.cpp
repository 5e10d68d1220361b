// test_records.cpp -- metamdbg_amd/host/records.hpp (the record files of `graph` indexed by several threads) against the serial walk:
// random files, files whose VALUES look like record headers (false candidates for the guessing threads), records longer than a chunk,
// empty records, a truncated file.  Test infrastructure; built and run by tests/test_hostfeed.py.
#include <cstdio>
#include <cstdlib>
#include <random>

#include "../../metamdbg_amd/host/records.hpp"

using namespace mdbgfeed;

static std::vector<uint64_t> serial(const std::vector<uint8_t> &f, bool *trunc) {
    std::vector<uint64_t> offs{0};
    size_t p = 0;
    *trunc = false;
    while (p < f.size()) {
        if (p + 5 > f.size()) { *trunc = true; break; }
        uint32_t n; memcpy(&n, f.data() + p, 4);
        if ((uint64_t)n > (f.size() - p - 5) / 4) { *trunc = true; break; }
        offs.push_back(offs.back() + n);
        p += 5 + (size_t)n * 4;
    }
    return offs;
}

static void put_record(std::vector<uint8_t> &f, const std::vector<uint32_t> &v, uint8_t circ) {
    const uint32_t n = (uint32_t)v.size();
    const size_t at = f.size();
    f.resize(at + 5 + (size_t)n * 4);
    memcpy(f.data() + at, &n, 4);
    f[at + 4] = circ;
    if (n) memcpy(f.data() + at + 5, v.data(), (size_t)n * 4);
}

int main() {
    std::mt19937_64 rng(20261001);
    int cases = 0, rewalks = 0;
    for (int mode = 0; mode < 6; mode++) {
        for (int rep = 0; rep < 3; rep++) {
            std::vector<uint8_t> f;
            const size_t target = ((size_t)9 << 20) + (rng() % ((size_t)4 << 20));
            while (f.size() < target) {
                size_t n = rng() % 80;
                if (mode == 1 && rng() % 50 == 0) n = 0;                                        // empty records
                if (mode == 2 && rng() % 4000 == 0) n = 300000 + rng() % 500000;                // records longer than a chunk of 16 threads
                std::vector<uint32_t> v(n);
                for (auto &x : v) {
                    x = (uint32_t)rng();
                    if (mode == 3) x = (uint32_t)(rng() % 40);                                  // values that read as plausible counts ...
                    if (mode == 4) x = (rng() & 1) ? (uint32_t)(rng() % 3) : (uint32_t)((rng() % 2) << 0);   // ... followed by bytes 0 / 1: whole fake chains
                    if (mode == 5) x = 0;                                                       // zeros everywhere: every position is a chain of empty records
                }
                put_record(f, v, (uint8_t)(rng() % 2));
            }
            for (int cut = 0; cut < 2; cut++) {
                if (cut) f.resize(f.size() - 3);                                                // truncated inside the last record
                bool trunc = false;
                const std::vector<uint64_t> want = serial(f, &trunc);
                for (int threads : {1, 2, 7, 16, 64}) {
                    const RecordIndex got = index_records(f.data(), f.size(), threads);
                    cases++;
                    rewalks += (int)got.rewalked;
                    if (got.truncated != trunc || (!trunc && got.offs != want)) {
                        fprintf(stderr, "mode %d rep %d cut %d threads %d: differs from the serial walk (%zu / %zu records, truncated %d / %d)\n", mode, rep, cut, threads,
                                got.n_records(), want.size() - 1, (int)got.truncated, (int)trunc);
                        return 1;
                    }
                }
            }
        }
    }
    // a small file takes one thread; an empty file is no records
    { std::vector<uint8_t> f; put_record(f, {1, 2, 3}, 0); const RecordIndex g = index_records(f.data(), f.size(), 16); if (g.chunks != 1 || g.offs != std::vector<uint64_t>{0, 3}) return 2; }
    { const RecordIndex g = index_records(nullptr, 0, 8); if (g.truncated || g.n_records() != 0) return 3; }
    printf("ok: %d cases, %d chunks walked again by the joining thread\n", cases, rewalks);
    return 0;
}
