// Drains metamdbg_amd/host/hostfeed.hpp over the given files and prints the delivery rate: how fast the host side can
// hand batches to the GPU (no GPU involved; buffers are plain malloc).
//   g++ -O2 -std=c++17 tools/host/hostfeed_bench.cpp -o /tmp/hostfeed_bench -lz -lpthread
//   /tmp/hostfeed_bench <chunkBytes> <threads> file...
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../metamdbg_amd/host/hostfeed.hpp"

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: hostfeed_bench <chunkBytes> <threads> file...\n"); return 2; }
    const size_t chunk = (size_t)atoll(argv[1]);
    const int threads = atoi(argv[2]);
    std::vector<std::string> files(argv + 3, argv + argc);
    const auto t0 = std::chrono::steady_clock::now();
    uint64_t reads = 0, bases = 0, batches = 0, packed = 0;
    try {
        mdbg_host::ReadFeeder feeder(files, chunk, threads, 0, [](size_t n) { return malloc(n); }, [](void *p) { free(p); });
        while (mdbg_host::ReadBatch *b = feeder.next()) {
            reads += b->n(); bases += b->nbases; batches++; packed += b->packed;
            feeder.recycle(b);
        }
    } catch (const std::exception &e) { fprintf(stderr, "exception: %s\n", e.what()); return 3; }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("{\"reads\": %llu, \"gbp\": %.4f, \"batches\": %llu, \"packed\": %llu, \"seconds\": %.3f, \"gbp_per_s\": %.3f, \"threads\": %d}\n",
           (unsigned long long)reads, bases / 1e9, (unsigned long long)batches, (unsigned long long)packed, s, bases / 1e9 / s, threads);
    return 0;
}
