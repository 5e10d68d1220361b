// One thread of metamdbg_amd/host/inflate.hpp over a gzip file (one member): MB/s of text, for bytes (the BGZF / one-thread path) and
// for 16-bit symbols (the several-threads gzip path, gzip_parallel.hpp).  The numbers of DESIGN.md section 6 ("the decoder's loop").
//   g++ -O2 -std=c++17 tools/host/inflate_bench.cpp -o /tmp/inflate_bench -lz
//   /tmp/inflate_bench reads.fasta.gz [room_bytes]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../metamdbg_amd/host/inflate.hpp"

using namespace mdbg_host;

template <typename T>
static void run(const std::vector<uint8_t> &raw, size_t n, size_t h, size_t room, const char *what) {
    std::vector<T> out(room);
    for (int rep = 0; rep < 3; rep++) {
        const auto t0 = std::chrono::steady_clock::now();
        InflaterT<T> inf;
        inf.reset(raw.data() + h, raw.data() + n);
        size_t produced = 0;
        const auto st = inf.run(out.data(), out.data() + out.size(), 0, &produced);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%s: %.1f MB in %.3f s = %.0f MB/s%s\n", what, produced / 1e6, s, produced / 1e6 / s,
               st == InflaterT<T>::STREAM_END ? "" : " (the room was used up before the end of the stream)");
    }
}

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: inflate_bench file.gz [room_bytes]\n"); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END);
    const size_t n = (size_t)ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> raw(n + 64);
    if (fread(raw.data(), 1, n, f) != n) return 1;
    fclose(f);
    const size_t h = gzip_header_size(raw.data(), n);
    if (!h) { fprintf(stderr, "not a gzip file\n"); return 1; }
    const size_t room = argc > 2 ? (size_t)atoll(argv[2]) : (size_t)1 << 30;
    run<uint8_t>(raw, n, h, room, "bytes          ");
    run<uint16_t>(raw, n, h, room, "16-bit symbols ");
    return 0;
}
