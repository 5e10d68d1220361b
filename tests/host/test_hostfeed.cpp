// CPU check of metamdbg_amd/host/hostfeed.hpp: the parallel chunked reader must deliver exactly the
// reads (bases, qualities, order) the sequential kseq-style reader (fastx.hpp) delivers.  Batches the workers packed
// to 2 bits are compared through the base codes ((c >> 1) & 3), must follow the device layout (reads on even words,
// zero padding) and must never contain a character with bit 3 set -- those chunks have to arrive as ASCII.
//   test_hostfeed <chunkBytes> <threads> <maxReadsPerFile> file...
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../metamdbg_amd/host/hostfeed.hpp"

int main(int argc, char **argv) {
    if (argc < 5) return 2;
    size_t chunk = (size_t)atoll(argv[1]);
    int threads = atoi(argv[2]);
    uint64_t maxReads = (uint64_t)atoll(argv[3]);
    std::vector<std::string> files(argv + 4, argv + argc);
    if (getenv("MDBG_TEST_DRAIN_ONLY")) {
        // damaged input: what matters is that the feeder ends with an error (exit code 3) instead of completing (0); what it
        // delivers before it notices differs legitimately from what zlib delivers before it notices
        try {
            mdbg_host::ReadFeeder feeder(files, chunk, threads, maxReads, [](size_t n) { return malloc(n); }, [](void *p) { free(p); });
            size_t n = 0;
            while (mdbg_host::ReadBatch *b = feeder.next()) { n += b->n(); feeder.recycle(b); }
            printf("completed %zu reads\n", n);
            return 0;
        } catch (const std::exception &e) { fprintf(stderr, "exception: %s\n", e.what()); return 3; }
    }
    // expected: sequential reader
    std::vector<std::string> expSeq, expQual;
    for (auto &f : files) {
        mdbg_host::FastxReader rd(f);
        uint64_t perFile = 0;
        for (;;) {
            if (maxReads > 0 && perFile > maxReads) break;
            std::string s, q;
            bool hq = false;
            if (!rd.next(s, q, hq)) break;
            perFile++;
            expSeq.push_back(s);
            expQual.push_back(hq ? q : std::string());
        }
    }
    size_t i = 0, nbatches = 0, npacked = 0, nodd = 0;
    try {
        mdbg_host::ReadFeeder feeder(files, chunk, threads, maxReads, [](size_t n) { return malloc(n); }, [](void *p) { free(p); });
        while (mdbg_host::ReadBatch *b = feeder.next()) {
            nbatches++;
            if (b->packed) {
                npacked++;
                if (b->wordOff.size() != (size_t)b->n() + 1 || b->lens.size() != b->n()) { fprintf(stderr, "packed batch: bad index sizes\n"); return 1; }
                if (b->oddOff.size() != b->odd.size() + 1 || b->oddOff.back() != b->oddBases.size()) { fprintf(stderr, "packed batch: bad odd-read index\n"); return 1; }
                for (size_t o = 1; o < b->odd.size(); o++) if (b->odd[o] <= b->odd[o - 1]) { fprintf(stderr, "odd reads not ascending\n"); return 1; }
            } else if (!b->odd.empty()) { fprintf(stderr, "ASCII batch with odd reads\n"); return 1; }
            size_t oddAt = 0;
            for (uint32_t r = 0; r < b->n(); r++, i++) {
                if (i >= expSeq.size()) { fprintf(stderr, "too many reads\n"); return 1; }
                if (b->packed) {
                    const std::string &e = expSeq[i];
                    const uint64_t w0 = b->wordOff[r], w1 = b->wordOff[r + 1];
                    if (b->lens[r] != e.size() || (w0 & 1) || w1 - w0 != ((e.size() + 63) / 64) * 2 || b->offsets[r + 1] - b->offsets[r] != e.size()) {
                        fprintf(stderr, "packed read %zu: bad layout\n", i); return 1;
                    }
                    // a read with anything but upper-case ACGT must be listed and carried again as characters; no other read may be
                    bool plain = true;
                    for (char c : e) plain = plain && (c == 'A' || c == 'C' || c == 'G' || c == 'T');
                    const bool listed = oddAt < b->odd.size() && b->odd[oddAt] == r;
                    if (listed == plain) { fprintf(stderr, "packed read %zu: %s\n", i, plain ? "listed as odd but plain" : "holds other characters and is not listed"); return 1; }
                    if (listed) {
                        if (b->oddBases.substr(b->oddOff[oddAt], b->oddOff[oddAt + 1] - b->oddOff[oddAt]) != e) { fprintf(stderr, "odd read %zu: characters differ\n", i); return 1; }
                        oddAt++; nodd++;
                    }
                    for (size_t k = 0; k < (w1 - w0) * 32; k++) {
                        const unsigned got = (unsigned)((b->words()[w0 + k / 32] >> (2 * (k % 32))) & 3u);
                        const unsigned want = k < e.size() ? (((unsigned char)e[k] >> 1) & 3u) : 0u;
                        if (got != want) { fprintf(stderr, "packed read %zu differs at base %zu\n", i, k); return 1; }
                    }
                } else {
                    std::string s(b->bases + b->offsets[r], b->bases + b->offsets[r + 1]);
                    if (s != expSeq[i]) { fprintf(stderr, "read %zu differs (len %zu vs %zu)\n", i, s.size(), expSeq[i].size()); return 1; }
                }
                if (b->hasQual) {
                    std::string q(b->quals + b->offsets[r], b->quals + b->offsets[r + 1]);
                    if (q != expQual[i]) { fprintf(stderr, "qual %zu differs\n", i); return 1; }
                } else if (!expQual[i].empty()) { fprintf(stderr, "missing qualities at %zu\n", i); return 1; }
            }
            if (b->packed && oddAt != b->odd.size()) { fprintf(stderr, "odd reads left over\n"); return 1; }
            feeder.recycle(b);
        }
    } catch (const std::exception &e) { fprintf(stderr, "exception: %s\n", e.what()); return 3; }
    if (i != expSeq.size()) { fprintf(stderr, "got %zu reads, expected %zu\n", i, expSeq.size()); return 1; }
    printf("ok %zu reads %zu batches %zu packed %zu odd\n", i, nbatches, npacked, nodd);
    return 0;
}
