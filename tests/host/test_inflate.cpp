// CPU check of metamdbg_amd/host/inflate.hpp against zlib: streams of every block type (stored, fixed, dynamic), level and
// strategy, texts from DNA-like to incompressible, decoded with the output room cut at random places (history kept in
// front of the room as the feed does), then damaged copies of the streams, which must end in CORRUPT or in a CRC /
// length mismatch -- never in a crash (the Python test builds this file with -fsanitize=address,undefined).
//   test_inflate <seed> <rounds>
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <string>
#include <vector>

#include "../../metamdbg_amd/host/crc32_fast.hpp"
#include "../../metamdbg_amd/host/inflate.hpp"

using mdbg_host::Inflater;

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t> &text, int level, int strategy, int memLevel, std::mt19937_64 &rng) {
    z_stream zs{};
    if (deflateInit2(&zs, level, Z_DEFLATED, -15, memLevel, strategy) != Z_OK) abort();
    std::vector<uint8_t> out(deflateBound(&zs, text.size()) + 1024 + text.size() / 100);
    zs.next_out = out.data(); zs.avail_out = (uInt)out.size();
    // feed in pieces with occasional flushes: block boundaries, empty stored blocks (Z_SYNC_FLUSH), dictionary resets (Z_FULL_FLUSH)
    size_t o = 0;
    while (o < text.size()) {
        size_t n = std::min<size_t>(text.size() - o, 1 + rng() % 200000);
        zs.next_in = const_cast<Bytef *>(text.data() + o); zs.avail_in = (uInt)n;
        const int fl = (rng() % 4 == 0) ? (rng() % 2 ? Z_SYNC_FLUSH : Z_FULL_FLUSH) : Z_NO_FLUSH;
        if (deflate(&zs, fl) == Z_STREAM_ERROR) abort();
        if (zs.avail_in) abort();
        o += n;
    }
    if (deflate(&zs, Z_FINISH) != Z_STREAM_END) abort();
    out.resize(zs.total_out);
    deflateEnd(&zs);
    return out;
}

static std::vector<uint8_t> make_text(std::mt19937_64 &rng, size_t n, int kind) {
    std::vector<uint8_t> t(n);
    static const char acgt[] = "ACGT";
    switch (kind) {
    case 0: for (auto &c : t) c = (uint8_t)acgt[rng() & 3]; break;                                   // DNA
    case 1: for (auto &c : t) c = (uint8_t)rng(); break;                                             // incompressible
    case 2: for (auto &c : t) c = (uint8_t)('a' + (rng() % 3 == 0)); break;                         // long runs / short distances
    case 3: {                                                                                      // FASTQ-like
        size_t o = 0;
        while (o < n) {
            std::string rec = "@read" + std::to_string(rng() % 100000) + "\n";
            const size_t L = 50 + rng() % 3000;
            for (size_t i = 0; i < L; i++) rec.push_back(acgt[rng() & 3]);
            rec += "\n+\n";
            for (size_t i = 0; i < L; i++) rec.push_back((char)(33 + rng() % 42));
            rec.push_back('\n');
            for (char c : rec) { if (o < n) t[o++] = (uint8_t)c; }
        }
        break;
    }
    default: {                                                                                     // repeats at all distances up to 32 KB
        size_t o = 0;
        while (o < n) {
            if (o > 0 && rng() % 3) {
                const size_t d = 1 + rng() % std::min<size_t>(o, 32768), L = 3 + rng() % 300;
                for (size_t i = 0; i < L && o < n; i++, o++) t[o] = t[o - d];
            } else t[o++] = (uint8_t)rng();
        }
    }
    }
    return t;
}

// decode with Inflater in random pieces of room, the way hostfeed.hpp does: history directly in front of the room
static int inflate_pieces(const std::vector<uint8_t> &comp, std::vector<uint8_t> &out, std::mt19937_64 &rng, size_t *in_used) {
    Inflater inf;
    inf.reset(comp.data(), comp.data() + comp.size());
    const size_t HIST = 32768;
    std::vector<uint8_t> buf;
    out.clear();
    size_t hist = 0;
    for (int guard = 0; guard < 1000000; guard++) {
        const size_t room = 300 + rng() % (rng() % 8 ? 5000 : 400000);
        buf.resize(HIST + room);
        // history: the last `hist` bytes of out, placed directly in front of the room
        hist = std::min(out.size(), HIST);
        if (hist) memcpy(buf.data() + HIST - hist, out.data() + out.size() - hist, hist);
        size_t produced = 0;
        const Inflater::Status st = inf.run(buf.data() + HIST, buf.data() + HIST + room, hist, &produced);
        out.insert(out.end(), buf.begin() + (long)HIST, buf.begin() + (long)(HIST + produced));
        if (st == Inflater::CORRUPT) return -1;
        if (st == Inflater::STREAM_END) { *in_used = (size_t)(inf.in_pos() - comp.data()); return 0; }
        if (out.size() > ((size_t)1 << 28)) return -2;        // a damaged stream may expand without end: give up
    }
    return -3;
}

int main(int argc, char **argv) {
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int rounds = argc > 2 ? atoi(argv[2]) : 40;
    std::mt19937_64 rng(seed);
    size_t streams = 0, damaged = 0, detected = 0;
    for (int r = 0; r < rounds; r++) {
        const int kind = r % 5;
        const size_t n = r % 7 == 0 ? rng() % 50 : 1000 + rng() % 1500000;     // some tiny and empty inputs
        const std::vector<uint8_t> text = make_text(rng, n, kind);
        static const int levels[] = {0, 1, 2, 4, 6, 9};
        static const int strategies[] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
        const int level = levels[rng() % 6], strategy = strategies[rng() % 5], memLevel = 1 + (int)(rng() % 9);
        std::vector<uint8_t> comp = deflate_raw(text, level, strategy, memLevel, rng);
        // trailing bytes behind the stream (a gzip trailer in real life): in_pos() must stop right behind the stream
        const size_t stream_len = comp.size();
        for (int i = 0; i < 8 + (int)(rng() % 40); i++) comp.push_back((uint8_t)rng());
        std::vector<uint8_t> out;
        size_t used = 0;
        const int rc = inflate_pieces(comp, out, rng, &used);
        if (rc != 0 || out != text || used != stream_len) {
            fprintf(stderr, "round %d (kind %d level %d strategy %d n %zu): rc %d, %zu bytes (want %zu), input used %zu (want %zu)\n", r, kind, level,
                    strategy, n, rc, out.size(), text.size(), used, stream_len);
            return 1;
        }
        streams++;
        // damage: flip bytes / truncate; the result must be CORRUPT or differ in CRC/length -- or, rarely, be the same text
        for (int d = 0; d < 6; d++) {
            std::vector<uint8_t> bad(comp.begin(), comp.begin() + (long)stream_len);
            if (bad.empty()) break;
            if (d % 3 == 2) bad.resize(rng() % bad.size());
            else for (int f = 0; f < 1 + (int)(rng() % 3); f++) bad[rng() % bad.size()] ^= (uint8_t)(1u << (rng() % 8));
            std::vector<uint8_t> o2;
            size_t u2 = 0;
            const int rc2 = inflate_pieces(bad, o2, rng, &u2);
            damaged++;
            if (rc2 != 0 || o2 != text) detected++;
        }
    }
    // arbitrary bytes as a deflate stream / gzip header: any verdict, no crash, no read outside the buffer (address sanitizer)
    {
        size_t verdicts[3] = {0, 0, 0};
        for (int t = 0; t < 4000; t++) {
            std::vector<uint8_t> junk(rng() % 600);
            for (auto &c : junk) c = (uint8_t)rng();
            if (t % 3 == 0 && junk.size() > 4) { junk[0] = (uint8_t)(4 | (rng() & 1)); }          // a dynamic-block header more often
            std::vector<uint8_t> o2;
            size_t u2 = 0;
            const int rc = inflate_pieces(junk, o2, rng, &u2);
            verdicts[rc == 0 ? 0 : rc == -1 ? 1 : 2]++;
            if (!junk.empty()) { junk[0] = 0x1f; if (junk.size() > 3) { junk[1] = 0x8b; junk[2] = 8; junk[3] &= 0x1f; } }
            const size_t h = mdbg_host::gzip_header_size(junk.data(), junk.size());
            if (h > junk.size()) { fprintf(stderr, "gzip_header_size beyond the buffer\n"); return 1; }
        }
        fprintf(stderr, "random bytes: %zu decoded, %zu rejected, %zu gave up\n", verdicts[0], verdicts[1], verdicts[2]);
    }
    // crc32_fast (carry-less multiplication) against zlib: lengths around the 64-byte folds, odd alignments, a running CRC
    {
        std::vector<uint8_t> buf(1 << 20);
        for (auto &c : buf) c = (uint8_t)rng();
        for (int t = 0; t < 3000; t++) {
            const size_t off = rng() % 64, len = t < 1200 ? (size_t)t : rng() % 200000;
            const uint32_t seed = t % 3 ? (uint32_t)rng() : 0;
            if ((uint32_t)crc32(seed, buf.data() + off, (uInt)len) != mdbg_host::crc32_fast(seed, buf.data() + off, len)) {
                fprintf(stderr, "crc32_fast differs from zlib: length %zu offset %zu\n", len, off);
                return 1;
            }
        }
        uint32_t a = 0, b = 0;
        for (size_t o = 0; o < buf.size();) {
            const size_t l = std::min<size_t>(buf.size() - o, rng() % 50000);
            a = (uint32_t)crc32(a, buf.data() + o, (uInt)l); b = mdbg_host::crc32_fast(b, buf.data() + o, l);
            o += l;
        }
        if (a != b) { fprintf(stderr, "running crc32_fast differs from zlib\n"); return 1; }
    }
    printf("ok %zu streams, %zu damaged copies (%zu changed the result or were rejected)\n", streams, damaged, detected);
    return 0;
}
