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

// ---- hand-built streams --------------------------------------------------------------------------------------------
struct BitWriter {
    std::vector<uint8_t> bytes;
    uint64_t acc = 0;
    int n = 0;
    void bits(uint32_t v, int count) {                        // LSB first (header fields, extra bits)
        acc |= (uint64_t)v << n; n += count;
        while (n >= 8) { bytes.push_back((uint8_t)acc); acc >>= 8; n -= 8; }
    }
    void code(uint32_t c, int len) {                          // Huffman codes travel most significant bit first
        for (int i = len - 1; i >= 0; i--) bits((c >> i) & 1u, 1);
    }
    void finish() { if (n) bits(0, 8 - n); }
};

// One dynamic block: literals A, C, G with 2-bit codes (00, 01, 10), end-of-block 110, length-258 code 111; a single
// 1-bit distance code (symbol 5: distances 7-8, one extra bit).  Four such literals fit one 11-bit look-up four times
// over, so one trip of the fast loop emits 16 literals AND a 258-byte match copied in 8-byte steps: the most a trip
// can write.  An earlier room test (258 + 16) was 6 bytes short of that.
static std::vector<uint8_t> literal_run_match_stream(std::vector<uint8_t> &text, int groups) {
    BitWriter w;
    w.bits(1, 1); w.bits(2, 2);                               // final block, dynamic
    w.bits(286 - 257, 5); w.bits(30 - 1, 5); w.bits(18 - 4, 4);
    // code-length alphabet: symbols 0..3 get 2-bit codes (00 01 10 11), sent in the order 16,17,18,0,8,7,9,6,10,5,11,4,12,3,13,2,14,1
    static const int order[18] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1};
    for (int s : order) w.bits(s <= 3 ? 2 : 0, 3);
    uint8_t lens[286 + 30] = {0};
    lens['A'] = 2; lens['C'] = 2; lens['G'] = 2; lens[256] = 3; lens[285] = 3;
    lens[286 + 5] = 1;
    for (uint8_t l : lens) w.code(l, 2);
    text.clear();
    auto lit = [&](char c) { w.code(c == 'A' ? 0 : c == 'C' ? 1 : 2, 2); text.push_back((uint8_t)c); };
    static const char acg[] = "ACG";
    unsigned x = 7;
    for (int i = 0; i < 320; i++) { x = x * 1103515245u + 12345u; lit(acg[(x >> 16) % 3]); }
    for (int g = 0; g < groups; g++) {
        for (int i = 0; i < 16; i++) { x = x * 1103515245u + 12345u; lit(acg[(x >> 16) % 3]); }
        w.code(7, 3);                                         // length 258 (symbol 285, no extra bits)
        w.code(0, 1); w.bits(1, 1);                           // distance symbol 5, extra bit 1: distance 8
        for (int i = 0; i < 258; i++) text.push_back(text[text.size() - 8]);
    }
    w.code(6, 3);                                             // end of block
    w.finish();
    for (int i = 0; i < 32; i++) w.bytes.push_back(0);        // readable input behind the stream (a gzip trailer in real life)
    return w.bytes;
}

// every room size against a buffer whose end is guarded: the decoder must stay inside [out, out_end)
static bool test_room_is_respected() {
    std::vector<uint8_t> text;
    const std::vector<uint8_t> comp = literal_run_match_stream(text, 12);
    {   // the stream is valid deflate: zlib agrees
        std::vector<uint8_t> ref(text.size() + 16);
        z_stream zs{};
        if (inflateInit2(&zs, -15) != Z_OK) return false;
        zs.next_in = const_cast<Bytef *>(comp.data()); zs.avail_in = (uInt)comp.size();
        zs.next_out = ref.data(); zs.avail_out = (uInt)ref.size();
        const int rc = inflate(&zs, Z_FINISH);
        const bool ok = rc == Z_STREAM_END && zs.total_out == text.size() && memcmp(ref.data(), text.data(), text.size()) == 0;
        inflateEnd(&zs);
        if (!ok) { fprintf(stderr, "hand-built stream: zlib disagrees (rc %d)\n", rc); return false; }
    }
    const size_t GUARD = 64;
    for (size_t room = (size_t)Inflater::MIN_ROOM; room < 1400; room++) {
        Inflater inf;
        inf.reset(comp.data(), comp.data() + comp.size());
        std::vector<uint8_t> out;
        for (int piece = 0; piece < 100000; piece++) {
            const size_t hist = std::min<size_t>(out.size(), 32768);
            // exact-size allocation + guard bytes behind out_end
            uint8_t *buf = (uint8_t *)malloc(hist + room + GUARD);
            if (hist) memcpy(buf, out.data() + out.size() - hist, hist);
            memset(buf + hist + room, 0xA5, GUARD);
            size_t produced = 0;
            const Inflater::Status st = inf.run(buf + hist, buf + hist + room, hist, &produced);
            bool guard_ok = true;
            for (size_t i = 0; i < GUARD; i++) guard_ok &= buf[hist + room + i] == 0xA5;
            if (!guard_ok || produced > room) {
                fprintf(stderr, "room %zu: the decoder wrote behind out_end (produced %zu)\n", room, produced);
                free(buf);
                return false;
            }
            out.insert(out.end(), buf + hist, buf + hist + produced);
            free(buf);
            if (st == Inflater::CORRUPT) { fprintf(stderr, "room %zu: hand-built stream rejected\n", room); return false; }
            if (st == Inflater::STREAM_END) break;
        }
        if (out != text) { fprintf(stderr, "room %zu: wrong text (%zu bytes, want %zu)\n", room, out.size(), text.size()); return false; }
    }
    // the same trip in the 16-bit symbol decoder (gzip_parallel.hpp): elements, not bytes
    for (size_t room = (size_t)mdbg_host::InflaterT<uint16_t>::MIN_ROOM; room < 700; room++) {
        mdbg_host::InflaterT<uint16_t> inf;
        inf.reset(comp.data(), comp.data() + comp.size());
        std::vector<uint16_t> out;
        for (int piece = 0; piece < 100000; piece++) {
            const size_t hist = std::min<size_t>(out.size(), 32768);
            uint16_t *buf = (uint16_t *)malloc((hist + room + GUARD) * 2);
            if (hist) memcpy(buf, out.data() + out.size() - hist, hist * 2);
            for (size_t i = 0; i < GUARD; i++) buf[hist + room + i] = 0xA5A5;
            size_t produced = 0;
            const auto st = inf.run(buf + hist, buf + hist + room, hist, &produced);
            bool guard_ok = true;
            for (size_t i = 0; i < GUARD; i++) guard_ok &= buf[hist + room + i] == 0xA5A5;
            if (!guard_ok || produced > room) { fprintf(stderr, "room %zu (u16): the decoder wrote behind out_end\n", room); free(buf); return false; }
            out.insert(out.end(), buf + hist, buf + hist + produced);
            free(buf);
            if (st == mdbg_host::InflaterT<uint16_t>::CORRUPT) { fprintf(stderr, "room %zu (u16): rejected\n", room); return false; }
            if (st == mdbg_host::InflaterT<uint16_t>::STREAM_END) break;
        }
        if (out.size() != text.size()) { fprintf(stderr, "room %zu (u16): wrong length\n", room); return false; }
        for (size_t i = 0; i < out.size(); i++) if (out[i] != text[i]) { fprintf(stderr, "room %zu (u16): wrong symbol\n", room); return false; }
    }
    // a dynamic block whose literal/length set is the single 1-bit end-of-block code (zlib accepts it): empty output
    {
        BitWriter w;
        w.bits(1, 1); w.bits(2, 2);
        w.bits(0, 5); w.bits(0, 5); w.bits(18 - 4, 4);         // 257 literal/length codes, 1 distance code
        static const int order[18] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1};
        for (int s : order) w.bits((s == 0 || s == 1 || s == 18) ? (s == 18 ? 1 : 2) : 0, 3);   // 18: code 0, 0: code 10, 1: code 11
        w.code(0, 1); w.bits(127, 7);                          // 138 zeros
        w.code(0, 1); w.bits(118 - 11, 7);                     // 118 zeros: symbols 0..255
        w.code(3, 2);                                          // symbol 256: length 1
        w.code(2, 2);                                          // the distance code: unused
        w.code(0, 1);                                          // end of block
        w.finish();
        for (int i = 0; i < 32; i++) w.bytes.push_back(0);
        std::vector<uint8_t> ref(16);
        z_stream zs{};
        inflateInit2(&zs, -15);
        zs.next_in = w.bytes.data(); zs.avail_in = (uInt)w.bytes.size();
        zs.next_out = ref.data(); zs.avail_out = (uInt)ref.size();
        const int rc = inflate(&zs, Z_FINISH);
        const size_t zout = zs.total_out;
        inflateEnd(&zs);
        if (rc != Z_STREAM_END || zout != 0) { fprintf(stderr, "end-of-block-only stream: zlib says %d\n", rc); return false; }
        Inflater inf;
        inf.reset(w.bytes.data(), w.bytes.data() + w.bytes.size());
        std::vector<uint8_t> buf(4096);
        size_t produced = 1;
        const Inflater::Status st = inf.run(buf.data(), buf.data() + buf.size(), 0, &produced);
        if (st != Inflater::STREAM_END || produced != 0) { fprintf(stderr, "end-of-block-only block: status %d\n", (int)st); return false; }
    }
    return true;
}

int main(int argc, char **argv) {
    if (!test_room_is_respected()) return 1;
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
