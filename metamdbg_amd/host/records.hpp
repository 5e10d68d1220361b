// records.hpp -- the record files `graph` reads at every k, indexed by several threads, EXACTLY.
//
// read_data_corrected.txt and unitig_data.txt are `u32 n; u8 circular; u32 m[n]` records back to back
// (readSelection/ReadSelection.hpp:1420-1426; the reference reads them one record at a time under one critical section,
// KminmerParserParallel, Commons.hpp:7394-7424).  A record can only be found by walking the counts from the start of the file: 10 M
// dependent steps over 1.55 GB for configs[2]'s read set, 0.2 s of a `graph` process that has 20 ms of kernels to run.  Here the file
// is cut into as many chunks as there are threads; every thread but the first GUESSES a record start in its chunk -- the first position
// from which a chain of 48 records looks sane (flag byte 0 or 1, count inside the file) -- and walks to the end of its chunk.  A guess
// is never trusted: the chunks are then joined in order, and a thread's walk is accepted only if it began exactly where the accepted
// walk before it ended (the first chunk begins at byte 0, so by induction every accepted start IS a record start).  A chunk whose guess
// does not meet its predecessor's end -- a false candidate in front of the true one, or no candidate at all (one record longer than a
// chunk) -- is walked again from the right place by the joining thread.  The result is the serial walk's, whatever the bytes.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

namespace mdbgfeed {

struct RecordIndex {
    std::vector<uint64_t> offs{0};     // minimizers before each record; n_records + 1 entries
    bool truncated = false;            // the file ends inside a record
    unsigned chunks = 1, rewalked = 0; // how it was found (tests)
    size_t n_records() const { return offs.size() - 1; }
};

namespace detail {
inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }

// a chain of `depth` records from p that looks sane (or reaches the end of the file exactly)
inline bool plausible_chain(const uint8_t *raw, size_t size, size_t p, int depth) {
    for (int i = 0; i < depth; i++) {
        if (p == size) return true;
        if (p + 5 > size) return false;
        const uint64_t n = rd32(raw + p);
        if (raw[p + 4] > 1) return false;
        if (n > (size - p - 5) / 4) return false;
        p += 5 + (size_t)n * 4;
    }
    return true;
}

// records from `from` while their start is below `until`; counts appended to `lens`; returns where the next record starts
// (*truncated: a record runs past the end of the file -- the walk stops in front of it)
inline size_t walk(const uint8_t *raw, size_t size, size_t from, size_t until, std::vector<uint32_t> &lens, bool *truncated) {
    size_t p = from;
    while (p < until && p < size) {
        if (p + 5 > size) { *truncated = true; break; }
        const uint64_t n = rd32(raw + p);
        if (n > (size - p - 5) / 4) { *truncated = true; break; }
        lens.push_back((uint32_t)n);
        p += 5 + (size_t)n * 4;
    }
    return p;
}
}  // namespace detail

inline RecordIndex index_records(const uint8_t *raw, size_t size, int threads) {
    using namespace detail;
    RecordIndex out;
    unsigned T = (unsigned)(threads < 1 ? 1 : threads > 64 ? 64 : threads);
    if (size < ((size_t)8 << 20)) T = 1;
    out.chunks = T;
    struct Part { size_t begin = 0, end = 0, start = (size_t)-1, stop = 0; std::vector<uint32_t> lens; bool truncated = false; };
    std::vector<Part> parts(T);
    for (unsigned t = 0; t < T; t++) { parts[t].begin = size / T * t; parts[t].end = t + 1 == T ? size : size / T * (t + 1); }
    auto work = [&](unsigned t) {
        Part &P = parts[t];
        P.lens.reserve((P.end - P.begin) / 96 + 16);
        size_t s = (size_t)-1;
        if (t == 0) s = 0;
        else for (size_t p = P.begin; p < P.end; p++) if (plausible_chain(raw, size, p, 48)) { s = p; break; }
        P.start = s;
        if (s != (size_t)-1) P.stop = walk(raw, size, s, P.end, P.lens, &P.truncated);
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < T; t++) pool.emplace_back(work, t);
        for (auto &th : pool) th.join();
    }
    // join: `pos` is where the accepted walks have got to
    size_t pos = 0, total = 0;
    for (unsigned t = 0; t < T; t++) total += parts[t].lens.size();
    out.offs.reserve(total + 1);
    uint64_t acc = 0;
    auto take = [&](const std::vector<uint32_t> &lens) { for (uint32_t n : lens) { acc += n; out.offs.push_back(acc); } };
    for (unsigned t = 0; t < T && !out.truncated; t++) {
        Part &P = parts[t];
        if (P.start == pos) { take(P.lens); pos = P.stop; out.truncated = P.truncated; continue; }
        if (pos >= P.end) continue;                                 // a record of an earlier chunk runs over this whole chunk
        std::vector<uint32_t> lens;                                 // the guess was not the record start the chain arrives at: walk again
        bool trunc = false;
        pos = walk(raw, size, pos, P.end, lens, &trunc);
        take(lens);
        out.truncated = trunc;
        out.rewalked++;
    }
    if (!out.truncated && pos != size) out.truncated = true;
    return out;
}

}  // namespace mdbgfeed
