// gzip_parallel.hpp -- an ordinary gzip file decoded by several threads.
//
// A deflate stream has no index and every block may refer to the 32 KB before it, so it is normally decoded by one
// thread (0.5 GB/s of text with inflate.hpp).  The two-pass scheme used here (known from pugz / rapidgzip) removes
// that dependency:
//   1. the compressed FILE is cut into chunks of equal size, whatever gzip members it is made of; for every chunk but the
//      first a thread looks for the first deflate block header behind the cut -- a position where a non-final dynamic
//      block parses with complete Huffman codes, decodes to plain text and is followed by more valid data;
//   2. a chunk is decoded from its header to the next chunk's header WITHOUT its window, into 16-bit symbols: bytes, or
//      "position p of the unknown window" (32768 + p), which matches copy around like bytes.  It must end exactly on the
//      header the next chunk found -- a wrongly guessed header makes the decoder run past it, which is an error.  Where a
//      member ends inside the chunk the decoder notes the place (its trailer: CRC-32 and length), reads the next member's
//      header and goes on: a member has no history, so nothing behind the border can refer to anything in front of it;
//   3. windows are resolved in chunk order (the last 32 KB of the member the chunk ends in) and every chunk's symbols are
//      translated to bytes by the pool, STRAIGHT INTO the caller's buffer (fill_begin / fill_wait: hostfeed.hpp cuts its slabs
//      from it; the length of every decoded chunk is known, so every piece has its place before it is translated); the
//      CRC-32s of the pieces are combined per member and checked against the member's trailer, like its length.
// Any inconsistency ends the run with an error naming the switch back to one decoding thread; nothing is guessed silently.
// Jobs never wait for each other: a chunk is decoded only once its stop position is known, translated only once its window is.
// (Until round 3 the chunks were cut per MEMBER and a member was started when the one before it had been read: a file of
// 8 MB members -- each smaller than the pool's look-ahead -- ran at a quarter of the rate of the same text in one member.)
#pragma once

#include <emmintrin.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "crc32_fast.hpp"
#include "inflate.hpp"

namespace mdbg_host {

class ParallelGzipReader {
public:
    static constexpr size_t WINDOW = 32768;
    static constexpr uint64_t NONE = ~0ull;

    ParallelGzipReader(const uint8_t *addr, size_t len, int threads, std::string path, size_t chunk_bytes)
        : addr_(addr), len_(len), path_(std::move(path)), chunk_(chunk_bytes < 65536 ? 65536 : chunk_bytes), nthreads_(threads < 1 ? 1 : threads) {
        pooled_ = !getenv("MDBG_HOST_GZIP_NO_POOL");                           // A/B switches for two round-3 changes
        table_translate_ = !getenv("MDBG_HOST_GZIP_BRANCHY");
        const size_t h = gzip_header_size(addr_, len_);
        if (!h) { fatal_ = "not a gzip file: " + path_; return; }
        chunks_.resize(len_ / chunk_ + 1);
        Chunk &c0 = chunks_[0];
        c0.start_bit = (uint64_t)h * 8;
        c0.find_taken = c0.start_known = c0.window_known = true;               // starts with the first member; nothing precedes it
        // A stream without dynamic-Huffman block headers to start from (stored blocks: gzip -0, incompressible data) cannot be
        // cut: ask before the first request and use one thread then.  Looked for in the second and third chunk.
        {
            InflaterT<uint16_t> probe;
            std::vector<uint16_t> scratch;
            for (size_t k = 1; k < std::min<size_t>(3, chunks_.size()) && !usable_; k++)
                usable_ = find_block(addr_, addr_ + len_, (uint64_t)k * chunk_ * 8, (uint64_t)(k + 1) * chunk_ * 8, probe, scratch) != NONE;
        }
        if (usable_) for (int i = 0; i < nthreads_; i++) pool_.emplace_back([this] { work(); });
    }
    bool usable() const { return usable_; }
    ~ParallelGzipReader() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : pool_) if (t.joinable()) t.join();
        if (getenv("MDBG_HOST_GZIP_TRACE"))
            fprintf(stderr, "[gzip on %d threads] thread-seconds: looking for headers %.3f, decoding %.3f, translating + CRC %.3f\n", nthreads_,
                    t_find_.load() * 1e-9, t_decode_.load() * 1e-9, t_translate_.load() * 1e-9);
    }

    bool direct() const { return true; }

    // Text straight to its place (what BgzfReader::fill_begin does for BGZF): the symbols of the next chunks -- as many as fit into
    // `room` bytes at `dst` -- are translated by the pool INTO `dst`; returns at once with the number of bytes that will be there
    // (it waits only for chunks that are not decoded yet), 0 at the end of the file.  fill_wait() returns when they are in place
    // and every member that ended among them has its CRC-32 and length checked.  One request at a time.
    size_t fill_begin(char *dst, size_t room) {
        constexpr size_t SUB = (size_t)1 << 20;                                // a job for one thread
        // the pool hears of the pieces however this call ends: one that throws at a damaged chunk has usually handed out the pieces
        // of the chunks in front of it, and fill_settle() waits for exactly those
        struct Wake { std::condition_variable &cv; ~Wake() { cv.notify_all(); } } wake{cv_};
        std::unique_lock<std::mutex> g(mu_);
        pieces_.clear();
        piecesNext_ = piecesDone_ = 0;
        pending_ = false;
        size_t placed = 0;
        while (placed < room && !ended_) {
            if (!fatal_.empty()) throw std::runtime_error(fatal_);
            if (consumed_ >= chunks_.size()) throw std::runtime_error("gzip stream without an end in " + path_ + hint());
            Chunk &c = chunks_[consumed_];
            if (!((c.decoded && c.window_known) || !c.error.empty())) {
                // the pool may be asleep: the chunks it was allowed to decode ahead were all done, and only this call moved the
                // window on (and handed it pieces to translate)
                cv_.notify_all();
                cv_.wait(g, [&] { return (c.decoded && c.window_known) || !c.error.empty() || !fatal_.empty(); });
            }
            if (!fatal_.empty()) throw std::runtime_error(fatal_);
            if (!c.error.empty()) throw std::runtime_error(c.error + hint());
            // up to the next member border inside the chunk (a piece never straddles one: the CRC-32s are combined per member),
            // the end of the chunk or the end of the room
            while (placed < room || (rborder_ < c.borders.size() && c.borders[rborder_].first == rpos_)) {
                if (rborder_ < c.borders.size() && c.borders[rborder_].first == rpos_) {
                    Piece pc;                                                  // a member ends here: its trailer is checked behind the pieces so far
                    pc.chunk = consumed_; pc.from = rpos_; pc.n = 0; pc.dst = dst + placed; pc.member_end = c.borders[rborder_].second;
                    pieces_.push_back(pc);
                    pending_ = true;
                    c.pieces_open++;
                    rborder_++;
                    continue;
                }
                const size_t upto = rborder_ < c.borders.size() ? c.borders[rborder_].first : c.nsym;
                if (rpos_ == upto) break;                                      // the chunk is used up
                const size_t n = std::min({upto - rpos_, room - placed, SUB});
                Piece pc;
                pc.chunk = consumed_; pc.from = rpos_; pc.n = n; pc.dst = dst + placed;
                pieces_.push_back(pc);
                pending_ = true;                                               // (from here on fill_wait / fill_settle have something to wait for,
                c.pieces_open++;                                               //  also when this call ends in an exception further down)
                rpos_ += n; placed += n;
            }
            if (rpos_ < c.nsym || rborder_ < c.borders.size()) break;          // the room is used up
            c.handed = true;
            if (c.pieces_open == 0) give_sym(c.sym, c.sym_cap);
            if (c.file_end) ended_ = true;                                     // what follows the last member is ignored, as gzread does
            consumed_++;
            rpos_ = 0; rborder_ = 0;
        }
        return placed;
    }
    void fill_wait() {
        if (!pending_) return;
        std::vector<Piece> done;
        {
            std::unique_lock<std::mutex> g(mu_);
            // (after a failure: not before every thread has let go of the caller's buffer)
            cvDone_.wait(g, [&] { return piecesDone_ == pieces_.size() || (!pieceError_.empty() && piecesBusy_ == 0); });
            pending_ = false;
            piecesNext_ = pieces_.size();
            if (!pieceError_.empty()) throw std::runtime_error(pieceError_ + hint());
            done.swap(pieces_);
            piecesNext_ = piecesDone_ = 0;
        }
        for (const Piece &pc : done) {
            crc_ = (uint32_t)crc32_combine(crc_, pc.crc, (z_off_t)pc.n);
            total_ += pc.n;
            if (pc.member_end == NONE) continue;
            const uint8_t *t = addr_ + pc.member_end;                          // trailer: CRC-32, ISIZE
            const uint32_t want_crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            const uint32_t want_len = t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
            if (want_crc != crc_ || want_len != (uint32_t)total_) throw std::runtime_error("gzip CRC / length mismatch in " + path_ + hint());
            total_ = 0;
            crc_ = (uint32_t)crc32(0L, Z_NULL, 0);
        }
    }
    void fill_settle() noexcept { try { fill_wait(); } catch (...) {} }
    bool at_end() {                                                            // nothing is left to ask for
        std::lock_guard<std::mutex> g(mu_);
        return ended_ && fatal_.empty();
    }

private:
    struct Chunk {
        uint64_t start_bit = NONE;          // first block header behind the cut, in bits from the start of the FILE; NONE: none here
        bool find_taken = false, start_known = false, decode_taken = false, decoded = false, window_known = false;
        std::unique_ptr<uint16_t[]> sym;    // WINDOW place-holders, then the symbols
        size_t nsym = 0, sym_cap = 0;
        std::unique_ptr<uint8_t[]> window;  // the WINDOW bytes in front of this chunk, right-aligned when fewer exist
        size_t window_valid = 0;
        // members that end inside the chunk: (symbols in front of the border, file offset of the member's trailer), ascending
        std::vector<std::pair<size_t, uint64_t>> borders;
        bool file_end = false;              // the last member ended in this chunk: later ones are never delivered
        std::string error;
        uint64_t uid = 0;                   // set when decoded: names the chunk's window in the translating threads' tables
        size_t pieces_open = 0;             // pieces of this chunk handed out and not yet translated
        bool handed = false;                // every symbol has been handed out
    };
    // the request that is out (fill_begin .. fill_wait); mu_ guards all of it
    struct Piece {
        size_t chunk = 0, from = 0, n = 0;  // symbols [from, from + n) of the chunk ...
        char *dst = nullptr;                // ... to this place
        uint32_t crc = 0;
        uint64_t member_end = NONE;         // a member's trailer lies behind this (empty) piece
    };

    std::string hint() const { return " (MDBG_HOST_GZIP_THREADS=1 decodes the stream on one thread)"; }

    // ---- finding a block header ------------------------------------------------------------------------------------
    static bool plausible_text(const uint16_t *s, size_t n) {
        for (size_t i = 0; i < n; i++) {
            const uint16_t v = s[i];
            // read files are text: no control characters (bytes >= 128 may be UTF-8 in a header line).  Bytes decoded from a
            // wrong position are uniformly random: a few hundred of them already contain one.
            if (v < 32 ? !(v == '\n' || v == '\r' || v == '\t') : v == 127) return false;
        }
        return true;
    }

    // first bit in [from, to) of `data` where a non-final dynamic block starts that decodes to text and is followed by
    // more valid data; NONE if there is none
    static uint64_t find_block(const uint8_t *data, const uint8_t *end, uint64_t from, uint64_t to, InflaterT<uint16_t> &probe, std::vector<uint16_t> &scratch) {
        if (scratch.size() < WINDOW + PROBE_ROOM) {
            scratch.resize(WINDOW + PROBE_ROOM);
            for (size_t i = 0; i < WINDOW; i++) scratch[i] = (uint16_t)(0x8000u + i);
        }
        const uint64_t last = (uint64_t)(end - data) * 8;
        for (uint64_t bit = from; bit < to && bit + 128 < last; bit++) {
            uint64_t w;
            memcpy(&w, data + (bit >> 3), 8);
            w >>= (bit & 7);
            if ((w & 7) != 4) continue;                                        // BFINAL = 0, BTYPE = 2 (dynamic Huffman)
            if (((w >> 3) & 31) > 29 || ((w >> 8) & 31) > 29) continue;        // HLIT, HDIST
            // the code-length code must be complete: sum over its codes of 2^(7 - length) == 2^7
            const unsigned hclen = (unsigned)((w >> 13) & 15) + 4;
            uint64_t cl;
            memcpy(&cl, data + ((bit + 17) >> 3), 8);
            cl >>= ((bit + 17) & 7);                                           // 57 bits left: 19 lengths of 3 bits
            unsigned kraft = 0;
            for (unsigned i = 0; i < hclen; i++) {
                const unsigned l = (unsigned)((cl >> (3 * i)) & 7);
                if (l) kraft += 128u >> l;
            }
            if (kraft != 128) continue;
            probe.reset_at_bit(data, bit, end);
            size_t produced = 0;
            const auto st = probe.run(scratch.data() + WINDOW, scratch.data() + scratch.size(), WINDOW, &produced);
            if (st == InflaterT<uint16_t>::CORRUPT) continue;                  // NEED_ROOM: 256 K symbols decoded cleanly
            if (st == InflaterT<uint16_t>::STREAM_END && probe.blocks_done() < 2) continue;
            if (!plausible_text(scratch.data() + WINDOW, produced)) continue;
            return bit;
        }
        return NONE;
    }

    // ---- the pool ------------------------------------------------------------------------------------------------------
    enum Job { NOTHING, FIND, DECODE, PIECE };

    // stop position of chunk k: the header found by the next chunk that has one.  false = not decided yet
    bool stop_of(size_t k, uint64_t *stop) const {
        size_t j = k + 1;
        while (j < chunks_.size() && chunks_[j].start_known && chunks_[j].start_bit == NONE) j++;
        if (j == chunks_.size()) { *stop = NONE; return true; }                // runs to the end of the file
        if (!chunks_[j].start_known) return false;
        *stop = chunks_[j].start_bit;
        return true;
    }

    Job next_job(size_t *k, uint64_t *stop) {                                  // mu_ held
        const size_t ahead = (size_t)nthreads_ + 2;
        const size_t hi = std::min(chunks_.size(), consumed_ + ahead);
        for (size_t i = consumed_; i < hi; i++) {
            Chunk &c = chunks_[i];
            if (i > end_chunk_) break;
            if (c.start_known && c.start_bit != NONE && !c.decode_taken && stop_of(i, stop)) { c.decode_taken = true; *k = i; return DECODE; }
        }
        // headers are looked for one chunk further than chunks are decoded -- and as far beyond that as it takes to find the
        // stop position of the last chunk in the window (a deflate block may be longer than a chunk)
        for (size_t i = consumed_; i < chunks_.size(); i++) {
            Chunk &c = chunks_[i];
            if (i > end_chunk_) break;
            if (!c.find_taken) { c.find_taken = true; *k = i; return FIND; }
            if (i >= hi && (!c.start_known || c.start_bit != NONE)) break;      // being looked for, or found: nothing further is needed yet
        }
        return NOTHING;
    }

    void work() {
        InflaterT<uint16_t> inf, probe;
        std::vector<uint16_t> scratch;
        std::unique_ptr<uint8_t[]> lut(new uint8_t[65536]());                  // symbol -> byte (translate_symbols); entries 256 .. 0x7fff are never produced
        for (unsigned v = 0; v < 256; v++) lut[v] = (uint8_t)v;
        uint64_t lut_of = 0;                                                   // the chunk (uid) whose window the table's upper half holds
        for (;;) {
            size_t k = 0, piece_index = 0;
            uint64_t stop = NONE;
            Job job = NOTHING;
            Piece piece;
            {
                std::unique_lock<std::mutex> g(mu_);
                for (;;) {
                    if (stop_) return;
                    if (piecesNext_ < pieces_.size() && pieceError_.empty()) {      // the reader's request goes first: it frees the symbols
                        piece = pieces_[piecesNext_];
                        piece_index = piecesNext_++;
                        piecesBusy_++;
                        k = piece.chunk;
                        job = PIECE;
                        break;
                    }
                    if (fatal_.empty()) job = next_job(&k, &stop);
                    if (job != NOTHING) break;
                    cv_.wait(g);
                }
            }
            Chunk &c = chunks_[k];
            const auto t0 = std::chrono::steady_clock::now();
            if (job == PIECE) {
                std::string err;
                uint32_t crc = 0;
                try {
                    if (piece.n) {
                        translate_symbols(c.sym.get() + WINDOW + piece.from, piece.n, c.window.get(), WINDOW - c.window_valid, (uint8_t *)piece.dst, lut.get(),
                                          lut_of != c.uid);
                        lut_of = c.uid;
                        crc = crc32_fast(0, (const uint8_t *)piece.dst, piece.n);
                    }
                } catch (const std::exception &e) { err = e.what(); }
                bool last;
                {
                    std::lock_guard<std::mutex> g(mu_);
                    piecesBusy_--;
                    if (err.empty()) { pieces_[piece_index].crc = crc; piecesDone_++; }
                    else if (pieceError_.empty()) pieceError_ = err;
                    if (--c.pieces_open == 0 && c.handed) give_sym(c.sym, c.sym_cap);
                    last = pieceError_.empty() ? piecesDone_ == pieces_.size() : piecesBusy_ == 0;
                }
                t_translate_ += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                if (last) cvDone_.notify_all();
                continue;
            }
            try {
                if (job == FIND) {
                    const uint64_t b = find_block(addr_, addr_ + len_, (uint64_t)k * chunk_ * 8, (uint64_t)(k + 1) * chunk_ * 8, probe, scratch);
                    std::lock_guard<std::mutex> g(mu_);
                    c.start_bit = b;
                    c.start_known = true;
                    if (b == NONE) { c.decode_taken = true; finish_decode(k, 0); }   // nothing starts here: the previous chunk runs through
                } else {
                    decode_chunk(k, stop, inf);
                }
            } catch (const std::exception &e) {
                // the reader meets the error when it gets to this chunk (chunks behind the end of the file never are)
                std::lock_guard<std::mutex> g(mu_);
                c.error = e.what();
                c.decoded = true;
            }
            const long long ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            (job == FIND ? t_find_ : t_decode_) += ns;
            cv_.notify_all();
        }
    }

    // From the chunk's header to `stop` (the next chunk's), across the borders of the members that end on the way.
    void decode_chunk(size_t k, uint64_t stop, InflaterT<uint16_t> &inf) {
        const uint8_t *end = addr_ + len_;
        Chunk &c = chunks_[k];
        inf.reset_at_bit(addr_, c.start_bit, end);
        if (stop != NONE) inf.set_stop_bit(stop);
        size_t cap = WINDOW + chunk_ * 5 + 65536;                              // symbols: [WINDOW place-holders][output]; grows as needed
        std::unique_ptr<uint16_t[]> sym = take_sym(&cap);
        for (size_t i = 0; i < WINDOW; i++) sym[i] = (uint16_t)(0x8000u + i);
        size_t n = 0;
        size_t member_from = 0;             // symbols in front of the member being decoded (0: it began before this chunk)
        bool own_history = k == 0;          // the member being decoded began inside this chunk (or with the file): nothing in front of it exists
        std::vector<std::pair<size_t, uint64_t>> borders;
        bool file_end = false;
        for (;;) {
            if (cap - (WINDOW + n) < 4096) {
                // a chunk that never reaches the next header (a long stretch of stored or fixed-code blocks) would grow without
                // bound: 64 chunks' worth of text is where this gives up
                if (n > chunk_ * 64 * 4) throw std::runtime_error("no block header to cut at for " + std::to_string(n) + " bytes of text in " + path_);
                size_t ncap = cap + cap / 2;
                std::unique_ptr<uint16_t[]> bigger(new uint16_t[ncap]);
                memcpy(bigger.get(), sym.get(), (WINDOW + n) * sizeof(uint16_t));
                sym = std::move(bigger);
                cap = ncap;
            }
            size_t produced = 0;
            // what lies in front of the output and may be referred to: the member's own text in this chunk, and -- for a member
            // that began before the chunk -- the place-holders of the unknown window
            const size_t hist = own_history ? n - member_from : WINDOW + n;
            const auto st = inf.run(sym.get() + WINDOW + n, sym.get() + cap, hist, &produced);
            n += produced;
            if (st == InflaterT<uint16_t>::CORRUPT) throw std::runtime_error("corrupt gzip data or a misjudged block boundary in " + path_);
            if (st == InflaterT<uint16_t>::AT_STOP) break;
            if (st == InflaterT<uint16_t>::STREAM_END) {
                const uint8_t *t = inf.in_pos();
                if ((size_t)(end - t) < 8) throw std::runtime_error("truncated gzip file: " + path_);
                borders.emplace_back(n, (uint64_t)(t - addr_));
                const size_t next = (size_t)(t - addr_) + 8;
                const size_t h = next < len_ ? gzip_header_size(addr_ + next, len_ - next) : 0;
                if (!h) { file_end = true; break; }                            // the end of the file (trailing bytes are ignored, as gzread does)
                if (stop != NONE && (uint64_t)(next + h) * 8 > stop) throw std::runtime_error("a misjudged block boundary in " + path_);
                inf.reset_at_bit(addr_, (uint64_t)(next + h) * 8, end);
                if (stop != NONE) inf.set_stop_bit(stop);
                member_from = n;
                own_history = true;
            }
        }
        std::lock_guard<std::mutex> g(mu_);
        c.sym = std::move(sym);
        c.sym_cap = cap;
        c.borders = std::move(borders);
        if (file_end) { c.file_end = true; if (k < end_chunk_) end_chunk_ = k; }
        finish_decode(k, n);
    }

    // mu_ held.  Marks chunk k decoded and resolves windows in chunk order: the window of chunk i + 1 is the last WINDOW bytes
    // of the member that chunk i ends in -- of its text in chunk i and, when the member began before chunk i, of chunk i's own
    // window.  (Under the same lock as `decoded`, so that no translation frees symbols the chain still needs.)
    void finish_decode(size_t k, size_t nsym) {
        chunks_[k].uid = ++uid_counter_;
        chunks_[k].nsym = nsym;
        chunks_[k].decoded = true;
        while (chain_next_ < chunks_.size()) {
            Chunk &c = chunks_[chain_next_];
            if (!c.decoded || !c.window_known) break;
            const size_t i = chain_next_++;
            if (!c.error.empty() || i + 1 >= chunks_.size() || i >= end_chunk_) continue;
            Chunk &nx = chunks_[i + 1];
            nx.window.reset(new uint8_t[WINDOW]);
            const size_t member_from = c.borders.empty() ? 0 : c.borders.back().first;       // the member chunk i ends in began here (0: before the chunk)
            const bool carries = c.borders.empty();                                           // ... and so chunk i's own window is part of its history
            const size_t from_text = std::min(c.nsym - member_from, WINDOW), from_win = WINDOW - from_text;
            if (from_win) {
                if (carries && c.window) memcpy(nx.window.get(), c.window.get() + from_text, from_win);
                else memset(nx.window.get(), 0, from_win);
            }
            nx.window_valid = std::min(WINDOW, from_text + (carries ? std::min(c.window_valid, from_win) : 0));
            for (size_t j = 0; j < from_text; j++) {
                const uint16_t v = c.sym[WINDOW + c.nsym - from_text + j];
                if (v < 256) { nx.window[from_win + j] = (uint8_t)v; continue; }
                const size_t p = v - 0x8000u;
                if (!carries || !c.window || p < WINDOW - c.window_valid) { nx.error = "gzip data refers to text before the start of the stream in " + path_; break; }
                nx.window[from_win + j] = c.window[p];
            }
            nx.window_known = true;
        }
    }

    // n symbols -> bytes at t.  win: the WINDOW bytes in front of the chunk (may be null), valid from min_p on.  lut: the thread's
    // symbol -> byte table; load_window: its upper half does not hold this chunk's window yet.
    void translate_symbols(const uint16_t *s, size_t n, const uint8_t *win, size_t min_p, uint8_t *t, uint8_t *lut, bool load_window) const {
        size_t i = 0;
        if (win && min_p == 0 && table_translate_) {
            // The usual case, a whole window in front of the chunk: every symbol the decoder can have produced is valid, and the
            // translation is one table look-up per symbol -- bytes map to themselves, 0x8000 + p to the window's byte p.  (In reads
            // the references do not die out behind the first 32 KB: a match copies them along with the bytes, and zlib finds a
            // match every few bases of a read, so "is it a byte?" is a coin flip the branch predictor loses: 4 ns per symbol.)
            if (load_window) memcpy(lut + 0x8000, win, WINDOW);
            for (; i + 16 <= n; i += 16) {
                const __m128i a = _mm_loadu_si128((const __m128i *)(s + i)), b = _mm_loadu_si128((const __m128i *)(s + i + 8));
                const __m128i high = _mm_and_si128(_mm_or_si128(a, b), _mm_set1_epi16((short)0xFF00));
                if (_mm_movemask_epi8(_mm_cmpeq_epi16(high, _mm_setzero_si128())) == 0xFFFF) {                 // sixteen bytes
                    _mm_storeu_si128((__m128i *)(t + i), _mm_packus_epi16(a, b));
                    continue;
                }
                for (unsigned j = 0; j < 16; j++) t[i + j] = lut[s[i + j]];
            }
            for (; i < n; i++) t[i] = lut[s[i]];
        }
        for (; i < n; i++) {
            const uint16_t v = s[i];
            if (v < 256) { t[i] = (uint8_t)v; continue; }
            const size_t p = v - 0x8000u;
            if (!win || p < min_p) throw std::runtime_error("gzip data refers to text before the start of the stream in " + path_);
            t[i] = win[p];
        }
    }

    // ---- buffers go round ------------------------------------------------------------------------------------------------
    // A chunk's symbols are 40 MB of address space: fresh from the allocator every time, they are mapped, faulted in page by page
    // and unmapped again by two dozen threads of one process at once -- which the kernel serialises (the parallel decoder was
    // no faster than one thread for it).  mu_ held in give_sym; pool_mu_ guards the list.
    std::unique_ptr<uint16_t[]> take_sym(size_t *cap) {
        {
            std::lock_guard<std::mutex> g(pool_mu_);
            for (size_t i = 0; i < idle_sym_.size(); i++)
                if (idle_sym_[i].second >= *cap) {
                    auto p = std::move(idle_sym_[i].first);
                    *cap = idle_sym_[i].second;
                    idle_sym_.erase(idle_sym_.begin() + (long)i);
                    return p;
                }
        }
        return std::unique_ptr<uint16_t[]>(new uint16_t[*cap]);
    }
    void give_sym(std::unique_ptr<uint16_t[]> &p, size_t cap) {
        if (!p) return;
        std::lock_guard<std::mutex> g(pool_mu_);
        if (pooled_ && idle_sym_.size() < (size_t)nthreads_ + 4) idle_sym_.emplace_back(std::move(p), cap);
        p.reset();
    }

    static constexpr size_t PROBE_ROOM = 1u << 18;

    const uint8_t *addr_;
    size_t len_;
    std::string path_;
    size_t chunk_;
    int nthreads_;
    std::vector<Chunk> chunks_;             // the file, cut every chunk_ bytes
    size_t consumed_ = 0;                   // first chunk not handed out entirely
    size_t chain_next_ = 0;                 // first chunk whose successor's window is not resolved yet
    size_t end_chunk_ = (size_t)-1;         // chunk in which the last member ended
    bool ended_ = false;                    // ... and it has been handed out
    size_t rpos_ = 0, rborder_ = 0;         // the reader inside chunk consumed_: symbols and borders handed out
    uint32_t crc_ = 0;                      // of the member being read, so far (fill_wait; the caller's thread only)
    uint64_t total_ = 0;
    std::mutex mu_;
    std::condition_variable cv_;
    std::mutex pool_mu_;
    std::vector<std::pair<std::unique_ptr<uint16_t[]>, size_t>> idle_sym_;
    std::vector<std::thread> pool_;
    std::atomic<long long> t_find_{0}, t_decode_{0}, t_translate_{0};
    bool stop_ = false, usable_ = false, pooled_ = true, table_translate_ = true;
    std::vector<Piece> pieces_;
    size_t piecesNext_ = 0, piecesDone_ = 0, piecesBusy_ = 0;
    bool pending_ = false;                  // (the caller's thread only)
    std::string pieceError_;
    std::condition_variable cvDone_;
    uint64_t uid_counter_ = 0;
    std::string fatal_;
};

}  // namespace mdbg_host
