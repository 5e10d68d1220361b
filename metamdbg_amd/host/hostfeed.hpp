// hostfeed.hpp -- parallel FASTA / FASTQ ingestion for the host tool.
//
// The reference parses with one kseq reader inside an `omp critical` (Commons.hpp:5868-5905), which
// caps it near 0.5 Gbp/s whatever the thread count.  Here a plain (uncompressed) file is mmap'ed and
// cut at record boundaries into chunks that worker threads parse independently into page-locked
// batches (read order preserved by sequence numbers).  A gzip file is one deflate stream and cannot be
// split: one thread inflates it into slabs cut at record starts and the same workers parse and pack the
// slabs (multi-line FASTQ inside gzip falls back to a sequential kseq-style reader).  Batches come out in
// file order, ready for mdbg_reads_from_packed / mdbg_reads_from_ascii.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "crc32_fast.hpp"
#include "fastx.hpp"
#include "gzip_parallel.hpp"
#include "inflate.hpp"

namespace mdbg_host {

struct ReadBatch {
    char *bases = nullptr;      // page-locked, `cap` bytes: ASCII bases, or (packed) the 2-bit words of the device layout
    char *quals = nullptr;      // page-locked, `cap` bytes
    size_t cap = 0;
    size_t nbases = 0;
    std::vector<uint64_t> offsets{0};     // base offsets of the reads (also the offsets of the qualities)
    bool hasQual = false;
    // packed == true: `bases` holds u64 words, 32 bases each (base i at bits [2i, 2i+2), code (c >> 1) & 3), read r in
    // words [wordOff[r], wordOff[r+1]) starting on an even word -- what mdbg_reads_from_packed takes.  Workers pack
    // while they parse (a quarter of the PCIe bytes); a chunk holding any character other than A, C, G, T (N, lower case,
    // IUPAC letters: the device packer keeps their invalid bits and character changes) or too many tiny reads for the
    // buffer is delivered as ASCII instead.
    bool packed = false;
    std::vector<uint64_t> wordOff{0};
    std::vector<uint32_t> lens;
    // packed batches: the few reads that hold something else than upper-case ACGT travel a second time as characters
    // (mdbg_reads_mark_ascii derives their side masks on the device); a chunk in which they are many is delivered as ASCII
    std::vector<uint32_t> odd;            // ascending read indices
    std::string oddBases;
    std::vector<uint64_t> oddOff{0};
    int file = 0;
    uint32_t n() const { return (uint32_t)(offsets.size() - 1); }
    const uint64_t *words() const { return reinterpret_cast<const uint64_t *>(bases); }
    void clear() {
        nbases = 0; offsets.assign(1, 0); hasQual = false; packed = false; wordOff.assign(1, 0); lens.clear();
        odd.clear(); oddBases.clear(); oddOff.assign(1, 0);
    }
};

// ---- 2-bit packing of ASCII bases on the host -------------------------------------------------------------------
// 8 characters -> 16 bits.  code = (c >> 1) & 3 (utils/kmer/Kmer.hpp:462); bit 3 of a character marks it invalid.
inline uint64_t pack8_swar(uint64_t x) {
    uint64_t y = (x >> 1) & 0x0303030303030303ull;
    y = (y | (y >> 6)) & 0x000F000F000F000Full;
    y = (y | (y >> 12)) & 0x000000FF000000FFull;
    return (y | (y >> 24)) & 0xFFFFull;
}
#if defined(__x86_64__)
__attribute__((target("bmi2"))) inline uint64_t pack8_pext(uint64_t x) { return __builtin_ia32_pext_di(x >> 1, 0x0303030303030303ull); }
inline bool have_bmi2() { static const bool v = __builtin_cpu_supports("bmi2") && !getenv("MDBG_HOST_NO_BMI2"); return v; }
#else
inline uint64_t pack8_pext(uint64_t x) { return pack8_swar(x); }
inline bool have_bmi2() { return false; }
#endif

// Bytes of x that are not one of 'A' (0x41), 'C' (0x43), 'G' (0x47), 'T' (0x54) leave a non-zero byte: with b1, b2 the
// code bits of a byte, the only valid spelling of that code is 0x40 | code << 1 | (T ? 0x10 : 0x01), T <=> b2 & ~b1.
inline uint64_t not_acgt8(uint64_t x) {
    const uint64_t t = (x >> 2) & ~(x >> 1) & 0x0101010101010101ull;
    const uint64_t expect = 0x4040404040404040ull | (x & 0x0606060606060606ull) | (t << 4) | (t ^ 0x0101010101010101ull);
    return x ^ expect;
}

// Appends bases to a stream of u64 words.  `invalid` becomes non-zero when a character other than A, C, G, T was seen:
// the packed form alone would lose what the reference still sees (bit 3 = invalid k-mers, utils/kmer/Kmer.hpp:462; any
// change of character = a new homopolymer run, Commons.hpp:4177-4178).
struct PackCursor {
    uint64_t *words;      // destination
    size_t capWords;
    size_t w = 0;         // current word
    unsigned fill = 0;    // bases already in words[w] (0..31)
    uint64_t cur = 0;
    uint64_t invalid = 0;
    bool overflow = false;

    template <bool PEXT>
    void append_impl(const char *p, size_t n) {
        // head: byte-wise until the word boundary
        while (n && fill) { push1((unsigned char)*p++); n--; }
        if (overflow) return;
        // body: 32 characters -> one word
        while (n >= 32) {
            if (w >= capWords) { overflow = true; return; }
            uint64_t x0, x1, x2, x3;
            memcpy(&x0, p, 8); memcpy(&x1, p + 8, 8); memcpy(&x2, p + 16, 8); memcpy(&x3, p + 24, 8);
            invalid |= not_acgt8(x0) | not_acgt8(x1) | not_acgt8(x2) | not_acgt8(x3);
            const uint64_t a = PEXT ? pack8_pext(x0) : pack8_swar(x0), b = PEXT ? pack8_pext(x1) : pack8_swar(x1);
            const uint64_t c = PEXT ? pack8_pext(x2) : pack8_swar(x2), d = PEXT ? pack8_pext(x3) : pack8_swar(x3);
            words[w++] = a | (b << 16) | (c << 32) | (d << 48);
            p += 32; n -= 32;
        }
        while (n) { push1((unsigned char)*p++); n--; }
    }
    void append(const char *p, size_t n) { if (have_bmi2()) append_impl<true>(p, n); else append_impl<false>(p, n); }
    void push1(unsigned char c) {
        invalid |= (uint64_t)(c != 'A' && c != 'C' && c != 'G' && c != 'T');
        cur |= (uint64_t)((c >> 1) & 3u) << (2 * fill);
        if (++fill == 32) {
            if (w >= capWords) { overflow = true; fill = 0; cur = 0; return; }
            words[w++] = cur; cur = 0; fill = 0;
        }
    }
    // end of a read: flush the partial word and pad with zero words to a 64-base unit (two words)
    void end_read() {
        if (fill) { if (w >= capWords) { overflow = true; } else words[w++] = cur; cur = 0; fill = 0; }
        if (w & 1) { if (w >= capWords) overflow = true; else words[w++] = 0; }
    }
    bool bad() const { return overflow || invalid != 0; }
};

inline bool zlib_inflate_requested() { static const bool v = getenv("MDBG_HOST_ZLIB_INFLATE") != nullptr; return v; }

// An ordinary gzip file (one or more members), memory-mapped and decoded by inflate.hpp in pieces of 4 MB on a thread of
// its own; a second thread checks every member's CRC-32 and length behind the decoder.  read() hands the text out in
// order and throws on damaged data.  Trailing bytes that are not a gzip member are ignored, as gzread does.
class GzipMemReader {
public:
    GzipMemReader(const uint8_t *addr, size_t len, std::string path) : addr_(addr), len_(len), path_(std::move(path)) {
        decoder_ = std::thread([this] { decode_loop(); });
        checker_ = std::thread([this] { check_loop(); });
    }
    ~GzipMemReader() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        if (decoder_.joinable()) decoder_.join();
        if (checker_.joinable()) checker_.join();
    }

    size_t read(char *dst, size_t want) {
        size_t got = 0;
        while (got < want) {
            if (!cur_ || cpos_ == cur_->len) {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return stop_ || !ready_.empty() || decoded_all_; });
                if (ready_.empty()) {
                    // everything decoded and handed out: the verdict of the checker is part of the end of the file
                    cv_.wait(g, [&] { return stop_ || checked_all_ || !error_.empty(); });
                    if (!error_.empty()) throw std::runtime_error(error_);
                    break;
                }
                cur_ = ready_.front();
                ready_.pop_front();
                cpos_ = 0;
                g.unlock();
                cv_.notify_all();
                continue;
            }
            const size_t n = std::min(want - got, cur_->len - cpos_);
            memcpy(dst + got, cur_->text() + cpos_, n);
            got += n; cpos_ += n;
        }
        return got;
    }

private:
    static constexpr size_t HIST = 32768, ROOM = (size_t)4 << 20;
    struct Piece {
        GzipMemReader *owner;
        std::unique_ptr<uint8_t[]> buf;    // HIST bytes of earlier text, then the text of this piece; recycled through the owner
        explicit Piece(GzipMemReader *o) : owner(o), buf(o->take_buffer()) {}
        ~Piece() { owner->give_buffer(std::move(buf)); }
        size_t len = 0;
        bool member_end = false;
        uint32_t crc = 0, isize = 0;       // the member's trailer (with member_end)
        const uint8_t *text() const { return buf.get() + HIST; }
    };

    // touched pages are worth keeping: a fresh 4 MB buffer per piece costs a thousand page faults
    std::unique_ptr<uint8_t[]> take_buffer() {
        {
            std::lock_guard<std::mutex> g(pool_mu_);
            if (!pool_.empty()) { std::unique_ptr<uint8_t[]> b = std::move(pool_.back()); pool_.pop_back(); return b; }
        }
        return std::unique_ptr<uint8_t[]>(new uint8_t[HIST + ROOM]);
    }
    void give_buffer(std::unique_ptr<uint8_t[]> b) {
        std::lock_guard<std::mutex> g(pool_mu_);
        pool_.push_back(std::move(b));
    }

    void fail(const std::string &msg) {
        {
            std::lock_guard<std::mutex> g(mu_);
            if (error_.empty()) error_ = msg;
            decoded_all_ = true;
        }
        cv_.notify_all();
    }

    void decode_loop() {
        Inflater inf;
        size_t pos = 0, hist = 0, members = 0;
        bool in_member = false;
        std::shared_ptr<Piece> prev;
        for (;;) {
            if (!in_member) {
                const size_t h = pos < len_ ? gzip_header_size(addr_ + pos, len_ - pos) : 0;
                if (!h) {
                    if (members == 0) { fail("not a gzip file: " + path_); return; }
                    break;                                  // end of file, or trailing bytes that are no gzip member
                }
                inf.reset(addr_ + pos + h, addr_ + len_);
                in_member = true;
                members++;
                hist = 0;
            }
            std::shared_ptr<Piece> p = std::make_shared<Piece>(this);
            if (hist) memcpy(p->buf.get() + HIST - hist, prev->text() + prev->len - hist, hist);
            size_t produced = 0;
            const Inflater::Status st = inf.run(p->buf.get() + HIST, p->buf.get() + HIST + ROOM, hist, &produced);
            if (st == Inflater::CORRUPT) { fail("corrupt gzip data in " + path_ + " (MDBG_HOST_ZLIB_INFLATE=1 decodes with zlib)"); return; }
            p->len = produced;
            if (st == Inflater::STREAM_END) {
                const uint8_t *t = inf.in_pos();
                if ((size_t)(addr_ + len_ - t) < 8) { fail("truncated gzip file: " + path_); return; }
                p->member_end = true;
                p->crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
                p->isize = t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
                pos = (size_t)(t - addr_) + 8;
                in_member = false;
            }
            hist = std::min(HIST, hist + produced);         // contiguous in p->buf: its own history, then its text
            prev = p;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return stop_ || (ready_.size() < 6 && check_.size() < 6); });
                if (stop_) return;
                ready_.push_back(p);
                check_.push_back(p);
            }
            cv_.notify_all();
        }
        {
            std::lock_guard<std::mutex> g(mu_);
            decoded_all_ = true;
        }
        cv_.notify_all();
    }

    void check_loop() {
        uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
        uint64_t n = 0;
        for (;;) {
            std::shared_ptr<Piece> p;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return stop_ || !check_.empty() || decoded_all_; });
                if (stop_) return;
                if (check_.empty()) { checked_all_ = true; g.unlock(); cv_.notify_all(); return; }
                p = check_.front();
                check_.pop_front();
            }
            cv_.notify_all();
            crc = crc32_fast(crc, p->text(), p->len);
            n += p->len;
            if (p->member_end) {
                if (crc != p->crc || (uint32_t)n != p->isize)
                    fail("gzip CRC / length mismatch in " + path_ + " (MDBG_HOST_ZLIB_INFLATE=1 decodes with zlib)");
                crc = (uint32_t)crc32(0L, Z_NULL, 0);
                n = 0;
            }
        }
    }

    const uint8_t *addr_;
    size_t len_;
    std::string path_;
    std::mutex pool_mu_;                                   // the buffer pool outlives every piece (declared before them)
    std::vector<std::unique_ptr<uint8_t[]>> pool_;
    size_t cpos_ = 0;
    std::shared_ptr<Piece> cur_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::shared_ptr<Piece>> ready_, check_;
    std::thread decoder_, checker_;
    bool stop_ = false, decoded_all_ = false, checked_all_ = false;
    std::string error_;
};

// BGZF (the blocked gzip that htslib writes: samtools fastq, bam2fastq, bgzip): every block of at most 64 KB is its
// own gzip member whose header carries the compressed block size, so the blocks of a memory-mapped file can be inflated
// by several threads at once.  read() hands the text out in file order.
class BgzfReader {
public:
    struct Block { size_t payload; uint32_t csize; uint32_t isize; uint32_t crc; };

    // true iff [addr, addr+len) is a sequence of BGZF blocks and nothing else
    static bool index(const unsigned char *addr, size_t len, std::vector<Block> &out) {
        size_t o = 0;
        out.clear();
        while (o < len) {
            if (len - o < 18 + 8) return false;
            const unsigned char *h = addr + o;
            if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
            const size_t xlen = h[10] | ((size_t)h[11] << 8);
            if (len - o < 12 + xlen + 8) return false;
            size_t bsize = 0;
            for (size_t x = 0; x + 4 <= xlen;) {               // extra subfields: SI1 SI2 SLEN(2) data
                const unsigned char *sf = h + 12 + x;
                const size_t slen = sf[2] | ((size_t)sf[3] << 8);
                if (sf[0] == 'B' && sf[1] == 'C' && slen == 2 && x + 6 <= xlen) bsize = (sf[4] | ((size_t)sf[5] << 8)) + 1;
                x += 4 + slen;
            }
            if ((h[3] & ~4u) || bsize < 12 + xlen + 8 || bsize > len - o) return false;   // other header fields are never set by BGZF writers
            const unsigned char *t = addr + o + bsize - 8;
            Block b;
            b.payload = o + 12 + xlen;
            b.csize = (uint32_t)(bsize - 12 - xlen - 8);
            b.crc = t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24);
            b.isize = t[4] | ((uint32_t)t[5] << 8) | ((uint32_t)t[6] << 16) | ((uint32_t)t[7] << 24);
            if (b.isize > (1u << 16)) return false;
            out.push_back(b);
            o += bsize;
        }
        return !out.empty();
    }

    BgzfReader(const unsigned char *addr, std::vector<Block> blocks, int threads, std::string path)
        : addr_(addr), blocks_(std::move(blocks)), path_(std::move(path)) {
        // groups of consecutive blocks holding about 4 MB of text
        size_t text = 0;
        groups_.push_back(0);
        for (size_t i = 0; i < blocks_.size(); i++) {
            if (text && text + blocks_[i].isize > ((size_t)4 << 20)) { groups_.push_back(i); text = 0; }
            text += blocks_[i].isize;
        }
        groups_.push_back(blocks_.size());
        window_ = (size_t)threads * 2 + 2;
        cum_.assign(blocks_.size() + 1, 0);
        for (size_t i = 0; i < blocks_.size(); i++) cum_[i + 1] = cum_[i] + blocks_[i].isize;
        direct_ = !getenv("MDBG_HOST_BGZF_COPY");       // A/B: the round-2 path (groups inflated into buffers of their own, read() copies)
        for (int i = 0; i < threads; i++) pool_.emplace_back([this] { if (direct_) fill_loop(); else inflate_loop(); });
    }
    ~BgzfReader() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &t : pool_) if (t.joinable()) t.join();
    }

    bool direct() const { return direct_; }

    // Text straight to its place.  A BGZF block says how much text it holds (ISIZE in its trailer), so the place of every block in
    // the caller's buffer is known before any is inflated: fill_begin() hands the next whole blocks that fit into `room` bytes at
    // `dst` to the pool and returns at once with the number of bytes they will put there (0: the next block does not fit, or the
    // file is at its end -- text_left() tells); fill_wait() returns when they are all in place.  One request at a time.
    // (Round 2 inflated groups of blocks into buffers of their own and one thread copied the text from there into the slabs it
    // cut: 3.9 GB/s of text on 24 inflating threads, that copy and the page faults of a fresh slab per 32 MB being the limit.)
    size_t fill_begin(char *dst, size_t room) {
        const size_t b0 = fillNext_;
        size_t b1 = b0;
        while (b1 < blocks_.size() && cum_[b1 + 1] - cum_[b0] <= room) b1++;
        fillNext_ = b1;
        if (cum_[b1] == cum_[b0]) return 0;               // (empty blocks -- the end marker is one -- hold nothing)
        {
            std::lock_guard<std::mutex> g(mu_);
            jobDst_ = dst; jobB0_ = b0; jobB1_ = b1; jobNext_ = b0; jobDone_ = 0;
            pending_ = true;
        }
        cv_.notify_all();
        return cum_[b1] - cum_[b0];
    }
    void fill_wait() {
        if (!pending_) return;
        std::unique_lock<std::mutex> g(mu_);
        // (after a failure: not before every thread has let go of the caller's buffer)
        cvDone_.wait(g, [&] { return (!error_.empty() && jobBusy_ == 0) || jobDone_ == jobB1_ - jobB0_; });
        pending_ = false;
        jobNext_ = jobB1_;                                // nothing of a failed request is taken up any more
        if (!error_.empty()) throw std::runtime_error(error_);
    }
    // the same without the exception (a caller unwinding: the buffer must not go before the pool has let go of it)
    void fill_settle() noexcept { try { fill_wait(); } catch (...) {} }
    size_t text_left() const { return cum_.back() - cum_[fillNext_]; }
    bool at_end() const { return text_left() == 0; }          // nothing is left to ask for

    // up to `want` bytes of text; 0 at the end of the file
    size_t read(char *dst, size_t want) {
        size_t got = 0;
        while (got < want) {
            std::unique_lock<std::mutex> g(mu_);
            cv_.wait(g, [&] { return !error_.empty() || cur_ + 1 >= groups_.size() || ready_.count(cur_); });
            if (!error_.empty()) throw std::runtime_error(error_);
            if (cur_ + 1 >= groups_.size()) break;
            Text &t = ready_[cur_];
            const size_t n = std::min(want - got, t.len - pos_);
            g.unlock();                                   // the entry of the current group is only touched by this thread
            memcpy(dst + got, t.buf.data() + pos_, n);
            got += n; pos_ += n;
            if (pos_ == t.len) {
                g.lock();
                ready_.erase(cur_);
                cur_++; pos_ = 0;
                g.unlock();
                cv_.notify_all();
            }
        }
        return got;
    }

private:
    struct Text { std::vector<char> buf; size_t len = 0; };

    void inflate_loop() {
        const bool use_zlib = zlib_inflate_requested();
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (use_zlib && inflateInit2(&zs, -15) != Z_OK) { fail("zlib initialisation failed"); return; }
        std::unique_ptr<Inflater> inf(use_zlib ? nullptr : new Inflater());
        for (;;) {
            size_t gi;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return stop_ || !error_.empty() || next_ + 1 >= groups_.size() || next_ < cur_ + window_; });
                if (stop_ || !error_.empty() || next_ + 1 >= groups_.size()) break;
                gi = next_++;
            }
            size_t total = 0;
            for (size_t b = groups_[gi]; b < groups_[gi + 1]; b++) total += blocks_[b].isize;
            Text text;
            text.len = total;
            text.buf.resize(total + 512);                  // the decoder wants Inflater::MIN_ROOM bytes of room in front of every symbol
            size_t o = 0;
            bool ok = true;
            for (size_t b = groups_[gi]; b < groups_[gi + 1] && ok; b++) {
                const Block &blk = blocks_[b];
                if (use_zlib) {
                    inflateReset(&zs);
                    zs.next_in = const_cast<Bytef *>(addr_ + blk.payload);
                    zs.avail_in = blk.csize;
                    zs.next_out = (Bytef *)text.buf.data() + o;
                    zs.avail_out = blk.isize;
                    ok = inflate(&zs, Z_FINISH) == Z_STREAM_END && zs.avail_out == 0;
                } else {
                    // blocks are independent deflate streams: no history in front of them
                    inf->reset(addr_ + blk.payload, addr_ + blk.payload + blk.csize);
                    size_t produced = 0;
                    uint8_t *out = (uint8_t *)text.buf.data() + o;
                    ok = inf->run(out, (uint8_t *)text.buf.data() + text.buf.size(), 0, &produced) == Inflater::STREAM_END && produced == blk.isize;
                }
                ok = ok && crc32_fast(0, (const unsigned char *)text.buf.data() + o, blk.isize) == blk.crc;
                o += blk.isize;
            }
            if (!ok) { fail("corrupt BGZF block in " + path_); break; }
            {
                std::lock_guard<std::mutex> g(mu_);
                ready_[gi] = std::move(text);
            }
            cv_.notify_all();
        }
        if (use_zlib) inflateEnd(&zs);
    }
    // a pool thread of the direct mode: takes runs of blocks of the current job; every block is inflated into the thread's own 64 KB
    // (the decoder may write a few bytes past the end of what it produces, and the neighbouring block's place belongs to another
    // thread) and copied to its place from there, cache-hot
    void fill_loop() {
        const bool use_zlib = zlib_inflate_requested();
        z_stream zs;
        memset(&zs, 0, sizeof(zs));
        if (use_zlib && inflateInit2(&zs, -15) != Z_OK) { fail("zlib initialisation failed"); return; }
        std::unique_ptr<Inflater> inf(use_zlib ? nullptr : new Inflater());
        std::vector<uint8_t> scratch(((size_t)1 << 16) + 1024);
        constexpr size_t RUN = 8;
        for (;;) {
            size_t a, b;
            char *dst;
            size_t base;
            {
                std::unique_lock<std::mutex> g(mu_);
                cv_.wait(g, [&] { return stop_ || !error_.empty() || jobNext_ < jobB1_; });
                if (stop_ || !error_.empty()) break;
                a = jobNext_;
                b = std::min(jobB1_, a + RUN);
                jobNext_ = b;
                jobBusy_++;
                dst = jobDst_;
                base = cum_[jobB0_];
            }
            bool ok = true;
            for (size_t i = a; i < b && ok; i++) {
                const Block &blk = blocks_[i];
                if (use_zlib) {
                    inflateReset(&zs);
                    zs.next_in = const_cast<Bytef *>(addr_ + blk.payload);
                    zs.avail_in = blk.csize;
                    zs.next_out = (Bytef *)scratch.data();
                    zs.avail_out = blk.isize;
                    ok = inflate(&zs, Z_FINISH) == Z_STREAM_END && zs.avail_out == 0;
                } else {
                    inf->reset(addr_ + blk.payload, addr_ + blk.payload + blk.csize);
                    size_t produced = 0;
                    ok = inf->run(scratch.data(), scratch.data() + scratch.size(), 0, &produced) == Inflater::STREAM_END && produced == blk.isize;
                }
                ok = ok && crc32_fast(0, scratch.data(), blk.isize) == blk.crc;
                if (ok && blk.isize) memcpy(dst + (cum_[i] - base), scratch.data(), blk.isize);
            }
            bool last;
            {
                std::lock_guard<std::mutex> g(mu_);
                jobBusy_--;
                if (ok) jobDone_ += b - a;
                else if (error_.empty()) error_ = "corrupt BGZF block in " + path_;
                last = !error_.empty() ? jobBusy_ == 0 : jobDone_ == jobB1_ - jobB0_;
            }
            if (last) cvDone_.notify_all();
            if (!ok) { cv_.notify_all(); break; }
        }
        if (use_zlib) inflateEnd(&zs);
    }
    void fail(const std::string &msg) {
        {
            std::lock_guard<std::mutex> g(mu_);
            if (error_.empty()) error_ = msg;
        }
        cv_.notify_all();
        cvDone_.notify_all();
    }

    const unsigned char *addr_;
    std::vector<Block> blocks_;
    std::vector<size_t> cum_;              // text bytes in front of every block
    bool direct_ = true;
    size_t fillNext_ = 0;                  // first block the next fill() takes
    char *jobDst_ = nullptr;               // the current job: blocks [jobB0_, jobB1_) to jobDst_ + (cum_[b] - cum_[jobB0_]); mu_ held
    size_t jobB0_ = 0, jobB1_ = 0, jobNext_ = 0, jobDone_ = 0, jobBusy_ = 0;
    bool pending_ = false;                 // a request is out (the caller's thread only)
    std::condition_variable cvDone_;       // the caller waits here; the pool waits on cv_
    std::vector<size_t> groups_;           // first block of every group, then the block count
    std::string path_;
    std::map<size_t, Text> ready_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<std::thread> pool_;
    size_t next_ = 0, cur_ = 0, pos_ = 0, window_ = 4;
    bool stop_ = false;
    std::string error_;
};

class ReadFeeder {
public:
    using Alloc = std::function<void *(size_t)>;
    using Free = std::function<void(void *)>;

    // consumers: how many threads call next(), each holding up to two batches at a time (the one on the device and the one it staged
    // ahead).  The pool is sized for that: with fewer buffers than 2 x consumers every buffer can be somebody's scanned-or-staged batch
    // while all of them wait in next() for a worker that has no buffer to parse into (round-3 ADVICE: --gpus 4 --threads 1 hung).
    ReadFeeder(std::vector<std::string> files, size_t chunkBytes, int threads, uint64_t maxReadsPerFile, Alloc alloc, Free freeFn, int consumers = 1)
        : files_(std::move(files)), chunk_(chunkBytes), maxReads_(maxReadsPerFile), free_(std::move(freeFn)),
          fileDone_(files_.size() ? files_.size() : 1) {
        for (auto &f : fileDone_) f.store(false);
        if (threads < 1) threads = 1;
        // Page-locking memory costs ~1 ms per MB, so buffers are kept few and small: a worker that packs to 2 bits needs a
        // quarter of the chunk (+ padding of every read to a 64-base unit), which pays for 12 packing workers where 6
        // copying ones were the limit; a buffer grows to the full chunk only if its chunk has to be delivered as ASCII.
        gzThreads_ = threads > 24 ? 24 : threads;      // decompression is plain CPU work: it may use more threads than there are buffers
        // (24 packing workers since round 3: two consumers drain 29 GB/s of FASTA and then wait for the parsers; the buffers are
        // page-locked by the workers themselves, side by side, the first time each is used)
        if (threads > (pack_ ? 24 : 6)) threads = pack_ ? 24 : 6;
        alloc_ = std::move(alloc);
        nThreads_ = threads;
        if (consumers < 1) consumers = 1;
        const int nbuf = threads + 1 + 2 * consumers;   // one per worker + what every consumer holds: the batch on the device and the one uploading
        for (int i = 0; i < nbuf; i++) {
            ReadBatch *b = new ReadBatch();
            b->cap = pack_ ? chunk_ / 4 + chunk_ / 32 + 4096 : chunk_ + 64;
            b->bases = nullptr;                     // page-locked on first use (ensure_bases), like the quality buffers (FASTQ only)
            all_.push_back(b);
            freeList_.push_back(b);
        }
        splitter_ = std::thread([this] { split(); });
        for (int i = 0; i < threads; i++) workers_.emplace_back([this] { work(); });
    }

    ~ReadFeeder() {
        {
            std::lock_guard<std::mutex> g(mu_);
            stop_ = true;
        }
        cvWork_.notify_all(); cvFree_.notify_all(); cvDone_.notify_all();
        if (splitter_.joinable()) splitter_.join();
        for (auto &t : workers_) if (t.joinable()) t.join();
        for (ReadBatch *b : all_) { if (b->bases) free_(b->bases); if (b->quals) free_(b->quals); delete b; }
        for (auto &m : maps_) if (m.addr) munmap((void *)m.addr, m.len);
    }

    // Next batch in read order, nullptr at the end.  Give it back with recycle().
    ReadBatch *next() {
        for (;;) {
            ReadBatch *b = nullptr;
            {
                std::unique_lock<std::mutex> g(mu_);
                cvDone_.wait(g, [&] { return stop_ || !error_.empty() || done_.count(nextSeq_) || (splitDone_ && nextSeq_ >= totalSeq_); });
                if (!error_.empty()) throw std::runtime_error(error_);
                if (stop_) return nullptr;
                auto it = done_.find(nextSeq_);
                if (it == done_.end()) return nullptr;   // all chunks delivered
                b = it->second;
                done_.erase(it);
                nextSeq_++;
            }
            // per-file read cap (the reference's `_maxReads`: readIndexPerDataset > maxReads stops the file)
            if (maxReads_ > 0) {
                if ((size_t)b->file >= perFile_.size()) perFile_.resize(b->file + 1, 0);
                uint64_t &seen = perFile_[b->file];
                const uint64_t allowed = maxReads_ + 1 > seen ? maxReads_ + 1 - seen : 0;
                if (b->n() > allowed) {
                    b->offsets.resize(allowed + 1); b->nbases = b->offsets.back();
                    if (b->packed) { b->wordOff.resize(allowed + 1); b->lens.resize(allowed); }
                }
                seen += b->n();
                if (seen >= maxReads_ + 1) fileDone_[b->file].store(true);   // later chunks of this file are skipped
            }
            if (b->n() == 0) { recycle(b); continue; }
            return b;
        }
    }

    void recycle(ReadBatch *b) {
        b->clear();
        {
            std::lock_guard<std::mutex> g(mu_);
            freeList_.push_back(b);
        }
        cvFree_.notify_all();
        cvWork_.notify_all();
    }

private:
    struct Mapping { const char *addr = nullptr; size_t len = 0; };
    // [begin, end) lies in a memory-mapped file, or in `slab` (inflated text of a gzip file, released with the item)
    struct Work { uint64_t seq; int file; const char *begin; const char *end; bool fastq; std::shared_ptr<char> slab; };

    static bool is_gzip(const std::string &path) {
        unsigned char m[2] = {0, 0};
        int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("File not found: " + path);
        ssize_t n = read(fd, m, 2);
        close(fd);
        return n == 2 && m[0] == 0x1f && m[1] == 0x8b;
    }

    static bool is_record_start(const char *c, const char *end, bool fastq) {
        if (!fastq) return *c == '>';
        if (*c != '@') return false;
        // a header is followed two lines later by a '+' line (a quality line starting with '@' is not)
        const char *l1 = (const char *)memchr(c, '\n', (size_t)(end - c));
        const char *l2 = l1 ? (const char *)memchr(l1 + 1, '\n', (size_t)(end - l1 - 1)) : nullptr;
        return l2 && l2 + 1 < end && l2[1] == '+';
    }

    // last record start in (p, lim]; p if there is none
    static const char *last_record_before(const char *p, const char *lim, const char *end, bool fastq) {
        const char *r = lim;
        while (r > p) {
            const char *nl = (const char *)memrchr(p, '\n', (size_t)(r - p));
            if (!nl) return p;
            const char *cand = nl + 1;
            if (cand < end && cand <= lim && is_record_start(cand, end, fastq)) return cand;
            r = nl;
        }
        return p;
    }

    void push_work(Work w) {
        {
            std::lock_guard<std::mutex> g(mu_);
            work_.push_back(std::move(w));
        }
        cvWork_.notify_one();
    }

    void split() {
        try {
            uint64_t seq = 0;
            for (size_t f = 0; f < files_.size(); f++) {
                const std::string &path = files_[f];
                if (is_gzip(path)) {
                    // one deflate stream: this thread inflates, the workers parse and pack the inflated slabs
                    seq = read_gz(path, (int)f, seq);
                    continue;
                }
                int fd = open(path.c_str(), O_RDONLY);
                if (fd < 0) throw std::runtime_error("File not found: " + path);
                struct stat st;
                fstat(fd, &st);
                if (st.st_size == 0) { close(fd); continue; }
                const char *addr = (const char *)mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
                close(fd);
                if (addr == MAP_FAILED) throw std::runtime_error("mmap failed: " + path);
                madvise((void *)addr, (size_t)st.st_size, MADV_SEQUENTIAL);
                maps_.push_back({addr, (size_t)st.st_size});
                const char *begin = addr, *end = addr + st.st_size;
                while (begin < end && (*begin == '\n' || *begin == '\r')) begin++;
                if (begin == end) continue;
                const bool fastq = *begin == '@';
                if (!fastq && *begin != '>') throw std::runtime_error("not FASTA/FASTQ: " + path);
                if (fastq && !looks_four_line(begin, end)) {
                    // multi-line FASTQ, which kseq accepts (Commons.hpp:82): the sequential kseq-style reader, as for gzip
                    seq = read_gz_sequential(path, (int)f, seq);
                    continue;
                }
                const char *p = begin;
                while (p < end && !fileDone_[f].load()) {
                    const char *q = end;
                    if ((size_t)(end - p) > chunk_) {
                        q = last_record_before(p, p + chunk_, end, fastq);
                        if (q == p) throw std::runtime_error("a single read is larger than the batch size; raise --batch-bases");
                    }
                    push_work({seq++, (int)f, p, q, fastq, nullptr});
                    p = q;
                }
            }
            {
                std::lock_guard<std::mutex> g(mu_);
                totalSeq_ = seq;
                splitDone_ = true;
            }
            cvWork_.notify_all(); cvDone_.notify_all();
        } catch (const std::exception &e) { fail(e.what()); }
    }

    // used by the sequential gzip reader only: earlier (lower sequence number) work items must already own
    // their buffers, otherwise later batches could starve them
    ReadBatch *take_free() {
        std::unique_lock<std::mutex> g(mu_);
        cvFree_.wait(g, [&] { return stop_ || (!freeList_.empty() && work_.empty()); });
        if (stop_) return nullptr;
        ReadBatch *b = freeList_.back();
        freeList_.pop_back();
        g.unlock();
        ensure_bases(b);
        return b;
    }

    void ensure_bases(ReadBatch *b) {       // outside mu_: page-locking takes about a millisecond per MB
        if (b->bases) return;
        b->bases = (char *)alloc_(b->cap);
        if (!b->bases) throw std::runtime_error("page-locked batch allocation failed");
    }

    void deliver(uint64_t seq, ReadBatch *b) {
        {
            std::lock_guard<std::mutex> g(mu_);
            done_[seq] = b;
        }
        cvDone_.notify_all();
    }

    // First records of an inflated FASTQ slab: 4 lines each?  (Multi-line FASTQ goes to the sequential reader.)
    static bool looks_four_line(const char *p, const char *end) {
        for (int rec = 0; rec < 256 && p < end; rec++) {
            const char *l[4];
            const char *c = p;
            for (int i = 0; i < 4; i++) {
                l[i] = (const char *)memchr(c, '\n', (size_t)(end - c));
                if (!l[i]) return true;              // slab ends inside this record: nothing seen against 4 lines
                c = l[i] + 1;
            }
            if (*p != '@' || l[1][1] != '+') return false;
            size_t ns = (size_t)(l[1] - l[0] - 1), nq = (size_t)(l[3] - l[2] - 1);
            if (ns && l[1][-1] == '\r') ns--;
            if (nq && l[3][-1] == '\r') nq--;
            if (ns != nq) return false;
            p = c;
        }
        return true;
    }

    // gzip: this thread only inflates.  The text is cut at record starts into slabs of at most one chunk, which the
    // workers parse and pack like slices of a memory-mapped file; at most threads + 2 slabs exist at a time.
    uint64_t read_gz(const std::string &path, int file, uint64_t seq) {
        // BGZF: the blocks are inflated by a pool of threads, this thread only stitches the text into slabs
        Mapping map;                       // declared first: unmapped after the readers (and their threads) are gone
        struct Unmap { Mapping *m; ~Unmap() { if (m->addr) munmap((void *)m->addr, m->len); } } unmap{&map};
        std::unique_ptr<BgzfReader> bgzf;
        std::unique_ptr<GzipMemReader> gzmem;
        std::unique_ptr<ParallelGzipReader> gzpar;
        {
            int fd = open(path.c_str(), O_RDONLY);
            if (fd < 0) throw std::runtime_error("File not found: " + path);
            struct stat st;
            fstat(fd, &st);
            void *addr = st.st_size ? mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0) : MAP_FAILED;
            close(fd);
            if (addr != MAP_FAILED) {
                map = {(const char *)addr, (size_t)st.st_size};
                madvise(addr, map.len, MADV_SEQUENTIAL);
                std::vector<BgzfReader::Block> blocks;
                if (!getenv("MDBG_HOST_NO_BGZF") && BgzfReader::index((const unsigned char *)addr, map.len, blocks))
                {
                    // (measured on the 10 Gbp set, 256 hardware threads: 24, 64 and 128 inflating threads give 1.48, 1.65 and 1.67 s of
                    // readSelection -- beyond 24 the threads that drive the GPU start to wait for a core)
                    int bthreads = gzThreads_;
                    if (const char *e = getenv("MDBG_HOST_BGZF_THREADS")) bthreads = std::max(1, atoi(e));
                    bgzf.reset(new BgzfReader((const unsigned char *)addr, std::move(blocks), bthreads, path));
                }
                else if (!zlib_inflate_requested()) {
                    // an ordinary gzip stream: several decoding threads when the file is worth it (gzip_parallel.hpp), else one
                    // (decoding without the window costs about twice the work: below six threads one thread is as fast)
                    int gthreads = gzThreads_ >= 6 ? gzThreads_ : 1;
                    if (const char *e = getenv("MDBG_HOST_GZIP_THREADS")) gthreads = atoi(e);
                    size_t gchunk = (size_t)4 << 20;
                    if (const char *e = getenv("MDBG_HOST_GZIP_CHUNK")) gchunk = (size_t)atoll(e);
                    if (gthreads >= 2 && map.len >= 4 * gchunk) {
                        gzpar.reset(new ParallelGzipReader((const uint8_t *)addr, map.len, gthreads, path, gchunk));
                        if (!gzpar->usable()) gzpar.reset();                  // no block header to cut at: one thread
                    }
                    if (!gzpar) gzmem.reset(new GzipMemReader((const uint8_t *)addr, map.len, path));
                }
            }
        }
        gzFile fp = nullptr;
        if (!bgzf && !gzmem && !gzpar) {
            fp = gzopen(path.c_str(), "r");
            if (!fp) throw std::runtime_error("File not found: " + path);
            gzbuffer(fp, 1 << 20);
        }
        struct Closer { gzFile f; ~Closer() { if (f) gzclose(f); } } closer{fp};
        if (bgzf && bgzf->direct()) {
            bool multiline = false;
            seq = stitch_direct(*bgzf, path, file, seq, &multiline);
            if (!multiline) return seq;
            bgzf.reset();
            return read_gz_sequential(path, file, seq);
        }
        if (gzpar) {
            bool multiline = false;
            seq = stitch_direct(*gzpar, path, file, seq, &multiline);
            if (!multiline) return seq;
            gzpar.reset();
            return read_gz_sequential(path, file, seq);
        }
        std::vector<char> carry;          // text behind the last cut: an incomplete record and the look-ahead
        bool first = true, fastq = false, eof = false;
        const size_t maxSlabs = (size_t)nThreads_ + 2;
        // a record start is recognised from the two lines that follow it, so text is inflated some way past the
        // chunk before the cut is chosen (enough for reads of a few Mbp; longer ones need a larger --batch-bases)
        const size_t cap = chunk_ + std::min<size_t>(chunk_, (size_t)4 << 20);
        // slabs go round: a fresh 36 MB array per slab is 9 k page faults on the one thread everything waits for
        struct SlabPool {
            std::mutex mu;
            std::vector<char *> idle;
            ~SlabPool() { for (char *p : idle) delete[] p; }
        };
        auto pool = std::make_shared<SlabPool>();
        auto take_slab = [&]() {
            char *p = nullptr;
            {
                std::lock_guard<std::mutex> g(pool->mu);
                if (!pool->idle.empty()) { p = pool->idle.back(); pool->idle.pop_back(); }
            }
            if (!p) p = new char[cap + 1];
            return std::shared_ptr<char>(p, [pool](char *q) { std::lock_guard<std::mutex> g(pool->mu); pool->idle.push_back(q); });
        };
        while ((!eof || !carry.empty()) && !fileDone_[file].load()) {
            {
                std::unique_lock<std::mutex> g(mu_);
                cvFree_.wait(g, [&] { return stop_ || slabsOut_ < maxSlabs; });
                if (stop_) return seq;
            }
            std::shared_ptr<char> slab = take_slab();
            char *buf = slab.get();
            size_t len = carry.size();
            if (len) memcpy(buf, carry.data(), len);
            carry.clear();
            while (!eof && len < cap) {
                long n;
                if (bgzf) n = (long)bgzf->read(buf + len, cap - len);
                else if (gzmem) n = (long)gzmem->read(buf + len, cap - len);
                else n = gzread(fp, buf + len, (unsigned)std::min<size_t>(cap - len, (size_t)1 << 30));
                if (n < 0) throw std::runtime_error("gzip read error: " + path);
                if (n == 0) {
                    if (fp) {   // zlib: a truncated file ends with 0 bytes read and an error state
                        int err = Z_OK;
                        gzerror(fp, &err);
                        if (err != Z_OK && err != Z_STREAM_END) throw std::runtime_error("gzip read error (truncated file?): " + path);
                    }
                    eof = true;
                    break;
                }
                len += (size_t)n;
            }
            const char *begin = buf, *end = buf + len;
            if (first) {
                while (begin < end && (*begin == '\n' || *begin == '\r')) begin++;
                if (begin == end) { if (eof) break; continue; }
                fastq = *begin == '@';
                if (!fastq && *begin != '>') throw std::runtime_error("not FASTA/FASTQ: " + path);
                if (fastq && !looks_four_line(begin, end)) {
                    if (fp) gzclose(fp);
                    closer.f = nullptr;
                    bgzf.reset();
                    gzmem.reset();
                    return read_gz_sequential(path, file, seq);
                }
                first = false;
            }
            const char *q = end;
            if ((size_t)(end - begin) > chunk_) {
                q = last_record_before(begin, begin + chunk_, end, fastq);
                if (q == begin) throw std::runtime_error("a single read is larger than the batch size; raise --batch-bases");
            }
            carry.assign(q, end);
            {
                std::lock_guard<std::mutex> g(mu_);
                slabsOut_++;
            }
            push_work({seq++, file, begin, q, fastq, slab});
        }
        return seq;
    }

    // Compressed text whose decoder can put it straight to its place (Src = BgzfReader: the pool inflates every block into the slab;
    // ParallelGzipReader: the pool translates every chunk's symbols into the slab): the slab AFTER the one being
    // cut is already filling while the cut is chosen: its text starts `head` bytes into the slab, and what the cut leaves over (an
    // incomplete record and the look-ahead) is copied in front of it afterwards -- the only text this thread touches.  How much the
    // next slab is asked to hold is steered so that the left-over stays about `look` bytes.
    template <typename Src>
    uint64_t stitch_direct(Src &bgzf, const std::string &path, int file, uint64_t seq, bool *multiline) {
        const size_t look = std::max<size_t>(std::min<size_t>(chunk_, (size_t)4 << 20), (size_t)128 << 10);   // (a block is up to 64 KB)
        const size_t head = 2 * look;
        const size_t cap = head + chunk_ + 2 * look;
        const size_t maxSlabs = (size_t)nThreads_ + 3;
        // slabs go round: a fresh 40 MB array per slab is 10 k page faults
        struct SlabPool {
            std::mutex mu;
            std::vector<char *> idle;
            ~SlabPool() { for (char *p : idle) delete[] p; }
        };
        auto pool = std::make_shared<SlabPool>();
        auto take_slab = [&]() -> std::shared_ptr<char> {
            {
                std::unique_lock<std::mutex> g(mu_);
                cvFree_.wait(g, [&] { return stop_ || slabsOut_ < maxSlabs; });
                if (stop_) return nullptr;
                slabsOut_++;
            }
            char *p = nullptr;
            {
                std::lock_guard<std::mutex> g(pool->mu);
                if (!pool->idle.empty()) { p = pool->idle.back(); pool->idle.pop_back(); }
            }
            if (!p) p = new char[cap + 1];
            return std::shared_ptr<char>(p, [pool](char *q) { std::lock_guard<std::mutex> g(pool->mu); pool->idle.push_back(q); });
        };
        auto give_back = [&]() {                          // a slab taken and not handed to the parsers
            std::lock_guard<std::mutex> g(mu_);
            slabsOut_--;
        };
        struct Settle { Src &b; ~Settle() { b.fill_settle(); } } settle{bgzf};   // declared after the slabs' pool, runs before it goes

        std::shared_ptr<char> cur = take_slab();
        if (!cur) return seq;
        size_t got = bgzf.fill_begin(cur.get() + head, chunk_ + look);
        bgzf.fill_wait();
        const char *begin = cur.get() + head, *end = begin + got;
        while (begin < end && (*begin == '\n' || *begin == '\r')) begin++;
        while (begin == end && !bgzf.at_end()) {        // a file that starts with empty lines only, a block's worth of them
            got = bgzf.fill_begin(cur.get() + head, chunk_ + look);
            bgzf.fill_wait();
            begin = cur.get() + head; end = begin + got;
            while (begin < end && (*begin == '\n' || *begin == '\r')) begin++;
        }
        if (begin == end) { give_back(); return seq; }
        const bool fastq = *begin == '@';
        if (!fastq && *begin != '>') throw std::runtime_error("not FASTA/FASTQ: " + path);
        if (fastq && !looks_four_line(begin, end)) { give_back(); *multiline = true; return seq; }

        bool held = true;                                 // `cur` is counted among the slabs out and not with the parsers
        for (;;) {
            if (fileDone_[file].load()) break;            // (the per-file read cap was reached)
            const size_t total = (size_t)(end - begin);
            const bool lastSlab = bgzf.at_end();
            // the cut first: the parsers get the slab before the next one is asked for (a request may have to wait for chunks that
            // are not decoded yet)
            const char *q = end;
            if (!lastSlab || total > chunk_) {
                if (total > chunk_) q = last_record_before(begin, begin + chunk_, end, fastq);
                else q = begin;                           // (cannot be: every request reaches past the chunk) keep everything for the next slab
                if (q == begin && total > chunk_) throw std::runtime_error("a single read is larger than the batch size; raise --batch-bases");
            }
            const size_t left = (size_t)(end - q);
            if (q > begin) push_work({seq++, file, begin, q, fastq, cur});      // (slabsOut_ counted when the slab was taken)
            else give_back();
            held = false;
            if (lastSlab && !left) break;
            // the next slab: its text `head` bytes in, so much of it that with the left-over in front it reaches `look` past the chunk
            std::shared_ptr<char> nxt = take_slab();
            if (!nxt) return seq;
            size_t nxtGot = 0;
            if (!lastSlab) {
                size_t want = chunk_ + look > left ? chunk_ + look - left : 0;
                want = std::max<size_t>(want, (size_t)128 << 10);
                nxtGot = bgzf.fill_begin(nxt.get() + head, std::min(want, cap - head));
            }
            if (left <= head) memcpy(nxt.get() + head - left, q, left);         // beside the pool, which writes behind `head`
            bgzf.fill_wait();
            if (left <= head) {
                begin = nxt.get() + head - left;
                end = nxt.get() + head + nxtGot;
            } else {
                // a left-over longer than the room in front (reads of several Mbp): a slab of its own size, outside the pool
                std::shared_ptr<char> big(new char[left + nxtGot + 1], std::default_delete<char[]>());
                memcpy(big.get(), q, left);
                memcpy(big.get() + left, nxt.get() + head, nxtGot);
                nxt = big;                                // (the pooled slab goes back; the count of slabs out stays)
                begin = big.get();
                end = begin + left + nxtGot;
            }
            cur = nxt;
            held = true;
        }
        if (held) give_back();
        return seq;
    }

    // gzip, multi-line FASTQ: sequential decode and parse on the splitter thread itself
    uint64_t read_gz_sequential(const std::string &path, int file, uint64_t seq) {
        FastxReader rd(path);
        if (!rd.ok()) throw std::runtime_error("File not found: " + path);
        std::string s, q;
        ReadBatch *b = take_free();
        if (!b) return seq;
        b->file = file;
        for (;;) {
            s.clear(); q.clear();
            bool hq = false;
            if (fileDone_[file].load()) break;
            if (!rd.next(s, q, hq)) break;
            if (s.size() > chunk_) throw std::runtime_error("a single read is larger than the batch size; raise --batch-bases");
            if ((b->n() && hq != b->hasQual) || b->nbases + s.size() > chunk_) {
                deliver(seq++, b);
                b = take_free();
                if (!b) return seq;
                b->file = file;
            }
            b->hasQual = hq;
            need_ascii_capacity(b);
            memcpy(b->bases + b->nbases, s.data(), s.size());
            if (hq) { need_quals(b); memcpy(b->quals + b->nbases, q.data(), s.size()); }
            b->nbases += s.size();
            b->offsets.push_back(b->nbases);
        }
        deliver(seq++, b);
        return seq;
    }

    void need_quals(ReadBatch *b) {
        if (!b->quals) { b->quals = (char *)alloc_(chunk_ + 64); if (!b->quals) throw std::runtime_error("page-locked batch allocation failed"); }
    }
    void need_ascii_capacity(ReadBatch *b) {
        if (b->cap >= chunk_ + 64 && b->bases) return;
        if (b->bases) free_(b->bases);
        if (b->cap < chunk_ + 64) b->cap = chunk_ + 64;
        b->bases = (char *)alloc_(b->cap);
        if (!b->bases) throw std::runtime_error("page-locked batch allocation failed");
    }

    // Same walk over the records as parse(), bases packed to 2 bits on the way.  false = deliver this chunk as ASCII.
    bool parse_packed(const Work &w, ReadBatch *b) {
        b->file = w.file;
        b->hasQual = w.fastq;
        if (w.fastq) need_quals(b);
        PackCursor pc{reinterpret_cast<uint64_t *>(b->bases), b->cap / 8};
        const char *p = w.begin, *end = w.end;
        while (p < end) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));   // header line
            if (!nl) break;
            p = nl + 1;
            size_t len = 0;
            const char *seq0 = p;                 // first sequence line of the record
            if (!w.fastq) {
                while (p < end && *p != '>') {
                    nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                    const char *le = nl ? nl : end;
                    size_t n = (size_t)(le - p);
                    if (n && p[n - 1] == '\r') n--;
                    pc.append(p, n);
                    len += n;
                    p = nl ? nl + 1 : end;
                }
            } else {
                nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                const char *le = nl ? nl : end;
                size_t n = (size_t)(le - p);
                if (n && p[n - 1] == '\r') n--;
                pc.append(p, n);
                len = n;
                p = nl ? nl + 1 : end;
                if (p >= end || *p != '+') throw std::runtime_error("FASTQ records are not 4-line; unwrap or gzip the file");
                nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                if (!nl) throw std::runtime_error("truncated FASTQ record");
                p = nl + 1;
                nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                le = nl ? nl : end;
                size_t nq = (size_t)(le - p);
                if (nq && p[nq - 1] == '\r') nq--;
                if (nq != n) throw std::runtime_error("FASTQ quality length differs from sequence length");
                memcpy(b->quals + b->nbases, p, n);
                p = nl ? nl + 1 : end;
            }
            pc.end_read();
            if (pc.overflow || len > 0xFFFFFFF0ull) return false;
            if (pc.invalid) {
                // something else than upper-case ACGT in this read: it goes along a second time as characters.  Many such reads
                // (a lower-case file, an assembly full of N): the chunk is delivered as ASCII as before.
                if (b->odd.size() >= 64 && b->odd.size() * 8 > b->lens.size() + 64) return false;
                b->odd.push_back((uint32_t)b->lens.size());
                if (!w.fastq) {
                    for (const char *q = seq0; q < end && *q != '>';) {
                        const char *nl2 = (const char *)memchr(q, '\n', (size_t)(end - q));
                        const char *le = nl2 ? nl2 : end;
                        size_t n = (size_t)(le - q);
                        if (n && q[n - 1] == '\r') n--;
                        b->oddBases.append(q, n);
                        q = nl2 ? nl2 + 1 : end;
                    }
                } else b->oddBases.append(seq0, len);
                b->oddOff.push_back(b->oddBases.size());
                pc.invalid = 0;
            }
            b->nbases += len;
            b->offsets.push_back(b->nbases);
            b->wordOff.push_back(pc.w);
            b->lens.push_back((uint32_t)len);
        }
        b->packed = true;
        return true;
    }

    void parse(const Work &w, ReadBatch *b) {
        b->file = w.file;
        b->hasQual = w.fastq;
        if (w.fastq) need_quals(b);
        need_ascii_capacity(b);
        const char *p = w.begin, *end = w.end;
        while (p < end) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));   // header line
            if (!nl) break;
            p = nl + 1;
            if (!w.fastq) {
                // sequence lines until the next '>' at a line start
                while (p < end && *p != '>') {
                    nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                    const char *le = nl ? nl : end;
                    size_t n = (size_t)(le - p);
                    if (n && p[n - 1] == '\r') n--;
                    memcpy(b->bases + b->nbases, p, n);
                    b->nbases += n;
                    p = nl ? nl + 1 : end;
                }
            } else {
                nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                const char *le = nl ? nl : end;
                size_t n = (size_t)(le - p);
                if (n && p[n - 1] == '\r') n--;
                memcpy(b->bases + b->nbases, p, n);
                p = nl ? nl + 1 : end;
                if (p >= end || *p != '+') throw std::runtime_error("FASTQ records are not 4-line; unwrap or gzip the file");
                nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                if (!nl) throw std::runtime_error("truncated FASTQ record");
                p = nl + 1;
                nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                le = nl ? nl : end;
                size_t nq = (size_t)(le - p);
                if (nq && p[nq - 1] == '\r') nq--;
                if (nq != n) throw std::runtime_error("FASTQ quality length differs from sequence length");
                memcpy(b->quals + b->nbases, p, n);
                b->nbases += n;
                p = nl ? nl + 1 : end;
            }
            b->offsets.push_back(b->nbases);
        }
    }

    void work() {
        try {
            for (;;) {
                Work w;
                ReadBatch *b = nullptr;
                {
                    // a work item is taken only together with a buffer: items leave the queue in sequence
                    // order, so the batch the consumer waits for always owns a buffer (no deadlock)
                    std::unique_lock<std::mutex> g(mu_);
                    cvWork_.wait(g, [&] { return stop_ || (!work_.empty() && !freeList_.empty()) || (splitDone_ && work_.empty()); });
                    if (stop_) return;
                    if (work_.empty()) return;
                    w = std::move(work_.front());
                    work_.pop_front();
                    b = freeList_.back();
                    freeList_.pop_back();
                }
                cvFree_.notify_all();   // the sequential gzip reader waits for an empty work queue
                ensure_bases(b);
                if (!fileDone_[w.file].load()) {
                    if (!pack_ || !parse_packed(w, b)) { b->clear(); parse(w, b); }
                } else b->file = w.file;
                if (!w.slab && w.end > w.begin) {
                    // a chunk of a memory-mapped plain file, parsed: its pages are let go of here, by the worker, 32 MB at a time --
                    // not at the end, all 50 GB at once, under the process's exit (0.3 s of page-table teardown nobody overlaps)
                    static const uintptr_t page = (uintptr_t)std::max<long>(4096, sysconf(_SC_PAGESIZE));
                    const uintptr_t a0 = ((uintptr_t)w.begin + page - 1) & ~(page - 1), a1 = (uintptr_t)w.end & ~(page - 1);
                    if (a1 > a0) (void)madvise((void *)a0, (size_t)(a1 - a0), MADV_DONTNEED);
                }
                if (w.slab) {
                    w.slab.reset();
                    {
                        std::lock_guard<std::mutex> g(mu_);
                        slabsOut_--;
                    }
                    cvFree_.notify_all();
                }
                deliver(w.seq, b);
            }
        } catch (const std::exception &e) { fail(e.what()); }
    }

    void fail(const std::string &msg) {
        {
            std::lock_guard<std::mutex> g(mu_);
            if (error_.empty()) error_ = msg;
        }
        cvDone_.notify_all(); cvWork_.notify_all(); cvFree_.notify_all();
    }

    std::vector<std::string> files_;
    size_t chunk_;
    uint64_t maxReads_;
    Alloc alloc_;
    Free free_;
    std::vector<ReadBatch *> all_, freeList_;
    std::deque<Work> work_;
    std::map<uint64_t, ReadBatch *> done_;
    std::vector<Mapping> maps_;
    std::vector<uint64_t> perFile_;
    std::vector<std::atomic<bool>> fileDone_;
    std::mutex mu_;
    std::condition_variable cvWork_, cvFree_, cvDone_;
    std::thread splitter_;
    std::vector<std::thread> workers_;
    uint64_t nextSeq_ = 0, totalSeq_ = 0;
    size_t slabsOut_ = 0;   // inflated gzip slabs queued or being parsed
    int nThreads_ = 1, gzThreads_ = 1;
    bool splitDone_ = false, stop_ = false;
    bool pack_ = getenv("MDBG_HOST_NO_PACK") == nullptr;   // pack to 2 bits on the host unless asked not to
    std::string error_;
};

}  // namespace mdbg_host
