// hostfeed.hpp -- parallel FASTA / FASTQ ingestion for the host tool.
//
// The reference parses with one kseq reader inside an `omp critical` (Commons.hpp:5868-5905), which
// caps it near 0.5 Gbp/s whatever the thread count.  Here a plain (uncompressed) file is mmap'ed and
// cut at record boundaries into chunks that worker threads parse independently into page-locked
// batches (read order preserved by sequence numbers); gzip files fall back to a sequential reader
// thread (one deflate stream cannot be split).  Batches come out in file order, ready for
// mdbg_reads_from_ascii.
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "fastx.hpp"

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
    // while they parse (a quarter of the PCIe bytes); a chunk holding a character with bit 3 set (N, n, ...) or too
    // many tiny reads for the buffer is delivered as ASCII instead.
    bool packed = false;
    std::vector<uint64_t> wordOff{0};
    std::vector<uint32_t> lens;
    int file = 0;
    uint32_t n() const { return (uint32_t)(offsets.size() - 1); }
    const uint64_t *words() const { return reinterpret_cast<const uint64_t *>(bases); }
    void clear() { nbases = 0; offsets.assign(1, 0); hasQual = false; packed = false; wordOff.assign(1, 0); lens.clear(); }
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

// Appends bases to a stream of u64 words.  `invalid` collects bit 3 of every character seen.
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
            invalid |= (x0 | x1 | x2 | x3);
            const uint64_t a = PEXT ? pack8_pext(x0) : pack8_swar(x0), b = PEXT ? pack8_pext(x1) : pack8_swar(x1);
            const uint64_t c = PEXT ? pack8_pext(x2) : pack8_swar(x2), d = PEXT ? pack8_pext(x3) : pack8_swar(x3);
            words[w++] = a | (b << 16) | (c << 32) | (d << 48);
            p += 32; n -= 32;
        }
        while (n) { push1((unsigned char)*p++); n--; }
    }
    void append(const char *p, size_t n) { if (have_bmi2()) append_impl<true>(p, n); else append_impl<false>(p, n); }
    void push1(unsigned char c) {
        invalid |= c;
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
    bool bad() const { return overflow || (invalid & 0x0808080808080808ull) != 0; }
};

class ReadFeeder {
public:
    using Alloc = std::function<void *(size_t)>;
    using Free = std::function<void(void *)>;

    ReadFeeder(std::vector<std::string> files, size_t chunkBytes, int threads, uint64_t maxReadsPerFile, Alloc alloc, Free free_)
        : files_(std::move(files)), chunk_(chunkBytes), maxReads_(maxReadsPerFile), free_(std::move(free_)),
          fileDone_(files_.size() ? files_.size() : 1) {
        for (auto &f : fileDone_) f.store(false);
        if (threads < 1) threads = 1;
        // Page-locking memory costs ~1 ms per MB, so buffers are kept few and small: a worker that packs to 2 bits needs a
        // quarter of the chunk (+ padding of every read to a 64-base unit), which pays for 12 packing workers where 6
        // copying ones were the limit; a buffer grows to the full chunk only if its chunk has to be delivered as ASCII.
        if (threads > (pack_ ? 12 : 6)) threads = pack_ ? 12 : 6;
        alloc_ = std::move(alloc);
        const int nbuf = threads + 2;
        for (int i = 0; i < nbuf; i++) {
            ReadBatch *b = new ReadBatch();
            b->cap = pack_ ? chunk_ / 4 + chunk_ / 32 + 4096 : chunk_ + 64;
            b->bases = (char *)alloc_(b->cap);      // quality buffers are allocated on first use (FASTQ only)
            if (!b->bases) throw std::runtime_error("page-locked batch allocation failed");
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
        for (ReadBatch *b : all_) { free_(b->bases); if (b->quals) free_(b->quals); delete b; }
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
    struct Work { uint64_t seq; int file; const char *begin; const char *end; bool fastq; bool gz; std::string path; };

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
                    // one deflate stream: a single sequential work item produces all its batches in order
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
                const char *p = begin;
                while (p < end && !fileDone_[f].load()) {
                    const char *q = end;
                    if ((size_t)(end - p) > chunk_) {
                        q = last_record_before(p, p + chunk_, end, fastq);
                        if (q == p) throw std::runtime_error("a single read is larger than the batch size; raise --batch-bases");
                    }
                    push_work({seq++, (int)f, p, q, fastq, false, ""});
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
        return b;
    }

    void deliver(uint64_t seq, ReadBatch *b) {
        {
            std::lock_guard<std::mutex> g(mu_);
            done_[seq] = b;
        }
        cvDone_.notify_all();
    }

    // gzip: sequential decode on the splitter thread itself
    uint64_t read_gz(const std::string &path, int file, uint64_t seq) {
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
        if (b->cap >= chunk_ + 64) return;
        free_(b->bases);
        b->cap = chunk_ + 64;
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
            if (pc.bad() || len > 0xFFFFFFF0ull) return false;
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
                if (!fileDone_[w.file].load()) {
                    if (!pack_ || !parse_packed(w, b)) { b->clear(); parse(w, b); }
                } else b->file = w.file;
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
    bool splitDone_ = false, stop_ = false;
    bool pack_ = getenv("MDBG_HOST_NO_PACK") == nullptr;   // pack to 2 bits on the host unless asked not to
    std::string error_;
};

}  // namespace mdbg_host
