// fastx.hpp -- FASTA / FASTQ (.gz or plain) record reader for the host tool, same record semantics
// as the kseq.h loop the reference parses with (KSEQ_INIT(gzFile, gzread), Commons.hpp:82, :5868-5905):
// multi-line sequences, '>' or '@' headers, optional '+' quality block (multi-line, length of the
// sequence).  Own implementation over zlib's gzread with a large buffer.
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace mdbg_host {

class FastxReader {
public:
    explicit FastxReader(const std::string &path) : buf_(1 << 22), path_(path) {
        fp_ = gzopen(path.c_str(), "r");
        if (fp_) gzbuffer(fp_, 1 << 20);
    }
    ~FastxReader() { if (fp_) gzclose(fp_); }
    bool ok() const { return fp_ != nullptr; }

    // Appends the next record's bases to `seq` and qualities to `qual` (nothing for FASTA).
    // Returns false at end of file.
    bool next(std::string &seq, std::string &qual, bool &has_qual) {
        int c;
        // find the next header
        if (last_ == 0) {
            while ((c = getc()) != -1 && c != '>' && c != '@') {}
            if (c == -1) return false;
            last_ = c;
        }
        // header line: skip it (names are not part of any output of this path)
        while ((c = getc()) != -1 && c != '\n') {}
        const size_t seq0 = seq.size();
        // sequence lines until the next record marker at a line start
        bool at_line_start = true;
        for (;;) {
            c = getc();
            if (c == -1) break;
            if (at_line_start && (c == '>' || c == '@' || c == '+')) break;
            if (c == '\n') { at_line_start = true; continue; }
            at_line_start = false;
            if (c != '\r' && c != ' ' && c != '\t') seq.push_back((char)c);   // kseq keeps isgraph() characters
        }
        has_qual = false;
        if (c == '>' || c == '@') { last_ = c; return true; }
        if (c == -1) { last_ = 0; return true; }        // last record of the file
        // '+' line, then qualities until as many as bases
        while ((c = getc()) != -1 && c != '\n') {}
        const size_t need = seq.size() - seq0;
        size_t got = 0;
        while (got < need && (c = getc()) != -1) {
            if (c >= 33 && c <= 127) { qual.push_back((char)c); got++; }
        }
        has_qual = true;
        last_ = 0;   // next header is searched for
        return true;
    }

private:
    int getc() {
        if (pos_ == len_) {
            if (eof_) return -1;
            int n = gzread(fp_, buf_.data(), (unsigned)buf_.size());
            if (n <= 0) {
                // damaged or truncated gzip data is an error, not the end of the reads
                int err = Z_OK;
                const char *msg = gzerror(fp_, &err);
                if (n < 0 || (err != Z_OK && err != Z_STREAM_END)) throw std::runtime_error("gzip read error in " + path_ + ": " + (msg ? msg : "?"));
                eof_ = true;
                return -1;
            }
            len_ = (size_t)n; pos_ = 0;
        }
        return (unsigned char)buf_[pos_++];
    }
    gzFile fp_ = nullptr;
    std::vector<char> buf_;
    std::string path_;
    size_t pos_ = 0, len_ = 0;
    bool eof_ = false;
    int last_ = 0;   // header character already consumed
};

}  // namespace mdbg_host
