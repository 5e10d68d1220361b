// inflate.hpp -- DEFLATE (RFC 1951) / gzip (RFC 1952) decoder for the host feed.
//
// Compressed reads are bound by the decoder, not by the GPU: zlib's inflate delivers a few hundred MB/s of text per
// stream.  This is an own implementation shaped for what the feed needs and zlib's stream API cannot assume: the WHOLE
// compressed file is in memory (memory-mapped), so the hot loop refills a 64-bit bit buffer with unaligned 8-byte
// loads, decodes through wide single-level-mostly tables (11 bits for literals/lengths, 8 for distances), emits up to
// four literals per table look-up (a second table holds, for every 11-bit index, the run of literals it decodes to on
// its own), resolves a whole match -- length code, its extra bits and the distance code -- with ONE look-up where the three
// fit the 11 index bits (a third table; reads compressed at the fast levels are such matches almost exclusively, and the
// loop is a chain of dependent look-ups) and copies matches 8 bytes at a time.  Output is produced in caller-sized pieces: run()
// stops in front of a symbol when fewer than MIN_ROOM (288) elements of room are left, so a match is never split and the only
// state carried between calls is the bit buffer, the current block's tables and the remainder of a stored block.
// Back-references reach into the text already produced, which the caller keeps directly in front of the output
// pointer (the last 32 KB suffice).
//
// Every gzip member's CRC-32 and length are checked by the caller (hostfeed.hpp): a decoding error ends the run with a
// message, never with a wrong read (MDBG_HOST_ZLIB_INFLATE=1 switches the feed back to zlib's inflate).
// tests/host/test_inflate.cpp checks the decoder against zlib over streams of every block type, compression level and
// strategy, with the output room cut at random places, and feeds it damaged streams under the address sanitizer.
#pragma once

#include <cstddef>
#include <cstdint>
#include <cstring>
#include <initializer_list>

namespace mdbg_host {

// OutT = uint8_t: the text.  OutT = uint16_t: symbols for decoding WITHOUT the preceding 32 KB (gzip_parallel.hpp): values
// < 256 are bytes, larger ones stand for a position of the unknown window and are copied around like bytes.
template <typename OutT>
class InflaterT {
public:
    enum Status { NEED_ROOM = 0, STREAM_END = 1, AT_STOP = 2, CORRUPT = -1 };
    // Room (in elements) one trip of the fast loop may write: four literal runs of up to four literals each (16; every run is
    // stored as four elements, the unused ones overwritten by what follows), then one maximal match (258) copied in 8-byte
    // steps that may overshoot its end by 7 bytes: 16 + 258 + 7 = 281, rounded up.  Callers pass the exact end of their
    // allocation as out_end, so this bound is what keeps the decoder inside it.
    static constexpr ptrdiff_t MIN_ROOM = 288;

    // start decoding a raw deflate stream at `in`; [in, in_end) must stay readable
    void reset(const uint8_t *in, const uint8_t *in_end) {
        base_ = in; in_ = in; in_end_ = in_end;
        bitbuf_ = 0; bitcnt_ = 0;
        state_ = BLOCK_HEADER; final_ = false; stored_left_ = 0;
        stop_bit_ = ~0ull; blocks_done_ = 0;
    }
    // start at a block header that lies `bit` bits behind `base` (gzip_parallel.hpp: decoding from the middle of a stream)
    void reset_at_bit(const uint8_t *base, uint64_t bit, const uint8_t *in_end) {
        reset(base, in_end);
        in_ = base + (bit >> 3);
        if (bit & 7) { bitbuf_ = (uint64_t)(*in_++ >> (bit & 7)); bitcnt_ = 8 - (int)(bit & 7); }
    }
    // bits consumed since `base`; exact at block boundaries
    uint64_t bit_position() const { return (uint64_t)(in_ - base_) * 8 - (uint64_t)bitcnt_; }
    // run() returns AT_STOP in front of the block header at this bit position (CORRUPT if the stream's block boundaries skip it)
    void set_stop_bit(uint64_t bit) { stop_bit_ = bit; }
    unsigned blocks_done() const { return blocks_done_; }
    bool in_final_block() const { return final_; }

    // Decodes into [out, out_end).  `hist` is the number of bytes of earlier output that lie directly in front of `out`
    // (at least min(32768, everything produced so far)).  *produced = bytes written.  Returns STREAM_END after the
    // final block (in_pos() is then the first byte behind the stream), NEED_ROOM when the room is used up (call again
    // with fresh room; up to MIN_ROOM elements of the old room may stay unused), CORRUPT on invalid data.
    Status run(OutT *out, OutT *out_end, size_t hist, size_t *produced) {
        OutT *const out0 = out;
        const OutT *const lowest = out - hist;
        Status st = NEED_ROOM;
        for (;;) {
            if (state_ == BLOCK_HEADER) {
                if (final_) { st = STREAM_END; break; }
                if (stop_bit_ != ~0ull) {
                    const uint64_t at = bit_position();
                    if (at == stop_bit_) { st = AT_STOP; break; }
                    if (at > stop_bit_) { st = CORRUPT; break; }
                }
                if (!need(3)) { st = CORRUPT; break; }
                final_ = take(1);
                const unsigned type = take(2);
                if (type == 0) {
                    drop_to_byte();
                    if (!need(32)) { st = CORRUPT; break; }
                    const unsigned len = take(16), nlen = take(16);
                    if ((len ^ 0xFFFFu) != nlen) { st = CORRUPT; break; }
                    stored_left_ = len;
                    state_ = STORED;
                } else if (type == 1) {
                    build_fixed();
                    state_ = HUFFMAN;
                } else if (type == 2) {
                    if (!read_dynamic()) { st = CORRUPT; break; }
                    state_ = HUFFMAN;
                } else { st = CORRUPT; break; }
            }
            if (state_ == STORED) {
                // the bit buffer holds whole bytes here: hand them back to the input first
                unread_bitbuf();
                size_t n = stored_left_;
                if ((size_t)(in_end_ - in_) < n) { st = CORRUPT; break; }
                const size_t room = (size_t)(out_end - out);
                if (n > room) n = room;
                if (sizeof(OutT) == 1) memcpy(out, in_, n);
                else for (size_t i = 0; i < n; i++) out[i] = (OutT)in_[i];
                out += n; in_ += n; stored_left_ -= (uint32_t)n;
                if (stored_left_) break;                      // room used up
                state_ = BLOCK_HEADER;
                blocks_done_++;
                continue;
            }
            // HUFFMAN
            const int r = decode_block(out, out_end, lowest);
            if (r < 0) { st = CORRUPT; break; }
            if (r == 0) break;                                // room used up
            state_ = BLOCK_HEADER;                            // end-of-block symbol
            blocks_done_++;
        }
        *produced = (size_t)(out - out0);
        return st;
    }

    // first byte behind the deflate stream (valid after STREAM_END)
    const uint8_t *in_pos() {
        drop_to_byte();
        unread_bitbuf();
        return in_;
    }

private:
    enum State { BLOCK_HEADER, STORED, HUFFMAN };
    static constexpr unsigned LIT_BITS = 11, DIST_BITS = 8;
    static constexpr uint32_t F_LITERAL = 1u << 15, F_EOB = 1u << 14, F_SUB = 1u << 13, F_INVALID = 1u << 12;
    // entry: bits [0,8) code length to drop, [8,12) extra bits (or sub-table bits), [12,16) flags, [16,32) literal /
    // length base / distance base / sub-table start
    static uint32_t entry(unsigned nbits, unsigned extra, uint32_t flags, unsigned value) {
        return nbits | (extra << 8) | flags | ((uint32_t)value << 16);
    }

    // ---- bit input ---------------------------------------------------------------------------------------------
    static uint64_t load64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }   // little-endian host (x86-64)
    // careful refill near the end of the input: byte by byte
    bool need(int n) {
        while (bitcnt_ < n) {
            if (in_ >= in_end_) return false;
            bitbuf_ |= (uint64_t)*in_++ << bitcnt_;
            bitcnt_ += 8;
        }
        return true;
    }
    unsigned take(int n) {
        const unsigned v = (unsigned)(bitbuf_ & ((1ull << n) - 1));
        bitbuf_ >>= n; bitcnt_ -= n;
        return v;
    }
    void drop_to_byte() { const int r = bitcnt_ & 7; bitbuf_ >>= r; bitcnt_ -= r; }
    void unread_bitbuf() { in_ -= bitcnt_ >> 3; bitbuf_ = 0; bitcnt_ = 0; }   // only with bitcnt_ % 8 == 0

    // ---- tables ------------------------------------------------------------------------------------------------
    // Canonical Huffman decoding table with `root` index bits and second-level tables for longer codes.
    // sym_entry(symbol) gives the entry without its code length.  Returns false for an over-subscribed or
    // (unless it is the single-code case zlib accepts) incomplete set.
    template <typename SymEntry>
    static bool build_table(const uint8_t *lens, unsigned nsym, unsigned root, uint32_t *table, unsigned table_cap, bool allow_incomplete,
                            SymEntry sym_entry) {
        unsigned count[16] = {0};
        for (unsigned s = 0; s < nsym; s++) count[lens[s]]++;
        count[0] = 0;
        unsigned maxlen = 15;
        while (maxlen > 0 && !count[maxlen]) maxlen--;
        for (unsigned i = 0; i < (1u << root); i++) table[i] = entry(1, 0, F_INVALID, 0);
        if (maxlen == 0) return allow_incomplete;             // no codes at all (a block without matches): never looked up
        int left = 1;
        for (unsigned l = 1; l <= 15; l++) {
            left = (left << 1) - (int)count[l];
            if (left < 0) return false;                       // over-subscribed
        }
        if (left > 0 && !(allow_incomplete && maxlen == 1)) return false;
        unsigned next_code[16];
        unsigned code = 0;
        for (unsigned l = 1; l <= 15; l++) { code = (code + count[l - 1]) << 1; next_code[l] = code; }
        unsigned used = 1u << root;                           // next free sub-table slot
        for (unsigned s = 0; s < nsym; s++) {
            const unsigned l = lens[s];
            if (!l) continue;
            const unsigned c = next_code[l]++;
            // deflate packs Huffman codes starting from their most significant bit: reverse to index by stream bits
            unsigned rev = 0;
            for (unsigned i = 0; i < l; i++) rev |= ((c >> i) & 1u) << (l - 1 - i);
            const uint32_t e = sym_entry(s);
            if (l <= root) {
                const uint32_t full = e | l;
                for (unsigned i = rev; i < (1u << root); i += 1u << l) table[i] = full;
            } else {
                const unsigned prefix = rev & ((1u << root) - 1);
                uint32_t &pe = table[prefix];
                if (!(pe & F_SUB)) {
                    // every sub-table is sized for the longest code of the whole set: never too small, and the bound
                    // (long codes x 2^(15 - root) entries) is what the table capacities are dimensioned for
                    const unsigned sub_bits = maxlen - root;
                    if (used + (1u << sub_bits) > table_cap) return false;
                    pe = entry(root, sub_bits, F_SUB, used);
                    for (unsigned i = 0; i < (1u << sub_bits); i++) table[used + i] = entry(1, 0, F_INVALID, 0);
                    used += 1u << sub_bits;
                }
                const unsigned sub_bits = (pe >> 8) & 15, start = pe >> 16;
                const uint32_t full = e | l;
                for (unsigned i = rev >> root; i < (1u << sub_bits); i += 1u << (l - root)) table[start + i] = full;
            }
        }
        return true;
    }

    static uint32_t litlen_entry(unsigned s) {
        static const uint16_t base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
        static const uint8_t extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
        if (s < 256) return entry(0, 0, F_LITERAL, s);
        if (s == 256) return entry(0, 0, F_EOB, 0);
        if (s < 286) return entry(0, extra[s - 257], 0, base[s - 257]);
        return entry(0, 0, F_INVALID, 0);                     // 286, 287: in the fixed code, never valid
    }
    static uint32_t dist_entry(unsigned s) {
        static const uint16_t base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073,
                                          4097, 6145, 8193, 12289, 16385, 24577};
        static const uint8_t extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
        if (s < 30) return entry(0, extra[s], 0, base[s]);
        return entry(0, 0, F_INVALID, 0);
    }

    void build_fixed() {
        uint8_t lens[288 + 32];
        for (unsigned i = 0; i < 144; i++) lens[i] = 8;
        for (unsigned i = 144; i < 256; i++) lens[i] = 9;
        for (unsigned i = 256; i < 280; i++) lens[i] = 7;
        for (unsigned i = 280; i < 288; i++) lens[i] = 8;
        for (unsigned i = 0; i < 32; i++) lens[288 + i] = 5;
        build_table(lens, 288, LIT_BITS, lit_, LIT_CAP, false, litlen_entry);
        build_table(lens + 288, 32, DIST_BITS, dist_, DIST_CAP, false, dist_entry);
        build_literal_runs();
        build_pairs();
    }

    // run_info_[i] / run_lits_[i]: the literals that the index bits i decode to on their own -- up to four, as long as
    // each next code is a literal that lies wholly inside the 11 bits.  info = bits consumed | count << 4; 0 = none.
    // pair_[i]: a match whose length code, length extra bits and distance code lie wholly inside the index bits i -- reads compressed at
    // the fast levels are such matches almost exclusively.  bits [0,4) code bits consumed in front of the distance's extra bits,
    // [4,8) number of those extra bits, [8,17) the length, [17,32) the distance's base.
    void build_pairs() {
        for (unsigned i = 0; i < (1u << LIT_BITS); i++) {
            pair_[i] = 0;
            const uint32_t e = lit_[i];
            if (e & (F_LITERAL | F_EOB | F_SUB | F_INVALID)) continue;
            const unsigned lb = e & 0xFF, lx = (e >> 8) & 15;
            if (lb + lx >= LIT_BITS) continue;
            const unsigned len = (e >> 16) + ((i >> lb) & ((1u << lx) - 1));
            const unsigned rest = LIT_BITS - (lb + lx);
            const uint32_t d = dist_[(i >> (lb + lx)) & ((1u << DIST_BITS) - 1)];      // the unknown high bits read as zero: fine while the code fits
            if ((d & (F_SUB | F_INVALID)) || (d & 0xFF) > rest) continue;
            pair_[i] = (lb + lx + (d & 0xFF)) | (((d >> 8) & 15) << 4) | (len << 8) | ((d >> 16) << 17);
        }
    }

    void build_literal_runs() {
        for (unsigned i = 0; i < (1u << LIT_BITS); i++) {
            unsigned pos = 0, count = 0;
            uint32_t lits = 0;
            while (count < 4) {
                const uint32_t e = lit_[i >> pos];            // the unknown high bits read as zero: fine while the code fits
                if (!(e & F_LITERAL) || (e & 0xFF) > LIT_BITS - pos) break;
                lits |= (e >> 16) << (8 * count);
                pos += e & 0xFF;
                count++;
            }
            run_info_[i] = (uint8_t)(count ? (pos | (count << 4)) : 0);
            run_lits_[i] = lits;
        }
    }

    bool read_dynamic() {
        if (!need(14)) return false;
        const unsigned hlit = take(5) + 257, hdist = take(5) + 1, hclen = take(4) + 4;
        if (hlit > 286 || hdist > 30) return false;
        static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
        uint8_t cl[19] = {0};
        for (unsigned i = 0; i < hclen; i++) {
            if (!need(3)) return false;
            cl[order[i]] = (uint8_t)take(3);
        }
        uint32_t cltab[128];
        if (!build_table(cl, 19, 7, cltab, 128, false, [](unsigned s) { return entry(0, 0, 0, s); })) return false;
        uint8_t lens[286 + 30 + 138];
        unsigned n = 0;
        const unsigned total = hlit + hdist;
        while (n < total) {
            if (!need(7 + 7)) {
                // near the end of the input fewer bits may be legitimate: only the code itself is required
                if (bitcnt_ == 0) return false;
            }
            const uint32_t e = cltab[bitbuf_ & 127];
            if (e & F_INVALID) return false;
            const unsigned l = e & 0xFF, sym = e >> 16;
            if ((int)l > bitcnt_) return false;
            bitbuf_ >>= l; bitcnt_ -= (int)l;
            if (sym < 16) { lens[n++] = (uint8_t)sym; continue; }
            unsigned rep, val = 0;
            if (sym == 16) {
                if (n == 0 || !need(2)) return false;
                val = lens[n - 1]; rep = 3 + take(2);
            } else if (sym == 17) {
                if (!need(3)) return false;
                rep = 3 + take(3);
            } else {
                if (!need(7)) return false;
                rep = 11 + take(7);
            }
            if (n + rep > total) return false;
            while (rep--) lens[n++] = (uint8_t)val;
        }
        if (lens[256] == 0) return false;                     // no end-of-block code
        // incomplete only in the one shape zlib accepts too: a single 1-bit code (a block holding nothing but its end-of-block)
        if (!build_table(lens, hlit, LIT_BITS, lit_, LIT_CAP, true, litlen_entry)) return false;
        if (!build_table(lens + hlit, hdist, DIST_BITS, dist_, DIST_CAP, true, dist_entry)) return false;
        build_literal_runs();
        build_pairs();
        return true;
    }

    // ---- the block loop ----------------------------------------------------------------------------------------
    // 1 = end of block, 0 = out of room, -1 = corrupt
    // (Two versions of the loop, picked when the program starts: with BMI2 the shifts by a table entry's bit counts are SHRX / BZHI --
    // no detour through CL, no flags -- and the loop, a chain of dependent look-ups and shifts, is 14 % faster on reads.)
    // (not under the sanitizers: the resolver of an ifunc runs before their run-time is up)
    // (glibc only: target_clones needs the dynamic loader's ifunc support -- not musl, not a static link)
#if defined(__x86_64__) && defined(__GNUC__) && defined(__GLIBC__) && !defined(__clang__) && !defined(MDBG_HOST_NO_TARGET_CLONES) && \
    !defined(__SANITIZE_THREAD__) && !defined(__SANITIZE_ADDRESS__)
    __attribute__((target_clones("bmi2", "default")))
#endif
    int decode_block(OutT *&out_ref, OutT *out_end, const OutT *lowest) {
        OutT *out = out_ref;
        const uint8_t *in = in_;
        uint64_t bitbuf = bitbuf_;
        int bitcnt = bitcnt_;
        int result = 0;
        const uint32_t *const lit = lit_, *const dist = dist_;
        // fast loop: at least 16 readable input bytes and MIN_ROOM elements of room (16 literals + one maximal match + copy overshoot)
        while (in_end_ - in >= 16 && out_end - out >= MIN_ROOM) {
            // refill to >= 56 bits
            bitbuf |= load64(in) << bitcnt;
            in += (63 - bitcnt) >> 3;
            bitcnt |= 56;
            // literal runs: one look-up yields up to four literals whose codes fit the 11 index bits together (DNA text has
            // 2- to 3-bit codes, quality strings 4 to 7); four look-ups per refill (4 x 11 <= 56 bits)
            {
                unsigned info = run_info_[bitbuf & ((1u << LIT_BITS) - 1)];
                if (info) {
                    int k = 0;
                    do {
                        store_run(out, run_lits_[bitbuf & ((1u << LIT_BITS) - 1)]);
                        out += info >> 4;
                        bitbuf >>= (info & 15); bitcnt -= (int)(info & 15);
                        info = run_info_[bitbuf & ((1u << LIT_BITS) - 1)];
                    } while (info && ++k < 4);
                    if (info) continue;                       // still literals: refill and go on
                    // a non-literal follows and up to 44 bits are gone: refill for length + distance (48 bits)
                    bitbuf |= load64(in) << bitcnt;
                    in += (63 - bitcnt) >> 3;
                    bitcnt |= 56;
                }
            }
            unsigned len, off;
            const uint32_t pe = pair_[bitbuf & ((1u << LIT_BITS) - 1)];
            if (__builtin_expect(pe != 0, 1)) {
                const unsigned tb = pe & 15, dx = (pe >> 4) & 15;
                len = (pe >> 8) & 511;
                off = (pe >> 17) + (unsigned)((bitbuf >> tb) & ((1u << dx) - 1));
                bitbuf >>= (tb + dx); bitcnt -= (int)(tb + dx);
            } else {
                uint32_t e = lit[bitbuf & ((1u << LIT_BITS) - 1)];
                if (e & F_SUB) {
                    e = lit[(e >> 16) + ((bitbuf >> LIT_BITS) & ((1u << ((e >> 8) & 15)) - 1))];
                    if (e & F_LITERAL) {
                        *out++ = (OutT)(uint8_t)(e >> 16);
                        bitbuf >>= (e & 0xFF); bitcnt -= (int)(e & 0xFF);
                        continue;
                    }
                }
                if (e & (F_EOB | F_INVALID)) {
                    if (e & F_INVALID) { result = -1; break; }
                    bitbuf >>= (e & 0xFF); bitcnt -= (int)(e & 0xFF);
                    result = 1;
                    break;
                }
                // length: code (<= 15) + extra (<= 5); then distance: code (<= 15) + extra (<= 13): 48 <= 56 bits
                bitbuf >>= (e & 0xFF); bitcnt -= (int)(e & 0xFF);
                len = (e >> 16) + (unsigned)(bitbuf & ((1u << ((e >> 8) & 15)) - 1));
                bitbuf >>= ((e >> 8) & 15); bitcnt -= (int)((e >> 8) & 15);
                uint32_t d = dist[bitbuf & ((1u << DIST_BITS) - 1)];
                if (d & F_SUB) d = dist[(d >> 16) + ((bitbuf >> DIST_BITS) & ((1u << ((d >> 8) & 15)) - 1))];
                if (d & F_INVALID) { result = -1; break; }
                bitbuf >>= (d & 0xFF); bitcnt -= (int)(d & 0xFF);
                off = (d >> 16) + (unsigned)(bitbuf & ((1u << ((d >> 8) & 15)) - 1));
                bitbuf >>= ((d >> 8) & 15); bitcnt -= (int)((d >> 8) & 15);
            }
            if ((size_t)(out - lowest) < off) { result = -1; break; }
            const OutT *src = out - off;
            OutT *const end = out + len;
            constexpr unsigned STEP = 8 / sizeof(OutT);        // elements per 8-byte copy
            if (__builtin_expect(off >= STEP, 1)) {
                // read text has short matches (3 to 8 bytes): the first 8 ELEMENTS without a loop and without a test -- one 8-byte
                // copy for bytes, two in a row for 16-bit symbols (4 elements each; the second may read what the first wrote, which
                // is what a match with 4 <= off < 8 means).  "Is it longer than 4?" would be a coin flip on 3- to 5-base matches.
                memcpy(out, src, 8);
                if (sizeof(OutT) == 2) memcpy(out + STEP, src + STEP, 8);
                if (__builtin_expect(len > 8, 0)) {
                    out += 8; src += 8;
                    do { memcpy(out, src, 8); out += STEP; src += STEP; } while (out < end);
                }
            } else if (off == 1) {
                if (sizeof(OutT) == 1) memset(out, (int)*src, len);
                else { const OutT v = *src; for (unsigned i = 0; i < len; i++) out[i] = v; }
            } else {
                do { *out++ = *src++; } while (out < end);
            }
            out = end;
        }
        if (result == 0) {
            // careful loop: near the end of the input or of the room; one symbol at a time, bits fetched on demand
            in_ = in; bitbuf_ = bitbuf; bitcnt_ = bitcnt;
            result = decode_block_careful(out, out_end, lowest);
            out_ref = out;
            return result;
        }
        in_ = in; bitbuf_ = bitbuf; bitcnt_ = bitcnt;
        out_ref = out;
        return result;
    }

    // up to four literals at once (the unused ones are overwritten by what follows)
    static void store_run(uint8_t *out, uint32_t lits) { memcpy(out, &lits, 4); }
    static void store_run(uint16_t *out, uint32_t lits) {
        uint64_t w = lits;
        w = (w | (w << 16)) & 0x0000FFFF0000FFFFull;
        w = (w | (w << 8)) & 0x00FF00FF00FF00FFull;
        memcpy(out, &w, 8);
    }

    // fetch bytes while available (up to 56 bits); decoding then checks that the bits it consumes exist
    void soft_refill() {
        while (bitcnt_ < 56 && in_ < in_end_) { bitbuf_ |= (uint64_t)*in_++ << bitcnt_; bitcnt_ += 8; }
    }
    bool drop(unsigned n) {
        if ((int)n > bitcnt_) return false;
        bitbuf_ >>= n; bitcnt_ -= (int)n;
        return true;
    }

    int decode_block_careful(OutT *&out, OutT *out_end, const OutT *lowest) {
        for (;;) {
            if (out_end - out < MIN_ROOM) {
                // out of room -- unless the input still allows the fast loop next time, which the caller decides
                return 0;
            }
            if (in_end_ - in_ >= 16 + 8) {
                // enough input again for the fast loop (we only came here for lack of room, which has been checked above)
                return decode_block(out, out_end, lowest);
            }
            soft_refill();
            uint32_t e = lit_[bitbuf_ & ((1u << LIT_BITS) - 1)];
            if (e & F_SUB) e = lit_[(e >> 16) + ((bitbuf_ >> LIT_BITS) & ((1u << ((e >> 8) & 15)) - 1))];
            if (e & F_INVALID) return -1;
            if (!drop(e & 0xFF)) return -1;
            if (e & F_LITERAL) { *out++ = (OutT)(uint8_t)(e >> 16); continue; }
            if (e & F_EOB) return 1;
            const unsigned lx = (e >> 8) & 15;
            soft_refill();
            unsigned len = (e >> 16) + (unsigned)(bitbuf_ & ((1u << lx) - 1));
            if (!drop(lx)) return -1;
            soft_refill();
            uint32_t d = dist_[bitbuf_ & ((1u << DIST_BITS) - 1)];
            if (d & F_SUB) d = dist_[(d >> 16) + ((bitbuf_ >> DIST_BITS) & ((1u << ((d >> 8) & 15)) - 1))];
            if (d & F_INVALID) return -1;
            if (!drop(d & 0xFF)) return -1;
            const unsigned dx = (d >> 8) & 15;
            soft_refill();
            const unsigned off = (d >> 16) + (unsigned)(bitbuf_ & ((1u << dx) - 1));
            if (!drop(dx)) return -1;
            if ((size_t)(out - lowest) < off) return -1;
            const OutT *src = out - off;
            while (len--) *out++ = *src++;
        }
    }

    static constexpr unsigned LIT_CAP = (1u << LIT_BITS) + 288 * 16, DIST_CAP = (1u << DIST_BITS) + 32 * 128;
    const uint8_t *base_ = nullptr, *in_ = nullptr, *in_end_ = nullptr;
    uint64_t stop_bit_ = ~0ull;
    unsigned blocks_done_ = 0;
    uint64_t bitbuf_ = 0;
    int bitcnt_ = 0;
    State state_ = BLOCK_HEADER;
    bool final_ = false;
    uint32_t stored_left_ = 0;
    uint32_t lit_[LIT_CAP];
    uint32_t dist_[DIST_CAP];
    uint32_t run_lits_[1u << LIT_BITS];    // up to four literals per index, little-endian
    uint8_t run_info_[1u << LIT_BITS];
    uint32_t pair_[1u << LIT_BITS];        // length code + extra + distance code inside the 11 index bits: 0 = no such pair here
};

using Inflater = InflaterT<uint8_t>;

// gzip member header (RFC 1952) at p: returns the offset of the deflate data, 0 if this is not a gzip member
inline size_t gzip_header_size(const uint8_t *p, size_t n) {
    if (n < 18 || p[0] != 0x1f || p[1] != 0x8b || p[2] != 8 || (p[3] & 0xE0)) return 0;
    const unsigned flg = p[3];
    size_t o = 10;
    if (flg & 4) {                                             // FEXTRA
        if (o + 2 > n) return 0;
        o += 2 + (p[o] | ((size_t)p[o + 1] << 8));
    }
    for (unsigned bit : {8u, 16u}) {                           // FNAME, FCOMMENT: zero-terminated
        if (!(flg & bit)) continue;
        while (o < n && p[o]) o++;
        o++;
    }
    if (flg & 2) o += 2;                                       // FHCRC
    return o + 8 <= n ? o : 0;
}

}  // namespace mdbg_host
