// reads.hip -- base-space read batches in HBM: ASCII -> 2-bit packing on the device, host-packed
// upload, and the seeded synthetic generator (twin of metamdbg_amd/synth.py) that builds
// benchmark inputs directly in HBM.
#include "common.hpp"
#include "objects.hpp"

#include <vector>

namespace mdbg {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    uint64_t z = x + 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// one wave per read, lanes stride over its words; each lane packs 32 ASCII bases into a u64.  Besides the 2-bit code and the
// invalid bit of every character (utils/kmer/Kmer.hpp:462) a third bit records where two neighbouring characters differ
// although code and invalid bit agree (mixed case, IUPAC letters): EncoderRLE compares characters (Commons.hpp:4177-4178).
__global__ __launch_bounds__(256) void pack_ascii_kernel(const uint8_t *bases, const uint64_t *base_off,
                                                         const uint64_t *word_off, uint32_t n_reads,
                                                         uint64_t *words, uint32_t *invalid, uint32_t *brk, uint32_t *any_flags, uint8_t *masked) {
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t seen = 0, seen_brk = 0;
    for (uint64_t r = wave; r < n_reads; r += nwaves) {
        const uint8_t *src = bases + base_off[r];
        const uint64_t L = base_off[r + 1] - base_off[r];
        const uint64_t w0 = word_off[r], nw = word_off[r + 1] - w0;
        uint32_t mine = 0;
        for (uint64_t w = lane; w < nw; w += 64) {
            uint64_t x = 0;
            uint32_t inv = 0, bk = 0;
            uint64_t b0 = w * 32;
            uint8_t prev = (b0 > 0 && b0 <= L) ? src[b0 - 1] : 0;
            bool have_prev = b0 > 0 && b0 <= L;
#pragma unroll 8
            for (int i = 0; i < 32; i++) {
                uint64_t bi = b0 + i;
                if (bi < L) {
                    uint8_t c = src[bi];
                    x |= (uint64_t)((c >> 1) & 3u) << (2 * i);    // utils/kmer/Kmer.hpp:462
                    inv |= (uint32_t)((c >> 3) & 1u) << i;
                    if (have_prev && c != prev && ((c ^ prev) & 0x0Eu) == 0) bk |= 1u << i;   // same code and invalid bit, other character
                    prev = c; have_prev = true;
                }
            }
            words[w0 + w] = x;
            invalid[w0 + w] = inv;
            brk[w0 + w] = bk;
            seen |= inv;
            seen_brk |= bk;
            mine |= inv | bk;
        }
        const bool any = __ballot(mine != 0u) != 0ull;      // the read carries a side-mask bit somewhere
        if (lane == 0) masked[r] = any ? 1 : 0;
    }
    if (seen) atomicOr(any_flags, 1u);
    if (seen_brk) atomicOr(any_flags, 2u);
}

// The side masks of LISTED reads of a batch uploaded as 2-bit words, from their characters: what pack_ascii_kernel derives for every
// read of an ASCII batch, for the few reads of a packed batch that hold something else than upper-case ACGT.  One wave per listed read.
__global__ __launch_bounds__(256) void mask_listed_kernel(const uint8_t *ascii, const uint64_t *a_off, const uint32_t *list, uint32_t n_list,
                                                          const uint64_t *word_off, uint32_t *invalid, uint32_t *brk, uint32_t *any_flags,
                                                          uint8_t *masked, uint8_t *listed_masked) {
    const unsigned lane = threadIdx.x & 63u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    uint32_t seen = 0, seen_brk = 0;
    for (uint64_t i = wave; i < n_list; i += nwaves) {
        const uint32_t r = list[i];
        const uint8_t *src = ascii + a_off[i];
        const uint64_t L = a_off[i + 1] - a_off[i];
        const uint64_t w0 = word_off[r], nw = word_off[r + 1] - w0;
        uint32_t mine = 0;
        for (uint64_t w = lane; w < nw; w += 64) {
            uint32_t inv = 0, bk = 0;
            const uint64_t b0 = w * 32;
            uint8_t prev = (b0 > 0 && b0 <= L) ? src[b0 - 1] : 0;
            bool have_prev = b0 > 0 && b0 <= L;
            for (int j = 0; j < 32; j++) {
                const uint64_t bi = b0 + j;
                if (bi < L) {
                    const uint8_t c = src[bi];
                    inv |= (uint32_t)((c >> 3) & 1u) << j;
                    if (have_prev && c != prev && ((c ^ prev) & 0x0Eu) == 0) bk |= 1u << j;
                    prev = c; have_prev = true;
                }
            }
            invalid[w0 + w] = inv;
            brk[w0 + w] = bk;
            seen |= inv; seen_brk |= bk; mine |= inv | bk;
        }
        const bool any = __ballot(mine != 0u) != 0ull;
        if (lane == 0) { masked[r] = any ? 1 : 0; listed_masked[i] = any ? 1 : 0; }
    }
    if (seen) atomicOr(any_flags, 1u);
    if (seen_brk) atomicOr(any_flags, 2u);
}

struct SynthArgs {
    uint64_t seed;
    uint32_t n_reads, read_len;
    uint64_t first_read;
    const uint64_t *species_len;      // n_species
    const uint64_t *species_off;      // n_species + 1
    const uint64_t *species_thr;      // n_species
    uint32_t n_species;
    uint64_t sub_thr;
    uint64_t ins_thr, del_thr;        // indel model (synth.read_codes): both 0 for substitution-only reads
    uint32_t window;                  // genome bases set aside per read (= read_len without indels)
    uint32_t words_per_read;
    uint64_t *words;
    uint8_t *qual;                    // nullptr or n_reads * read_len
};

// (grid-stride: a launch of 2^32 threads or more wraps around -- 40 M reads x 314 words ran as the first 12.64 M reads and left the rest
// zero, which the N = 1 point of a 40 M-read strong-scaling run showed as 11.8 minimizers per read)
__global__ __launch_bounds__(256) void synth_kernel(SynthArgs a) {
    const uint64_t total = (uint64_t)a.n_reads * a.words_per_read;
    const uint64_t base = mix64(a.seed ^ 0xA5A5A5A5A5A5A5A5ull);
    const uint64_t gkey = mix64(a.seed);
    for (uint64_t gid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (uint64_t)gridDim.x * blockDim.x) {
        const uint32_t rl = (uint32_t)(gid / a.words_per_read);
        const uint32_t w = (uint32_t)(gid % a.words_per_read);
        const uint64_t r = a.first_read + rl;
        const uint32_t L = a.read_len;
        // layout (synth.read_layout)
        const uint64_t u_species = mix64(base + 4 * r), u_start = mix64(base + 4 * r + 1), u_strand = mix64(base + 4 * r + 2);
        uint32_t sp = 0;
        while (sp + 1 < a.n_species && !(u_species < a.species_thr[sp])) sp++;
        const uint64_t span = a.species_len[sp] - a.window + 1;
        const uint64_t start = a.species_off[sp] + (u_start % span);
        const unsigned strand = (unsigned)(u_strand & 1ull);
        const uint64_t ekey = mix64(mix64(a.seed ^ 0x5EED5EED5EED5EEDull) + r);
        const uint64_t qkey = mix64(mix64(a.seed ^ 0x0123456789ABCDEFull) + r);
        uint64_t x = 0;
        for (int i = 0; i < 32; i++) {
            uint32_t bi = w * 32 + i;
            if (bi >= L) break;
            uint64_t gpos = strand ? (start + (L - 1) - bi) : (start + bi);
            unsigned code = (unsigned)(mix64(gkey + gpos) >> 62);
            if (strand) code ^= 2u;
            uint64_t e = mix64(ekey + bi);
            if (e < a.sub_thr) code = (code + 1u + (unsigned)(mix64(e) % 3ull)) & 3u;
            x |= (uint64_t)code << (2 * i);
            if (a.qual) a.qual[(uint64_t)rl * L + bi] = (uint8_t)(mix64(qkey + bi) % 30ull + 43ull);
        }
        a.words[gid] = x;
    }
}

// Reads with insertions and deletions (synth.read_codes): read position i shows genome base k(i) = i - #insertions before i
// + #deletions up to i of the read's window, so a word needs the event counts of everything before it.  One wave per
// read: pass 1 counts (deletions - insertions) per 32-position word into LDS, a wave scan turns them into the offset at
// the start of every word, pass 2 generates the words.
constexpr int SYNTH_MAX_WORDS = 2048;            // read_len <= 65536 for reads with indels
__global__ __launch_bounds__(256) void synth_indel_kernel(SynthArgs a) {
    __shared__ int32_t lds_delta[4][SYNTH_MAX_WORDS];
    const unsigned lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    int32_t *delta = lds_delta[wv];
    const uint32_t L = a.read_len, W = a.window;
    const uint32_t nw = (L + 31u) / 32u;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    const uint64_t base = mix64(a.seed ^ 0xA5A5A5A5A5A5A5A5ull);
    const uint64_t gkey = mix64(a.seed);
    const uint64_t t_id = a.ins_thr + a.del_thr;
    const uint64_t t_all = (t_id + a.sub_thr < t_id) ? ~0ull : t_id + a.sub_thr;
    for (uint64_t rl = wave; rl < a.n_reads; rl += nwaves) {
        const uint64_t r = a.first_read + rl;
        const uint64_t u_species = mix64(base + 4 * r), u_start = mix64(base + 4 * r + 1), u_strand = mix64(base + 4 * r + 2);
        uint32_t sp = 0;
        while (sp + 1 < a.n_species && !(u_species < a.species_thr[sp])) sp++;
        const uint64_t span = a.species_len[sp] - W + 1;
        const uint64_t start = a.species_off[sp] + (u_start % span);
        const unsigned strand = (unsigned)(u_strand & 1ull);
        const uint64_t ekey = mix64(mix64(a.seed ^ 0x5EED5EED5EED5EEDull) + r);
        const uint64_t qkey = mix64(mix64(a.seed ^ 0x0123456789ABCDEFull) + r);
        // pass 1: net offset change inside every word
        for (uint32_t w = lane; w < nw; w += 64) {
            int32_t d = 0;
            for (uint32_t i = 0; i < 32; i++) {
                const uint32_t bi = w * 32 + i;
                if (bi >= L) break;
                const uint64_t e = mix64(ekey + bi);
                if (e < a.ins_thr) d--; else if (e < t_id) d++;
            }
            delta[w] = d;
        }
        wave_lds_sync();
        // exclusive scan over the words (nw <= 2048: each lane sums a contiguous run, then a wave scan of the run totals)
        const uint32_t per = (nw + 63u) / 64u;
        int32_t run = 0;
        for (uint32_t j = 0; j < per; j++) { const uint32_t w = lane * per + j; if (w < nw) run += delta[w]; }
        int32_t incl = run;
        for (int dlt = 1; dlt < 64; dlt <<= 1) { int32_t t = __shfl_up(incl, dlt, 64); if (lane >= (unsigned)dlt) incl += t; }
        int32_t acc = incl - run;
        wave_lds_sync();
        for (uint32_t j = 0; j < per; j++) { const uint32_t w = lane * per + j; if (w < nw) { int32_t d = delta[w]; delta[w] = acc; acc += d; } }
        wave_lds_sync();
        // pass 2
        for (uint32_t w = lane; w < a.words_per_read; w += 64) {
            uint64_t x = 0;
            if (w < nw) {
                int64_t off = delta[w];           // (#deletions - #insertions) before position 32 w
                for (uint32_t i = 0; i < 32; i++) {
                    const uint32_t bi = w * 32 + i;
                    if (bi >= L) break;
                    const uint64_t e = mix64(ekey + bi);
                    unsigned code;
                    if (e < a.ins_thr) {
                        code = (unsigned)(mix64(e ^ 0x1B5E47EDull) >> 62);
                        off--;                    // this position consumed no genome base
                    } else {
                        if (e < t_id) off++;
                        int64_t k = (int64_t)bi + off;
                        if (k > (int64_t)W - 1) k = (int64_t)W - 1;
                        const uint64_t gpos = strand ? (start + (W - 1) - (uint64_t)k) : (start + (uint64_t)k);
                        code = (unsigned)(mix64(gkey + gpos) >> 62);
                        if (strand) code ^= 2u;
                        if (e >= t_id && e < t_all) code = (code + 1u + (unsigned)(mix64(e) % 3ull)) & 3u;
                    }
                    x |= (uint64_t)code << (2 * i);
                    if (a.qual) a.qual[rl * L + bi] = (uint8_t)(mix64(qkey + bi) % 30ull + 43ull);
                }
            }
            a.words[rl * a.words_per_read + w] = x;
        }
        wave_lds_sync();
    }
}

__global__ void fill_u32_kernel(uint32_t *p, uint64_t n, uint32_t v) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ void iota_scaled_u64_kernel(uint64_t *p, uint64_t n, uint64_t scale) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = i * scale;
}

}  // namespace mdbg

using namespace mdbg;

static inline uint64_t words_for(uint64_t len) { return ((len + 63) / 64) * 2; }  // reads start on 16-byte units

extern "C" int mdbg_reads_from_ascii(mdbg_ctx *ctx, const char *bases, const char *quals, const uint64_t *offsets,
                                     uint32_t n_reads, mdbg_reads **out) try {
    if (!ctx || !out || (n_reads && (!bases || !offsets))) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_from_ascii: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    mdbg_reads *r = new mdbg_reads();
    auto fail = [&](int rc) { delete r; return rc; };
    r->n_reads = n_reads;
    std::vector<uint64_t> woff((size_t)n_reads + 1, 0);
    std::vector<uint32_t> lens(n_reads);
    uint64_t nb = n_reads ? offsets[n_reads] - offsets[0] : 0;
    for (uint32_t i = 0; i < n_reads; i++) {
        uint64_t L = offsets[i + 1] - offsets[i];
        if (L > 0xFFFFFFF0ull) return fail(set_error(ctx, MDBG_ERANGE, "read %u longer than 2^32 bases", i));
        lens[i] = (uint32_t)L;
        if (lens[i] > r->max_len) r->max_len = lens[i];
        woff[i + 1] = woff[i] + words_for(L);
    }
    r->n_bases = nb;
    r->n_words = woff[n_reads];
    int rc;
    DevBuf<uint8_t> d_ascii;
    DevBuf<uint64_t> d_boff;
    DevBuf<uint32_t> d_any;
    if ((rc = r->d_words.alloc(ctx, r->n_words)) || (rc = r->d_invalid.alloc(ctx, r->n_words)) || (rc = r->d_break.alloc(ctx, r->n_words)) ||
        (rc = r->d_word_off.alloc(ctx, (size_t)n_reads + 1)) || (rc = r->d_len.alloc(ctx, n_reads)) ||
        (rc = d_ascii.alloc(ctx, nb)) || (rc = d_boff.alloc(ctx, (size_t)n_reads + 1)) || (rc = d_any.alloc(ctx, 1)) ||
        (rc = r->d_masked.alloc(ctx, n_reads)))
        return fail(rc);
    std::vector<uint64_t> rel((size_t)n_reads + 1, 0);
    for (uint32_t i = 0; i <= n_reads && n_reads; i++) rel[i] = offsets[i] - offsets[0];
    hipError_t e;
#define CK(x) if ((e = (x)) != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "%s: %s", #x, hipGetErrorString(e)))
    CK(hipMemcpyAsync(r->d_word_off.p, woff.data(), woff.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    CK(hipMemcpyAsync(d_boff.p, rel.data(), rel.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    if (n_reads) CK(hipMemcpyAsync(r->d_len.p, lens.data(), lens.size() * 4, hipMemcpyHostToDevice, ctx->stream));
    if (nb) CK(hipMemcpyAsync(d_ascii.p, bases + offsets[0], nb, hipMemcpyHostToDevice, ctx->stream));
    CK(hipMemsetAsync(d_any.p, 0, 4, ctx->stream));
    if (n_reads) {
        unsigned blocks = grid_for((uint64_t)n_reads * 64, 256, (unsigned)ctx->n_cu * 16u);
        LaunchTimer timer(ctx, "pack_ascii");
        hipLaunchKernelGGL(pack_ascii_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_ascii.p, d_boff.p,
                           r->d_word_off.p, n_reads, r->d_words.p, r->d_invalid.p, r->d_break.p, d_any.p, r->d_masked.p);
    }
    uint32_t any = 0;
    CK(hipMemcpyAsync(&any, d_any.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    if (quals && nb) {
        if ((rc = r->d_qual.alloc(ctx, nb + 32)) || (rc = r->d_qual_off.alloc(ctx, (size_t)n_reads + 1))) return fail(rc);   // + 32: kernels read whole 16-byte pieces
        CK(hipMemcpyAsync(r->d_qual.p, quals + offsets[0], nb, hipMemcpyHostToDevice, ctx->stream));
        CK(hipMemcpyAsync(r->d_qual_off.p, rel.data(), rel.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        r->has_qual = true;
    }
    CK(hipStreamSynchronize(ctx->stream));
#undef CK
    r->has_break = (any & 2u) != 0;
    r->has_invalid = any != 0;            // the scan's side-mask variant reads d_invalid whenever either mask matters
    if (!r->has_invalid) { r->d_invalid.release(); r->d_masked.release(); }
    if (!r->has_break) r->d_break.release();
    if (r->has_invalid) {                 // which reads: the list mdbg_scan hands to the general kernel
        std::vector<uint8_t> flag(n_reads);
        if ((e = memcpy_sync(ctx, flag.data(), r->d_masked.p, n_reads, hipMemcpyDeviceToHost)) != hipSuccess)
            return fail(set_error(ctx, MDBG_EHIP, "masked flags copy failed: %s", hipGetErrorString(e)));
        std::vector<uint32_t> list;
        for (uint32_t i = 0; i < n_reads; i++) if (flag[i]) list.push_back(i);
        r->n_masked = (uint32_t)list.size();
        if ((rc = r->d_masked_list.alloc(ctx, list.size()))) return fail(rc);
        if (!list.empty() && (e = memcpy_sync(ctx, r->d_masked_list.p, list.data(), list.size() * 4, hipMemcpyHostToDevice)) != hipSuccess)
            return fail(set_error(ctx, MDBG_EHIP, "masked list upload failed: %s", hipGetErrorString(e)));
    }
    *out = r;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_reads_from_packed(mdbg_ctx *ctx, const uint64_t *words, const uint64_t *word_offsets,
                                      const uint32_t *lengths, uint32_t n_reads, mdbg_reads **out) try {
    if (!ctx || !out || (n_reads && (!words || !word_offsets || !lengths)))
        return set_error(ctx, MDBG_EINVAL, "mdbg_reads_from_packed: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    mdbg_reads *r = new mdbg_reads();
    auto fail = [&](int rc) { delete r; return rc; };
    r->n_reads = n_reads;
    uint64_t base = n_reads ? word_offsets[0] : 0;
    r->n_words = n_reads ? word_offsets[n_reads] - base : 0;
    std::vector<uint64_t> rel((size_t)n_reads + 1, 0);
    for (uint32_t i = 0; i < n_reads; i++) {
        rel[i] = word_offsets[i] - base;
        uint64_t nw = word_offsets[i + 1] - word_offsets[i];
        if ((rel[i] & 1) || nw * 32 < lengths[i])
            return fail(set_error(ctx, MDBG_EINVAL, "read %u: word offset must be even and cover the read", i));
        r->n_bases += lengths[i];
        if (lengths[i] > r->max_len) r->max_len = lengths[i];
    }
    rel[n_reads] = r->n_words;
    int rc;
    if ((rc = r->d_words.alloc(ctx, r->n_words + 2)) || (rc = r->d_word_off.alloc(ctx, (size_t)n_reads + 1)) ||
        (rc = r->d_len.alloc(ctx, n_reads)))
        return fail(rc);
    hipError_t e;
#define CK(x) if ((e = (x)) != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "%s: %s", #x, hipGetErrorString(e)))
    if (r->n_words) CK(hipMemcpyAsync(r->d_words.p, words + base, r->n_words * 8, hipMemcpyHostToDevice, ctx->stream));
    CK(hipMemcpyAsync(r->d_word_off.p, rel.data(), rel.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    if (n_reads) CK(hipMemcpyAsync(r->d_len.p, lengths, (size_t)n_reads * 4, hipMemcpyHostToDevice, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
#undef CK
    *out = r;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_reads_from_packed_async(mdbg_ctx *ctx, const uint64_t *words, const uint64_t *word_offsets,
                                            const uint32_t *lengths, uint32_t n_reads, mdbg_reads **out) try {
    if (!ctx || !out || (n_reads && (!words || !word_offsets || !lengths)))
        return set_error(ctx, MDBG_EINVAL, "mdbg_reads_from_packed_async: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (!ctx->upload_stream) MDBG_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
    std::unique_ptr<mdbg_reads> r(new mdbg_reads());
    r->n_reads = n_reads;
    const uint64_t base = n_reads ? word_offsets[0] : 0;
    r->n_words = n_reads ? word_offsets[n_reads] - base : 0;
    r->h_rel.assign((size_t)n_reads + 1, 0);
    for (uint32_t i = 0; i < n_reads; i++) {
        r->h_rel[i] = word_offsets[i] - base;
        const uint64_t nw = word_offsets[i + 1] - word_offsets[i];
        if ((r->h_rel[i] & 1) || nw * 32 < lengths[i])
            return set_error(ctx, MDBG_EINVAL, "read %u: word offset must be even and cover the read", i);
        r->n_bases += lengths[i];
        if (lengths[i] > r->max_len) r->max_len = lengths[i];
    }
    r->h_rel[n_reads] = r->n_words;
    r->h_len.assign(lengths, lengths + n_reads);
    MDBG_TRY(r->d_words.alloc(ctx, r->n_words + 2));
    MDBG_TRY(r->d_word_off.alloc(ctx, (size_t)n_reads + 1));
    MDBG_TRY(r->d_len.alloc(ctx, n_reads));
    // the blocks come from the context's pool, whose reuse is ordered by the context's stream: the upload starts after what that
    // stream has queued so far (the last user of a recycled block), not after what it queues later
    hipEvent_t ev = nullptr;
    MDBG_HIP_CHECK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->upload_stream, ev, 0);
    // the two small arrays first: they usually sit in pageable memory, and a copy from pageable memory returns only when it is
    // done -- queued behind the words it made this call wait for the whole upload (round 3: the pipelined PCIe leg overlapped nothing)
    if (e == hipSuccess) e = hipMemcpyAsync(r->d_word_off.p, r->h_rel.data(), r->h_rel.size() * 8, hipMemcpyHostToDevice, ctx->upload_stream);
    if (e == hipSuccess && n_reads) e = hipMemcpyAsync(r->d_len.p, lengths, (size_t)n_reads * 4, hipMemcpyHostToDevice, ctx->upload_stream);
    if (e == hipSuccess && r->n_words) e = hipMemcpyAsync(r->d_words.p, words + base, r->n_words * 8, hipMemcpyHostToDevice, ctx->upload_stream);
    if (e == hipSuccess) e = hipEventRecord(ev, ctx->upload_stream);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(ctx->upload_stream);
        (void)hipEventDestroy(ev);
        return set_error(ctx, MDBG_EHIP, "mdbg_reads_from_packed_async: %s", hipGetErrorString(e));
    }
    r->ready = ev;
    *out = r.release();
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_reads_attach_qualities_async(mdbg_ctx *ctx, mdbg_reads *r, const char *quals, const uint64_t *offsets) try {
    if (!ctx || !r || (r->n_reads && (!quals || !offsets))) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_attach_qualities_async: null argument");
    if (r->has_qual) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_attach_qualities_async: the reads already carry qualities");
    if (!r->ready || r->h_len.size() != r->n_reads)
        return set_error(ctx, MDBG_EINVAL, "mdbg_reads_attach_qualities_async: for reads made by mdbg_reads_from_packed_async");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    const uint32_t n = r->n_reads;
    if (!n) return MDBG_OK;
    r->h_qrel.resize((size_t)n + 1);
    for (uint32_t i = 0; i <= n; i++) r->h_qrel[i] = offsets[i] - offsets[0];
    for (uint32_t i = 0; i < n; i++)
        if (r->h_qrel[i + 1] - r->h_qrel[i] != r->h_len[i])
            return set_error(ctx, MDBG_EINVAL, "read %u: %llu qualities for %u bases", i, (unsigned long long)(r->h_qrel[i + 1] - r->h_qrel[i]), r->h_len[i]);
    const uint64_t nb = r->h_qrel[n];
    MDBG_TRY(r->d_qual.alloc(ctx, nb + 32));
    MDBG_TRY(r->d_qual_off.alloc(ctx, (size_t)n + 1));
    // behind the words on the upload stream; `ready` moves behind the qualities.  (The blocks come from the pool: as in
    // mdbg_reads_from_packed_async the upload is ordered after what the context's stream has queued so far.)
    hipEvent_t ev = nullptr;
    MDBG_HIP_CHECK(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, ctx->stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(ctx->upload_stream, ev, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(r->d_qual_off.p, r->h_qrel.data(), r->h_qrel.size() * 8, hipMemcpyHostToDevice, ctx->upload_stream);
    if (e == hipSuccess && nb) e = hipMemcpyAsync(r->d_qual.p, quals + offsets[0], nb, hipMemcpyHostToDevice, ctx->upload_stream);
    if (e == hipSuccess) e = hipEventRecord(ev, ctx->upload_stream);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(ctx->upload_stream);
        (void)hipEventDestroy(ev);
        return set_error(ctx, MDBG_EHIP, "mdbg_reads_attach_qualities_async: %s", hipGetErrorString(e));
    }
    (void)hipEventDestroy(r->ready);        // superseded: the new event is recorded behind everything the old one covered
    r->ready = ev;
    r->has_qual = true;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_reads_mark_ascii(mdbg_ctx *ctx, mdbg_reads *r, const uint32_t *read_index, uint32_t n_listed, const char *ascii,
                                     const uint64_t *ascii_offsets) try {
    if (!ctx || !r || (n_listed && (!read_index || !ascii || !ascii_offsets))) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_mark_ascii: null argument");
    if (r->has_invalid) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_mark_ascii: the batch already carries side masks");
    if (!n_listed) return MDBG_OK;
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    MDBG_HIP_CHECK(ctx, reads_ready_on(ctx, r));          // an upload still in flight: the kernel below runs behind it
    const uint32_t n = r->n_reads;
    std::vector<uint32_t> lens = r->h_len;
    if (lens.size() != n) {
        lens.resize(n);
        MDBG_HIP_CHECK(ctx, reads_ready_host(r));
        MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, lens.data(), r->d_len.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    }
    std::vector<uint64_t> rel((size_t)n_listed + 1);
    for (uint32_t i = 0; i <= n_listed; i++) rel[i] = ascii_offsets[i] - ascii_offsets[0];
    for (uint32_t i = 0; i < n_listed; i++) {
        if (read_index[i] >= n || (i && read_index[i] <= read_index[i - 1]))
            return set_error(ctx, MDBG_EINVAL, "mdbg_reads_mark_ascii: read indices must be ascending and inside the batch");
        if (rel[i + 1] - rel[i] != lens[read_index[i]])
            return set_error(ctx, MDBG_EINVAL, "mdbg_reads_mark_ascii: read %u has %u bases, %llu characters given", read_index[i], lens[read_index[i]],
                             (unsigned long long)(rel[i + 1] - rel[i]));
    }
    DevBuf<uint8_t> d_ascii, d_lm;
    DevBuf<uint64_t> d_aoff;
    DevBuf<uint32_t> d_list, d_any;
    MDBG_TRY(r->d_invalid.alloc(ctx, r->n_words));
    MDBG_TRY(r->d_break.alloc(ctx, r->n_words));
    MDBG_TRY(r->d_masked.alloc(ctx, n));
    MDBG_TRY(d_ascii.alloc(ctx, rel[n_listed]));
    MDBG_TRY(d_aoff.alloc(ctx, rel.size()));
    MDBG_TRY(d_list.alloc(ctx, n_listed));
    MDBG_TRY(d_lm.alloc(ctx, n_listed));
    MDBG_TRY(d_any.alloc(ctx, 1));
    MDBG_HIP_CHECK(ctx, hipMemsetAsync(r->d_invalid.p, 0, r->n_words * 4, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipMemsetAsync(r->d_break.p, 0, r->n_words * 4, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipMemsetAsync(r->d_masked.p, 0, n, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipMemsetAsync(d_any.p, 0, 4, ctx->stream));
    if (rel[n_listed]) MDBG_HIP_CHECK(ctx, hipMemcpyAsync(d_ascii.p, ascii + ascii_offsets[0], rel[n_listed], hipMemcpyHostToDevice, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipMemcpyAsync(d_aoff.p, rel.data(), rel.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipMemcpyAsync(d_list.p, read_index, (size_t)n_listed * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(mask_listed_kernel, dim3(grid_for((uint64_t)n_listed * 64, 256, (unsigned)ctx->n_cu * 16u)), dim3(256), 0, ctx->stream, d_ascii.p,
                       d_aoff.p, d_list.p, n_listed, r->d_word_off.p, r->d_invalid.p, r->d_break.p, d_any.p, r->d_masked.p, d_lm.p);
    uint32_t any = 0;
    std::vector<uint8_t> lm(n_listed);
    MDBG_HIP_CHECK(ctx, hipMemcpyAsync(&any, d_any.p, 4, hipMemcpyDeviceToHost, ctx->stream));
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, lm.data(), d_lm.p, n_listed, hipMemcpyDeviceToHost));
    r->has_break = (any & 2u) != 0;
    r->has_invalid = any != 0;
    if (!r->has_invalid) { r->d_invalid.release(); r->d_masked.release(); }
    if (!r->has_break) r->d_break.release();
    if (r->has_invalid) {
        std::vector<uint32_t> list;
        for (uint32_t i = 0; i < n_listed; i++) if (lm[i]) list.push_back(read_index[i]);
        r->n_masked = (uint32_t)list.size();
        MDBG_TRY(r->d_masked_list.alloc(ctx, list.size()));
        if (!list.empty()) MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, r->d_masked_list.p, list.data(), list.size() * 4, hipMemcpyHostToDevice));
    }
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_reads_wait(mdbg_ctx *ctx, const mdbg_reads *r) try {
    if (!ctx || !r) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_wait: null argument");
    MDBG_HIP_CHECK(ctx, reads_ready_host(r));
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_reads_attach_qualities(mdbg_ctx *ctx, mdbg_reads *r, const char *quals, const uint64_t *offsets) try {
    if (!ctx || !r || (r->n_reads && (!quals || !offsets))) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_attach_qualities: null argument");
    if (r->has_qual) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_attach_qualities: the reads already carry qualities");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    MDBG_HIP_CHECK(ctx, reads_ready_host(r));
    const uint32_t n = r->n_reads;
    if (!n) return MDBG_OK;
    std::vector<uint32_t> lens(n);
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, lens.data(), r->d_len.p, (size_t)n * 4, hipMemcpyDeviceToHost));
    std::vector<uint64_t> rel((size_t)n + 1);
    for (uint32_t i = 0; i <= n; i++) rel[i] = offsets[i] - offsets[0];
    for (uint32_t i = 0; i < n; i++)
        if (rel[i + 1] - rel[i] != lens[i]) return set_error(ctx, MDBG_EINVAL, "read %u: %llu qualities for %u bases", i,
                                                            (unsigned long long)(rel[i + 1] - rel[i]), lens[i]);
    const uint64_t nb = rel[n];
    MDBG_TRY(r->d_qual.alloc(ctx, nb + 32));
    MDBG_TRY(r->d_qual_off.alloc(ctx, (size_t)n + 1));
    if (nb) MDBG_HIP_CHECK(ctx, hipMemcpyAsync(r->d_qual.p, quals + offsets[0], nb, hipMemcpyHostToDevice, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipMemcpyAsync(r->d_qual_off.p, rel.data(), rel.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    MDBG_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    r->has_qual = true;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_reads_synthetic(mdbg_ctx *ctx, uint64_t seed, uint32_t n_reads, uint32_t read_len,
                                    uint64_t first_read, const uint64_t *species_len,
                                    const uint64_t *species_threshold, uint32_t n_species,
                                    uint64_t sub_threshold, uint64_t ins_threshold, uint64_t del_threshold, uint32_t window,
                                    int with_quality, mdbg_reads **out) try {
    if (!ctx || !out || !species_len || !species_threshold || !n_species || !read_len)
        return set_error(ctx, MDBG_EINVAL, "mdbg_reads_synthetic: bad argument");
    const bool indels = ins_threshold != 0 || del_threshold != 0;
    if (!indels) window = read_len;
    if (window < read_len) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_synthetic: window %u shorter than a read", window);
    if (indels && (read_len + 31u) / 32u > (uint32_t)SYNTH_MAX_WORDS)
        return set_error(ctx, MDBG_ERANGE, "mdbg_reads_synthetic: reads with indels are limited to %d bases", SYNTH_MAX_WORDS * 32);
    if (ins_threshold > (1ull << 62) || del_threshold > (1ull << 62)) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_synthetic: indel rate above 1/4");
    for (uint32_t s = 0; s < n_species; s++)
        if (species_len[s] < window) return set_error(ctx, MDBG_EINVAL, "species %u shorter than a read's window", s);
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    mdbg_reads *r = new mdbg_reads();
    auto fail = [&](int rc) { delete r; return rc; };
    const uint32_t wpr = (uint32_t)words_for(read_len);
    r->n_reads = n_reads;
    r->n_bases = (uint64_t)n_reads * read_len;
    r->n_words = (uint64_t)n_reads * wpr;
    r->max_len = read_len;
    std::vector<uint64_t> soff((size_t)n_species + 1, 0);
    for (uint32_t s = 0; s < n_species; s++) soff[s + 1] = soff[s] + species_len[s];
    DevBuf<uint64_t> d_slen, d_soff, d_sthr;
    int rc;
    if ((rc = r->d_words.alloc(ctx, r->n_words + 2)) || (rc = r->d_word_off.alloc(ctx, (size_t)n_reads + 1)) ||
        (rc = r->d_len.alloc(ctx, n_reads)) || (rc = d_slen.alloc(ctx, n_species)) ||
        (rc = d_soff.alloc(ctx, (size_t)n_species + 1)) || (rc = d_sthr.alloc(ctx, n_species)))
        return fail(rc);
    if (with_quality) {
        if ((rc = r->d_qual.alloc(ctx, r->n_bases + 32)) || (rc = r->d_qual_off.alloc(ctx, (size_t)n_reads + 1))) return fail(rc);
        r->has_qual = true;
    }
    hipError_t e;
#define CK(x) if ((e = (x)) != hipSuccess) return fail(set_error(ctx, MDBG_EHIP, "%s: %s", #x, hipGetErrorString(e)))
    CK(hipMemcpyAsync(d_slen.p, species_len, (size_t)n_species * 8, hipMemcpyHostToDevice, ctx->stream));
    CK(hipMemcpyAsync(d_soff.p, soff.data(), soff.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    CK(hipMemcpyAsync(d_sthr.p, species_threshold, (size_t)n_species * 8, hipMemcpyHostToDevice, ctx->stream));
    SynthArgs a{seed, n_reads, read_len, first_read, d_slen.p, d_soff.p, d_sthr.p, n_species, sub_threshold, ins_threshold,
                del_threshold, window, wpr, r->d_words.p, with_quality ? r->d_qual.p : nullptr};
    uint64_t total = (uint64_t)n_reads * wpr;
    if (total) {
        if (indels) hipLaunchKernelGGL(synth_indel_kernel, dim3(grid_for((uint64_t)n_reads * 64, 256, (unsigned)ctx->n_cu * 16u)), dim3(256), 0, ctx->stream, a);
        else hipLaunchKernelGGL(synth_kernel, dim3(grid_for(total, 256, 1u << 22)), dim3(256), 0, ctx->stream, a);
        hipLaunchKernelGGL(fill_u32_kernel, dim3(grid_for(n_reads, 256)), dim3(256), 0, ctx->stream, r->d_len.p, (uint64_t)n_reads, read_len);
    }
    hipLaunchKernelGGL(iota_scaled_u64_kernel, dim3(grid_for((uint64_t)n_reads + 1, 256)), dim3(256), 0, ctx->stream,
                       r->d_word_off.p, (uint64_t)n_reads + 1, (uint64_t)wpr);
    if (with_quality)
        hipLaunchKernelGGL(iota_scaled_u64_kernel, dim3(grid_for((uint64_t)n_reads + 1, 256)), dim3(256), 0, ctx->stream,
                           r->d_qual_off.p, (uint64_t)n_reads + 1, (uint64_t)read_len);
    CK(hipGetLastError());
    CK(hipStreamSynchronize(ctx->stream));
#undef CK
    *out = r;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_reads_info(const mdbg_reads *r, uint32_t *n_reads, uint64_t *n_bases, uint64_t *n_words) {
    if (!r) return MDBG_EINVAL;
    if (n_reads) *n_reads = r->n_reads;
    if (n_bases) *n_bases = r->n_bases;
    if (n_words) *n_words = r->n_words;
    return MDBG_OK;
}

extern "C" int mdbg_reads_get(mdbg_ctx *ctx, const mdbg_reads *r, uint32_t index, char *bases, char *quals, uint32_t *length) try {
    if (!ctx || !r || index >= r->n_reads) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_get: bad argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    MDBG_HIP_CHECK(ctx, reads_ready_host(r));
    uint64_t off[2];
    uint32_t L;
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, off, r->d_word_off.p + index, 16, hipMemcpyDeviceToHost));
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &L, r->d_len.p + index, 4, hipMemcpyDeviceToHost));
    if (length) *length = L;
    if (bases) {
        std::vector<uint64_t> w(off[1] - off[0]);
        std::vector<uint32_t> inv;
        if (!w.empty()) MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, w.data(), r->d_words.p + off[0], w.size() * 8, hipMemcpyDeviceToHost));
        if (r->has_invalid) {
            inv.resize(w.size());
            if (!w.empty()) MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, inv.data(), r->d_invalid.p + off[0], inv.size() * 4, hipMemcpyDeviceToHost));
        }
        static const char code2ascii[4] = {'A', 'C', 'T', 'G'};
        for (uint32_t i = 0; i < L; i++) {
            bases[i] = code2ascii[(w[i / 32] >> (2 * (i % 32))) & 3];
            if (r->has_invalid && ((inv[i / 32] >> (i % 32)) & 1)) bases[i] = 'N';
        }
    }
    if (quals) {
        if (!r->has_qual) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_get: batch has no qualities");
        uint64_t qo[2];
        MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, qo, r->d_qual_off.p + index, 16, hipMemcpyDeviceToHost));
        if (L) MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, quals, r->d_qual.p + qo[0], L, hipMemcpyDeviceToHost));
    }
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_reads_export_ascii(mdbg_ctx *ctx, const mdbg_reads *r, uint32_t first, uint32_t count,
                                       char *bases, uint64_t *offsets, uint64_t *n_bytes) try {
    if (!ctx || !r || (uint64_t)first + count > r->n_reads) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_export_ascii: bad range");
    if (r->has_invalid) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_export_ascii: batch holds N bases; use mdbg_reads_get");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    MDBG_HIP_CHECK(ctx, reads_ready_host(r));
    std::vector<uint64_t> woff((size_t)count + 1);
    std::vector<uint32_t> lens(count);
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, woff.data(), r->d_word_off.p + first, woff.size() * 8, hipMemcpyDeviceToHost));
    if (count) MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, lens.data(), r->d_len.p + first, (size_t)count * 4, hipMemcpyDeviceToHost));
    uint64_t total = 0;
    for (uint32_t i = 0; i < count; i++) total += lens[i];
    if (n_bytes) *n_bytes = total;
    if (!bases) return MDBG_OK;
    std::vector<uint64_t> w(woff[count] - woff[0]);
    if (!w.empty()) MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, w.data(), r->d_words.p + woff[0], w.size() * 8, hipMemcpyDeviceToHost));
    static const char code2ascii[4] = {'A', 'C', 'T', 'G'};
    // four bases per table look-up (a byte of packed codes -> four characters): exports of 10 Gbp are part of every bench run
    static const std::vector<uint32_t> quad = [] {
        std::vector<uint32_t> t(256);
        for (unsigned v = 0; v < 256; v++) {
            uint32_t x = 0;
            for (unsigned j = 0; j < 4; j++) x |= (uint32_t)(unsigned char)code2ascii[(v >> (2 * j)) & 3] << (8 * j);
            t[v] = x;
        }
        return t;
    }();
    uint64_t o = 0;
    for (uint32_t i = 0; i < count; i++) {
        if (offsets) offsets[i] = o;
        const uint64_t *rw = w.data() + (woff[i] - woff[0]);
        const uint32_t L = lens[i], full = L & ~3u;
        char *dst = bases + o;
        for (uint32_t b = 0; b < full; b += 4) {
            const uint32_t q = quad[(rw[b >> 5] >> (2 * (b & 31))) & 255u];
            memcpy(dst + b, &q, 4);
        }
        for (uint32_t b = full; b < L; b++) dst[b] = code2ascii[(rw[b >> 5] >> (2 * (b & 31))) & 3];
        o += L;
    }
    if (offsets) offsets[count] = o;
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_reads_export_qualities(mdbg_ctx *ctx, const mdbg_reads *r, uint32_t first, uint32_t count, char *quals, uint64_t *n_bytes) try {
    if (!ctx || !r || (uint64_t)first + count > r->n_reads) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_export_qualities: bad range");
    if (!r->has_qual) return set_error(ctx, MDBG_EINVAL, "mdbg_reads_export_qualities: the batch has no qualities");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    MDBG_HIP_CHECK(ctx, reads_ready_host(r));
    uint64_t range[2] = {0, 0};
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &range[0], r->d_qual_off.p + first, 8, hipMemcpyDeviceToHost));
    MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, &range[1], r->d_qual_off.p + first + count, 8, hipMemcpyDeviceToHost));
    if (n_bytes) *n_bytes = range[1] - range[0];
    if (quals && range[1] > range[0]) MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, quals, r->d_qual.p + range[0], range[1] - range[0], hipMemcpyDeviceToHost));
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" int mdbg_memcpy_device(mdbg_ctx *ctx, void *dst, const void *src, uint64_t bytes) try {
    if (!ctx || (bytes && (!dst || !src))) return set_error(ctx, MDBG_EINVAL, "mdbg_memcpy_device: null argument");
    MDBG_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    if (bytes) MDBG_HIP_CHECK(ctx, memcpy_sync(ctx, dst, src, bytes, hipMemcpyDeviceToDevice));
    return MDBG_OK;
} MDBG_API_CATCH(ctx)

extern "C" void mdbg_reads_free(mdbg_reads *r) { delete r; }
