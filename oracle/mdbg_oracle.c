/*
 * mdbg_oracle.c -- CPU restatement of metaMDBG's minimizer + k-min-mer hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see mdbg_oracle.h).  Parity status: PINNED against the
 * reference's own code built into oracle/_ref/ and the fixtures under tests/golden/.
 * Citations are file:line under /root/reference/src.
 */
#include "mdbg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------ */
/* MurmurHash3 x64-128 (utils/MurmurHash3.cpp:52-81 helpers, :328-405 body)   */
/* ------------------------------------------------------------------------ */

static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }

static inline uint64_t fmix64(uint64_t k) /* utils/MurmurHash3.cpp:72-81 */
{
    k ^= k >> 33;
    k *= 0xff51afd7ed558ccdULL;
    k ^= k >> 33;
    k *= 0xc4ceb9fe1a85ec53ULL;
    k ^= k >> 33;
    return k;
}

void orc_murmur3_x64_128(const void *key, int len, uint32_t seed, uint64_t out[2])
{
    const uint8_t *data = (const uint8_t *)key;
    const int nblocks = len / 16;
    uint64_t h1 = seed, h2 = seed;
    const uint64_t c1 = 0x87c37b91114253d5ULL, c2 = 0x4cf5ad432745937fULL;

    for (int i = 0; i < nblocks; i++) { /* :346-359 */
        uint64_t k1, k2;
        memcpy(&k1, data + 16 * i, 8);
        memcpy(&k2, data + 16 * i + 8, 8);
        k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1;
        h1 = rotl64(h1, 27); h1 += h2; h1 = h1 * 5 + 0x52dce729;
        k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2;
        h2 = rotl64(h2, 31); h2 += h1; h2 = h2 * 5 + 0x38495ab5;
    }

    const uint8_t *tail = data + nblocks * 16; /* :364-389 */
    uint64_t k1 = 0, k2 = 0;
    int rem = len & 15;
    for (int i = rem - 1; i >= 8; i--) k2 ^= (uint64_t)tail[i] << (8 * (i - 8));
    if (rem > 8) { k2 *= c2; k2 = rotl64(k2, 33); k2 *= c1; h2 ^= k2; }
    for (int i = (rem > 8 ? 8 : rem) - 1; i >= 0; i--) k1 ^= (uint64_t)tail[i] << (8 * i);
    if (rem > 0) { k1 *= c1; k1 = rotl64(k1, 31); k1 *= c2; h1 ^= k1; }

    h1 ^= (uint64_t)len; h2 ^= (uint64_t)len; /* :394-404 */
    h1 += h2; h2 += h1;
    h1 = fmix64(h1); h2 = fmix64(h2);
    h1 += h2; h2 += h1;
    out[0] = h1; out[1] = h2;
}

uint64_t orc_kmer_hash(uint64_t v) /* utils/kmer/Kmer.hpp:1421 -> utils/MurmurHash3.cpp:246-325 */
{
    uint64_t out[2];
    orc_murmur3_x64_128(&v, 8, 42, out);
    return out[0];
}

uint64_t orc_density_threshold(float density)
{
    /* utils/kmer/Kmer.hpp:1357-1358: u64 max -> double is 2^64; bound = (double)density * 2^64.
     * :1434 compares (double)hash < bound.  (double)h is monotone in h, so binary-search the
     * first h whose conversion reaches the bound. */
    const double bound = (double)density * 18446744073709551616.0;
    if (!((double)UINT64_MAX >= bound)) return UINT64_MAX; /* every hash passes (density >= 1) */
    uint64_t lo = 0, hi = UINT64_MAX; /* invariant: (double)hi >= bound */
    while (lo < hi) {
        uint64_t mid = lo + (hi - lo) / 2;
        if ((double)mid >= bound) hi = mid; else lo = mid + 1;
    }
    return lo;
}

/* ------------------------------------------------------------------------ */
/* base space                                                               */
/* ------------------------------------------------------------------------ */

size_t orc_hpc_encode(const char *seq, size_t len, int hpc, char *out_seq, uint64_t *rle_pos)
{
    size_t n = 0;
    if (hpc) { /* Commons.hpp:4169-4191 */
        char last = '#';
        uint64_t last_pos = 0;
        for (size_t i = 0; i < len; i++) {
            char c = seq[i];
            if (c == last) continue;
            if (last != '#') {
                out_seq[n] = last;
                rle_pos[n] = last_pos;
                n++;
                last_pos = i;
            }
            last = c;
        }
        out_seq[n] = last;
        rle_pos[n] = last_pos;
        n++;
        rle_pos[n] = len;
        return n;
    }
    /* Commons.hpp:4193-4199 */
    memcpy(out_seq, seq, len);
    for (size_t i = 0; i < len; i++) rle_pos[i] = i;
    return len;
}

size_t orc_kmer_iterate(const char *seq, size_t len, unsigned K, uint64_t *kmers, uint8_t *dirs)
{
    /* utils/kmer/Kmer.hpp:531-611; ConvertASCII :462; comp_NT :31 (A<->T, C<->G == code ^ 2) */
    if (len < K) return 0;
    const uint64_t mask = (K >= 32) ? ~0ULL : ((1ULL << (2 * K)) - 1);
    uint64_t fwd = 0, rev = 0;
    int bad = -1;
    for (unsigned i = 0; i < K; i++) { /* polynom + reverse (:594-602) */
        unsigned c = ((unsigned char)seq[i] >> 1) & 3;
        fwd = (fwd << 2) + c;
        rev = (rev >> 2) + ((uint64_t)(c ^ 2) << (2 * (K - 1)));
        if (((unsigned char)seq[i] >> 3) & 1) bad = (int)i;
    }
    size_t n = 0;
    int d = (fwd < rev) ? 0 : 1;
    kmers[n] = (bad < 0) ? (d ? rev : fwd) : UINT64_MAX;
    dirs[n] = (uint8_t)d;
    n++;
    for (size_t idx = K; idx < len; idx++) {
        unsigned c = ((unsigned char)seq[idx] >> 1) & 3;
        if (((unsigned char)seq[idx] >> 3) & 1) bad = (int)K - 1; else bad--;
        fwd = ((fwd << 2) + c) & mask;
        rev = ((rev >> 2) + ((uint64_t)(c ^ 2) << (2 * (K - 1)))) & mask;
        d = (fwd < rev) ? 0 : 1;
        kmers[n] = (bad < 0) ? (d ? rev : fwd) : UINT64_MAX;
        dirs[n] = (uint8_t)d;
        n++;
    }
    return n;
}

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

size_t orc_minimizer_parse(const char *seq, size_t len, unsigned K, float density,
                           const uint32_t *repetitive, size_t n_rep,
                           uint32_t *out_min, uint32_t *out_pos, uint8_t *out_dir)
{
    return orc_minimizer_parse_trim(seq, len, K, density, repetitive, n_rep, 1, out_min, out_pos, out_dir);
}

size_t orc_minimizer_parse_trim(const char *seq, size_t len, unsigned K, float density,
                                const uint32_t *repetitive, size_t n_rep, size_t trim_bps,
                                uint32_t *out_min, uint32_t *out_pos, uint8_t *out_dir)
{
    if (len < K) return 0;
    size_t nk = len - K + 1;
    uint64_t *kmers = (uint64_t *)malloc(nk * sizeof(uint64_t));
    uint8_t *dirs = (uint8_t *)malloc(nk);
    orc_kmer_iterate(seq, len, K, kmers, dirs);
    /* utils/kmer/Kmer.hpp:1357-1358, compared as double at :1434 -- done literally here */
    const double bound = (double)density * 18446744073709551616.0;
    uint32_t *rep_sorted = NULL;
    if (n_rep) {
        rep_sorted = (uint32_t *)malloc(n_rep * sizeof(uint32_t));
        memcpy(rep_sorted, repetitive, n_rep * sizeof(uint32_t));
        qsort(rep_sorted, n_rep, sizeof(uint32_t), cmp_u32);
    }
    size_t n = 0;
    for (size_t pos = trim_bps; pos + trim_bps < nk; pos++) { /* :1395; _trimBps = 1 by default (:1362) */
        uint64_t v = kmers[pos];
        uint64_t h = orc_kmer_hash(v);
        if ((double)h < bound) {
            uint32_t v32 = (uint32_t)v; /* MinimizerType is u32 (:23) */
            if (n_rep && bsearch(&v32, rep_sorted, n_rep, sizeof(uint32_t), cmp_u32)) continue; /* :1437 */
            out_min[n] = v32;
            out_pos[n] = (uint32_t)pos;
            out_dir[n] = dirs[pos];
            n++;
        }
    }
    free(rep_sorted);
    free(kmers);
    free(dirs);
    return n;
}

double orc_sequence_complexity(const char *seq, size_t len)
{
    /* readSelection/ReadSelection.hpp:1171-1228; KmerModelDirect(3)::iterate (Kmer.hpp:745-799)
     * yields FORWARD 3-mers.  w = 64, step = 32, l = w - 2. */
    const size_t w = 64, step = 32;
    const double l = (double)(w - 2);
    double nb_windows = 0, window_score_sum = 0;
    if (len >= 3) {
        size_t nk = len - 3 + 1;
        int16_t *kmers = (int16_t *)malloc(nk * sizeof(int16_t));
        for (size_t i = 0; i < nk; i++) {
            int badc = 0, v = 0;
            for (int j = 0; j < 3; j++) {
                unsigned char ch = (unsigned char)seq[i + j];
                v = (v << 2) + ((ch >> 1) & 3);
                badc |= (ch >> 3) & 1;
            }
            /* the reference stores -1 for a 3-mer touching N and then indexes kmerCounts[-1] (heap
             * corruption, :1196); with no defined behaviour to restate, N counts with its 2-bit code */
            (void)badc;
            kmers[i] = (int16_t)v;
        }
        for (size_t ii = 0; ii < nk; ii += step) {
            double counts[64];
            for (int i = 0; i < 64; i++) counts[i] = 0;
            size_t nb = 0;
            for (size_t i = ii; i < nk; i++) {
                if (kmers[i] >= 0) counts[kmers[i]] += 1;
                nb++;
                if (nb == w) break;
            }
            if (nb < w) continue;
            double score = 0;
            for (int i = 0; i < 64; i++) score += counts[i] * (counts[i] - 1) / 2.0;
            score /= (l - 1);
            nb_windows += 1;
            window_score_sum += score;
        }
        free(kmers);
    }
    return window_score_sum / nb_windows;
}

static float g_q2err[256];
static int g_q2err_init = 0;
static void init_q2err(void) /* readSelection/ReadSelection.hpp:101-104 */
{
    if (g_q2err_init) return;
    for (int q = 0; q < 256; q++) g_q2err[q] = 0;
    for (int q = 33; q <= 127; q++) {
        float qq = (float)(uint8_t)(q - 33);     /* Commons.hpp:2338-2341 */
        g_q2err[q] = powf(10.0f, -qq / 10.0f);
    }
    g_q2err_init = 1;
}

float orc_mean_read_quality(const char *qual, size_t len)
{
    init_q2err();
    long double error_sum = 0; /* readSelection/ReadSelection.hpp:870-879 */
    if (qual)
        for (size_t i = 0; i < len; i++) error_sum += g_q2err[(uint8_t)qual[i]];
    size_t n = qual ? len : 0;
    float mean_err = (float)(error_sum / n);
    return -10.0f * log10f(mean_err);
}

void orc_read_selection(const char *seq, const char *qual, size_t len,
                        const orc_scan_params *p, orc_read_record *rec)
{
    /* readSelection/ReadSelection.hpp:669-1158 */
    char *rle = (char *)malloc(len + 2);
    uint64_t *rle_pos = (uint64_t *)malloc((len + 2) * sizeof(uint64_t));
    size_t hl = orc_hpc_encode(seq, len, p->hpc, rle, rle_pos);
    size_t cap = hl ? hl : 1;
    rec->minimizers = (uint32_t *)malloc(cap * sizeof(uint32_t));
    rec->pos = (uint32_t *)malloc(cap * sizeof(uint32_t));
    rec->dir = (uint8_t *)malloc(cap);
    rec->qual = (uint8_t *)malloc(cap);
    size_t n = orc_minimizer_parse(rle, hl, p->K, p->density, p->repetitive, p->n_rep,
                                   rec->minimizers, rec->pos, rec->dir);
    rec->hpc_length = (uint32_t)hl;
    rec->read_length = (uint32_t)len;
    rec->mean_quality = orc_mean_read_quality(qual, len);
    rec->low_complexity = 0;
    rec->low_quality = 0;
    double cx = orc_sequence_complexity(seq, len);
    if (cx > 5) { rec->low_complexity = 1; n = 0; }                 /* :890-899 */
    if (rec->mean_quality < p->min_read_quality) { rec->low_quality = 1; n = 0; } /* :901-909 */
    for (size_t i = 0; i < n; i++) {
        if (!qual) { rec->qual[i] = 1; continue; }                   /* :1047-1051 */
        /* getMinQuality over [rle[pos], rle[pos+K]) of q-33 as u8 (:1302-1320) */
        uint64_t s = rle_pos[rec->pos[i]], e = rle_pos[rec->pos[i] + p->K];
        uint8_t mq = 255;
        for (uint64_t j = s; j < e; j++) {
            uint8_t q = (uint8_t)qual[j];
            q = (uint8_t)(q - 33);
            if (q < mq) mq = q;
        }
        rec->qual[i] = mq;
    }
    rec->n = (uint32_t)n;
    free(rle);
    free(rle_pos);
}

void orc_correction_scan(const char *seq, const char *qual, size_t len,
                         const orc_scan_params *p, orc_read_record *rec)
{
    /* readSelection/ReadCorrection.hpp:2318-2372 */
    char *rle = (char *)malloc(len + 2);
    uint64_t *rle_pos = (uint64_t *)malloc((len + 2) * sizeof(uint64_t));
    size_t hl = orc_hpc_encode(seq, len, p->hpc, rle, rle_pos);
    size_t cap = hl ? hl : 1;
    rec->minimizers = (uint32_t *)malloc(cap * sizeof(uint32_t));
    rec->pos = (uint32_t *)malloc(cap * sizeof(uint32_t));
    rec->dir = (uint8_t *)malloc(cap);
    rec->qual = (uint8_t *)malloc(cap);
    size_t n = orc_minimizer_parse(rle, hl, p->K, p->density, p->repetitive, p->n_rep,
                                   rec->minimizers, rec->pos, rec->dir);
    rec->hpc_length = (uint32_t)hl;
    rec->read_length = (uint32_t)len;
    rec->mean_quality = 0.0f;                                        /* :2370 */
    rec->low_complexity = 0;
    rec->low_quality = 0;
    for (size_t i = 0; i < n; i++) {
        if (!qual) { rec->qual[i] = 1; continue; }                   /* :2328-2332 */
        uint64_t s = rle_pos[rec->pos[i]], e = rle_pos[rec->pos[i] + p->K - 1];   /* :2340 */
        uint8_t mq = 255;
        for (uint64_t j = s; j <= e; j++) {                          /* :2472, inclusive */
            uint8_t q = (uint8_t)qual[j];
            q = (uint8_t)(q - 33);
            if (q < mq) mq = q;
        }
        rec->qual[i] = mq;
    }
    rec->n = (uint32_t)n;
    free(rle);
    free(rle_pos);
}

size_t orc_apply_density_threshold(const uint32_t *minimizers, size_t n, float density, uint8_t *keep)
{
    /* Commons.hpp:2524-2533: double bound = density(f32) * (u64)-1 -- a float product, exact (power of two);
     * the key hashed is the stored minimizer widened to u64 */
    uint64_t max_hash = (uint64_t)-1;
    double bound = density * max_hash;
    size_t kept = 0;
    for (size_t i = 0; i < n; i++) {
        keep[i] = (double)orc_kmer_hash((uint64_t)minimizers[i]) < bound;
        kept += keep[i];
    }
    return kept;
}

void orc_read_record_free(orc_read_record *rec)
{
    free(rec->minimizers); free(rec->pos); free(rec->dir); free(rec->qual);
    rec->minimizers = NULL; rec->pos = NULL; rec->dir = NULL; rec->qual = NULL;
}

size_t orc_write_read_record(const orc_read_record *rec, uint8_t *buf)
{
    /* readSelection/ReadSelection.hpp:415-467: u32 n; u8 circ=0; u32 m[n]; u32 pos[n]; u8 dir[n];
     * u8 qual[n]; f32 meanQ; u32 readLen */
    uint8_t *p = buf;
    uint32_t n = rec->n;
    memcpy(p, &n, 4); p += 4;
    *p++ = 0;
    memcpy(p, rec->minimizers, 4 * (size_t)n); p += 4 * (size_t)n;
    memcpy(p, rec->pos, 4 * (size_t)n); p += 4 * (size_t)n;
    memcpy(p, rec->dir, n); p += n;
    memcpy(p, rec->qual, n); p += n;
    memcpy(p, &rec->mean_quality, 4); p += 4;
    memcpy(p, &rec->read_length, 4); p += 4;
    return (size_t)(p - buf);
}

static int cmp_u32_desc(const void *a, const void *b) { return -cmp_u32(a, b); }

uint32_t orc_compute_n50(const uint32_t *lengths, size_t n)
{
    /* Commons.hpp:2291-2322: sort descending, cumulative, reverse both; first i (ascending length
     * order, cumul from the big end) whose cumul < total/2 gives n50; default = largest. */
    if (n == 0) return 0;
    uint32_t *s = (uint32_t *)malloc(n * sizeof(uint32_t));
    uint64_t *cum = (uint64_t *)malloc(n * sizeof(uint64_t));
    memcpy(s, lengths, n * sizeof(uint32_t));
    qsort(s, n, sizeof(uint32_t), cmp_u32_desc);
    uint64_t c = 0;
    for (size_t i = 0; i < n; i++) { c += s[i]; cum[i] = c; }
    /* after the two std::reverse calls: lens[i] = s[n-1-i], cumuls[i] = cum[n-1-i] */
    uint32_t n50 = s[0];            /* allReadLengths[size-1] after reverse == largest */
    uint64_t half = cum[n - 1] / 2; /* readLengthCumuls[0] after reverse == total */
    for (size_t i = 0; i < n; i++) {
        if (cum[n - 1 - i] < half) { n50 = s[n - 1 - i]; break; }
    }
    free(s); free(cum);
    return n50;
}

uint32_t orc_compute_mean_length(const uint32_t *lengths, size_t n)
{
    long double sum = 0, cnt = 0; /* Commons.hpp:2324-2336 */
    for (size_t i = 0; i < n; i++) { sum += lengths[i]; cnt += 1; }
    return (uint32_t)(uint64_t)(sum / cnt);
}

int orc_compute_last_k(float density, size_t n50, size_t first_k, size_t max_k)
{
    size_t last_k = (size_t)(n50 * density * 2.0f); /* Commons.hpp:1726-1741 (float arithmetic) */
    if (max_k > 0) last_k = max_k;
    if (last_k < first_k + 2) last_k = first_k + 2;
    return (int)last_k;
}

size_t orc_purge_palindrome(uint32_t *m, size_t n, size_t first_k, size_t last_k)
{
    /* Commons.hpp:1617-1723.  Window = first k non-banned minimizers from i; palindrome test is
     * KmerVec::isPalindrome (:918-921): first k/2 entries equal the reversed tail. */
    uint8_t *banned = (uint8_t *)calloc(n ? n : 1, 1);
    uint32_t *win = (uint32_t *)malloc((last_k ? last_k : 1) * sizeof(uint32_t));
    for (;;) {
        int has = 0;
        for (size_t k = first_k; k < last_k && !has; k++) {
            int i_max = (int)n - (int)k + 1;
            for (int i = 0; i < i_max && !has; i++) {
                if (banned[i]) continue;
                size_t cnt = 0;
                for (size_t j = (size_t)i; j < n && cnt < k; j++) {
                    if (banned[j]) continue;
                    win[cnt++] = m[j];
                }
                if (cnt == k) {
                    int pal = 1;
                    for (size_t t = 0; t < k / 2; t++)
                        if (win[t] != win[k - 1 - t]) { pal = 0; break; }
                    if (pal) { banned[i] = 1; has = 1; }
                }
            }
        }
        if (!has) break;
    }
    size_t o = 0;
    for (size_t i = 0; i < n; i++) if (!banned[i]) m[o++] = m[i];
    free(banned); free(win);
    return o;
}

/* ------------------------------------------------------------------------ */
/* minimizer space                                                          */
/* ------------------------------------------------------------------------ */

int orc_kminmer_normalize(const uint32_t *v, unsigned k, uint32_t *out)
{
    /* Commons.hpp:886-916: strictly smaller forward keeps orientation; tie or larger reverses. */
    int reversed = 1;
    for (unsigned i = 0; i < k; i++) {
        uint32_t a = v[i], b = v[k - 1 - i];
        if (a == b) continue;
        reversed = (a < b) ? 0 : 1;
        break;
    }
    if (reversed) for (unsigned i = 0; i < k; i++) out[i] = v[k - 1 - i];
    else          for (unsigned i = 0; i < k; i++) out[i] = v[i];
    return reversed;
}

void orc_kminmer_hash128(const uint32_t *v, unsigned k, uint64_t *hi, uint64_t *lo)
{
    uint64_t out[2]; /* Commons.hpp:941-969 */
    orc_murmur3_x64_128(v, (int)(k * 4), 0, out);
    *hi = out[0];
    *lo = out[1];
}

void orc_kminmer_table_free(orc_kminmer_table *t)
{
    free(t->vecs); free(t->hash_lo); free(t->hash_hi); free(t->abundance);
    memset(t, 0, sizeof(*t));
}

/* instances: all windows of k consecutive minimizers of every sequence, normalised
 * (MDBG::getKminmers_complete, Commons.hpp:5282-5361). */
typedef struct { uint64_t n; uint32_t *vecs; uint64_t *seq_first; /* n_seqs+1 */ } inst_list;

static void enumerate_instances(const uint32_t *mins, const uint64_t *off, uint64_t n_seqs, unsigned k, inst_list *L)
{
    uint64_t total = 0;
    L->seq_first = (uint64_t *)malloc((n_seqs + 1) * sizeof(uint64_t));
    for (uint64_t r = 0; r < n_seqs; r++) {
        uint64_t n = off[r + 1] - off[r];
        L->seq_first[r] = total;
        if (n >= k) total += n - k + 1;
    }
    L->seq_first[n_seqs] = total;
    L->n = total;
    L->vecs = (uint32_t *)malloc((total ? total : 1) * k * sizeof(uint32_t));
    for (uint64_t r = 0; r < n_seqs; r++) {
        uint64_t n = off[r + 1] - off[r];
        if (n < k) continue;
        for (uint64_t i = 0; i + k <= n; i++)
            orc_kminmer_normalize(mins + off[r] + i, k, L->vecs + (L->seq_first[r] + i) * k);
    }
}

static unsigned g_sort_k;
static const uint32_t *g_sort_vecs;
static int cmp_inst(const void *a, const void *b) /* KmerVec operator< (Commons.hpp:754-773) */
{
    const uint32_t *x = g_sort_vecs + (size_t)(*(const uint64_t *)a) * g_sort_k;
    const uint32_t *y = g_sort_vecs + (size_t)(*(const uint64_t *)b) * g_sort_k;
    for (unsigned i = 0; i < g_sort_k; i++) {
        if (x[i] == y[i]) continue;
        return x[i] < y[i] ? -1 : 1;
    }
    return 0;
}

static uint64_t *sorted_instance_order(const inst_list *L, unsigned k)
{
    uint64_t *idx = (uint64_t *)malloc((L->n ? L->n : 1) * sizeof(uint64_t));
    for (uint64_t i = 0; i < L->n; i++) idx[i] = i;
    g_sort_k = k; g_sort_vecs = L->vecs;
    qsort(idx, L->n, sizeof(uint64_t), cmp_inst);
    return idx;
}

static void table_reserve(orc_kminmer_table *t, uint64_t cap, unsigned k, int with_vecs)
{
    memset(t, 0, sizeof(*t));
    t->k = k;
    if (!cap) cap = 1;
    t->vecs = with_vecs ? (uint32_t *)malloc(cap * k * sizeof(uint32_t)) : NULL;
    t->hash_lo = (uint64_t *)malloc(cap * sizeof(uint64_t));
    t->hash_hi = (uint64_t *)malloc(cap * sizeof(uint64_t));
    t->abundance = (uint32_t *)malloc(cap * sizeof(uint32_t));
}

static void table_push(orc_kminmer_table *t, const uint32_t *vec, uint32_t abundance)
{
    uint64_t i = t->n++;
    if (t->vecs) memcpy(t->vecs + i * t->k, vec, t->k * sizeof(uint32_t));
    orc_kminmer_hash128(vec, t->k, &t->hash_hi[i], &t->hash_lo[i]);
    t->abundance[i] = abundance;
}

static uint32_t median_u32(uint32_t *v, size_t n) /* Utils::compute_median (Commons.hpp:2972-2988) */
{
    if (n == 0) return 0;
    qsort(v, n, sizeof(uint32_t), cmp_u32);
    if (n % 2 == 0) return (uint32_t)(v[n / 2 - 1] + v[n / 2]) / 2; /* u32 arithmetic, as T=u_int32_t */
    return v[n / 2];
}

void orc_kminmer_count_first(const uint32_t *mins, const uint64_t *off, uint64_t n_reads,
                             unsigned k, uint32_t min_abundance, orc_kminmer_table *out)
{
    inst_list L;
    enumerate_instances(mins, off, n_reads, k, &L);
    uint64_t *idx = sorted_instance_order(&L, k);
    uint32_t *cnt = (uint32_t *)malloc((L.n ? L.n : 1) * sizeof(uint32_t));
    table_reserve(out, L.n, k, 1);
    /* dereplicatePartition run-length (graph/CreateMdbg.hpp:3812-3840) + dumpKminmer filter (:3862-3883) */
    for (uint64_t s = 0; s < L.n;) {
        uint64_t e = s + 1;
        while (e < L.n && cmp_inst(&idx[s], &idx[e]) == 0) e++;
        uint32_t c = (uint32_t)(e - s);
        int solid = (c > 1) && !(c < min_abundance);
        for (uint64_t t = s; t < e; t++) cnt[idx[t]] = solid ? c : 0; /* 0 == not in solid table */
        if (solid) table_push(out, L.vecs + idx[s] * k, c);
        s = e;
    }
    out->n_solid = out->n;
    if (min_abundance <= 1) { /* graph/CreateMdbg.cpp:317-319 -> CreateMdbg.hpp:4562-4640 */
        uint32_t *ab = (uint32_t *)malloc(sizeof(uint32_t) * 1);
        size_t ab_cap = 1;
        for (uint64_t r = 0; r < n_reads; r++) {
            uint64_t f = L.seq_first[r], n = L.seq_first[r + 1] - f;
            if (n > ab_cap) { ab_cap = n; ab = (uint32_t *)realloc(ab, ab_cap * sizeof(uint32_t)); }
            int all_one = 1;
            for (uint64_t i = 0; i < n; i++) {
                if (cnt[f + i]) { ab[i] = cnt[f + i]; all_one = 0; } else ab[i] = 1;
            }
            uint32_t median = median_u32(ab, n);
            double cutoff = median * 0.1f; /* u32 * float -> float, widened (:4610) */
            if (cutoff > 1) continue;
            if (all_one) continue;
            for (uint64_t i = 0; i < n; i++)
                if (!cnt[f + i]) table_push(out, L.vecs + (f + i) * k, 1);
        }
        free(ab);
    }
    free(cnt); free(idx); free(L.vecs); free(L.seq_first);
}

/* ---- abundance map (sorted arrays + pending overlay list) ---------------- */

typedef struct { uint64_t hi, lo; uint32_t a; uint32_t seq; } amap_ent;

static int cmp_ent(const void *x, const void *y)
{
    const amap_ent *a = (const amap_ent *)x, *b = (const amap_ent *)y;
    if (a->hi != b->hi) return a->hi < b->hi ? -1 : 1;
    if (a->lo != b->lo) return a->lo < b->lo ? -1 : 1;
    return (a->seq > b->seq) - (a->seq < b->seq);
}

static void amap_from_ents(amap_ent *e, uint64_t n, orc_abundance_map *m)
{
    /* later entries (higher seq) override earlier ones for equal keys; a==UINT32_MAX marks
     * "set to 0 if present" (overlay with unitig abundance 1). */
    qsort(e, n, sizeof(amap_ent), cmp_ent);
    m->hi = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    m->lo = (uint64_t *)malloc((n ? n : 1) * sizeof(uint64_t));
    m->abundance = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    uint64_t o = 0;
    for (uint64_t s = 0; s < n;) {
        uint64_t t = s;
        int present = 0; uint32_t val = 0;
        while (t < n && e[t].hi == e[s].hi && e[t].lo == e[s].lo) {
            if (e[t].a == UINT32_MAX) { if (present) val = 0; }
            else { present = 1; val = e[t].a; }
            t++;
        }
        if (present) { m->hi[o] = e[s].hi; m->lo[o] = e[s].lo; m->abundance[o] = val; o++; }
        s = t;
    }
    m->n = o;
}

void orc_abundance_map_from_records(const uint8_t *rec, uint64_t n_rec, orc_abundance_map *m)
{
    amap_ent *e = (amap_ent *)malloc((n_rec ? n_rec : 1) * sizeof(amap_ent));
    uint64_t n = 0;
    for (uint64_t i = 0; i < n_rec; i++) {
        uint32_t a;
        memcpy(&a, rec + 20 * i + 16, 4);
        if (a == 1) continue; /* graph/CreateMdbg.cpp:3445 */
        memcpy(&e[n].lo, rec + 20 * i, 8);
        memcpy(&e[n].hi, rec + 20 * i + 8, 8);
        e[n].a = a; e[n].seq = 0;
        n++;
    }
    amap_from_ents(e, n, m);
    free(e);
}

/* overlays are accumulated in a side list hung off the map until finish() */
static amap_ent *g_pending = NULL;
static uint64_t g_pending_n = 0, g_pending_cap = 0;
static uint32_t g_pending_seq = 1;

void orc_abundance_map_overlay(orc_abundance_map *m, const uint32_t *unitig, uint32_t n, unsigned kprev, uint32_t a)
{
    (void)m;
    if (n < kprev) return;
    uint32_t *tmp = (uint32_t *)malloc(kprev * sizeof(uint32_t));
    for (uint32_t i = 0; i + kprev <= n; i++) {
        if (g_pending_n == g_pending_cap) {
            g_pending_cap = g_pending_cap ? g_pending_cap * 2 : 1024;
            g_pending = (amap_ent *)realloc(g_pending, g_pending_cap * sizeof(amap_ent));
        }
        orc_kminmer_normalize(unitig + i, kprev, tmp); /* graph/CreateMdbg.cpp:3489-3491 */
        amap_ent *e = &g_pending[g_pending_n++];
        orc_kminmer_hash128(tmp, kprev, &e->hi, &e->lo);
        e->a = (a == 1) ? UINT32_MAX : a; /* :3493-3506 */
        e->seq = g_pending_seq++;
    }
    free(tmp);
}

void orc_abundance_map_finish(orc_abundance_map *m)
{
    uint64_t n = m->n + g_pending_n;
    amap_ent *e = (amap_ent *)malloc((n ? n : 1) * sizeof(amap_ent));
    for (uint64_t i = 0; i < m->n; i++) { e[i].hi = m->hi[i]; e[i].lo = m->lo[i]; e[i].a = m->abundance[i]; e[i].seq = 0; }
    memcpy(e + m->n, g_pending, g_pending_n * sizeof(amap_ent));
    free(m->hi); free(m->lo); free(m->abundance);
    amap_from_ents(e, n, m);
    free(e);
    free(g_pending); g_pending = NULL; g_pending_n = g_pending_cap = 0; g_pending_seq = 1;
}

int orc_abundance_map_get(const orc_abundance_map *m, uint64_t hi, uint64_t lo, uint32_t *a)
{
    uint64_t l = 0, r = m->n;
    while (l < r) {
        uint64_t mid = l + (r - l) / 2;
        if (m->hi[mid] < hi || (m->hi[mid] == hi && m->lo[mid] < lo)) l = mid + 1; else r = mid;
    }
    if (l < m->n && m->hi[l] == hi && m->lo[l] == lo) { *a = m->abundance[l]; return 1; }
    return 0;
}

void orc_abundance_map_free(orc_abundance_map *m)
{
    free(m->hi); free(m->lo); free(m->abundance);
    memset(m, 0, sizeof(*m));
}

void orc_kminmer_count_refined(const uint32_t *mins, const uint64_t *off, uint64_t n_seqs,
                               unsigned k, const orc_abundance_map *prev, orc_kminmer_table *out)
{
    inst_list L;
    enumerate_instances(mins, off, n_seqs, k, &L);
    uint64_t *idx = sorted_instance_order(&L, k);
    table_reserve(out, L.n, k, 1);
    uint32_t *tmp = (uint32_t *)malloc(k * sizeof(uint32_t));
    for (uint64_t s = 0; s < L.n;) {
        uint64_t e = s + 1;
        while (e < L.n && cmp_inst(&idx[s], &idx[e]) == 0) e++;
        const uint32_t *vec = L.vecs + idx[s] * k;
        /* getRefinedAbundance (graph/CreateMdbg.hpp:3933-3970): (k-1)-min-mers of the canonical vec */
        uint32_t min_ab = UINT32_MAX;
        for (unsigned i = 0; i + (k - 1) <= k; i++) {
            uint64_t hi, lo; uint32_t a;
            orc_kminmer_normalize(vec + i, k - 1, tmp);
            orc_kminmer_hash128(tmp, k - 1, &hi, &lo);
            if (orc_abundance_map_get(prev, hi, lo, &a)) {
                if (a == 0) { min_ab = 1; break; }
                if (a < min_ab) min_ab = a;
            } else { min_ab = 1; break; }
        }
        if (min_ab > 1) table_push(out, vec, min_ab); /* dumpKminmer: abundance <= 1 dropped (:3867) */
        s = e;
    }
    out->n_solid = out->n;
    free(tmp); free(idx); free(L.vecs); free(L.seq_first);
}

typedef struct { uint64_t hi, lo; uint32_t a; } idx_ent;
static int cmp_idx_ent(const void *x, const void *y)
{
    const idx_ent *a = (const idx_ent *)x, *b = (const idx_ent *)y;
    if (a->hi != b->hi) return a->hi < b->hi ? -1 : 1;
    if (a->lo != b->lo) return a->lo < b->lo ? -1 : 1;
    return 0;
}

void orc_kminmer_index(const uint32_t *mins, const uint64_t *off, uint64_t n_seqs,
                       unsigned k, const orc_abundance_map *prev, orc_kminmer_table *out)
{
    uint64_t cap = 0;
    for (uint64_t r = 0; r < n_seqs; r++) { uint64_t n = off[r + 1] - off[r]; if (n >= k) cap += n - k + 1; }
    idx_ent *ents = (idx_ent *)malloc((cap ? cap : 1) * sizeof(idx_ent));
    uint64_t ne = 0;
    uint32_t *tmp = (uint32_t *)malloc(k * sizeof(uint32_t));
    for (uint64_t r = 0; r < n_seqs; r++) {
        const uint32_t *m = mins + off[r];
        uint64_t n = off[r + 1] - off[r];
        if (n < k) continue;
        /* getPrevAbundances (graph/CreateMdbg.hpp:1240-1265): one per (k-1)-min-mer, missing => 1 */
        uint64_t np = n - (k - 1) + 1;
        uint32_t *pa = (uint32_t *)malloc(np * sizeof(uint32_t));
        for (uint64_t i = 0; i < np; i++) {
            uint64_t hi, lo; uint32_t a;
            orc_kminmer_normalize(m + i, k - 1, tmp);
            orc_kminmer_hash128(tmp, k - 1, &hi, &lo);
            pa[i] = orc_abundance_map_get(prev, hi, lo, &a) ? a : 1;
        }
        for (uint64_t i = 0; i + k <= n; i++) {
            uint32_t a = pa[i] < pa[i + 1] ? pa[i] : pa[i + 1]; /* getAbundance (:988-1010) */
            if (a <= 1) continue;                               /* :1446-1448 */
            orc_kminmer_normalize(m + i, k, tmp);
            orc_kminmer_hash128(tmp, k, &ents[ne].hi, &ents[ne].lo);
            ents[ne].a = a;
            ne++;
        }
        free(pa);
    }
    qsort(ents, ne, sizeof(idx_ent), cmp_idx_ent);
    table_reserve(out, ne, k, 0);
    for (uint64_t s = 0; s < ne;) { /* insert-if-absent: one record per key (:1450-1459) */
        uint64_t e = s + 1;
        while (e < ne && ents[e].hi == ents[s].hi && ents[e].lo == ents[s].lo) e++;
        uint64_t i = out->n++;
        out->hash_hi[i] = ents[s].hi; out->hash_lo[i] = ents[s].lo; out->abundance[i] = ents[s].a;
        s = e;
    }
    out->n_solid = out->n;
    free(tmp); free(ents);
}

void orc_small_contigs(const uint32_t *mins, const uint64_t *off, uint64_t n_seqs, unsigned k, unsigned kprev,
                       const orc_abundance_map *prev, uint8_t *flags)
{
    uint32_t *tmp = (uint32_t *)malloc((kprev ? kprev : 1) * sizeof(uint32_t));
    for (uint64_t r = 0; r < n_seqs; r++) {
        const uint32_t *m = mins + off[r];
        uint64_t n = off[r + 1] - off[r];
        flags[r] = 0;
        if (n >= k || n < kprev) continue;       /* has k-min-mers (:1330) / empty prevAbundances (undefined in the reference) */
        uint32_t a = 0xFFFFFFFFu;
        uint64_t np = n - kprev + 1, use = np < 2 ? np : 2;   /* getAbundance(0, .): prev[0], or min(prev[0], prev[1]) (:990-1006) */
        for (uint64_t i = 0; i < use; i++) {
            uint64_t hi, lo; uint32_t v;
            orc_kminmer_normalize(m + i, kprev, tmp);
            orc_kminmer_hash128(tmp, kprev, &hi, &lo);
            if (!orc_abundance_map_get(prev, hi, lo, &v)) v = 1;
            if (v < a) a = v;
        }
        flags[r] = a > 1;
    }
    free(tmp);
}

uint64_t orc_edge_index(const uint32_t *vecs, uint64_t n, unsigned k, uint64_t *out_hi, uint64_t *out_lo, uint64_t *checksum)
{
    idx_ent *e = (idx_ent *)malloc((n ? 2 * n : 1) * sizeof(idx_ent));
    uint32_t *tmp = (uint32_t *)malloc(k * sizeof(uint32_t));
    for (uint64_t i = 0; i < n; i++) { /* partitionNode (graph/CreateMdbg.hpp:4083-4100): prefix = first k-1, suffix = last k-1 */
        orc_kminmer_normalize(vecs + i * k, k - 1, tmp);
        orc_kminmer_hash128(tmp, k - 1, &e[2 * i].hi, &e[2 * i].lo);
        orc_kminmer_normalize(vecs + i * k + 1, k - 1, tmp);
        orc_kminmer_hash128(tmp, k - 1, &e[2 * i + 1].hi, &e[2 * i + 1].lo);
        e[2 * i].a = e[2 * i + 1].a = 0;
    }
    qsort(e, 2 * n, sizeof(idx_ent), cmp_idx_ent);
    uint64_t m = 0, sum = 0;
    for (uint64_t i = 0; i < 2 * n; i++) {
        if (i && e[i].hi == e[i - 1].hi && e[i].lo == e[i - 1].lo) continue;   /* dereplicatePartition (:4147-4180) */
        out_hi[m] = e[i].hi; out_lo[m] = e[i].lo; m++;
        sum += e[i].lo;                                                        /* u64 += u128 keeps the low word */
    }
    *checksum = sum;
    free(e); free(tmp);
    return m;
}

uint64_t orc_unitig_edge_index(const uint32_t *minimizers, const uint64_t *offsets, uint64_t n_seqs, unsigned k,
                               uint64_t *out_hi, uint64_t *out_lo, uint64_t *checksum)
{
    idx_ent *e = (idx_ent *)malloc((n_seqs ? 4 * n_seqs : 1) * sizeof(idx_ent));
    uint32_t *node = (uint32_t *)malloc(k * sizeof(uint32_t));
    uint32_t *tmp = (uint32_t *)malloc(k * sizeof(uint32_t));
    uint64_t m = 0;
    for (uint64_t s = 0; s < n_seqs; s++) {
        const uint32_t *u = minimizers + offsets[s];
        const uint64_t n = offsets[s + 1] - offsets[s];
        if (n < k) continue;                                   /* minimizersToKminmers yields nothing */
        const uint64_t last = n - k;
        for (int which = 0; which < 2; which++) {
            if (which == 1 && last == 0) break;                /* startNode == endNode (:4378) */
            orc_kminmer_normalize(u + (which ? last : 0), k, node);          /* :4366 / :4380 */
            orc_kminmer_normalize(node + 1, k - 1, tmp);                     /* suffix (:4395) */
            orc_kminmer_hash128(tmp, k - 1, &e[m].hi, &e[m].lo); e[m].a = 0; m++;
            orc_kminmer_normalize(node, k - 1, tmp);                         /* prefix (:4394) */
            orc_kminmer_hash128(tmp, k - 1, &e[m].hi, &e[m].lo); e[m].a = 0; m++;
        }
    }
    qsort(e, m, sizeof(idx_ent), cmp_idx_ent);
    uint64_t d = 0, sum = 0;
    for (uint64_t i = 0; i < m; i++) {
        if (i && e[i].hi == e[i - 1].hi && e[i].lo == e[i - 1].lo) continue;
        out_hi[d] = e[i].hi; out_lo[d] = e[i].lo; d++;
        sum += e[i].lo;
    }
    *checksum = sum;
    free(e); free(node); free(tmp);
    return d;
}
