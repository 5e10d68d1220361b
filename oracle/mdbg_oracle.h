/*
 * mdbg_oracle.h -- CPU restatement of metaMDBG's minimizer + k-min-mer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker for the HIP kernels in
 * metamdbg_amd/csrc.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may link or call it.  The product path never does.
 *
 * Parity status: PINNED.  Every function here is checked (tests/test_oracle_golden.py,
 * tests/golden/) against output of the reference's own code compiled from
 * /root/reference by oracle/Makefile into oracle/_ref/ (see oracle/ref_driver.cpp),
 * and against the known-answer scalars in SURVEY.md section 8(c).
 *
 * Each function cites the reference file:line it restates (paths relative to
 * /root/reference/src).  Plain C99, single-threaded, no dependencies but libm.
 */
#ifndef MDBG_ORACLE_H
#define MDBG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- hashing ----------------------------------------------------------- */

/* Standard MurmurHash3_x64_128 (utils/MurmurHash3.cpp:328-405). out[0]=h1, out[1]=h2. */
void orc_murmur3_x64_128(const void *key, int len, uint32_t seed, uint64_t out[2]);

/* The reference's modified variant returning h1 only (utils/MurmurHash3.cpp:246-325),
 * specialised to what MinimizerParser calls: 8-byte key, seed 42 (utils/kmer/Kmer.hpp:1421). */
uint64_t orc_kmer_hash(uint64_t kmer_value);

/* Smallest u64 T such that  (double)h < (double)density_f32 * 2^64  <=>  h < T
 * (utils/kmer/Kmer.hpp:1357-1358 builds the bound, :1434 compares u64 against double). */
uint64_t orc_density_threshold(float density);

/* ---- base space (readSelection) ---------------------------------------- */

/* EncoderRLE::execute (Commons.hpp:4163-4203).
 * hpc != 0: homopolymer-compress; out_seq gets L' chars, rle_pos gets L'+1 entries
 *           (start of each run, then `len` as sentinel).  Empty input emits '#'.
 * hpc == 0: copy; rle_pos[i] = i for i < len (no sentinel).
 * Buffers must hold len+1 entries.  Returns L'. */
size_t orc_hpc_encode(const char *seq, size_t len, int hpc, char *out_seq, uint64_t *rle_pos);

/* KmerModel::iterate (utils/kmer/Kmer.hpp:531-611): rolling canonical K-mers.
 * kmers[i] = min(fwd, revcomp), dirs[i] = 0 if fwd < rev else 1 (tie -> 1, :427);
 * a k-mer overlapping a char with bit 3 set (N/n, :462) gets value UINT64_MAX.
 * Returns the number of k-mers (0 if len < K). */
size_t orc_kmer_iterate(const char *seq, size_t len, unsigned K, uint64_t *kmers, uint8_t *dirs);

/* MinimizerParser::parse (utils/kmer/Kmer.hpp:1373-1456).  Positions 1..nK-2 only;
 * select iff hash < bound; drop if (u32)value is in `repetitive` (n_rep entries, may be 0).
 * Outputs must hold max(len,1) entries.  Returns the number selected. */
size_t orc_minimizer_parse(const char *seq, size_t len, unsigned K, float density,
                           const uint32_t *repetitive, size_t n_rep,
                           uint32_t *out_min, uint32_t *out_pos, uint8_t *out_dir);
/* Same with MinimizerParser::_trimBps explicit: 1 is the constructor's default (Kmer.hpp:1362); GenerateGfa's
 * LoadUnitigsFunctor sets 0 (graph/GenerateGfa.hpp:366) so the first and last l-mer of a unitig can be selected. */
size_t orc_minimizer_parse_trim(const char *seq, size_t len, unsigned K, float density,
                                const uint32_t *repetitive, size_t n_rep, size_t trim_bps,
                                uint32_t *out_min, uint32_t *out_pos, uint8_t *out_dir);

/* ReadSelectionFunctor::computeSequenceComplexity (readSelection/ReadSelection.hpp:1171-1228)
 * with w=64, step=32 on the ORIGINAL (not HPC) sequence.  NaN when there is no full window.
 * 3-mers overlapping an N index kmerCounts[-1] in the reference (UB, it aborts); here N counts
 * with its 2-bit code (c>>1)&3 like any other character. */
double orc_sequence_complexity(const char *seq, size_t len);

/* Mean read quality (readSelection/ReadSelection.hpp:870-879; table :101-104;
 * Utils::transformQuality Commons.hpp:2338-2341).  qual may be NULL/len 0 -> NaN. */
float orc_mean_read_quality(const char *qual, size_t len);

typedef struct {
    uint32_t n;          /* number of minimizers (after filters) */
    uint32_t *minimizers;
    uint32_t *pos;       /* position in HPC coordinates */
    uint8_t  *dir;
    uint8_t  *qual;      /* per-minimizer min quality, or 1 when no qualities */
    float    mean_quality;
    uint32_t read_length; /* original length */
    uint32_t hpc_length;
    int low_complexity;
    int low_quality;
} orc_read_record;

typedef struct {
    unsigned K;              /* minimizer size l (<=16) */
    float density;           /* assembly density, e.g. 0.005f */
    int hpc;                 /* homopolymer compression on (HiFi) */
    float min_read_quality;  /* --min-read-quality */
    const uint32_t *repetitive; size_t n_rep;
} orc_scan_params;

/* ReadSelectionFunctor::operator() (readSelection/ReadSelection.hpp:669-1158): one read ->
 * one record.  qual == NULL for FASTA.  Arrays in rec are malloc'd; free with orc_read_record_free. */
void orc_read_selection(const char *seq, const char *qual, size_t len,
                        const orc_scan_params *p, orc_read_record *rec);
void orc_read_record_free(orc_read_record *rec);

/* ReadCorrection::ReadSelectionFunctor::operator() up to its record sink (readSelection/ReadCorrection.hpp:2318-2372):
 * HPC + MinimizerParser at p->density (the correction density), NO complexity / quality filters, mean quality 0,
 * per-minimizer min quality over the INCLUSIVE span [rle[pos], rle[pos+K-1]] (:2340, getMinQuality :2467-2481). */
void orc_correction_scan(const char *seq, const char *qual, size_t len,
                         const orc_scan_params *p, orc_read_record *rec);

/* Utils::applyDensityThreshold (Commons.hpp:2507-2550): keep[i] = 1 iff Murmur(minimizer i) < density * 2^64.
 * Returns the number kept. */
size_t orc_apply_density_threshold(const uint32_t *minimizers, size_t n, float density, uint8_t *keep);

/* Serialise a record exactly as ReadSelection::writeRead does (ReadSelection.hpp:415-467).
 * Returns bytes written (13 + 10 n). buf must hold that many. */
size_t orc_write_read_record(const orc_read_record *rec, uint8_t *buf);

/* Utils::computeN50 (Commons.hpp:2291-2322) / computeMeanLength (:2324-2336). */
uint32_t orc_compute_n50(const uint32_t *lengths, size_t n);
uint32_t orc_compute_mean_length(const uint32_t *lengths, size_t n);
/* Commons::computeLastK (Commons.hpp:1726-1741). */
int orc_compute_last_k(float density, size_t n50, size_t first_k, size_t max_k);

/* Commons::purgePalindrome (Commons.hpp:1617-1723). In place; returns the new length. */
size_t orc_purge_palindrome(uint32_t *minimizers, size_t n, size_t first_k, size_t last_k);

/* ---- minimizer space (graph) ------------------------------------------- */

/* KmerVec::normalize (Commons.hpp:886-916): writes the canonical orientation of v[0..k) to out;
 * returns isReversed (1 when reverse is strictly smaller OR the vector is a palindrome). */
int orc_kminmer_normalize(const uint32_t *v, unsigned k, uint32_t *out);

/* KmerVec::hash128 (Commons.hpp:941-969): Murmur3 x64-128 seed 0 over k*4 bytes.
 * hi = out[0], lo = out[1]; on disk (little-endian u128) lo comes first. */
void orc_kminmer_hash128(const uint32_t *v, unsigned k, uint64_t *hi, uint64_t *lo);

typedef struct {
    uint64_t n;         /* records */
    unsigned k;
    uint32_t *vecs;     /* n*k u32, canonical vectors (NULL when not produced, k>=firstK+2) */
    uint64_t *hash_lo;  /* n */
    uint64_t *hash_hi;  /* n */
    uint32_t *abundance;/* n */
    uint64_t n_solid;   /* first n_solid records are solid, the rest rescued (first pass) */
} orc_kminmer_table;
void orc_kminmer_table_free(orc_kminmer_table *t);

/* First pass (k = firstK): KminmerCounter (graph/CreateMdbg.hpp:3591-3883) followed by
 * rescueKminmers (graph/CreateMdbg.hpp:4514-4640) when min_abundance <= 1.
 * Reads are given CSR style: minimizers[offsets[r] .. offsets[r+1]).
 * Record order: solid records sorted by canonical vector, then rescued ones in read order
 * (the reference's order is unspecified; compare as multisets). */
void orc_kminmer_count_first(const uint32_t *minimizers, const uint64_t *offsets, uint64_t n_reads,
                             unsigned k, uint32_t min_abundance, orc_kminmer_table *out);

/* Previous-iteration abundance lookup used at k > firstK: sorted array of (hi,lo)->abundance. */
typedef struct { uint64_t n; uint64_t *hi; uint64_t *lo; uint32_t *abundance; } orc_abundance_map;
/* Build from 20-byte records as CreateMdbg::loadRefinedAbundances does for
 * kminmerData_abundance_prev.txt (graph/CreateMdbg.cpp:3436-3447): abundance==1 skipped. */
void orc_abundance_map_from_records(const uint8_t *records, uint64_t n_records, orc_abundance_map *m);
/* Overlay one unitig's refined abundance (graph/CreateMdbg.cpp:3466-3507): every kprev-min-mer of the
 * unitig gets `a`, or 0 if a==1 and the key exists.  Call orc_abundance_map_finish afterwards. */
void orc_abundance_map_overlay(orc_abundance_map *m, const uint32_t *unitig, uint32_t n, unsigned kprev, uint32_t a);
void orc_abundance_map_finish(orc_abundance_map *m);
int  orc_abundance_map_get(const orc_abundance_map *m, uint64_t hi, uint64_t lo, uint32_t *abundance);
void orc_abundance_map_free(orc_abundance_map *m);

/* k = firstK+1: KminmerCounter over reads (+ unitigs) with getRefinedAbundance
 * (graph/CreateMdbg.hpp:3933-4005): abundance = min over the vector's (k-1)-min-mers of prev
 * (missing or 0 => 1); keep > 1. */
void orc_kminmer_count_refined(const uint32_t *minimizers, const uint64_t *offsets, uint64_t n_seqs,
                               unsigned k, const orc_abundance_map *prev, orc_kminmer_table *out);

/* k >= firstK+2: IndexKminmerFunctor (graph/CreateMdbg.hpp:1240-1265, :1450-1459): per sequence,
 * prev[i] = abundance of the i-th (k-1)-min-mer (missing => 1); k-min-mer i gets min(prev[i],prev[i+1]);
 * inserted if > 1.  out->vecs is NULL; records sorted by (hi,lo). */
void orc_kminmer_index(const uint32_t *minimizers, const uint64_t *offsets, uint64_t n_seqs,
                       unsigned k, const orc_abundance_map *prev, orc_kminmer_table *out);

/* The small-contig branch of IndexKminmerFunctor (graph/CreateMdbg.hpp:1330-1352), taken in the unitig pass when
 * k > 8: a unitig with no k-min-mer whose getAbundance(0, prevAbundances) (:988-1010) exceeds 1 is written to
 * smallContigs_k<k>.bin as `u32 n; u8 circular; u32 m[n]` instead of being indexed.  prevAbundances holds one entry per
 * kprev-min-mer of the unitig (missing => 1, :1240-1265): with one entry the value is that entry, with more it is
 * min(prev[0], prev[1]); with none (n < kprev) the reference reads prev[0] of an empty vector (undefined), restated here
 * as "not a small contig".  flags[u] = 1 where the unitig is written.  The caller applies the k > 8 condition. */
void orc_small_contigs(const uint32_t *minimizers, const uint64_t *offsets, uint64_t n_seqs, unsigned k, unsigned kprev,
                       const orc_abundance_map *prev, uint8_t *flags);

/* EdgeIndexer (graph/CreateMdbg.hpp:4010-4230): the distinct identities of the normalised
 * (k-1)-prefix and (k-1)-suffix of every k-min-mer vector (the content of edges.bin, sorted by (hi,lo));
 * *checksum = sum of the identities truncated to u64, as indexEdges logs it (graph/CreateMdbg.cpp:1184).
 * out_hi / out_lo must hold 2 n entries; returns the number of distinct edges. */
uint64_t orc_edge_index(const uint32_t *vecs, uint64_t n, unsigned k, uint64_t *out_hi, uint64_t *out_lo, uint64_t *checksum);

/* UnitigEdgeIndexer (graph/CreateMdbg.hpp:4234-4512): per unitig (minimizer sequence), the first and the last k-min-mer
 * (only one when they are the same vector, :4378), each normalised (:4366, :4380), then the identities of its
 * normalised (k-1)-prefix and -suffix (:4391-4402); de-duplicated, sorted by (hi,lo).  Sequences shorter than k have no
 * k-min-mer.  out_hi / out_lo must hold 4 n_seqs entries; returns the number of distinct edges. */
uint64_t orc_unitig_edge_index(const uint32_t *minimizers, const uint64_t *offsets, uint64_t n_seqs, unsigned k,
                               uint64_t *out_hi, uint64_t *out_lo, uint64_t *checksum);

#ifdef __cplusplus
}
#endif
#endif
