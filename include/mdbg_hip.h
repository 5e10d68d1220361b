/*
 * mdbg_hip.h -- C ABI of libmdbg_hip.so: metaMDBG's read -> minimizer -> k-min-mer hot path
 * as hand-written HIP kernels for MI355X (gfx950).
 *
 * The reference has no FFI: its boundary is two child processes (`readSelection`, `graph`)
 * talking through files.  Each entry point below replaces the
 * compute inside one reference function, cited as file:line under /root/reference/src, and is
 * what a maintainer would call from that function (binding stubs: INTEGRATION.md).
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative
 * MDBG_E* code, with a message available from mdbg_last_error(); objects are opaque handles
 * owned by the library and released with the matching *_free; a context is bound to one HIP
 * device and one stream and is not thread-safe (use one context per host thread / stream,
 * as the reference uses one functor copy per OpenMP thread: Commons.hpp:5846-5914).
 * Several contexts on one device may be driven concurrently from several threads; that is the intended way to keep
 * two or three batches in flight (the table kernels of one batch overlap the scan of the next -- bench.py).
 * mdbg_scan calls on the same device take turns: one scan kernel runs at a time, also across processes (an advisory file lock named
 * after the device's PCI address; MDBG_SCAN_NO_XPROC_LOCK=1 keeps the rule inside the process).
 * There is no CPU fallback: without a usable GPU every call fails with MDBG_ENODEV.
 *
 * Environment (read by the library): MDBG_SCAN_READS_PER_WAVE, MDBG_TABLE_BLOCKS_PER_CU -- defaults of the options of
 * mdbg_set_option; MDBG_TRACE -- one line per purge with the number of suspect reads.
 */
#ifndef MDBG_HIP_H
#define MDBG_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDBG_OK        0
#define MDBG_EINVAL   -1   /* bad argument */
#define MDBG_ENODEV   -2   /* no HIP device / wrong architecture */
#define MDBG_ENOMEM   -3   /* device or host allocation failed */
#define MDBG_EHIP     -4   /* HIP runtime error (see mdbg_last_error) */
#define MDBG_ERANGE   -5   /* size exceeds an internal limit (see message) */
#define MDBG_EPEER    -6   /* collective calls: another rank reported a failure; this rank's data is intact, nothing was exchanged */

typedef struct mdbg_ctx mdbg_ctx;
typedef struct mdbg_reads mdbg_reads;             /* base-space reads resident in HBM, 2-bit packed */
typedef struct mdbg_minimizers mdbg_minimizers;   /* minimizer-space reads resident in HBM (CSR) */
typedef struct mdbg_table mdbg_table;             /* k-min-mer table resident in HBM */

/* ---- context ------------------------------------------------------------------------- */
int  mdbg_create(int device, mdbg_ctx **ctx);
void mdbg_destroy(mdbg_ctx *ctx);
const char *mdbg_last_error(const mdbg_ctx *ctx);      /* ctx may be NULL: last creation error of the calling thread */
int  mdbg_synchronize(mdbg_ctx *ctx);
/* Queues a kernel of one wave that idles for `microseconds` on the context's stream and returns at once.  HIP multiplexes streams
 * onto a few hardware queues; two contexts whose streams share one never overlap their kernels.  A caller that keeps several
 * batches in flight on one device finds out by spinning on two contexts at once: different queues finish together, a shared
 * queue takes twice as long (bench.py's overlap probe; no reference analogue -- the reference's threads share one CPU scheduler). */
int  mdbg_stream_spin(mdbg_ctx *ctx, uint32_t microseconds);
void *mdbg_stream(mdbg_ctx *ctx);                       /* the hipStream_t every launch goes to */
int  mdbg_device_info(mdbg_ctx *ctx, char *arch, size_t arch_len, int *n_cu, uint64_t *hbm_bytes);
int  mdbg_device_clock_khz(mdbg_ctx *ctx, int *clock_khz);   /* peak engine clock (hipDeviceProp_t::clockRate) */
/* Per-context tuning, value <= 0 restores the default:
 *   "table_blocks_per_cu"   resident blocks per CU of the kernels that walk every k-min-mer instance (default: unlimited;
 *                           1..3 when several contexts share a device, so that they do not displace another context's scan)
 *   "table_grid_blocks"     g > 0: those kernels run as exactly g workgroups (grid-stride), e.g. half a block per CU beside a scan;
 *                           0 (default): "table_blocks_per_cu" decides
 *   "table_cu_count"        c > 0: every kernel of the context except the block-structured scan kernel is confined to c compute
 *                           units (spread over the XCDs), the scan kernel runs on a stream of its own over all of them; for
 *                           several contexts in flight on one device (the other batches' table kernels then displace a running
 *                           scan on those CUs only).  Call it between steps: it synchronises and re-creates the stream, so a
 *                           handle obtained from mdbg_stream before is dead.  Default 0: one unconfined stream
 *   "pool_cache_percent"    the context keeps freed device blocks for reuse up to this share of the device memory (default 55:
 *                           several contexts share a device; a context that has the device to itself may take 90)
 *   "pool_trim"             any value: the blocks kept for reuse are given back to the device now
 *   "scan_reads_per_wave"   reads a scan wave processes before it retires (default 2)
 *   "scan_candidate_slack"  tests only: the block-structured scan records candidate positions by the upper half of the
 *                           hash and confirms each with the full hash; a read with a false candidate is re-run.  False
 *                           candidates occur about once in 2^31 positions; this widens the test (units of 2^32 of the
 *                           hash range) so that the re-run path can be exercised.  Default 0; results never depend on it
 *   "index_tuning"          the passes above firstK over the one-slot tables (bits; default 19; negative: the default): 1 = a slot's key and
 *                           value fetched in one trip, 2 = the insert first looks at a window's home slot with plain loads (a key found
 *                           there is done without an atomic), 4 = two windows of a lane in flight (measured: no gain), 8 = look-up and
 *                           insert in one kernel (measured: slower), 16 = both slots of a key's home sector fetched at once (look-up
 *                           and the insert's first look), 32 = 32 lanes a sequence instead of 16, 128 = always insert first and look the
 *                           previous table up only for the k-windows not seen in the new one, 256 = never (neither: a sample of the
 *                           sequences decides per pass -- insert first when fewer than 0.45 of the windows are never inserted);
 *                           0 = the kernels of rounds 1 - 4.  Results never depend on it
 *   "keep_index_table"      1 (default) = the hash table an index pass filled stays with its result as the look-up structure the next
 *                           pass reads; 0 = it is dropped and built again from the rows on first use (rounds 1 - 5)
 *   "index_table_form"      0 = those passes over bucket tables (three keys per 64-byte sector: a third of the bytes, measured no
 *                           faster), 1 or negative = one 32-byte slot per key (default)
 *   "refined_form"          0 = k = firstK+1 done like an index pass (a look-up per (k-1)-window; measured slower), 1 or negative =
 *                           every distinct key first, then two look-ups per key (default)
 *   "scan_quality_stream"   FASTQ: 1 (default) = the per-read quality sums run beside the scan kernel on a side stream, 0 = in front of it
 *   "test_exchange_fail_phase"  tests only: this rank fails inside the next mdbg_shard_exchange before the counts travel (1),
 *                           when the receive buffers are allocated (2) or in the owner's reduction (3); one-shot
 *   "test_corrupt_replies"  tests only: the next exchange hands back one reply with a wrong count; one-shot
 * The environment variables MDBG_TABLE_BLOCKS_PER_CU / MDBG_SCAN_READS_PER_WAVE set the defaults at mdbg_create. */
int  mdbg_set_option(mdbg_ctx *ctx, const char *name, int64_t value);

/* Profiling aid: accumulated HIP-event time (ms) and launch count of the kernel named
 * `kernel` ("scan", "kminmer_insert", ...) since the last reset; events are recorded on
 * the context's stream only while timing is enabled. */
int  mdbg_timing_enable(mdbg_ctx *ctx, int on);
int  mdbg_timing_reset(mdbg_ctx *ctx);
int  mdbg_timing_get(mdbg_ctx *ctx, const char *kernel, double *ms_total, uint64_t *launches);

/* ---- base-space reads ---------------------------------------------------------------- */
/* Replaces the per-read copy in ReadParserParallel::parse (Commons.hpp:5868-5905): the caller
 * hands a batch of parsed reads (ASCII, concatenated; read r = bases[offsets[r] .. offsets[r+1]))
 * and optional phred+33 qualities with the same offsets (NULL for FASTA).  The batch is packed
 * to 2 bits/base (code (c>>1)&3, utils/kmer/Kmer.hpp:462) on the device.  Characters with
 * bit 3 set (N, n) are kept in a side bitmask, and so is every place where two neighbouring characters
 * differ although their codes agree ("aA", IUPAC letters): the reference's homopolymer compression compares
 * raw characters (Commons.hpp:4177-4178), and so does this path -- any byte string gives the reference's
 * minimizers.  Batches of plain upper-case ACGT carry neither mask and take the fast kernel variant. */
int  mdbg_reads_from_ascii(mdbg_ctx *ctx, const char *bases, const char *quals, const uint64_t *offsets,
                           uint32_t n_reads, mdbg_reads **out);
/* Already packed on the host: words[] holds 32 bases per u64 (base i of a word at bits [2i,2i+2)),
 * read r occupies words[word_offsets[r] .. word_offsets[r+1]) and starts on an even word.  The packed form
 * cannot say more than the code: use it for reads made of A, C, G, T only (the host feed does). */
int  mdbg_reads_from_packed(mdbg_ctx *ctx, const uint64_t *words, const uint64_t *word_offsets,
                            const uint32_t *lengths, uint32_t n_reads, mdbg_reads **out);
/* The same without waiting for the copies: they are queued on an upload stream of the context (a copy engine: they run beside
 * the kernels of the context's own stream) and the call returns (offsets and lengths are copied before it does; the words are
 * what travels behind the caller's back).  `words` must stay untouched until
 * mdbg_reads_wait returns (or the reads are freed); page-locked memory (mdbg_host_alloc) is what makes the copy asynchronous.
 * Calls that consume the reads order themselves after the upload: mdbg_scan on the device, without a host wait -- so a feeder
 * uploads batch i+1 while batch i is scanned, inside one context: the replacement of the reference's one-reader critical
 * section (Commons.hpp:5868-5905) is then bound by the link, not by link + kernels. */
int  mdbg_reads_from_packed_async(mdbg_ctx *ctx, const uint64_t *words, const uint64_t *word_offsets, const uint32_t *lengths,
                                  uint32_t n_reads, mdbg_reads **out);
/* The few reads of a PACKED batch (mdbg_reads_from_packed / _async) that hold characters other than upper-case A, C, G, T -- an N,
 * a soft-masked stretch, an IUPAC letter: the 2-bit words alone lose what the reference still sees (invalid k-mers, utils/kmer/
 * Kmer.hpp:574-580; a change of character starts a homopolymer run, Commons.hpp:4177-4178).  The caller hands those reads again
 * as characters (ascending indices; read read_index[i] = ascii[ascii_offsets[i] .. ascii_offsets[i+1])) and the side masks of the
 * batch are derived from them on the device, exactly as mdbg_reads_from_ascii derives them for every read.  mdbg_scan then keeps
 * the batch on its fast kernel and sends only the marked reads through the general one.  Once per batch, before any scan. */
int  mdbg_reads_mark_ascii(mdbg_ctx *ctx, mdbg_reads *r, const uint32_t *read_index, uint32_t n_listed, const char *ascii,
                           const uint64_t *ascii_offsets);
/* Qualities for reads made by mdbg_reads_from_packed_async, queued behind their words (same rules: `quals` stays untouched until
 * mdbg_reads_wait; a read has one quality per base, checked at once against the lengths given at upload). */
int  mdbg_reads_attach_qualities_async(mdbg_ctx *ctx, mdbg_reads *r, const char *quals, const uint64_t *offsets);
int  mdbg_reads_wait(mdbg_ctx *ctx, const mdbg_reads *r);
/* Phred+33 qualities (Read::_qual) for reads made by mdbg_reads_from_packed: read r = quals[offsets[r] .. offsets[r+1]),
 * offsets[r+1] - offsets[r] must equal its length.  Once per reads object. */
int  mdbg_reads_attach_qualities(mdbg_ctx *ctx, mdbg_reads *reads, const char *quals, const uint64_t *offsets);
/* Seeded synthetic read set generated directly in HBM (bench/test harness; same generator as
 * metamdbg_amd/synth.py).  thresholds[s] = cumulative species weight as u64.  Per read position one u64 draw e decides:
 * e < ins_threshold an inserted base, then del_threshold a skipped genome base, then sub_threshold a substitution
 * (HiFi 0.1 % substitutions; ONT R10 1 % + 0.5 % + 0.5 %).  window = genome bases set aside per read
 * (>= read_len; ignored without indels). */
int  mdbg_reads_synthetic(mdbg_ctx *ctx, uint64_t seed, uint32_t n_reads, uint32_t read_len,
                          uint64_t first_read, const uint64_t *species_len,
                          const uint64_t *species_threshold, uint32_t n_species,
                          uint64_t sub_threshold, uint64_t ins_threshold, uint64_t del_threshold, uint32_t window,
                          int with_quality, mdbg_reads **out);
int  mdbg_reads_info(const mdbg_reads *r, uint32_t *n_reads, uint64_t *n_bases, uint64_t *n_words);
/* Copy read `index` back as ASCII (buffer of at least its length; quals may be NULL). */
int  mdbg_reads_get(mdbg_ctx *ctx, const mdbg_reads *r, uint32_t index, char *bases, char *quals, uint32_t *length);
/* Bulk export of reads [first, first+count) as concatenated ASCII: offsets gets count+1 entries
 * (relative to bases[0]); bases must hold the sum of their lengths (use mdbg_reads_info / a first
 * call with bases == NULL to size it: *n_bytes is always set). */
int  mdbg_reads_export_ascii(mdbg_ctx *ctx, const mdbg_reads *r, uint32_t first, uint32_t count,
                             char *bases, uint64_t *offsets, uint64_t *n_bytes);
/* The phred+33 bytes of reads [first, first+count), concatenated in read order (same offsets as mdbg_reads_export_ascii gives
 * for the bases: a read has one quality per base); *n_bytes is always set, quals may be NULL to size the buffer. */
int  mdbg_reads_export_qualities(mdbg_ctx *ctx, const mdbg_reads *r, uint32_t first, uint32_t count, char *quals, uint64_t *n_bytes);
void mdbg_reads_free(mdbg_reads *r);

/* ---- reads -> minimizers --------------------------------------------------------------- */
typedef struct {
    uint32_t minimizer_size;      /* l, <= 16            (Parameters::_minimizerSize) */
    float    density;             /* f32 as stored in parameters.gz (Parameters::_minimizerDensity_assembly) */
    int32_t  hpc;                 /* Parameters::_useHomopolymerCompression */
    float    min_read_quality;    /* --min-read-quality   (ReadSelection.hpp:901) */
    const uint32_t *repetitive;   /* host array: repetitiveMinimizers.bin contents, may be NULL */
    uint32_t n_repetitive;
    int32_t  apply_read_filters;  /* 1: complexity + quality filters of ReadSelectionFunctor (ReadSelection.hpp:890-915);
                                     0: bare MinimizerParser::parse as CountMinimizerFunctor uses it (:599-611) */
    int32_t  quality_window;      /* per-minimizer minimum quality over the ORIGINAL coordinates
                                     0: [rle[pos], rle[pos+l])    ReadSelectionFunctor   (ReadSelection.hpp:1135, :1302-1320)
                                     1: [rle[pos], rle[pos+l-1]]  the correction scan, ReadCorrection::ReadSelectionFunctor
                                                                  (ReadCorrection.hpp:2340, :2467-2481); used with
                                                                  density = _minimizerDensity_correction, apply_read_filters = 0 */
    int32_t  no_end_trim;         /* 0: MinimizerParser's default _trimBps = 1, the first and last l-mer of a read are never
                                     selected (utils/kmer/Kmer.hpp:1362, :1395); 1: _trimBps = 0 as GenerateGfa's
                                     LoadUnitigsFunctor sets it for unitig sequences (graph/GenerateGfa.hpp:366) */
    int32_t  ignore_qualities;    /* 1: the read set's qualities are not looked at, as if it had none (every minimizer's quality
                                     is 1, the mean read quality NaN): CountMinimizerFunctor's census of minimizer values
                                     (ReadSelection.hpp:565-625) over reads that are resident with their qualities */
} mdbg_scan_params;

/* Replaces, for a whole batch, EncoderRLE::execute + MinimizerParser::parse + complexity /
 * quality filters + per-minimizer min quality of ReadSelectionFunctor::operator()
 * (readSelection/ReadSelection.hpp:669-1158; utils/kmer/Kmer.hpp:1373-1456; Commons.hpp:4163-4203).
 * Output order = read order, minimizers of a read in position order. */
int  mdbg_scan(mdbg_ctx *ctx, const mdbg_reads *reads, const mdbg_scan_params *params, mdbg_minimizers **out);

int  mdbg_minimizers_info(const mdbg_minimizers *m, uint32_t *n_reads, uint64_t *n_minimizers);
/* Host copies; any pointer may be NULL.  offsets has n_reads+1 entries; per-read arrays n_reads.
 * mean_quality is the f32 of ReadSelection.hpp:878-879 (NaN without qualities). */
int  mdbg_minimizers_to_host(mdbg_ctx *ctx, const mdbg_minimizers *m, uint64_t *offsets,
                             uint32_t *minimizers, uint32_t *positions, uint8_t *directions, uint8_t *qualities,
                             uint32_t *read_lengths, float *mean_quality, uint8_t *read_flags);
#define MDBG_READ_LOW_COMPLEXITY 1u
#define MDBG_READ_LOW_QUALITY    2u
/* Upload minimizer-space sequences (read_data_corrected.txt / unitig_data.txt contents as CSR). */
int  mdbg_minimizers_from_host(mdbg_ctx *ctx, const uint32_t *minimizers, const uint64_t *offsets,
                               uint32_t n_reads, mdbg_minimizers **out);
/* Device pointers for zero-copy consumers (torch.from_dlpack-style wrapping in the harness). */
int  mdbg_minimizers_device_ptrs(const mdbg_minimizers *m, const uint64_t **d_offsets, const uint32_t **d_minimizers);
void mdbg_minimizers_free(mdbg_minimizers *m);
/* The reads of `parts[0]`, then `parts[1]`, ... as one set, appended on the device (CSR offsets rebased; positions, directions,
 * qualities and the per-read fields follow when every part is a scan output).  For read sets scanned in several resident
 * pieces -- 10 M ONT reads with their qualities are 250 GB, more than fits beside their outputs -- whose palindrome purge and
 * k-min-mer table are over ALL the reads: the reference counts k-min-mers over the whole read_data_corrected.txt however the
 * reads were parsed (graph/CreateMdbg.cpp:290-328).  The parts stay valid and are not modified. */
int  mdbg_minimizers_concat(mdbg_ctx *ctx, const mdbg_minimizers *const *parts, uint32_t n_parts, mdbg_minimizers **out);
/* The other way round: reads [first_read, first_read + n_reads) of `in` as a set of their own (values and offsets; a device copy).
 * What a rank of a sharded job holds of a read set (contiguous read ranges); a job that checks the table of a
 * whole set against the union of its shards cuts the shards with it instead of scanning the reads again. */
int  mdbg_minimizers_slice(mdbg_ctx *ctx, const mdbg_minimizers *in, uint32_t first_read, uint32_t n_reads, mdbg_minimizers **out);

/* Replaces Utils::applyDensityThreshold over every read (Commons.hpp:2507-2550; callers ReadCorrection.hpp:6385, :6435,
 * Commons.hpp:2563 getLowDensityMinimizerRead, :7252-7742 the minimizer-read parsers): keeps the minimizers whose
 * selection hash is below `density` -- the down-sampling from the correction density to the assembly density.
 * Positions / directions / qualities / per-read fields follow when `in` carries them. */
int  mdbg_apply_density_threshold(mdbg_ctx *ctx, const mdbg_minimizers *in, float density, mdbg_minimizers **out);

/* Replaces Commons::purgePalindrome over every read (Commons.hpp:1617-1723, driven by
 * ReadSelection::purgePalindromes, ReadSelection.hpp:1374-1431). */
int  mdbg_purge_palindromes(mdbg_ctx *ctx, const mdbg_minimizers *in, uint32_t first_k, uint32_t last_k,
                            mdbg_minimizers **out);

/* Replaces CountMinimizerFunctor + the global map of determineRepetitiveMinimizers
 * (ReadSelection.hpp:497-625): counts every minimizer value of `m` on the device and returns the
 * top max(1, floor(1e-5 * distinct)) by count (ties broken by smaller value; the reference's
 * tie order is std::sort-unstable).  out must hold *n_out entries on input. */
int  mdbg_repetitive_minimizers(mdbg_ctx *ctx, const mdbg_minimizers *m, uint32_t *out, uint32_t *n_out);
/* The same census fed batch by batch -- the reads of determineRepetitiveMinimizers arrive in chunks (the first
 * 1 000 001 reads of every input file, ReadSelection.hpp:497-561): mdbg_census_add counts the values of one batch's
 * minimizers on the device (nothing travels; one count per possible value -- 4 bytes x the power of two above the largest
 * value seen: 4 GiB at l = 15, 16 GiB at l = 16 --, one add per minimizer), mdbg_census_top is the selection above over
 * everything added so far.
 * mdbg_repetitive_minimizers(m) = create, add(m), top, free. */
typedef struct mdbg_census mdbg_census;
int  mdbg_census_create(mdbg_ctx *ctx, mdbg_census **out);
int  mdbg_census_add(mdbg_ctx *ctx, mdbg_census *census, const mdbg_minimizers *m);
int  mdbg_census_top(mdbg_ctx *ctx, const mdbg_census *census, uint32_t *out, uint32_t *n_out);
void mdbg_census_free(mdbg_census *census);

/* ---- minimizers -> k-min-mer table ----------------------------------------------------- */
/* First pass, k = firstK: KminmerCounter::execute + rescueKminmers
 * (graph/CreateMdbg.hpp:3591-3883, :4514-4640; graph/CreateMdbg.cpp:290-328). */
int  mdbg_kminmer_count_first(mdbg_ctx *ctx, const mdbg_minimizers *reads, uint32_t k, uint32_t min_abundance,
                              mdbg_table **out);
/* Previous-iteration abundances for k > firstK (CreateMdbg::loadRefinedAbundances,
 * graph/CreateMdbg.cpp:3401-3709): from kminmerData_abundance_prev.txt records (lo,hi,abundance;
 * abundance==1 skipped) ... */
int  mdbg_prev_from_records(mdbg_ctx *ctx, const uint8_t *records20, uint64_t n_records, mdbg_table **out);
/* ... then overlaid with the refined abundance of each unitig of unitigGraph_prev.nodes.bin
 * (unitigs as CSR; abundance[u] per unitig, 0xFFFFFFFF = unitig has no refined abundance and is skipped). */
int  mdbg_prev_overlay_unitigs(mdbg_ctx *ctx, mdbg_table *prev, const mdbg_minimizers *unitigs,
                               const uint32_t *abundance, uint32_t k_prev);
/* k = firstK+1: KminmerCounter with getRefinedAbundance (graph/CreateMdbg.hpp:3933-4005) over
 * reads and (optionally, may be NULL) unitig_data.txt sequences. */
int  mdbg_kminmer_count_refined(mdbg_ctx *ctx, const mdbg_minimizers *reads, const mdbg_minimizers *unitigs,
                                uint32_t k, const mdbg_table *prev, mdbg_table **out);
/* k >= firstK+2: IndexKminmerFunctor (graph/CreateMdbg.hpp:1240-1265, :1450-1459) over reads and unitigs. */
int  mdbg_kminmer_index(mdbg_ctx *ctx, const mdbg_minimizers *reads, const mdbg_minimizers *unitigs,
                        uint32_t k, const mdbg_table *prev, mdbg_table **out);

/* Replaces the small-contig branch of IndexKminmerFunctor (graph/CreateMdbg.hpp:1330-1352), taken in the unitig pass of
 * `graph` when k > 8: a unitig with no k-min-mer whose getAbundance(0, prevAbundances) (:988-1010, prevAbundances over
 * its k_prev-min-mers, missing => 1, :1240-1265) exceeds 1 is appended to smallContigs/smallContigs_k<k>.bin
 * (`u32 n; u8 circular; u32 m[n]`) instead of being indexed.  flags (host, one per unitig) is set to 1 for those
 * unitigs; the caller applies the k > 8 condition and writes the records.  Unitigs shorter than k_prev, for which the
 * reference reads an empty vector (undefined), are never flagged. */
int  mdbg_small_contigs(mdbg_ctx *ctx, const mdbg_minimizers *unitigs, uint32_t k, uint32_t k_prev, const mdbg_table *prev,
                        uint8_t *flags);

/* n_records = rows of kminmerData_abundance.txt; n_solid of them are solid (the rest rescued,
 * abundance 1); has_vectors = whether kminmerData_min.txt rows exist (k <= firstK+1). */
int  mdbg_table_info(const mdbg_table *t, uint32_t *k, uint64_t *n_records, uint64_t *n_solid, int *has_vectors);
/* Host copies in file layout: records20 = n_records x {u64 lo, u64 hi, u32 abundance} packed
 * (MDBG::writeKminmerAbundance, Commons.hpp:4463-4472); vectors = n_records x k u32
 * (MDBG::writeKminmer, Commons.hpp:4429-4446), same row order.  Either may be NULL. */
int  mdbg_table_to_host(mdbg_ctx *ctx, const mdbg_table *t, uint8_t *records20, uint32_t *vectors);
/* Rows [first, first + count) only, same layout: a writer streams a large table (the ONT preset's first pass leaves 75 M records per
 * 20 Gbp, most of them rescued singletons) to its files through two page-locked buffers instead of holding it all on the host. */
int  mdbg_table_to_host_range(mdbg_ctx *ctx, const mdbg_table *t, uint64_t first, uint64_t count, uint8_t *records20, uint32_t *vectors);
/* What the pass that built the table walked: stats[0] = minimizers read (M), [1] = k-min-mer instances (I = sum over the
 * sequences of max(0, n - k + 1)), [2] = distinct keys it inserted, [3] = slots of the hash table it used.  With the table's
 * rows D these are the terms of the step's algorithmic bytes 4 M + 16 I + 20 D; zeros for tables that were
 * not built from sequences (mdbg_prev_from_records, the edge indexes, mdbg_shard_keep); a share of a sharded first pass reports
 * its rank's own reads and local keys. */
int  mdbg_table_stats(const mdbg_table *t, uint64_t stats[4]);
/* How the context's last mdbg_kminmer_count_first ran.  The reference counts the first pass's k-min-mers in `_nbPartitions`
 * partitions chosen by `vecHash % P`, each sorted and run-length counted (KminmerCounter, graph/CreateMdbg.hpp:3714-3851; P from
 * graph/CreateMdbg.cpp:222-225); the library either partitions the instances by bits of the identity and counts every bucket in LDS
 * (csrc/partition.hip) or meets them in one table in HBM (csrc/kminmer.hip; small inputs, k > 32).  info[0] = 2 / 1 for the two,
 * [1] = groups of keys processed one after the other (the analogue of _nbPartitions: more than one when the instance records of
 * all keys would not fit the memory budget), [2] = bucket bits per group, [3] = levels of the radix split, [4] = attempts (a
 * bucket that outgrows its LDS table makes the pass repeat with more buckets), [5] = LDS slots per bucket, [6] = buckets per
 * group, [7] = k-min-mer instances.  Options (mdbg_set_option): "first_pass_mode" 0 by size / 1 one table / 2 partitioned,
 * "partition_auto_min" (minimizers from which mode 0 partitions); for a context that shares its device with another context's
 * scan (a block only runs beside a scan if it fits what the scan's blocks leave of a CU) "partition_tile" 2048,
 * "partition_slot_list" 0 and "partition_lds_slots" 1024 select the kernels' 24 KB forms and "scan_lds_reserve" (bytes) makes
 * the scan leave that much of every CU's LDS free; for tests "partition_bits", "partition_lds_slots" (256 / 1024 / 2048),
 * "partition_max_records" (instances per group).  mdbg_shard_begin counts a rank's share the same way. */
int  mdbg_first_pass_info(const mdbg_ctx *ctx, uint64_t info[8]);
/* Order-independent sums over the rows, wrapping at 2^64, computed on the device (nothing but 32 bytes travels):
 *   sums[0] = sum abundance * hash_lo -- the "Checksum kminmer abundance" the reference logs when it loads the table again
 *             (CreateMdbg::loadRefinedAbundances, graph/CreateMdbg.cpp:3300-3321, :3397: `abundance * vecHash` truncated to u64),
 *   sums[1] = sum abundance, sums[2] = sum hash_hi,
 *   sums[3] = sum (v[0] + 3 v[1] + 5 v[2] + ...) * (hash_lo | 1) over the rows' vectors (0 when the table has none).
 * The sums of the per-rank tables of a sharded pass add up to those of the single-GPU table (the union of the shares is the
 * table): what a multi-GPU job checks itself with. */
int  mdbg_table_checksum(mdbg_ctx *ctx, const mdbg_table *t, uint64_t sums[4]);
/* In-process lookup for the reference's graph multiplexer (isEdgeSupported etc.,
 * graph/CreateMdbg.cpp:3990): abundance of each of n keys, 0 when absent. */
int  mdbg_table_lookup(mdbg_ctx *ctx, const mdbg_table *t, const uint64_t *hash_lo, const uint64_t *hash_hi,
                       uint64_t n, uint32_t *abundance);
/* SURVEY.md 8(f) N2 -- EdgeIndexer (graph/CreateMdbg.hpp:4010-4230, driven by indexEdges, graph/CreateMdbg.cpp:1178-1186):
 * the distinct identities of the normalised (k-1)-prefix and (k-1)-suffix of every k-min-mer vector of `nodes`
 * (a table with vectors, i.e. the rows of kminmerData_min.txt) = the content of edges.bin.  The result is a table
 * without vectors whose abundance column is 0; *checksum = sum of the identities truncated to u64 as the reference
 * logs it.  Fetch the 16-byte records with mdbg_table_keys_to_host (u128 little-endian: lo, hi). */
int  mdbg_edge_index(mdbg_ctx *ctx, const mdbg_table *nodes, mdbg_table **edges, uint64_t *checksum);
int  mdbg_table_keys_to_host(mdbg_ctx *ctx, const mdbg_table *t, uint64_t *keys_lo_hi);
/* Replaces UnitigEdgeIndexer::partitionEdges + dereplicatePartitions (graph/CreateMdbg.hpp:4234-4512): the distinct
 * (k-1)-prefix/suffix identities of the FIRST and LAST k-min-mer of every unitig of unitigGraph.nodes.bin (uploaded as
 * minimizer-space sequences; sequences shorter than k contribute nothing).  *checksum as in mdbg_edge_index (may be NULL). */
int  mdbg_unitig_edge_index(mdbg_ctx *ctx, const mdbg_minimizers *unitigs, uint32_t k, mdbg_table **edges, uint64_t *checksum);
void mdbg_table_free(mdbg_table *t);

/* ---- files of records handed over as BYTES (round 6) ---------------------------------------------------------------------------
 * What `graph` reads at every k -- read_data_corrected.txt and unitig_data.txt (`u32 n; u8 circular; u32 m[n]` per record,
 * readSelection/ReadSelection.hpp:1420-1426, read back by KminmerParserParallel under one critical section, Commons.hpp:7394-7424)
 * and kminmerData_abundance_prev.txt (20-byte records, loaded by CreateMdbg::loadRefinedAbundances, graph/CreateMdbg.cpp:3401-3491) --
 * need not be parsed into arrays on the host first: the file's bytes travel as they are, in pieces, from page-locked slabs
 * (mdbg_host_alloc) beside whatever else the process does, and the records are taken apart on the device.  One `graph` process per k
 * is how the reference runs the loop (pipeline/AssemblyPipeline.hpp:763-792): 97 of them in a default assembly of 10 kb reads.
 *
 * mdbg_bytes_create       a device buffer of n_bytes.
 * mdbg_bytes_upload_async `n` bytes from `host` to offset `at`, queued on the context's upload stream; returns at once when `host` is
 *                         page-locked.  *ticket (optional) names the copy: `host` may be reused once mdbg_bytes_upload_done(ticket) says so.
 * mdbg_bytes_upload_done  1 = the copy named by `ticket` (and every earlier one) is complete, 0 = not yet (wait != 0: block until it is).
 * mdbg_minimizers_from_record_bytes
 *                         the minimizer-space read set of a record file whose bytes are in `records`: `offsets` (host, n_reads + 1) are
 *                         the minimizers before each read, as the caller's walk over the record headers found them -- record r then
 *                         starts at byte 5 r + 4 offsets[r].  Every record's own count is checked against the offsets on the device
 *                         (MDBG_EINVAL on a mismatch: the bytes are not that file).  circular (optional, host, n_reads) receives the
 *                         records' flag bytes.  Uploads still in flight are waited for on the device, not by the host.
 * mdbg_prev_from_record_bytes
 *                         mdbg_prev_from_records for a table file whose bytes are in `records`. */
typedef struct mdbg_bytes mdbg_bytes;
int  mdbg_bytes_create(mdbg_ctx *ctx, uint64_t n_bytes, mdbg_bytes **out);
int  mdbg_bytes_upload_async(mdbg_ctx *ctx, mdbg_bytes *b, uint64_t at, const void *host, uint64_t n, uint64_t *ticket);
int  mdbg_bytes_upload_done(mdbg_ctx *ctx, mdbg_bytes *b, uint64_t ticket, int wait);
void mdbg_bytes_free(mdbg_bytes *b);
int  mdbg_minimizers_from_record_bytes(mdbg_ctx *ctx, const mdbg_bytes *records, const uint64_t *offsets, uint32_t n_reads,
                                       uint8_t *circular, mdbg_minimizers **out);
int  mdbg_prev_from_record_bytes(mdbg_ctx *ctx, const mdbg_bytes *records, uint64_t n_records, mdbg_table **out);

/* Page-locked host memory for read batches handed to mdbg_reads_from_ascii / _from_packed: uploads from it
 * run at PCIe rate instead of through the driver's staging copies.  Release with mdbg_host_free. */
int  mdbg_host_alloc(mdbg_ctx *ctx, size_t bytes, void **out);
void mdbg_host_free(mdbg_ctx *ctx, void *p);

/* Stream-ordered device-to-device copy on the context stream followed by a synchronize; lets a
 * harness move library-owned rows into buffers it owns (e.g. torch tensors for RCCL). */
int  mdbg_memcpy_device(mdbg_ctx *ctx, void *dst, const void *src, uint64_t bytes);

/* ---- sharded first pass (one process per GPU; the caller moves the bytes, e.g. RCCL all-to-all over xGMI) ----
 * Reads shard across ranks; only the counts are global.  Keys are partitioned by owner rank (top bits of hash_hi
 * scaled to [0, n_ranks), n_ranks <= 64).  Per rank and step:
 *     mdbg_shard_begin   local counts -> rows [hash_lo, hash_hi, count] grouped by owner                (send: all-to-all)
 *     mdbg_shard_reduce  owner sums the rows it received; reply[i] answers received row i                 (send back: all-to-all
 *                        with the transposed split sizes; replies arrive in the order the rows were sent)
 *     mdbg_shard_finish  global counts -> this rank's share of the solid rows + rescue of its own reads
 * A reply is the GLOBAL count of the row's key (low 32 bits) plus bit 63 on exactly one of the rows of each key: the
 * rank that sent that row lists the key in its table, with the vector from its own reads -- vectors never travel.
 * The union over ranks of the finished tables equals mdbg_kminmer_count_first on the union of the reads.
 * Nearest reference analogue: KminmerCounter's on-disk partitioning `vecHash % _nbPartitions`
 * (graph/CreateMdbg.hpp:3714-3724).  Rows are mdbg_row_words(k) = 3 u64 words each. */
typedef struct mdbg_shard mdbg_shard;
uint32_t mdbg_row_words(uint32_t k);
/* *d_rows: device rows of every distinct local key, owner 0 first; counts[r] = rows for rank r.  `reads` must stay
 * alive until mdbg_shard_free. */
int  mdbg_shard_begin(mdbg_ctx *ctx, const mdbg_minimizers *reads, uint32_t k, uint32_t n_ranks,
                      mdbg_shard **shard, const uint64_t **d_rows, uint64_t *counts);
/* d_recv: device rows received from all ranks (any order; free to reuse once the call returns).
 * *d_reply: n_recv u64, aligned with d_recv, valid until mdbg_shard_free. */
int  mdbg_shard_reduce(mdbg_ctx *ctx, mdbg_shard *shard, const uint64_t *d_recv, uint64_t n_recv, const uint64_t **d_reply);
/* d_replies: n_sent u64, the replies for the rows of mdbg_shard_begin in the order they were sent. */
int  mdbg_shard_finish(mdbg_ctx *ctx, mdbg_shard *shard, const uint64_t *d_replies, uint32_t min_abundance, mdbg_table **out);
/* Both exchanges of a sharded pass among `n` shards that live on one device (begun with n_ranks = n on contexts of that device):
 * rows to their owners, mdbg_shard_reduce on every owner, the replies back in the order the rows were sent -- what
 * mdbg_shard_exchange does between GPUs, with device-to-device copies in place of the wire.  counts[src * n + dst] = rows shard
 * src holds for owner dst (the `counts` of its mdbg_shard_begin / _from_table), d_rows[src] its rows; d_replies[src] receives the
 * replies for mdbg_shard_finish / _keep (owned by the shard).  For a job that checks itself against the union of shards (bench.py's
 * self-checks: the table of a whole read set against the shares of its halves) and for tests; nothing in the reference
 * corresponds to it -- its partitions meet on disk (graph/CreateMdbg.hpp:3714-3851). */
int  mdbg_shard_exchange_local(mdbg_ctx *ctx, mdbg_shard *const *shards, uint32_t n, const uint64_t *const *d_rows, const uint64_t *counts,
                               const uint64_t **d_replies);
void mdbg_shard_free(mdbg_shard *shard);
/* Sharded k > firstK (the refined pass and the index passes, graph/CreateMdbg.cpp:391-468).  The abundance of a k-min-mer at
 * these k is a function of the key and the previous table, not of a count, so nothing is summed: every rank holds the whole
 * previous table (20 bytes per key: a few hundred MB at 40 M reads -- all-gather the records of the previous step and load them
 * with mdbg_prev_from_records), runs mdbg_kminmer_count_refined / mdbg_kminmer_index over ITS reads, and the ranks then only
 * agree on who lists a key that several of them found:
 *     mdbg_shard_from_table  rows [hash_lo, hash_hi, abundance] of the local table grouped by owner           (send: all-to-all)
 *     mdbg_shard_reduce      the owner marks the first row of every key (bit 63 of its reply), as in the first pass
 *     mdbg_shard_keep        the rows this rank was told to list, vectors included when the table has them
 * The union over ranks of the kept tables equals the single-GPU table over the union of the reads.  `local` must stay alive
 * until mdbg_shard_free; the exchange in between is mdbg_shard_exchange or the caller's own. */
int  mdbg_shard_from_table(mdbg_ctx *ctx, const mdbg_table *local, uint32_t n_ranks, mdbg_shard **shard, const uint64_t **d_rows,
                           uint64_t *counts);
int  mdbg_shard_keep(mdbg_ctx *ctx, mdbg_shard *shard, const uint64_t *d_replies, mdbg_table **out);

/* ---- the exchange inside the library: peer copies or RCCL point-to-point, over xGMI --------------------------------
 * For callers that do not want to move the bytes themselves (the C++ pipeline: src/pipeline has no communication layer).
 * One communicator per context that takes part; rank 0 makes the id (128 bytes) and hands it to the other ranks by whatever
 * means it has (a file in the shared tmp dir, a socket, MPI, torch.distributed -- bench.py broadcasts it).  Ranks may be
 * processes (one per GPU) or threads of one process driving one context each (mdbg_tool graph --gpus G).
 *     mdbg_comm_unique_id    ncclGetUniqueId (128 random bytes when RCCL is not installed: enough for MDBG_COMM_PEER)
 *     mdbg_comm_create_mode  collective, every rank calls it with the same id and mode:
 *         MDBG_COMM_PEER   the two all-to-alls as PEER COPIES.  Every rank stages its rows (grouped by owner) and later its replies in
 *                          a buffer of its own; every owner PULLS its slice from every sender with a device-to-device copy, one
 *                          stream per peer, so the seven xGMI links of a GPU carry their slices at once; the replies travel the
 *                          same way back.  Between processes the buffers are shared by hipIpcGetMemHandle / hipIpcOpenMemHandle,
 *                          between threads of one process they are plain pointers (peer access enabled between the devices).  The
 *                          hand-shakes -- the count matrix, "my rows are staged", the status words of the failure protocol below --
 *                          are words in a block of host memory the ranks share (csrc/peerlink.hpp: a POSIX shared-memory object named
 *                          after the id, unlinked once everybody is attached), polled with a deadline (MDBG_PEER_TIMEOUT_S, default
 *                          120): no collective kernel has to become resident beside another batch's scan -- RCCL's needs 37.6 KB of
 *                          LDS a block, a scan leaves 28 -- and no exchange can hang.  Creation ends with a self-test: a small
 *                          exchange of known rows through the very buffers, whose verdicts the ranks agree on.
 *         MDBG_COMM_RCCL   ncclCommInitRank; every transfer an ncclSend / ncclRecv pair inside one group.  RCCL is loaded on first use.
 *         MDBG_COMM_AUTO   peer copies if every rank attaches and passes the self-test, otherwise -- all ranks together -- RCCL
 *                          (mdbg_comm_note says why).  A caller with other batches' scans in flight then keeps them off the device
 *                          during an exchange (bench.py's exchange gate).
 *         MDBG_COMM_DEFAULT  the environment's MDBG_COMM_MODE = peer | rccl | auto, "auto" when unset
 *     mdbg_comm_create       = mdbg_comm_create_mode(..., MDBG_COMM_DEFAULT, ...)
 *     mdbg_comm_adopt        wraps a communicator the caller already has (an ncclComm_t); mdbg_comm_destroy leaves it alive
 *     mdbg_comm_mode         the transport a communicator ended up with (MDBG_COMM_PEER or MDBG_COMM_RCCL)
 *     mdbg_comm_destroy      collective in peer mode (a rank frees its staging buffers once its peers have let go of them; bounded wait)
 * mdbg_kminmer_count_first_sharded = mdbg_shard_begin -> rows to their owner ranks -> mdbg_shard_reduce -> replies back ->
 * mdbg_shard_finish, all on the context's stream (the pulls on side streams it waits for).  Collective: every rank of the
 * communicator calls it, in the same order if a rank drives several communicators.  The union over ranks of the tables equals
 * mdbg_kminmer_count_first over the union of the reads.  Replaces, across GPUs, KminmerCounter's partition-to-disk +
 * per-partition dereplication (graph/CreateMdbg.hpp:3714-3724, :3744-3851). */
typedef struct mdbg_comm mdbg_comm;
#define MDBG_COMM_ID_BYTES 128
int  mdbg_comm_unique_id(uint8_t *id128);
#define MDBG_COMM_DEFAULT (-1)
#define MDBG_COMM_RCCL    0
#define MDBG_COMM_PEER    1
#define MDBG_COMM_AUTO    2
int  mdbg_comm_create(mdbg_ctx *ctx, const uint8_t *id128, int rank, int n_ranks, mdbg_comm **out);
int  mdbg_comm_create_mode(mdbg_ctx *ctx, const uint8_t *id128, int rank, int n_ranks, int mode, mdbg_comm **out);
int  mdbg_comm_mode(const mdbg_comm *comm);
const char *mdbg_comm_note(const mdbg_comm *comm);     /* "auto" that ended on RCCL: why the peer copies were not taken; else "" */
int  mdbg_comm_adopt(mdbg_ctx *ctx, void *nccl_comm, int rank, int n_ranks, mdbg_comm **out);
void mdbg_comm_destroy(mdbg_comm *comm);
/* What the communicator has carried: stats[0] = rank, [1] = ranks as the caller gave them, [2] = ncclCommCount (RCCL: create / adopt
 * fail unless it equals [1] and ncclCommUserRank equals [0]; 0 for peer copies), [3] = completed exchanges, [4] / [5] = bytes that went
 * to / came from OTHER ranks (rows and replies; peer copies: what the peers pulled from this rank / what it pulled), [6] = bytes of the
 * rank's own share (device-to-device copies, never on the wire), [7] = ncclCommUserRank; *exchange_ms (may be NULL) = host wall
 * time spent inside mdbg_shard_exchange. */
int  mdbg_comm_stats(const mdbg_comm *comm, uint64_t stats[8], double *exchange_ms);
/* Where the time inside mdbg_shard_exchange went (peer copies; RCCL reports [1] = [2] = 0): ms[0] = the whole call (= *exchange_ms above),
 * ms[1] = the owner's reduction (mdbg_shard_reduce: this rank's device work), ms[2] = waiting -- for this rank's own copies to land and
 * for its peers to reach a phase.  ms[0] - ms[1] - ms[2] is the transport's own host time. */
int  mdbg_comm_times(const mdbg_comm *comm, double ms[3]);
int  mdbg_kminmer_count_first_sharded(mdbg_ctx *ctx, mdbg_comm *comm, const mdbg_minimizers *reads, uint32_t k,
                                      uint32_t min_abundance, mdbg_table **out);
/* The collective middle part alone, between mdbg_shard_begin and mdbg_shard_finish (a caller with several batches in flight
 * overlaps the local halves and takes turns on the wire): d_rows / counts as mdbg_shard_begin returned them; *d_replies (one
 * u64 per sent row, in the order sent) is what mdbg_shard_finish takes and stays valid until the next exchange on `comm`. */
int  mdbg_shard_exchange(mdbg_ctx *ctx, mdbg_comm *comm, mdbg_shard *shard, const uint64_t *d_rows, const uint64_t *counts,
                         const uint64_t **d_replies);
/* Failure behaviour of the collective calls.  An exchange is: counts all-gathered -> rows to owners -> reduction -> replies.
 * Before each transfer the ranks agree (a status word per rank: all-gathered over RCCL, or written to the shared control block of
 * the peer copies, where waiting for a rank that died ends at a deadline with MDBG_EPEER) that everybody got that far: a rank that failed
 * locally -- bad argument, allocation, reduction -- still takes part in that small collective with its error code, returns its
 * own error, and every other rank returns MDBG_EPEER naming it; nobody is left waiting in a receive.  Send / receive groups
 * are closed on every path (an open group would swallow the thread's next RCCL call); a communicator on which an RCCL call
 * itself failed is marked broken: later exchanges on it fail at once and mdbg_comm_destroy aborts it.
 * mdbg_shard_abort is for the caller that drives begin / exchange / finish itself: when its local half (mdbg_shard_begin,
 * mdbg_shard_from_table, anything before the exchange) failed with `code`, it calls this INSTEAD of mdbg_shard_exchange so
 * that the peers, who are about to enter theirs, return MDBG_EPEER.  Returns MDBG_OK once the peers have been told. */
int  mdbg_shard_abort(mdbg_ctx *ctx, mdbg_comm *comm, int code);

#ifdef __cplusplus
}
#endif
#endif
