/*
 * ref_binding.cpp -- the binding INTEGRATION.md describes, compiled INTO the reference's own tools.
 *
 * TEST INFRASTRUCTURE ONLY (like everything under oracle/).  #included at the end of ref_driver.cpp when oracle/Makefile builds
 * oracle/_ref/refdrv_hip (-DMDBG_WITH_HIP_BINDING, linked against metamdbg_amd/libmdbg_hip.so).  No reference source is copied
 * or edited: the two classes below DERIVE from the reference's ReadSelection and CreateMdbg where those lie under /root/reference
 * and replace only what INTEGRATION.md sections 1 and 2 say a maintainer would replace -- the compute between the reference's
 * parser and the reference's record writer, and the production of the k-min-mer tables in front of the reference's graph stage.
 * Everything around it is the reference's code, running unchanged:
 *
 *   refdrv_hip readSelection_hip <same argv as readSelection>
 *       ReadSelection::parseArgs, ReadParserParallel + kseq (Commons.hpp:5846-5914), ReadSelection::writeRead -- the ordered
 *       writer of read_data_init.txt (:386-491) --, ReadSelection::computeReadStats (:305-384), Commons::computeLastK, Tool::end
 *       are the reference's; ReadSelectionFunctor::operator() (:669-1158), determineRepetitiveMinimizers' counting (:497-625) and
 *       purgePalindromes' per-read work (:1374-1431) are calls into the library through include/mdbg_hip.h.
 *   refdrv_hip graph_hip <same argv as graph>
 *       CreateMdbg::parseArgs and everything behind the tables (createGfa / computeNextUnitigGraph, as `graph_from_tables` above
 *       runs them) are the reference's; KminmerCounter + rescueKminmers / loadRefinedAbundances + IndexKminmerFunctor
 *       (graph/CreateMdbg.cpp:290-468) are calls into the library.
 *
 * tests/test_gpu_reference_binding.py runs both beside the unmodified `refdrv readSelection` / `refdrv graph` on the same inputs and
 * compares the files: the proof that the C ABI is a drop-in for the path INSIDE the reference, not only beside it.
 */
#include "../include/mdbg_hip.h"

#include <mutex>

namespace hipbind {

static void check(mdbg_ctx *ctx, int rc, const char *what)
{
    if (rc == MDBG_OK) return;
    Logger::get().error() << what << ": " << mdbg_last_error(ctx);      /* the reference's convention for fatal conditions, Commons.hpp:5749-5752 */
    exit(1);
}

/* ---- readSelection ------------------------------------------------------------------------------------------------------------ */
class ReadSelectionHip : public ReadSelection {
public:
    mdbg_ctx *_gpu = nullptr;
    /* the batch the parser threads append to */
    std::mutex _batchMutex;
    std::string _batchBases, _batchQuals;
    std::vector<uint64_t> _batchOffsets{0};
    std::vector<Read> _batchReads;                 /* index + length is all writeRead looks at: _seq keeps the length only */
    bool _batchHasQual = false;
    size_t _batchLimit = (size_t)64 << 20;         /* bases per device batch */
    /* which scan a flush runs: the census of determineRepetitiveMinimizers or the main pass */
    bool _censusPass = false;
    mdbg_census *_census = nullptr;
    std::vector<uint32_t> _repetitiveVec;
    struct Kept { mdbg_minimizers *mins; };
    std::vector<Kept> _kept;                       /* minimizer reads left in HBM for purgePalindromes */

    struct BatchingFunctor {                       /* what ReadParserParallel::parse copies per OpenMP thread and calls per read */
        ReadSelectionHip &_parent;
        explicit BatchingFunctor(ReadSelectionHip &parent) : _parent(parent) {}
        BatchingFunctor(const BatchingFunctor &copy) : _parent(copy._parent) {}
        void operator()(const Read &read) { _parent.addRead(read); }
    };

    void addRead(const Read &read)
    {
        std::lock_guard<std::mutex> g(_batchMutex);
        const bool hasQual = !read._qual.empty() && !_censusPass;
        if (!_batchReads.empty() && (hasQual != _batchHasQual || _batchBases.size() + read._seq.size() > _batchLimit)) flushBatch();
        _batchHasQual = hasQual;
        _batchBases += read._seq;
        if (hasQual) _batchQuals += read._qual;
        _batchOffsets.push_back(_batchBases.size());
        Read light;
        light._index = read._index;
        light._datasetIndex = read._datasetIndex;
        light._seq.assign(read._seq.size(), 'N');  /* writeRead records read._seq.size() */
        _batchReads.push_back(std::move(light));
    }

    /* replaces N calls of ReadSelectionFunctor::operator() (or CountMinimizerFunctor::operator()); _batchMutex held */
    void flushBatch()
    {
        if (_batchReads.empty()) return;
        const uint32_t n = (uint32_t)_batchReads.size();
        mdbg_reads *reads = nullptr;
        mdbg_minimizers *mins = nullptr;
        check(_gpu, mdbg_reads_from_ascii(_gpu, _batchBases.data(), _batchHasQual ? _batchQuals.data() : nullptr, _batchOffsets.data(), n, &reads),
              "mdbg_reads_from_ascii");
        mdbg_scan_params p{};
        p.minimizer_size = (uint32_t)_params._minimizerSize;
        p.hpc = _params._useHomopolymerCompression ? 1 : 0;
        if (_censusPass) {                         /* CountMinimizerFunctor, ReadSelection.hpp:565-625 */
            p.density = _params._minimizerDensity_correction;
            p.apply_read_filters = 0;
            check(_gpu, mdbg_scan(_gpu, reads, &p, &mins), "mdbg_scan");
            check(_gpu, mdbg_census_add(_gpu, _census, mins), "mdbg_census_add");
            mdbg_minimizers_free(mins);
        } else {
            p.density = _params._minimizerDensity_assembly;
            p.min_read_quality = _minReadQuality;
            p.repetitive = _repetitiveVec.empty() ? nullptr : _repetitiveVec.data();
            p.n_repetitive = (uint32_t)_repetitiveVec.size();
            p.apply_read_filters = 1;
            check(_gpu, mdbg_scan(_gpu, reads, &p, &mins), "mdbg_scan");
            uint32_t nr = 0;
            uint64_t total = 0;
            mdbg_minimizers_info(mins, &nr, &total);
            std::vector<uint64_t> off((size_t)n + 1);
            std::vector<uint32_t> m(total), pos(total), len(n);
            std::vector<uint8_t> dir(total), qual(total), flags(n);
            std::vector<float> meanQ(n);
            check(_gpu, mdbg_minimizers_to_host(_gpu, mins, off.data(), m.data(), pos.data(), dir.data(), qual.data(), len.data(), meanQ.data(),
                                                flags.data()), "mdbg_minimizers_to_host");
            for (uint32_t r = 0; r < n; r++) {
                /* the counters ReadSelectionFunctor keeps beside the records (ReadSelection.hpp:890-915) */
                if (flags[r] & MDBG_READ_LOW_COMPLEXITY) _nbLowComplexityReads += 1;
                if (flags[r] & MDBG_READ_LOW_QUALITY) _nbLowQualityReads += 1;
                else { _readQualitySum += meanQ[r]; _readQualityN += 1; }
                const vector<MinimizerType> rm(m.begin() + (long)off[r], m.begin() + (long)off[r + 1]);
                const vector<u_int32_t> rp(pos.begin() + (long)off[r], pos.begin() + (long)off[r + 1]);
                const vector<u_int8_t> rd(dir.begin() + (long)off[r], dir.begin() + (long)off[r + 1]);
                const vector<u_int8_t> rq(qual.begin() + (long)off[r], qual.begin() + (long)off[r + 1]);
                writeRead(_batchReads[r], rm, rp, rd, rq, meanQ[r]);           /* the reference's ordered writer, unchanged */
            }
            if (_params._useHomopolymerCompression || _skipCorrection) _kept.push_back({mins});
            else mdbg_minimizers_free(mins);
        }
        mdbg_reads_free(reads);
        _batchBases.clear(); _batchQuals.clear(); _batchOffsets.assign(1, 0); _batchReads.clear();
    }

    /* determineRepetitiveMinimizers (ReadSelection.hpp:497-561) with the counting on the device */
    void determineRepetitiveMinimizersHip()
    {
        ofstream outputFile(_inputDir + "/repetitiveMinimizers.bin");
        if (_params._useHomopolymerCompression) { outputFile.close(); return; }
        check(_gpu, mdbg_census_create(_gpu, &_census), "mdbg_census_create");
        _censusPass = true;
        ReadParserParallel readParser(_inputFilename, false, false, _nbCores);
        readParser._maxReads = 1000000;
        readParser.parse(BatchingFunctor(*this));
        { std::lock_guard<std::mutex> g(_batchMutex); flushBatch(); }
        _censusPass = false;
        uint32_t cap = 1u << 16;
        _repetitiveVec.resize(cap);
        check(_gpu, mdbg_census_top(_gpu, _census, _repetitiveVec.data(), &cap), "mdbg_census_top");
        _repetitiveVec.resize(cap);
        mdbg_census_free(_census);
        for (uint32_t v : _repetitiveVec) _isRepetitiveMinimizer.insert((MinimizerType)v);
        for (const MinimizerType &minimizer : _isRepetitiveMinimizer) outputFile.write((const char *)&minimizer, sizeof(minimizer));
        outputFile.close();
    }

    /* purgePalindromes (ReadSelection.hpp:1374-1431) on the batches kept in HBM */
    void purgePalindromesHip()
    {
        _kminmerSizeLast = Commons::computeLastK(_params._minimizerDensity_assembly, _n50ReadLength, _params._kminmerSizeFirst, 0);
        _file_readData = ofstream(_inputDir + "/read_data_corrected.txt");
        for (Kept &k : _kept) {
            mdbg_minimizers *pur = nullptr;
            check(_gpu, mdbg_purge_palindromes(_gpu, k.mins, (uint32_t)_params._kminmerSizeFirst, (uint32_t)_kminmerSizeLast, &pur), "mdbg_purge_palindromes");
            mdbg_minimizers_free(k.mins);
            uint32_t n = 0;
            uint64_t total = 0;
            mdbg_minimizers_info(pur, &n, &total);
            std::vector<uint64_t> off((size_t)n + 1);
            std::vector<uint32_t> m(total);
            check(_gpu, mdbg_minimizers_to_host(_gpu, pur, off.data(), m.data(), nullptr, nullptr, nullptr, nullptr, nullptr, nullptr), "mdbg_minimizers_to_host");
            mdbg_minimizers_free(pur);
            for (uint32_t r = 0; r < n; r++) {     /* the record PurgePalindromeFunctor writes (:1417-1427) */
                u_int32_t size = (u_int32_t)(off[r + 1] - off[r]);
                _file_readData.write((const char *)&size, sizeof(size));
                u_int8_t isCircular = CONTIG_LINEAR;
                _file_readData.write((const char *)&isCircular, sizeof(isCircular));
                _file_readData.write((const char *)(m.data() + off[r]), size * sizeof(MinimizerType));
            }
        }
        _file_readData.close();
    }

    /* ReadSelection::execute + ReadSelection::readSelection (:91-111, :251-303) with the three compute steps redirected */
    void execute() override
    {
        _nbKmers = 0;
        _nbBases = 0;
        _nbSelectedMinimizers = 0;
        _nbLowQualityReads = 0;
        _nbLowComplexityReads = 0;
        _readQualitySum = 0;
        _readQualityN = 0;
        check(nullptr, mdbg_create(0, &_gpu), "mdbg_create");
        if (const char *e = getenv("MDBG_BINDING_BATCH_BASES")) _batchLimit = (size_t)atoll(e);      /* tests: many small batches */

        _nextReadIndexWriter = 0;
        _debug_nbMinimizers = 0;
        _file_readData = ofstream(_outputFilename);
        determineRepetitiveMinimizersHip();
        ReadParserParallel readParser(_inputFilename, false, false, _nbCores);
        readParser.parse(BatchingFunctor(*this));
        { std::lock_guard<std::mutex> g(_batchMutex); flushBatch(); }
        _file_readData.close();
        computeReadStats();                         /* the reference's, on the counters kept above */
        if (_params._useHomopolymerCompression || _skipCorrection) purgePalindromesHip();
        Logger::get().debug() << "Nb low quality reads:    " << _nbLowQualityReads;
        Logger::get().debug() << "Nb low complexity reads: " << _nbLowComplexityReads;
        mdbg_destroy(_gpu);
    }
};

/* ---- graph -------------------------------------------------------------------------------------------------------------------- */
static std::vector<uint8_t> file_bytes(const std::string &path, bool required)
{
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        if (required) { Logger::get().error() << "File not found: " << path; exit(1); }
        return {};
    }
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

/* "u32 n; u8 circ; u32 m[n]" records (read_data_corrected.txt, unitig_data.txt: Commons.hpp:7413, :7485) -> CSR */
static void parse_minimizer_reads(const std::vector<uint8_t> &raw, std::vector<uint32_t> &mins, std::vector<uint64_t> &offs, std::vector<uint8_t> *circular)
{
    offs.assign(1, 0);
    for (size_t o = 0; o + 5 <= raw.size();) {
        uint32_t n;
        memcpy(&n, raw.data() + o, 4);
        if (circular) circular->push_back(raw[o + 4]);
        o += 5;
        if (o + (size_t)n * 4 > raw.size()) { Logger::get().error() << "truncated minimizer read file"; exit(1); }
        const size_t base = mins.size();
        mins.resize(base + n);
        if (n) memcpy(mins.data() + base, raw.data() + o, (size_t)n * 4);
        o += (size_t)n * 4;
        offs.push_back(mins.size());
    }
}

static mdbg_minimizers *upload(mdbg_ctx *gpu, const std::vector<uint32_t> &mins, const std::vector<uint64_t> &offs)
{
    mdbg_minimizers *out = nullptr;
    static const uint32_t none = 0;
    check(gpu, mdbg_minimizers_from_host(gpu, mins.empty() ? &none : mins.data(), offs.data(), (uint32_t)(offs.size() - 1), &out), "mdbg_minimizers_from_host");
    return out;
}

/* CreateMdbg::createMDBG up to the moment its tables are on disk (graph/CreateMdbg.cpp:290-468, :515-522), INTEGRATION.md section 2 */
static void produce_tables(CreateMdbg &g)
{
    mdbg_ctx *gpu = nullptr;
    check(nullptr, mdbg_create(0, &gpu), "mdbg_create");
    const std::string dir = g._outputDir;
    const uint32_t k = (uint32_t)g._kminmerSize, firstK = (uint32_t)g._kminmerSizeFirst;
    std::vector<uint32_t> mins;
    std::vector<uint64_t> offs;
    parse_minimizer_reads(file_bytes(dir + "/read_data_corrected.txt", true), mins, offs, nullptr);
    mdbg_minimizers *reads = upload(gpu, mins, offs);
    mdbg_table *table = nullptr;
    /* every `graph` run truncates smallContigs/smallContigs_k<k>.bin (graph/CreateMdbg.cpp:258-259) */
    ofstream small(dir + "/smallContigs/smallContigs_k" + std::to_string(k) + ".bin", std::ios::binary);
    if (g._isFirstPass) {
        check(gpu, mdbg_kminmer_count_first(gpu, reads, k, (uint32_t)g._minAbundance, &table), "mdbg_kminmer_count_first");
    } else {
        /* loadRefinedAbundances' inputs (graph/CreateMdbg.cpp:3401-3709) */
        const std::vector<uint8_t> prevRec = file_bytes(dir + "/kminmerData_abundance_prev.txt", true);
        mdbg_table *prev = nullptr;
        check(gpu, mdbg_prev_from_records(gpu, prevRec.data(), prevRec.size() / 20, &prev), "mdbg_prev_from_records");
        const std::vector<uint8_t> ab = file_bytes(dir + "/unitigGraph.nodes.refined_abundances.bin", false);      /* (u32 unitigName, u32 abundance)* */
        std::vector<std::pair<uint32_t, uint32_t>> name2ab(ab.size() / 8);
        for (size_t i = 0; i < name2ab.size(); i++) { memcpy(&name2ab[i].first, ab.data() + 8 * i, 4); memcpy(&name2ab[i].second, ab.data() + 8 * i + 4, 4); }
        std::stable_sort(name2ab.begin(), name2ab.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
        const std::vector<uint8_t> nodes = file_bytes(dir + "/unitigGraph_prev.nodes.bin", false);                  /* (u32 size; u32 m[size]; u32 unitigIndex)* */
        std::vector<uint32_t> um, uab;
        std::vector<uint64_t> uoff{0};
        for (size_t o = 0; o + 4 <= nodes.size();) {
            uint32_t n;
            memcpy(&n, nodes.data() + o, 4); o += 4;
            if (o + (size_t)n * 4 + 4 > nodes.size()) { Logger::get().error() << "truncated unitigGraph_prev.nodes.bin"; exit(1); }
            const size_t base = um.size();
            um.resize(base + n);
            if (n) memcpy(um.data() + base, nodes.data() + o, (size_t)n * 4);
            o += (size_t)n * 4;
            uint32_t idx;
            memcpy(&idx, nodes.data() + o, 4); o += 4;
            uoff.push_back(um.size());
            uint32_t v = 0xFFFFFFFFu;              /* no refined abundance: skipped (CreateMdbg.cpp:3483); several entries: the map keeps the last */
            for (auto it = std::lower_bound(name2ab.begin(), name2ab.end(), std::make_pair(idx / 2, 0u), [](const auto &a, const auto &b) { return a.first < b.first; });
                 it != name2ab.end() && it->first == idx / 2; ++it) v = it->second;
            uab.push_back(v);
        }
        if (!uab.empty()) {
            mdbg_minimizers *un = upload(gpu, um, uoff);
            check(gpu, mdbg_prev_overlay_unitigs(gpu, prev, un, uab.data(), (uint32_t)g._kminmerSizePrev), "mdbg_prev_overlay_unitigs");
            mdbg_minimizers_free(un);
        }
        mdbg_minimizers *unitigs = nullptr;
        std::vector<uint32_t> unitigMins;
        std::vector<uint64_t> unitigOffs;
        std::vector<uint8_t> unitigCirc;
        if (fs::exists(dir + "/unitig_data.txt")) {
            parse_minimizer_reads(file_bytes(dir + "/unitig_data.txt", true), unitigMins, unitigOffs, &unitigCirc);
            unitigs = upload(gpu, unitigMins, unitigOffs);
        }
        if (unitigs && k > 8 && k != firstK + 1) { /* IndexKminmerFunctor's small contigs, graph/CreateMdbg.hpp:1330-1352 */
            const uint32_t nUnitigs = (uint32_t)(unitigOffs.size() - 1);
            std::vector<uint8_t> isSmall(nUnitigs);
            check(gpu, mdbg_small_contigs(gpu, unitigs, k, (uint32_t)g._kminmerSizePrev, prev, isSmall.data()), "mdbg_small_contigs");
            for (uint32_t u = 0; u < nUnitigs; u++) {
                if (!isSmall[u]) continue;
                const uint32_t n = (uint32_t)(unitigOffs[u + 1] - unitigOffs[u]);
                small.write((const char *)&n, 4);
                small.write((const char *)&unitigCirc[u], 1);
                small.write((const char *)(unitigMins.data() + unitigOffs[u]), (std::streamsize)n * 4);
            }
        }
        if (k == firstK + 1) check(gpu, mdbg_kminmer_count_refined(gpu, reads, unitigs, k, prev, &table), "mdbg_kminmer_count_refined");
        else check(gpu, mdbg_kminmer_index(gpu, reads, unitigs, k, prev, &table), "mdbg_kminmer_index");
        if (unitigs) mdbg_minimizers_free(unitigs);
        mdbg_table_free(prev);
    }
    small.close();
    uint64_t n = 0, nSolid = 0;
    int hasVec = 0;
    mdbg_table_info(table, nullptr, &n, &nSolid, &hasVec);
    std::vector<uint8_t> rec(n * 20);
    std::vector<uint32_t> vec(hasVec ? n * k : 0);
    check(gpu, mdbg_table_to_host(gpu, table, rec.data(), hasVec ? vec.data() : nullptr), "mdbg_table_to_host");
    auto put = [&](const std::string &name, const void *p, size_t bytes) {
        ofstream f(dir + name, std::ios::binary);
        f.write((const char *)p, (std::streamsize)bytes);
    };
    put("/kminmerData_abundance.txt", rec.data(), rec.size());
    if (hasVec) put("/kminmerData_min.txt", vec.data(), vec.size() * 4);
    if (g._isFirstPass) put("/kminmerData_abundance_init.txt", rec.data(), rec.size());                       /* graph/CreateMdbg.cpp:515-522 */
    if (k == firstK + 1) put("/kminmerData_abundance_init_k" + std::to_string(firstK + 1) + ".txt", rec.data(), rec.size());
    if (g._isFirstPass) {
        Logger::get().debug() << "Nb solid kminmers: " << nSolid;
        Logger::get().debug() << "Nb rescued kminmers: " << (n - nSolid);
    }
    mdbg_table_free(table);
    mdbg_minimizers_free(reads);
    mdbg_destroy(gpu);
}

}  // namespace hipbind

static int graph_hip(int argc, char **argv)
{
    {
        CreateMdbg g;                               /* the reference's own argument parsing: _outputDir, _kminmerSize, _isFirstPass, _minAbundance */
        g.parseArgs(argc, argv);
        hipbind::produce_tables(g);
    }
    return graph_from_tables(argc, argv);           /* the rest of the reference's `graph`, on those tables */
}

/* ---- SURVEY 8(f) N1 and N2 inside the reference's process (round-5 VERDICT item 8) -----------------------------------------------
 * What could NOT be done, and why: ReadCorrection as a tool (readSelection/ReadCorrection.hpp) keeps its records in BGZF partitions
 * (htslib) compressed with TurboPFor, and CreateMdbg::createGfa calls the non-virtual indexEdges() (graph/CreateMdbg.cpp:1178) whose second
 * half builds a BooPHF over the keys: neither links here without ext/ libraries built by their own build systems, and indexEdges()
 * cannot be redirected from a derived class without repeating its body.  So the two rows are bound at the narrowest seam that does link:
 *
 *   refdrv_hip fn_corrscan_hip <K> <density> <hpc>      same stdin / stdout as `refdrv fn_corrscan`: the compute of
 *       ReadCorrection::ReadSelectionFunctor::operator() (ReadCorrection.hpp:2298-2342 -- EncoderRLE, MinimizerParser::parse at the
 *       correction density, getMinQuality over [rle[pos], rle[pos + l - 1]]) done by ONE mdbg_scan over all the lines
 *       (quality_window = 1, apply_read_filters = 0); what the functor does with the result (CompressedMinimizerRead, bgzf_write) is untouched.
 *   refdrv_hip edges_hip <tmpDir> --threads N [--firstpass]      after `graph_hip` left its tables in <tmpDir>: the REFERENCE's own
 *       EdgeIndexer (graph/CreateMdbg.hpp:4010-4230, constructed on the reference's CreateMdbg exactly as indexEdges() does, :1181-1183)
 *       writes edges.bin from kminmerData_min.txt, mdbg_edge_index makes the same set from the same vectors, and the two are compared
 *       key by key (count, the checksum the reference logs, every 128-bit identity); then the same for the reference's UnitigEdgeIndexer
 *       (:4234-4512) over the unitigGraph.nodes.bin the graph stage left, against mdbg_unitig_edge_index. */
static int fn_corrscan_hip(int argc, char **argv)
{
    if (argc < 5) return 2;
    const size_t K = std::stoul(argv[2]);
    const float density = std::stof(argv[3]);
    const bool hpc = std::stoi(argv[4]) != 0;
    std::string bases, quals, line;
    std::vector<uint64_t> offs{0};
    while (std::getline(std::cin, line)) {
        std::istringstream is(line);
        std::string seq, qual;
        is >> seq >> qual;
        if (qual.size() != seq.size()) { std::cerr << "fn_corrscan_hip: one quality per base\n"; return 2; }
        bases += seq; quals += qual;
        offs.push_back(bases.size());
    }
    const uint32_t n = (uint32_t)(offs.size() - 1);
    mdbg_ctx *gpu = nullptr;
    hipbind::check(nullptr, mdbg_create(0, &gpu), "mdbg_create");
    mdbg_reads *reads = nullptr;
    hipbind::check(gpu, mdbg_reads_from_ascii(gpu, bases.data(), quals.data(), offs.data(), n, &reads), "mdbg_reads_from_ascii");
    mdbg_scan_params P;
    memset(&P, 0, sizeof P);
    P.minimizer_size = (uint32_t)K; P.density = density; P.hpc = hpc ? 1 : 0;
    P.apply_read_filters = 0; P.quality_window = 1;
    mdbg_minimizers *m = nullptr;
    hipbind::check(gpu, mdbg_scan(gpu, reads, &P, &m), "mdbg_scan");
    uint64_t total = 0;
    mdbg_minimizers_info(m, nullptr, &total);
    std::vector<uint64_t> mo((size_t)n + 1);
    std::vector<uint32_t> mv(total + 1), mp(total + 1);
    std::vector<uint8_t> md(total + 1), mq(total + 1);
    hipbind::check(gpu, mdbg_minimizers_to_host(gpu, m, mo.data(), mv.data(), mp.data(), md.data(), mq.data(), nullptr, nullptr, nullptr), "mdbg_minimizers_to_host");
    for (uint32_t r = 0; r < n; r++) {
        std::cout << (mo[r + 1] - mo[r]);
        for (uint64_t i = mo[r]; i < mo[r + 1]; i++) std::cout << " " << mv[i] << ":" << mp[i] << ":" << (int)md[i] << ":" << (int)mq[i];
        std::cout << "\n";
    }
    mdbg_minimizers_free(m);
    mdbg_reads_free(reads);
    mdbg_destroy(gpu);
    return 0;
}

static int edges_hip(int argc, char **argv)
{
    CreateMdbg g;
    g.parseArgs(argc, argv);
    g._readStats.load(g._outputDir + "/read_stats.txt");
    g._nbPartitions = g._readStats._nbBases / 20000000000ull;          /* as CreateMdbg::createMDBG sets it (graph/CreateMdbg.cpp:290-299) */
    g._nbPartitions = max(g._nbPartitions, g._nbCores);
    g._nbPartitions = max(g._nbPartitions, 1);
    g._nbPartitions = min(g._nbPartitions, 5000);
    const uint32_t k = (uint32_t)g._kminmerSize;
    /* the reference's EdgeIndexer on the vectors in the directory, as indexEdges() runs it */
    CreateMdbg::EdgeIndexer ref(g);
    ref.execute();
    std::vector<u_int128_t> want;
    {
        ifstream f(ref.getOutputFilename(), std::ios::binary);
        u_int128_t e;
        while (f.read((char *)&e, sizeof e)) want.push_back(e);
    }
    fs::remove(ref.getOutputFilename());
    std::sort(want.begin(), want.end());
    /* the library on the same vectors: the table the vectors came from is rebuilt from read_data_corrected.txt (a table with vectors is
     * what mdbg_edge_index takes), its vectors checked against the file's, then indexed */
    mdbg_ctx *gpu = nullptr;
    hipbind::check(nullptr, mdbg_create(0, &gpu), "mdbg_create");
    std::vector<uint32_t> mins;
    std::vector<uint64_t> offs;
    hipbind::parse_minimizer_reads(hipbind::file_bytes(g._outputDir + "/read_data_corrected.txt", true), mins, offs, nullptr);
    mdbg_minimizers *reads = hipbind::upload(gpu, mins, offs);
    mdbg_table *table = nullptr;
    if (!g._isFirstPass) { std::cerr << "edges_hip: run after graph_hip --firstpass (the pass whose table has vectors and needs no previous table)\n"; return 2; }
    hipbind::check(gpu, mdbg_kminmer_count_first(gpu, reads, k, (uint32_t)g._minAbundance, &table), "mdbg_kminmer_count_first");
    uint64_t n = 0;
    int hasVec = 0;
    mdbg_table_info(table, nullptr, &n, nullptr, &hasVec);
    std::vector<uint32_t> vec(n * k);
    hipbind::check(gpu, mdbg_table_to_host(gpu, table, nullptr, vec.data()), "mdbg_table_to_host");
    {
        const std::vector<uint8_t> onDisk = hipbind::file_bytes(g._outputDir + "/kminmerData_min.txt", true);
        std::vector<std::vector<uint32_t>> a(n), b(onDisk.size() / 4 / k);
        for (uint64_t i = 0; i < n; i++) a[i].assign(vec.begin() + i * k, vec.begin() + (i + 1) * k);
        for (size_t i = 0; i < b.size(); i++) { b[i].resize(k); memcpy(b[i].data(), onDisk.data() + i * k * 4, k * 4); }
        std::sort(a.begin(), a.end()); std::sort(b.begin(), b.end());
        if (a != b) { std::cerr << "edges_hip: kminmerData_min.txt is not this table's vectors\n"; return 3; }
    }
    mdbg_table *edges = nullptr;
    uint64_t checksum = 0;
    hipbind::check(gpu, mdbg_edge_index(gpu, table, &edges, &checksum), "mdbg_edge_index");
    uint64_t ne = 0;
    mdbg_table_info(edges, nullptr, &ne, nullptr, nullptr);
    std::vector<uint64_t> keys(ne * 2);
    hipbind::check(gpu, mdbg_table_keys_to_host(gpu, edges, keys.data()), "mdbg_table_keys_to_host");
    std::vector<u_int128_t> got(ne);
    for (uint64_t i = 0; i < ne; i++) got[i] = ((u_int128_t)keys[2 * i + 1] << 64) | keys[2 * i];
    std::sort(got.begin(), got.end());
    const bool same = got == want && ne == ref._nbEdges && checksum == ref._checksum;
    std::cout << "edges_hip: reference EdgeIndexer " << ref._nbEdges << " keys, checksum " << ref._checksum << "; mdbg_edge_index " << ne << " keys, checksum " << checksum
              << "; " << (same ? "equal" : "DIFFERENT") << "\n";
    /* the second half of N2: the reference's UnitigEdgeIndexer (graph/CreateMdbg.hpp:4234-4512) over unitigGraph.nodes.bin -- left by the graph
     * stage that `graph_hip` ran on the library's tables -- against mdbg_unitig_edge_index over the same unitigs */
    bool same_u = true;
    if (fs::exists(g._outputDir + "/unitigGraph.nodes.bin")) {
        CreateMdbg::UnitigEdgeIndexer uref(g);
        uref.execute();
        std::vector<u_int128_t> uwant;
        {
            ifstream f(uref.getOutputFilename(), std::ios::binary);
            u_int128_t e;
            while (f.read((char *)&e, sizeof e)) uwant.push_back(e);
        }
        fs::remove(uref.getOutputFilename());
        std::sort(uwant.begin(), uwant.end());
        const std::vector<uint8_t> nodes = hipbind::file_bytes(g._outputDir + "/unitigGraph.nodes.bin", true);     /* (u32 size; u32 m[size]; u32 unitigIndex)* */
        std::vector<uint32_t> um;
        std::vector<uint64_t> uoff{0};
        for (size_t o = 0; o + 4 <= nodes.size();) {
            uint32_t sz;
            memcpy(&sz, nodes.data() + o, 4); o += 4;
            const size_t base = um.size();
            um.resize(base + sz);
            if (sz) memcpy(um.data() + base, nodes.data() + o, (size_t)sz * 4);
            o += (size_t)sz * 4 + 4;
            uoff.push_back(um.size());
        }
        mdbg_minimizers *unitigs = hipbind::upload(gpu, um, uoff);
        mdbg_table *uedges = nullptr;
        hipbind::check(gpu, mdbg_unitig_edge_index(gpu, unitigs, k, &uedges, nullptr), "mdbg_unitig_edge_index");
        uint64_t nu = 0;
        mdbg_table_info(uedges, nullptr, &nu, nullptr, nullptr);
        std::vector<uint64_t> ukeys(nu * 2);
        hipbind::check(gpu, mdbg_table_keys_to_host(gpu, uedges, ukeys.data()), "mdbg_table_keys_to_host");
        std::vector<u_int128_t> ugot(nu);
        for (uint64_t i = 0; i < nu; i++) ugot[i] = ((u_int128_t)ukeys[2 * i + 1] << 64) | ukeys[2 * i];
        std::sort(ugot.begin(), ugot.end());
        same_u = ugot == uwant && nu == uref._nbEdges;
        std::cout << "edges_hip: reference UnitigEdgeIndexer " << uref._nbEdges << " keys over " << (uoff.size() - 1) << " unitigs; mdbg_unitig_edge_index " << nu << " keys; "
                  << (same_u ? "equal" : "DIFFERENT") << "\n";
        mdbg_table_free(uedges); mdbg_minimizers_free(unitigs);
    }
    mdbg_table_free(edges); mdbg_table_free(table); mdbg_minimizers_free(reads); mdbg_destroy(gpu);
    return same && same_u ? 0 : 1;
}

static int read_selection_hip(int argc, char **argv)
{
    hipbind::ReadSelectionHip().run(argc, argv);    /* Tool::run: parseArgs, execute (above), end -- perf.bin as the reference writes it */
    return 0;
}
