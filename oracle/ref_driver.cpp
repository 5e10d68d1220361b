/*
 * ref_driver.cpp -- thin main() over the REFERENCE's own classes, compiled in place from
 * /root/reference by oracle/Makefile into oracle/_ref/refdrv.
 *
 * TEST INFRASTRUCTURE ONLY.  No reference source is copied: this file only #includes the
 * reference headers where they lie and dispatches to them, the way
 * /root/reference/src/MdbgAssembler.cpp:97-172 dispatches its sub-commands.  It exists so the
 * oracle restatement (mdbg_oracle.c) and the HIP path can be checked against the real
 * reference, and so bench.py can time the reference's CPU path (cpu_baseline.kind="reference").
 *
 * Sub-commands that run reference tools unchanged (same argv as the reference executable):
 *   refdrv readSelection <tmpDir> <out> <input.txt> --threads N --min-read-quality Q [...]
 *   refdrv graph <tmpDir> --threads N [--min-abundance M] [--firstpass]
 *   refdrv contig ... / refdrv toMinspace ...      (producers of the k>4 inputs)
 *   refdrv graph_from_tables <tmpDir> --threads N [--firstpass]
 *        the REST of the reference's `graph` command after its tables exist: CreateMdbg::createGfa() (k <= firstK+1,
 *        graph/CreateMdbg.cpp:527-553, :658-) or computeNextUnitigGraph() (k >= firstK+2, :525) run on tables some
 *        other producer wrote into <tmpDir> -- the hand-over test: mdbg_tool's tables in, the reference's own graph
 *        stage on top, products compared with a pure reference run.  At k >= firstK+2 the table the graph stage
 *        queries in memory (_mdbgNodesLight: isEdgeSupported :3990, removeUnsupportedUnitigs :4156, createEdgeNode
 *        :5013) is filled from the 20-byte records of kminmerData_abundance.txt.
 * Built as oracle/_ref/refdrv_hip (with ref_binding.cpp, linked against libmdbg_hip.so) it also has
 *   refdrv_hip readSelection_hip ... / refdrv_hip graph_hip ...
 *        the reference's own tools with the hot path replaced by calls through include/mdbg_hip.h, as INTEGRATION.md describes.
 *   refdrv_hip fn_corrscan_hip ... / refdrv_hip edges_hip ...
 *        SURVEY 8(f) N1 / N2 at the seams that link here (ref_binding.cpp says which and why)
 * Function-level probes (text on stdin -> text on stdout), used by tests/golden/make_golden.py:
 *   refdrv fn_scan <K> <density> <hpc>     : lines "<seq>"              -> "n v:pos:dir ..."
 *   refdrv fn_scan_notrim <K> <density> <hpc> : same with MinimizerParser::_trimBps = 0 (GenerateGfa's unitig scan)
 *   refdrv fn_purge <firstK> <lastK>       : lines "m0 m1 ..."          -> purged list
 *   refdrv fn_kminmer <k>                  : lines "m0 .. m(k-1)"       -> "rev hi lo c0 .. c(k-1)"
 *   refdrv fn_murmur                       : lines "<u64>"              -> Murmur3_x64_128(&v,8,42)
 *   refdrv fn_lastk <density> <n50> <firstK> <maxK>
 *   refdrv fn_density <density>            : lines "m0 m1 ..."          -> indices kept by Utils::applyDensityThreshold
 *   refdrv fn_corrscan <K> <density> <hpc> : lines "<seq> <qual>"       -> "n v:pos:dir:minq ..." as the correction scan
 *                                            (ReadCorrection::ReadSelectionFunctor) computes them
 */
#include "Commons.hpp"
#include "readSelection/ReadSelection.hpp"
#include "graph/CreateMdbg.hpp"
#include "assembly/GenerateContigs.hpp"
#include "toBasespace/ToMinspace.hpp"
#include "readSelection/ReadCorrection.hpp"

#include <iostream>
#include <sstream>

static std::vector<uint64_t> parse_u64s(const std::string &line)
{
    std::vector<uint64_t> v;
    std::istringstream is(line);
    uint64_t x;
    while (is >> x) v.push_back(x);
    return v;
}

static int fn_scan(int argc, char **argv, size_t trim)
{
    if (argc < 5) return 2;
    size_t K = std::stoul(argv[2]);
    float density = std::stof(argv[3]);
    bool hpc = std::stoi(argv[4]) != 0;
    unordered_set<MinimizerType> rep;
    for (int i = 5; i < argc; i++) rep.insert((MinimizerType)std::stoul(argv[i]));
    MinimizerParser parser(K, density, rep);
    parser._trimBps = trim;                 /* 1 = the constructor's default; 0 as graph/GenerateGfa.hpp:366 sets it */
    EncoderRLE enc;
    std::string line;
    while (std::getline(std::cin, line)) {
        std::string rle;
        vector<u_int64_t> rlePos;
        enc.execute(line.c_str(), line.size(), rle, rlePos, hpc);
        vector<MinimizerType> m;
        vector<u_int32_t> pos;
        vector<u_int8_t> dir;
        parser.parse(rle, m, pos, dir);
        std::cout << m.size() << " " << rle.size();
        for (size_t i = 0; i < m.size(); i++) std::cout << " " << m[i] << ":" << pos[i] << ":" << (int)dir[i];
        std::cout << "\n";
    }
    return 0;
}

static int fn_purge(int argc, char **argv)
{
    if (argc < 4) return 2;
    size_t firstK = std::stoul(argv[2]), lastK = std::stoul(argv[3]);
    std::string line;
    while (std::getline(std::cin, line)) {
        vector<MinimizerType> m;
        for (uint64_t x : parse_u64s(line)) m.push_back((MinimizerType)x);
        vector<MinimizerType> r = Commons::purgePalindrome(m, firstK, lastK);
        for (size_t i = 0; i < r.size(); i++) std::cout << (i ? " " : "") << r[i];
        std::cout << "\n";
    }
    return 0;
}

static int fn_kminmer(int argc, char **argv)
{
    if (argc < 3) return 2;
    std::string line;
    while (std::getline(std::cin, line)) {
        KmerVec vec;
        for (uint64_t x : parse_u64s(line)) vec._kmers.push_back((MinimizerType)x);
        bool rev;
        KmerVec c = vec.normalize(rev);
        u_int128_t h = c.hash128();
        std::cout << (rev ? 1 : 0) << " " << (uint64_t)(h >> 64) << " " << (uint64_t)h;
        for (auto m : c._kmers) std::cout << " " << m;
        std::cout << "\n";
    }
    return 0;
}

static int fn_density(int argc, char **argv)
{
    if (argc < 3) return 2;
    float density = std::stof(argv[2]);
    std::string line;
    while (std::getline(std::cin, line)) {
        vector<MinimizerType> m, mf;
        for (uint64_t x : parse_u64s(line)) m.push_back((MinimizerType)x);
        vector<u_int32_t> pos(m.size()), posf;
        for (size_t i = 0; i < m.size(); i++) pos[i] = (u_int32_t)i;
        vector<u_int8_t> dir(m.size(), 0), qual(m.size(), 0), dirf, qualf;
        Utils::applyDensityThreshold(density, m, pos, dir, qual, mf, posf, dirf, qualf);
        std::cout << mf.size();
        for (size_t i = 0; i < mf.size(); i++) std::cout << " " << posf[i];
        std::cout << "\n";
    }
    return 0;
}

/* The statements of ReadCorrection::ReadSelectionFunctor::operator() (ReadCorrection.hpp:2318-2343) up to the
 * record sink, on the reference's own EncoderRLE / MinimizerParser / getMinQuality.  getMinQuality reads no
 * member, so it is called on raw storage: building a ReadCorrection would pull the whole correction tool in. */
static int fn_corrscan(int argc, char **argv)
{
    if (argc < 5) return 2;
    size_t K = std::stoul(argv[2]);
    float density = std::stof(argv[3]);
    bool hpc = std::stoi(argv[4]) != 0;
    unordered_set<MinimizerType> rep;
    MinimizerParser parser(K, density, rep);
    EncoderRLE enc;
    alignas(16) static unsigned char storage[sizeof(ReadCorrection::ReadSelectionFunctor)];
    auto *f = reinterpret_cast<ReadCorrection::ReadSelectionFunctor *>(storage);
    std::string line;
    while (std::getline(std::cin, line)) {
        std::istringstream is(line);
        std::string seq, qual;
        is >> seq >> qual;
        std::string rle;
        vector<u_int64_t> rlePos;
        enc.execute(seq.c_str(), seq.size(), rle, rlePos, hpc);
        vector<MinimizerType> m;
        vector<u_int32_t> pos;
        vector<u_int8_t> dir;
        parser.parse(rle, m, pos, dir);
        std::cout << m.size();
        for (size_t i = 0; i < m.size(); i++) {
            u_int8_t q = f->getMinQuality(0, seq, qual, rlePos[pos[i]], rlePos[pos[i] + K - 1]);
            std::cout << " " << m[i] << ":" << pos[i] << ":" << (int)dir[i] << ":" << (int)q;
        }
        std::cout << "\n";
    }
    return 0;
}

/* What CreateMdbg::execute() + createMDBG() do around the table production (graph/CreateMdbg.cpp:168-196, :199-232,
 * :515-553), with the tables taken from the files in <tmpDir> instead of being computed. */
static int graph_from_tables(int argc, char **argv)
{
    CreateMdbg g;
    g.parseArgs(argc, argv);
    g._checksum_unitigNodes = 0;
    g._checksum_unitigEdges = 0;
    g._checksum_unitigAbundances = 0;
    g._nbUnitigEdges = 0;
    g._nbUnitigNodes = 0;
    g._mutexes.resize(1000);
    for (size_t i = 0; i < g._mutexes.size(); i++) omp_init_lock(&g._mutexes[i]);
    g._readStats.load(g._outputDir + "/read_stats.txt");
    g._nbPartitions = g._readStats._nbBases / 20000000000ull;
    g._nbPartitions = max(g._nbPartitions, g._nbCores);
    g._nbPartitions = max(g._nbPartitions, 1);
    g._nbPartitions = min(g._nbPartitions, 5000);
    if (g._kminmerSize > g._kminmerSizeFirst + 1) {
        ifstream f(g._outputDir + "/kminmerData_abundance.txt", std::ios::binary);
        u_int64_t n = 0;
        while (true) {
            u_int128_t hash;
            u_int32_t abundance;
            f.read((char *)&hash, sizeof(hash));
            if (f.eof()) break;
            f.read((char *)&abundance, sizeof(abundance));
            g._mdbgNodesLight[hash] = abundance;
            n++;
        }
        Logger::get().debug() << "graph_from_tables: " << n << " records into _mdbgNodesLight";
        g.computeNextUnitigGraph();
    } else {
        g.createGfa();
        Logger::get().debug() << "Checksum unitig nodes:   " << g._checksum_unitigNodes;        /* as CreateMdbg.cpp:574-576 */
        Logger::get().debug() << "Checksum unitig edges:   " << g._checksum_unitigEdges;
        Logger::get().debug() << "Checksum unitig abundance:   " << g._checksum_unitigAbundances;
    }
    for (size_t i = 0; i < g._mutexes.size(); i++) omp_destroy_lock(&g._mutexes[i]);
    g.end();
    return 0;
}

static int fn_murmur()
{
    std::string line;
    while (std::getline(std::cin, line)) {
        uint64_t v = std::stoull(line);
        std::cout << MurmurHash3_x64_128((const char *)&v, sizeof(v), 42) << "\n";
    }
    return 0;
}

#ifdef MDBG_WITH_HIP_BINDING
#include "ref_binding.cpp"      /* refdrv_hip only: readSelection_hip / graph_hip, the reference's tools with the hot path behind the C ABI */
#endif

int main(int argc, char **argv)
{
    if (argc < 2) { std::cerr << "usage: refdrv <sub-command> ...\n"; return 2; }
    std::string cmd = argv[1];
    if (cmd == "fn_scan") return fn_scan(argc, argv, 1);
    if (cmd == "fn_scan_notrim") return fn_scan(argc, argv, 0);
    if (cmd == "fn_purge") return fn_purge(argc, argv);
    if (cmd == "fn_density") return fn_density(argc, argv);
    if (cmd == "fn_corrscan") return fn_corrscan(argc, argv);
    if (cmd == "fn_kminmer") return fn_kminmer(argc, argv);
    if (cmd == "fn_murmur") return fn_murmur();
#ifdef MDBG_WITH_HIP_BINDING
    if (cmd == "fn_corrscan_hip") return fn_corrscan_hip(argc, argv);
#endif
    if (cmd == "fn_lastk") {
        if (argc < 6) return 2;
        std::cout << Commons::computeLastK(std::stof(argv[2]), std::stoul(argv[3]), std::stoul(argv[4]), std::stoul(argv[5])) << "\n";
        return 0;
    }
    /* Tool dispatch: drop argv[1] like MdbgAssembler.cpp:106-110 */
    std::vector<char *> args(argv, argv + argc);
    args.erase(args.begin() + 1);
    int n = argc - 1;
    if (cmd == "readSelection") ReadSelection().run(n, args.data());
    else if (cmd == "graph") CreateMdbg().run(n, args.data());
    else if (cmd == "contig") GenerateContigs().run(n, args.data());
    else if (cmd == "toMinspace") ToMinspace().run(n, args.data());
    else if (cmd == "graph_from_tables") return graph_from_tables(n, args.data());
#ifdef MDBG_WITH_HIP_BINDING
    else if (cmd == "readSelection_hip") return read_selection_hip(n, args.data());
    else if (cmd == "graph_hip") return graph_hip(n, args.data());
    else if (cmd == "edges_hip") return edges_hip(n, args.data());
#endif
    else { std::cerr << "unknown sub-command " << cmd << "\n"; return 2; }
    return 0;
}
