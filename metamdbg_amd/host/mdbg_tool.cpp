// mdbg_tool.cpp -- C++ host side over the C ABI (include/mdbg_hip.h): drop-in producers of the
// reference's on-disk products for the minimizer + k-min-mer path, same argv as the reference's
// sub-commands:
//
//   mdbg_tool readSelection <tmpDir> <outFile> <inputList> --threads N --min-read-quality F
//                           [--output-quality] [--skip-correction]
//       (readSelection/ReadSelection.hpp:113-246, :251-303)  writes read_data_init.txt, read_stats.txt,
//       repetitiveMinimizers.bin and, for HiFi or --skip-correction, read_data_corrected.txt
//   mdbg_tool graph <tmpDir> --threads N [--min-abundance M] [--firstpass]
//       (graph/CreateMdbg.cpp:11-166, :199-598 up to the tables) writes kminmerData_min.txt,
//       kminmerData_abundance.txt (+ _init copies); it stops where the reference goes on to build
//       the graph (createGfa / computeNextUnitigGraph are out of scope).
//
// Both read <tmpDir>/parameters.gz and write <tmpDir>/perf.bin like Tool::end (Commons.hpp:8088-8107).
// Exit status: 0 on success, 1 on any failure (the parent aborts on non-zero, Commons.hpp:2862-2876).
// All compute goes through libmdbg_hip.so; there is no CPU fallback.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/wait.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <string>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <dlfcn.h>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/mdbg_hip.h"
#include "fastx.hpp"
#include "hostfeed.hpp"
#include "records.hpp"

namespace {

// _exit, not exit: with --gpus G a failing rank thread ends the process while its peers may sit inside an RCCL collective
// waiting for it; exit() would run the static destructors of HIP / RCCL under them and can hang, and the parent
// (Utils::executeCommand, Commons.hpp:2855) only needs the non-zero status.  Output files are flushed as they are written.
[[noreturn]] void die(const std::string &msg) {
    fprintf(stderr, "mdbg_tool: %s\n", msg.c_str());
    fflush(nullptr);
    _exit(1);
}

mdbg_ctx *g_ctx = nullptr;
std::vector<mdbg_ctx *> g_more_ctx;      // the other consumers' contexts (readSelection)

// Every output file is written and closed: the process ends here.  Tearing the HIP contexts and their memory pools down block by
// block (mdbg_destroy) took 0.2 s of a 2.6 s run over 50 Gbp; the driver reclaims a process's device memory when it exits.
[[noreturn]] void finish();

// phase timings on stderr when MDBG_TRACE is set
struct Trace {
    double t0;
    static double now() { struct timeval tv; gettimeofday(&tv, nullptr); return tv.tv_sec + 1e-6 * tv.tv_usec; }
    Trace() : t0(now()) {}
    void mark(const char *what) { if (getenv("MDBG_TRACE")) fprintf(stderr, "[mdbg_tool] %8.3f s  %s\n", now() - t0, what); }
} g_trace;
// The process the caller waits for is a thin launcher (main): this one -- the worker, its child -- tells it through a pipe that every
// output file is written and closed, and the launcher exits at once with the status byte.  What the worker still has to do then is
// die: 0.11 - 0.28 s in which the kernel releases its GPU state (tools/exit_cost.py, profiles/round5_k_tool_exit_cost_*.json: whatever
// the tool tears down first), a quarter of a `graph` process at k >= 5 -- it now passes beside the caller's next step instead of in
// front of it.  The worker's stdout / stderr are closed first so that a caller reading them through pipes sees their end.
// MDBG_TOOL_NO_DETACH=1: one process, as before.
int g_done_fd = -1;
void signal_done(int status) {
    if (g_done_fd < 0) return;
    fflush(nullptr);
    const char c = (char)status;
    ssize_t w;
    do w = write(g_done_fd, &c, 1); while (w < 0 && errno == EINTR);
    close(g_done_fd);
    g_done_fd = -1;
    const int nul = open("/dev/null", O_WRONLY);
    if (nul >= 0) { dup2(nul, 1); dup2(nul, 2); if (nul > 2) close(nul); }
}

[[noreturn]] void finish() {
    fflush(nullptr);
    // (measuring aid: what part of the time between "done" and the parent's wait() returning is the first context's teardown)
    if (const char *e = getenv("MDBG_TOOL_EXIT_TRACE")) {
        const int how = atoi(e);       // 1: only say when; 2: the first context destroyed first; 3: every context; 4: hipDeviceReset; 5: idle for 0.2 s; 6: 3 + 4
        if (g_ctx && how == 2) { mdbg_destroy(g_ctx); g_trace.mark("exit trace: the first context destroyed"); }
        if (how == 3 || how == 6) {
            for (mdbg_ctx *c : g_more_ctx) mdbg_destroy(c);
            if (g_ctx) mdbg_destroy(g_ctx);
            g_trace.mark("exit trace: every context destroyed");
        }
        if (how == 4 || how == 6) {
            if (auto reset = (int (*)())dlsym(RTLD_DEFAULT, "hipDeviceReset")) { const int rc = reset(); g_trace.mark(rc ? "exit trace: hipDeviceReset failed" : "exit trace: hipDeviceReset"); }
        }
        if (how == 5) { usleep(200000); g_trace.mark("exit trace: idle for 0.2 s"); }
        fprintf(stderr, "[mdbg_tool] exit trace: clock started at %.6f, _exit at %.6f (epoch seconds)\n", g_trace.t0, Trace::now());
    }
    signal_done(0);
    _exit(0);
}
void check(int rc, const char *what) {
    if (rc != MDBG_OK) die(std::string(what) + ": " + mdbg_last_error(g_ctx));
}

// Appends to <parent of tmpDir>/metaMDBG.log, the file Tool::openLogFile points the reference's logger at (Commons.hpp:8070-8086).
struct LogFile {
    std::ofstream f;
    void open(const std::string &tmpDir) {
        std::string d = tmpDir;
        while (d.size() > 1 && d.back() == '/') d.pop_back();
        const size_t cut = d.find_last_of('/');
        std::string parent = cut == std::string::npos ? std::string() : d.substr(0, cut == 0 ? 1 : cut);
        if (parent.empty()) parent = tmpDir;
        if (!f.is_open()) f.open(parent + "/metaMDBG.log", std::ios::app);       // (asmStep: the second command of the process finds it open)
    }
    void line(const std::string &s) { if (f) { f << s << "\n"; f.flush(); } }
} g_log;

// ---- parameters.gz (pipeline/AssemblyPipeline.hpp:1479-1517 / Commons.hpp:1475-1497) -----------------
struct Parameters {
    size_t minimizerSize = 0, kminmerSize = 0;
    float densityAssembly = 0;
    size_t firstK = 0;
    float spacingMean = 0, kLenMean = 0, kOvlMean = 0;
    size_t prevK = 0, lastK = 0, meanReadLength = 0;
    float densityCorrection = 0;
    bool hpc = false;
    int dataType = 0;
    size_t snpmerSize = 0;

    void load(const std::string &path) {
        gzFile f = gzopen(path.c_str(), "rb");
        if (!f) die("cannot open " + path);
        auto rd = [&](void *p, unsigned n) { if (gzread(f, p, n) != (int)n) die("short read of " + path); };
        rd(&minimizerSize, sizeof(size_t)); rd(&kminmerSize, sizeof(size_t)); rd(&densityAssembly, sizeof(float));
        rd(&firstK, sizeof(size_t)); rd(&spacingMean, sizeof(float)); rd(&kLenMean, sizeof(float)); rd(&kOvlMean, sizeof(float));
        rd(&prevK, sizeof(size_t)); rd(&lastK, sizeof(size_t)); rd(&meanReadLength, sizeof(size_t));
        rd(&densityCorrection, sizeof(float)); rd(&hpc, sizeof(bool)); rd(&dataType, sizeof(int)); rd(&snpmerSize, sizeof(size_t));
        gzclose(f);
    }
};

std::vector<uint8_t> read_file(const std::string &path, bool required = true) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { if (required) die("File not found: " + path); return {}; }
    f.seekg(0, std::ios::end);
    std::vector<uint8_t> v((size_t)f.tellg());
    f.seekg(0);
    if (!v.empty()) f.read((char *)v.data(), (std::streamsize)v.size());
    return v;
}

// Utils::computeN50 / computeMeanLength / Commons::computeLastK (Commons.hpp:2291-2336, :1726-1741)
uint32_t compute_n50(std::vector<uint32_t> len) {
    if (len.empty()) return 0;
    std::sort(len.begin(), len.end(), std::greater<uint32_t>());
    std::vector<uint64_t> cum(len.size());
    uint64_t c = 0;
    for (size_t i = 0; i < len.size(); i++) { c += len[i]; cum[i] = c; }
    const size_t n = len.size();
    uint32_t n50 = len[0];
    const uint64_t half = cum[n - 1] / 2;
    for (size_t i = 0; i < n; i++) if (cum[n - 1 - i] < half) { n50 = len[n - 1 - i]; break; }
    return n50;
}
uint32_t compute_mean_length(const std::vector<uint32_t> &len) {
    long double sum = 0, n = 0;
    for (uint32_t l : len) { sum += l; n += 1; }
    return (uint32_t)(uint64_t)(sum / n);
}
int compute_last_k(float density, size_t n50, size_t firstK, size_t maxK) {
    size_t lastK = (size_t)(n50 * density * 2.0f);
    if (maxK > 0) lastK = maxK;
    lastK = std::max(lastK, firstK + 2);
    return (int)lastK;
}

void write_perf(const std::string &tmpDir) {   // Tool::end
    struct rusage r;
    getrusage(RUSAGE_SELF, &r);
    double cpu = r.ru_utime.tv_sec + r.ru_stime.tv_sec + 1e-6 * (r.ru_utime.tv_usec + r.ru_stime.tv_usec);
    double rss = (double)r.ru_maxrss * 1024.0 / 1024.0 / 1024.0 / 1024.0;   // ru_maxrss is kB -> GB
    std::ofstream f(tmpDir + "/perf.bin", std::ios::binary);
    f.write((const char *)&cpu, sizeof(cpu));
    f.write((const char *)&rss, sizeof(rss));
}

// ---- argv ---------------------------------------------------------------------------------------------
struct Args {
    std::vector<std::string> pos;
    int threads = 1;
    float minReadQuality = 0;
    bool skipCorrection = false, outputQuality = false, firstPass = false;
    uint32_t minAbundance = 0;
    size_t batchBases = (size_t)32 << 20;  // bytes of input per device batch (not a reference flag)
    int gpus = 1;                          // contexts / devices the work is spread over (not a reference flag)
    bool verify = true;                    // graph --gpus G: rank 0 repeats the pass alone and the job compares (not a reference flag)
    bool thenGraph = false;                // asmStep: graph --firstpass follows in this process, on the minimizers still on the device (not a reference flag)
};
Args parse_args(int argc, char **argv, int first) {
    Args a;
    for (int i = first; i < argc; i++) {
        std::string s = argv[i];
        auto val = [&]() -> std::string { if (i + 1 >= argc) die("missing value for " + s); return argv[++i]; };
        if (s == "--threads") a.threads = atoi(val().c_str());
        else if (s == "--min-read-quality") a.minReadQuality = (float)atof(val().c_str());
        else if (s == "--min-abundance") a.minAbundance = (uint32_t)atoi(val().c_str());
        else if (s == "--skip-correction") a.skipCorrection = true;
        else if (s == "--output-quality") a.outputQuality = true;
        else if (s == "--firstpass") a.firstPass = true;
        else if (s == "--batch-bases") a.batchBases = (size_t)atoll(val().c_str());
        else if (s == "--gpus") a.gpus = std::max(1, std::min(64, atoi(val().c_str())));
        else if (s == "--no-verify") a.verify = false;
        else if (s == "--verify") a.verify = true;
        else if (s.rfind("--", 0) == 0) die("unknown flag " + s);
        else a.pos.push_back(s);
    }
    return a;
}

// ---- batches of parsed reads ---------------------------------------------------------------------------------
using mdbg_host::ReadBatch;

std::vector<std::string> read_input_list(const std::string &inputList) {
    std::ifstream lst(inputList);
    if (!lst) die("File not found: " + inputList);
    std::vector<std::string> files;
    std::string path;
    while (std::getline(lst, path)) if (!path.empty()) files.push_back(path);
    return files;
}

// Feeds `fn` with batches of reads, in read order, from the files of inputList; at most maxReadsPerFile + 1 reads
// of each file when maxReadsPerFile > 0 (the reference's `_maxReads` check, Commons.hpp:5873).  Parsing runs on
// `threads` workers into page-locked buffers (hostfeed.hpp) while fn -- upload, kernels, record writing -- runs here.
void for_each_batch(const std::string &inputList, size_t batchBases, int threads, uint64_t maxReadsPerFile,
                    const std::function<void(ReadBatch &)> &fn) {
    auto alloc = [](size_t n) -> void * { void *p = nullptr; return mdbg_host_alloc(g_ctx, n, &p) == MDBG_OK ? p : nullptr; };
    auto release = [](void *p) { mdbg_host_free(g_ctx, p); };
    try {
        mdbg_host::ReadFeeder feeder(read_input_list(inputList), batchBases, threads, maxReadsPerFile, alloc, release);
        while (ReadBatch *b = feeder.next()) {
            fn(*b);
            feeder.recycle(b);
        }
    } catch (const std::exception &e) { die(e.what()); }
}

// The same one batch ahead: stage(b) starts bringing batch i+1 to the device (an upload that runs beside the kernels) before
// process(handle, b) works on batch i; the batch's page-locked buffer is recycled after process.  The feeder is taken apart on a thread
// of its own (un-pinning its buffers and unmapping the input are not worth waiting for).
template <typename Handle>
void for_each_batch_ahead(const std::string &inputList, size_t batchBases, int threads, uint64_t maxReadsPerFile,
                          const std::function<Handle(ReadBatch &)> &stage, const std::function<void(Handle, ReadBatch &)> &process) {
    auto alloc = [](size_t n) -> void * { void *p = nullptr; return mdbg_host_alloc(g_ctx, n, &p) == MDBG_OK ? p : nullptr; };
    auto release = [](void *p) { mdbg_host_free(g_ctx, p); };
    try {
        std::unique_ptr<mdbg_host::ReadFeeder> feeder(new mdbg_host::ReadFeeder(read_input_list(inputList), batchBases, threads, maxReadsPerFile, alloc, release));
        ReadBatch *nb = feeder->next();
        Handle nh{};
        if (nb) nh = stage(*nb);
        while (nb) {
            ReadBatch *cb = nb;
            Handle ch = nh;
            nb = feeder->next();
            if (nb) nh = stage(*nb);
            process(ch, *cb);
            feeder->recycle(cb);
        }
        std::thread([f = feeder.release()] { delete f; }).detach();
    } catch (const std::exception &e) { die(e.what()); }
}

// a parsed batch -> reads in HBM: 2-bit words when the feeder packed the chunk, ASCII (packed on the device) otherwise
void check_on(mdbg_ctx *ctx, int rc, const char *what) {
    if (rc) die(std::string(what) + ": " + mdbg_last_error(ctx));
}

mdbg_reads *upload_batch(mdbg_ctx *ctx, ReadBatch &b, bool withQual) {
    mdbg_reads *reads = nullptr;
    if (b.packed) {
        check_on(ctx, mdbg_reads_from_packed(ctx, b.words(), b.wordOff.data(), b.lens.data(), b.n(), &reads), "mdbg_reads_from_packed");
        if (withQual && b.hasQual) check_on(ctx, mdbg_reads_attach_qualities(ctx, reads, b.quals, b.offsets.data()), "mdbg_reads_attach_qualities");
        if (!b.odd.empty())
            check_on(ctx, mdbg_reads_mark_ascii(ctx, reads, b.odd.data(), (uint32_t)b.odd.size(), b.oddBases.data(), b.oddOff.data()), "mdbg_reads_mark_ascii");
    } else {
        check_on(ctx, mdbg_reads_from_ascii(ctx, b.bases, withQual && b.hasQual ? b.quals : nullptr, b.offsets.data(), b.n(), &reads), "mdbg_reads_from_ascii");
    }
    return reads;
}

mdbg_scan_params scan_params(const Parameters &P, float density, const std::vector<uint32_t> &rep, float minQ, bool filters) {
    mdbg_scan_params p{};
    p.minimizer_size = (uint32_t)P.minimizerSize;
    p.density = density;
    p.hpc = P.hpc ? 1 : 0;
    p.min_read_quality = minQ;
    p.repetitive = rep.empty() ? nullptr : rep.data();
    p.n_repetitive = (uint32_t)rep.size();
    p.apply_read_filters = filters ? 1 : 0;
    return p;
}

[[noreturn]] void graph_main(Args a, const std::string &dir, mdbg_minimizers *resident);

// ---- readSelection --------------------------------------------------------------------------------------------
// (asmStep = true: `mdbg_tool asmStep <the arguments of readSelection> [--min-abundance M]` -- this command and `graph --firstpass` in ONE
// process: the reference's pipeline runs them as two children one after the other, pipeline/AssemblyPipeline.hpp:716-740 and :763-792; here the
// second finds the library context alive and the corrected minimizers still on the device instead of creating one and parsing
// read_data_corrected.txt back.  Every file of both commands is written as by the two of them.)
int run_read_selection(int argc, char **argv, bool asmStep = false) {
    Args a = parse_args(argc, argv, 2);
    a.thenGraph = asmStep;
    if (a.pos.size() != 3) die(std::string("usage: mdbg_tool ") + (asmStep ? "asmStep" : "readSelection") + " <tmpDir> <outFile> <inputList> --threads N --min-read-quality F [--skip-correction]" +
                               (asmStep ? " [--min-abundance M]" : ""));
    if (asmStep && a.gpus > 1) die("asmStep runs on one device; with --gpus G use the two commands readSelection and graph");
    const std::string tmpDir = a.pos[0], outFile = a.pos[1], inputList = a.pos[2];
    Parameters P;
    P.load(tmpDir + "/parameters.gz");
    g_log.open(tmpDir);
    g_log.line("mdbg_tool readSelection (MI355X) " + inputList);
    check(mdbg_create(0, &g_ctx), "mdbg_create");
    g_trace.mark("context created");

    // repetitive minimizers (ReadSelection.hpp:497-561): HiFi writes an empty file
    std::vector<uint32_t> rep;
    {
        std::ofstream repFile(tmpDir + "/repetitiveMinimizers.bin", std::ios::binary);
        if (!P.hpc) {
            // every batch's minimizer values are counted where they are (mdbg_census_*): nothing but the pick comes back
            mdbg_census *census = nullptr;
            check(mdbg_census_create(g_ctx, &census), "mdbg_census_create");
            // (the census scan never looks at qualities -- CountMinimizerFunctor, ReadSelection.hpp:565-625 -- so they do not travel)
            for_each_batch_ahead<mdbg_reads *>(inputList, a.batchBases, a.threads, 1000000,
                [&](ReadBatch &b) -> mdbg_reads * {
                    mdbg_reads *reads = nullptr;
                    if (b.packed) {
                        check(mdbg_reads_from_packed_async(g_ctx, b.words(), b.wordOff.data(), b.lens.data(), b.n(), &reads), "mdbg_reads_from_packed_async");
                        if (!b.odd.empty())
                            check(mdbg_reads_mark_ascii(g_ctx, reads, b.odd.data(), (uint32_t)b.odd.size(), b.oddBases.data(), b.oddOff.data()), "mdbg_reads_mark_ascii");
                    } else reads = upload_batch(g_ctx, b, false);
                    return reads;
                },
                [&](mdbg_reads *reads, ReadBatch &) {
                    mdbg_minimizers *mins = nullptr;
                    mdbg_scan_params p = scan_params(P, P.densityCorrection, {}, 0, false);
                    check(mdbg_scan(g_ctx, reads, &p, &mins), "mdbg_scan");
                    check(mdbg_census_add(g_ctx, census, mins), "mdbg_census_add");
                    mdbg_minimizers_free(mins);
                    check(mdbg_reads_wait(g_ctx, reads), "mdbg_reads_wait");       // before the batch's buffer is recycled
                    mdbg_reads_free(reads);
                });
            g_trace.mark("census done");
            uint32_t cap = 1u << 16;
            rep.resize(cap);
            check(mdbg_census_top(g_ctx, census, rep.data(), &cap), "mdbg_census_top");
            rep.resize(cap);
            mdbg_census_free(census);
            // test hook: ties among equally frequent minimizers are broken arbitrarily by the reference
            // (std::sort on counts, ReadSelection.hpp:522-524); a fixture can pin the reference's pick
            if (const char *forced = getenv("MDBG_TOOL_REPETITIVE")) {
                std::vector<uint8_t> raw = read_file(forced);
                rep.assign(raw.size() / 4, 0);
                if (!raw.empty()) memcpy(rep.data(), raw.data(), rep.size() * 4);
            }
            repFile.write((const char *)rep.data(), (std::streamsize)(rep.size() * 4));
        }
        repFile.close();
        if (!repFile) die("writing " + tmpDir + "/repetitiveMinimizers.bin failed");
    }

    // main pass: read_data_init.txt in read order (ReadSelection.hpp:386-491)
    const int outFd = open(outFile.c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
    if (outFd < 0) die("cannot write " + outFile);
    std::vector<uint32_t> allReadSizes;
    uint64_t nbKmers = 0, nbBases = 0, nbSelected = 0;
    long double qualitySum = 0, qualityN = 0;
    const bool needCorrected = P.hpc || a.skipCorrection;
    if (a.thenGraph && !needCorrected) die("asmStep: without --skip-correction an ONT read set goes through the reference's read correction before `graph` (not part of this tool)");
    std::vector<std::pair<size_t, mdbg_minimizers *>> purgedResident;     // asmStep: the purged groups, by the number of their first batch (guarded by fifoMu)
    // Batches are turned into host arrays by consumer threads, each with its own library context (stream, pool), so the
    // upload of one batch can overlap the scan and the download of another; a writer thread builds the records and
    // writes them in batch order while later batches are on the device.  One consumer is the default: at 28 GB/s of
    // FASTA the pass is bound by the parser workers and the page-locked buffers, and a second consumer measured no
    // faster (profiles/r01f_tool_consumers.json); MDBG_TOOL_CONSUMERS=2..4 is there for slower-to-scan inputs.
    // the arrays of one batch in ONE page-locked slab (the download is plain DMA, nothing is zero-filled); slabs go round between
    // the consumers and the writer
    struct HostBatch {
        uint64_t seq = 0;
        uint32_t n = 0; uint64_t t = 0;
        char *slab = nullptr; size_t cap = 0;
        uint64_t *off = nullptr; uint32_t *m = nullptr, *pos = nullptr, *len = nullptr; uint8_t *dir = nullptr, *qual = nullptr, *flags = nullptr; float *meanQ = nullptr;
        void shape(uint32_t n_, uint64_t t_) {
            n = n_; t = t_;
            const size_t need = ((size_t)n + 1) * 8 + t * 8 + (size_t)n * 8 + ((t + 7) & ~(size_t)7) * 2 + (((size_t)n + 7) & ~(size_t)7) + 64;
            if (need > cap) {
                if (slab) mdbg_host_free(g_ctx, slab);
                cap = need + need / 4;
                void *p = nullptr;
                check(mdbg_host_alloc(g_ctx, cap, &p), "mdbg_host_alloc");
                slab = (char *)p;
            }
            char *q = slab;
            off = (uint64_t *)q; q += ((size_t)n + 1) * 8;
            m = (uint32_t *)q; q += t * 4;
            pos = (uint32_t *)q; q += t * 4;
            len = (uint32_t *)q; q += (size_t)n * 4;
            meanQ = (float *)q; q += (size_t)n * 4;
            dir = (uint8_t *)q; q += (t + 7) & ~(size_t)7;
            qual = (uint8_t *)q; q += (t + 7) & ~(size_t)7;
            flags = (uint8_t *)q;
        }
        // offsets and values only (the purge pass)
        void shape_values(uint32_t n_, uint64_t t_) {
            n = n_; t = t_;
            const size_t need = ((size_t)n + 1) * 8 + t * 4 + 64;
            if (need > cap) {
                if (slab) mdbg_host_free(g_ctx, slab);
                cap = need + need / 4;
                void *p = nullptr;
                check(mdbg_host_alloc(g_ctx, cap, &p), "mdbg_host_alloc");
                slab = (char *)p;
            }
            off = (uint64_t *)slab;
            m = (uint32_t *)(slab + ((size_t)n + 1) * 8);
            pos = len = nullptr; dir = qual = flags = nullptr; meanQ = nullptr;
        }
    };
    std::vector<HostBatch *> spareBatches;        // guarded by fifoMu
    // The purge pass copies GROUPS of batches back (see there), each into one slab several times a batch's: taking the main pass's slabs
    // and growing them on the way (free + allocate page-locked memory, 2 - 4 ms a time) was 0.11 s of the 0.19 s the pass took over 50 Gbp.
    // A helper allocates them beside the main pass, as soon as the first batch has shown how large a batch is.
    constexpr size_t GROUP = 32;
    std::vector<HostBatch *> groupSlabs;          // guarded by fifoMu
    std::atomic<bool> groupSlabsStop{false};
    std::thread groupSlabHelper;
    std::once_flag groupSlabOnce;
    struct Kept { uint64_t seq; mdbg_ctx *ctx; mdbg_minimizers *mins; };
    std::vector<Kept> kept;                 // device-resident minimizer reads, purged once N50 is known
    // --gpus G: one consumer per device; batches go to the consumers in turn, every batch is scanned (and later purged) where it
    // landed, and the ordered writer, N50 and read_stats.txt see the batches in read order whatever device produced them
    // (contiguous read ranges, stats reduced on the host).
    // two consumers per device by default: with the parsers packing on 12 threads it is the consumer -- upload, scan, download, hand-over,
    // per batch of a few thousand reads -- that a 50 Gbp pass waits for (round 3: the parsers were idle 60 % of the time)
    int nConsumers = std::max(1, a.gpus) * (a.threads >= 8 ? 2 : 1);
    if (const char *e = getenv("MDBG_TOOL_CONSUMERS")) nConsumers = std::max(std::max(1, a.gpus), std::min(4 * std::max(1, a.gpus), atoi(e)));
    // how many: what the purge pass can have in flight (six groups' pieces waiting for the writers, one being filled, per consumer) --
    // of a long input.  A short one (10 Gbp: ten groups) gets a third of its groups' worth: pinning 200 MB beside a main pass of
    // 0.25 s measured no faster than growing slabs on the way (0.53 against 0.50 s inside the tool, ten runs each, noise as large)
    size_t groupSlabsWanted = 0;
    {
        uint64_t inputBytes = 0;
        for (const std::string &f : read_input_list(inputList)) { struct stat st; if (stat(f.c_str(), &st) == 0) inputBytes += (uint64_t)st.st_size; }
        const uint64_t groups = inputBytes / ((uint64_t)std::max<size_t>(1, a.batchBases) * GROUP);
        groupSlabsWanted = (size_t)std::min<uint64_t>(7 * (uint64_t)nConsumers + 2, groups * (uint64_t)nConsumers / 3);
    }
    std::vector<mdbg_ctx *> ctxs{g_ctx};
    for (int i = 1; i < nConsumers; i++) {
        mdbg_ctx *c = nullptr;
        if (mdbg_create(i % std::max(1, a.gpus), &c) != MDBG_OK) die(std::string("mdbg_create(device ") + std::to_string(i % a.gpus) + "): " + mdbg_last_error(nullptr));
        ctxs.push_back(c);
        g_more_ctx.push_back(c);
    }

    std::map<uint64_t, HostBatch *> pending;     // finished batches waiting for their turn at the writer
    uint64_t nextWrite = 0;
    std::mutex fifoMu;
    std::condition_variable fifoCv;
    bool fifoDone = false;
    // The record bytes of a batch -- `u32 n; u8 circ; u32 m[n]; u32 pos[n]; u8 dir[n]; u8 qual[n]; f32 meanQ; u32 len` per read -- are
    // built by a few builder threads and written where they belong in the file (pwrite: the size of every earlier batch is known as
    // soon as it has been scanned); one thread writing 1.9 GB of a 50 Gbp read set alone finished 0.45 s behind the consumers.  The
    // statistics (long-double sums: order matters) are accumulated by one thread in read order behind the builders.
    std::map<uint64_t, HostBatch *> buildQ;             // by batch number: a builder takes the lowest one whose place in the file is known
    std::map<uint64_t, uint64_t> sizeOf, offOf;         // batch -> record bytes / file offset
    uint64_t prefSeq = 0, prefOff = 0;
    bool buildDone = false;
    size_t inFlight = 0;                                // batches between the consumers and the statistics thread
    const uint64_t inFlightWindow = 12 + 2 * (uint64_t)nConsumers;
    auto register_size = [&](uint64_t seq, uint64_t bytes) {     // fifoMu held
        sizeOf[seq] = bytes;
        for (auto it = sizeOf.find(prefSeq); it != sizeOf.end(); it = sizeOf.find(prefSeq)) {
            offOf[prefSeq] = prefOff;
            prefOff += it->second;
            sizeOf.erase(it);
            prefSeq++;
        }
    };
    auto build = [&] {
        std::vector<char> rec;
        for (;;) {
            HostBatch *hb = nullptr;
            uint64_t at = 0;
            {
                std::unique_lock<std::mutex> lk(fifoMu);
                // never hold a batch while waiting for its offset (round-3 ADVICE): the offset needs the sizes of all earlier batches, and
                // with every builder parked on a later batch the one the statistics thread is waiting for sat in the queue unbuilt
                auto placed = [&]() -> std::map<uint64_t, HostBatch *>::iterator {
                    for (auto it = buildQ.begin(); it != buildQ.end(); ++it) if (offOf.count(it->first)) return it;
                    return buildQ.end();
                };
                fifoCv.wait(lk, [&] { return placed() != buildQ.end() || (buildDone && buildQ.empty()); });
                auto it = placed();
                if (it == buildQ.end()) return;
                hb = it->second;
                buildQ.erase(it);
                at = offOf[hb->seq];
                offOf.erase(hb->seq);
            }
            rec.resize(hb->t * 10 + (size_t)hb->n * 13);
            char *dst = rec.data();
            for (uint32_t r = 0; r < hb->n; r++) {
                const uint64_t s0 = hb->off[r];
                const uint32_t k = (uint32_t)(hb->off[r + 1] - s0);
                memcpy(dst, &k, 4); dst[4] = 0; dst += 5;
                memcpy(dst, hb->m + s0, (size_t)k * 4); dst += (size_t)k * 4;
                memcpy(dst, hb->pos + s0, (size_t)k * 4); dst += (size_t)k * 4;
                memcpy(dst, hb->dir + s0, k); dst += k;
                memcpy(dst, hb->qual + s0, k); dst += k;
                memcpy(dst, &hb->meanQ[r], 4); memcpy(dst + 4, &hb->len[r], 4); dst += 8;
            }
            for (size_t done = 0; done < rec.size();) {
                const ssize_t w = pwrite(outFd, rec.data() + done, rec.size() - done, (off_t)(at + done));
                if (w < 0) { if (errno == EINTR) continue; die("write to " + outFile + " failed"); }
                done += (size_t)w;
            }
            {
                std::lock_guard<std::mutex> lk(fifoMu);
                pending.emplace(hb->seq, hb);
            }
            fifoCv.notify_all();
        }
    };
    std::vector<std::thread> builders;
    for (int i = 0, nb = std::max(1, std::min(4, a.threads / 4)); i < nb; i++) builders.emplace_back(build);
    std::thread watchdog;
    std::atomic<bool> watchStop{false};
    if (const char *e = getenv("MDBG_TOOL_WATCHDOG_S")) {        // debugging aid: the state of the pipeline every so many seconds
        const int every = std::max(1, atoi(e));
        watchdog = std::thread([&, every] {
            for (int t = 0; !watchStop.load(); t++) {
                std::this_thread::sleep_for(std::chrono::milliseconds(100));
                if (t % (10 * every) != 10 * every - 1) continue;
                std::lock_guard<std::mutex> lk(fifoMu);
                fprintf(stderr, "[mdbg_tool watchdog] buildQ %zu pending %zu inFlight %zu nextWrite %llu prefSeq %llu sizeOf %zu offOf %zu spare %zu\n", buildQ.size(),
                        pending.size(), inFlight, (unsigned long long)nextWrite, (unsigned long long)prefSeq, sizeOf.size(), offOf.size(), spareBatches.size());
                if (!sizeOf.empty()) fprintf(stderr, "    first registered-but-unplaced batch %llu\n", (unsigned long long)sizeOf.begin()->first);
                if (!buildQ.empty()) fprintf(stderr, "    buildQ front %llu\n", (unsigned long long)buildQ.begin()->first);
            }
        });
    }
    std::thread writer([&] {
        for (;;) {
            HostBatch *hb = nullptr;
            {
                std::unique_lock<std::mutex> lk(fifoMu);
                fifoCv.wait(lk, [&] { return pending.count(nextWrite) || (fifoDone && pending.empty()); });
                auto it = pending.find(nextWrite);
                if (it == pending.end()) return;
                hb = it->second;
                pending.erase(it);
                nextWrite++;
            }
            fifoCv.notify_all();
            for (uint32_t r = 0; r < hb->n; r++) {
                const uint32_t k = (uint32_t)(hb->off[r + 1] - hb->off[r]);
                allReadSizes.push_back(hb->len[r]);
                nbSelected += k;
                nbKmers += (uint64_t)((size_t)hb->len[r] - P.minimizerSize + 1);   // size_t arithmetic as in :480
                nbBases += hb->len[r];
                if (!(hb->flags[r] & MDBG_READ_LOW_QUALITY)) { qualitySum += hb->meanQ[r]; qualityN += 1; }   // :911-914
            }
            {
                std::lock_guard<std::mutex> lk(fifoMu);
                spareBatches.push_back(hb);                  // its slab serves a later batch
                inFlight--;
            }
            fifoCv.notify_all();
        }
    });

    double tUpload = 0, tScan = 0, tDownload = 0, tQueue = 0, tWait = 0;
    uint64_t nBatches = 0;
    {
        // (nullptr on failure: the feeder's workers -- many threads, page-locking side by side -- throw into its error path; `check` would
        // have them all write the shared context's error string and _exit at once)
        auto alloc = [](size_t n) -> void * { void *p = nullptr; return mdbg_host_alloc(g_ctx, n, &p) == MDBG_OK ? p : nullptr; };
        auto release = [](void *p) { mdbg_host_free(g_ctx, p); };
        std::unique_ptr<mdbg_host::ReadFeeder> feeder;
        try { feeder.reset(new mdbg_host::ReadFeeder(read_input_list(inputList), a.batchBases, a.threads, 0, alloc, release, nConsumers)); }
        catch (const std::exception &e) { die(e.what()); }
        std::mutex feedMu, statMu;
        uint64_t nextSeq = 0;
        auto consume = [&](int ci) {
            mdbg_ctx *ctx = ctxs[(size_t)ci];
            double up = 0, sc = 0, dn = 0, qu = 0, wt = 0;
            uint64_t nb = 0;
            // One batch ahead: the upload of batch i+1 is queued (mdbg_reads_from_packed_async: the context's upload stream, a copy
            // engine) before batch i is scanned and its minimizers come back, so the link carries reads in one direction and
            // minimizers in the other while the kernels run.  Packed chunks take this route, with or without qualities; chunks
            // delivered as ASCII (a character with bit 3 set) are uploaded synchronously as before.
            struct Staged { ReadBatch *b = nullptr; uint64_t seq = 0; mdbg_reads *reads = nullptr; bool live = false; };
            auto stage = [&]() -> Staged {
                Staged st;
                const double tw = g_trace.now();
                try {
                    std::lock_guard<std::mutex> g(feedMu);       // batches leave the feeder in read order
                    st.b = feeder->next();
                    st.seq = nextSeq++;
                } catch (const std::exception &e) { die(e.what()); }
                if (!st.b) return st;
                st.live = true;
                if (st.seq == 0) g_trace.mark("first batch parsed");
                const double t0 = g_trace.now();
                if (st.b->packed) {
                    check_on(ctx, mdbg_reads_from_packed_async(ctx, st.b->words(), st.b->wordOff.data(), st.b->lens.data(), st.b->n(), &st.reads),
                             "mdbg_reads_from_packed_async");
                    if (st.b->hasQual)
                        check_on(ctx, mdbg_reads_attach_qualities_async(ctx, st.reads, st.b->quals, st.b->offsets.data()), "mdbg_reads_attach_qualities_async");
                    if (!st.b->odd.empty())         // the few reads with an N or lower case, again as characters (a small synchronous copy)
                        check_on(ctx, mdbg_reads_mark_ascii(ctx, st.reads, st.b->odd.data(), (uint32_t)st.b->odd.size(), st.b->oddBases.data(),
                                                            st.b->oddOff.data()), "mdbg_reads_mark_ascii");
                } else {
                    st.reads = upload_batch(ctx, *st.b, true);
                    feeder->recycle(st.b);                       // the page-locked buffer is free again once the upload is done
                    st.b = nullptr;
                }
                wt += t0 - tw; up += g_trace.now() - t0;
                return st;
            };
            Staged next = stage();
            while (next.live) {
                Staged cur = next;
                next = stage();
                const uint64_t seq = cur.seq;
                const double t1 = g_trace.now();
                mdbg_minimizers *mins = nullptr;
                mdbg_scan_params p = scan_params(P, P.densityAssembly, rep, a.minReadQuality, true);
                check_on(ctx, mdbg_scan(ctx, cur.reads, &p, &mins), "mdbg_scan");
                if (cur.b) {                                     // the scan has read the words: the upload is long done
                    check_on(ctx, mdbg_reads_wait(ctx, cur.reads), "mdbg_reads_wait");
                    feeder->recycle(cur.b);
                }
                mdbg_reads_free(cur.reads);
                const double t2 = g_trace.now();
                HostBatch *hb = nullptr;
                {
                    std::lock_guard<std::mutex> lk(fifoMu);
                    if (!spareBatches.empty()) { hb = spareBatches.back(); spareBatches.pop_back(); }
                }
                if (!hb) hb = new HostBatch();
                hb->seq = seq;
                {
                    uint32_t bn; uint64_t bt;
                    mdbg_minimizers_info(mins, &bn, &bt);
                    hb->shape(bn, bt);
                    if (needCorrected && groupSlabsWanted && !getenv("MDBG_TOOL_NO_GROUP_SLABS"))       // (the variable: A/B of the helper)
                        std::call_once(groupSlabOnce, [&, bn, bt] {
                            // a consumer's share of a group (batches are handed out as consumers come free: a little more than GROUP / consumers)
                            const size_t share = GROUP / (size_t)nConsumers + 2;
                            const uint64_t rn = (uint64_t)share * bn, rt = (uint64_t)share * bt;
                            groupSlabHelper = std::thread([&, rn, rt] {
                                for (size_t i = 0; i < groupSlabsWanted && !groupSlabsStop.load(); i++) {
                                    HostBatch *g = new HostBatch();
                                    g->shape_values((uint32_t)std::min<uint64_t>(rn, 0xFFFFFFFFu), rt);
                                    std::lock_guard<std::mutex> lk(fifoMu);
                                    groupSlabs.push_back(g);
                                }
                            });
                        });
                }
                check_on(ctx, mdbg_minimizers_to_host(ctx, mins, hb->off, hb->m, hb->pos, hb->dir, hb->qual, hb->len, hb->meanQ, hb->flags), "to_host");
                const double t3 = g_trace.now();
                {
                    std::unique_lock<std::mutex> lk(fifoMu);
                    if (needCorrected) kept.push_back(Kept{seq, ctx, mins});
                    register_size(seq, hb->t * 10 + (uint64_t)hb->n * 13);
                    fifoCv.notify_all();        // builders may be waiting for exactly this size to learn their offsets (and this thread may
                                                // be about to wait itself: a wake-up left for after the wait below never comes)
                    // Bounded by a WINDOW OF BATCH NUMBERS above the one the statistics thread is waiting for, not by a count (round-3
                    // ADVICE: with three or more consumers, twelve later batches could fill the count while batch `nextWrite` sat
                    // staged-ahead and unscanned with a consumer parked right here).  A consumer's staged batch always comes after the one
                    // it is holding, so whoever holds batch nextWrite -- scanned or staged -- is never the one waiting: no cycle.
                    fifoCv.wait(lk, [&] { return seq < nextWrite + inFlightWindow; });
                    inFlight++;
                    buildQ.emplace(seq, hb);
                }
                if (!needCorrected) mdbg_minimizers_free(mins);
                fifoCv.notify_all();
                const double t4 = g_trace.now();
                sc += t2 - t1; dn += t3 - t2; qu += t4 - t3; nb++;
            }
            std::lock_guard<std::mutex> g(statMu);
            tWait += wt; tUpload += up; tScan += sc; tDownload += dn; tQueue += qu; nBatches += nb;
        };
        g_trace.mark("feeder started");
        std::vector<std::thread> consumers;
        for (int i = 1; i < nConsumers; i++) consumers.emplace_back(consume, i);
        consume(0);
        for (auto &t : consumers) t.join();
        g_trace.mark("last batch scanned and handed to the writer");
        // the feeder is taken apart behind the purge pass, not in front of it: un-pinning its 27 buffers and unmapping the input cost
        // 0.45 s of a 1.85 s run when it sat here (and as much under the process's exit when it was simply left to the system)
        std::thread([f = feeder.release()] { delete f; }).detach();
    }
    if (getenv("MDBG_TRACE"))
        fprintf(stderr, "[mdbg_tool] %llu batches on %d consumer(s), summed over them: waiting for the feeder %.3f s, upload (queued ahead when packed) %.3f s, scan %.3f s, "
                        "download %.3f s, writer queue %.3f s\n", (unsigned long long)nBatches, nConsumers, tWait, tUpload, tScan, tDownload, tQueue);
    {
        std::lock_guard<std::mutex> lk(fifoMu);
        buildDone = true;
    }
    fifoCv.notify_all();
    for (auto &t : builders) t.join();
    {
        std::lock_guard<std::mutex> lk(fifoMu);
        fifoDone = true;
    }
    fifoCv.notify_all();
    writer.join();
    watchStop = true;
    if (watchdog.joinable()) watchdog.join();
    if (close(outFd) != 0) die("closing " + outFile + " failed");
    g_trace.mark("main pass done (parse + scan + read_data_init.txt)");

    // read_stats.txt (ReadSelection.hpp:305-384)
    const uint64_t nbReads = allReadSizes.size();
    const uint32_t n50 = compute_n50(allReadSizes);
    const uint32_t meanLen = nbReads ? compute_mean_length(allReadSizes) : 0;
    {
        float density = (float)((long double)nbSelected / (long double)nbKmers);
        float avgQ = (float)(qualitySum / qualityN);
        std::ofstream st(tmpDir + "/read_stats.txt", std::ios::binary);
        st.write((const char *)&nbReads, 8); st.write((const char *)&n50, 4); st.write((const char *)&density, 4);
        st.write((const char *)&nbBases, 8); st.write((const char *)&avgQ, 4); st.write((const char *)&meanLen, 4);
        st.write((const char *)&nbSelected, 8);
        st.close();
        if (!st) die("writing " + tmpDir + "/read_stats.txt failed");
    }

    g_log.line("\tNb reads: " + std::to_string(nbReads) + "  bases: " + std::to_string(nbBases) + "  minimizers: " + std::to_string(nbSelected) +
               "  N50: " + std::to_string(n50));
    // purgePalindromes (ReadSelection.hpp:1374-1431): lastK from the N50, --max-k ignored
    if (needCorrected) {
        const int lastK = compute_last_k(P.densityAssembly, n50, P.firstK, 0);
        std::sort(kept.begin(), kept.end(), [](const Kept &x, const Kept &y) { return x.seq < y.seq; });
        // A second pass shaped like the first, on GROUPS of batches: a batch is a few thousand reads and 16 k minimizers, and purging them
        // one by one -- two kernels, three waits and two copies each -- was 0.39 ms a batch, 0.18 to 0.23 s of a 50 Gbp FASTA / 20 Gbp FASTQ
        // run.  Every consumer appends the batches its context holds of a group of 32 consecutive ones on the device
        // (mdbg_minimizers_concat), purges them with one call, copies values and offsets back into one page-locked slab, and the writer
        // builds the `u32 n; u8 circular = 0; u32 m[n]` records batch by batch in read order from the slabs of both.
        struct Piece { HostBatch *hb; uint64_t firstRead; uint32_t nReads; std::atomic<int> *left; uint64_t bytes; };    // batch i = reads [firstRead, +nReads) of hb
        // Where a batch's records go in the file is known once every earlier batch has been purged (their sizes add up); from then on
        // any thread may build them and write them in place.  (One thread writing the 0.8 GB of a 50 Gbp read set through a stream was
        // what the pass took: 0.19 s, whatever the purging cost.)
        const int corrFd = open((tmpDir + "/read_data_corrected.txt").c_str(), O_CREAT | O_TRUNC | O_WRONLY, 0644);
        if (corrFd < 0) die("cannot write " + tmpDir + "/read_data_corrected.txt");
        std::map<uint64_t, Piece> placed2;              // purged, not yet given its place: position in read order -> piece
        std::deque<std::pair<Piece, uint64_t>> ready2;  // placed: piece, file offset
        uint64_t prefSeq2 = 0, prefOff2 = 0;            // batches [0, prefSeq2) have their place; the file is prefOff2 bytes so far
        size_t outstanding2 = 0;                        // pieces purged and not yet written
        bool done2 = false;
        auto write2 = [&] {
            std::string rec;
            for (;;) {
                Piece pc{};
                uint64_t at = 0;
                {
                    std::unique_lock<std::mutex> lk(fifoMu);
                    fifoCv.wait(lk, [&] { return !ready2.empty() || (done2 && outstanding2 == 0); });
                    if (ready2.empty()) return;
                    pc = ready2.front().first; at = ready2.front().second;
                    ready2.pop_front();
                }
                const HostBatch *hb = pc.hb;
                const uint64_t r0 = pc.firstRead, r1 = r0 + pc.nReads;
                rec.resize((size_t)pc.bytes);
                char *dst = &rec[0];
                for (uint64_t r = r0; r < r1; r++) {
                    const uint32_t k = (uint32_t)(hb->off[r + 1] - hb->off[r]);
                    memcpy(dst, &k, 4); dst[4] = 0;
                    memcpy(dst + 5, hb->m + hb->off[r], (size_t)k * 4);
                    dst += 5 + (size_t)k * 4;
                }
                for (size_t done = 0; done < rec.size();) {
                    const ssize_t w = pwrite(corrFd, rec.data() + done, rec.size() - done, (off_t)(at + done));
                    if (w < 0) { if (errno == EINTR) continue; die("write to " + tmpDir + "/read_data_corrected.txt failed"); }
                    done += (size_t)w;
                }
                const bool last = pc.left->fetch_sub(1) == 1;           // the slab's last batch
                if (last) delete pc.left;
                {
                    std::lock_guard<std::mutex> lk(fifoMu);
                    if (last) groupSlabs.push_back(pc.hb);
                    outstanding2--;
                }
                fifoCv.notify_all();
            }
        };
        groupSlabsStop = true;
        if (groupSlabHelper.joinable()) groupSlabHelper.join();
        if (getenv("MDBG_TRACE")) fprintf(stderr, "[mdbg_tool] %zu of %zu group slabs were allocated beside the main pass\n", groupSlabs.size(), groupSlabsWanted);
        // (eight, not four: over 50 Gbp the last record was written 0.16 s after the last group had been purged)
        std::vector<std::thread> writers2;
        for (int i = 0, nw = std::max(1, std::min(8, a.threads / 4)); i < nw; i++) writers2.emplace_back(write2);
        double pConcat = 0, pPurge = 0, pFree = 0, pSlab = 0, pCopy = 0, pPlace = 0;      // MDBG_TRACE: where the purging threads' time goes
        auto purge_own = [&](int ci) {
            mdbg_ctx *ctx = ctxs[(size_t)ci];
            std::vector<const mdbg_minimizers *> parts;
            std::vector<size_t> idx;
            std::vector<uint32_t> nReadsOf;
            double tc = 0, tp = 0, tf = 0, ts = 0, ty = 0, tl = 0;
            struct Sum { double &a, &b, &c, &d, &e, &f, &A, &B, &C, &D, &E, &F; std::mutex &mu;
                         ~Sum() { std::lock_guard<std::mutex> g(mu); A += a; B += b; C += c; D += d; E += e; F += f; } }
                sum{tc, tp, tf, ts, ty, tl, pConcat, pPurge, pFree, pSlab, pCopy, pPlace, fifoMu};
            for (size_t g0 = 0; g0 < kept.size(); g0 += GROUP) {
                parts.clear(); idx.clear(); nReadsOf.clear();
                for (size_t i = g0; i < std::min(kept.size(), g0 + GROUP); i++) {
                    if (kept[i].ctx != ctx) continue;
                    uint32_t bn = 0;
                    mdbg_minimizers_info(kept[i].mins, &bn, nullptr);
                    parts.push_back(kept[i].mins); idx.push_back(i); nReadsOf.push_back(bn);
                }
                if (parts.empty()) continue;
                const double t0 = g_trace.now();
                mdbg_minimizers *all = nullptr, *pur = nullptr;
                if (parts.size() > 1) check_on(ctx, mdbg_minimizers_concat(ctx, parts.data(), (uint32_t)parts.size(), &all), "mdbg_minimizers_concat");
                const double t1 = g_trace.now();
                check_on(ctx, mdbg_purge_palindromes(ctx, all ? all : parts[0], (uint32_t)P.firstK, (uint32_t)lastK, &pur), "mdbg_purge_palindromes");
                const double t2 = g_trace.now();
                if (all) mdbg_minimizers_free(all);
                for (size_t i : idx) mdbg_minimizers_free(kept[i].mins);
                const double t3 = g_trace.now();
                HostBatch *hb = nullptr;
                {
                    std::lock_guard<std::mutex> lk(fifoMu);
                    if (!groupSlabs.empty()) { hb = groupSlabs.back(); groupSlabs.pop_back(); }
                    else if (!spareBatches.empty()) { hb = spareBatches.back(); spareBatches.pop_back(); }
                }
                if (!hb) hb = new HostBatch();
                uint32_t bn; uint64_t bt;
                mdbg_minimizers_info(pur, &bn, &bt);
                hb->shape_values(bn, bt);
                const double t4 = g_trace.now();
                check_on(ctx, mdbg_minimizers_to_host(ctx, pur, hb->off, hb->m, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr), "to_host");
                if (a.thenGraph) { std::lock_guard<std::mutex> lk(fifoMu); purgedResident.emplace_back(idx[0], pur); }
                else mdbg_minimizers_free(pur);
                const double t5 = g_trace.now();
                std::atomic<int> *left = new std::atomic<int>((int)idx.size());
                {
                    std::unique_lock<std::mutex> lk(fifoMu);
                    // bounded -- but the group that holds the next batch to be placed always gets in
                    fifoCv.wait(lk, [&] { return outstanding2 < 6 * GROUP || prefSeq2 / GROUP == g0 / GROUP; });
                    uint64_t first = 0;
                    for (size_t j = 0; j < idx.size(); j++) {
                        const uint64_t bytes = (hb->off[first + nReadsOf[j]] - hb->off[first]) * 4 + (uint64_t)nReadsOf[j] * 5;
                        placed2.emplace((uint64_t)idx[j], Piece{hb, first, nReadsOf[j], left, bytes});
                        first += nReadsOf[j];
                    }
                    outstanding2 += idx.size();
                    for (auto it = placed2.find(prefSeq2); it != placed2.end(); it = placed2.find(prefSeq2)) {
                        ready2.emplace_back(it->second, prefOff2);
                        prefOff2 += it->second.bytes;
                        placed2.erase(it);
                        prefSeq2++;
                    }
                }
                fifoCv.notify_all();
                tc += t1 - t0; tp += t2 - t1; tf += t3 - t2; ts += t4 - t3; ty += t5 - t4; tl += g_trace.now() - t5;
            }
        };
        std::vector<std::thread> purgers;
        for (int i = 1; i < nConsumers; i++) purgers.emplace_back(purge_own, i);
        purge_own(0);
        for (auto &t : purgers) t.join();
        {
            std::lock_guard<std::mutex> lk(fifoMu);
            done2 = true;
        }
        fifoCv.notify_all();
        g_trace.mark("purge pass: every batch purged");
        for (auto &t : writers2) t.join();
        if (getenv("MDBG_TRACE"))
            fprintf(stderr, "[mdbg_tool] purge pass, summed over the %d purging thread(s): appending the batches of a group %.3f s, purging %.3f s, freeing %.3f s, "
                            "page-locked slab %.3f s, copy back %.3f s, waiting for a place %.3f s\n", nConsumers, pConcat, pPurge, pFree, pSlab, pCopy, pPlace);
        if (prefSeq2 != kept.size()) die("internal error: a purged batch was never placed");
        if (close(corrFd) != 0) die("closing " + tmpDir + "/read_data_corrected.txt failed");
    }
    g_trace.mark("read_data_corrected.txt written");
    if (a.thenGraph) {
        // the corrected reads as they sit on the device, in the order of the file (what order they are counted in changes nothing: the
        // tables are multisets; it is kept all the same)
        std::sort(purgedResident.begin(), purgedResident.end());
        std::vector<const mdbg_minimizers *> parts;
        for (auto &pr : purgedResident) parts.push_back(pr.second);
        mdbg_minimizers *all = nullptr;
        if (parts.size() == 1) all = purgedResident[0].second;
        else if (parts.size() > 1) {
            check(mdbg_minimizers_concat(g_ctx, parts.data(), (uint32_t)parts.size(), &all), "mdbg_minimizers_concat");
            for (auto &pr : purgedResident) mdbg_minimizers_free(pr.second);
        }
        g_trace.mark("asmStep: the corrected reads appended on the device");
        a.firstPass = true;
        a.pos = {tmpDir};
        graph_main(a, tmpDir, all);        // (all == null: no read at all -- the file path handles that like the reference does)
    }
    write_perf(tmpDir);
    g_trace.mark("done");
    finish();
}

// ---- graph ---------------------------------------------------------------------------------------------------------
// a vector whose resize() leaves new elements uninitialised (they are overwritten at once: 0.8 GB of zero-filling is 0.1 s)
template <typename T>
struct DefaultInit : std::allocator<T> {
    template <typename U> struct rebind { using other = DefaultInit<U>; };
    using std::allocator<T>::allocator;
    template <typename U> void construct(U *p) noexcept(std::is_nothrow_default_constructible<U>::value) { ::new ((void *)p) U; }
    template <typename U, typename... A> void construct(U *p, A &&...args) { ::new ((void *)p) U(std::forward<A>(args)...); }
};
using U32Vec = std::vector<uint32_t, DefaultInit<uint32_t>>;

// "u32 n; u8 circ; u32 m[n]" records (read_data_corrected.txt, unitig_data.txt) -> CSR.  One walk over the record headers for the
// offsets, then the values are copied by a few threads (a 50 Gbp read set is 0.8 GB of these records: 0.45 s of a 0.75 s `graph` when
// it was one loop over a zero-filled copy of the file).
template <typename Vec>
void parse_minimizer_reads(const uint8_t *raw, size_t size, Vec &mins, std::vector<uint64_t> &offs,
                           std::vector<uint8_t> *circular = nullptr, int threads = 1) {
    offs.assign(1, 0);
    offs.reserve(size / 64 + 16);
    std::vector<size_t> at;                       // byte position of every record's values
    at.reserve(size / 64 + 16);
    size_t o = 0;
    uint64_t total = 0;
    while (o + 5 <= size) {
        uint32_t n;
        memcpy(&n, raw + o, 4);
        if (circular) circular->push_back(raw[o + 4]);
        o += 5;
        if (o + (size_t)n * 4 > size) die("truncated minimizer read file");
        at.push_back(o);
        o += (size_t)n * 4;
        total += n;
        offs.push_back(total);
    }
    const size_t base = mins.size();
    mins.resize(base + total);
    const size_t nRec = at.size();
    const unsigned nThr = (unsigned)std::max(1, std::min(threads, 16));
    auto copy = [&](unsigned t) {
        for (size_t r = nRec * t / nThr, r1 = nRec * (t + 1) / nThr; r < r1; r++) {
            const size_t k = (size_t)(offs[r + 1] - offs[r]);
            if (k) memcpy(mins.data() + base + offs[r], raw + at[r], k * 4);
        }
    };
    if (nThr == 1 || total < (1u << 20)) { for (unsigned t = 0; t < nThr; t++) copy(t); }
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nThr; t++) pool.emplace_back(copy, t);
        for (auto &th : pool) th.join();
    }
}
void parse_minimizer_reads(const std::vector<uint8_t> &raw, std::vector<uint32_t> &mins, std::vector<uint64_t> &offs,
                           std::vector<uint8_t> *circular = nullptr) {
    parse_minimizer_reads(raw.data(), raw.size(), mins, offs, circular, 1);
}

// a whole file mapped read-only (no copy, no zero-fill)
struct MappedFile {
    const uint8_t *p = nullptr;
    size_t n = 0;
    explicit MappedFile(const std::string &path, bool populate = true, bool required = true) {
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) { if (required) die("File not found: " + path); return; }
        struct stat st;
        if (fstat(fd, &st) != 0) die("cannot stat " + path);
        n = (size_t)st.st_size;
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE | (populate ? MAP_POPULATE : 0), fd, 0);   // (not populated: the threads that walk it fault it in, each its own part)
            if (m == MAP_FAILED) die("cannot map " + path);
            p = (const uint8_t *)m;
        }
        close(fd);
    }
    ~MappedFile() { if (p) munmap((void *)p, n); }
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
};

// A file's bytes to the device as they are (mdbg_bytes_*): `workers` threads copy pieces of the mapped file into page-locked slabs of their
// own (two each: one travels while the other is filled) and queue them on the library's upload stream.  The 1.55 GB of configs[2]'s
// read_data_corrected.txt go over in the time the link needs; through one pageable hipMemcpy of a parsed copy they took three times that,
// after 0.5 s of parsing (profiles/round6_*_graph_per_k_*.json).
// page-locked slabs of 8 MB, kept from one upload to the next (locking 128 MB of pages costs as much as sending them)
struct SlabPool {
    static size_t bytes() { static const size_t b = [] { const char *e = getenv("MDBG_TOOL_SLAB_MB"); const int v = e ? atoi(e) : 0; return (size_t)(v >= 1 && v <= 256 ? v : 8) << 20; }(); return b; }
    std::mutex mu;
    std::map<mdbg_ctx *, std::vector<void *>> idle;      // by the context that locked the pages (the ranks of --gpus G sit on different devices)
    void *get(mdbg_ctx *ctx) {
        { std::lock_guard<std::mutex> g(mu); auto &v = idle[ctx]; if (!v.empty()) { void *p = v.back(); v.pop_back(); return p; } }
        void *p = nullptr;
        return mdbg_host_alloc(ctx, bytes(), &p) == MDBG_OK ? p : nullptr;
    }
    void put(mdbg_ctx *ctx, void *p) { if (p) { std::lock_guard<std::mutex> g(mu); idle[ctx].push_back(p); } }
} g_slabs;

mdbg_bytes *upload_file_bytes(mdbg_ctx *ctx, const uint8_t *p, size_t n, int threads) {
    mdbg_bytes *b = nullptr;
    check_on(ctx, mdbg_bytes_create(ctx, n, &b), "mdbg_bytes_create");
    if (!n) return b;
    const size_t piece = SlabPool::bytes();
    const size_t nPieces = (n + piece - 1) / piece;
    static const size_t maxW = [] { const char *e = getenv("MDBG_TOOL_UPLOAD_WORKERS"); const int v = e ? atoi(e) : 0; return (size_t)(v >= 1 && v <= 64 ? v : 8); }();
    const unsigned W = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)std::max(1, threads), maxW, nPieces}));
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    auto work = [&]() {
        void *slab[2] = {nullptr, nullptr};
        uint64_t ticket[2] = {0, 0};
        for (int s = 0; s < 2; s++) if (!(slab[s] = g_slabs.get(ctx))) { failed = 1; return; }
        for (int turn = 0;; turn ^= 1) {
            const size_t i = next.fetch_add(1);
            if (i >= nPieces || failed) break;
            if (ticket[turn] && mdbg_bytes_upload_done(ctx, b, ticket[turn], 1) != 1) { failed = 1; break; }
            const size_t at = i * piece, cnt = std::min(piece, n - at);
            memcpy(slab[turn], p + at, cnt);
            if (mdbg_bytes_upload_async(ctx, b, at, slab[turn], cnt, &ticket[turn]) != MDBG_OK) { failed = 1; break; }
        }
        for (int s = 0; s < 2; s++) {
            if (ticket[s] && mdbg_bytes_upload_done(ctx, b, ticket[s], 1) != 1) failed = 1;
            g_slabs.put(ctx, slab[s]);
        }
    };
    std::vector<std::thread> pool;
    for (unsigned w = 0; w < W; w++) pool.emplace_back(work);
    for (auto &t : pool) t.join();
    if (failed) die(std::string("upload of a record file failed: ") + mdbg_last_error(ctx));
    return b;
}

// What `graph` reads besides the reads when k > firstK, parsed once on the host (every rank builds its own device copy).
struct PrevInputs {
    std::unique_ptr<MappedFile> prevRec;          // kminmerData_abundance_prev.txt, mapped: its bytes travel as they are (mdbg_prev_from_record_bytes)
    std::vector<uint32_t> um, uab;                // unitigGraph_prev.nodes.bin sequences + the refined abundance of each (0xFFFFFFFF = none)
    std::vector<uint64_t> uoff{0};
    std::vector<uint32_t> unitigMins;             // unitig_data.txt
    std::vector<uint64_t> unitigOffs;
    std::vector<uint8_t> unitigCirc;
    bool hasUnitigs = false;
};

void load_prev_inputs(const std::string &dir, PrevInputs &in) {
    // loadRefinedAbundances (graph/CreateMdbg.cpp:3401-3709)
    in.prevRec.reset(new MappedFile(dir + "/kminmerData_abundance_prev.txt"));
    if (in.prevRec->n % 20) die("kminmerData_abundance_prev.txt is not a whole number of 20-byte records");
    // unitigGraph.nodes.refined_abundances.bin: (u32 unitigName, u32 abundance)*
    std::vector<uint8_t> ab = read_file(dir + "/unitigGraph.nodes.refined_abundances.bin", false);
    std::vector<std::pair<uint32_t, uint32_t>> name2ab(ab.size() / 8);
    for (size_t i = 0; i < name2ab.size(); i++) { memcpy(&name2ab[i].first, ab.data() + 8 * i, 4); memcpy(&name2ab[i].second, ab.data() + 8 * i + 4, 4); }
    std::sort(name2ab.begin(), name2ab.end());
    // unitigGraph_prev.nodes.bin: (u32 size; u32 m[size]; u32 unitigIndex)*, name = index / 2
    std::vector<uint8_t> nodes = read_file(dir + "/unitigGraph_prev.nodes.bin", false);
    for (size_t o = 0; o + 4 <= nodes.size();) {
        uint32_t n; memcpy(&n, nodes.data() + o, 4); o += 4;
        if (o + (size_t)n * 4 + 4 > nodes.size()) die("truncated unitigGraph_prev.nodes.bin");
        const size_t base = in.um.size();
        in.um.resize(base + n);
        if (n) memcpy(in.um.data() + base, nodes.data() + o, (size_t)n * 4);
        o += (size_t)n * 4;
        uint32_t idx; memcpy(&idx, nodes.data() + o, 4); o += 4;
        in.uoff.push_back(in.um.size());
        auto it = std::lower_bound(name2ab.begin(), name2ab.end(), std::make_pair(idx / 2, 0u));
        // several entries for one name: the reference's map keeps the last one
        uint32_t v = 0; bool found = false;
        for (; it != name2ab.end() && it->first == idx / 2; ++it) { v = it->second; found = true; }
        in.uab.push_back(found ? v : 0xFFFFFFFFu);   // 0xFFFFFFFF = no refined abundance: skipped (CreateMdbg.cpp:3483)
    }
    std::ifstream probe(dir + "/unitig_data.txt", std::ios::binary);
    if (probe) {
        parse_minimizer_reads(read_file(dir + "/unitig_data.txt", true), in.unitigMins, in.unitigOffs, &in.unitigCirc);
        in.hasUnitigs = true;
    }
}

struct RankTable {
    std::vector<uint8_t> rec;
    std::vector<uint32_t> vec;
    uint64_t n = 0, nSolid = 0;
    int hasVec = 0;
    uint64_t sums[4] = {0, 0, 0, 0};  // mdbg_table_checksum of the share: [0] is the "Checksum kminmer abundance" the reference logs
    std::string smallContigs;        // records for smallContigs_k<k>.bin (the rank that holds unitig_data.txt)
};

// What a pass at k > firstK looks its reads up in, on the device: the previous table with the refined abundances laid over it, and -- on the
// rank / piece that holds them -- the unitigs of unitig_data.txt, whose short ones become smallContigs_k<k>.bin records.
struct PrevOnDevice {
    mdbg_table *prev = nullptr;
    mdbg_minimizers *unitigs = nullptr;
    void build(mdbg_ctx *ctx, const Parameters &P, const PrevInputs &in, bool withUnitigs, std::string &smallContigs, int threads = 4) {
        const uint32_t k = (uint32_t)P.kminmerSize;
        {
            mdbg_bytes *rb = upload_file_bytes(ctx, in.prevRec->p, in.prevRec->n, threads);
            check_on(ctx, mdbg_prev_from_record_bytes(ctx, rb, in.prevRec->n / 20, &prev), "mdbg_prev_from_record_bytes");
            mdbg_bytes_free(rb);
        }
        if (!in.uab.empty()) {
            mdbg_minimizers *un = nullptr;
            check_on(ctx, mdbg_minimizers_from_host(ctx, in.um.data(), in.uoff.data(), (uint32_t)in.uab.size(), &un), "mdbg_minimizers_from_host");
            check_on(ctx, mdbg_prev_overlay_unitigs(ctx, prev, un, in.uab.data(), (uint32_t)P.prevK), "mdbg_prev_overlay_unitigs");
            mdbg_minimizers_free(un);
        }
        if (in.hasUnitigs && withUnitigs)
            check_on(ctx, mdbg_minimizers_from_host(ctx, in.unitigMins.data(), in.unitigOffs.data(), (uint32_t)(in.unitigOffs.size() - 1), &unitigs),
                     "mdbg_minimizers_from_host");
        // smallContigs_k<k>.bin is filled by the unitig pass of IndexKminmerFunctor when k > 8 (graph/CreateMdbg.hpp:1330-1352);
        // those unitigs have no k-min-mer, so they add nothing to the table.
        if (unitigs && k > 8 && k != P.firstK + 1) {
            const uint32_t nUnitigs = (uint32_t)(in.unitigOffs.size() - 1);
            std::vector<uint8_t> isSmall(nUnitigs);
            check_on(ctx, mdbg_small_contigs(ctx, unitigs, k, (uint32_t)P.prevK, prev, isSmall.data()), "mdbg_small_contigs");
            for (uint32_t u = 0; u < nUnitigs; u++) {
                if (!isSmall[u]) continue;
                const uint32_t n = (uint32_t)(in.unitigOffs[u + 1] - in.unitigOffs[u]);
                const uint8_t circ = in.unitigCirc[u];   // the record's own flag byte (Commons.hpp:7413, :7485)
                smallContigs.append((const char *)&n, 4); smallContigs.append((const char *)&circ, 1);
                smallContigs.append((const char *)(in.unitigMins.data() + in.unitigOffs[u]), (size_t)n * 4);
            }
        }
    }
    void free() {
        if (unitigs) mdbg_minimizers_free(unitigs);
        if (prev) mdbg_table_free(prev);
        unitigs = nullptr; prev = nullptr;
    }
};

// Where the reads of read_data_corrected.txt are: the mapped file itself (the bytes path: any range of whole records is a file of its own --
// record r of the range starts at byte 5 r + 4 off[r] of the range -- so every rank, piece or self-check takes its reads straight from the
// file's bytes) or, with MDBG_TOOL_PARSE_ON_HOST, the array a host parse made.
struct ReadSource {
    const uint8_t *raw = nullptr;                 // the mapped file, or null
    const U32Vec *mins = nullptr;                 // the parsed values (raw == null)
    const std::vector<uint64_t> *offs = nullptr;  // minimizers before each read, n_reads + 1
    int threads = 4;
    uint64_t n_minimizers() const { return offs->back(); }
};

mdbg_minimizers *upload_read_range(mdbg_ctx *ctx, const ReadSource &src, size_t r0, size_t r1) {
    const std::vector<uint64_t> &offs = *src.offs;
    std::vector<uint64_t> rel(offs.begin() + (long)r0, offs.begin() + (long)r1 + 1);
    mdbg_minimizers *reads = nullptr;
    if (!src.raw) {
        check_on(ctx, mdbg_minimizers_from_host(ctx, src.mins->data(), rel.data(), (uint32_t)(r1 - r0), &reads), "mdbg_minimizers_from_host");
        return reads;
    }
    const uint64_t base = rel[0];
    for (uint64_t &o : rel) o -= base;
    const size_t b0 = 5 * r0 + 4 * (size_t)offs[r0], b1 = 5 * r1 + 4 * (size_t)offs[r1];
    mdbg_bytes *rb = upload_file_bytes(ctx, src.raw + b0, b1 - b0, src.threads);
    check_on(ctx, mdbg_minimizers_from_record_bytes(ctx, rb, rel.data(), (uint32_t)(r1 - r0), nullptr, &reads), "mdbg_minimizers_from_record_bytes");
    mdbg_bytes_free(rb);
    return reads;
}

// One rank's part of `graph`: reads [r0, r1) of read_data_corrected.txt on `ctx`; with a communicator the table is built
// across the ranks (include/mdbg_hip.h "the exchange inside the library") and `out` is this rank's share of it.
// unitig_data.txt -- sequences, not reads -- goes to rank 0 only.
void graph_rank(mdbg_ctx *ctx, mdbg_comm *comm, int rank, const Parameters &P, const Args &a, const ReadSource &src,
                size_t r0, size_t r1, const PrevInputs &in, RankTable &out, bool rowsToHost = true,
                const std::function<void(mdbg_ctx *, mdbg_table *)> &sink = nullptr, mdbg_minimizers *resident = nullptr) {
    const uint32_t k = (uint32_t)P.kminmerSize;
    mdbg_minimizers *reads = resident;          // asmStep: the reads are on the device already (freed here like an uploaded set)
    if (!reads) reads = upload_read_range(ctx, src, r0, r1);
    mdbg_table *table = nullptr;
    if (a.firstPass) {
        if (comm) check_on(ctx, mdbg_kminmer_count_first_sharded(ctx, comm, reads, k, a.minAbundance, &table), "mdbg_kminmer_count_first_sharded");
        else check_on(ctx, mdbg_kminmer_count_first(ctx, reads, k, a.minAbundance, &table), "mdbg_kminmer_count_first");
    } else {
        PrevOnDevice dev;
        dev.build(ctx, P, in, rank == 0, out.smallContigs, a.threads);
        if (rank == 0) g_trace.mark("graph: the previous table on the device");
        mdbg_table *prev = dev.prev;
        mdbg_minimizers *unitigs = dev.unitigs;
        mdbg_table *local = nullptr;
        if (k == P.firstK + 1) check_on(ctx, mdbg_kminmer_count_refined(ctx, reads, unitigs, k, prev, &local), "mdbg_kminmer_count_refined");
        else check_on(ctx, mdbg_kminmer_index(ctx, reads, unitigs, k, prev, &local), "mdbg_kminmer_index");
        if (comm) {
            // the ranks agree on who lists a key several of them found (include/mdbg_hip.h, "Sharded k > firstK")
            mdbg_shard *sh = nullptr;
            const uint64_t *dRows = nullptr, *dReplies = nullptr;
            std::vector<uint64_t> counts(64, 0);
            check_on(ctx, mdbg_shard_from_table(ctx, local, (uint32_t)a.gpus, &sh, &dRows, counts.data()), "mdbg_shard_from_table");
            check_on(ctx, mdbg_shard_exchange(ctx, comm, sh, dRows, counts.data(), &dReplies), "mdbg_shard_exchange");
            check_on(ctx, mdbg_shard_keep(ctx, sh, dReplies, &table), "mdbg_shard_keep");
            mdbg_shard_free(sh);
            mdbg_table_free(local);
        } else table = local;
        dev.free();
    }
    mdbg_table_info(table, nullptr, &out.n, &out.nSolid, &out.hasVec);
    check_on(ctx, mdbg_table_checksum(ctx, table, out.sums), "mdbg_table_checksum");
    if (rank == 0) g_trace.mark("graph: the pass done, the table on the device");
    if (sink) sink(ctx, table);            // the rows go straight to their files (one rank: run_graph)
    else if (rowsToHost) {
        out.rec.resize(out.n * 20);
        out.vec.resize(out.hasVec ? out.n * k : 0);
        check_on(ctx, mdbg_table_to_host(ctx, table, out.rec.data(), out.hasVec ? out.vec.data() : nullptr), "mdbg_table_to_host");
    }
    mdbg_table_free(table);
    mdbg_minimizers_free(reads);
}

// The rows of a table to their files without a host copy of the whole table: pieces of 2 M rows come down into one of two page-locked
// buffers while the piece before is written -- every file by a thread of its own (the 20-byte records go to kminmerData_abundance.txt and to
// its `_init` copy, graph/CreateMdbg.cpp:515-522; the vectors to kminmerData_min.txt).  The ONT preset's first pass leaves 75 M records per
// 20 Gbp: 2.7 GB of rows, 4.2 GB of files -- 1.2 s of a 1.65 s `graph` when they went through pageable vectors and one writing thread.
// (rowBase > 0: the table is the next share of a pass made in pieces -- its rows follow the rowBase rows already in the files)
void stream_table_to_files(mdbg_ctx *ctx, mdbg_table *table, uint32_t k, const std::vector<std::string> &recFiles, const std::string &vecFile,
                           uint64_t rowBase = 0, bool firstShare = true) {
    uint64_t n = 0;
    int hasVec = 0;
    mdbg_table_info(table, nullptr, &n, nullptr, &hasVec);
    const bool vec = hasVec && !vecFile.empty();
    std::vector<int> recFd, vecFd;
    // The files of the k before are overwritten IN PLACE and cut to their new length at the end (ftruncate below): truncating a 300 MB
    // file of the page cache first gives its pages back one by one only for the writes to ask for them again (0.03 - 0.05 s a file at
    // configs[2]'s size).  What the next stage finds is the same: these bytes, this length.
    for (const std::string &f : recFiles) {
        const int fd = open(f.c_str(), O_CREAT | O_WRONLY, 0644);
        if (fd < 0) die("cannot write " + f);
        recFd.push_back(fd);
    }
    if (vec) {
        const int fd = open(vecFile.c_str(), O_CREAT | O_WRONLY, 0644);
        if (fd < 0) die("cannot write " + vecFile);
        vecFd.push_back(fd);
    }
    const uint64_t piece = (uint64_t)1 << 21;
    struct Buf { uint8_t *rec = nullptr; uint32_t *vec = nullptr; };
    Buf buf[2];
    if (n) for (Buf &b : buf) {
        void *p = nullptr;
        check_on(ctx, mdbg_host_alloc(ctx, std::min(n, piece) * 20, &p), "mdbg_host_alloc"); b.rec = (uint8_t *)p;
        if (vec) { check_on(ctx, mdbg_host_alloc(ctx, std::min(n, piece) * k * 4, &p), "mdbg_host_alloc"); b.vec = (uint32_t *)p; }
    }
    auto put = [](int fd, const void *data, size_t bytes, uint64_t at) {
        for (size_t done = 0; done < bytes;) {
            const ssize_t w = pwrite(fd, (const char *)data + done, bytes - done, (off_t)(at + done));
            if (w < 0) { if (errno == EINTR) continue; die("writing a table file failed"); }
            done += (size_t)w;
        }
    };
    std::vector<std::thread> writers;           // of the piece before the one coming down
    for (uint64_t first = 0, i = 0; first < n; first += piece, i++) {
        const uint64_t cnt = std::min(piece, n - first);
        Buf &b = buf[i & 1];
        if (i >= 2) { /* buf[i & 1] was written out two pieces ago: its writers were joined below */ }
        check_on(ctx, mdbg_table_to_host_range(ctx, table, first, cnt, b.rec, vec ? b.vec : nullptr), "mdbg_table_to_host_range");
        for (auto &t : writers) t.join();        // the other buffer is free again, this one is full
        writers.clear();
        // a piece of a file by four threads (round 6: one thread a file wrote the 322 MB of configs[2]'s table at k >= 6 in 0.1 s -- page
        // allocation in the page cache, not the copy -- half of what was left of a `graph` process)
        auto spread = [&](int fd, const void *data, size_t bytes, uint64_t at) {
            const size_t parts = bytes > ((size_t)8 << 20) ? 4 : 1;
            for (size_t q = 0; q < parts; q++) {
                const size_t a = bytes * q / parts / 4096 * 4096, e = q + 1 == parts ? bytes : bytes * (q + 1) / parts / 4096 * 4096;
                writers.emplace_back(put, fd, (const void *)((const char *)data + a), e - a, at + a);
            }
        };
        for (int fd : recFd) spread(fd, (const void *)b.rec, (size_t)(cnt * 20), (rowBase + first) * 20);
        for (int fd : vecFd) spread(fd, (const void *)b.vec, (size_t)(cnt * k * 4), (rowBase + first) * k * 4);
    }
    for (auto &t : writers) t.join();
    for (int fd : recFd) if (ftruncate(fd, (off_t)((rowBase + n) * 20)) != 0 || close(fd) != 0) die("closing a table file failed");
    for (int fd : vecFd) if (ftruncate(fd, (off_t)((rowBase + n) * k * 4)) != 0 || close(fd) != 0) die("closing a table file failed");
    for (Buf &b : buf) { if (b.rec) mdbg_host_free(ctx, b.rec); if (b.vec) mdbg_host_free(ctx, b.vec); }
}

// `graph` in pieces on one device: a read set with more minimizers than one call of the library takes (2^32: the flat index of an instance's first
// minimizer is 32 bits in the records of the first pass) is cut into contiguous read ranges, every range is put through the pass as ONE RANK OF A
// SHARDED JOB would be -- mdbg_shard_begin (or the range's own table -> mdbg_shard_from_table at k > firstK), the exchange among the pieces on the
// device (mdbg_shard_exchange_local), mdbg_shard_finish / _keep -- and the shares are written one after the other: their union IS the table of the
// whole set (include/mdbg_hip.h, "sharded first pass"; the reference meets the same limit-free way on disk, `vecHash % _nbPartitions`,
// graph/CreateMdbg.hpp:3714-3851).  The minimizers of all the pieces stay on the device until the last share is written (4 bytes each).
// cuts: read indexes, cuts[p] .. cuts[p + 1] = piece p.
void graph_pieces(mdbg_ctx *ctx, const Parameters &P, const Args &a, const ReadSource &src, const std::vector<size_t> &cuts,
                  const PrevInputs &in, RankTable &out, const std::function<void(mdbg_ctx *, mdbg_table *, uint64_t, bool)> &sink) {
    const uint32_t k = (uint32_t)P.kminmerSize;
    const uint32_t n = (uint32_t)(cuts.size() - 1);
    if (n > 64) die("graph: more than 64 pieces (" + std::to_string(n) + "): raise MDBG_TOOL_MAX_MINIMIZERS");
    // What this way of running the pass keeps on the device AT ONCE (round-4 ADVICE): every piece's minimizers (4 bytes each) and, until the
    // last share is written, every piece's shard state -- the instance records of the partitioned local pass (20 bytes per k-min-mer instance)
    // or the local table and its instance slots (up to 36) -- so about 40 bytes per minimizer of the whole read set.  A read set that needs
    // pieces at the default limit (3.5 * 10^9 minimizers a piece: a terabase of HiFi reads) does not fit one MI355X that way; it is a job for
    // --gpus G (a piece per device).  Said here, before the first upload, instead of as an allocation failure an hour in.
    {
        char arch[64]; int nCu = 0; uint64_t hbm = 0;
        check_on(ctx, mdbg_device_info(ctx, arch, sizeof arch, &nCu, &hbm), "mdbg_device_info");
        const uint64_t need = src.n_minimizers() * 40ull;
        if (hbm && need > hbm - hbm / 8)
            die("graph: " + std::to_string(src.n_minimizers()) + " minimizers in " + std::to_string(n) + " pieces need about " + std::to_string(need >> 30) + " GiB on the device at once, it has " +
                std::to_string(hbm >> 30) + " GiB: run the pass over several devices (--gpus G)");
    }
    std::vector<mdbg_minimizers *> reads(n, nullptr);
    std::vector<mdbg_shard *> shards(n, nullptr);
    std::vector<mdbg_table *> local(n, nullptr);
    std::vector<const uint64_t *> dRows(n, nullptr), dReplies(n, nullptr);
    std::vector<uint64_t> counts((size_t)n * n + 64, 0);
    PrevOnDevice dev;
    if (!a.firstPass) dev.build(ctx, P, in, true, out.smallContigs, a.threads);
    for (uint32_t p = 0; p < n; p++) {
        reads[p] = upload_read_range(ctx, src, cuts[p], cuts[p + 1]);
        std::vector<uint64_t> c(64, 0);
        if (a.firstPass) check_on(ctx, mdbg_shard_begin(ctx, reads[p], k, n, &shards[p], &dRows[p], c.data()), "mdbg_shard_begin");
        else {
            mdbg_minimizers *unitigs = p == 0 ? dev.unitigs : nullptr;      // sequences, not reads: with the first piece
            if (k == P.firstK + 1) check_on(ctx, mdbg_kminmer_count_refined(ctx, reads[p], unitigs, k, dev.prev, &local[p]), "mdbg_kminmer_count_refined");
            else check_on(ctx, mdbg_kminmer_index(ctx, reads[p], unitigs, k, dev.prev, &local[p]), "mdbg_kminmer_index");
            check_on(ctx, mdbg_shard_from_table(ctx, local[p], n, &shards[p], &dRows[p], c.data()), "mdbg_shard_from_table");
        }
        for (uint32_t d = 0; d < n; d++) counts[(size_t)p * n + d] = c[d];
    }
    dev.free();
    check_on(ctx, mdbg_shard_exchange_local(ctx, shards.data(), n, dRows.data(), counts.data(), dReplies.data()), "mdbg_shard_exchange_local");
    uint64_t rowBase = 0;
    for (uint32_t p = 0; p < n; p++) {
        mdbg_table *share = nullptr;
        if (a.firstPass) check_on(ctx, mdbg_shard_finish(ctx, shards[p], dReplies[p], a.minAbundance, &share), "mdbg_shard_finish");
        else check_on(ctx, mdbg_shard_keep(ctx, shards[p], dReplies[p], &share), "mdbg_shard_keep");
        uint64_t rows = 0, solid = 0, sums[4] = {0, 0, 0, 0};
        int hasVec = 0;
        mdbg_table_info(share, nullptr, &rows, &solid, &hasVec);
        check_on(ctx, mdbg_table_checksum(ctx, share, sums), "mdbg_table_checksum");
        out.n += rows; out.nSolid += solid; out.hasVec |= hasVec;
        for (int i = 0; i < 4; i++) out.sums[i] += sums[i];
        sink(ctx, share, rowBase, p == 0);
        rowBase += rows;
        mdbg_table_free(share);
        mdbg_shard_free(shards[p]);
        if (local[p]) mdbg_table_free(local[p]);
        mdbg_minimizers_free(reads[p]);
    }
}

int run_graph(int argc, char **argv) {
    Args a = parse_args(argc, argv, 2);
    if (a.pos.size() != 1) die("usage: mdbg_tool graph <tmpDir> --threads N [--min-abundance M] [--firstpass] [--gpus G [--no-verify]]");
    graph_main(a, a.pos[0], nullptr);
}

// `resident`: the reads of read_data_corrected.txt as asmStep left them on the device of g_ctx (null: the file is read)
[[noreturn]] void graph_main(Args a, const std::string &dir, mdbg_minimizers *resident) {
    Parameters P;
    P.load(dir + "/parameters.gz");
    g_log.open(dir);
    const uint32_t k = (uint32_t)P.kminmerSize;
    g_log.line("mdbg_tool graph (MI355X) k = " + std::to_string(k) + (a.firstPass ? " --firstpass" : "") +
               (a.gpus > 1 ? " --gpus " + std::to_string(a.gpus) : ""));
    // --gpus G: contiguous read ranges, one context (device r, one host thread) per rank, the exchange inside the library over
    // RCCL.  MDBG_TOOL_SHARDED=1 takes the same path with one rank (a communicator of one: what a one-GPU box can exercise).
    const int G = std::max(1, a.gpus);
    const bool sharded = G > 1 || getenv("MDBG_TOOL_SHARDED") != nullptr;
    uint64_t maxMins = 3500000000ull;    // more minimizers than one call of the library takes: the pass in pieces (MDBG_TOOL_MAX_MINIMIZERS: the most one piece may hold)
    if (const char *e = getenv("MDBG_TOOL_MAX_MINIMIZERS")) maxMins = std::max<uint64_t>(1, strtoull(e, nullptr, 10));
    if (resident) {                      // only the plain one-rank pass takes the reads where they are; anything else reads the file it has just written
        uint64_t total = 0;
        mdbg_minimizers_info(resident, nullptr, &total);
        if (sharded || total > maxMins) { mdbg_minimizers_free(resident); resident = nullptr; }
    }
    // one rank: the library context (HIP start-up: 0.09 - 0.17 s, a third of this command on a 10 Gbp read set) is created on a thread of its
    // own while this one reads the input files
    int ctxRc = MDBG_OK;
    std::thread ctxThread;
    if (!sharded && !g_ctx) ctxThread = std::thread([&] { ctxRc = mdbg_create(0, &g_ctx); });
    U32Vec mins;
    std::vector<uint64_t> offs{0};
    // One rank (the reference's own call): read_data_corrected.txt is not parsed into an array here.  Its record headers are walked by
    // several threads (host/records.hpp: exact) while the context comes up, its bytes then travel as they are and the records are taken
    // apart on the device (mdbg_minimizers_from_record_bytes) -- by every rank of a --gpus G job and every piece of a pass in pieces for its own
    // range of records.  MDBG_TOOL_PARSE_ON_HOST=1: the way of rounds 1 - 5 (one walk, a host copy of the values, one pageable upload).
    const bool byBytes = !resident && !getenv("MDBG_TOOL_PARSE_ON_HOST");
    std::unique_ptr<MappedFile> corrected;
    if (!resident) {
        offs.clear();
        corrected.reset(new MappedFile(dir + "/read_data_corrected.txt", !byBytes));
        if (byBytes) {
            mdbgfeed::RecordIndex idx = mdbgfeed::index_records(corrected->p, corrected->n, std::max(1, std::min(a.threads, 16)));
            if (idx.truncated) die("truncated minimizer read file");
            offs = std::move(idx.offs);
            g_trace.mark("graph: read_data_corrected.txt mapped, its records indexed");
        } else {
            parse_minimizer_reads(corrected->p, corrected->n, mins, offs, nullptr, std::max(1, a.threads));
            corrected.reset();
            g_trace.mark("graph: read_data_corrected.txt read and parsed");
        }
    }
    const size_t nReads = offs.size() - 1;
    if (nReads >= (1ull << 32)) die("graph: more than 2^32 records in read_data_corrected.txt");
    ReadSource src;
    src.raw = byBytes ? corrected->p : nullptr; src.mins = &mins; src.offs = &offs; src.threads = std::max(1, a.threads);
    PrevInputs in;
    if (!a.firstPass) load_prev_inputs(dir, in);
    // every `graph` run truncates smallContigs/smallContigs_k<k>.bin (graph/CreateMdbg.cpp:258-259)
    std::ofstream small(dir + "/smallContigs/smallContigs_k" + std::to_string(k) + ".bin", std::ios::binary);

    bool streamed = false;               // the table files were written as the rows came down (one rank)
    std::vector<RankTable> parts((size_t)G);
    if (!sharded) {
        if (ctxThread.joinable()) ctxThread.join();
        check(ctxRc, "mdbg_create");
        g_trace.mark("graph: context created");
        // one rank: the rows are streamed to their files (graph/CreateMdbg.cpp:451-464, :515-522 for the copies)
        std::vector<std::string> recFiles{dir + "/kminmerData_abundance.txt"};
        if (a.firstPass) recFiles.push_back(dir + "/kminmerData_abundance_init.txt");
        if (k == P.firstK + 1) recFiles.push_back(dir + "/kminmerData_abundance_init_k" + std::to_string(P.firstK + 1) + ".txt");
        std::vector<size_t> cuts{0};
        for (size_t r = 0, first = 0; r < nReads; r++) {
            if (offs[r + 1] - offs[r] > maxMins) die("graph: one read has more minimizers than a piece may hold");
            if (offs[r + 1] - offs[first] > maxMins) { cuts.push_back(r); first = r; }
        }
        cuts.push_back(nReads);
        if (byBytes && cuts.size() <= 2) {
            resident = upload_read_range(g_ctx, src, 0, nReads);
            g_trace.mark("graph: the records' bytes on the device, taken apart there");
        }
        if (cuts.size() > 2) {
            g_log.line("\tThe pass runs in " + std::to_string(cuts.size() - 1) + " pieces of at most " + std::to_string(maxMins) + " minimizers");
            graph_pieces(g_ctx, P, a, src, cuts, in, parts[0], [&](mdbg_ctx *c, mdbg_table *t, uint64_t rowBase, bool firstShare) {
                stream_table_to_files(c, t, k, recFiles, dir + "/kminmerData_min.txt", rowBase, firstShare);
            });
        } else
            graph_rank(g_ctx, nullptr, 0, P, a, src, 0, nReads, in, parts[0], false,
                       [&](mdbg_ctx *c, mdbg_table *t) { stream_table_to_files(c, t, k, recFiles, dir + "/kminmerData_min.txt"); }, resident);
        streamed = true;
        g_trace.mark("graph: table built and written");
    } else {
        uint8_t id[MDBG_COMM_ID_BYTES];
        check(mdbg_comm_unique_id(id), "mdbg_comm_unique_id");
        std::vector<mdbg_ctx *> ctxs((size_t)G, nullptr);
        std::vector<std::thread> th;
        for (int r = 0; r < G; r++) {
            th.emplace_back([&, r] {
                mdbg_ctx *ctx = nullptr;
                // (MDBG_TOOL_SHARE_GPU=1, tests: every rank on device 0 -- the G-rank job on a one-GPU box; the exchange then runs as peer
                // copies between the ranks' staging buffers on that device, RCCL refuses two ranks per device)
                const int dev = getenv("MDBG_TOOL_SHARE_GPU") ? 0 : r;
                if (mdbg_create(dev, &ctx) != MDBG_OK) die(std::string("mdbg_create(device ") + std::to_string(dev) + "): " + mdbg_last_error(nullptr));
                ctxs[(size_t)r] = ctx;
                mdbg_comm *comm = nullptr;
                // MDBG_COMM_MODE = peer | rccl | auto (default auto: peer copies between the devices of this process, RCCL if they fail their self-test)
                check_on(ctx, mdbg_comm_create(ctx, id, r, G, &comm), "mdbg_comm_create");
                if (r == 0 && getenv("MDBG_TRACE"))
                    fprintf(stderr, "[mdbg_tool] exchange among %d ranks: %s%s%s\n", G, mdbg_comm_mode(comm) == MDBG_COMM_PEER ? "peer copies" : "RCCL",
                            *mdbg_comm_note(comm) ? " (no peer copies: " : "", *mdbg_comm_note(comm) ? (std::string(mdbg_comm_note(comm)) + ")").c_str() : "");
                graph_rank(ctx, comm, r, P, a, src, nReads * (size_t)r / (size_t)G, nReads * (size_t)(r + 1) / (size_t)G, in, parts[(size_t)r]);
                mdbg_comm_destroy(comm);
            });
        }
        for (auto &t : th) t.join();
        g_ctx = ctxs[0];
        for (int r = 1; r < G; r++) mdbg_destroy(ctxs[(size_t)r]);
    }
    uint64_t n = 0, nSolid = 0, sums[4] = {0, 0, 0, 0};
    int hasVec = 0;                  // any rank's view: a rank without reads (G > reads) still reports the table's kind, but do not rely on it
    for (const RankTable &t : parts) {
        n += t.n; nSolid += t.nSolid; hasVec |= t.hasVec;
        for (int i = 0; i < 4; i++) sums[i] += t.sums[i];
        small.write(t.smallContigs.data(), (std::streamsize)t.smallContigs.size());
    }
    // the line the reference itself logs when it loads this table again at the next k (graph/CreateMdbg.cpp:3321, :3397)
    g_log.line("\tChecksum kminmer abundance: " + std::to_string(sums[0]));
    if (sharded && a.verify) {
        // The job checks itself: rank 0 repeats the pass alone over all the reads, and record count, solid count and the four
        // order-independent sums of the table (mdbg_table_checksum) must equal the sums over the ranks' shares -- the union of the
        // shares IS the single-GPU table.  A wrong exchange (a lost row, a count summed twice, two listers for one key) fails here,
        // in the run that produced it.  --no-verify skips it.
        RankTable whole;
        graph_rank(g_ctx, nullptr, 0, P, a, src, 0, nReads, in, whole, false);
        const bool same = whole.n == n && whole.nSolid == nSolid && whole.sums[0] == sums[0] && whole.sums[1] == sums[1] &&
                          whole.sums[2] == sums[2] && whole.sums[3] == sums[3];
        const std::string what = "records " + std::to_string(n) + " / " + std::to_string(whole.n) + ", solid " + std::to_string(nSolid) + " / " +
                                 std::to_string(whole.nSolid) + ", checksum " + std::to_string(sums[0]) + " / " + std::to_string(whole.sums[0]);
        g_log.line(std::string("\tSelf-check of the ") + std::to_string(G) + "-rank table against the single-GPU pass: " + (same ? "ok" : "FAILED") + " (" + what + ")");
        if (getenv("MDBG_TRACE")) fprintf(stderr, "[mdbg_tool] self-check %s: %s\n", same ? "ok" : "FAILED", what.c_str());
        if (!same) die("graph --gpus " + std::to_string(G) + ": the sharded table differs from the single-GPU pass (" + what + ")");
    }
    if (a.firstPass) {   // the two counts the reference logs after its first pass (graph/CreateMdbg.cpp:300-328)
        g_log.line("\tNb solid kminmers: " + std::to_string(nSolid));
        g_log.line("\tNb rescued kminmers: " + std::to_string(n - nSolid));
    } else {
        g_log.line("\tNb kminmers: " + std::to_string(n));
    }
    // the ranks' shares one after the other: the record order of these files is unspecified in the reference as well
    auto write_records = [&](const std::string &to) {
        std::ofstream f(dir + to, std::ios::binary);
        for (const RankTable &t : parts) f.write((const char *)t.rec.data(), (std::streamsize)t.rec.size());
        f.close();
        if (!f) die("writing " + dir + to + " failed");
    };
    if (!streamed) {
        write_records("/kminmerData_abundance.txt");
        if (hasVec) {
            std::ofstream f(dir + "/kminmerData_min.txt", std::ios::binary);
            for (const RankTable &t : parts) f.write((const char *)t.vec.data(), (std::streamsize)(t.vec.size() * 4));
            f.close();
            if (!f) die("writing " + dir + "/kminmerData_min.txt failed");
        }
        // graph/CreateMdbg.cpp:515-522
        if (a.firstPass) write_records("/kminmerData_abundance_init.txt");
        if (k == P.firstK + 1) write_records("/kminmerData_abundance_init_k" + std::to_string(P.firstK + 1) + ".txt");
    }
    small.close();
    if (!small) die("writing smallContigs_k" + std::to_string(k) + ".bin failed");
    g_trace.mark("graph: tables written");
    write_perf(dir);
    g_trace.mark("graph: done");
    finish();
}

}  // namespace

int main(int argc, char **argv) {
    if (argc < 2) die("usage: mdbg_tool <readSelection|graph> ...");
    setenv("GPU_MAX_HW_QUEUES", "8", 0);   // two contexts must not share a hardware queue; before HIP starts
    const std::string cmd = argv[1];
    if (!getenv("MDBG_TOOL_NO_DETACH") && (cmd == "readSelection" || cmd == "graph" || cmd == "asmStep")) {
        // launcher and worker (see signal_done); before anything of HIP exists in this process
        int fds[2];
        if (pipe(fds) == 0) {
            const pid_t pid = fork();
            if (pid > 0) {
                close(fds[1]);
                char c = 0;
                ssize_t r;
                do r = read(fds[0], &c, 1); while (r < 0 && errno == EINTR);
                if (r == 1) _exit((unsigned char)c);          // the worker's files are complete
                int st = 0;                                    // the worker ended without saying so: its status is the command's
                while (waitpid(pid, &st, 0) < 0 && errno == EINTR) {}
                _exit(WIFEXITED(st) ? WEXITSTATUS(st) : 1);
            }
            if (pid == 0) { close(fds[0]); g_done_fd = fds[1]; fcntl(g_done_fd, F_SETFD, FD_CLOEXEC); }
            else { close(fds[0]); close(fds[1]); }             // no fork: one process
        }
    }
    if (cmd == "readSelection") return run_read_selection(argc, argv);
    if (cmd == "graph") return run_graph(argc, argv);
    if (cmd == "asmStep") return run_read_selection(argc, argv, true);          // readSelection + graph --firstpass in one process, one context
    die("unknown sub-command " + cmd + " (only the hot-path tools exist here: readSelection, graph, asmStep)");
}
