// peerlink.hpp -- host side of the peer-copy exchange (csrc/multigpu.hip): the control block the ranks of one communicator share.
//
// The two all-to-alls of a sharded pass can run as plain device-to-device copies between staging buffers (every owner PULLS its
// slice from every sender: over xGMI between GPUs that is the copy engines' work, nothing has to be resident on a compute unit
// beside another batch's scan).  What the ranks then need from each other is small and lives here, in host memory every rank
// maps: per rank and phase a status word (0 or its negative MDBG_E* code), its row of the count matrix, the handles of its
// staging buffers, and a monotonic `tick` that says how far it has come.  A rank ARRIVES at a phase by writing its words and then
// storing the tick (release); it WAITS for a phase by polling every rank's tick (acquire).  That is a barrier and an all-gather in
// one, it costs microseconds when the ranks are in step, and -- unlike a collective kernel or a host collective of another
// library -- it cannot hang: a wait has a deadline, and a rank that failed locally still arrives, with its code, so its peers
// return MDBG_EPEER at the same phase instead of waiting for rows that never come (include/mdbg_hip.h, "Failure behaviour of the
// collective calls").
//
// The block is a POSIX shared-memory object named after the communicator id (ranks are processes: one per GPU; or threads of one
// process: mdbg_tool graph --gpus G -- same code, every rank maps it).  Rank 0 creates it, everybody attaches, rank 0 unlinks the
// name once all are in: nothing is left under /dev/shm whatever happens later.
//
// Nearest reference analogue of the exchange it serves: KminmerCounter's partitioning by `vecHash % _nbPartitions`
// (graph/CreateMdbg.hpp:3714-3724); the reference has no process-to-process layer of its own.
//
// No HIP in this header: tests/host/test_peerlink.cpp drives it with plain processes and threads on a CPU.
#pragma once
#include <atomic>
#include <cerrno>
#include <signal.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

namespace mdbg {

constexpr int PEER_MAX_RANKS = 64;
constexpr int PEER_PHASES = 4;           // per exchange: counts known, rows staged + buffers ready, rows summed + replies staged, replies pulled
constexpr uint32_t PEER_MAGIC = 0x4d444247u;

// a staging buffer as its owner publishes it
struct PeerBufWords {
    uint64_t generation;                 // bumped whenever the owner replaced the buffer: peers re-map when it changed
    uint64_t pointer;                    // the device address in the owner's process (what a rank of the SAME process uses directly)
    uint64_t capacity;                   // bytes
    unsigned char handle[64];            // hipIpcMemHandle_t (what a rank of ANOTHER process opens)
};

// what a rank says at the phases of one exchange; two copies, used by even and odd exchanges in turn: words written for exchange
// E are overwritten for E + 2 only, which a rank reaches after every rank has ARRIVED in E + 1, i.e. is done reading E's
struct PeerWords {
    int64_t status[PEER_PHASES];
    uint64_t counts[PEER_MAX_RANKS];     // phase 0: rows this rank holds for every owner
    PeerBufWords rows, replies;          // phase 0 / phase 1: where its rows / its replies are staged
};

struct alignas(64) PeerSlot {
    std::atomic<uint64_t> tick;          // the last phase this rank has arrived at (monotonic; see PeerCtl::tick_of); TICK_CLOSING when it leaves
    std::atomic<uint64_t> reached;       // the last phase of an exchange it arrived at (what `tick` was before it left)
    int32_t pid, device;                 // written before the attach tick
    uint64_t process_token;              // a number drawn once per process: ranks with the same token are threads of ONE process (plain pointers).  (The pid
                                         // alone said so until round 5: two containers that share /dev/shm but not their pid namespaces can hold equal pids.)
    PeerWords words[2];
};

struct PeerHeader {
    std::atomic<uint32_t> magic;         // set by rank 0 when the block is initialised
    uint32_t n_ranks;
    uint64_t bytes;
};

class PeerCtl {
  public:
    PeerCtl() = default;
    PeerCtl(const PeerCtl &) = delete;
    PeerCtl &operator=(const PeerCtl &) = delete;
    ~PeerCtl() { detach(); }

    static size_t bytes_for(int n) { return slot_offset() + (size_t)n * sizeof(PeerSlot); }
    // exchange e >= 1, phase p in 0 .. PEER_PHASES-1; tick 1 = attached, the last ticks = closing
    static uint64_t tick_of(uint64_t exchange, int phase) { return exchange * 4 + (uint64_t)phase + 1; }
    static constexpr uint64_t TICK_ATTACHED = 1;
    static constexpr uint64_t TICK_CLOSING = ~0ull - 1;

    // Name of the shared object for a communicator id (128 bytes: an ncclUniqueId or any random bytes the ranks agree on).
    static std::string name_for(const uint8_t *id128) {
        uint64_t h[2] = {0xcbf29ce484222325ull, 0x84222325cbf29ce4ull};
        for (int i = 0; i < 128; i++) {
            h[i & 1] = (h[i & 1] ^ id128[i]) * 0x100000001b3ull;
            h[(i + 1) & 1] += h[i & 1] >> 29;
        }
        char buf[64];
        snprintf(buf, sizeof buf, "/mdbg_peer_%016llx%016llx", (unsigned long long)h[0], (unsigned long long)h[1]);
        return buf;
    }

    // Collective: rank 0 creates and initialises the block, the others wait for it to appear; returns "" or what went wrong.
    // Every rank then fills its identity, calls arrive(TICK_ATTACHED) and wait_all(TICK_ATTACHED) before anything else.
    std::string attach(const std::string &name, int rank, int n_ranks, double timeout_s) {
        detach();
        if (n_ranks < 1 || n_ranks > PEER_MAX_RANKS || rank < 0 || rank >= n_ranks) return "bad rank / rank count";
        name_ = name; rank_ = rank; n_ = n_ranks;
        const size_t bytes = bytes_for(n_ranks);
        const auto deadline = std::chrono::steady_clock::now() + std::chrono::duration<double>(timeout_s);
        int fd = -1;
        if (rank == 0) {
            (void)shm_unlink(name.c_str());                       // a leftover of a job that died with this very id: not ours
            fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
            if (fd < 0) return "shm_open(create " + name + "): " + strerror(errno);
            if (ftruncate(fd, (off_t)bytes) != 0) { std::string e = strerror(errno); close(fd); shm_unlink(name.c_str()); return "ftruncate: " + e; }
            created_ = true;
        } else {
            for (;;) {
                fd = shm_open(name.c_str(), O_RDWR, 0600);
                if (fd >= 0) {
                    struct stat st;
                    if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) break;
                    close(fd); fd = -1;                           // created but not sized yet
                }
                if (std::chrono::steady_clock::now() > deadline) return "the control block " + name + " did not appear (is /dev/shm shared by the ranks?)";
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
        }
        void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (p == MAP_FAILED) { std::string e = strerror(errno); if (created_) { shm_unlink(name.c_str()); created_ = false; } return "mmap: " + e; }
        base_ = (unsigned char *)p; bytes_ = bytes;
        if (rank == 0) {
            memset(base_, 0, bytes);
            header()->n_ranks = (uint32_t)n_ranks;
            header()->bytes = bytes;
            header()->magic.store(PEER_MAGIC, std::memory_order_release);
        } else {
            while (header()->magic.load(std::memory_order_acquire) != PEER_MAGIC) {
                if (std::chrono::steady_clock::now() > deadline) { detach(); return "the control block " + name + " was never initialised"; }
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
            if (header()->n_ranks != (uint32_t)n_ranks) { detach(); return "the control block was made for another number of ranks"; }
        }
        return "";
    }

    // rank 0, once wait_all(TICK_ATTACHED) has passed: the name is no longer needed
    void unlink_name() { if (created_) { shm_unlink(name_.c_str()); created_ = false; } }

    void detach() {
        if (created_) { shm_unlink(name_.c_str()); created_ = false; }
        if (base_) munmap(base_, bytes_);
        base_ = nullptr; bytes_ = 0;
    }

    bool attached() const { return base_ != nullptr; }
    int rank() const { return rank_; }
    int n_ranks() const { return n_; }
    PeerSlot *slot(int r) { return (PeerSlot *)(base_ + slot_offset()) + r; }
    PeerSlot *mine() { return slot(rank_); }
    PeerWords *words(int r, uint64_t exchange) { return &slot(r)->words[exchange & 1]; }

    // this rank has come as far as `tick` (its words for that phase are written)
    void arrive(uint64_t tick) {
        if (tick < TICK_CLOSING) mine()->reached.store(tick, std::memory_order_relaxed);
        mine()->tick.store(tick, std::memory_order_release);
    }
    // after a wait_all(tick) that passed: rank r is beyond `tick` only because it LEFT (closing passes every wait) without ever
    // arriving there -- its words for that phase are not to be read.  (A rank that arrived and then left is fine: what it wrote stays.)
    bool left_before(int r, uint64_t tick) {
        return slot(r)->tick.load(std::memory_order_acquire) >= TICK_CLOSING && slot(r)->reached.load(std::memory_order_relaxed) < tick;
    }

    // Until every rank has arrived at `tick` (or beyond).  Returns -1 when all have, else the first rank that had not when the
    // deadline passed.  Busy for a few microseconds, then yielding, then sleeping: ranks in step meet within the busy part.
    // a process that can be seen from here and has not ended (a zombie -- ended, not yet reaped by its parent -- still answers kill(pid, 0))
    static bool process_alive(int pid) {
        if (pid <= 0 || !(kill(pid, 0) == 0 || errno == EPERM)) return false;
        char path[64], buf[512];
        snprintf(path, sizeof path, "/proc/%d/stat", pid);
        FILE *f = fopen(path, "r");
        if (!f) return true;                                     // no /proc: kill's word stands
        const size_t n = fread(buf, 1, sizeof buf - 1, f);
        fclose(f);
        buf[n] = 0;
        const char *p = strrchr(buf, ')');                       // "pid (comm) S ..."
        return !(p && p[1] == ' ' && (p[2] == 'Z' || p[2] == 'X'));
    }

    // `late_s` > 0 tells LATE from DEAD (round-5 ADVICE): once the deadline has passed, a rank whose process can still be seen
    // (kill(pid, 0)) is given until `late_s` -- a rank that is merely slow (a terabase shard, a table growing, a debugger) should not end
    // the job where RCCL would simply wait; a rank whose process is gone, or cannot be seen from here (another pid namespace), ends the
    // wait at the deadline as before.  Liveness only ever EXTENDS a wait, it never ends one early.
    int wait_all(uint64_t tick, double timeout_s, double late_s = 0.0) {
        const auto t0 = std::chrono::steady_clock::now();
        unsigned spins = 0;
        for (int r = 0; r < n_; r++) {
            while (slot(r)->tick.load(std::memory_order_acquire) < tick) {
                if (++spins < 4096) {
#if defined(__x86_64__) || defined(__i386__)
                    __builtin_ia32_pause();
#endif
                    continue;
                }
                if ((spins & 63) == 0) {
                    const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                    if (waited > timeout_s) {
                        const int pid = slot(r)->pid;
                        const bool alive = late_s > timeout_s && waited <= late_s && process_alive(pid);
                        if (!alive) return r;
                    }
                }
                if (spins < 65536) sched_yield(); else std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
        return -1;
    }

  private:
    static size_t slot_offset() { return (sizeof(PeerHeader) + 63) / 64 * 64; }
    PeerHeader *header() { return (PeerHeader *)base_; }
    unsigned char *base_ = nullptr;
    size_t bytes_ = 0;
    std::string name_;
    int rank_ = 0, n_ = 1;
    bool created_ = false;
};

}  // namespace mdbg
