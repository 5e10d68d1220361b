// test_peerlink.cpp -- metamdbg_amd/csrc/peerlink.hpp (the control block of the peer-copy exchange) driven on a CPU: ranks as forked
// PROCESSES and as THREADS of one process, hundreds of exchanges of three phases, every rank reading every rank's words after
// every wait; a rank that fails locally (its code is seen by all at that very phase, the next exchange is in step again); a rank
// that never arrives (the others' wait ends at its deadline and names it); a rank that leaves; the name gone from /dev/shm.
// usage: test_peerlink <ranks> <exchanges>          prints "ok ..." and exits 0
#include <sys/wait.h>

#include <cstdlib>
#include <thread>
#include <vector>

#include "../../metamdbg_amd/csrc/peerlink.hpp"

using namespace mdbg;

static uint64_t word(uint64_t e, int r, int d) { return e * 1000003ull + (uint64_t)r * 131 + (uint64_t)d * 7 + 1; }

// what csrc/multigpu.hip's peer_phase does with the block: say rc, wait, read what everybody said
static int phase(PeerCtl &c, uint64_t e, int p, int rc, double timeout, int *who) {
    c.words(c.rank(), e)->status[p] = rc;
    c.arrive(PeerCtl::tick_of(e, p));
    const int late = c.wait_all(PeerCtl::tick_of(e, p), timeout);
    if (late >= 0) { *who = late; return -100; }
    for (int r = 0; r < c.n_ranks(); r++) {
        if (c.left_before(r, PeerCtl::tick_of(e, p))) { *who = r; return -200; }
        if (c.words(r, e)->status[p] != 0) { *who = r; return (int)c.words(r, e)->status[p]; }
    }
    return 0;
}

// one rank's life; returns 0 or a line number
static int rank_main(const std::string &name, int rank, int n, uint64_t exchanges, uint64_t fail_at, uint64_t stall_at) {
    PeerCtl c;
    const std::string err = c.attach(name, rank, n, 10.0);
    if (!err.empty()) { fprintf(stderr, "rank %d: %s\n", rank, err.c_str()); return __LINE__; }
    c.mine()->pid = (int32_t)getpid();
    c.mine()->device = rank;
    c.arrive(PeerCtl::TICK_ATTACHED);
    if (c.wait_all(PeerCtl::TICK_ATTACHED, 10.0) >= 0) return __LINE__;
    c.unlink_name();
    for (int r = 0; r < n; r++) if (c.slot(r)->device != r) return __LINE__;
    for (uint64_t e = 1; e <= exchanges; e++) {
        PeerWords *w = c.words(rank, e);
        for (int d = 0; d < n; d++) w->counts[d] = word(e, rank, d);
        w->rows.generation = e; w->rows.pointer = word(e, rank, 99);
        int who = -1;
        int rc = phase(c, e, 0, 0, 10.0, &who);
        if (rc) return __LINE__;
        for (int r = 0; r < n; r++) {
            for (int d = 0; d < n; d++) if (c.words(r, e)->counts[d] != word(e, r, d)) return __LINE__;
            if (c.words(r, e)->rows.generation != e || c.words(r, e)->rows.pointer != word(e, r, 99)) return __LINE__;
        }
        if (e == stall_at) {
            // the last rank never arrives at phase 1 of this exchange: the others name it after their deadline and stop; it leaves
            if (rank == n - 1) { std::this_thread::sleep_for(std::chrono::milliseconds(700)); c.arrive(PeerCtl::TICK_CLOSING); return 0; }
            const auto t0 = std::chrono::steady_clock::now();
            rc = phase(c, e, 1, 0, 0.3, &who);
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rc != -100 || who != n - 1 || waited < 0.25 || waited > 5.0) return __LINE__;
            // ... and a later wait passes over a rank that has left, which the reader must notice
            std::this_thread::sleep_for(std::chrono::milliseconds(900));
            rc = phase(c, e, 2, 0, 5.0, &who);           // (the others arrive, the one that left passes the wait without having arrived)
            return rc == -200 && who == n - 1 ? 0 : __LINE__;
        }
        // phase 1: on exchange fail_at rank 1 (or 0, alone) reports a local failure; everybody must see exactly that, here
        const int failing = n > 1 ? 1 : 0;
        const int mine = (e == fail_at && rank == failing) ? -3 : 0;
        w->replies.generation = e + 5;
        rc = phase(c, e, 1, mine, 10.0, &who);
        if (e == fail_at) {
            if (rank == failing ? rc != -3 : (rc != -3 || who != failing)) return __LINE__;
            continue;                                 // the exchange is abandoned by all; the next one starts in step
        }
        if (rc) return __LINE__;
        for (int r = 0; r < n; r++) if (c.words(r, e)->replies.generation != e + 5) return __LINE__;
        rc = phase(c, e, 2, 0, 10.0, &who);
        if (rc) return __LINE__;
        // the words of exchange e - 1 (other parity) are still what their rank wrote: nobody is two exchanges ahead
        if (e > 1 && e - 1 != fail_at)
            for (int r = 0; r < n; r++) if (c.words(r, e - 1)->counts[0] != word(e - 1, r, 0) && c.words(r, e - 1)->counts[0] != word(e + 1, r, 0)) return __LINE__;
    }
    c.arrive(PeerCtl::TICK_CLOSING);
    if (c.wait_all(PeerCtl::TICK_CLOSING, 10.0) >= 0) return __LINE__;
    return 0;
}

static std::string fresh_name(const char *tag) {
    uint8_t id[128];
    for (int i = 0; i < 128; i++) id[i] = (uint8_t)(i * 37 + getpid() + tag[0] * 3 + tag[1]);
    return PeerCtl::name_for(id);
}

static bool name_exists(const std::string &name) { return access(("/dev/shm" + name).c_str(), F_OK) == 0; }

static int run_processes(int n, uint64_t exchanges, uint64_t fail_at, uint64_t stall_at, const char *tag) {
    const std::string name = fresh_name(tag);
    std::vector<pid_t> kids;
    for (int r = 0; r < n; r++) {
        pid_t p = fork();
        if (p == 0) { int l = rank_main(name, r, n, exchanges, fail_at, stall_at); if (l) fprintf(stderr, "%s: rank %d failed at line %d\n", tag, r, l); _exit(l == 0 ? 0 : 1); }
        kids.push_back(p);
    }
    int bad = 0;
    for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); bad += !(WIFEXITED(st) && WEXITSTATUS(st) == 0); }
    if (name_exists(name)) { fprintf(stderr, "%s: the name is still under /dev/shm\n", tag); bad++; }
    return bad;
}

static int run_threads(int n, uint64_t exchanges, uint64_t fail_at, const char *tag) {
    const std::string name = fresh_name(tag);
    std::vector<int> rc((size_t)n, -1);
    std::vector<std::thread> th;
    for (int r = 0; r < n; r++) th.emplace_back([&, r] { rc[r] = rank_main(name, r, n, exchanges, fail_at, 0); });
    for (auto &t : th) t.join();
    int bad = 0;
    for (int r = 0; r < n; r++) if (rc[r]) { fprintf(stderr, "%s: rank %d failed at line %d\n", tag, r, rc[r]); bad++; }
    if (name_exists(name)) bad++;
    return bad;
}

// LATE is not DEAD (round-5 ADVICE): with a `late_s` behind the deadline, rank 1 that is alive and merely slow (1.2 s for a 0.3 s deadline) is
// waited for; rank 1 that has ENDED -- a zombie its parent has not reaped yet still answers kill(pid, 0) -- ends the wait at the deadline.
static int run_late_and_dead() {
    int bad = 0;
    for (int dead = 0; dead < 2; dead++) {
        const std::string name = fresh_name(dead ? "z1" : "l1");
        const pid_t kid = fork();
        if (kid == 0) {
            PeerCtl c;
            if (!c.attach(name, 1, 2, 10.0).empty()) _exit(2);
            c.mine()->pid = (int32_t)getpid();
            c.arrive(PeerCtl::TICK_ATTACHED);
            if (c.wait_all(PeerCtl::TICK_ATTACHED, 10.0) >= 0) _exit(3);
            if (dead) _exit(0);                                  // gone without a word (the parent below does not reap it before its own wait is over)
            std::this_thread::sleep_for(std::chrono::milliseconds(1200));
            c.arrive(PeerCtl::tick_of(1, 0));
            c.arrive(PeerCtl::TICK_CLOSING);
            _exit(0);
        }
        PeerCtl c;
        if (!c.attach(name, 0, 2, 10.0).empty()) return 1;
        c.mine()->pid = (int32_t)getpid();
        c.arrive(PeerCtl::TICK_ATTACHED);
        if (c.wait_all(PeerCtl::TICK_ATTACHED, 10.0) >= 0) return 1;
        c.unlink_name();
        c.arrive(PeerCtl::tick_of(1, 0));
        const auto t0 = std::chrono::steady_clock::now();
        const int late = c.wait_all(PeerCtl::tick_of(1, 0), 0.3, 20.0);
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (dead ? (late != 1 || waited < 0.25 || waited > 3.0) : (late != -1 || waited < 0.8)) { fprintf(stderr, "late/dead case %d: late %d after %.2f s\n", dead, late, waited); bad++; }
        int st = 0;
        waitpid(kid, &st, 0);
        c.arrive(PeerCtl::TICK_CLOSING);
    }
    return bad;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 4;
    const uint64_t ex = argc > 2 ? strtoull(argv[2], nullptr, 10) : 300;
    int bad = 0;
    bad += run_processes(n, ex, ex / 3 + 1, 0, "p1");          // processes, one local failure on the way
    bad += run_threads(n, ex, ex / 2 + 1, "t1");               // threads of one process
    bad += run_processes(1, 20, 5, 0, "s1");                   // a communicator of one
    bad += run_processes(n, 12, 0, 9, "d1");                   // a rank that never arrives, then leaves
    bad += run_late_and_dead();                                // a rank that is late but alive is waited for, a rank that has ended is not
    // two names from two ids differ; one id gives one name
    uint8_t a[128] = {0}, b[128] = {0};
    b[127] = 1;
    if (PeerCtl::name_for(a) == PeerCtl::name_for(b) || PeerCtl::name_for(a) != PeerCtl::name_for(a)) bad++;
    if (sizeof(PeerBufWords::handle) != 64) bad++;
    if (bad) { printf("FAILED: %d\n", bad); return 1; }
    printf("ok: %d ranks x %llu exchanges as processes and as threads, a local failure, a rank that never arrives, a rank that leaves\n", n, (unsigned long long)ex);
    return 0;
}
