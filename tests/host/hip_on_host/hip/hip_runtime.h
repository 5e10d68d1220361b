// hip_runtime.h -- TEST INFRASTRUCTURE: a stand-in for the HIP runtime that keeps "device" memory in host memory, so that the SHIPPED
// sources of the exchange between GPUs (metamdbg_amd/csrc/multigpu.hip + peerlink.hpp, context.hip, common.hpp, objects.hpp) compile with
// g++ and run as several PROCESSES on a CPU (tests/test_exchange_on_cpu.py; round-5 VERDICT item 5: the CPU multi-rank test covered the
// Python harness, not the library's transport).  Never shipped, never seen by hipcc: only tests/ puts this directory on an include path.
//
//   * hipMalloc      a POSIX shared-memory object mapped into the caller (so that another process can map it too)
//   * hipIpc*        the handle is the object's name: hipIpcOpenMemHandle maps it -- the cross-process path of csrc/multigpu.hip's peer_map
//   * copies, sets   memcpy / memset at once; streams and events exist as tokens, everything is complete when the call returns
//   * kernels        hipLaunchKernelGGL runs the kernel function for every (block, thread) in turn -- enough for kernels without barriers,
//                    shared memory or cross-lane traffic, which is what the sources above launch
// What it cannot show: anything about a GPU.  What it does show: the count matrix, the four phases and their status words, the staging
// buffers' publication / mapping / growth, deadlines and MDBG_EPEER -- the host logic of the transport, in real processes.
#pragma once
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorNotReady = 600 };
inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : e == hipErrorOutOfMemory ? "out of memory" : e == hipErrorNotReady ? "not ready" : "invalid value"; }
inline hipError_t hipGetLastError() { return hipSuccess; }

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct FakeStream { int id; };
struct FakeEvent { double t; };
typedef FakeStream *hipStream_t;
typedef FakeEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipIpcMemLazyEnablePeerAccess = 1 };
enum hipDeviceAttribute_t { hipDeviceAttributeWallClockRate = 1 };
struct hipIpcMemHandle_t { char reserved[64]; };
struct hipDeviceProp_t { char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; size_t maxSharedMemoryPerMultiProcessor; int clockRate; };

namespace fakehip {
inline thread_local dim3 tl_block_idx, tl_thread_idx, tl_block_dim, tl_grid_dim;
struct Region { std::string name; size_t bytes; bool opened; };
inline std::mutex &mu() { static std::mutex m; return m; }
inline std::map<void *, Region> &regions() { static std::map<void *, Region> r; return r; }
inline std::atomic<unsigned> &counter() { static std::atomic<unsigned> c{0}; return c; }
inline void unlink_all() {                      // at exit: nothing of this process is left under /dev/shm
    std::lock_guard<std::mutex> g(mu());
    for (auto &kv : regions()) if (!kv.second.opened) shm_unlink(kv.second.name.c_str());
}
inline double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
}
#define threadIdx fakehip::tl_thread_idx
#define blockIdx fakehip::tl_block_idx
#define blockDim fakehip::tl_block_dim
#define gridDim fakehip::tl_grid_dim

inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int *n) { const char *e = getenv("FAKEHIP_DEVICES"); *n = e ? atoi(e) : 8; return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t *p, int) { memset(p, 0, sizeof *p); strcpy(p->gcnArchName, "gfx950:host-stand-in"); p->multiProcessorCount = 4; p->totalGlobalMem = (size_t)8 << 30; p->maxSharedMemoryPerMultiProcessor = 163840; p->clockRate = 2400000; return hipSuccess; }
inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 100000; return hipSuccess; }
inline hipError_t hipDeviceCanAccessPeer(int *can, int a, int b) { const char *e = getenv("FAKEHIP_NO_PEER_ACCESS"); *can = !(e && a != b); return hipSuccess; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }

inline hipError_t hipMalloc(void **out, size_t bytes) {
    static const bool reg = [] { (void)fakehip::mu(); (void)fakehip::regions(); atexit(fakehip::unlink_all); return true; }();   // (the map outlives the handler)
    (void)reg;
    if (bytes == 0) bytes = 1;
    char name[64];
    snprintf(name, sizeof name, "/fakehip_%d_%u", (int)getpid(), fakehip::counter().fetch_add(1));
    const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) return hipErrorOutOfMemory;
    if (ftruncate(fd, (off_t)bytes) != 0) { close(fd); shm_unlink(name); return hipErrorOutOfMemory; }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { shm_unlink(name); return hipErrorOutOfMemory; }
    std::lock_guard<std::mutex> g(fakehip::mu());
    fakehip::regions()[p] = {name, bytes, false};
    *out = p;
    return hipSuccess;
}
inline hipError_t hipFree(void *p) {
    if (!p) return hipSuccess;
    std::lock_guard<std::mutex> g(fakehip::mu());
    auto it = fakehip::regions().find(p);
    if (it == fakehip::regions().end()) return hipErrorInvalidValue;
    munmap(p, it->second.bytes);
    if (!it->second.opened) shm_unlink(it->second.name.c_str());
    fakehip::regions().erase(it);
    return hipSuccess;
}
inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *h, void *p) {
    std::lock_guard<std::mutex> g(fakehip::mu());
    auto it = fakehip::regions().find(p);
    if (it == fakehip::regions().end()) return hipErrorInvalidValue;
    memset(h, 0, sizeof *h);
    snprintf(h->reserved, sizeof h->reserved, "%s:%zu", it->second.name.c_str(), it->second.bytes);
    return hipSuccess;
}
inline hipError_t hipIpcOpenMemHandle(void **out, hipIpcMemHandle_t h, unsigned) {
    char name[64];
    size_t bytes = 0;
    const char *colon = strrchr(h.reserved, ':');
    if (!colon || (size_t)(colon - h.reserved) >= sizeof name) return hipErrorInvalidValue;
    memcpy(name, h.reserved, (size_t)(colon - h.reserved)); name[colon - h.reserved] = 0;
    bytes = (size_t)strtoull(colon + 1, nullptr, 10);
    const int fd = shm_open(name, O_RDWR, 0600);
    if (fd < 0) return hipErrorInvalidValue;
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> g(fakehip::mu());
    fakehip::regions()[p] = {name, bytes, true};
    *out = p;
    return hipSuccess;
}
inline hipError_t hipIpcCloseMemHandle(void *p) { return hipFree(p); }
inline hipError_t hipHostMalloc(void **out, size_t bytes, unsigned) { *out = malloc(bytes ? bytes : 1); return *out ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }

inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }

inline hipError_t hipStreamCreate(hipStream_t *s) { *s = new FakeStream{0}; return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { return hipStreamCreate(s); }
inline hipError_t hipExtStreamCreateWithCUMask(hipStream_t *s, uint32_t, const uint32_t *) { return hipStreamCreate(s); }
inline hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new FakeEvent{0}; return hipSuccess; }
inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = fakehip::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }

// every (block, thread) of a launch in turn, on the calling thread
template <typename K, typename... A>
inline void fakehip_launch(K kernel, dim3 grid, dim3 block, A... args) {
    fakehip::tl_grid_dim = grid; fakehip::tl_block_dim = block;
    for (unsigned b = 0; b < grid.x; b++)
        for (unsigned t = 0; t < block.x; t++) {
            fakehip::tl_block_idx = dim3(b, 0, 0); fakehip::tl_thread_idx = dim3(t, 0, 0);
            kernel(args...);
        }
}
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) fakehip_launch(kernel, grid, block, __VA_ARGS__)
inline long long wall_clock64() { return (long long)(fakehip::now() * 100.0); }
inline void __builtin_amdgcn_s_sleep(int) {}
