// rccl.h -- TEST INFRASTRUCTURE: the declarations csrc/multigpu.hip needs to compile against tests/host/hip_on_host (the host stand-in
// for HIP).  The library resolves RCCL at run time with dlopen; on a CPU there is none, so MDBG_COMM_RCCL reports that and only the
// peer-copy transport runs (tests/test_exchange_on_cpu.py).
#pragma once
#include <cstddef>
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4, ncclInvalidUsage = 5 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclChar = 0, ncclUint8 = 1, ncclInt32 = 2, ncclInt = 2, ncclUint32 = 3, ncclInt64 = 4, ncclUint64 = 5 } ncclDataType_t;
extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId *);
ncclResult_t ncclCommInitRank(ncclComm_t *, int, ncclUniqueId, int);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclCommAbort(ncclComm_t);
ncclResult_t ncclCommCount(const ncclComm_t, int *);
ncclResult_t ncclCommUserRank(const ncclComm_t, int *);
const char *ncclGetErrorString(ncclResult_t);
ncclResult_t ncclGroupStart();
ncclResult_t ncclGroupEnd();
ncclResult_t ncclSend(const void *, size_t, ncclDataType_t, int, ncclComm_t, void *);
ncclResult_t ncclRecv(void *, size_t, ncclDataType_t, int, ncclComm_t, void *);
ncclResult_t ncclAllGather(const void *, void *, size_t, ncclDataType_t, ncclComm_t, void *);
}
