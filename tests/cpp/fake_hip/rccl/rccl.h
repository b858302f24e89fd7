// A FAKE rccl.h for the sanitizer builds: the few types and prototypes fx_comm.cpp names.  fake_rccl.cpp builds the matching
// librccl.so.1 (one process = one rank; send / receive to oneself) that fx_comm.cpp's dlopen finds first through LD_LIBRARY_PATH.
#ifndef FX_FAKE_RCCL_H
#define FX_FAKE_RCCL_H
#include <hip/hip_runtime.h>
#define NCCL_UNIQUE_ID_BYTES 128
typedef struct { char internal[NCCL_UNIQUE_ID_BYTES]; } ncclUniqueId;
typedef struct fake_nccl_comm* ncclComm_t;
typedef enum { ncclSuccess = 0, ncclUnhandledCudaError = 1, ncclSystemError = 2, ncclInternalError = 3, ncclInvalidArgument = 4 } ncclResult_t;
typedef enum { ncclInt8 = 0, ncclInt32 = 2, ncclFloat32 = 7 } ncclDataType_t;
extern "C" {
ncclResult_t ncclGetUniqueId(ncclUniqueId*);
ncclResult_t ncclCommInitRank(ncclComm_t*, int, ncclUniqueId, int);
ncclResult_t ncclCommDestroy(ncclComm_t);
ncclResult_t ncclCommCount(const ncclComm_t, int*);
ncclResult_t ncclAllGather(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t);
ncclResult_t ncclGroupStart(void);
ncclResult_t ncclGroupEnd(void);
ncclResult_t ncclSend(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
ncclResult_t ncclRecv(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t);
const char* ncclGetErrorString(ncclResult_t);
}
#endif
