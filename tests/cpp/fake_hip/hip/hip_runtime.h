// A FAKE hip_runtime.h for sanitizer builds of the host shim on the CPU (tests/test_host_sanitized_cpu.py): the types, constants and
// entry points fx_capi.cpp / fx_comm.cpp use, backed by malloc in fake_hip.cpp, every call countable and failable on demand.
// Nothing here is HIP and nothing here ships: the product is built against the real runtime by feature-extractor_amd/build.py.
#ifndef FX_FAKE_HIP_RUNTIME_H
#define FX_FAKE_HIP_RUNTIME_H

#include <cmath>
#include <cstddef>
#include <cstdint>

#define __host__
#define __device__
#define __global__

typedef enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorNotReady = 600, hipErrorLaunchFailure = 719, hipErrorUnknown = 999 } hipError_t;
typedef struct fake_stream* hipStream_t;
typedef struct fake_event* hipEvent_t;
typedef struct fake_graph* hipGraph_t;
typedef struct fake_graph_exec* hipGraphExec_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
typedef enum { hipStreamCaptureModeGlobal = 0, hipStreamCaptureModeThreadLocal = 1 } hipStreamCaptureMode;
typedef struct { char gcnArchName[256]; int multiProcessorCount; } hipDeviceProp_t;
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0, hipHostMallocCoherent = 0x40000000 };

extern "C" {
const char* hipGetErrorString(hipError_t);
hipError_t hipGetLastError(void);
hipError_t hipGetDeviceCount(int*);
hipError_t hipGetDeviceProperties(hipDeviceProp_t*, int);
hipError_t hipSetDevice(int);
hipError_t hipMalloc(void**, size_t);
hipError_t hipFree(void*);
hipError_t hipHostMalloc(void**, size_t, unsigned);
hipError_t hipHostFree(void*);
hipError_t hipHostGetDevicePointer(void**, void*, unsigned);
hipError_t hipMemcpy(void*, const void*, size_t, hipMemcpyKind);
hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t);
hipError_t hipMemsetAsync(void*, int, size_t, hipStream_t);
hipError_t hipStreamCreateWithFlags(hipStream_t*, unsigned);
hipError_t hipStreamDestroy(hipStream_t);
hipError_t hipStreamSynchronize(hipStream_t);
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned);
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode);
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t*);
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, void*, void*, size_t);
hipError_t hipGraphDestroy(hipGraph_t);
hipError_t hipGraphExecDestroy(hipGraphExec_t);
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t);
hipError_t hipEventCreate(hipEvent_t*);
hipError_t hipEventCreateWithFlags(hipEvent_t*, unsigned);
hipError_t hipEventDestroy(hipEvent_t);
hipError_t hipEventRecord(hipEvent_t, hipStream_t);
hipError_t hipEventSynchronize(hipEvent_t);
hipError_t hipEventQuery(hipEvent_t);
hipError_t hipEventElapsedTime(float*, hipEvent_t, hipEvent_t);

// ---- the test driver's handles on the fake ----
void fake_hip_reset(void);                 // counters to zero, no failure armed (live resources are kept: they are what the leak check reads)
void fake_hip_fail_at(long call);          // the call with this 1-based index (counted from the last reset) fails; 0 = none
long fake_hip_calls(void);                 // calls since the last reset
long fake_hip_live(void);                  // device + host allocations, streams, events, graphs alive right now
long fake_hip_live_bytes(void);
int  fake_hip_failed(void);                // whether the armed failure has fired
const char* fake_hip_failed_name(void);    // ... and in which entry point
hipError_t fake_hip_count(const char* name);   // for the kernel-launch stubs: counts as a call, fails if armed
// every hop the shim has handed to a frame / hop kernel since fake_hop_log_clear(), channel by channel in the order of analysis: the
// samples of channel c are bytes [fake_hop_log_bytes(c), + fake_hop_log_size(c)) -- what the analysers would have consumed
void fake_hop_log_clear(void);
void fake_hop_log_enable(int on);
const unsigned char* fake_hop_log_bytes(int channel);
size_t fake_hop_log_size(int channel);
}
#endif
