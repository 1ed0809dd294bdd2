/* A host-only stand-in for <hip/hip_runtime.h>: exactly the runtime surface dvi_ekf_amd/csrc/eskf_capi.cpp uses, backed by
 * malloc / memcpy (tests/hipstub/hip_stub.cpp).  TEST INFRASTRUCTURE: it lets the C-ABI layer -- handle lifetime, workspace
 * growth, argument validation, schedule walking, stream fork / join -- be built with -fsanitize=address,undefined and run on
 * the CPU (tests/test_capi_sanitized.py); GPU sanitizers are not available on the pool.  Never part of the product build. */
#pragma once
#include <stddef.h>
#include <stdint.h>

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2, hipErrorStreamCaptureImplicit = 906 };
typedef struct hipstub_stream* hipStream_t;
typedef struct hipstub_event* hipEvent_t;
typedef enum { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 } hipMemcpyKind;
typedef enum { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 } hipStreamCaptureStatus;
enum { hipEventDisableTiming = 2, hipStreamNonBlocking = 1 };

#ifdef __cplusplus
extern "C" {
#endif
hipError_t hipGetDeviceCount(int* n);
hipError_t hipGetDevice(int* d);
hipError_t hipSetDevice(int d);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipMallocAsync(void** p, size_t bytes, hipStream_t s);
hipError_t hipFreeAsync(void* p, hipStream_t s);
hipError_t hipMemsetAsync(void* p, int value, size_t bytes, hipStream_t s);
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind, hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize(void);
hipError_t hipStreamIsCapturing(hipStream_t s, hipStreamCaptureStatus* st);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
/* test controls */
void hipstub_set_capturing(hipStream_t s, int on);      /* what hipStreamIsCapturing reports for s */
void hipstub_fail_next_malloc(int n);                   /* the n-th hipMalloc from now fails (0 = none) */
long hipstub_live_allocations(void);
long hipstub_launches(void);
#ifdef __cplusplus
}
#endif
