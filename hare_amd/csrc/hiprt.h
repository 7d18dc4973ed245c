// hiprt.h -- the HIP runtime, bound at run time.
//
// libhare_hip.so does not link libamdhip64: a host process may already have a HIP runtime mapped
// (PyTorch-ROCm bundles its own copy under another soname), and two runtimes in one process do
// not share streams or allocations.  On first use we bind to the runtime that is ALREADY loaded
// if there is one, else to $HARE_HIP_RUNTIME, else to the system's libamdhip64.so.  Kernels are
// shipped as a gfx950 code object embedded in the library and launched with the module API.
#pragma once
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <string>

namespace hare {

struct HipApi {
    hipError_t (*GetDeviceCount)(int*);
    hipError_t (*SetDevice)(int);
    hipError_t (*GetDevice)(int*);
    hipError_t (*DeviceGetAttribute)(int*, hipDeviceAttribute_t, int);
    hipError_t (*Malloc)(void**, size_t);
    hipError_t (*Free)(void*);
    hipError_t (*Memcpy)(void*, const void*, size_t, hipMemcpyKind);
    hipError_t (*MemcpyAsync)(void*, const void*, size_t, hipMemcpyKind, hipStream_t);
    hipError_t (*MemsetAsync)(void*, int, size_t, hipStream_t);
    hipError_t (*StreamCreate)(hipStream_t*);
    hipError_t (*StreamDestroy)(hipStream_t);
    hipError_t (*StreamSynchronize)(hipStream_t);
    hipError_t (*DeviceSynchronize)(void);
    hipError_t (*ModuleLoadData)(hipModule_t*, const void*);
    hipError_t (*ModuleUnload)(hipModule_t);
    hipError_t (*ModuleGetFunction)(hipFunction_t*, hipModule_t, const char*);
    hipError_t (*ModuleLaunchKernel)(hipFunction_t, unsigned, unsigned, unsigned, unsigned, unsigned, unsigned,
                                     unsigned, hipStream_t, void**, void**);
    const char* (*GetErrorString)(hipError_t);
    hipError_t (*GetLastError)(void);
    hipError_t (*EventCreate)(hipEvent_t*);
    hipError_t (*EventCreateWithFlags)(hipEvent_t*, unsigned int);
    hipError_t (*Memset)(void*, int, size_t);
    hipError_t (*EventDestroy)(hipEvent_t);
    hipError_t (*EventRecord)(hipEvent_t, hipStream_t);
    hipError_t (*EventSynchronize)(hipEvent_t);
    hipError_t (*StreamWaitEvent)(hipStream_t, hipEvent_t, unsigned int);
    hipError_t (*EventElapsedTime)(float*, hipEvent_t, hipEvent_t);
    hipError_t (*HostMalloc)(void**, size_t, unsigned int);
    hipError_t (*HostFree)(void*);
    hipError_t (*StreamIsCapturing)(hipStream_t, hipStreamCaptureStatus*);   // optional (null when the runtime lacks it)
    std::string path;   // which runtime was bound
};

// hipMalloc (0) / hipFree (1) / host-side waits -- hipDeviceSynchronize, hipStreamSynchronize, hipEventSynchronize -- (2) this library has
// made in this process so far (tests: a stream-ordered entry point makes none of them).
long long hip_call_count(int which);

// Binds on first call; returns nullptr and sets `err` if no runtime can be loaded.
const HipApi* hip_api(std::string* err);

}  // namespace hare
