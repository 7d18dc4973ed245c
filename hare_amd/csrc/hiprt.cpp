// hiprt.cpp -- run-time binding of the HIP runtime (see hiprt.h).
#include "hiprt.h"
#include <atomic>

#include <dlfcn.h>
#include <link.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>

namespace hare {
namespace {

int find_loaded_cb(struct dl_phdr_info* info, size_t, void* data)
{
    if (info->dlpi_name && strstr(info->dlpi_name, "libamdhip64.so")) {
        *static_cast<std::string*>(data) = info->dlpi_name;
        return 1;
    }
    return 0;
}

HipApi g_api;
bool g_ok = false;
std::string g_err;
std::once_flag g_once;

template <class F>
bool bind(void* h, const char* name, F& fn, std::string& err)
{
    void* p = dlsym(h, name);
    if (!p) {
        err = std::string("HIP runtime lacks symbol ") + name;
        return false;
    }
    fn = reinterpret_cast<F>(p);
    return true;
}

struct RealCalls {
    hipError_t (*Malloc)(void**, size_t) = nullptr;
    hipError_t (*Free)(void*) = nullptr;
    hipError_t (*DeviceSynchronize)(void) = nullptr;
    hipError_t (*StreamSynchronize)(hipStream_t) = nullptr;
    hipError_t (*EventSynchronize)(hipEvent_t) = nullptr;
} g_real;
std::atomic<long long> g_calls[3];      // 0 hipMalloc, 1 hipFree, 2 host-side waits (device / stream / event synchronise)
hipError_t counted_malloc(void** p, size_t n) { g_calls[0]++; return g_real.Malloc(p, n); }
hipError_t counted_free(void* p) { g_calls[1]++; return g_real.Free(p); }
hipError_t counted_device_sync(void) { g_calls[2]++; return g_real.DeviceSynchronize(); }
hipError_t counted_stream_sync(hipStream_t st) { g_calls[2]++; return g_real.StreamSynchronize(st); }
hipError_t counted_event_sync(hipEvent_t e) { g_calls[2]++; return g_real.EventSynchronize(e); }

void do_bind()
{
    std::string loaded;
    dl_iterate_phdr(find_loaded_cb, &loaded);
    void* h = nullptr;
    std::string tried;
    auto attempt = [&](const std::string& p) {
        if (h || p.empty()) return;
        h = dlopen(p.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (h) g_api.path = p;
        else tried += p + " ";
    };
    attempt(loaded);
    if (const char* e = getenv("HARE_HIP_RUNTIME")) attempt(e);
    attempt("libamdhip64.so.7");
    attempt("libamdhip64.so");
    if (const char* r = getenv("ROCM_PATH")) attempt(std::string(r) + "/lib/libamdhip64.so");
    attempt("/opt/rocm/lib/libamdhip64.so");
    if (!h) {
        g_err = "cannot load a HIP runtime (tried: " + tried + ")";
        return;
    }
    std::string e;
    bool ok = bind(h, "hipGetDeviceCount", g_api.GetDeviceCount, e) && bind(h, "hipSetDevice", g_api.SetDevice, e) &&
              bind(h, "hipGetDevice", g_api.GetDevice, e) &&
              bind(h, "hipDeviceGetAttribute", g_api.DeviceGetAttribute, e) && bind(h, "hipMalloc", g_api.Malloc, e) &&
              bind(h, "hipFree", g_api.Free, e) && bind(h, "hipMemcpy", g_api.Memcpy, e) &&
              bind(h, "hipMemcpyAsync", g_api.MemcpyAsync, e) && bind(h, "hipMemsetAsync", g_api.MemsetAsync, e) &&
              bind(h, "hipStreamCreate", g_api.StreamCreate, e) && bind(h, "hipStreamDestroy", g_api.StreamDestroy, e) &&
              bind(h, "hipStreamSynchronize", g_api.StreamSynchronize, e) &&
              bind(h, "hipDeviceSynchronize", g_api.DeviceSynchronize, e) &&
              bind(h, "hipModuleLoadData", g_api.ModuleLoadData, e) && bind(h, "hipModuleUnload", g_api.ModuleUnload, e) &&
              bind(h, "hipModuleGetFunction", g_api.ModuleGetFunction, e) &&
              bind(h, "hipModuleLaunchKernel", g_api.ModuleLaunchKernel, e) &&
              bind(h, "hipGetErrorString", g_api.GetErrorString, e) && bind(h, "hipGetLastError", g_api.GetLastError, e) && bind(h, "hipEventCreate", g_api.EventCreate, e) &&
              bind(h, "hipEventCreateWithFlags", g_api.EventCreateWithFlags, e) && bind(h, "hipMemset", g_api.Memset, e) &&
              bind(h, "hipEventDestroy", g_api.EventDestroy, e) && bind(h, "hipEventRecord", g_api.EventRecord, e) &&
              bind(h, "hipEventSynchronize", g_api.EventSynchronize, e) && bind(h, "hipStreamWaitEvent", g_api.StreamWaitEvent, e) &&
              bind(h, "hipEventElapsedTime", g_api.EventElapsedTime, e) && bind(h, "hipHostMalloc", g_api.HostMalloc, e) &&
              bind(h, "hipHostFree", g_api.HostFree, e);
    if (!ok) {
        g_err = e + " (" + g_api.path + ")";
        return;
    }
    {
        std::string ignored;                                   // optional entry points: absent from an old runtime = feature off
        if (!bind(h, "hipStreamIsCapturing", g_api.StreamIsCapturing, ignored)) g_api.StreamIsCapturing = nullptr;
    }
    // the calls a stream-ordered entry point must never make are counted (hip_call_count; hare_scene_get_option "hip_malloc_calls", ...):
    // tests hold hare_shoot_device to "no allocation, no free, no host-side wait"
    g_real.Malloc = g_api.Malloc; g_api.Malloc = counted_malloc;
    g_real.Free = g_api.Free; g_api.Free = counted_free;
    g_real.DeviceSynchronize = g_api.DeviceSynchronize; g_api.DeviceSynchronize = counted_device_sync;
    g_real.StreamSynchronize = g_api.StreamSynchronize; g_api.StreamSynchronize = counted_stream_sync;
    g_real.EventSynchronize = g_api.EventSynchronize; g_api.EventSynchronize = counted_event_sync;
    g_ok = true;
}

}  // namespace

long long hip_call_count(int which) { return which >= 0 && which < 3 ? g_calls[which].load() : -1; }

const HipApi* hip_api(std::string* err)
{
    std::call_once(g_once, do_bind);
    if (!g_ok) {
        if (err) *err = g_err;
        return nullptr;
    }
    return &g_api;
}

}  // namespace hare
