// tests/host_shim/stub_hip.cpp -- a HIP runtime made of host memory, for driving the product's HOST layer on a CPU under ThreadSanitizer
// (tests/test_host_sanitizers.py; no GPU, no libamdhip64).  "Device" memory is zero-filled host memory tagged with the device that was
// current when it was allocated (eight devices), copies happen at once, streams and events are inert handles that are always complete, and every
// thread has its own current device -- which is what the product's DeviceScope and the placement of the lazily built tables are
// checked against.  Test infrastructure only: nothing here is linked into libmi355kzg.so.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace {
constexpr int N_DEVICES = 8;          // the multi-device scenarios list up to eight
thread_local int tl_device = 0;
std::mutex g_mu;
std::unordered_map<const void*, int> g_owner;          // allocation -> device
struct StubStream { int priority; };
}  // namespace

extern "C" int stub_hip_live_allocations() {
    std::lock_guard<std::mutex> g(g_mu);
    return (int)g_owner.size();
}

hipError_t hipDeviceGetPCIBusId(char*, int, int) { return hipErrorInvalidValue; }      // no sysfs entry to bind the pool to: the workers float
hipError_t hipGetDeviceCount(int* n) { *n = N_DEVICES; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= N_DEVICES) return hipErrorInvalidDevice; tl_device = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = tl_device; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorName(hipError_t) { return "stub"; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 1; *greatest = -1; return hipSuccess; }
hipError_t hipMemGetInfo(size_t* free_b, size_t* total_b) { *free_b = (size_t)250 << 30; *total_b = (size_t)288 << 30; return hipSuccess; }

hipError_t hipMalloc(void** p, size_t bytes) {
    *p = calloc(1, bytes ? bytes : 1);
    if (!*p) return hipErrorOutOfMemory;
    std::lock_guard<std::mutex> g(g_mu);
    g_owner[*p] = tl_device;
    return hipSuccess;
}
hipError_t hipFree(void* p) {
    if (!p) return hipSuccess;
    {
        std::lock_guard<std::mutex> g(g_mu);
        g_owner.erase(p);
    }
    free(p);
    return hipSuccess;
}
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned int) { *p = calloc(1, bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
    std::lock_guard<std::mutex> g(g_mu);
    auto it = g_owner.find(p);
    if (it == g_owner.end()) return hipErrorInvalidValue;
    memset(a, 0, sizeof(*a));
    a->type = hipMemoryTypeDevice;
    a->device = it->second;
    return hipSuccess;
}
// (launch graphs, ctx.h lane_graph_run: this runtime has none -- every sequence falls back to plain launches, which is a tested path of its own)
hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { return hipErrorNotSupported; }
hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t* g) { *g = nullptr; return hipErrorNotSupported; }
hipError_t hipGraphExecUpdate(hipGraphExec_t, hipGraph_t, hipGraphNode_t*, hipGraphExecUpdateResult*) { return hipErrorNotSupported; }
hipError_t hipGraphInstantiate(hipGraphExec_t*, hipGraph_t, hipGraphNode_t*, char*, size_t) { return hipErrorNotSupported; }
hipError_t hipGraphDestroy(hipGraph_t) { return hipSuccess; }
hipError_t hipGraphLaunch(hipGraphExec_t, hipStream_t) { return hipErrorNotSupported; }
hipError_t hipGraphExecDestroy(hipGraphExec_t) { return hipSuccess; }
hipError_t hipMemsetAsync(void* dst, int v, size_t n, hipStream_t) { if (n) memset(dst, v, n); return hipSuccess; }
hipError_t hipMemset(void* dst, int v, size_t n) { if (n) memset(dst, v, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) { if (n) memcpy(dst, src, n); return hipSuccess; }

hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned int, int priority) { *s = (hipStream_t) new StubStream{priority}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = (hipStream_t) new StubStream{0}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete (StubStream*)s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned int) { return hipSuccess; }
hipError_t hipStreamGetPriority(hipStream_t s, int* p) { *p = s ? ((StubStream*)s)->priority : 0; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t) new int(0); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t) new int(0); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete (int*)e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
