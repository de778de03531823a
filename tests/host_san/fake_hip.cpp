// fake_hip.cpp -- TEST INFRASTRUCTURE: a host stand-in for the HIP runtime entry points the library's host code calls, so
// that runtime.cpp / capi.cpp / host_pipeline.cpp / linalg.cpp run under ThreadSanitizer and AddressSanitizer in a
// container without a GPU (tests/test_host_sanitizers.py).  "Device" memory is host memory, streams execute at once on
// the calling thread.  Nothing here is part of the product; the product links libamdhip64.
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>

extern "C" {
hipError_t hipMalloc(void **p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) std::memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t)
{
  for (size_t r = 0; r < h; ++r) std::memmove((char *)d + r * dp, (const char *)s + r * sp, w);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
static std::atomic<long> g_streams{0};
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)(uintptr_t)(++g_streams * 64); return hipSuccess; }
hipError_t hipStreamCreate(hipStream_t *s) { return hipStreamCreateWithFlags(s, 0); }
hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { return hipSuccess; }
struct FakeEvent { std::chrono::steady_clock::time_point t; };
hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t) new FakeEvent(); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete (FakeEvent *)e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { ((FakeEvent *)e)->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b)
{
  *ms = std::chrono::duration<float, std::milli>(((FakeEvent *)b)->t - ((FakeEvent *)a)->t).count();
  return hipSuccess;
}
// several fake devices (MM3D_FAKE_DEVICES, default 1): the current device is per thread, like HIP's; memory is host memory
// whichever "device" it belongs to, so a peer copy is a memmove
static thread_local int t_device = 0;
static int fake_device_count() { const char *e = std::getenv("MM3D_FAKE_DEVICES"); const int n = e ? std::atoi(e) : 1; return n > 0 ? n : 1; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= fake_device_count()) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipGetDeviceCount(int *n) { *n = fake_device_count(); return hipSuccess; }
hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t) { if (n) std::memmove(d, s, n); return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 256; return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "fake HIP runtime"; }
}
