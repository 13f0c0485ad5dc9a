// A HOST stand-in for the 33 HIP runtime entry points libkraken_amd.so imports, for ONE purpose: running the C ABI's C++ side (plan
// compiler, weight packing, length / shape arithmetic, workspace sizing, argument checks, every launcher's host half) under the host
// AddressSanitizer on a machine without a GPU (GPU ASan needs xnack+ code objects this pool does not run, and
// AMD's ASan runtime intercepts the HSA allocator, so the real runtime cannot be driven under it either: profiles/r06_asan_host.txt).
// "Device" memory is host memory from malloc -- so ASan sees every hipMemcpy / hipMemset that runs past a packed-weight or workspace
// buffer -- and a kernel launch is a successful no-op: NOTHING is computed.  Test infrastructure; never linked into the product.
#include <cstdlib>
#include <cstring>
#include <cstdint>

extern "C" {
typedef int hipError_t;
typedef void* hipStream_t;
typedef void* hipEvent_t;
struct dim3_ { unsigned x, y, z; };

static long g_launches = 0, g_allocs = 0;
long fake_hip_launches() { return g_launches; }
long fake_hip_live_allocations() { return g_allocs; }

void** __hipRegisterFatBinary(const void*) { static void* h; return &h; }
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipUnregisterFatBinary(void**) {}
static thread_local struct { dim3_ g, b; size_t shmem; hipStream_t s; } g_cfg;
hipError_t __hipPushCallConfiguration(dim3_ g, dim3_ b, size_t shmem, hipStream_t s) { g_cfg = {g, b, shmem, s}; return 0; }
hipError_t __hipPopCallConfiguration(dim3_* g, dim3_* b, size_t* shmem, hipStream_t* s) {
    *g = g_cfg.g; *b = g_cfg.b; *shmem = g_cfg.shmem; *s = g_cfg.s; return 0;
}
hipError_t hipLaunchKernel(const void*, dim3_, dim3_, void**, size_t, hipStream_t) { ++g_launches; return 0; }
hipError_t hipFuncSetAttribute(const void*, int, int) { return 0; }

hipError_t hipGetDeviceCount(int* n) { *n = 1; return 0; }
hipError_t hipGetDevice(int* d) { *d = 0; return 0; }
hipError_t hipSetDevice(int d) { return d == 0 ? 0 : 101; }
hipError_t hipDeviceSynchronize() { return 0; }
hipError_t hipGetLastError() { return 0; }
const char* hipGetErrorString(hipError_t) { return "fake HIP runtime (tools/asan/fake_hip.cpp)"; }

hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); ++g_allocs; return *p ? 0 : 2; }
hipError_t hipMallocAsync(void** p, size_t n, hipStream_t) { return hipMalloc(p, n); }
hipError_t hipFree(void* p) { if (p) --g_allocs; free(p); return 0; }
hipError_t hipFreeAsync(void* p, hipStream_t) { return hipFree(p); }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(1, n ? n : 1); return *p ? 0 : 2; }
hipError_t hipHostFree(void* p) { free(p); return 0; }
hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return 0; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, int) { memmove(d, s, n); return 0; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, int, hipStream_t) { memmove(d, s, n); return 0; }
hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return 0; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return 0; }
hipError_t hipMemset2DAsync(void* d, size_t pitch, int v, size_t w, size_t h, hipStream_t) {
    for (size_t r = 0; r < h; ++r) memset(static_cast<char*>(d) + r * pitch, v, w);
    return 0;
}

hipError_t hipEventCreate(hipEvent_t* e) { *e = malloc(8); return 0; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = malloc(8); return 0; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return 0; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return 0; }
hipError_t hipEventSynchronize(hipEvent_t) { return 0; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return 0; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return 0; }
}
