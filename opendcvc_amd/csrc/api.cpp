// api.cpp - device discovery, pinned host memory and stream-ordered copies of the C ABI
// (include/dcvc_amd.h).  The error state lives in error.cpp (host-only: also part of the sanitizer build).
#include "common.hpp"

extern "C" {

int dcvc_device_count(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        dcvc::set_error("hipGetDeviceCount failed: %s", hipGetErrorString(e));
        return dcvc::E_HIP;
    }
    return n;
}

void* dcvc_host_alloc(size_t bytes)
{
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) {
        dcvc::set_error("hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

void dcvc_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

void* dcvc_host_device_ptr(void* host)
{
    void* d = nullptr;
    if (!host || hipHostGetDevicePointer(&d, host, 0) != hipSuccess) {
        dcvc::set_error("hipHostGetDevicePointer failed");
        return nullptr;
    }
    return d;
}

int dcvc_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream)
{
    DCVC_REQUIRE(dst_host && src_dev, "dcvc_memcpy_d2h: null pointer");
    DCVC_HIP(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
    return 0;
}

int dcvc_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream)
{
    DCVC_REQUIRE(dst_dev && src_host, "dcvc_memcpy_h2d: null pointer");
    DCVC_HIP(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return 0;
}

int dcvc_stream_sync(void* stream)
{
    DCVC_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

}  // extern "C"
