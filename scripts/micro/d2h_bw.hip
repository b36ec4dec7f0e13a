// D2H bandwidth probes: one hipMemcpyAsync, the same split over 2 / 4 streams, and a copy kernel writing page-locked
// host memory directly.  Build: hipcc --offload-arch=gfx950 -O3 d2h_bw.hip -o d2h_bw
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_copy(const double2* __restrict__ src, double2* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    const size_t bytes = 168u << 20;
    void *d, *h;
    CK(hipMalloc(&d, bytes));
    CK(hipHostMalloc(&h, bytes, hipHostMallocDefault));
    CK(hipMemset(d, 1, bytes));
    hipStream_t st[4];
    for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto now = [] { return std::chrono::steady_clock::now(); };
    for (int nst : {1, 2, 4}) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = now();
            for (int s = 0; s < nst; ++s)
                CK(hipMemcpyAsync((char*)h + s * (bytes / nst), (char*)d + s * (bytes / nst), bytes / nst, hipMemcpyDeviceToHost, st[s]));
            CK(hipDeviceSynchronize());
            double ms = std::chrono::duration<double, std::milli>(now() - t0).count();
            if (rep == 2) printf("hipMemcpyAsync x%d streams: %.2f ms  %.1f GB/s\n", nst, ms, bytes / ms / 1e6);
        }
    }
    for (int blocks : {64, 256, 1024}) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipDeviceSynchronize());
            auto t0 = now();
            k_copy<<<blocks, 256, 0, st[0]>>>((const double2*)d, (double2*)h, bytes / 16);
            CK(hipDeviceSynchronize());
            double ms = std::chrono::duration<double, std::milli>(now() - t0).count();
            if (rep == 2) printf("copy kernel %d blocks: %.2f ms  %.1f GB/s\n", blocks, ms, bytes / ms / 1e6);
        }
    }
    return 0;
}
