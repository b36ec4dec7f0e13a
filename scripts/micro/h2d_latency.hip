// How long does a small host->device table take to arrive while a bulk D2H result copy is in flight?  (The batched
// call's table uploads were seen to start 1.3 ms late in the steady state of bench.py.)  Variants:
//   a  copy kernel reading a page-locked host block (gd_stage_h2d)          b  hipMemcpyAsync from page-locked memory
//   c  CPU stores straight into fine-grained DEVICE memory (large BAR), then a kernel reads it
//   d  the table as a kernel argument (4 KB by value)
// each with and without a 512-MB D2H copy running on another stream.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <chrono>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Tab { double v[500]; };
__global__ void k_copy(const double* __restrict__ src, double* __restrict__ dst, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ void k_arg(Tab t, double* __restrict__ dst) {
    int i = threadIdx.x;
    if (i < 500) dst[i] = t.v[i];
}
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t big = 512u << 20;
    const int n = 2500;  // 20 KB table
    void *d_big, *h_big, *h_tab, *d_tab, *d_fine = nullptr;
    CK(hipMalloc(&d_big, big)); CK(hipHostMalloc(&h_big, big, hipHostMallocDefault));
    CK(hipHostMalloc(&h_tab, n * 8, hipHostMallocDefault)); CK(hipMalloc(&d_tab, n * 8));
    hipError_t ef = hipExtMallocWithFlags(&d_fine, n * 8, hipDeviceMallocFinegrained);
    printf("fine-grained device block: %s\n", ef == hipSuccess ? "ok" : hipGetErrorString(ef));
    hipStream_t s, c; CK(hipStreamCreate(&s)); CK(hipStreamCreate(&c));
    std::vector<double> src(n, 1.0);
    Tab t; for (int i = 0; i < 500; ++i) t.v[i] = i;
    for (int busy = 0; busy < 2; ++busy) {
        for (char variant : {'a', 'b', 'c', 'd'}) {
            if (variant == 'c' && ef != hipSuccess) continue;
            double worst = 0, sum = 0; int reps = 6;
            for (int r = 0; r < reps; ++r) {
                CK(hipDeviceSynchronize());
                if (busy) CK(hipMemcpyAsync(h_big, d_big, big, hipMemcpyDeviceToHost, c));
                // let the copy get going
                double t0 = now(); while (now() - t0 < 1.0) {}
                t0 = now();
                if (variant == 'a') { memcpy(h_tab, src.data(), n * 8); k_copy<<<(n + 255) / 256, 256, 0, s>>>((double*)h_tab, (double*)d_tab, n); }
                if (variant == 'b') { memcpy(h_tab, src.data(), n * 8); CK(hipMemcpyAsync(d_tab, h_tab, n * 8, hipMemcpyHostToDevice, s)); }
                if (variant == 'c') { memcpy(d_fine, src.data(), n * 8); k_copy<<<(n + 255) / 256, 256, 0, s>>>((double*)d_fine, (double*)d_tab, n); }
                if (variant == 'd') { k_arg<<<1, 512, 0, s>>>(t, (double*)d_tab); }
                CK(hipStreamSynchronize(s));
                const double dt = now() - t0;
                sum += dt; if (dt > worst) worst = dt;
            }
            printf("bulk D2H in flight: %d  variant %c: mean %.3f ms  worst %.3f ms\n", busy, variant, sum / reps, worst);
        }
    }
    return 0;
}
