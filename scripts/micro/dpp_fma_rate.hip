// Issue rate of fp64 FMAs on gfx950 with a row weight that is (a) a plain VGPR, (b) broadcast inside the instruction
// (v_fmac_f64_dpp row_newbcast), (c) an SGPR pair filled by two v_readlane_b32 per four FMAs.
// 8 waves per CU (two per SIMD) as in k_kopt2d_res, 8 independent accumulators per lane.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/dpp_fma_rate.hip -o /tmp/dpp && /tmp/dpp
#include <hip/hip_runtime.h>
#include <cstdio>
#define ITERS 4096
template <int MODE>
__global__ void __launch_bounds__(512) k(double* out, const double* in, long long* cyc) {
    double a[8], m[8];
    for (int i = 0; i < 8; ++i) a[i] = 0, m[i] = in[threadIdx.x + 64 * i];
    double y = in[threadIdx.x & 15];
    __syncthreads();
    const long long t0 = clock64();
    const long long w0 = wall_clock64();
    for (int it = 0; it < ITERS; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[i]) : "v"(y), "v"(m[i]));
        } else if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(y), "v"(m[i]));
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const double s = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(y), 3 + h),
                                                  __builtin_amdgcn_readlane(__double2loint(y), 3 + h));
                asm volatile("" ::"s"(s));
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[4 * h + i]) : "s"(s), "v"(m[4 * h + i]));
            }
        }
    }
    const long long t1 = clock64();
    const long long w1 = wall_clock64();
    double r = 0;
    for (int i = 0; i < 8; ++i) r += a[i];
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0, cyc[1] = w1 - w0;
}
int main() {
    double *out, *in;
    long long* cyc;
    hipMalloc(&out, 256 * 512 * 8);
    hipMalloc(&in, 4096 * 8);
    hipMemset(in, 0, 4096 * 8);
    hipMalloc(&cyc, 16);
    long long h[2];
    const char* names[3] = {"plain VGPR weight", "DPP row_newbcast", "2 readlane + 4 FMA (SGPR weight)"};
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) k<0><<<256, 512>>>(out, in, cyc);
            if (mode == 1) k<1><<<256, 512>>>(out, in, cyc);
            if (mode == 2) k<2><<<256, 512>>>(out, in, cyc);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
        printf("%-36s clock64 %lld  wall_clock64(100MHz) %lld  per FMA of a wave: %.2f clock64 ticks, %.2f ns (two waves per SIMD)\n",
               names[mode], h[0], h[1], (double)h[0] / (ITERS * 8), 10.0 * h[1] / (ITERS * 8));
    }
    return 0;
}
