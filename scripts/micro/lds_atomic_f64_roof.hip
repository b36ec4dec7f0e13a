// LDS-atomic roof of the real-weight 2D binning: how many ds_add_f64 (and, for comparison, ds_add_u64 / ds_add_u32) lanes
// per clock a CU retires on uniformly random entries of a 128-KB table (16 384 doubles = one 64 x 256 stripe of a 256 x 256
// fp64 grid), with the block shape of the binning kernels: 1024 threads, 128 KB of LDS, one block per CU, no memory
// traffic (addresses and addends come from registers).  One JSON object per type on stdout.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o lds_atomic_f64_roof scripts/micro/lds_atomic_f64_roof.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class T>
__global__ void __launch_bounds__(1024) k_roof(const unsigned* __restrict__ seeds, int iters, double* __restrict__ sink) {
    extern __shared__ double sh_raw[];
    T* sh = reinterpret_cast<T*>(sh_raw);
    constexpr unsigned ENTRIES = 128 * 1024 / sizeof(T);
    for (unsigned i = threadIdx.x; i < ENTRIES; i += 1024) sh[i] = (T)0;
    __syncthreads();
    unsigned r[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) r[j] = seeds[(blockIdx.x * 16 + j) * 1024 + threadIdx.x];
    for (int it = 0; it < iters; ++it) {
        const unsigned s = (unsigned)it * 0x9E3779B1u;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const unsigned a = (r[j] ^ s) & (ENTRIES - 1);
            atomicAdd(&sh[a], (T)(1 + (r[j] >> 28)));
        }
    }
    __syncthreads();
    double acc = 0;
    for (unsigned i = threadIdx.x; i < ENTRIES; i += 1024) acc += (double)sh[i];
    sink[blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <class T>
static int run(const char* name, int cus, int iters, const unsigned* d_seeds, double* d_sink) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute((const void*)k_roof<T>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CHECK(hipEventRecord(e0));
        k_roof<T><<<cus, 1024, 128 * 1024>>>(d_seeds, iters, d_sink);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double adds = (double)cus * 1024.0 * 16.0 * iters;
    const double per_cu = adds / (best * 1e-3) / cus;
    printf("{\"type\": \"%s\", \"cus\": %d, \"ms\": %.4f, \"adds\": %.4g, \"adds_per_s_per_cu\": %.4g, \"lanes_per_clk_per_cu_at_2p4GHz\": %.3f, "
           "\"ms_for_1p2e10_adds\": %.3f}\n", name, cus, best, adds, per_cu, per_cu / 2.4e9, 1.2e10 / (per_cu * cus) * 1e3);
    return 0;
}

int main(int argc, char** argv) {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    std::vector<unsigned> seeds((size_t)cus * 16 * 1024);
    unsigned long long st = 88172645463325252ull;
    for (auto& v : seeds) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (unsigned)(st >> 11); }
    unsigned* d_seeds;
    double* d_sink;
    CHECK(hipMalloc(&d_seeds, seeds.size() * 4));
    CHECK(hipMalloc(&d_sink, (size_t)cus * 1024 * 8));
    CHECK(hipMemcpy(d_seeds, seeds.data(), seeds.size() * 4, hipMemcpyHostToDevice));
    if (run<double>("ds_add_f64", cus, iters, d_seeds, d_sink)) return 1;
    if (run<unsigned long long>("ds_add_u64", cus, iters, d_seeds, d_sink)) return 1;
    if (run<unsigned int>("ds_add_u32", cus, iters, d_seeds, d_sink)) return 1;
    if (run<float>("ds_add_f32", cus, iters, d_seeds, d_sink)) return 1;
    return 0;
}
