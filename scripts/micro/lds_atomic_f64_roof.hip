// LDS-atomic roof of the real-weight 2D binning: how many ds_add_f64 (and, for comparison, ds_add_u64 / ds_add_u32) lanes
// per clock a CU retires on uniformly random entries of a 128-KB table (16 384 doubles = one 64 x 256 stripe of a 256 x 256
// fp64 grid), with the block shape of the binning kernels: 1024 threads, 128 KB of LDS, one block per CU, no memory
// traffic (addresses and addends come from registers).  One JSON object per type on stdout.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o lds_atomic_f64_roof scripts/micro/lds_atomic_f64_roof.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <class T>
__global__ void __launch_bounds__(1024) k_roof(const unsigned* __restrict__ seeds, int iters, double* __restrict__ sink, int scramble) {
    extern __shared__ double sh_raw[];
    T* sh = reinterpret_cast<T*>(sh_raw);
    constexpr unsigned ENTRIES = 128 * 1024 / sizeof(T);
    for (unsigned i = threadIdx.x; i < ENTRIES; i += 1024) sh[i] = (T)0;
    __syncthreads();
    unsigned r[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) r[j] = seeds[(blockIdx.x * 16 + j) * 1024 + threadIdx.x];
    for (int it = 0; it < iters; ++it) {
        const unsigned s = scramble ? (unsigned)it * 0x9E3779B1u : 0u;
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
        k_roof<T><<<cus, 1024, 128 * 1024>>>(d_seeds, iters, d_sink, strstr(name, "gaussian") == nullptr);
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
    {   // the same adds on the addresses a 2D Gaussian sample set produces in ONE 64 x 256 stripe of its grid (sigma = 28 bins
        // in x, rows uniform within the stripe's populated half): what the real-weight binning kernels see
        std::vector<unsigned> g(seeds.size());
        unsigned long long s2 = 0x9E3779B97F4A7C15ull;
        auto rnd = [&]() { s2 ^= s2 << 13; s2 ^= s2 >> 7; s2 ^= s2 << 17; return (unsigned)(s2 >> 11); };
        for (auto& v : g) {
            double u = (rnd() + 1.0) / 4294967297.0, w = rnd() / 4294967296.0;
            int x = (int)lrint(128 + 28 * sqrt(-2 * log(u)) * cos(6.283185307179586 * w));
            x = x < 0 ? 0 : (x > 255 ? 255 : x);
            double u2 = (rnd() + 1.0) / 4294967297.0, w2 = rnd() / 4294967296.0;
            int y = (int)lrint(fabs(28 * sqrt(-2 * log(u2)) * cos(6.283185307179586 * w2)));  // distance from the grid's centre row
            y = y > 63 ? 63 : y;
            v = (rnd() & 0xf0000000u) | (unsigned)(y * 256 + x);
        }
        CHECK(hipMemcpy(d_seeds, g.data(), g.size() * 4, hipMemcpyHostToDevice));
        if (run<double>("ds_add_f64 gaussian (fixed addresses per lane: xor with 0)", cus, iters, d_seeds, d_sink)) return 1;
        CHECK(hipMemcpy(d_seeds, seeds.data(), seeds.size() * 4, hipMemcpyHostToDevice));
    }
    if (run<unsigned long long>("ds_add_u64", cus, iters, d_seeds, d_sink)) return 1;
    if (run<unsigned int>("ds_add_u32", cus, iters, d_seeds, d_sink)) return 1;
    if (run<float>("ds_add_f32", cus, iters, d_seeds, d_sink)) return 1;
    return 0;
}
