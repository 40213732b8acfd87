// Dependent-issue latencies on gfx950 that bound the 16 x 16 pivot loop of k_potrf_inv_mf: cycles (s_memtime) per link of a
// chain of v_fma_f64, of v_rsq_f64, of v_readlane + VALU, of MFMA f64 -> VALU -> MFMA (alone and with a second independent
// MFMA on the same pipe).  One wave.   hipcc --offload-arch=gfx950 -O3 tools/lat_bench.hip -o tools/lat_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double double4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlane_f64(double v, int src) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), src), __builtin_amdgcn_readlane(__double2loint(v), src));
}
template <int MODE>
__global__ __launch_bounds__(64) void k(double* io, unsigned long long* cyc) {
    double x = io[threadIdx.x], a = io[64 + threadIdx.x], b = io[128 + threadIdx.x];
    double4_t D = {x, a, b, x}, X = {a, b, x, a};
    __builtin_amdgcn_s_waitcnt(0);
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(x), "+v"(D), "+v"(X) :: "memory");
    constexpr int N = 128;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        if (MODE == 0) x = fma(x, a, b);
        if (MODE == 1) x = __builtin_amdgcn_rsq(x);
        if (MODE == 2) { const double s = readlane_f64(x, 5); x = fma(s, a, b); }
        if (MODE == 3) { D = __builtin_amdgcn_mfma_f64_16x16x4f64(x, a, D, 0, 0, 0); x = D[0] * b; }
        if (MODE == 4) { D = __builtin_amdgcn_mfma_f64_16x16x4f64(x, a, D, 0, 0, 0); X = __builtin_amdgcn_mfma_f64_16x16x4f64(x, b, X, 0, 0, 0); x = D[0] * b; }
        if (MODE == 5) { D = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, D, 0, 0, 0); }   // back-to-back accumulation, no VALU link
        if (MODE == 6) x = x * a;
        if (MODE == 7) { const double s = readlane_f64(x, 5); x = a * s; }
    }
    asm volatile("s_nop 0\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(x), "+v"(D), "+v"(X) :: "memory");
    io[192 + threadIdx.x] = x + D[0] + D[1] + D[2] + D[3] + X[0] + X[1];
    if (threadIdx.x == 0) cyc[MODE] = (t1 - t0) / N;
}
int main() {
    double* io; unsigned long long* cyc;
    hipMalloc(&io, 256 * 8); hipMalloc(&cyc, 8 * 8);
    double h[256]; for (int i = 0; i < 256; ++i) h[i] = 1.0 + 1e-3 * i;
    hipMemcpy(io, h, sizeof h, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(k<0>, 1, 64, 0, 0, io, cyc); hipLaunchKernelGGL(k<1>, 1, 64, 0, 0, io, cyc); hipLaunchKernelGGL(k<2>, 1, 64, 0, 0, io, cyc);
        hipLaunchKernelGGL(k<3>, 1, 64, 0, 0, io, cyc); hipLaunchKernelGGL(k<4>, 1, 64, 0, 0, io, cyc); hipLaunchKernelGGL(k<5>, 1, 64, 0, 0, io, cyc);
        hipLaunchKernelGGL(k<6>, 1, 64, 0, 0, io, cyc); hipLaunchKernelGGL(k<7>, 1, 64, 0, 0, io, cyc);
        hipDeviceSynchronize();
    }
    unsigned long long c[8]; hipMemcpy(c, cyc, sizeof c, hipMemcpyDeviceToHost);
    const char* name[8] = {"v_fma_f64 chain", "v_rsq_f64 chain", "readlane x2 + v_fma_f64", "mfma_f64_16x16x4 -> v_mul_f64 -> mfma", "same + one independent mfma per link",
                           "mfma back-to-back accumulate", "v_mul_f64 chain", "readlane x2 + v_mul_f64"};
    for (int m = 0; m < 8; ++m) printf("%-44s %4llu cycles per link\n", name[m], c[m]);
    return 0;
}
