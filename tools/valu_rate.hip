// Micro-benchmark: VALU issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 at 1/2/4/8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));

template <int ILP>
__global__ void k_fma(float* out, int iters, float a, float b) {
    float x[ILP];
    for (int i = 0; i < ILP; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    }
    float s = 0;
    for (int i = 0; i < ILP; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int ILP>
__global__ void k_pkfma(float* out, int iters, float a, float b) {
    float2v x[ILP];
    float2v av = {a, a * 1.0001f}, bv = {b, b * 0.999f};
    for (int i = 0; i < ILP; ++i) x[i] = float2v{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < ILP; ++i) x[i] = __builtin_elementwise_fma(x[i], av, bv);
    }
    float s = 0;
    for (int i = 0; i < ILP; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
float timeit(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    f();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
    const int iters = 20000;
    constexpr int ILP = 8;
    for (int wps : {1, 2, 4, 8}) {
        int blocks = 256 * wps;  // 256-thread blocks = 4 waves = 1 wave per SIMD per block
        float ms = timeit([&] { hipLaunchKernelGGL(k_fma<ILP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); });
        double inst = (double)iters * ILP * wps;  // wave-instructions per SIMD
        printf("v_fma_f32    waves/SIMD=%d  %.3f ms  cycles/inst/SIMD @2.4GHz = %.2f  TFLOP/s=%.1f\n", wps, ms, ms * 1e-3 * 2.4e9 / inst,
               inst * 1024 * 64 * 2 / (ms * 1e-3) / 1e12);
        ms = timeit([&] { hipLaunchKernelGGL(k_pkfma<ILP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); });
        printf("v_pk_fma_f32 waves/SIMD=%d  %.3f ms  cycles/inst/SIMD @2.4GHz = %.2f  TFLOP/s=%.1f\n", wps, ms, ms * 1e-3 * 2.4e9 / inst,
               inst * 1024 * 64 * 4 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
