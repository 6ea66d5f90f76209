// Sustained MFMA rate on the whole chip: v_mfma_f32_32x32x16_bf16 and v_mfma_f32_32x32x2_f32 from registers only.
// build: hipcc -O3 --offload-arch=gfx950 -o mfma_rate tools/micro/mfma_rate.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(256) void spin(float* out, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  uint4 a = {threadIdx.x, 1u, 2u, 3u}, b = {5u, threadIdx.x, 7u, 8u};
  float fa = threadIdx.x, fb = 1.f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[i], 0, 0, 0);
      }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  float* out;
  hipMalloc(&out, 4096 * 256 * sizeof(float));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int kind = 0; kind < 2; ++kind)
    for (int blocks : {256, 512}) {
      for (int iters : {2000, 20000}) {
        if (kind == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), 0, 0, out, 10); else hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, 0, out, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(spin<0>, dim3(blocks), dim3(256), 0, 0, out, iters); else hipLaunchKernelGGL(spin<1>, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 32 * (kind == 0 ? 32768.0 : 4096.0);
        printf("%s  blocks %d  iters %d: %.3f ms  %.1f TFLOP/s\n", kind == 0 ? "bf16 32x32x16" : "f32 32x32x2", blocks, iters, ms, flop / ms / 1e9);
      }
    }
  return 0;
}
