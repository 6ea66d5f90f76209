// Does the WIDTH of a lane's access matter for a kernel that streams many per-pixel planes (the step kernel of the none-mode loop reads
// 40-odd 4-byte-per-pixel streams)?  Same bytes (N pixels x 32 planes of 4 B read, 8 planes written), three shapes:
//   A: 32 plane streams, 4 B per lane and instruction (a wave covers 256 contiguous bytes)          -- what the step kernel does
//   B: the same planes, 8 B per lane (two adjacent pixels per lane)
//   C: planes interleaved four at a time ([8][N][4]): 16 B per lane
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 -o /tmp/ps tools/micro/plane_streams.hip && /tmp/ps
#include <hip/hip_runtime.h>
#include <cstdio>
constexpr int NP = 32, NW = 8;
template <int MODE>
__global__ __launch_bounds__(256, 4) void k(const float* __restrict__ in, float* __restrict__ out, long N) {
  if (MODE == 0) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) {
      float acc = 0.f;
#pragma unroll
      for (int p = 0; p < NP; ++p) acc += in[p * N + i];
#pragma unroll
      for (int p = 0; p < NW; ++p) out[p * N + i] = acc + p;
    }
  } else if (MODE == 1) {
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2; i < N; i += (long)gridDim.x * 512) {
      float2 acc = make_float2(0.f, 0.f);
#pragma unroll
      for (int p = 0; p < NP; ++p) { const float2 v = *reinterpret_cast<const float2*>(in + p * N + i); acc.x += v.x; acc.y += v.y; }
#pragma unroll
      for (int p = 0; p < NW; ++p) *reinterpret_cast<float2*>(out + p * N + i) = make_float2(acc.x + p, acc.y + p);
    }
  } else {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) {
      float acc = 0.f;
#pragma unroll
      for (int p = 0; p < NP / 4; ++p) { const float4 v = *reinterpret_cast<const float4*>(in + ((long)p * N + i) * 4); acc += (v.x + v.y) + (v.z + v.w); }
#pragma unroll
      for (int p = 0; p < NW / 4; ++p) *reinterpret_cast<float4*>(out + ((long)p * N + i) * 4) = make_float4(acc, acc + 1, acc + 2, acc + 3);
    }
  }
}
template <int MODE>
float run(const float* in, float* out, long N) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  const int grid = (int)((N + 511) / 512);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, N);
  hipEventRecord(a);
  for (int w = 0; w < 20; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256), 0, 0, in, out, N);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 20;
}
int main() {
  const long N = 8L * 512 * 512;
  float *in, *out;
  hipMalloc(&in, sizeof(float) * NP * N); hipMalloc(&out, sizeof(float) * NW * N);
  hipMemset(in, 0, sizeof(float) * NP * N);
  const double bytes = (double)(NP + NW) * 4 * N;
  const float t0 = run<0>(in, out, N), t1 = run<1>(in, out, N), t2 = run<2>(in, out, N);
  printf("bytes %.1f MB\n4 B/lane  %.1f us  %.2f TB/s\n8 B/lane  %.1f us  %.2f TB/s\n16 B/lane %.1f us  %.2f TB/s\n", bytes / 1e6, t0 * 1e3, bytes / t0 / 1e9,
         t1 * 1e3, bytes / t1 / 1e9, t2 * 1e3, bytes / t2 / 1e9);
  return 0;
}
