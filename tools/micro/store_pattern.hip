// How much does the shape of a wave's stores matter for a 537 MB write?  A: 8 rows x 128 B per store instruction (the MFMA-epilogue
// pattern of posmlp_kernels.hip), B: one whole 1 KB row per instruction, C: 2 rows x 512 B.  Same bytes, same grid.
// build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 -o /tmp/sp tools/micro/store_pattern.hip && /tmp/sp
#include <hip/hip_runtime.h>
#include <cstdio>
template <int PAT>
__global__ __launch_bounds__(256) void wr(float* __restrict__ out, int tiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int tile = blockIdx.x * 4 + wave; tile < tiles; tile += gridDim.x * 4) {
    float* base = out + (size_t)tile * 32 * 256;
    const float4 v = make_float4(tile, lane, 1.f, 2.f);
    if (PAT == 0) {
#pragma unroll
      for (int ni = 0; ni < 8; ++ni)
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) *reinterpret_cast<float4*>(base + ((lane >> 3) + 8 * ps) * 256 + ni * 32 + (lane & 7) * 4) = v;
    } else if (PAT == 1) {
#pragma unroll
      for (int r = 0; r < 32; ++r) *reinterpret_cast<float4*>(base + r * 256 + lane * 4) = v;
    } else {
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r2 = 0; r2 < 16; ++r2) *reinterpret_cast<float4*>(base + (2 * r2 + (lane >> 5)) * 256 + h * 128 + (lane & 31) * 4) = v;
    }
  }
}
int main() {
  const size_t M = 512 * 512 * 2;   // two [M,256] matrices = 537 MB
  float* out;
  (void)hipMalloc(&out, M * 256 * sizeof(float));
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int tiles = (int)(M / 32);
  for (int pat = 0; pat < 3; ++pat)
    for (int grid : {1024, 4096}) {
      for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        for (int i = 0; i < 10; ++i) {
          if (pat == 0) hipLaunchKernelGGL(wr<0>, dim3(grid), dim3(256), 0, 0, out, tiles);
          else if (pat == 1) hipLaunchKernelGGL(wr<1>, dim3(grid), dim3(256), 0, 0, out, tiles);
          else hipLaunchKernelGGL(wr<2>, dim3(grid), dim3(256), 0, 0, out, tiles);
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("pattern %c grid %d: %.1f us  %.2f TB/s\n", "ABC"[pat], grid, ms * 100, M * 1024.0 / (ms / 10 * 1e-3) / 1e12);
      }
    }
  return 0;
}
