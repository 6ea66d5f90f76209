// Sine-layer GEMMs of the PosMLP (mymodels/mlps.py:102-103 `SineLayer.forward = sin(linear(x))`, :216-236 the layer loop and
// the output heads, and their autograd backward), hand-written for gfx950.  Three families, one C ABI section at the end:
//   * exact-f32 MFMA (v_mfma_f32_32x32x2_f32): mlp_gemm_nt_wide / _pipe / mlp_gemm_nt, mlp_wgrad_tn -- any shape (round 1; notes below)
//   * split-operand products on the bf16 matrix pipe (v_mfma_f32_32x32x16_bf16, f32 = three bf16 pieces, 6 of 9 partial products,
//     f32 accumulate; as accurate as the exact-f32 kernels): mlp_nt_bx, mlp_wgrad_bx -- the 256-wide layers at image size, which
//     is where the iteration's time is (258 / 243 / 180 us per layer at 512x512 against 357 / 318 / 283)
//   * small point sets (<= 1024 rows, the envmap MLP): mlp_small_nt / mlp_small_tn; and the skinny ends at image size: output
//     layer + 'arm' head, its backward, the 5-row and 15-column weight gradients (mlp_skinny_*), HBM-bound streaming kernels
//
// The coordinate MLP of hot loop B runs over M = H*W points (262 144 at 512x512) with layers of width <= 256: nine
// [M,256]x[256,256] products per iteration (311 GFLOP) plus, in a stock composition, one full pass over an [M,256] matrix for
// every sin, every d_y*cos(pre) and every bias gradient.  Here those passes live in the GEMM epilogues:
//
//   mlp_gemm_nt<EPI_SINCOS>   S, C = sin, cos(X W^T + b)         forward of one sine layer; C is kept for the backward, pre is not
//   mlp_gemm_nt<EPI_MULC>     G' = (G Wt^T) * C'  (+ column sums of G' per workgroup: the bias gradient of the layer below)
//   mlp_wgrad_tn              dW partials = G^T X over a slab of rows; mlp_wgrad_reduce adds the slabs (fixed order)
//
// All matrices are row-major fp32 with a leading dimension that is a multiple of 4 floats (activations: 256, padded).
// A 32x32x2 MFMA takes one f32 per lane for A and one for B (lane l: A[row l&31][k l>>5], B[k l>>5][col l&31]); K <= 256 runs in
// k-tiles of 32 through LDS, global -> registers -> LDS with the fetch one k-tile ahead.  Three kernels implement the NT product:
//   mlp_gemm_nt_wide   128 rows x 256 columns per 256-thread workgroup (2x2 waves of 64x128), 2 per CU: the 256-wide layers
//   mlp_gemm_nt_pipe   128 x 128 tiles, double-buffered LDS, the previous tile's epilogue interleaved: layers of <= 128 outputs
//   mlp_gemm_nt        any shape, ragged rows
// Measured at M = 512x512 (tools/mlp_bench.py): forward layer 369 us (BLAS product + sin pass: 424), dL/d input 327 us (BLAS + the
// d_y cos(pre) pass + the bias-gradient pass: 555), weight gradient 303 + 21 us (split-K BLAS: 308); the MFMA pipe is busy
// 57-77 % of the time (the floor of 8.4 M MFMAs x 64 cycles on 1024 SIMDs is 238 us at 2.2 GHz).
// What was tried on top of mlp_gemm_nt_wide and did not pay (bias-only epilogue, 347 us): cycle stamps give 470 (fetch issue) + 8280
// (128 MFMAs) + 1250 (mask + 12 ds_write_b128) + 180 (two barriers) cycles per k-tile for a workgroup alone on its CU, and two
// co-resident workgroups run in lock-step (both multiplying at half rate, then both stashing), so the pair is no faster than one.
// Offsetting half of the workgroups with s_sleep, s_setprio by block index or by hardware wave slot: no change.  One 512-thread
// workgroup per CU with both k-tile buffers in LDS (stash and fetch free of barriers against the products): 366 us.  Writing 1/16 of
// the outputs: 310 us, i.e. the 4-byte stores of the MFMA register layout cost 37 us per output matrix, and the k-loop itself sits
// at the 303 us of mlp_wgrad_tn.  The epilogue now goes through a per-wave LDS transpose (16-byte stores and 16-byte loads of the cos
// factors): dL/d input 349 -> 327 us, the forward unchanged (371 -> 369 us).  Also without effect: stripping the k-loop from 350
// to 110 non-MFMA instructions (no k masks, no modulo, scalar-base addressing), a one-workgroup-per-CU variant with both k-tile
// buffers in LDS and the staging interleaved with the MFMA groups (9750 instead of 10 200 cycles per k-tile; slower overall),
// iglp_opt / sched_group_barrier orderings, explicit double-buffered operand fragments (hipcc re-orders them anyway), running the
// weight gradient on a second stream, and one workgroup per CU with 512 registers, both LDS buffers and the fetch TWO k-tiles ahead
// (exact vmcnt counts in the loop: 389 us).  Cycle stamps per workgroup show the two co-resident workgroups entering together and each
// k-tile step taking 12.5-20.7 thousand cycles while both are active (10.2 thousand alone), epilogues 13-43 thousand (4.2 alone).  Ablation of mlp_gemm_nt_pipe
// (bias epilogue): MFMA + LDS reads + barriers + stores 298 us, + LDS writes 319, + weight fetch 327, + activation fetch 375; its
// sin/cos epilogue adds 35 us although it rides inside the next tile's MFMA stream.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <type_traits>

#include "../../include/matpbr.h"
#include "../../include/matpbr_experimental.h"
#include "posmlp_device.hpp"

namespace {


constexpr int kBM = 128, kBN = 128, kBK = 32;
constexpr int kPersistent = 512;   // two workgroups per CU x 256 CUs
constexpr int EPI_SINCOS = 0, EPI_MULC = 1, EPI_BIAS = 2;

// the output layer handed to the forward kernel of the last sine layer, which then finishes the network in its epilogue
struct HeadArgs {
  const float* w;      // [5, ldw >= 256]
  int ldw;
  const float* bias;   // [5]
  ArmHead h;
};

// Columns at or beyond N of the 256-wide outputs are left alone: the buffer of a skip layer keeps x0 there (mymodels/mlps.py
// :214-217 concatenates it every forward; a caller that owns the buffers writes it once).
// out[m][N + j] = tail[m][j], j < 256 - N: the x0 columns of a skip layer's buffer, rewritten after a forward that stored whole
// 16-byte words over them (guarding the one straddling word of every row in the epilogue costs ~30 us per layer, this pass ~8)
__global__ __launch_bounds__(256) void mlp_tail_copy_kernel(float* __restrict__ out, int ldo, const float* __restrict__ tail, int ldt, long M, int N) {
  const int w = 256 - N;
  const long total = M * w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long m = i / w;
    const int j = (int)(i - m * w);
    out[m * ldo + N + j] = tail[m * ldt + j];
  }
}
__device__ __forceinline__ void store4_upto(float* dst, float4 v, int valid, bool all) {   // all: uniform (N == 256), a scalar branch
  if (all) {
    *reinterpret_cast<float4*>(dst) = v;
    return;
  }
  if (valid >= 4) {
    *reinterpret_cast<float4*>(dst) = v;
  } else {
    if (valid > 0) dst[0] = v.x;
    if (valid > 1) dst[1] = v.y;
    if (valid > 2) dst[2] = v.z;
  }
}

struct NtArgs {
  const float* A;      // [M, lda]   rows = points
  const float* B;      // [N, ldb]   rows = output features, k contiguous
  const float* bias;   // [N]                     (EPI_SINCOS / EPI_BIAS)
  const float* cmul;   // [M, ldo]   cos(pre) of the layer below (EPI_MULC)
  float* out0;         // [M, ldo]   S | G' | pre
  float* out1;         // [M, ldo]   C            (EPI_SINCOS)
  float* colsum;       // [gridDim.x, 256] per-workgroup column sums of out0 (EPI_MULC), may be null
  int M;
  int N, K, lda, ldb, ldo;
  const float* tail;   // nullable [M, ldt]: the values that belong in columns N.. of out0 (x0 of a skip layer).  Given: the epilogue
  int ldt;             // stores whole 16-byte words over them and the caller rewrites them (mlp_tail_copy_kernel); null: they are guarded
  int cmul_sin;        // EPI_MULC: `cmul` holds the SINES of the layer below with the sign of their cosine in the last mantissa bit
  const float* x0;     // mlp_nt_bx<.., W0>: [M, ldx0 >= 16] the network's input rows (columns beyond d0 zero)
  int ldx0;
  float* w0_part;      // mlp_nt_bx<.., W0>: [4 gridDim.x][16][256] partial first-layer weight gradients (the layout of the skinny kernels)
  const unsigned* a_tmax;   // mlp_nt_gx<EPI_MULC, 3>: [M / 128] max |A| of every 128-row tile (f32 bit patterns): the tile's block exponent
  unsigned* o_tmax;         // nullable: the same of the rows written (atomic max into a zeroed array: the next product's a_tmax)
};

// LDS image of a k-tile: [row][k] with a pitch of 36 floats: 16-byte writes and 16-byte reads are both conflict-free (8 lanes x
// 4 banks cover the 32 banks; SQ_LDS_BANK_CONFLICT = 0).  Lane half h of an MFMA takes k = 16 h + step inside the tile (A and
// B agree, so the sum over k is unchanged), which makes a lane's 16 operands of a k-tile four consecutive float4.
constexpr int kLd = 36;
constexpr int kTileFloats = kBM * kLd;

// Global -> register fetch of one k-tile (4 row chunks of A, 4 of B per thread: 16 bytes each).  NO predicates and no branches:
// addresses are clamped into the matrix and the out-of-range elements are zeroed later, in `stash`, when the data has arrived.
// A predicate (or a select on the loaded value) next to the load makes the compiler wait for the fetch it has just issued -- and a
// load inside a branch makes every later s_waitcnt a vmcnt(0): measured 4800-7000 stall cycles per 4200-cycle k-tile.
__device__ __forceinline__ void fetch_a(const NtArgs& p, float4 (&ra)[4], int row0, int kslot, int crow, int ck) {
  int k = kslot * kBK + ck;
  const int kmax = ((p.K + 3) & ~3) - 4;
  k = k < kmax ? k : kmax;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int m = row0 + crow + 32 * q;
    m = m < p.M ? m : p.M - 1;
    ra[q] = *reinterpret_cast<const float4*>(p.A + (size_t)m * p.lda + k);
  }
}

__device__ __forceinline__ void fetch_b(const NtArgs& p, float4 (&rb)[4], int col0, int kslot, int crow, int ck) {
  int k = kslot * kBK + ck;
  const int kmax = ((p.K + 3) & ~3) - 4;
  k = k < kmax ? k : kmax;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    int n = col0 + crow + 32 * q;
    n = n < p.N ? n : p.N - 1;
    rb[q] = *reinterpret_cast<const float4*>(p.B + (size_t)n * p.ldb + k);
  }
}

__device__ __forceinline__ float4 kmask(float4 v, int k, int K, bool keep) {
  v.x = (keep && k < K) ? v.x : 0.f;
  v.y = (keep && k + 1 < K) ? v.y : 0.f;
  v.z = (keep && k + 2 < K) ? v.z : 0.f;
  v.w = (keep && k + 3 < K) ? v.w : 0.f;
  return v;
}

template <bool ROWS_FULL>
__device__ __forceinline__ void stash(const NtArgs& p, const float4 (&ra)[4], const float4 (&rb)[4], float* sA, float* sB, int row0, int col0,
                                      int kslot, int crow, int ck) {
  const int k = kslot * kBK + ck;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool mrow = ROWS_FULL || (row0 + crow + 32 * q < p.M);
    *reinterpret_cast<float4*>(sA + (crow + 32 * q) * kLd + ck) = kmask(ra[q], k, p.K, mrow);
    *reinterpret_cast<float4*>(sB + (crow + 32 * q) * kLd + ck) = kmask(rb[q], k, p.K, col0 + crow + 32 * q < p.N);
  }
}

// 64 MFMAs of one k-tile on a wave's 64x64 block; `mid(j)` runs after each of the four operand groups
template <class Mid>
__device__ __forceinline__ void nt_tile_mma(f32x16 (&acc)[2][2], const float* pa, const float* pb, Mid&& mid) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 a4[2], b4[2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) a4[mi] = *reinterpret_cast<const float4*>(pa + mi * 32 * kLd + 4 * j);
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) b4[ni] = *reinterpret_cast<const float4*>(pb + ni * 32 * kLd + 4 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const float av = e == 0 ? a4[mi].x : e == 1 ? a4[mi].y : e == 2 ? a4[mi].z : a4[mi].w;
          const float bv = e == 0 ? b4[ni].x : e == 1 ? b4[ni].y : e == 2 ? b4[ni].z : b4[ni].w;
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi][ni], 0, 0, 0);
        }
    mid(j);
  }
}

// Work items are (row tile, 128-column half).  Ids b and b + 8 sit on the same XCD (ids are dealt round-robin over the 8 XCDs) and
// take the two halves of one row tile, so its A rows cross the fabric once and the second read hits that XCD's L2.  A workgroup
// keeps one column half over its persistent loop (gridDim.x is a multiple of 16 whenever there are two halves).
struct TileMap {
  int halves, half, col0, my_count;
  __device__ TileMap(int rtiles, int N) {
    halves = (N + kBN - 1) / kBN;
    const int ntiles = (halves == 2) ? ((rtiles + 7) / 8) * 16 : rtiles;
    half = (halves == 2) ? (int)((blockIdx.x >> 3) & 1) : 0;
    col0 = half * kBN;
    my_count = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    // the last group of 16 ids may name row tiles past the end; a workgroup's tiles ascend, so those are a suffix of its list
    while (my_count > 0 && row_of(my_count - 1) >= rtiles * kBM) --my_count;
  }
  __device__ int row_of(int j) const {                 // first row of this workgroup's j-th tile (may lie past M: an empty tile)
    const int id = (int)blockIdx.x + j * (int)gridDim.x;
    return ((halves == 2) ? (id >> 4) * 8 + (id & 7) : id) * kBM;
  }
};

// General shapes (K <= 224: the first layer, the backward of the 5-wide output layer, narrow networks; and the ragged last row
// tile of every layer): one tile at a time, double-buffered LDS, one barrier per k-tile, predicated epilogue.
template <int EPI>
__global__ __launch_bounds__(256, 2) void mlp_gemm_nt(NtArgs p) {
  __shared__ __attribute__((aligned(16))) float sA[2][kTileFloats];
  __shared__ __attribute__((aligned(16))) float sB[2][kTileFloats];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int crow = tid >> 3, ck = (tid & 7) * 4;   // staging: 4 consecutive k of one row per chunk, 8 chunks per 32-wide row
  const int nk = (p.K + kBK - 1) / kBK;
  const int aoff = (wm * 64 + li) * kLd + 16 * lh, boff = (wn * 64 + li) * kLd + 16 * lh;
  const TileMap tm((p.M + kBM - 1) / kBM, p.N);
  const int col0 = tm.col0;
  float csum[2] = {0.f, 0.f};

  for (int j = 0; j < tm.my_count; ++j) {
    const int row0 = tm.row_of(j);
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    float4 ra[4], rb[4];
    __syncthreads();                                // the previous tile's readers are done with both buffers
    fetch_a(p, ra, row0, 0, crow, ck);
    fetch_b(p, rb, col0, 0, crow, ck);
    stash<false>(p, ra, rb, sA[0], sB[0], row0, col0, 0, crow, ck);
    fetch_a(p, ra, row0, nk > 1 ? 1 : 0, crow, ck);
    fetch_b(p, rb, col0, nk > 1 ? 1 : 0, crow, ck);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      nt_tile_mma(acc, sA[cur] + aoff, sB[cur] + boff, [](int) {});
      const int k1 = kt + 1 < nk ? kt + 1 : kt, k2 = kt + 2 < nk ? kt + 2 : nk - 1;   // past the end: harmless re-fetch / re-stash
      stash<false>(p, ra, rb, sA[cur ^ 1], sB[cur ^ 1], row0, col0, k1, crow, ck);
      fetch_a(p, ra, row0, k2, crow, ck);
      fetch_b(p, rb, col0, k2, crow, ck);
      __syncthreads();
    }

    // epilogue: accumulator register r of lane (li, lh) is row (r&3) + 8 (r>>2) + 4 lh, column li of its 32x32 tile.
    // gfx950 retires loads and stores through one in-order counter, so every load of the tile is issued (clamped addresses,
    // no branch) before its first store.
    int ncol[2];
    bool nok[2];
    float bn[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
      const int n = col0 + wn * 64 + ni * 32 + li;
      nok[ni] = n < p.N;
      ncol[ni] = nok[ni] ? n : p.N - 1;
      bn[ni] = (EPI != EPI_MULC) ? p.bias[ncol[ni]] : 0.f;
    }
    if (row0 + kBM <= p.M && p.ldo >= col0 + kBN) {
      // interior tile whose 128 columns all exist (those past N are left alone): the per-wave LDS transpose of mlp_gemm_nt_wide, 16-byte
      // stores and 16-byte loads of the cos factors.  Both k-tile buffers are idle here (the next tile starts with a barrier).
      float* scr = sA[0] + wave * (32 * kLd);
      const int t_row = lane >> 3, t_col = (lane & 7) * 4;
      float4 cs4[2] = {make_float4(0.f, 0.f, 0.f, 0.f), make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const size_t tile_row = (size_t)(row0 + wm * 64 + mi * 32 + t_row) * p.ldo;
        float4 cv4[2][4];
        if (EPI == EPI_MULC) {
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
              cv4[ni][ps] = *reinterpret_cast<const float4*>(p.cmul + tile_row + (size_t)(8 * ps) * p.ldo + col0 + wn * 64 + ni * 32 + t_col);
        }
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const size_t o0 = tile_row + col0 + wn * 64 + ni * 32 + t_col;
          float second[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[mi][ni][r];
            if (EPI == EPI_SINCOS) {
              sincos_cw(v + bn[ni], v, second[r]);
            } else if (EPI == EPI_BIAS) {
              v += bn[ni];
            }
            scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * kLd + li] = v;
          }
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            float4 v = *reinterpret_cast<const float4*>(scr + (t_row + 8 * ps) * kLd + t_col);
            if (EPI == EPI_MULC) {
              v.x *= cv4[ni][ps].x; v.y *= cv4[ni][ps].y; v.z *= cv4[ni][ps].z; v.w *= cv4[ni][ps].w;
              cs4[ni].x += v.x; cs4[ni].y += v.y; cs4[ni].z += v.z; cs4[ni].w += v.w;
            }
            store4_upto(p.out0 + o0 + (size_t)(8 * ps) * p.ldo, v, p.N - (col0 + wn * 64 + ni * 32 + t_col), p.N >= 256 || p.tail != nullptr);
          }
          if (EPI == EPI_SINCOS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * kLd + li] = second[r];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
              store4_upto(p.out1 + o0 + (size_t)(8 * ps) * p.ldo, *reinterpret_cast<const float4*>(scr + (t_row + 8 * ps) * kLd + t_col),
                          p.N - (col0 + wn * 64 + ni * 32 + t_col), p.N >= 256 || p.tail != nullptr);   // the cos tail is scratch
          }
        }
      }
      if (EPI == EPI_MULC) {
        // fold the transposed column sums (4 columns x 8 row slots per lane) back to the per-column accumulators of the lane layout:
        // through the same scratch, one row per row slot
        __syncthreads();
        float* fold = sB[0];            // [4 waves][8][64]
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) *reinterpret_cast<float4*>(fold + (wave * 8 + t_row) * 64 + ni * 32 + t_col) = cs4[ni];
        __syncthreads();
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          float t = 0.f;
          if (lh == 0) {
#pragma unroll
            for (int rs = 0; rs < 8; ++rs) t += fold[(wave * 8 + rs) * 64 + ni * 32 + li];
          }
          csum[ni] += nok[ni] ? t : 0.f;
        }
      }
    } else {
      float cv[2][2][16];
      if (EPI == EPI_MULC) {
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              int m = row0 + wm * 64 + mi * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
              m = m < p.M ? m : p.M - 1;
              cv[mi][ni][r] = p.cmul[(size_t)m * p.ldo + ncol[ni]];
            }
      }
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
          const int mb = row0 + wm * 64 + mi * 32 + 4 * lh;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mb + (r & 3) + 8 * (r >> 2);
            float v0, v1 = 0.f;
            if (EPI == EPI_SINCOS) {
              sincos_cw(acc[mi][ni][r] + bn[ni], v0, v1);
            } else if (EPI == EPI_BIAS) {
              v0 = acc[mi][ni][r] + bn[ni];
            } else {
              v0 = acc[mi][ni][r] * cv[mi][ni][r];
            }
            if (nok[ni] && m < p.M) {
              const size_t o = (size_t)m * p.ldo + ncol[ni];
              p.out0[o] = v0;
              if (EPI == EPI_SINCOS) p.out1[o] = v1;
              if (EPI == EPI_MULC) csum[ni] += v0;
            }
          }
        }
    }
  }
  if (EPI == EPI_MULC && p.colsum != nullptr) {
    // column sums over every row this workgroup produced (it keeps one column half): 2 lane halves x 2 row-waves -> LDS, fixed order
    float* red = sA[0];   // [4][128]: slot = wm*2 + lh
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) red[(wm * 2 + lh) * 128 + wn * 64 + ni * 32 + li] = csum[ni];
    __syncthreads();
    const int c = tid & 127;
    const float v = (red[c] + red[128 + c]) + (red[256 + c] + red[384 + c]);
    p.colsum[(size_t)blockIdx.x * 256 + tid] = ((tid >> 7) == tm.half) ? v : 0.f;
  }
}

// The hot shape: 224 < K <= 256 (8 k-tiles: every hidden layer), M a multiple of 128 (the host sends a ragged last tile to
// mlp_gemm_nt), output row stride >= 128 x halves (every column of a half exists; columns N.. of the outputs are scratch).
//   * no branch and no predicate anywhere in the steady state, so every s_waitcnt carries an exact count;
//   * the k-tiles of consecutive row tiles form one stream through the two LDS buffers: A rows (HBM) are fetched two k-tiles
//     ahead into alternating register sets, the weights (L2) one ahead, B before A so that waiting for B does not wait for A;
//   * the epilogue of tile t (sincos / * cos, stores) is cut into 8 pieces that ride inside the 8 k-tile steps of tile t+1: one
//     wave keeps the matrix pipe (4 MFMAs = 256 cycles per step, 8 of them holding the issue port) and the VALU + store path
//     busy together, and the output leaves as a steady stream.  The cos factors of piece j are loaded at the top of step j, ahead of
//     that step's stores (loads and stores retire through one in-order counter on gfx950).
template <int EPI>
__global__ __launch_bounds__(256, 2) void mlp_gemm_nt_pipe(NtArgs p) {
  __shared__ __attribute__((aligned(16))) float sA[2][kTileFloats];
  __shared__ __attribute__((aligned(16))) float sB[2][kTileFloats];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int crow = tid >> 3, ck = (tid & 7) * 4;
  const int aoff = (wm * 64 + li) * kLd + 16 * lh, boff = (wn * 64 + li) * kLd + 16 * lh;
  const TileMap tm(p.M / kBM, p.N);
  const int col0 = tm.col0;
  if (tm.my_count == 0) {                            // uniform; such a workgroup still owes its (zero) column-sum row
    if (EPI == EPI_MULC && p.colsum != nullptr) p.colsum[(size_t)blockIdx.x * 256 + tid] = 0.f;
    return;
  }
  const int last = tm.my_count - 1;

  // addressing: one 32-bit per-lane offset per stream, everything else uniform (scalar base + immediate), rows >= N of the
  // weights clamped once for the whole kernel
  const int a_lane = crow * p.lda + ck;
  int b_lane[4];
  bool b_keep[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int n = col0 + crow + 32 * q;
    b_keep[q] = n < p.N;
    b_lane[q] = (b_keep[q] ? n : p.N - 1) * p.ldb + ck;
  }
  const int st_lane = crow * kLd + ck;
  int o_lane[2];
  float bn[2], csum[2] = {0.f, 0.f};
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const int n = col0 + wn * 64 + ni * 32 + li;
    o_lane[ni] = (wm * 64 + 4 * lh) * p.ldo + n;
    bn[ni] = (EPI != EPI_MULC) ? p.bias[n < p.N ? n : p.N - 1] : 0.f;
  }

  auto fetchA = [&](float4 (&ra)[4], int row0, int kslot) {
    const float* base = p.A + (size_t)row0 * p.lda + kslot * kBK;
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[q] = *reinterpret_cast<const float4*>(base + (size_t)(32 * q) * p.lda + a_lane);
  };
  auto fetchB = [&](float4 (&rb)[4], int kslot) {
    const float* base = p.B + kslot * kBK;
#pragma unroll
    for (int q = 0; q < 4; ++q) rb[q] = *reinterpret_cast<const float4*>(base + b_lane[q]);
  };
  auto stashAB = [&](const float4 (&ra)[4], const float4 (&rb)[4], float* dA, float* dB, int kslot) {
    const int k = kslot * kBK + ck;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      *reinterpret_cast<float4*>(dA + 32 * q * kLd + st_lane) = kmask(ra[q], k, p.K, true);
      *reinterpret_cast<float4*>(dB + 32 * q * kLd + st_lane) = kmask(rb[q], k, p.K, b_keep[q]);
    }
  };

  f32x16 prev[2][2];
  int prev_row0 = 0;
  // one eighth of the previous tile's epilogue: tile (mi, ni) = (j>>2, (j>>1)&1), registers 8 (j&1) .. +8
  auto piece_load = [&](int j, float (&cv)[8]) {
    if (EPI != EPI_MULC) return;
    const int mi = j >> 2, ni = (j >> 1) & 1, r0 = (j & 1) * 8;
    const float* base = p.cmul + (size_t)(prev_row0 + mi * 32) * p.ldo;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int r = r0 + q;
      cv[q] = base[(size_t)((r & 3) + 8 * (r >> 2)) * p.ldo + o_lane[ni]];
    }
  };
  // ... stored two values at a time (quarter h of piece j), one quarter after each operand group of the running k-tile, fenced
  // so that the scheduler does not merge the sincos chains of a whole piece (that needs ~50 more registers and spills)
  auto piece_store = [&](int j, int h, const float (&cv)[8]) {
    const int mi = j >> 2, ni = (j >> 1) & 1, r0 = (j & 1) * 8;
    float* base0 = p.out0 + (size_t)(prev_row0 + mi * 32) * p.ldo;
    float* base1 = (EPI == EPI_SINCOS) ? p.out1 + (size_t)(prev_row0 + mi * 32) * p.ldo : nullptr;
#pragma unroll
    for (int q = 2 * h; q < 2 * h + 2; ++q) {
      const int r = r0 + q;
      const size_t o = (size_t)((r & 3) + 8 * (r >> 2)) * p.ldo + o_lane[ni];
      const float v = prev[mi][ni][r];
      if (EPI == EPI_SINCOS) {
        float sv, cs;
        sincos_cw(v + bn[ni], sv, cs);
        base0[o] = sv;
        base1[o] = cs;
      } else if (EPI == EPI_BIAS) {
        base0[o] = v + bn[ni];
      } else {
        const float g = v * cv[q];
        base0[o] = g;
        csum[ni] += g;
      }
    }
  };

  // k-tiles are walked in an order rotated by the row-tile index, so that workgroups do not all pull the same 128-byte column
  // slot of their 1-KB rows at the same time
  auto slot = [&](int row0, int kt) { return (kt + (row0 >> 7)) & 7; };

  float4 ra[4], rb[4];
  {
    const int r0 = tm.row_of(0);
    fetchA(ra, r0, slot(r0, 0));
    fetchB(rb, slot(r0, 0));
    stashAB(ra, rb, sA[0], sB[0], slot(r0, 0));
    fetchB(rb, slot(r0, 1));
    fetchA(ra, r0, slot(r0, 1));
  }
  __syncthreads();

  auto tile = [&](int j, auto with_prev) {
    constexpr bool kPrev = decltype(with_prev)::value;
    const int row0 = tm.row_of(j), row1 = tm.row_of(j < last ? j + 1 : last);   // past the end: re-read the last tile (unused)
    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
#pragma unroll
    for (int kt = 0; kt < 8; ++kt) {
      const int cur = kt & 1;                        // 8 k-tiles per row tile: the buffer parity restarts with every tile
      float cv[8];
      if (kPrev) piece_load(kt, cv);
      nt_tile_mma(acc, sA[cur] + aoff, sB[cur] + boff, [&](int h) {
        if (kPrev) piece_store(kt, h, cv);
      });
      const int rs = kt < 7 ? row0 : row1, r2 = kt < 6 ? row0 : row1;
      stashAB(ra, rb, sA[cur ^ 1], sB[cur ^ 1], slot(rs, (kt + 1) & 7));
      fetchB(rb, slot(r2, (kt + 2) & 7));
      fetchA(ra, r2, slot(r2, (kt + 2) & 7));
      __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) prev[mi][ni] = acc[mi][ni];
    prev_row0 = row0;
  };

  tile(0, std::false_type{});
  for (int j = 1; j <= last; ++j) tile(j, std::true_type{});
  // drain: the last tile's epilogue
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    float cv[8];
    piece_load(j, cv);
#pragma unroll
    for (int h = 0; h < 4; ++h) piece_store(j, h, cv);
  }
  if (EPI == EPI_MULC && p.colsum != nullptr) {
    float* red = sA[0];   // [4][128]: slot = wm*2 + lh
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) red[(wm * 2 + lh) * 128 + wn * 64 + ni * 32 + li] = csum[ni];
    __syncthreads();
    const int c = tid & 127;
    const float v = (red[c] + red[128 + c]) + (red[256 + c] + red[384 + c]);
    p.colsum[(size_t)blockIdx.x * 256 + tid] = ((tid >> 7) == tm.half) ? v : 0.f;
  }
}

// Full-width variant: a workgroup owns 128 rows x 256 columns (2x2 waves of 64x128 = 2x4 MFMA tiles, 128 accumulator registers),
// the tiling of mlp_wgrad_tn: 128 MFMAs per wave between barriers instead of 64 and each activation row fetched once instead of
// once per column half.  No room for the previous tile's accumulators, so the epilogue is not interleaved with the next tile: it
// overlaps with the co-resident workgroup's products instead.  Same preconditions as mlp_gemm_nt_pipe, N > 128.
template <int EPI>
__global__ __launch_bounds__(256, 2) void mlp_gemm_nt_wide(NtArgs p) {
  __shared__ __attribute__((aligned(16))) float sA[kBM * kLd];
  __shared__ __attribute__((aligned(16))) float sB[256 * kLd];
  __shared__ __attribute__((aligned(16))) float sScr[4 * 32 * kLd];   // per-wave transpose scratch of the epilogue
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int crow = tid >> 3, ck = (tid & 7) * 4;
  const float* pa = sA + (wm * 64 + li) * kLd + 16 * lh;
  const float* pb = sB + (wn * 128 + li) * kLd + 16 * lh;
  const int rtiles = p.M / kBM;
  const int nk = (p.K + kBK - 1) / kBK;
  const int my_count = ((int)blockIdx.x < rtiles) ? (rtiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
  if (my_count == 0) {
    if (EPI == EPI_MULC && p.colsum != nullptr) p.colsum[(size_t)blockIdx.x * 256 + tid] = 0.f;
    return;
  }
  auto row_of = [&](int j) { return ((int)blockIdx.x + (j < my_count ? j : my_count - 1) * (int)gridDim.x) * kBM; };
  auto slot = [&](int row0, int kt) { return (kt + (row0 >> 7)) % nk; };

  const int a_lane = crow * p.lda + ck;
  int b_lane[8];
  bool b_keep[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int n = crow + 32 * q;
    b_keep[q] = n < p.N;
    b_lane[q] = (b_keep[q] ? n : p.N - 1) * p.ldb + ck;
  }
  const int st_lane = crow * kLd + ck;
  float bn[4];
  float4 csum4[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = wn * 128 + ni * 32 + li;
    bn[ni] = (EPI != EPI_MULC) ? p.bias[n < p.N ? n : p.N - 1] : 0.f;
    csum4[ni] = make_float4(0.f, 0.f, 0.f, 0.f);
  }

  float4 ra[4], rb[8];
  auto fetch = [&](int row0, int kslot) {
    const float* ba = p.A + (size_t)row0 * p.lda + kslot * kBK;
    const float* bb = p.B + kslot * kBK;
#pragma unroll
    for (int q = 0; q < 4; ++q) ra[q] = *reinterpret_cast<const float4*>(ba + (size_t)(32 * q) * p.lda + a_lane);
#pragma unroll
    for (int q = 0; q < 8; ++q) rb[q] = *reinterpret_cast<const float4*>(bb + b_lane[q]);
  };
  auto stash_all = [&](int kslot) {
    const int k = kslot * kBK + ck;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(sA + 32 * q * kLd + st_lane) = kmask(ra[q], k, p.K, true);
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(sB + 32 * q * kLd + st_lane) = kmask(rb[q], k, p.K, b_keep[q]);
  };

  {
    const int r0 = row_of(0);
    fetch(r0, slot(r0, 0));
    stash_all(slot(r0, 0));
  }
  __syncthreads();
  for (int j = 0; j < my_count; ++j) {
    const int row0 = row_of(j), row1 = row_of(j + 1);
    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    for (int kt = 0; kt < nk; ++kt) {
      const int rn = kt + 1 < nk ? row0 : row1, kn = kt + 1 < nk ? kt + 1 : 0;   // the stream continues into the next tile
      fetch(rn, slot(rn, kn));
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 a4[2], b4[4];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a4[mi] = *reinterpret_cast<const float4*>(pa + mi * 32 * kLd + 4 * g);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) b4[ni] = *reinterpret_cast<const float4*>(pb + ni * 32 * kLd + 4 * g);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
              const float av = e == 0 ? a4[mi].x : e == 1 ? a4[mi].y : e == 2 ? a4[mi].z : a4[mi].w;
              const float bv = e == 0 ? b4[ni].x : e == 1 ? b4[ni].y : e == 2 ? b4[ni].z : b4[ni].w;
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[mi][ni], 0, 0, 0);
            }
      }
      __syncthreads();
      stash_all(slot(rn, kn));
      __syncthreads();
    }
    // Epilogue through a per-wave LDS transpose.  The MFMA layout gives a lane one column and 16 scattered rows of a 32x32 tile:
    // stored as is, that is 16 four-byte store instructions per tile and output (the epilogue was store-issue-bound: 37 us per
    // output matrix).  Each tile goes to a [32][36] scratch image instead (conflict-free both ways) and comes back as 4 float4
    // per lane along the rows: 16-byte stores, 16-byte loads of the cos factors, a quarter of the memory instructions.
    // For the cos factors: the loads of a column-tile pair are all issued before its stores (in-order load/store counter).
    float* scr = sScr + wave * (32 * kLd);
    const int t_row = lane >> 3, t_col = (lane & 7) * 4;              // float4 position in the transposed image: rows t_row + 8 pass
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const size_t tile_row = (size_t)(row0 + wm * 64 + mi * 32 + t_row) * p.ldo;
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        float4 cv[2][4];
        if (EPI == EPI_MULC) {
#pragma unroll
          for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
              cv[n2][ps] = *reinterpret_cast<const float4*>(p.cmul + tile_row + (size_t)(8 * ps) * p.ldo + wn * 128 + (nh * 2 + n2) * 32 + t_col);
        }
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2) {
          const int ni = nh * 2 + n2;
          const size_t o0 = tile_row + wn * 128 + ni * 32 + t_col;
          float second[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[mi][ni][r];
            if (EPI == EPI_SINCOS) {
              sincos_cw(v + bn[ni], v, second[r]);
            } else if (EPI == EPI_BIAS) {
              v += bn[ni];
            }
            scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * kLd + li] = v;
          }
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            float4 v = *reinterpret_cast<const float4*>(scr + (t_row + 8 * ps) * kLd + t_col);
            if (EPI == EPI_MULC) {
              v.x *= cv[n2][ps].x; v.y *= cv[n2][ps].y; v.z *= cv[n2][ps].z; v.w *= cv[n2][ps].w;
              csum4[ni].x += v.x; csum4[ni].y += v.y; csum4[ni].z += v.z; csum4[ni].w += v.w;
            }
            store4_upto(p.out0 + o0 + (size_t)(8 * ps) * p.ldo, v, p.N - (wn * 128 + ni * 32 + t_col), p.N >= 256 || p.tail != nullptr);
          }
          if (EPI == EPI_SINCOS) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * kLd + li] = second[r];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps)
              store4_upto(p.out1 + o0 + (size_t)(8 * ps) * p.ldo, *reinterpret_cast<const float4*>(scr + (t_row + 8 * ps) * kLd + t_col),
                          p.N - (wn * 128 + ni * 32 + t_col), p.N >= 256 || p.tail != nullptr);
          }
        }
      }
    }
  }
  if (EPI == EPI_MULC && p.colsum != nullptr) {
    // a lane holds 4 columns x (its 8 row slots of every tile): 8 lanes per column group x 2 row-waves -> LDS, fixed-order sum
    float* red = sB;   // [16][256]: slot = wm*8 + t_row
    __syncthreads();
    const int t_row = lane >> 3, t_col = (lane & 7) * 4;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      *reinterpret_cast<float4*>(red + (wm * 8 + t_row) * 256 + wn * 128 + ni * 32 + t_col) = csum4[ni];
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) t += red[sl * 256 + tid];
    p.colsum[(size_t)blockIdx.x * 256 + tid] = t;
  }
}

// dW[n][k] partial over a slab of rows: grid (slabs, 2 halves of n).  G [M, ldg] (columns n), X [M, ldx] (columns k).
constexpr int kWM = 32;   // rows (reduction) per LDS tile
__global__ __launch_bounds__(256, 2) void mlp_wgrad_tn(const float* __restrict__ G, int ldg, const float* __restrict__ X, int ldx,
                                                       float* __restrict__ partial, long M, long rows_per_slab, int K) {
  __shared__ float sG[kWM * 128];
  __shared__ float sX[kWM * 256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * 128;
  const long m_begin = (long)blockIdx.x * rows_per_slab;
  const long m_end = (m_begin + rows_per_slab < M) ? m_begin + rows_per_slab : M;

  float4 rg[4], rx[8];
  // G tile: 32 rows x 128 cols = 1024 float4, 32 per row; X tile: 32 x 256 = 2048 float4, 64 per row
  auto gload = [&](long m0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = tid + 256 * q;
      const long m = m0 + (c >> 5);
      const int n = n0 + (c & 31) * 4;          // columns at or past ldg do not exist; those past N are computed and dropped
      rg[q] = (m < m_end && n < ldg) ? *reinterpret_cast<const float4*>(G + m * ldg + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int c = tid + 256 * q;
      const long m = m0 + (c >> 6);
      const int k = (c & 63) * 4;
      rx[q] = (m < m_end && k < K) ? *reinterpret_cast<const float4*>(X + m * ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(sG + (tid + 256 * q) * 4) = rg[q];
#pragma unroll
    for (int q = 0; q < 8; ++q) *reinterpret_cast<float4*>(sX + (tid + 256 * q) * 4) = rx[q];
  };

  f32x16 acc[2][4];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  if (m_begin < m_end) {
    gload(m_begin);
    sstore();
    __syncthreads();
    const float* pa = sG + lh * 128 + wm * 64 + li;
    const float* pb = sX + lh * 256 + wn * 128 + li;
    for (long m0 = m_begin; m0 < m_end; m0 += kWM) {
      const bool more = m0 + kWM < m_end;
      if (more) gload(m0 + kWM);
#pragma unroll
      for (int kk = 0; kk < kWM / 2; ++kk) {
        float a[2], b[4];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) a[mi] = pa[2 * kk * 128 + mi * 32];
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) b[ni] = pb[2 * kk * 256 + ni * 32];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
      }
      __syncthreads();
      if (more) {
        sstore();
        __syncthreads();
      }
    }
  }
  float* out = partial + (long)blockIdx.x * 256 * 256;
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wm * 64 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int k = wn * 128 + ni * 32 + li;
        out[n * 256 + k] = acc[mi][ni][r];
      }
}

// The folds of the per-workgroup partial sums the gradient kernels leave behind, as device functions over a workgroup of 1024 threads and 1024
// floats of LDS: launched one by one behind their producers (the entry points without a `defer` record), or all of an iteration's together in
// mlp_reduce_jobs_kernel (matpbr_mlp_reduce_jobs: nothing but the optimiser reads a weight or bias gradient) -- the same sums in the same order.
__device__ __forceinline__ void wgrad_reduce_body(float* red, const float* __restrict__ partial, int slabs, float* __restrict__ dW, int N, int K, int ldw,
                                                  int block) {
  // 256 outputs per workgroup x 4 slices of the slabs (a chain of slabs / 16 dependent rounds instead of slabs / 4), LDS fold
  const int t = threadIdx.x & 255, sl = threadIdx.x >> 8;
  const int idx = block * 256 + t;   // over 256 x 256
  const int per = (slabs + 3) / 4, c0 = sl * per, c1 = c0 + per < slabs ? c0 + per : slabs;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c = c0;
  for (; c + 3 < c1; c += 4) {
    s0 += partial[(long)(c + 0) * 65536 + idx];
    s1 += partial[(long)(c + 1) * 65536 + idx];
    s2 += partial[(long)(c + 2) * 65536 + idx];
    s3 += partial[(long)(c + 3) * 65536 + idx];
  }
  for (; c < c1; ++c) s0 += partial[(long)c * 65536 + idx];
  red[sl * 256 + t] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  const int n = idx >> 8, k = idx & 255;
  if (sl == 0 && n < N && k < K) dW[(long)n * ldw + k] = (red[t] + red[256 + t]) + (red[512 + t] + red[768 + t]);
}
__global__ __launch_bounds__(1024) void mlp_wgrad_reduce(const float* __restrict__ partial, int slabs, float* __restrict__ dW, int N, int K,
                                                         int ldw) {
  __shared__ float red[1024];
  wgrad_reduce_body(red, partial, slabs, dW, N, K, ldw, (int)blockIdx.x);
}

// column sums of the per-workgroup partials [groups, 256] -> out[n]: one workgroup per column (its first 256 threads), fixed-order tree
__device__ __forceinline__ void colsum_reduce_body(float* red, const float* __restrict__ part, int groups, float* __restrict__ out, int col) {
  const bool act = threadIdx.x < 256;
  float s = 0.f;
  if (act) {
    for (int g = threadIdx.x; g < groups; g += 256) s += part[(long)g * 256 + col];
    red[threadIdx.x] = s;
  }
  __syncthreads();
#pragma unroll
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[col] = red[0];
}
__global__ __launch_bounds__(256) void mlp_colsum_reduce(const float* __restrict__ part, int groups, float* __restrict__ out) {
  __shared__ float red[256];
  colsum_reduce_body(red, part, groups, out, (int)blockIdx.x);
}



// ---------------------------------------------------------------------------------------------------------------------------
// Small point sets (M <= kSmallM rows): the 16x32 envmap MLP of hot loop A (mymodels/mlps.py PosMLP(output_type='envmap'), 512
// points; inverse_img_w_mi.py:117-124,238-239).  A 128-row persistent tile would occupy 4-8 CUs; here every wave owns one 32x32
// output tile over the whole reduction and feeds the MFMA straight from L2 (the operands of all layers together are < 3 MB):
// 128 waves for a [512,256]x[256,256] product, ~130 MFMAs each.  Same entry points, same epilogues, same results contract.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr long kSmallM = 1024;

__device__ __forceinline__ float4 ld4_masked(const float* p, int k, int K) {   // 4 consecutive k, zero beyond K (address clamped by the caller)
  float4 v = *reinterpret_cast<const float4*>(p);
  v.x = k < K ? v.x : 0.f;
  v.y = k + 1 < K ? v.y : 0.f;
  v.z = k + 2 < K ? v.z : 0.f;
  v.w = k + 3 < K ? v.w : 0.f;
  return v;
}

// C[m][n] = sum_k A[m][k] B[n][k] with the epilogues of mlp_gemm_nt: one workgroup per 32x32 tile of C.  Its four waves split the
// reduction (k-quarters of <= 64), so every operand load of the tile is in flight at once (the data sits in another XCD's L2 or in
// the Infinity Cache: ~2 us away) and 128 workgroups cover a [512,256] product; the partial tiles are folded through LDS in fixed
// order and each wave finishes a quarter of the rows.  Lane l: row / column l & 31; lane half h = l >> 5 takes k = 8 j + 4 h .. + 3
// of every 8-wide chunk j (A and B agree on the pairing, so the sum over k is unchanged).
// EPI_MULC writes per-row-tile column sums to colsum[tile_m][256].
// BKN: B is given as [K][ldb] (k-major: the forward weight of the layer, used as is by the backward product G W) instead of [N][ldb].
template <int EPI, bool BKN = false>
__device__ __forceinline__ void small_nt_body(const NtArgs& p, int block) {
  __shared__ float s_part[4][16 * 64];
  __shared__ float s_col[4][32];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int tiles_n = (p.N + 31) >> 5;
  const int tm = block / tiles_n, tn = block - tm * tiles_n;
  const int row = min(tm * 32 + li, p.M - 1), col = min(tn * 32 + li, p.N - 1);
  const float* pa = p.A + (size_t)row * p.lda;
  const float* pb = BKN ? p.B + col : p.B + (size_t)col * p.ldb;
  const int kpad = (p.K + 3) & ~3;
  const int kq = (((p.K + 3) / 4) + 7) & ~7;      // k per wave, a multiple of 8 (<= 64)
  const int k_begin = wave * kq;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  {
    float4 a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k_begin + 8 * j + 4 * lh;
      const int kc = k < kpad ? k : kpad - 4;     // clamped address, masked value
      const bool mine = 8 * j < kq;
      a[j] = ld4_masked(pa + kc, k, mine ? p.K : 0);
      if (BKN) {                                  // four rows of the k-major matrix: lanes run along n (coalesced)
        const int K_ = mine ? p.K : 0;
        b[j].x = k < K_ ? pb[(size_t)min(k, p.K - 1) * p.ldb] : 0.f;
        b[j].y = k + 1 < K_ ? pb[(size_t)min(k + 1, p.K - 1) * p.ldb] : 0.f;
        b[j].z = k + 2 < K_ ? pb[(size_t)min(k + 2, p.K - 1) * p.ldb] : 0.f;
        b[j].w = k + 3 < K_ ? pb[(size_t)min(k + 3, p.K - 1) * p.ldb] : 0.f;
      } else {
        b[j] = ld4_masked(pb + kc, k, mine ? p.K : 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (8 * j < kq && k_begin + 8 * j < p.K) {  // wave-uniform
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].x, b[j].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].y, b[j].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].z, b[j].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j].w, b[j].w, acc, 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) s_part[wave][r * 64 + lane] = acc[r];
  __syncthreads();
  const int n = tn * 32 + li;
  const bool ncol = n < p.N;
  const float bias = (EPI != EPI_MULC && ncol) ? p.bias[n] : 0.f;
  float csum = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {                   // wave w finishes accumulator rows 4 w .. 4 w + 3
    const int r = 4 * wave + q;
    const float v0 = ((s_part[0][r * 64 + lane] + s_part[1][r * 64 + lane]) + s_part[2][r * 64 + lane]) + s_part[3][r * 64 + lane];
    const int m = tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (m < p.M && ncol) {
      const size_t o = (size_t)m * p.ldo + n;
      if (EPI == EPI_SINCOS) {
        float sv, cv;
        sincos_cw(v0 + bias, sv, cv);
        p.out0[o] = sv;
        p.out1[o] = cv;
      } else if (EPI == EPI_BIAS) {
        p.out0[o] = v0 + bias;
      } else {
        const float v = v0 * p.cmul[o];
        p.out0[o] = v;
        csum += v;
      }
    }
  }
  if (EPI == EPI_MULC && p.colsum != nullptr) {
    csum += __shfl_xor(csum, 32);                 // the two lane halves hold different rows of the same column
    if (lh == 0) s_col[wave][li] = csum;
    __syncthreads();
    if (wave == 0 && lh == 0 && ncol) p.colsum[(size_t)tm * 256 + n] = (s_col[0][li] + s_col[1][li]) + (s_col[2][li] + s_col[3][li]);
  }
}

template <int EPI, bool BKN = false>
__global__ __launch_bounds__(256) void mlp_small_nt(const NtArgs p) { small_nt_body<EPI, BKN>(p, (int)blockIdx.x); }

// dW[n][k] = sum_m G[m][n] X[m][k]: one workgroup per 32x32 tile of dW; its four waves split the reduction over m (<= kSmallM)
// into quarters of <= 256 rows, issue every load of a 128-row half at once (64 + 64 dwords per lane), and fold their partial
// tiles through LDS in fixed order.
__device__ __forceinline__ void small_tn_body(const float* __restrict__ G, int ldg, const float* __restrict__ X, int ldx, float* __restrict__ dW,
                                              int ldw, int M, int N, int K, int block) {
  __shared__ float s_part[3][16 * 64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  const int tiles_k = (K + 31) >> 5;
  const int tn = block / tiles_k, tk = block - tn * tiles_k;
  const int n = tn * 32 + li, k = tk * 32 + li;
  const bool nok = n < N, kok = k < K;
  const float* pg = G + (nok ? n : N - 1);
  const float* px = X + (kok ? k : K - 1);
  const int quarter = ((M + 3) / 4 + 1) & ~1;     // rows per wave, even
  const int m_begin = wave * quarter, m_end = min(M, m_begin + quarter);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int m0 = m_begin; m0 < m_end; m0 += 128) {
    float a[64], b[64];
#pragma unroll
    for (int j = 0; j < 64; ++j) {
      const int m = m0 + 2 * j + lh;
      const int mc = m < M ? m : M - 1;
      const float av = pg[(size_t)mc * ldg], bv = px[(size_t)mc * ldx];
      a[j] = (m < m_end && nok) ? av : 0.f;
      b[j] = (m < m_end && kok) ? bv : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 64; ++j)
      if (m0 + 2 * j < m_end) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
  }
  if (wave > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) s_part[wave - 1][r * 64 + lane] = acc[r];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = ((acc[r] + s_part[0][r * 64 + lane]) + s_part[1][r * 64 + lane]) + s_part[2][r * 64 + lane];
      const int nn = tn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (nn < N && kok) dW[(size_t)nn * ldw + k] = v;
    }
  }
}

__global__ __launch_bounds__(256) void mlp_small_tn(const float* __restrict__ G, int ldg, const float* __restrict__ X, int ldx,
                                                    float* __restrict__ dW, int ldw, int M, int N, int K) {
  small_tn_body(G, ldg, X, ldx, dW, ldw, M, N, K, (int)blockIdx.x);
}

// One backward step of a small-M network (the 16 x 32 envmap MLP of hot loop A) in ONE launch.  With g = dL/d pre of layer l in hand, three
// pieces of work are independent of each other: the weight gradient of layer l (g^T x), the input gradient into layer l - 1 ((g W) * cos,
// with per-row-tile column sums for that layer's bias gradient), and the fold of the column sums the step BEFORE left behind (= the bias
// gradient of layer l).  As separate launches they were 3 of the iteration's 26 launches per layer, each a few microseconds of work behind
// ~6 us of launch; here workgroups take one of the roles by index.
struct SmallBwdStep {
  NtArgs d;                 // input gradient (mlp_small_nt<EPI_MULC, BKN>); nD == 0: none
  int nD;
  const float* G; int ldg; const float* X; int ldx; float* dW; int ldw; int M, N, K;   // weight gradient (mlp_small_tn)
  int nW;
  const float* part_in; int part_stride, groups_in; float* d_bias; int n_bias;          // bias gradient: d_bias[c] = sum_g part_in[g stride + c]
};
__global__ __launch_bounds__(256) void mlp_small_bwd_step_kernel(const SmallBwdStep a) {
  int b = (int)blockIdx.x;
  if (b < a.nD) {
    small_nt_body<EPI_MULC, true>(a.d, b);
    return;
  }
  b -= a.nD;
  if (b < a.nW) {
    small_tn_body(a.G, a.ldg, a.X, a.ldx, a.dW, a.ldw, a.M, a.N, a.K, b);
    return;
  }
  if (a.n_bias <= 4 && a.groups_in > 64) {
    // the output layer: a handful of columns, one row of g per point (512 groups).  One thread per column walked the 512 rows in turn --
    // a chain of 512 dependent loads, 31 us, the longest kernel of the envmap-MLP iteration; here the rows are spread over the workgroup
    // (thread t: rows t, t + 256, ... in turn) and the 256 partial sums folded by a fixed tree: deterministic, a few microseconds
    __shared__ float s_b[256][4];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int g = (int)threadIdx.x; g < a.groups_in; g += 256) {
#pragma unroll
      for (int c = 0; c < 4; ++c)
        if (c < a.n_bias) v[c] += a.part_in[(long)g * a.part_stride + c];
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) s_b[threadIdx.x][c] = v[c];
    __syncthreads();
    for (int half = 128; half > 0; half >>= 1) {
      if ((int)threadIdx.x < half) {
#pragma unroll
        for (int c = 0; c < 4; ++c) s_b[threadIdx.x][c] += s_b[threadIdx.x + half][c];
      }
      __syncthreads();
    }
    if ((int)threadIdx.x < a.n_bias) a.d_bias[threadIdx.x] = s_b[0][threadIdx.x];
    return;
  }
  const int c = (b - a.nW) * 256 + (int)threadIdx.x;        // fixed order over the groups: deterministic
  if (c < a.n_bias) {
    float s = 0.f;
    int g = 0;
    for (; g + 8 <= a.groups_in; g += 8) {          // eight groups requested together, added in turn (the same sum, one round trip instead of eight)
      float t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) t[j] = a.part_in[(long)(g + j) * a.part_stride + c];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += t[j];
    }
    for (; g < a.groups_in; ++g) s += a.part_in[(long)g * a.part_stride + c];
    a.d_bias[c] = s;
  }
}

template <int EPI, bool BKN = false>
int launch_small_nt(NtArgs p, long M, hipStream_t stream) {
  p.M = (int)M;
  const int tiles = (int)((M + 31) / 32) * ((p.N + 31) / 32);
  hipLaunchKernelGGL((mlp_small_nt<EPI, BKN>), dim3((unsigned)tiles), dim3(256), 0, stream, p);
  return (int)((M + 31) / 32);   // column-sum groups
}


// ---------------------------------------------------------------------------------------------------------------------------
// Split-operand products on the bf16 matrix pipe ("bx" kernels).
// An f32 number is the exact sum of three bf16 numbers, x = x1 + x2 + x3 (round-to-nearest pieces of the running residual: 8 + 8 + 8
// significand bits), and a product of two bf16 numbers is exact in f32.  A x B = sum_{i,j} A_i B_j therefore runs on
// v_mfma_f32_32x32x16_bf16 with f32 accumulation: all 9 products reproduce the f32 product exactly before accumulation; the 6
// products with i + j <= 4 drop terms below 2^-24 |a||b| (the size of one f32 rounding).  The bf16 pipe is 16x the f32-input
// MFMA (32 cycles per 32x32x16 step against 8 x 64), so 6 (9) products cost 0.375 (0.56) of the f32 kernel's matrix time, and
// the layer becomes bound by its HBM traffic (read X, write sin and cos: 805 MB per 512x512 layer).
//
// No LDS in the k-loop, no barriers: every wave owns a 64 x 128 block of the output (2 x 4 accumulator tiles) and loads its
// operands already in MFMA layout -- lane (l & 31, l >> 5) reads 16 consecutive k of its A row per 32-wide super-step (a full
// 128-byte line per row) and splits them in registers; the weights are split once per call into `wsplit`, ordered so that a wave's
// read of one operand is 1 KB contiguous, and come from L2.  One workgroup (2 x 2 waves, 128 x 256 outputs) per CU, persistent.
// ---------------------------------------------------------------------------------------------------------------------------
// wsplit layout: [K/32 super-steps][2 steps][3 pieces][2 lane halves g][256 rows n] x uint4 (8 bf16: k = 32 ks + 16 g + 8 s + j).
// f16 form (MATPBR_WSPLIT_F16X2): [K/32][2 steps][2 pieces][2 g][256 n] x uint4 of 8 f16, the values scaled by 256 (16 KB per half step).
// One super-step (3072 uint4 = 48 KB) is also the LDS image of the weights for that super-step: a wave's read of one operand is
// two contiguous 512-byte runs (conflict-free), the global -> LDS copy is a straight copy.
constexpr int kBxStage = 2 * 3 * 2 * 256;   // uint4 per super-step
__host__ __device__ inline size_t wsplit_index(int ks, int s, int piece, int n, int g) {
  return ((((size_t)ks * 2 + s) * 3 + piece) * 2 + g) * 256 + n;
}
// One thread = the 8 consecutive k of (super-step ks, step s, row n, lane half g).  tr: element (n, k) of the operand is B[k * ldb + n]
// (the forward weight serving as the backward product's operand); f16: the two-piece f16 form
__device__ __forceinline__ void split_weights_item(const float* __restrict__ B, int ldb, int N, int K, bool tr, bool f16, uint4* __restrict__ out, int idx) {
  const int nks = (K + 31) / 32;
  if (idx >= nks * 2 * 256 * 2) return;
  const int g = idx & 1, n = (idx >> 1) & 255, s = (idx >> 9) & 1, ks = idx >> 10;
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const int k = 32 * ks + 16 * g + 8 * s + q;
    v[q] = (n < N && k < K) ? (tr ? B[(size_t)k * ldb + n] : B[(size_t)n * ldb + k]) : 0.f;
  }
  if (f16) {
    unsigned p[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split2h(v[2 * q] * kF16WScale, v[2 * q + 1] * kF16WScale, p[0][q], p[1][q]);
#pragma unroll
    for (int piece = 0; piece < 2; ++piece)
      out[((((size_t)ks * 2 + s) * 2 + piece) * 2 + g) * 256 + n] = make_uint4(p[piece][0], p[piece][1], p[piece][2], p[piece][3]);
    return;
  }
  unsigned p[3][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) split3(v[2 * q], v[2 * q + 1], p[0][q], p[1][q], p[2][q]);
#pragma unroll
  for (int piece = 0; piece < 3; ++piece) out[wsplit_index(ks, s, piece, n, g)] = make_uint4(p[piece][0], p[piece][1], p[piece][2], p[piece][3]);
}
__global__ __launch_bounds__(256) void mlp_split_weights_kernel(const float* __restrict__ B, int ldb, int N, int K, int flags, uint4* __restrict__ out) {
  split_weights_item(B, ldb, N, K, (flags & 1) != 0, (flags & 2) != 0, out, blockIdx.x * 256 + threadIdx.x);
}

// Several operands in one launch (the weights of every layer change together, once per optimiser step): blockIdx.y = job
struct SplitJobs {
  const float* w[8];
  uint4* out[8];
  int ldw[8], N[8], K[8], transposed[8];   // transposed: bit 0 = transposed operand, bit 1 = the f16 form
};
__global__ __launch_bounds__(256) void mlp_split_weights_multi_kernel(const SplitJobs jobs) {
  const int j = blockIdx.y;
  split_weights_item(jobs.w[j], jobs.ldw[j], jobs.N[j], jobs.K[j], (jobs.transposed[j] & 1) != 0, (jobs.transposed[j] & 2) != 0, jobs.out[j],
                     blockIdx.x * 256 + threadIdx.x);
}

// Eight waves per workgroup (4 row groups of 32 x 2 column halves of 128; 128 x 256 outputs per workgroup, one workgroup per
// CU): two waves per SIMD, so that one wave's operand split / LDS traffic / epilogue runs under the other's products.
constexpr int kBxThreads = 512;
// HEAD (forward of the LAST sine layer, N == 256): the epilogue also forms the five outputs of the network's output layer for its
// 128 rows -- each lane dots the 16-byte words of sines it is about to store with the matching weights (from a 5 KB LDS image), an
// 8-lane DPP fold and one LDS exchange between the two column halves complete the rows -- and runs the 'arm' head on them: the
// separate pass over the 268 MB of sines (76 us at 512 x 512) disappears.
// W0 (EPI_MULC, the input gradient INTO THE FIRST LAYER): G' is not stored -- its only consumers are the first layer's bias gradient (the
// column sums, formed here anyway) and weight gradient dW0[n][k] = sum_m G'[m][n] x0[m][k], k < 16, which the epilogue accumulates
// itself: each 32 x 32 block of G' sits in the wave's LDS slice for the transposition, the wave's 32 rows of x0 beside it, and sixteen
// v_mfma_f32_16x16x4f32 per block fold them into 8 x 4 accumulator registers per wave.  Saves the 268 MB store, the 268 MB read of the
// skinny weight-gradient pass and its launch.
typedef float f32x4v __attribute__((ext_vector_type(4)));
// GL: both operands of the main loop arrive by LDS-DMA (global_load_lds_dwordx4: no register round trip, no ds_write, full 128-byte
// lines of the rows, the rows read once per workgroup instead of once per column half).  Weights: two 48 KB buffers, one super-step
// ahead; rows: a ring of three 16 KB buffers [128 rows][8 x 16 B] (chunk c of row r at slot c ^ (r >> 1 & 7): the fragment reads of
// 16 consecutive rows cover 16 distinct 16-byte bank groups), two super-steps ahead.  The loads are inline asm, invisible to hipcc's
// wait bookkeeping: one counted s_waitcnt vmcnt(4) + s_barrier per super-step retires everything but the four youngest pieces (the
// rows of step g + 2) and publishes it.  Waves 0-3 issue all sixteen pieces of a step; waves 4-7 (the other wave of each SIMD) go
// straight to the products.  Slice w of a weight buffer (bytes [6144 w, 6144 w + 6144)) is wave w's transposition scratch during the
// epilogue, in the buffer that is filled next (one more barrier per tile separates the two uses).
constexpr int kGlRows = 128 * 32 * 4;                                    // bytes of one row buffer
constexpr size_t kGlSmem = 2 * kBxStage * sizeof(uint4) + 3 * kGlRows;   // 144 KB
template <int EPI, int NPROD, bool FULL, bool HEAD = false, bool W0 = false, bool GL = false>   // FULL: all 256 output columns exist (N == 256): unguarded 16-byte stores
__global__ __launch_bounds__(kBxThreads, 1) void mlp_nt_bx(const NtArgs p, const uint4* __restrict__ wsplit, const HeadArgs hd) {
  extern __shared__ __align__(16) unsigned char bx_smem[];
  uint4* sB = reinterpret_cast<uint4*>(bx_smem);                         // [2 buffers][kBxStage]
  float* sScr = reinterpret_cast<float*>(bx_smem + 2 * kBxStage * sizeof(uint4));   // [8 waves][32][kLd]  (GL: the row ring instead)
  float* sRed = reinterpret_cast<float*>(bx_smem);                       // [32][256] after the last tile (aliases sB)
  float* sExtra = GL ? reinterpret_cast<float*>(bx_smem + kGlSmem) : sScr + 8 * 32 * kLd;
  float* sW4 = sExtra;                                                   // HEAD: [5][256] output-layer weights
  float* sComb = sW4 + 5 * 256;                                          // HEAD: [128 rows][2 column halves][8]
  float* sX0 = sExtra;                                  // W0: [8 waves][32 rows][16]  (HEAD and W0 never meet)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int nks = (p.K + 31) / 32;
  const int tiles = p.M / kBM;
  constexpr int kCopy = kBxStage / kBxThreads;                           // uint4 per thread and super-step of the weight stream
  float bn[4];
  float4 csum4[4];
  f32x4v acc0[4][2];                                                     // W0: dW0 of this wave's rows: [column block][16-column half], a 16 x 16 tile each
  if (W0) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc0[ni][ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    const int n = wn * 128 + ni * 32 + li;
    bn[ni] = (EPI != EPI_MULC) ? p.bias[n < p.N ? n : p.N - 1] : 0.f;
    csum4[ni] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int b_lane = lh * 256 + wn * 128 + li;                           // + ((s * 3 + piece) * 2) * 256 + ni * 32
  // the weights of super-step 0 are resident in buffer 0 at the start of every tile: the stream of super-steps runs across tiles
  uint4 bnext[kCopy];
  if (!GL) {
#pragma unroll
    for (int q = 0; q < kCopy; ++q) sB[tid + kBxThreads * q] = wsplit[tid + kBxThreads * q];
  }
  if (HEAD) {
    for (int i = tid; i < 5 * 256; i += kBxThreads) sW4[i] = hd.w[(size_t)(i >> 8) * hd.ldw + (i & 255)];
  }
  __syncthreads();
  int buf = 0;
  // ---- GL: lane constants of the two streams
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds0 = GL ? lds_byte_address(bx_smem) : 0u;
  // waves 0-3 issue every piece (their own slices and those of waves 4-7, their partners on the SIMDs): the other four start a
  // super-step with the fragment reads and the products, so the matrix pipes run while the loads are being issued
  const bool issuer = wave_u >= 4;
  const unsigned gl_w_dst = lds0, gl_a_dst = lds0 + 2u * kBxStage * 16u;
  unsigned gl_w_voff[2], gl_a_voff[2][2], gl_rd[4];
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int vw = (wave & 3) + 4 * v;                                   // the slice owner this lane copies for
    gl_w_voff[v] = (unsigned)(384 * vw + lane) * 16u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int row = 16 * vw + 8 * j + (lane >> 3);
      gl_a_voff[v][j] = (unsigned)(row * p.lda + 4 * ((lane & 7) ^ ((row >> 1) & 7))) * 4u;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = wm * 32 + li;
    gl_rd[q] = (unsigned)(2 * kBxStage * 16 + row * 128 + (((4 * lh + q) ^ ((row >> 1) & 7)) * 16));
  }
  int a_slot = 0;                                                        // ring slot of the current super-step's rows
  auto gl_issue_w = [&](int ks_w, int wb) {
    const char* src = reinterpret_cast<const char*>(wsplit) + (size_t)ks_w * (kBxStage * 16);
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int j = 0; j < 6; ++j)
        glds16(src + 1024 * j, gl_w_voff[v], gl_w_dst + (unsigned)wb * (kBxStage * 16u) + (unsigned)((wave_u & 3) + 4 * v) * 6144u + 1024u * j);
  };
  auto gl_issue_a = [&](int tile_a, int ks_a, int slot) {
    const float* src = p.A + (size_t)tile_a * kBM * p.lda + 32 * ks_a;
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        glds16(src, gl_a_voff[v][j], gl_a_dst + (unsigned)slot * (unsigned)kGlRows + (unsigned)((wave_u & 3) + 4 * v) * 2048u + 1024u * j);
  };
  if (GL) {
    const int t0 = blockIdx.x;
    if (issuer) {
      gl_issue_w(0, 0);
      gl_issue_a(t0, 0, 0);
      gl_issue_a(nks > 1 ? t0 : (t0 + (int)gridDim.x < tiles ? t0 + (int)gridDim.x : t0), nks > 1 ? 1 : 0, 1);
      asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_barrier" ::: "memory");
  }

  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int row0 = tile * kBM;
    const float* pa0 = p.A + (size_t)(row0 + wm * 32 + li) * p.lda + 16 * lh;
    f32x16 acc[4];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

    float4 raw[4];
    if (!GL) {
#pragma unroll
      for (int q = 0; q < 4; ++q) raw[q] = *reinterpret_cast<const float4*>(pa0 + 4 * q);
    }
    for (int ks = 0; ks < nks; ++ks) {
      const int ksn = ks + 1 < nks ? ks + 1 : 0;                           // next super-step of the stream (wraps into the next tile)
      float4 cur[4];
      const int ks_stamp = ks & 7; (void)ks_stamp;
      if (GL) {
        // the weights of the next super-step and the rows of the one after it (beyond this workgroup's last tile: its rows again, unused)
        int t2 = tile, k2 = ks + 2;
        if (k2 >= nks) { k2 -= nks; t2 = tile + (int)gridDim.x < tiles ? tile + (int)gridDim.x : tile; }
        if (k2 >= nks) k2 = 0;                                             // a one-step reduction
        if (issuer) {
          gl_issue_w(ksn, buf ^ 1);
          gl_issue_a(t2, k2, a_slot >= 1 ? a_slot - 1 : 2);                // (a_slot + 2) % 3
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) cur[q] = *reinterpret_cast<const float4*>(bx_smem + gl_rd[q] + a_slot * kGlRows);
      } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) cur[q] = raw[q];
      // in flight during this super-step's products: the next weights (global -> registers) and the next 16 k of the rows
      const uint4* wsrc = wsplit + (size_t)ksn * kBxStage + tid;
#pragma unroll
      for (int q = 0; q < kCopy; ++q) bnext[q] = wsrc[kBxThreads * q];
      if (ks + 1 < nks) {
#pragma unroll
        for (int q = 0; q < 4; ++q) raw[q] = *reinterpret_cast<const float4*>(pa0 + 32 * (ks + 1) + 4 * q);
      }
      }
      if (32 * (ks + 1) > p.K) {                            // ragged reduction (K = 241): columns at and beyond K are scratch, not zeros
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = 32 * ks + 16 * lh + 4 * q;
          cur[q].x = k < p.K ? cur[q].x : 0.f;
          cur[q].y = k + 1 < p.K ? cur[q].y : 0.f;
          cur[q].z = k + 2 < p.K ? cur[q].z : 0.f;
          cur[q].w = k + 3 < p.K ? cur[q].w : 0.f;
        }
      }
      const uint4* sb = sB + buf * kBxStage + b_lane;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint4 bq[3][4];
#pragma unroll
        for (int piece = 0; piece < 3; ++piece)
#pragma unroll
          for (int ni = 0; ni < 4; ++ni) bq[piece][ni] = sb[((s * 3 + piece) * 2) * 256 + ni * 32];
        uint4 aq[3];
        {
          const float4 u = cur[2 * s], v = cur[2 * s + 1];
          split3(u.x, u.y, aq[0].x, aq[1].x, aq[2].x);
          split3(u.z, u.w, aq[0].y, aq[1].y, aq[2].y);
          split3(v.x, v.y, aq[0].z, aq[1].z, aq[2].z);
          split3(v.z, v.w, aq[0].w, aq[1].w, aq[2].w);
        }
        // products from the smallest terms up: (a3 b3, a2 b3, a3 b2 with NPROD = 9), a3 b1, a1 b3, a2 b2, a2 b1, a1 b2, a1 b1
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          constexpr int ia[9] = {2, 1, 2, 2, 0, 1, 1, 0, 0}, ib[9] = {2, 2, 1, 0, 2, 1, 0, 1, 0};
          if (NPROD == 6 && t < 3) continue;
#pragma unroll
          for (int ni = 0; ni < 4; ++ni)
            acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aq[ia[t]]), __builtin_bit_cast(bf16x8, bq[ib[t]][ni]),
                                                              acc[ni], 0, 0, 0);
        }
      }
      if (GL) {
        // everything but the four youngest pieces (the rows of step + 2) has landed; the barrier publishes it
        // (lgkmcnt: this wave's fragment reads of the buffers that the next step's DMA overwrites are complete, not merely issued)
        if (issuer) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        a_slot = a_slot == 2 ? 0 : a_slot + 1;
      } else {
      // the other buffer was last read in the previous super-step, and every wave has passed that step's barrier
      uint4* sdst = sB + (buf ^ 1) * kBxStage + tid;
#pragma unroll
      for (int q = 0; q < kCopy; ++q) sdst[kBxThreads * q] = bnext[q];
      __syncthreads();
      }
      buf ^= 1;
    }
    // epilogue through a per-wave LDS transpose (16-byte stores / cos loads), as mlp_gemm_nt_wide
    // (GL: in the slice of the weight buffer read last that this wave fills next)
    float* scr = GL ? reinterpret_cast<float*>(bx_smem + (size_t)(buf ^ 1) * (kBxStage * 16) + wave * 6144) : sScr + wave * (32 * kLd);
    const int t_row = lane >> 3, t_col = (lane & 7) * 4;
    const size_t tile_row = (size_t)(row0 + wm * 32 + t_row) * p.ldo;
    float* sx = sX0 + wave * (32 * 16);
    if (W0) {                                                 // this wave's 32 rows of x0: lane = (row, half)
      const float* xs = p.x0 + (size_t)(row0 + wm * 32 + (lane >> 1)) * p.ldx0 + 8 * (lane & 1);
      const float4 xa = *reinterpret_cast<const float4*>(xs), xb = *reinterpret_cast<const float4*>(xs + 4);
      *reinterpret_cast<float4*>(sx + (lane >> 1) * 16 + 8 * (lane & 1)) = xa;
      *reinterpret_cast<float4*>(sx + (lane >> 1) * 16 + 8 * (lane & 1) + 4) = xb;
    }
    float hacc[4][5];
    if (HEAD) {
#pragma unroll
      for (int ps = 0; ps < 4; ++ps)
#pragma unroll
        for (int j = 0; j < 5; ++j) hacc[ps][j] = 0.f;
    }
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      float4 cv[2][4];
      if (EPI == EPI_MULC) {
#pragma unroll
        for (int n2 = 0; n2 < 2; ++n2)
#pragma unroll
          for (int ps = 0; ps < 4; ++ps)
            cv[n2][ps] = *reinterpret_cast<const float4*>(p.cmul + tile_row + (size_t)(8 * ps) * p.ldo + wn * 128 + (nh * 2 + n2) * 32 + t_col);
      }
#pragma unroll
      for (int n2 = 0; n2 < 2; ++n2) {
        const int ni = nh * 2 + n2;
        const size_t o0 = tile_row + wn * 128 + ni * 32 + t_col;
        float second[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[ni][r];
          if (EPI == EPI_SINCOS) {
            if (p.out1 == nullptr) v = sin_packed(v + bn[ni]);          // uniform: the sines carry the sign of their cosine
            else sincos_cw(v + bn[ni], v, second[r]);
          } else if (EPI == EPI_BIAS) {
            v += bn[ni];
          }
          scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * kLd + li] = v;
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          float4 v = *reinterpret_cast<const float4*>(scr + (t_row + 8 * ps) * kLd + t_col);
          if (EPI == EPI_MULC) {
            if (p.cmul_sin) cv[n2][ps] = cos_from_packed_sin(cv[n2][ps]);
            v.x *= cv[n2][ps].x; v.y *= cv[n2][ps].y; v.z *= cv[n2][ps].z; v.w *= cv[n2][ps].w;
            csum4[ni].x += v.x; csum4[ni].y += v.y; csum4[ni].z += v.z; csum4[ni].w += v.w;
          }
          if (W0) *reinterpret_cast<float4*>(scr + (t_row + 8 * ps) * kLd + t_col) = v;   // G' back into the block, for the product below
          if (HEAD) {
#pragma unroll
            for (int j = 0; j < 5; ++j) {
              const float4 w4 = *reinterpret_cast<const float4*>(sW4 + j * 256 + wn * 128 + ni * 32 + t_col);
              hacc[ps][j] = __builtin_fmaf(v.x, w4.x, __builtin_fmaf(v.y, w4.y, __builtin_fmaf(v.z, w4.z, __builtin_fmaf(v.w, w4.w, hacc[ps][j]))));
            }
          }
          if (!W0) store4_upto(p.out0 + o0 + (size_t)(8 * ps) * p.ldo, v, p.N - (wn * 128 + ni * 32 + t_col), FULL || wn * 128 + ni * 32 + 32 <= p.N);
        }
        if (W0) {     // dW0[16 ct + i][j] += sum over the block's 32 rows of G'[row][16 ct + i] x0[row][j]  (A: lane = (i, k), B: lane = (j, k))
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
              acc0[ni][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(scr[(4 * kk + (lane >> 4)) * kLd + 16 * ct + (lane & 15)],
                                                                  sx[(4 * kk + (lane >> 4)) * 16 + (lane & 15)], acc0[ni][ct], 0, 0, 0);
        }
        if (EPI == EPI_SINCOS && p.out1 != nullptr) {
#pragma unroll
          for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * kLd + li] = second[r];
#pragma unroll
          for (int ps = 0; ps < 4; ++ps)
            store4_upto(p.out1 + o0 + (size_t)(8 * ps) * p.ldo, *reinterpret_cast<const float4*>(scr + (t_row + 8 * ps) * kLd + t_col),
                        p.N - (wn * 128 + ni * 32 + t_col), FULL || wn * 128 + ni * 32 + 32 <= p.N);
        }
      }
    }
    if (HEAD) {
      // the 8 lanes of a row (lane & 7) hold its partial dot products over this wave's 128 columns: fold (total in lane 8 g + 7)
#pragma unroll
      for (int ps = 0; ps < 4; ++ps)
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          float v = hacc[ps][j];
          v = dpp_add_f<0x111>(v);    // row_shr:1
          v = dpp_add_f<0x112>(v);    // row_shr:2
          v = dpp_add_f<0x114>(v);    // row_shr:4
          if ((lane & 7) == 7) sComb[((wm * 32 + t_row + 8 * ps) * 2 + wn) * 8 + j] = v;
        }
      __syncthreads();
      if (tid < kBM) {                                        // one thread per row of the tile: both column halves + bias, then the head
        float v5[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) v5[j] = (sComb[(tid * 2) * 8 + j] + sComb[(tid * 2 + 1) * 8 + j]) + hd.bias[j];
        arm_head_store(hd.h, (long)row0 + tid, v5);
      }
      // sComb is written again at the end of the next tile, eight barriers from here
    }
    if (GL) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the scratch slices are the next LDS-DMA targets (of waves 0-3)
  }
  if (GL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the unused look-ahead pieces: nothing may land after the workgroup ends
  if (W0) {                                                   // lane holds dW0[n = .. + 4 (lane >> 4) + r][k = lane & 15]
    const size_t slab = (size_t)blockIdx.x * 4 + wm;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          p.w0_part[(slab * 16 + (lane & 15)) * 256 + wn * 128 + ni * 32 + 16 * ct + 4 * (lane >> 4) + r] = acc0[ni][ct][r];
  }
  if (EPI == EPI_MULC && p.colsum != nullptr) {
    const int t_row = lane >> 3, t_col = (lane & 7) * 4;
    __syncthreads();                                          // sRed aliases the weight buffers: every wave is done with them
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      *reinterpret_cast<float4*>(sRed + (wm * 8 + t_row) * 256 + wn * 128 + ni * 32 + t_col) = csum4[ni];
    __syncthreads();
    if (tid < 256) {
      float t = 0.f;
#pragma unroll
      for (int sl = 0; sl < 32; ++sl) t += sRed[sl * 256 + tid];
      p.colsum[(size_t)blockIdx.x * 256 + tid] = t;
    }
  }
}

// mlp_nt_gx: the same product as mlp_nt_bx (same pieces, same order of accumulation: the same bits) as TWO independent 256-thread
// workgroups per CU.  A workgroup owns a 128 x 256 tile with 2 x 2 waves of 64 x 128 (128 accumulator registers per wave): every B
// fragment read from LDS feeds two products instead of one, and the two workgroups of a CU are not coupled by a barrier -- one's
// epilogue and LDS round trips run under the other's products.  80 KB of LDS per workgroup: weights in two 24 KB buffers per HALF
// super-step (16 k), one step ahead; rows in two 16 KB granules of 32 k (full 128-byte lines), one granule ahead; all by LDS-DMA
// (inline asm, counted by hand: vmcnt(4) after the first half of a granule, vmcnt(0) after the second).  Wave w fills slice w of every
// buffer; slice w of the weight buffer it fills next is its transposition scratch in the epilogue.  HEAD: the output-layer weights and
// the row exchange live in the row granule that is free during the epilogue (re-read per tile, two more barriers per tile).
constexpr int kGxThreads = 256;
constexpr int kGxW = kBxStage / 2 * 16;                                  // bytes of one weight buffer (24 KB)
constexpr size_t kGxSmem = 2 * kGxW + 2 * kGlRows;                       // 80 KB
template <int EPI, int NPROD, bool HEAD, bool PK, bool W0 = false>   // PK: the sines carry the sign of their cosine, no cosines are stored (out1 == nullptr); W0: as mlp_nt_bx
__global__ __launch_bounds__(kGxThreads, 2) void mlp_nt_gx(const NtArgs p, const uint4* __restrict__ wsplit, const HeadArgs hd) {
  extern __shared__ __align__(16) unsigned char gx_smem[];
  // NPROD == 3: two f16 pieces per operand (split2h), three products; the weight image of a half step is 16 KB ([2 pieces][2 g][256 n] x 16 B),
  // slice (piece, g) = wave w's four DMA pieces, kept at the 6144-byte slice pitch of the bf16 form (the epilogue's scratch needs 4608)
  constexpr bool F16 = NPROD == 3;
  constexpr bool BLK = F16 && EPI == EPI_MULC;                            // the rows are loss gradients: one exponent per 128-row tile (block_scale)
  static_assert(!F16 || EPI != EPI_BIAS, "the two-piece f16 form: sine layers forward (sines in, unit scale) and input gradients (block-scaled rows)");
  constexpr int kSlicePitch = F16 ? 384 : 256;                            // uint4 between (piece, g) slices in LDS
  constexpr size_t kStepBytes = F16 ? 16384 : (size_t)kGxW;               // one half step of the global image
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, li = lane & 31, lh = lane >> 5;
  const int ng = (p.K + 31) / 32;                                        // granules per tile (a ragged last one: columns at and beyond K are scratch)
  const int tiles = p.M / kBM;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds0 = lds_byte_address(gx_smem);
  const unsigned w_dst = lds0 + (unsigned)wave_u * 6144u, a_dst = lds0 + 2u * kGxW + (unsigned)wave_u * 4096u;
  const unsigned w_voff = (unsigned)((F16 ? 256 : 384) * wave + lane) * 16u;
  unsigned a_voff[4], a_rd[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int row = 32 * wave + 8 * j + (lane >> 3);
    a_voff[j] = (unsigned)(row * p.lda + 4 * ((lane & 7) ^ ((row >> 1) & 7))) * 4u;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) a_rd[q] = (unsigned)(2 * kGxW + (wm * 64 + li) * 128 + (((4 * lh + q) ^ ((li >> 1) & 7)) * 16));
  float bn[4];
  float4 csum4[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) {
    bn[ni] = (EPI != EPI_MULC) ? p.bias[wn * 128 + ni * 32 + li] : 0.f;
    csum4[ni] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  f32x4v acc0[4][2];                                                     // W0: dW0 of this wave's rows: [column block][16-column half], a 16 x 16 tile each
  if (W0) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) acc0[ni][ct] = f32x4v{0.f, 0.f, 0.f, 0.f};
  }
  const int b_lane = lh * kSlicePitch + wn * 128 + li;
  auto issue_w = [&](int step_w, int wb) {                                // step_w: half super-step index within the reduction
    const char* src = reinterpret_cast<const char*>(wsplit) + (size_t)step_w * kStepBytes;
    if (F16) {
      glds16_x4(src, w_voff, w_dst + (unsigned)wb * (unsigned)kGxW);
    } else {
      glds16_x3(src, w_voff, w_dst + (unsigned)wb * (unsigned)kGxW);
      glds16_x3(src + 3072, w_voff, w_dst + (unsigned)wb * (unsigned)kGxW + 3072u);
    }
  };
  auto issue_a = [&](int tile_a, int g_a, int ab) {
    const float* src = p.A + (size_t)tile_a * kBM * p.lda + 32 * g_a;
    glds16_x4v(src, a_voff[0], a_voff[1] - 1024u, a_voff[2] - 2048u, a_voff[3] - 3072u, a_dst + (unsigned)ab * (unsigned)kGlRows);
  };
  issue_w(0, 0);
  issue_a(blockIdx.x, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  int wb = 0, ab = 0;

  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const int row0 = tile * kBM;
    float a_scale = 1.0f, a_unscale = kF16WUnscale, omax = 0.f;
    if (BLK) {
      block_scale(p.a_tmax[tile], a_scale, a_unscale);
      a_unscale *= kF16WUnscale;
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    for (int g = 0; g < ng; ++g) {
      int t2 = tile, g2 = g + 1;                                           // the next granule of the stream (beyond the last tile: this one again, unused)
      if (g2 >= ng) { g2 = 0; t2 = tile + (int)gridDim.x < tiles ? tile + (int)gridDim.x : tile; }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int step_n = s == 0 ? 2 * g + 1 : (g + 1 < ng ? 2 * g + 2 : 0);
        issue_w(step_n, wb ^ 1);
        if (s == 0) issue_a(t2, g2, ab ^ 1);
        uint4 aq[2][3];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          float4 u = *reinterpret_cast<const float4*>(gx_smem + a_rd[2 * s] + ab * kGlRows + mi * 4096);
          float4 v = *reinterpret_cast<const float4*>(gx_smem + a_rd[2 * s + 1] + ab * kGlRows + mi * 4096);
          if (32 * (g + 1) > p.K) {                           // uniform: scratch columns are not zeros
            const int k = 32 * g + 16 * lh + 8 * s;
            u.x = k < p.K ? u.x : 0.f; u.y = k + 1 < p.K ? u.y : 0.f; u.z = k + 2 < p.K ? u.z : 0.f; u.w = k + 3 < p.K ? u.w : 0.f;
            v.x = k + 4 < p.K ? v.x : 0.f; v.y = k + 5 < p.K ? v.y : 0.f; v.z = k + 6 < p.K ? v.z : 0.f; v.w = k + 7 < p.K ? v.w : 0.f;
          }
          if (BLK) {
            u.x *= a_scale; u.y *= a_scale; u.z *= a_scale; u.w *= a_scale;
            v.x *= a_scale; v.y *= a_scale; v.z *= a_scale; v.w *= a_scale;
          }
          if (F16) {
            split2h(u.x, u.y, aq[mi][0].x, aq[mi][1].x);
            split2h(u.z, u.w, aq[mi][0].y, aq[mi][1].y);
            split2h(v.x, v.y, aq[mi][0].z, aq[mi][1].z);
            split2h(v.z, v.w, aq[mi][0].w, aq[mi][1].w);
          } else {
            split3(u.x, u.y, aq[mi][0].x, aq[mi][1].x, aq[mi][2].x);
            split3(u.z, u.w, aq[mi][0].y, aq[mi][1].y, aq[mi][2].y);
            split3(v.x, v.y, aq[mi][0].z, aq[mi][1].z, aq[mi][2].z);
            split3(v.z, v.w, aq[mi][0].w, aq[mi][1].w, aq[mi][2].w);
          }
        }
        const uint4* sb = reinterpret_cast<const uint4*>(gx_smem + wb * kGxW) + b_lane;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          uint4 bq[3];
#pragma unroll
          for (int piece = 0; piece < (F16 ? 2 : 3); ++piece) bq[piece] = sb[(piece * 2) * kSlicePitch + ni * 32];
          if constexpr (F16) {                                 // p2 q1, p1 q2, p1 q1: the small terms first
#pragma unroll
            for (int t = 0; t < 3; ++t) {
              constexpr int ia[3] = {1, 0, 0}, ib[3] = {0, 1, 0};
#pragma unroll
              for (int mi = 0; mi < 2; ++mi)
                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, aq[mi][ia[t]]), __builtin_bit_cast(f16x8, bq[ib[t]]),
                                                                     acc[mi][ni], 0, 0, 0);
            }
          }
          // products from the smallest terms up, as mlp_nt_bx
#pragma unroll
          for (int t = 0; t < (F16 ? 0 : 9); ++t) {
            constexpr int ia[9] = {2, 1, 2, 2, 0, 1, 1, 0, 0}, ib[9] = {2, 2, 1, 0, 2, 1, 0, 1, 0};
            if (NPROD == 6 && t < 3) continue;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aq[mi][ia[t]]), __builtin_bit_cast(bf16x8, bq[ib[t]]),
                                                                    acc[mi][ni], 0, 0, 0);
          }
        }
        if (s == 0) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        wb ^= 1;
      }
      ab ^= 1;
    }
    // ---- epilogue.  Scratch: this wave's slice of the weight buffer read last (it fills it next).  HEAD: weights and row exchange in
    // the row granule read last (the next DMA into it is issued after the barriers below)
    float* scr = reinterpret_cast<float*>(gx_smem + (size_t)(wb ^ 1) * kGxW + wave * 6144);
    float* sW4 = reinterpret_cast<float*>(gx_smem + 2 * kGxW + (size_t)(ab ^ 1) * kGlRows);   // [5][256]
    float* sComb = sW4 + 5 * 256;                                                          // [128 rows][2 column halves][8]
    const int t_row = lane >> 3, t_col = (lane & 7) * 4;
    if (HEAD) {
      for (int i = tid; i < 5 * 256; i += kGxThreads) sW4[i] = hd.w[(size_t)(i >> 8) * hd.ldw + (i & 255)];
      __syncthreads();
    }
    // W0: this wave's 64 rows of x0 in ITS slice of the free row granule (the slice it fills next itself): lane = row
    float* sx = reinterpret_cast<float*>(gx_smem + 2 * kGxW + (size_t)(ab ^ 1) * kGlRows + wave * 4096);
    if (W0) {
      const float* xs = p.x0 + (size_t)(row0 + wm * 64 + lane) * p.ldx0;
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(sx + lane * 16 + 4 * ((q + lane) & 3)) = *reinterpret_cast<const float4*>(xs + 4 * ((q + lane) & 3));
    }
    // eight 32 x 32 blocks per wave, b = (row half mi, column block ni); EPI_MULC: the cos operand of block b + 1 is requested
    // before block b is worked on (two sets of four registers quadruples: what one column pair took before)
    float4 cvb[2][4];
    auto cv_load = [&](int b, float4 (&dst)[4]) {
      const float* src = p.cmul + (size_t)(row0 + wm * 64 + (b >> 2) * 32 + t_row) * p.ldo + wn * 128 + (b & 3) * 32 + t_col;
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) dst[ps] = *reinterpret_cast<const float4*>(src + (size_t)(8 * ps) * p.ldo);
    };
    if (EPI == EPI_MULC) cv_load(0, cvb[0]);
    float hacc[4][5];
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const int mi = b >> 2, ni = b & 3;
      const size_t o0 = (size_t)(row0 + wm * 64 + mi * 32 + t_row) * p.ldo + wn * 128 + ni * 32 + t_col;
      if (EPI == EPI_MULC && b + 1 < 8) cv_load(b + 1, cvb[(b + 1) & 1]);
      if (HEAD && ni == 0) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
#pragma unroll
          for (int j = 0; j < 5; ++j) hacc[ps][j] = 0.f;
      }
      float second[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[mi][ni][r];
        if (BLK) v *= a_unscale;
        if (EPI == EPI_SINCOS) {
          const float pre = F16 ? __builtin_fmaf(v, kF16WUnscale, bn[ni]) : v + bn[ni];
          if (PK) v = sin_packed(pre);
          else sincos_cw(pre, v, second[r]);
        }
        scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * kLd + li] = v;
      }
      float4 vrow[4];
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        float4 v = *reinterpret_cast<const float4*>(scr + (t_row + 8 * ps) * kLd + t_col);
        if (EPI == EPI_MULC) {
          float4 c4 = cvb[b & 1][ps];
          if (p.cmul_sin) c4 = cos_from_packed_sin(c4);
          v.x *= c4.x; v.y *= c4.y; v.z *= c4.z; v.w *= c4.w;
          csum4[ni].x += v.x; csum4[ni].y += v.y; csum4[ni].z += v.z; csum4[ni].w += v.w;
          if (BLK && !W0) omax = __builtin_fmaxf(__builtin_fmaxf(omax, __builtin_fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y))),
                                                   __builtin_fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
        }
        if (EPI == EPI_SINCOS && !HEAD && ni == 3 && p.tail != nullptr && wn == 1) {
            // a skip layer's buffer: the columns at and beyond N hold x0 (mymodels/mlps.py:214-217), written here instead of by a copy launch
            const int col = 224 + t_col;
            const float* tl = p.tail + (size_t)(row0 + wm * 64 + mi * 32 + t_row + 8 * ps) * p.ldt - p.N;
            if (col >= p.N) v.x = tl[col];
            if (col + 1 >= p.N) v.y = tl[col + 1];
            if (col + 2 >= p.N) v.z = tl[col + 2];
            if (col + 3 >= p.N) v.w = tl[col + 3];
        }
        if (W0) *reinterpret_cast<float4*>(scr + (t_row + 8 * ps) * kLd + t_col) = v;   // G' back into the block, for the product below
        else *reinterpret_cast<float4*>(p.out0 + o0 + (size_t)(8 * ps) * p.ldo) = v;
        vrow[ps] = v;
      }
      if (W0) {     // dW0[16 ct + i][j] += sum over the block's 32 rows of G'[row][16 ct + i] x0[row][j]  (A: lane = (i, k), B: lane = (j, k))
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int kk = 0; kk < 8; ++kk)
            acc0[ni][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(scr[(4 * kk + (lane >> 4)) * kLd + 16 * ct + (lane & 15)],
                                                                sx[(mi * 32 + 4 * kk + (lane >> 4)) * 16 + (lane & 15)], acc0[ni][ct], 0, 0, 0);
      }
      if (HEAD) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
          const float4 w4 = *reinterpret_cast<const float4*>(sW4 + j * 256 + wn * 128 + ni * 32 + t_col);
#pragma unroll
          for (int ps = 0; ps < 4; ++ps)
            hacc[ps][j] = __builtin_fmaf(vrow[ps].x, w4.x, __builtin_fmaf(vrow[ps].y, w4.y, __builtin_fmaf(vrow[ps].z, w4.z, __builtin_fmaf(vrow[ps].w, w4.w, hacc[ps][j]))));
        }
      }
      if (EPI == EPI_SINCOS && !PK) {
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * kLd + li] = second[r];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
          *reinterpret_cast<float4*>(p.out1 + o0 + (size_t)(8 * ps) * p.ldo) = *reinterpret_cast<const float4*>(scr + (t_row + 8 * ps) * kLd + t_col);
      }
      if (HEAD && ni == 3) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
#pragma unroll
          for (int j = 0; j < 5; ++j) {
            float v = hacc[ps][j];
            v = dpp_add_f<0x111>(v);
            v = dpp_add_f<0x112>(v);
            v = dpp_add_f<0x114>(v);
            if ((lane & 7) == 7) sComb[((wm * 64 + mi * 32 + t_row + 8 * ps) * 2 + wn) * 8 + j] = v;
          }
      }
    }
    if (HEAD) {
      __syncthreads();
      if (tid < kBM) {
        float v5[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) v5[j] = (sComb[(tid * 2) * 8 + j] + sComb[(tid * 2 + 1) * 8 + j]) + hd.bias[j];
        arm_head_store(hd.h, (long)row0 + tid, v5);
      }
      __syncthreads();                                        // the granule is the next DMA target
    }
    if (BLK && !W0 && p.o_tmax != nullptr) {                    // this wave's share of the tile's largest |G'|: the next product's block exponent
      const float wmx = wave_max_lane63(omax);
      if (lane == 63) atomicMax(p.o_tmax + tile, __float_as_uint(wmx));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the scratch slice is this wave's next DMA target
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the unused look-ahead pieces: nothing may land after the workgroup ends
  if (W0) {                                                     // lane holds dW0[n = .. + 4 (lane >> 4) + r][k = lane & 15]
    const size_t slab = (size_t)blockIdx.x * 2 + wm;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          p.w0_part[(slab * 16 + (lane & 15)) * 256 + wn * 128 + ni * 32 + 16 * ct + 4 * (lane >> 4) + r] = acc0[ni][ct][r];
  }
  if (EPI == EPI_MULC && p.colsum != nullptr) {
    float* sRed = reinterpret_cast<float*>(gx_smem);                      // [16][256]
    const int t_row = lane >> 3, t_col = (lane & 7) * 4;
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
      *reinterpret_cast<float4*>(sRed + (wm * 8 + t_row) * 256 + wn * 128 + ni * 32 + t_col) = csum4[ni];
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < 16; ++sl) t += sRed[sl * 256 + tid];
    p.colsum[(size_t)blockIdx.x * 256 + tid] = t;
  }
}

// dW partials = G^T X over a slab of rows with split operands.  One workgroup owns the whole 256 x 256 output of its slab: eight
// waves, wave (nq, kq) accumulates 64 x 128 of it (8 accumulator tiles), two waves per SIMD so that one wave's staging work (VALU
// split, LDS traffic) runs in the shadow of the other's MFMAs.  Both operands are activations laid out [row m][column] with m the
// reduction index, so the MFMA fragments (8 consecutive m of one column per lane) are built by the staging pass: a thread takes 2
// adjacent columns x 8 rows (8 8-byte loads; a wave's load is a 512-byte run of a row), cuts the 16 values into three bf16 pieces
// once -- every staged element is then read by 2 (G) or 4 (X) waves -- and writes one 16-byte LDS word per (piece, column).
// Columns are stored permuted within their group of 16 (slot = 8 (c & 1) + (c >> 1 & 7)): the stores of 8 adjacent threads and the
// fragment reads of 8 adjacent lanes both cover 8 distinct 16-byte bank groups; lane l of a fragment therefore holds column
// wg_perm(l) of its tile, undone when the partials are stored.
// Stage = 16 rows x 512 columns x 3 pieces = 48 KB, double-buffered; the rows of the next kWgDepth stages are in registers or in
// flight (the loop runs at the latency of its loads unless enough bytes are in flight).
// Columns at or beyond N (K) are computed and dropped by mlp_wgrad_reduce: dW[n][k] depends on column n of G and k of X only.
constexpr int kWgStage = 2 * 3 * 2 * 256;                  // uint4 per buffer: [G|X][piece][row half][column slot]
constexpr size_t kWgSmem = 2 * kWgStage * sizeof(uint4);
#ifndef MATPBR_WG_DEPTH
#define MATPBR_WG_DEPTH 2
#endif
constexpr int kWgDepth = MATPBR_WG_DEPTH;      // (measurement builds: tools/ab.sh)
constexpr int kWgThreads = 512;
__device__ __forceinline__ int wg_perm(int i) { return (i & 16) + 2 * (i & 7) + ((i >> 3) & 1); }   // fragment lane -> column within a tile of 32
// EARLY: stage before the products of a step instead of after them.  The two waves of a SIMD (w and w + 4) meet at the barrier of
// every step; with opposite orders one of them splits and stores while the other one multiplies.
template <int NPROD, bool EARLY>
__device__ __forceinline__ void wgrad_bx_body(const float* __restrict__ G, int ldg, const float* __restrict__ X, int ldx,
                                              float* __restrict__ partial, long M, long rows_per_slab, uint4* sW, const unsigned* __restrict__ g_tmax) {
  // NPROD == 3: two f16 pieces per operand.  X (sines, the x0 tail) as it is; G under ONE exponent per slab, from the largest of the
  // tile maxima of the slab's rows (block_scale); the partial sums are scaled back when they are stored
  constexpr bool F16 = NPROD == 3;
  constexpr int NP = F16 ? 2 : 3;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nq = wave >> 1, kq = wave & 1, li = lane & 31, lh = lane >> 5;
  const long m_begin = (long)blockIdx.x * rows_per_slab;
  const long m_end = (m_begin + rows_per_slab < M) ? m_begin + rows_per_slab : M;
  const int steps = m_begin < m_end ? (int)((m_end - m_begin) / 16) : 0;
  float g_scale = 1.0f, g_unscale = 1.0f;
  if (F16 && steps > 0) {
    unsigned mb = 0;
    for (long t = m_begin / kBM; t <= (m_end - 1) / kBM; ++t) mb = g_tmax[t] > mb ? g_tmax[t] : mb;      // magnitudes: bit patterns order as values
    block_scale(mb, g_scale, g_unscale);
  }

  f32x16 acc[2][4];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int ki = 0; ki < 4; ++ki)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][ki][r] = 0.f;

  // staging: this thread's 2 columns (pair cp of the 256 pairs of G | X) and row half sh
  const int cp = tid & 255, sh = tid >> 8;
  const float* src = (cp >> 7) ? X + (m_begin + 8 * sh) * ldx + 2 * (cp & 127) : G + (m_begin + 8 * sh) * ldg + 2 * (cp & 127);
  const long ld = (cp >> 7) ? ldx : ldg;
  const float my_scale = (F16 && !(cp >> 7)) ? g_scale : 1.0f;
  uint4* sdst = sW + (((cp >> 7) * NP) * 2 + sh) * 256 + ((cp & 127) >> 3) * 16 + (cp & 7);   // + buf * kWgStage + (piece * 2) * 256 + 8 * column
  float2 raw[kWgDepth][8];
  auto load_stage = [&](int slot, int step) {
#pragma unroll
    for (int j = 0; j < 8; ++j) raw[slot][j] = *reinterpret_cast<const float2*>(src + (16L * step + j) * ld);
  };
  auto split_store = [&](int slot, int buf) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      uint4 pc[3];
      const float2* r = raw[slot];
#define MATPBR_COMP(v) (c == 0 ? (v).x : (v).y)
      if (F16) {
        split2h(MATPBR_COMP(r[0]) * my_scale, MATPBR_COMP(r[1]) * my_scale, pc[0].x, pc[1].x);
        split2h(MATPBR_COMP(r[2]) * my_scale, MATPBR_COMP(r[3]) * my_scale, pc[0].y, pc[1].y);
        split2h(MATPBR_COMP(r[4]) * my_scale, MATPBR_COMP(r[5]) * my_scale, pc[0].z, pc[1].z);
        split2h(MATPBR_COMP(r[6]) * my_scale, MATPBR_COMP(r[7]) * my_scale, pc[0].w, pc[1].w);
      } else {
        split3(MATPBR_COMP(r[0]), MATPBR_COMP(r[1]), pc[0].x, pc[1].x, pc[2].x);
        split3(MATPBR_COMP(r[2]), MATPBR_COMP(r[3]), pc[0].y, pc[1].y, pc[2].y);
        split3(MATPBR_COMP(r[4]), MATPBR_COMP(r[5]), pc[0].z, pc[1].z, pc[2].z);
        split3(MATPBR_COMP(r[6]), MATPBR_COMP(r[7]), pc[0].w, pc[1].w, pc[2].w);
      }
#undef MATPBR_COMP
#pragma unroll
      for (int piece = 0; piece < NP; ++piece) sdst[buf * kWgStage + (piece * 2) * 256 + 8 * c] = pc[piece];
    }
  };
  // unconditional (the last steps re-stage the last rows, which nobody reads): loads under a branch could not be counted.
  // The scheduling fences keep the loads in stage order: the count of loads in flight that the compiler derives for the waits in
  // the loop is the minimum over the loop entry and the back edge.
  auto stage = [&](int slot, int buf, int next_stage) {
    __builtin_amdgcn_sched_barrier(0);
    split_store(slot, buf);
    __builtin_amdgcn_sched_barrier(0);
    load_stage(slot, next_stage < steps ? next_stage : steps - 1);
    __builtin_amdgcn_sched_barrier(0);
  };

  if (steps > 0) {
    load_stage(0, 0);
    split_store(0, 0);
#pragma unroll
    for (int d = 0; d < kWgDepth; ++d) {
      __builtin_amdgcn_sched_barrier(0);
      load_stage((1 + d) % kWgDepth, 1 + d < steps ? 1 + d : steps - 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // step st multiplies from buffer st & 1 and stages step st + 1 (registers (st + 1) % depth) into the other one.  Whole groups of kWgDepth
    // steps in the loop, the rest behind it: an exit from the middle of the unrolled body shares the loop's latch, and the waits the compiler
    // derives at the loop header are then the merge of both orders of the loads in flight -- vmcnt(0) at every group's first step
    auto step = [&](auto u_tag, int st) {
        constexpr int u = decltype(u_tag)::value;
        const int cur = st & 1;
        if (EARLY) stage((u + 1) % kWgDepth, cur ^ 1, st + 1 + kWgDepth);
        const uint4* sa = sW + cur * kWgStage + lh * 256 + nq * 64 + li;             // + (piece * 2) * 256 + ni * 32
        const uint4* sb = sW + cur * kWgStage + (NP * 2 + lh) * 256 + kq * 128 + li;  // + (piece * 2) * 256 + ki * 32
        uint4 a[3][2];
#pragma unroll
        for (int piece = 0; piece < NP; ++piece)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) a[piece][ni] = sa[(piece * 2) * 256 + ni * 32];
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          uint4 b[3][2];
#pragma unroll
          for (int piece = 0; piece < NP; ++piece)
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) b[piece][k2] = sb[(piece * 2) * 256 + (kh * 2 + k2) * 32];
          if constexpr (F16) {
#pragma unroll
            for (int t = 0; t < 3; ++t) {
              constexpr int ia[3] = {1, 0, 0}, ib[3] = {0, 1, 0};
#pragma unroll
              for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
                  acc[ni][kh * 2 + k2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ia[t]][ni]),
                                                                               __builtin_bit_cast(f16x8, b[ib[t]][k2]), acc[ni][kh * 2 + k2], 0, 0, 0);
            }
          }
#pragma unroll
          for (int t = 0; t < (F16 ? 0 : 9); ++t) {
            constexpr int ia[9] = {2, 1, 2, 2, 0, 1, 1, 0, 0}, ib[9] = {2, 2, 1, 0, 2, 1, 0, 1, 0};
            if (NPROD == 6 && t < 3) continue;
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
              for (int ni = 0; ni < 2; ++ni)
                acc[ni][kh * 2 + k2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ia[t]][ni]),
                                                                              __builtin_bit_cast(bf16x8, b[ib[t]][k2]), acc[ni][kh * 2 + k2], 0, 0, 0);
          }
        }
        if (!EARLY) stage((u + 1) % kWgDepth, cur ^ 1, st + 1 + kWgDepth);
        __syncthreads();
    };
    int st0 = 0;
    for (; st0 + kWgDepth <= steps; st0 += kWgDepth) {
      step(std::integral_constant<int, 0>{}, st0);
      if constexpr (kWgDepth > 1) step(std::integral_constant<int, 1>{}, st0 + 1);
      if constexpr (kWgDepth > 2) step(std::integral_constant<int, 2>{}, st0 + 2);
      if constexpr (kWgDepth > 3) step(std::integral_constant<int, 3>{}, st0 + 3);
    }
    if (st0 < steps) step(std::integral_constant<int, 0>{}, st0);
    if constexpr (kWgDepth > 2) if (st0 + 1 < steps) step(std::integral_constant<int, 1>{}, st0 + 1);
    if constexpr (kWgDepth > 3) if (st0 + 2 < steps) step(std::integral_constant<int, 2>{}, st0 + 2);
  }
  float* out = partial + (long)blockIdx.x * 256 * 256;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int ki = 0; ki < 4; ++ki)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nq * 64 + ni * 32 + wg_perm((r & 3) + 8 * (r >> 2) + 4 * lh);
        const int k = kq * 128 + ki * 32 + wg_perm(li);
        out[n * 256 + k] = F16 ? acc[ni][ki][r] * g_unscale : acc[ni][ki][r];
      }
}

template <int NPROD>
__global__ __launch_bounds__(kWgThreads, 1) void mlp_wgrad_bx(const float* __restrict__ G, int ldg, const float* __restrict__ X, int ldx,
                                                              float* __restrict__ partial, long M, long rows_per_slab, const unsigned* __restrict__ g_tmax) {
  extern __shared__ __align__(16) unsigned char wg_smem[];
  uint4* sW = reinterpret_cast<uint4*>(wg_smem);
  if (threadIdx.x >> 8)
    wgrad_bx_body<NPROD, true>(G, ldg, X, ldx, partial, M, rows_per_slab, sW, g_tmax);
  else
    wgrad_bx_body<NPROD, false>(G, ldg, X, ldx, partial, M, rows_per_slab, sW, g_tmax);
}

// mlp_wgrad_hx: the two-piece f16 weight gradient with its rows brought in by LDS-DMA and cut ONCE (see EXPERIMENTS.md, "where the weight
// gradient's time goes").  A wave that ISSUES memory instructions while the memory pipeline is full waits at the issue -- and does not
// multiply meanwhile: with every wave issuing its share after the same barrier, rows that arrive in 65-85 us by themselves and products that
// take 71 us by themselves took their SUM (131-145 us, stamps).  Here only waves 0..3 (one per SIMD) issue the DMA of a step, first thing after
// the barrier, while their partners on the same SIMDs (waves 4..7) have the matrix pipe; then they multiply: loop 120 us, the pair with its
// fold 150-153 us against 158-161 of mlp_wgrad_bx<3> on the same boxes (tools/ab.sh tools/wg_time.py "-DMATPBR_WG_HX=0").
//   * rows: f32 as they are, 16 rows x [G | X] = 32 KB a stage, 32 one-KB pieces (eight per issuing wave, scalar bases), ring of three stages;
//   * every thread cuts ITS 2 columns x 8 rows of the NEXT step out of the landed stage (what mlp_wgrad_bx cuts) into mlp_wgrad_bx's fragment
//     image (one 32 KB buffer), behind the products of the CURRENT step, whose fragments are already in registers;
//   * two barriers a step: A -- the pieces of this step are written and the rows of the next have landed; B -- everybody holds its
//     fragments and its rows, the piece buffer and the oldest raw stage may be rewritten.
// Same pieces, same products in the same order as mlp_wgrad_bx<3>: the same bits.
#ifndef MATPBR_WG_HX
#define MATPBR_WG_HX 1                         // (measurement builds: 0 = the register-staged mlp_wgrad_bx<3>)
#endif
constexpr int kHxRaw = 16 * 2 * 1024;                      // bytes of a raw stage: [G rows 0..15][X rows 0..15], 1 KB each
constexpr int kHxPieces = 2 * 2 * 2 * 256;                 // uint4 of the piece buffer: [G|X][piece][row half][column slot]
constexpr size_t kHxSmem = 3 * kHxRaw + kHxPieces * sizeof(uint4);     // 128 KB
__global__ __launch_bounds__(kWgThreads, 1) void mlp_wgrad_hx(const float* __restrict__ G, int ldg, const float* __restrict__ X, int ldx,
                                                              float* __restrict__ partial, long M, long rows_per_slab, const unsigned* __restrict__ g_tmax) {
  extern __shared__ __align__(16) unsigned char hx_smem[];
  uint4* sP = reinterpret_cast<uint4*>(hx_smem + 3 * kHxRaw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nq = wave >> 1, kq = wave & 1, li = lane & 31, lh = lane >> 5;
  // this workgroup's rows: the 128-row tiles T - 1 - b, T - 1 - b - grid, ... of the T tiles, each from its last rows to its first.  Interleaved,
  // so that at any moment the launch reads ONE region of G and of X; DESCENDING, because the kernels on either side of it (the input
  // gradients: tiles in ascending order) end at the last rows and start at the first -- what one kernel touched last is what the next
  // touches first, and the 256 MB Infinity Cache still holds it (tools/ab.sh tools/wg_time.py "-DMATPBR_WG_ASCENDING")
  (void)rows_per_slab;
  const int tiles = (int)(M / kBM);
  const int t_first = tiles - 1 - (int)blockIdx.x;
  const int my_tiles = t_first >= 0 ? t_first / (int)gridDim.x + 1 : 0;
  const int steps = my_tiles * (kBM / 16);
  float g_scale = 1.0f, g_unscale = 1.0f;
  if (steps > 0) {
    unsigned mb = 0;
    for (int t = t_first; t >= 0; t -= (int)gridDim.x) mb = g_tmax[t] > mb ? g_tmax[t] : mb;      // magnitudes: bit patterns order as values
    block_scale(mb, g_scale, g_unscale);
  }
  f32x16 acc[2][4];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int ki = 0; ki < 4; ++ki)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ni][ki][r] = 0.f;
  if (steps > 0) {
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool loader = wave_u < 4;                               // (uniform per wave)
    const unsigned lds0 = lds_byte_address(hx_smem);
    unsigned vg[4], vx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      vg[j] = (unsigned)(j * ldg) * 4u + (unsigned)lane * 16u - 1024u * j;
      vx[j] = (unsigned)(j * ldx) * 4u + (unsigned)lane * 16u - 1024u * j;
    }
    auto issue = [&](int step, int buf) {                         // a loader wave's rows 4 w .. 4 w + 3 of the step, of G and of X
      const int sc = step < steps ? step : steps - 1;              // (beyond the slab: the last rows again, into a buffer nobody reads)
      const long row = ((long)t_first - (long)(sc >> 3) * gridDim.x) * kBM + 16 * (7 - (sc & 7)) + 4 * wave_u;
      const unsigned dst = lds0 + (unsigned)buf * (unsigned)kHxRaw + (unsigned)wave_u * 4096u;
      glds16_x4v(G + row * ldg, vg[0], vg[1], vg[2], vg[3], dst);
      glds16_x4v(X + row * ldx, vx[0], vx[1], vx[2], vx[3], dst + 16u * 1024u);
    };
    // the cut: this thread's 2 columns (pair cp of the 256 pairs of G | X) and row half sh, as mlp_wgrad_bx stages them
    const int cp = tid & 255, sh = tid >> 8;
    const float my_scale = (cp >> 7) ? 1.0f : g_scale;
    const unsigned raw_off = (unsigned)((cp >> 7) * 16 * 1024 + (8 * sh) * 1024 + (cp & 127) * 8);          // + buf * kHxRaw + j * 1024
    uint4* sdst = sP + (((cp >> 7) * 2) * 2 + sh) * 256 + ((cp & 127) >> 3) * 16 + (cp & 7);                  // + (piece * 2) * 256 + 8 * column
    float2 raw[8];
    auto read_raw = [&](int buf) {
#pragma unroll
      for (int j = 0; j < 8; ++j) raw[j] = *reinterpret_cast<const float2*>(hx_smem + buf * kHxRaw + raw_off + j * 1024);
    };
    auto cut_store = [&]() {
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        uint4 pc[2];
#define MATPBR_COMP(v) (c == 0 ? (v).x : (v).y)
        split2h(MATPBR_COMP(raw[0]) * my_scale, MATPBR_COMP(raw[1]) * my_scale, pc[0].x, pc[1].x);
        split2h(MATPBR_COMP(raw[2]) * my_scale, MATPBR_COMP(raw[3]) * my_scale, pc[0].y, pc[1].y);
        split2h(MATPBR_COMP(raw[4]) * my_scale, MATPBR_COMP(raw[5]) * my_scale, pc[0].z, pc[1].z);
        split2h(MATPBR_COMP(raw[6]) * my_scale, MATPBR_COMP(raw[7]) * my_scale, pc[0].w, pc[1].w);
#undef MATPBR_COMP
#pragma unroll
        for (int piece = 0; piece < 2; ++piece) sdst[(piece * 2) * 256 + 8 * c] = pc[piece];
      }
    };
    auto products = [&](const uint4 (&a)[2][2], const uint4 (&b)[2][4]) {
#pragma unroll
      for (int ki = 0; ki < 4; ++ki)
#pragma unroll
        for (int t = 0; t < 3; ++t) {                                    // p2 q1, p1 q2, p1 q1: the small terms first
          constexpr int ia[3] = {1, 0, 0}, ib[3] = {0, 1, 0};
#pragma unroll
          for (int ni = 0; ni < 2; ++ni)
            acc[ni][ki] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[ia[t]][ni]), __builtin_bit_cast(f16x8, b[ib[t]][ki]), acc[ni][ki], 0, 0, 0);
        }
    };
    if (loader) { issue(0, 0); issue(1, 1); }
    if (loader) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");                              // the rows of step 0 have landed
    read_raw(0);
    cut_store();
    if (loader) issue(2, 2);
    const uint4* sa = sP + lh * 256 + nq * 64 + li;                      // + (piece * 2) * 256 + ni * 32
    const uint4* sb = sP + (2 * 2 + lh) * 256 + kq * 128 + li;           // + (piece * 2) * 256 + ki * 32
    int nxt = 1;                                                         // the raw buffer of step st + 1
    for (int st = 0; st < steps; ++st) {
      if (loader) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // the rows of st + 1 have landed (st + 2 in flight)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // A: pieces of st written by everybody
      uint4 a[2][2], b[2][4];
#pragma unroll
      for (int piece = 0; piece < 2; ++piece) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) a[piece][ni] = sa[(piece * 2) * 256 + ni * 32];
#pragma unroll
        for (int ki = 0; ki < 4; ++ki) b[piece][ki] = sb[(piece * 2) * 256 + ki * 32];
      }
      read_raw(nxt);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");    // B: everybody holds its fragments and its rows of st + 1
      if (loader) {
        issue(st + 3, nxt == 0 ? 2 : nxt - 1);                            // into the buffer of step st (cut during step st - 1)
        __builtin_amdgcn_sched_barrier(0);
      }
      products(a, b);
      cut_store();                                                       // the pieces of step st + 1 (beyond the slab: of its last rows, unused)
      nxt = nxt == 2 ? 0 : nxt + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                     // the look-ahead pieces: nothing may land after the workgroup has ended
  }
  float* out = partial + (long)blockIdx.x * 256 * 256;
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int ki = 0; ki < 4; ++ki)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = nq * 64 + ni * 32 + wg_perm((r & 3) + 8 * (r >> 2) + 4 * lh);
        const int k = kq * 128 + ki * 32 + wg_perm(li);
        out[n * 256 + k] = acc[ni][ki][r] * g_unscale;
      }
}

constexpr size_t kBxSmem = 2 * kBxStage * sizeof(uint4) + 8 * 32 * kLd * sizeof(float);   // 96 KB of weights + 36 KB of epilogue scratch
constexpr size_t kBxSmemHead = kBxSmem + (5 * 256 + 128 * 2 * 8) * sizeof(float);           // + output-layer weights and the row exchange
// More than 64 KB of dynamic LDS needs an opt-in attribute, which HIP keeps per device: one bit per (kernel, device), set under the
// device that is current at the launch (a process may drive several GPUs through the C ABI, from several threads).  A failure is
// reported to the caller and retried at the next launch.
std::atomic<int> g_nt_gl{2};
std::atomic<int> g_nt_w0_gx{0};   // measurement: mode 3 of matpbr_mlp_set_lds_dma                  // LDS-DMA main loop where the shape allows (matpbr_mlp_set_lds_dma: A/B switch)
inline bool gl_ok(const NtArgs& p) {
  return g_nt_gl.load(std::memory_order_relaxed) != 0 && p.K > 32 && (long)kBM * p.lda * 4 < (1l << 31);   // (a ragged K: the rows hold 32-column granules, lda >= 32 ceil(K / 32))
}
template <auto Kernel>
bool lds_opt_in(size_t bytes) {
  static std::atomic<unsigned long long> done{0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  const unsigned long long bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return true;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) return false;
  done.fetch_or(bit, std::memory_order_release);
  return true;
}
template <int EPI, int NPROD, bool FULL>
bool launch_nt_bx_full(const NtArgs& p, const uint4* wsplit, unsigned grid, hipStream_t stream) {
  if constexpr (FULL) if (gl_ok(p)) {
    if (!lds_opt_in<&mlp_nt_bx<EPI, NPROD, true, false, false, true>>(kGlSmem)) return false;
    hipLaunchKernelGGL((mlp_nt_bx<EPI, NPROD, true, false, false, true>), dim3(grid), dim3(kBxThreads), kGlSmem, stream, p, wsplit, HeadArgs{});
    return true;
  }
  if (!lds_opt_in<&mlp_nt_bx<EPI, NPROD, FULL>>(kBxSmem)) return false;
  hipLaunchKernelGGL((mlp_nt_bx<EPI, NPROD, FULL>), dim3(grid), dim3(kBxThreads), kBxSmem, stream, p, wsplit, HeadArgs{});
  return true;
}
constexpr size_t kBxSmemW0 = kBxSmem + 8 * 32 * 16 * sizeof(float);                          // + the waves' rows of x0
template <int NPROD>
bool launch_nt_bx_w0(const NtArgs& p, const uint4* wsplit, unsigned grid, hipStream_t stream) {
  if (gl_ok(p)) {
    constexpr size_t bytes = kGlSmem + 8 * 32 * 16 * sizeof(float);     // 160 KB: the whole LDS of a CU
    if (!lds_opt_in<&mlp_nt_bx<EPI_MULC, NPROD, true, false, true, true>>(bytes)) return false;
    hipLaunchKernelGGL((mlp_nt_bx<EPI_MULC, NPROD, true, false, true, true>), dim3(grid), dim3(kBxThreads), bytes, stream, p, wsplit, HeadArgs{});
    return true;
  }
  if (!lds_opt_in<&mlp_nt_bx<EPI_MULC, NPROD, true, false, true>>(kBxSmemW0)) return false;
  hipLaunchKernelGGL((mlp_nt_bx<EPI_MULC, NPROD, true, false, true>), dim3(grid), dim3(kBxThreads), kBxSmemW0, stream, p, wsplit, HeadArgs{});
  return true;
}
template <int NPROD>
bool launch_nt_bx_head(const NtArgs& p, const uint4* wsplit, const HeadArgs& hd, unsigned grid, hipStream_t stream) {
  if (gl_ok(p)) {
    constexpr size_t bytes = kGlSmem + (5 * 256 + 128 * 2 * 8) * sizeof(float);
    if (!lds_opt_in<&mlp_nt_bx<EPI_SINCOS, NPROD, true, true, false, true>>(bytes)) return false;
    hipLaunchKernelGGL((mlp_nt_bx<EPI_SINCOS, NPROD, true, true, false, true>), dim3(grid), dim3(kBxThreads), bytes, stream, p, wsplit, hd);
    return true;
  }
  if (!lds_opt_in<&mlp_nt_bx<EPI_SINCOS, NPROD, true, true>>(kBxSmemHead)) return false;
  hipLaunchKernelGGL((mlp_nt_bx<EPI_SINCOS, NPROD, true, true>), dim3(grid), dim3(kBxThreads), kBxSmemHead, stream, p, wsplit, hd);
  return true;
}
template <int EPI, int NPROD>
bool launch_nt_bx_one(const NtArgs& p, const uint4* wsplit, unsigned grid, hipStream_t stream) {
  // the input-gradient buffers have no tail to protect: columns at or beyond N of G' are scratch for every consumer
  if (p.N >= 256 || EPI == EPI_MULC || p.tail != nullptr) return launch_nt_bx_full<EPI, NPROD, true>(p, wsplit, grid, stream);
  return launch_nt_bx_full<EPI, NPROD, false>(p, wsplit, grid, stream);
}
template <int EPI, int NPROD, bool HEAD, bool PK>
bool launch_nt_gx_pk(const NtArgs& p, const uint4* wsplit, const HeadArgs& hd, unsigned grid, hipStream_t stream) {
  if (!lds_opt_in<&mlp_nt_gx<EPI, NPROD, HEAD, PK>>(kGxSmem)) return false;
  hipLaunchKernelGGL((mlp_nt_gx<EPI, NPROD, HEAD, PK>), dim3(grid), dim3(kGxThreads), kGxSmem, stream, p, wsplit, hd);
  return true;
}
template <int EPI, int NPROD, bool HEAD>
bool launch_nt_gx(const NtArgs& p, const uint4* wsplit, const HeadArgs& hd, unsigned grid, hipStream_t stream) {
  if constexpr (EPI == EPI_SINCOS) {
    if (p.out1 == nullptr) return launch_nt_gx_pk<EPI, NPROD, HEAD, true>(p, wsplit, hd, grid, stream);
  }
  return launch_nt_gx_pk<EPI, NPROD, HEAD, false>(p, wsplit, hd, grid, stream);
}
inline bool gx_ok(const NtArgs& p) { return g_nt_gl.load(std::memory_order_relaxed) == 2 && gl_ok(p); }
template <int EPI>
int launch_nt_bx(NtArgs p, const uint4* wsplit, int nprod, hipStream_t stream) {   // p.M a multiple of 128; returns the grid, -1 when the launch could not be set up
  const int tiles = p.M / kBM;
  if (nprod == 3) {                                                       // the two-piece f16 forms run on mlp_nt_gx only
    if constexpr (EPI != EPI_BIAS) {
      if (p.K > 32 && (long)kBM * p.lda * 4 < (1l << 31) && (p.N >= 256 || EPI == EPI_MULC || p.tail != nullptr) && (EPI != EPI_MULC || p.a_tmax != nullptr)) {
        const unsigned grid2 = (unsigned)(tiles < 512 ? tiles : 512);
        return launch_nt_gx<EPI, 3, false>(p, wsplit, HeadArgs{}, grid2, stream) ? (int)grid2 : -1;
      }
    }
    return -1;
  }
  if constexpr (EPI != EPI_BIAS) {
    if (gx_ok(p) && (p.N >= 256 || EPI == EPI_MULC || p.tail != nullptr)) {
      const unsigned grid2 = (unsigned)(tiles < 512 ? tiles : 512);
      const bool ok2 = nprod == 9 ? launch_nt_gx<EPI, 9, false>(p, wsplit, HeadArgs{}, grid2, stream) : launch_nt_gx<EPI, 6, false>(p, wsplit, HeadArgs{}, grid2, stream);
      return ok2 ? (int)grid2 : -1;
    }
  }
  const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
  const bool ok = nprod == 9 ? launch_nt_bx_one<EPI, 9>(p, wsplit, grid, stream) : launch_nt_bx_one<EPI, 6>(p, wsplit, grid, stream);
  return ok ? (int)grid : -1;
}

// (row tile, column half) work items; a workgroup must keep one column half across its persistent loop: grid multiple of 16
inline unsigned nt_grid(long M, int N) {
  const long rt = (M + kBM - 1) / kBM;
  const long items = N > kBN ? ((rt + 7) / 8) * 16 : rt;
  return (unsigned)(items < kPersistent ? items : kPersistent);
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Launches the product over all M rows: the pipelined kernel over the full 128-row tiles when the shape allows it, the general
// kernel over the ragged rest (or over everything).  Returns the number of column-sum rows written to p.colsum.
template <int EPI>
int launch_nt(NtArgs p, long M, hipStream_t stream) {
  const int halves = (p.N + kBN - 1) / kBN;
  const bool pipe = p.K > 224 && p.ldo >= halves * kBN && p.lda >= 256 && p.ldb >= 256 && M >= kBM;
  int groups = 0;
  long done = 0;
  // 256-wide outputs go to the full-width kernel (measured at M = 512x512: forward 371 vs 416 us, dL/d input 349 vs 356 us, whole
  // pos_mlp iteration 3.86 vs 4.04 ms); the pipelined kernel keeps the layers of at most 128 outputs
  if (pipe && halves == 2) {
    done = M / kBM * kBM;
    p.M = (int)done;
    const long rt = done / kBM;
    const unsigned grid = (unsigned)(rt < kPersistent ? rt : kPersistent);
    hipLaunchKernelGGL(mlp_gemm_nt_wide<EPI>, dim3(grid), dim3(256), 0, stream, p);
    groups += (int)grid;
  } else if (pipe) {
    done = M / kBM * kBM;
    p.M = (int)done;
    const unsigned grid = nt_grid(done, p.N);
    hipLaunchKernelGGL(mlp_gemm_nt_pipe<EPI>, dim3(grid), dim3(256), 0, stream, p);
    groups += (int)grid;
  }
  if (done < M) {
    NtArgs q = p;
    q.A = p.A + (size_t)done * p.lda;
    q.out0 = p.out0 + (size_t)done * p.ldo;
    if (p.out1) q.out1 = p.out1 + (size_t)done * p.ldo;
    if (p.cmul) q.cmul = p.cmul + (size_t)done * p.ldo;
    if (p.tail) q.tail = p.tail + (size_t)done * p.ldt;
    if (p.colsum) q.colsum = p.colsum + (size_t)groups * 256;
    q.M = (int)(M - done);
    const unsigned grid = nt_grid(M - done, p.N);
    hipLaunchKernelGGL(mlp_gemm_nt<EPI>, dim3(grid), dim3(256), 0, stream, q);
    groups += (int)grid;
  }
  return groups;
}

// ---------------------------------------------------------------------------------------------------------------------------
// Thin reductions (K <= 16) into a 256-wide layer at image size: the first layer's forward (x0 [M,15] -> sin, cos [M,241]) and the
// input gradient of the last sine layer (d x [M,5] -> G' [M,256] * cos).  There is no k-loop to speak of (8 or 4 MFMA steps per
// 32 x 32 tile): these are store-bound passes (537 MB), and what the general kernel loses on them is the shape of its launch --
// 128 x 128 tiles in 2 column halves, k-tiles staged through LDS with barriers.  Here a wave owns 32 whole rows: the weights of
// all 8 column tiles sit in registers, A comes straight from the rows (64 bytes each), the tile goes through the wave's own LDS
// slice for 16-byte stores, no barrier anywhere; many independent waves per CU keep the stores flowing.
// ---------------------------------------------------------------------------------------------------------------------------
template <int EPI, int KP>   // KP: K padded to 8 or 16 (the row stride of A and B covers it; padding columns hold zeros)
__global__ __launch_bounds__(256) void mlp_thin_k_kernel(const NtArgs p, int tiles) {
  __shared__ __align__(16) float scr_all[4][32 * kLd];
  __shared__ __align__(16) float sWt[256 * KP];                // the weights [n][k], zero beyond N / K: 16 KB, read per column tile
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
  float* scr = scr_all[wave];
  constexpr int KS = KP / 2;                                   // MFMA steps (32x32x2)
  for (int i = threadIdx.x; i < 256 * KP; i += 256) {
    const int nn = i / KP, k = i - nn * KP;
    sWt[i] = (nn < p.N && k < p.K) ? p.B[(size_t)nn * p.ldb + k] : 0.f;
  }
  float bn[8];
#pragma unroll
  for (int ni = 0; ni < 8; ++ni) bn[ni] = (EPI != EPI_MULC && ni * 32 + li < p.N) ? p.bias[ni * 32 + li] : 0.f;
  __syncthreads();
  const int t_row = lane >> 3, t_col = (lane & 7) * 4;
  float4 csum[8];
#pragma unroll
  for (int ni = 0; ni < 8; ++ni) csum[ni] = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int tile = blockIdx.x * 4 + wave; tile < tiles; tile += gridDim.x * 4) {
    const int row0 = tile * 32;
    // A: row li, k = 2 s + lh (both halves read the same 4 * KP bytes of the row)
    float av[KS];
    {
      const int m = row0 + li < p.M ? row0 + li : p.M - 1;
      const float4* ap = reinterpret_cast<const float4*>(p.A + (size_t)m * p.lda);
#pragma unroll
      for (int q = 0; q < KP / 4; ++q) {
        const float4 v = ap[q];
        av[2 * q] = lh ? v.y : v.x;
        av[2 * q + 1] = lh ? v.w : v.z;
      }
    }
    const bool full_rows = row0 + 32 <= p.M;
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) {
      if (ni * 32 >= p.N) break;                               // uniform
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      float wb[KS];
#pragma unroll
      for (int q = 0; q < KP / 4; ++q) {                       // row (ni * 32 + li) of the weights, k = 2 s + lh
        const float4 v = *reinterpret_cast<const float4*>(sWt + (ni * 32 + li) * KP + 4 * q);
        wb[2 * q] = lh ? v.y : v.x;
        wb[2 * q + 1] = lh ? v.w : v.z;
      }
#pragma unroll
      for (int sidx = 0; sidx < KS; ++sidx) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[sidx], wb[sidx], acc, 0, 0, 0);
      float4 cv[4];
      const size_t o0 = (size_t)(row0 + t_row) * p.ldo + ni * 32 + t_col;
      if (EPI == EPI_MULC) {
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
          cv[ps] = (full_rows || row0 + t_row + 8 * ps < p.M) ? *reinterpret_cast<const float4*>(p.cmul + o0 + (size_t)(8 * ps) * p.ldo)
                                                            : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      float second[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = acc[r];
        if (EPI == EPI_SINCOS) {
          if (p.out1 == nullptr) v = sin_packed<true>(v + bn[ni]);
          else sincos_cw(v + bn[ni], v, second[r]);
        } else if (EPI == EPI_BIAS) v += bn[ni];
        scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * kLd + li] = v;
      }
      const bool all_cols = ni * 32 + 32 <= p.N || p.tail != nullptr;    // uniform; with a tail the caller rewrites columns N..
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        float4 v = *reinterpret_cast<const float4*>(scr + (t_row + 8 * ps) * kLd + t_col);
        const bool row_ok = full_rows || row0 + t_row + 8 * ps < p.M;
        if (EPI == EPI_MULC) {
          if (p.cmul_sin) cv[ps] = cos_from_packed_sin(cv[ps]);
          v.x *= cv[ps].x; v.y *= cv[ps].y; v.z *= cv[ps].z; v.w *= cv[ps].w;
          if (!row_ok) v = make_float4(0.f, 0.f, 0.f, 0.f);   // a row past the end (its cos operand was not read: a packed zero decodes to cos = 1)
          csum[ni].x += v.x; csum[ni].y += v.y; csum[ni].z += v.z; csum[ni].w += v.w;
        }
        if (row_ok)
          store4_upto(p.out0 + o0 + (size_t)(8 * ps) * p.ldo, v, p.N - (ni * 32 + t_col), all_cols);
      }
      if (EPI == EPI_SINCOS && p.out1 != nullptr) {
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[((r & 3) + 8 * (r >> 2) + 4 * lh) * kLd + li] = second[r];
#pragma unroll
        for (int ps = 0; ps < 4; ++ps)
          if (full_rows || row0 + t_row + 8 * ps < p.M)
            store4_upto(p.out1 + o0 + (size_t)(8 * ps) * p.ldo, *reinterpret_cast<const float4*>(scr + (t_row + 8 * ps) * kLd + t_col),
                        p.N - (ni * 32 + t_col), all_cols);
      }
    }
  }
  if (EPI == EPI_MULC && p.colsum != nullptr) {                // per-workgroup column sums of G' (the bias gradient of the layer below)
    __shared__ float red[32][256];
    __syncthreads();
#pragma unroll
    for (int ni = 0; ni < 8; ++ni) *reinterpret_cast<float4*>(&red[wave * 8 + t_row][ni * 32 + t_col]) = csum[ni];
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int sl = 0; sl < 32; ++sl) t += red[sl][threadIdx.x];
    p.colsum[(size_t)blockIdx.x * 256 + threadIdx.x] = t;
  }
}
constexpr int kThinBlocks = 1024;                            // workgroups of the thin-K kernels (4 per CU)
inline int thin_pad(int K) { return K <= 8 ? 8 : 16; }      // the floats of a row of x the thin-K kernel reads (the weights are read element by element: ldw >= K)
template <int EPI>
int launch_thin_k(const NtArgs& p, hipStream_t stream) {     // returns the number of workgroups (rows of the colsum partials)
  const int tiles = (p.M + 31) / 32;
  const int grid = (tiles + 3) / 4 < kThinBlocks ? (tiles + 3) / 4 : kThinBlocks;
  if (p.K <= 8) hipLaunchKernelGGL((mlp_thin_k_kernel<EPI, 8>), dim3(grid), dim3(256), 0, stream, p, tiles);
  else hipLaunchKernelGGL((mlp_thin_k_kernel<EPI, 16>), dim3(grid), dim3(256), 0, stream, p, tiles);
  return grid;
}

// ---------------------------------------------------------------------------------------------------------------------------
// The skinny ends of the network at image size (M = H*W rows): the output layer ([M,256] x [256,J], J <= 8) with the 'arm' head
// of the residual coordinate MLP, the output layer's weight gradient and the first layer's (15 inputs).  One streaming pass over
// the 256-wide matrix each (268 MB at 512 x 512): HBM-bound, VALU only.
// ---------------------------------------------------------------------------------------------------------------------------
// out[m][j] = bias[j] + sum_k X[m][k] W[j][k].  A wave owns 64 consecutive rows: lane l holds columns 4l..4l+3 of the weights
// and of the row being multiplied (one whole 1 KB row per wave load -- 128-byte segments of many rows run at half the bandwidth),
// J DPP tree sums per row leave the totals in lane 63, from where they are handed to lane (row & 63): after 64 rows every lane
// holds the J sums of ITS row and the head runs once with all lanes busy.  Four rows in flight per wave.
template <int J, bool HEAD>
__global__ __launch_bounds__(256) void mlp_skinny_nt_kernel(const float* __restrict__ X, int ldx, const float* __restrict__ W, int ldw,
                                                            const float* __restrict__ bias, float* __restrict__ out, int ldo, long M, int K,
                                                            const ArmHead h) {
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), waves = (long)gridDim.x * 4;
  const int k0 = 4 * lane;
  const bool in = k0 < K;                                     // K is a multiple of 4
  float4 w[J];
#pragma unroll
  for (int j = 0; j < J; ++j) w[j] = in ? *reinterpret_cast<const float4*>(W + (size_t)j * ldw + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
  const long tiles = (M + 63) / 64;
  constexpr int R = 4;
  for (long tile = wave; tile < tiles; tile += waves) {
    const long row0 = tile * 64;
    float mine[J];
#pragma unroll
    for (int j = 0; j < J; ++j) mine[j] = 0.f;
    float4 x[R];
    auto load_rows = [&](int g) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        long m = row0 + g * R + r;
        m = m < M ? m : M - 1;
        x[r] = in ? *reinterpret_cast<const float4*>(X + m * ldx + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    load_rows(0);
    for (int g = 0; g < 64 / R; ++g) {
      float4 cur[R];
#pragma unroll
      for (int r = 0; r < R; ++r) cur[r] = x[r];
      if (g + 1 < 64 / R) load_rows(g + 1);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int row = g * R + r;                            // uniform
#pragma unroll
        for (int j = 0; j < J; ++j) {
          const float t = wave_sum_lane63(fmaf(cur[r].x, w[j].x, fmaf(cur[r].y, w[j].y, fmaf(cur[r].z, w[j].z, cur[r].w * w[j].w))));
          const float total = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, t), 63));
          mine[j] = lane == row ? total : mine[j];
        }
      }
    }
    const long m = row0 + lane;
    if (m < M) {
      float v[J];
#pragma unroll
      for (int j = 0; j < J; ++j) v[j] = mine[j] + bias[j];
      if (out != nullptr) {
#pragma unroll
        for (int j = 0; j < J; ++j) out[m * ldo + j] = v[j];
      }
      if (HEAD) arm_head_store(h, m, v);
    }
  }
}

template <int J, bool HEAD>
void launch_skinny_nt(const float* x, int ldx, const float* w, int ldw, const float* bias, float* out, int ldo, long M, int K, const ArmHead& h,
                      hipStream_t stream) {
  const long tiles = (M + 63) / 64;
  const unsigned grid = (unsigned)((tiles + 3) / 4 < 2048 ? (tiles + 3) / 4 : 2048);
  hipLaunchKernelGGL((mlp_skinny_nt_kernel<J, HEAD>), dim3(grid), dim3(256), 0, stream, x, ldx, w, ldw, bias, out, ldo, M, K, h);
}

// d x[m][j] = g_y[j] * 1.3 * (1 - tanh(x)^2) for the live channels (g_y = d maps; the roughness map is 0.93 y + 0.07; the clamp
// of :235 is a straight-through one), 0 for the others and for the padding columns 5..7.
__global__ __launch_bounds__(256) void arm_head_bwd_kernel(const float* __restrict__ g_a, const float* __restrict__ g_r, const float* __restrict__ g_m,
                                                           const float* __restrict__ th, float* __restrict__ d_x, long M) {
  for (long m = (long)blockIdx.x * 256 + threadIdx.x; m < M; m += (long)gridDim.x * 256) {
    const float4 t0 = *reinterpret_cast<const float4*>(th + m * 8);
    const float t4 = th[m * 8 + 4];
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    float o4 = 0.f;
    if (g_a) {
      o.x = (g_a[m * 3 + 0] * 1.3f) * (1.f - t0.x * t0.x);
      o.y = (g_a[m * 3 + 1] * 1.3f) * (1.f - t0.y * t0.y);
      o.z = (g_a[m * 3 + 2] * 1.3f) * (1.f - t0.z * t0.z);
    }
    if (g_r) o.w = ((g_r[m] * 0.93f) * 1.3f) * (1.f - t0.w * t0.w);
    if (g_m) o4 = (g_m[m] * 1.3f) * (1.f - t4 * t4);
    *reinterpret_cast<float4*>(d_x + m * 8) = o;
    *reinterpret_cast<float4*>(d_x + m * 8 + 4) = make_float4(o4, 0.f, 0.f, 0.f);
  }
}

// partial[slab][j][c] = sum over the slab's rows of S[m][j] B[m][c] (J columns of the skinny S, 256 of the wide B), and
// bpart[slab][j] = sum of S[m][j].  Thread (cq, rs): columns 4cq..4cq+3 of B, every 4th row of the slab; the S row is uniform
// across a wave (scalar loads).  The four row slices are folded through LDS in fixed order.
constexpr int kSkinnySlabs = 1024;
constexpr int kOutJ = 5;                                   // outputs of the 'arm' network's last layer (mymodels/mlps.py:233-236)
struct SkinnyDgrad {          // the input-gradient half of mlp_skinny_tn_kernel<8, true>
  const float* W;             // [Jv][ldw] the layer's forward weight (row j = output j, 256 inputs)
  int ldw, Jv;
  const float* cmul;          // nullable [M][ldb] cos(pre); null: B holds the sign-packed sines
  float* G;                   // [M][ldg]
  int ldg;
  float* gsum_part;           // [slabs][256]
  unsigned* tmax;             // nullable [M / 128]: atomic max of |G| per 128-row tile (zeroed by the caller): the block exponents of the f16 products
};
template <int J, bool DGRAD = false, int JV = kOutJ>      // JV (DGRAD): the valid columns of S -- 5 ('arm'), 8 ('armn': round 6)
__global__ __launch_bounds__(256) void mlp_skinny_tn_kernel(const float* __restrict__ S, int lds, const float* __restrict__ B, int ldb,
                                                            float* __restrict__ partial, float* __restrict__ bpart, long M, long rows_per_slab,
                                                            const SkinnyDgrad dg) {
  __shared__ float red[3][8][256];                         // the fold runs 8 columns of S at a time (24 KB: four workgroups per CU)
  __shared__ float bred[4][J];
  const int cq = threadIdx.x & 63, rs = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long m_begin = (long)blockIdx.x * rows_per_slab;
  const long m_end = (m_begin + rows_per_slab < M) ? m_begin + rows_per_slab : M;
  float4 acc[J];
  float bs[J];
#pragma unroll
  for (int j = 0; j < J; ++j) { acc[j] = make_float4(0.f, 0.f, 0.f, 0.f); bs[j] = 0.f; }
  // DGRAD (the output layer of the 'arm' network, J = 8, 5 valid): the same pass over the wide matrix B = sines of the last sine layer also
  // forms that layer's pre-activation gradient G[m][n] = (sum_j S[m][j] W[j][n]) cos(pre[m][n]) and its column sums -- a K = 5 product
  // is five FMAs per element, and the separate input-gradient pass read the 268 MB of sines a second time
  float4 wv[DGRAD ? JV : 1], gsum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (DGRAD) {
#pragma unroll
    for (int j = 0; j < JV; ++j) wv[j] = j < dg.Jv ? *reinterpret_cast<const float4*>(dg.W + (size_t)j * dg.ldw + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  constexpr int U = 4;
  float tmx = 0.f;                                          // largest |G| of the rows since the last tile boundary (this wave's rows)
  long tmx_tile = m_begin / kBM;
  auto tmx_flush = [&]() {
    if (DGRAD && dg.tmax != nullptr) {
      const float wmx = wave_max_lane63(tmx);
      if (cq == 63) atomicMax(dg.tmax + tmx_tile, __float_as_uint(wmx));
    }
    tmx = 0.f;
  };
  for (long m0 = m_begin + rs; m0 < m_end; m0 += 4 * U) {
    float4 b[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long m = m0 + 4 * u;
      b[u] = m < m_end ? *reinterpret_cast<const float4*>(B + m * ldb + 4 * cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long m = m0 + 4 * u < m_end ? m0 + 4 * u : m_end - 1;   // past the end: b is zero, any finite row of S will do
      const float* srow = S + m * lds;
      const bool live = m0 + 4 * u < m_end;
      float4 dot = make_float4(0.f, 0.f, 0.f, 0.f);
      constexpr int JA = DGRAD ? JV : J;                     // DGRAD: at most JV columns of S are valid (the others are padding: their sums stay zero)
#pragma unroll
      for (int j = 0; j < JA; ++j) {
        const float sv = srow[j];
        acc[j].x = fmaf(sv, b[u].x, acc[j].x);
        acc[j].y = fmaf(sv, b[u].y, acc[j].y);
        acc[j].z = fmaf(sv, b[u].z, acc[j].z);
        acc[j].w = fmaf(sv, b[u].w, acc[j].w);
        bs[j] += live ? sv : 0.f;
        if (DGRAD && j < JV) {
          dot.x = fmaf(sv, wv[j].x, dot.x); dot.y = fmaf(sv, wv[j].y, dot.y); dot.z = fmaf(sv, wv[j].z, dot.z); dot.w = fmaf(sv, wv[j].w, dot.w);
        }
      }
      if (DGRAD && live) {
        const float4 c = dg.cmul ? *reinterpret_cast<const float4*>(dg.cmul + m * ldb + 4 * cq) : cos_from_packed_sin(b[u]);
        const float4 gv = make_float4(dot.x * c.x, dot.y * c.y, dot.z * c.z, dot.w * c.w);
        *reinterpret_cast<float4*>(dg.G + m * dg.ldg + 4 * cq) = gv;
        gsum.x += gv.x; gsum.y += gv.y; gsum.z += gv.z; gsum.w += gv.w;
        if (m / kBM != tmx_tile) {                            // uniform over the wave (m is): a new tile begins
          tmx_flush();
          tmx_tile = m / kBM;
        }
        tmx = __builtin_fmaxf(__builtin_fmaxf(tmx, __builtin_fmaxf(__builtin_fabsf(gv.x), __builtin_fabsf(gv.y))), __builtin_fmaxf(__builtin_fabsf(gv.z), __builtin_fabsf(gv.w)));
      }
    }
  }
  if (DGRAD) tmx_flush();
  if (DGRAD) {                                             // column sums of G over the slab: the four row-waves in fixed order
    if (rs > 0) *reinterpret_cast<float4*>(&red[rs - 1][0][4 * cq]) = gsum;
    __syncthreads();
    if (rs == 0) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        const float4 o = *reinterpret_cast<const float4*>(&red[q][0][4 * cq]);
        gsum.x += o.x; gsum.y += o.y; gsum.z += o.z; gsum.w += o.w;
      }
      *reinterpret_cast<float4*>(dg.gsum_part + (size_t)blockIdx.x * 256 + 4 * cq) = gsum;
    }
    __syncthreads();
  }
  if (cq == 0) {
#pragma unroll
    for (int j = 0; j < J; ++j) bred[rs][j] = bs[j];
  }
#pragma unroll
  for (int j0 = 0; j0 < J; j0 += 8) {
    if (j0 > 0) __syncthreads();                          // the previous eight have been folded
    if (rs > 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(&red[rs - 1][j][4 * cq]) = acc[j0 + j];
    }
    __syncthreads();
    if (rs == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float4 t = acc[j0 + j];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const float4 o = *reinterpret_cast<const float4*>(&red[q][j][4 * cq]);
          t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
        }
        *reinterpret_cast<float4*>(partial + ((size_t)blockIdx.x * J + j0 + j) * 256 + 4 * cq) = t;
      }
    }
  }
  if (rs == 0 && bpart != nullptr && threadIdx.x < J)
    bpart[(size_t)blockIdx.x * J + threadIdx.x] = (bred[0][threadIdx.x] + bred[1][threadIdx.x]) + (bred[2][threadIdx.x] + bred[3][threadIdx.x]);
}

// out[j * ld_j + c * ld_c] = sum over slabs of partial[slab][j][c] (j < Jv, c < C); d_b[j] = sum of bpart[slab][j].  One
// workgroup per (j, 64 columns): 16 slab slices x 64 columns, fixed order.
__device__ __forceinline__ void skinny_reduce_body(float* red_, int block, const float* __restrict__ partial, const float* __restrict__ bpart, int slabs, int J, int Jv,
                                                             int C, float* __restrict__ out, long ld_j, long ld_c, float* __restrict__ d_b,
                                                             const float* __restrict__ gsum_part, int C2, float* __restrict__ d_b2) {
  float (*red)[64] = reinterpret_cast<float (*)[64]>(red_);
  int j = block >> 2;
  const int cl = threadIdx.x & 63, c = (block & 3) * 64 + cl, sl = threadIdx.x >> 6;
  if (j == J) {                                              // the extra workgroups: column sums of the fused input gradient -> its bias gradient
    partial = gsum_part; J = 1; j = 0; Jv = 1; C = C2; out = d_b2; ld_j = 0; ld_c = 1; d_b = nullptr;
  }
  if (j >= Jv) return;                                       // uniform per workgroup
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int q = sl;
  for (; q + 48 < slabs; q += 64) {
    a0 += partial[((size_t)q * J + j) * 256 + c];
    a1 += partial[((size_t)(q + 16) * J + j) * 256 + c];
    a2 += partial[((size_t)(q + 32) * J + j) * 256 + c];
    a3 += partial[((size_t)(q + 48) * J + j) * 256 + c];
  }
  for (; q < slabs; q += 16) a0 += partial[((size_t)q * J + j) * 256 + c];
  red[sl][cl] = (a0 + a1) + (a2 + a3);
  __syncthreads();
  if (sl == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) t += red[u][cl];
    out[j * ld_j + c * ld_c] = t;
  }
  if (d_b != nullptr && bpart != nullptr && (block & 3) == 0) {   // the bias gradient: 1024 threads over the slabs, LDS tree
    __syncthreads();
    float t = 0.f;
    for (int q2 = threadIdx.x; q2 < slabs; q2 += 1024) t += bpart[(size_t)q2 * J + j];
    red[sl][cl] = t;
    __syncthreads();
    if (sl == 0) {
      float u = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) u += red[r][cl];
      red[0][cl] = u;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      float u = 0.f;
      for (int r = 0; r < 64; ++r) u += red[0][r];
      d_b[j] = u;
    }
  }
}
__global__ __launch_bounds__(1024) void mlp_skinny_tn_reduce(const float* __restrict__ partial, const float* __restrict__ bpart, int slabs, int J, int Jv,
                                                             int C, float* __restrict__ out, long ld_j, long ld_c, float* __restrict__ d_b,
                                                             const float* __restrict__ gsum_part, int C2, float* __restrict__ d_b2) {
  __shared__ float red[1024];
  skinny_reduce_body(red, (int)blockIdx.x, partial, bpart, slabs, J, Jv, C, out, ld_j, ld_c, d_b, gsum_part, C2, d_b2);
}

// Every fold of an iteration's backward pass in ONE launch (matpbr_mlp_reduce_jobs): workgroup -> job by the running count of workgroups
constexpr int kMaxReduceJobs = 16;
struct ReduceJobs {
  MatpbrReduceJob j[kMaxReduceJobs];
  int first[kMaxReduceJobs + 1];
  int n;
};
__global__ __launch_bounds__(1024) void mlp_reduce_jobs_kernel(const ReduceJobs js) {
  __shared__ float red[1024];
  int k = 0;
  while (k + 1 < js.n && (int)blockIdx.x >= js.first[k + 1]) ++k;          // (uniform)
  const MatpbrReduceJob& q = js.j[k];
  const int block = (int)blockIdx.x - js.first[k];
  if (q.kind == MATPBR_REDUCE_WGRAD) wgrad_reduce_body(red, q.src, q.groups, q.dst, q.n0, q.n1, q.n2, block);
  else if (q.kind == MATPBR_REDUCE_COLSUM) colsum_reduce_body(red, q.src, q.groups, q.dst, block);
  else skinny_reduce_body(red, block, q.src, q.src_b, q.groups, q.n0, q.n1, q.n2, q.dst, q.ld_j, q.ld_c, q.dst_b, q.src_g, q.n3, q.dst_g);
}
inline int reduce_job_blocks(const MatpbrReduceJob& q) {
  if (q.kind == MATPBR_REDUCE_WGRAD) return 256;
  if (q.kind == MATPBR_REDUCE_COLSUM) return q.n0;
  return q.n0 * 4 + (q.dst_g ? 4 : 0);
}
inline void launch_reduce_job(const MatpbrReduceJob& q, hipStream_t st) {          // the job by itself, behind its producer
  if (q.kind == MATPBR_REDUCE_WGRAD) hipLaunchKernelGGL(mlp_wgrad_reduce, dim3(256), dim3(1024), 0, st, q.src, q.groups, q.dst, q.n0, q.n1, q.n2);
  else if (q.kind == MATPBR_REDUCE_COLSUM) hipLaunchKernelGGL(mlp_colsum_reduce, dim3(q.n0), dim3(256), 0, st, q.src, q.groups, q.dst);
  else hipLaunchKernelGGL(mlp_skinny_tn_reduce, dim3(reduce_job_blocks(q)), dim3(1024), 0, st, q.src, q.src_b, q.groups, q.n0, q.n1, q.n2, q.dst, q.ld_j, q.ld_c,
                          q.dst_b, q.src_g, q.n3, q.dst_g);
}
// reduce now, or leave the record to the caller (matpbr_mlp_reduce_jobs)
inline void reduce_or_defer(const MatpbrReduceJob& q, MatpbrReduceJob* defer, hipStream_t st) {
  if (defer) *defer = q;
  else launch_reduce_job(q, st);
}

}  // namespace

extern "C" {

int matpbr_mlp_set_lds_dma(int mode) {
  g_nt_w0_gx.store(mode == 3 ? 1 : 0, std::memory_order_relaxed);
  const int was = g_nt_gl.exchange(mode < 0 ? 0 : (mode > 2 ? 2 : mode), std::memory_order_relaxed);
  return was;
}

int matpbr_mlp_layer_fwd_tail(const float* x, int ldx, const float* w, int ldw, const float* bias, float* s_out, float* c_out, int ldo,
                              const float* tail, int ldt, long M, int N, int K, void* stream) {
  if (!x || !w || !bias || !s_out || M <= 0 || M > 0x7fffff00L || N <= 0 || N > 256 || K <= 0 || K > 256) return MATPBR_ERR_INVALID_ARG;
  if ((ldx & 3) || (ldw & 3) || ldx < ((K + 3) & ~3) || ldw < ((K + 3) & ~3) || ldo < N || !aligned16(x) || !aligned16(w))
    return MATPBR_ERR_INVALID_ARG;
  if (tail && (ldo < 256 || ldt < 256 - N || !c_out)) return MATPBR_ERR_INVALID_ARG;   // the tail fills columns N..255 of a 256-wide sine layer
  NtArgs p{x, w, bias, nullptr, s_out, c_out, nullptr, 0, N, K, ldx, ldw, ldo};
  p.tail = tail; p.ldt = ldt;
  const bool thin = M > kSmallM && K <= 16 && ldx >= thin_pad(K) && ldw >= ((K + 3) & ~3) && ldo >= 256 && !(ldo & 3) && aligned16(s_out) &&
                    (!c_out || aligned16(c_out));
  if (M <= kSmallM) {
    if (c_out) launch_small_nt<EPI_SINCOS>(p, M, (hipStream_t)stream);
    else launch_small_nt<EPI_BIAS>(p, M, (hipStream_t)stream);
  } else if (thin) {
    p.M = (int)M;
    if (c_out) launch_thin_k<EPI_SINCOS>(p, (hipStream_t)stream);
    else launch_thin_k<EPI_BIAS>(p, (hipStream_t)stream);
  } else if (c_out)
    launch_nt<EPI_SINCOS>(p, M, (hipStream_t)stream);
  else
    launch_nt<EPI_BIAS>(p, M, (hipStream_t)stream);
  if (tail && N < 256)
    hipLaunchKernelGGL(mlp_tail_copy_kernel, dim3((unsigned)((M * (256 - N) + 255) / 256 < 2048 ? (M * (256 - N) + 255) / 256 : 2048)), dim3(256), 0,
                       (hipStream_t)stream, s_out, ldo, tail, ldt, M, N);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

int matpbr_mlp_layer_fwd_sgn(const float* x, int ldx, const float* w, int ldw, const float* bias, float* s_out, int ldo, const float* tail, int ldt,
                             long M, int N, int K, void* stream) {
  if (!x || !w || !bias || !s_out || M <= 0 || M > 0x7fffff00L || N <= 0 || N > 256 || K <= 0 || K > 16) return MATPBR_ERR_INVALID_ARG;
  // the thin-K kernel only (the first layer of the coordinate MLP at image size: 15 inputs, 10 for 'armn'): every other shape keeps its cosines
  if (M <= kSmallM || (ldx & 3) || (ldw & 3) || ldx < thin_pad(K) || ldw < ((K + 3) & ~3) || ldo < 256 || (ldo & 3) || !aligned16(x) || !aligned16(w) ||
      !aligned16(s_out))
    return MATPBR_ERR_UNSUPPORTED;
  if (tail && ldt < 256 - N) return MATPBR_ERR_INVALID_ARG;
  NtArgs p{x, w, bias, nullptr, s_out, nullptr, nullptr, (int)M, N, K, ldx, ldw, ldo};
  p.tail = tail; p.ldt = ldt;
  launch_thin_k<EPI_SINCOS>(p, (hipStream_t)stream);
  if (tail && N < 256)
    hipLaunchKernelGGL(mlp_tail_copy_kernel, dim3((unsigned)((M * (256 - N) + 255) / 256 < 2048 ? (M * (256 - N) + 255) / 256 : 2048)), dim3(256), 0,
                       (hipStream_t)stream, s_out, ldo, tail, ldt, M, N);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

int matpbr_mlp_layer_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias, float* s_out, float* c_out, int ldo, long M,
                         int N, int K, void* stream) {
  return matpbr_mlp_layer_fwd_tail(x, ldx, w, ldw, bias, s_out, c_out, ldo, nullptr, 0, M, N, K, stream);
}

size_t matpbr_mlp_bwd_input_workspace_bytes(long M) {
  (void)M;
  return (size_t)((kPersistent + 16) > kThinBlocks ? (kPersistent + 16) : kThinBlocks) * 256 * sizeof(float);
}

static int mlp_layer_bwd_input_impl(const float* g, int ldg, const float* wt, int ldwt, const float* c_prev, float* g_prev, int ldo,
                                    float* d_bias_prev, void* workspace, size_t workspace_bytes, long M, int n_prev, int n_red, int sgn, void* stream) {
  if (!g || !wt || !c_prev || !g_prev || M <= 0 || M > 0x7fffff00L || n_prev <= 0 || n_prev > 256 || n_red <= 0 || n_red > 256)
    return MATPBR_ERR_INVALID_ARG;
  if ((ldg & 3) || (ldwt & 3) || ldg < ((n_red + 3) & ~3) || ldwt < ((n_red + 3) & ~3) || ldo < n_prev || !aligned16(g) || !aligned16(wt))
    return MATPBR_ERR_INVALID_ARG;
  if (d_bias_prev && (!workspace || workspace_bytes < matpbr_mlp_bwd_input_workspace_bytes(M))) return MATPBR_ERR_WORKSPACE;
  NtArgs p{g, wt, nullptr, c_prev, g_prev, nullptr, d_bias_prev ? (float*)workspace : nullptr, 0, n_prev, n_red, ldg, ldwt, ldo};
  p.cmul_sin = sgn;
  const bool thin = M > kSmallM && n_red <= 16 && ldg >= (n_red <= 8 ? 8 : 16) && ldo >= 256 && !(ldo & 3) && aligned16(c_prev) && aligned16(g_prev);
  if (sgn && !thin) return MATPBR_ERR_UNSUPPORTED;
  int groups;
  if (thin) {
    p.M = (int)M;
    groups = launch_thin_k<EPI_MULC>(p, (hipStream_t)stream);
  } else {
    groups = M <= kSmallM ? launch_small_nt<EPI_MULC>(p, M, (hipStream_t)stream) : launch_nt<EPI_MULC>(p, M, (hipStream_t)stream);
  }
  if (d_bias_prev)
    hipLaunchKernelGGL(mlp_colsum_reduce, dim3(n_prev), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, groups, d_bias_prev);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

int matpbr_mlp_layer_bwd_input(const float* g, int ldg, const float* wt, int ldwt, const float* c_prev, float* g_prev, int ldo,
                               float* d_bias_prev, void* workspace, size_t workspace_bytes, long M, int n_prev, int n_red, void* stream) {
  return mlp_layer_bwd_input_impl(g, ldg, wt, ldwt, c_prev, g_prev, ldo, d_bias_prev, workspace, workspace_bytes, M, n_prev, n_red, 0, stream);
}
int matpbr_mlp_layer_bwd_input_sgn(const float* g, int ldg, const float* wt, int ldwt, const float* s_prev, float* g_prev, int ldo,
                                   float* d_bias_prev, void* workspace, size_t workspace_bytes, long M, int n_prev, int n_red, void* stream) {
  return mlp_layer_bwd_input_impl(g, ldg, wt, ldwt, s_prev, g_prev, ldo, d_bias_prev, workspace, workspace_bytes, M, n_prev, n_red, 1, stream);
}

int matpbr_mlp_small_bwd_step(const float* g, int ldg, const float* w, int ldw, const float* c_prev, float* g_prev, int ldo, float* colsum_out,
                              int n_prev, const float* x, int ldx, float* d_w, int ldw_out, int K, const float* colsum_in, int colsum_stride,
                              int groups_in, float* d_bias, long M, int n_red, void* stream) {
  if (!g || !x || !d_w || M <= 0 || M > kSmallM || n_red <= 0 || n_red > 256 || K <= 0 || K > 256) return MATPBR_ERR_INVALID_ARG;
  if ((ldg & 3) || (ldx & 3) || ldg < ((n_red + 3) & ~3) || ldx < ((K + 3) & ~3) || ldw_out < K || !aligned16(g) || !aligned16(x)) return MATPBR_ERR_INVALID_ARG;
  SmallBwdStep a{};
  if (w) {
    if (!c_prev || !g_prev || n_prev <= 0 || n_prev > 256 || ldw < n_prev || ldo < n_prev) return MATPBR_ERR_INVALID_ARG;
    a.d = NtArgs{g, w, nullptr, c_prev, g_prev, nullptr, colsum_out, (int)M, n_prev, n_red, ldg, ldw, ldo};
    a.nD = (int)((M + 31) / 32) * ((n_prev + 31) / 32);
  }
  a.G = g; a.ldg = ldg; a.X = x; a.ldx = ldx; a.dW = d_w; a.ldw = ldw_out; a.M = (int)M; a.N = n_red; a.K = K;
  a.nW = ((n_red + 31) / 32) * ((K + 31) / 32);
  int nB = 0;
  if (d_bias) {
    if (!colsum_in || groups_in <= 0 || colsum_stride < n_red) return MATPBR_ERR_INVALID_ARG;
    a.part_in = colsum_in; a.part_stride = colsum_stride; a.groups_in = groups_in; a.d_bias = d_bias; a.n_bias = n_red;
    nB = (n_red + 255) / 256;
  }
  hipLaunchKernelGGL(mlp_small_bwd_step_kernel, dim3((unsigned)(a.nD + a.nW + nB)), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

size_t matpbr_mlp_wsplit_bytes(int K) { return K > 0 ? (size_t)((K + 31) / 32) * 2 * 3 * 256 * 2 * sizeof(uint4) : 0; }

static int split_weights_one(const float* w, int ldw, int N, int K, int flags, void* wsplit, void* stream) {
  if (!w || !wsplit || N <= 0 || N > 256 || K <= 0 || K > 256 || ldw < ((flags & 1) ? N : K)) return MATPBR_ERR_INVALID_ARG;
  const int n = ((K + 31) / 32) * 2 * 256 * 2;
  hipLaunchKernelGGL(mlp_split_weights_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, ldw, N, K, flags, (uint4*)wsplit);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}
int matpbr_mlp_split_weights(const float* w, int ldw, int N, int K, void* wsplit, void* stream) { return split_weights_one(w, ldw, N, K, 0, wsplit, stream); }
int matpbr_mlp_split_weights_fmt(const float* w, int ldw, int N, int K, int flags, void* wsplit, void* stream) {
  if (flags & ~(MATPBR_WSPLIT_TRANSPOSED | MATPBR_WSPLIT_F16X2)) return MATPBR_ERR_INVALID_ARG;
  return split_weights_one(w, ldw, N, K, flags, wsplit, stream);
}

int matpbr_mlp_split_weights_multi(const float* const* w, const int* ldw, const int* N, const int* K, const int* transposed, void* const* wsplit,
                                   int n_jobs, void* stream) {
  if (!w || !ldw || !N || !K || !transposed || !wsplit || n_jobs <= 0 || n_jobs > 8) return MATPBR_ERR_INVALID_ARG;
  SplitJobs jobs{};
  int kmax = 0;
  for (int j = 0; j < n_jobs; ++j) {
    if (!w[j] || !wsplit[j] || N[j] <= 0 || N[j] > 256 || K[j] <= 0 || K[j] > 256 || (transposed[j] & ~3) || ldw[j] < ((transposed[j] & 1) ? N[j] : K[j])) return MATPBR_ERR_INVALID_ARG;
    jobs.w[j] = w[j]; jobs.out[j] = (uint4*)wsplit[j]; jobs.ldw[j] = ldw[j]; jobs.N[j] = N[j]; jobs.K[j] = K[j]; jobs.transposed[j] = transposed[j];
    kmax = K[j] > kmax ? K[j] : kmax;
  }
  const int n = ((kmax + 31) / 32) * 2 * 256 * 2;
  hipLaunchKernelGGL(mlp_split_weights_multi_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)n_jobs), dim3(256), 0, (hipStream_t)stream, jobs);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

int matpbr_mlp_layer_fwd_bx_tail(const float* x, int ldx, const void* wsplit, const float* bias, float* s_out, float* c_out, int ldo,
                                 const float* tail, int ldt, long M, int N, int K, int nprod, void* stream) {
  // c_out == NULL: the sines carry the sign of their cosine in their last mantissa bit and no cosines are written (matpbr_mlp_layer_fwd_bx_sgn)
  if (!x || !wsplit || !bias || !s_out || M <= 0 || N <= 0 || N > 256 || K <= 0 || K > 256) return MATPBR_ERR_INVALID_ARG;
  if ((nprod != 3 && nprod != 6 && nprod != 9) || (M % kBM) || M > 0x7fffff00L || ldo < 256 || (ldo & 3) || (ldx & 3) || ldx < ((K + 31) & ~31) || !aligned16(x) ||
      !aligned16(s_out) || (c_out && !aligned16(c_out)))
    return MATPBR_ERR_UNSUPPORTED;
  if (nprod == 3 && (K <= 32 || (N < 256 && !tail))) return MATPBR_ERR_UNSUPPORTED;   // mlp_nt_gx: whole 16-byte stores over a skip layer's tail
  if (tail && ldt < 256 - N) return MATPBR_ERR_INVALID_ARG;
  NtArgs p{x, nullptr, bias, nullptr, s_out, c_out, nullptr, (int)M, N, K, ldx, 0, ldo};
  p.tail = tail; p.ldt = ldt;
  if (launch_nt_bx<EPI_SINCOS>(p, (const uint4*)wsplit, nprod, (hipStream_t)stream) < 0) return MATPBR_ERR_LAUNCH;
  if (tail && N < 256 && !((nprod == 3 || gx_ok(p)) && N > 224))     // mlp_nt_gx writes the tail in its epilogue (it sits in the last 32-column block)
    hipLaunchKernelGGL(mlp_tail_copy_kernel, dim3((unsigned)((M * (256 - N) + 255) / 256 < 2048 ? (M * (256 - N) + 255) / 256 : 2048)), dim3(256), 0,
                       (hipStream_t)stream, s_out, ldo, tail, ldt, M, N);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

int matpbr_mlp_layer_fwd_bx_head(const float* x, int ldx, const void* wsplit, const float* bias, float* s_out, float* c_out, int ldo,
                                 const float* w_out, int ldw_out, const float* bias_out, const float* start, int lds, float* th, float* map_a,
                                 float* map_r, float* map_m, long M, int K, int nprod, void* stream) {
  if (!x || !wsplit || !bias || !s_out || !w_out || !bias_out || !start || !th || M <= 0 || K <= 0 || K > 256 || lds < 5 || ldw_out < 256)
    return MATPBR_ERR_INVALID_ARG;
  if ((nprod != 3 && nprod != 6 && nprod != 9) || (M % kBM) || M > 0x7fffff00L || ldo < 256 || (ldo & 3) || (ldx & 3) || ldx < ((K + 31) & ~31) || !aligned16(x) ||
      !aligned16(s_out) || (c_out && !aligned16(c_out)))
    return MATPBR_ERR_UNSUPPORTED;
  NtArgs p{x, nullptr, bias, nullptr, s_out, c_out, nullptr, (int)M, 256, K, ldx, 0, ldo};
  const HeadArgs hd{w_out, ldw_out, bias_out, ArmHead{start, lds, th, map_a, map_r, map_m}};
  const int tiles = (int)(M / kBM);
  const unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
  if (nprod == 3) {                          // two f16 pieces: the packed-sine form of mlp_nt_gx
    if (c_out != nullptr || K <= 32 || (long)kBM * ldx * 4 >= (1l << 31)) return MATPBR_ERR_UNSUPPORTED;
    const unsigned grid2 = (unsigned)(tiles < 512 ? tiles : 512);
    const bool ok3 = launch_nt_gx<EPI_SINCOS, 3, true>(p, (const uint4*)wsplit, hd, grid2, (hipStream_t)stream);
    return ok3 && hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
  }
  if (gx_ok(p) && c_out == nullptr) {        // with stored cosines the two-workgroup form spills: the 512-thread kernel
    const unsigned grid2 = (unsigned)(tiles < 512 ? tiles : 512);
    const bool ok2 = nprod == 9 ? launch_nt_gx<EPI_SINCOS, 9, true>(p, (const uint4*)wsplit, hd, grid2, (hipStream_t)stream)
                                : launch_nt_gx<EPI_SINCOS, 6, true>(p, (const uint4*)wsplit, hd, grid2, (hipStream_t)stream);
    return ok2 && hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
  }
  const bool ok = nprod == 9 ? launch_nt_bx_head<9>(p, (const uint4*)wsplit, hd, grid, (hipStream_t)stream)
                             : launch_nt_bx_head<6>(p, (const uint4*)wsplit, hd, grid, (hipStream_t)stream);
  return ok && hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

int matpbr_mlp_layer_fwd_bx(const float* x, int ldx, const void* wsplit, const float* bias, float* s_out, float* c_out, int ldo, long M, int N,
                            int K, int nprod, void* stream) {
  return matpbr_mlp_layer_fwd_bx_tail(x, ldx, wsplit, bias, s_out, c_out, ldo, nullptr, 0, M, N, K, nprod, stream);
}

static int mlp_layer_bwd_input_bx_impl(const float* g, int ldg, const void* wtsplit, const float* c_prev, float* g_prev, int ldo, float* d_bias_prev,
                                       void* workspace, size_t workspace_bytes, long M, int n_prev, int n_red, int nprod, int sgn, void* stream,
                                       const unsigned* g_tile_max = nullptr, unsigned* out_tile_max = nullptr, MatpbrReduceJob* defer = nullptr) {
  if (!g || !wtsplit || !c_prev || !g_prev || M <= 0 || n_prev <= 0 || n_prev > 256 || n_red <= 0 || n_red > 256) return MATPBR_ERR_INVALID_ARG;
  if (nprod == 3 && (!g_tile_max || n_red <= 32)) return MATPBR_ERR_INVALID_ARG;
  if ((nprod != 3 && nprod != 6 && nprod != 9) || (M % kBM) || M > 0x7fffff00L || ldo < 256 || (ldo & 3) || (ldg & 3) || ldg < ((n_red + 31) & ~31) || !aligned16(g) ||
      !aligned16(g_prev) || !aligned16(c_prev))
    return MATPBR_ERR_UNSUPPORTED;
  if (d_bias_prev && (!workspace || workspace_bytes < matpbr_mlp_bwd_input_workspace_bytes(M))) return MATPBR_ERR_WORKSPACE;
  NtArgs p{g, nullptr, nullptr, c_prev, g_prev, nullptr, d_bias_prev ? (float*)workspace : nullptr, (int)M, n_prev, n_red, ldg, 0, ldo};
  p.cmul_sin = sgn;
  p.a_tmax = g_tile_max; p.o_tmax = out_tile_max;
  const int groups = launch_nt_bx<EPI_MULC>(p, (const uint4*)wtsplit, nprod, (hipStream_t)stream);
  if (groups < 0) return MATPBR_ERR_LAUNCH;
  if (defer) defer->kind = MATPBR_REDUCE_NONE;
  if (d_bias_prev) {
    MatpbrReduceJob q{};
    q.kind = MATPBR_REDUCE_COLSUM; q.groups = groups; q.src = (const float*)workspace; q.dst = d_bias_prev; q.n0 = n_prev;
    reduce_or_defer(q, defer, (hipStream_t)stream);
  }
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

static int mlp_first_layer_bwd_impl(const float* g, int ldg, const void* wtsplit, const float* c_prev, int ldc, int sgn, const float* x0, int ldx0,
                                    float* d_w0, long ld_j, long ld_c, int d0, float* d_bias0, void* workspace, size_t workspace_bytes, void* workspace2,
                                    size_t workspace2_bytes, long M, int n0, int n_red, int nprod, const unsigned* g_tile_max, void* stream,
                                    MatpbrReduceJob* defer = nullptr) {
  if (!g || !wtsplit || !c_prev || !x0 || !d_w0 || M <= 0 || n0 <= 0 || n0 > 256 || n_red <= 0 || n_red > 256 || d0 <= 0 || d0 > 16) return MATPBR_ERR_INVALID_ARG;
  if (nprod == 3 && (!g_tile_max || n_red <= 32)) return MATPBR_ERR_INVALID_ARG;
  if ((nprod != 3 && nprod != 6 && nprod != 9) || (M % kBM) || M > 0x7fffff00L || ldc < 256 || (ldc & 3) || (ldg & 3) || ldg < ((n_red + 31) & ~31) || ldx0 < 16 || (ldx0 & 3) ||
      !aligned16(g) || !aligned16(c_prev) || !aligned16(x0))
    return MATPBR_ERR_UNSUPPORTED;
  if (d_bias0 && (!workspace || workspace_bytes < matpbr_mlp_bwd_input_workspace_bytes(M))) return MATPBR_ERR_WORKSPACE;
  if (!workspace2 || workspace2_bytes < matpbr_mlp_skinny_workspace_bytes(16)) return MATPBR_ERR_WORKSPACE;
  NtArgs p{g, nullptr, nullptr, c_prev, nullptr, nullptr, d_bias0 ? (float*)workspace : nullptr, (int)M, n0, n_red, ldg, 0, ldc};
  p.cmul_sin = sgn;
  p.x0 = x0; p.ldx0 = ldx0; p.w0_part = (float*)workspace2;
  p.a_tmax = g_tile_max;
  const int tiles = (int)(M / kBM);
  unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
  int slabs = (int)grid * 4;
  bool ok;
  if (nprod == 3) {                                        // block-scaled f16 pieces: the two-workgroup form (fewer operand registers: no spills)
    if ((long)kBM * ldg * 4 >= (1l << 31)) return MATPBR_ERR_UNSUPPORTED;
    grid = (unsigned)(tiles < 512 ? tiles : 512);
    slabs = (int)grid * 2;
    ok = lds_opt_in<&mlp_nt_gx<EPI_MULC, 3, false, false, true>>(kGxSmem);
    if (ok) hipLaunchKernelGGL((mlp_nt_gx<EPI_MULC, 3, false, false, true>), dim3(grid), dim3(kGxThreads), kGxSmem, (hipStream_t)stream, p, (const uint4*)wtsplit, HeadArgs{});
  } else if (gx_ok(p) && g_nt_w0_gx.load(std::memory_order_relaxed) != 0) {
    grid = (unsigned)(tiles < 512 ? tiles : 512);
    slabs = (int)grid * 2;
    if (nprod == 9) {
      ok = lds_opt_in<&mlp_nt_gx<EPI_MULC, 9, false, false, true>>(kGxSmem);
      if (ok) hipLaunchKernelGGL((mlp_nt_gx<EPI_MULC, 9, false, false, true>), dim3(grid), dim3(kGxThreads), kGxSmem, (hipStream_t)stream, p, (const uint4*)wtsplit, HeadArgs{});
    } else {
      ok = lds_opt_in<&mlp_nt_gx<EPI_MULC, 6, false, false, true>>(kGxSmem);
      if (ok) hipLaunchKernelGGL((mlp_nt_gx<EPI_MULC, 6, false, false, true>), dim3(grid), dim3(kGxThreads), kGxSmem, (hipStream_t)stream, p, (const uint4*)wtsplit, HeadArgs{});
    }
  } else {
    ok = nprod == 9 ? launch_nt_bx_w0<9>(p, (const uint4*)wtsplit, grid, (hipStream_t)stream) : launch_nt_bx_w0<6>(p, (const uint4*)wtsplit, grid, (hipStream_t)stream);
  }
  if (!ok) return MATPBR_ERR_LAUNCH;
  if (defer) defer[0].kind = MATPBR_REDUCE_NONE;
  if (d_bias0) {
    MatpbrReduceJob q{};
    q.kind = MATPBR_REDUCE_COLSUM; q.groups = (int)grid; q.src = (const float*)workspace; q.dst = d_bias0; q.n0 = n0;
    reduce_or_defer(q, defer, (hipStream_t)stream);
  }
  MatpbrReduceJob q2{};
  q2.kind = MATPBR_REDUCE_SKINNY; q2.groups = slabs; q2.src = (const float*)workspace2; q2.dst = d_w0; q2.n0 = 16; q2.n1 = d0; q2.n2 = n0; q2.ld_j = ld_j; q2.ld_c = ld_c;
  reduce_or_defer(q2, defer ? defer + 1 : nullptr, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

int matpbr_mlp_first_layer_bwd_bx(const float* g, int ldg, const void* wtsplit, const float* c_prev, int ldc, int sgn, const float* x0, int ldx0,
                                  float* d_w0, long ld_j, long ld_c, int d0, float* d_bias0, void* workspace, size_t workspace_bytes, void* workspace2,
                                  size_t workspace2_bytes, long M, int n0, int n_red, int nprod, void* stream) {
  if (nprod == 3) return MATPBR_ERR_INVALID_ARG;           // the f16 form needs the tile maxima: matpbr_mlp_first_layer_bwd_blk
  return mlp_first_layer_bwd_impl(g, ldg, wtsplit, c_prev, ldc, sgn, x0, ldx0, d_w0, ld_j, ld_c, d0, d_bias0, workspace, workspace_bytes, workspace2,
                                  workspace2_bytes, M, n0, n_red, nprod, nullptr, stream);
}
int matpbr_mlp_first_layer_bwd_blk(const float* g, int ldg, const void* g_tile_max, const void* wtsplit, const float* s_prev, int lds, const float* x0, int ldx0,
                                   float* d_w0, long ld_j, long ld_c, int d0, float* d_bias0, void* workspace, size_t workspace_bytes, void* workspace2,
                                   size_t workspace2_bytes, long M, int n0, int n_red, MatpbrReduceJob* defer2, void* stream) {
  return mlp_first_layer_bwd_impl(g, ldg, wtsplit, s_prev, lds, 1, x0, ldx0, d_w0, ld_j, ld_c, d0, d_bias0, workspace, workspace_bytes, workspace2,
                                  workspace2_bytes, M, n0, n_red, 3, (const unsigned*)g_tile_max, stream, defer2);
}
int matpbr_mlp_layer_bwd_input_blk(const float* g, int ldg, const void* g_tile_max, const void* wtsplit, const float* s_prev, float* g_prev, int ldo,
                                   void* out_tile_max, float* d_bias_prev, void* workspace, size_t workspace_bytes, long M, int n_prev, int n_red,
                                   MatpbrReduceJob* defer, void* stream) {
  return mlp_layer_bwd_input_bx_impl(g, ldg, wtsplit, s_prev, g_prev, ldo, d_bias_prev, workspace, workspace_bytes, M, n_prev, n_red, 3, 1, stream,
                                     (const unsigned*)g_tile_max, (unsigned*)out_tile_max, defer);
}

int matpbr_mlp_layer_bwd_input_bx(const float* g, int ldg, const void* wtsplit, const float* c_prev, float* g_prev, int ldo, float* d_bias_prev,
                                  void* workspace, size_t workspace_bytes, long M, int n_prev, int n_red, int nprod, void* stream) {
  if (nprod == 3) return MATPBR_ERR_INVALID_ARG;
  return mlp_layer_bwd_input_bx_impl(g, ldg, wtsplit, c_prev, g_prev, ldo, d_bias_prev, workspace, workspace_bytes, M, n_prev, n_red, nprod, 0, stream);
}
int matpbr_mlp_layer_bwd_input_bx_sgn(const float* g, int ldg, const void* wtsplit, const float* s_prev, float* g_prev, int ldo, float* d_bias_prev,
                                      void* workspace, size_t workspace_bytes, long M, int n_prev, int n_red, int nprod, void* stream) {
  if (nprod == 3) return MATPBR_ERR_INVALID_ARG;
  return mlp_layer_bwd_input_bx_impl(g, ldg, wtsplit, s_prev, g_prev, ldo, d_bias_prev, workspace, workspace_bytes, M, n_prev, n_red, nprod, 1, stream);
}

static int wgrad_slabs(long M) {
  long s = (M + 255) / 256;     // >= 256 rows per slab
  if (s > 256) s = 256;
  if (s < 1) s = 1;
  return (int)s;
}

size_t matpbr_mlp_bwd_weight_workspace_bytes(long M) { return (size_t)wgrad_slabs(M) * 256 * 256 * sizeof(float); }

int matpbr_mlp_layer_bwd_weight(const float* g, int ldg, const float* x, int ldx, float* d_w, int ldw, void* workspace,
                                size_t workspace_bytes, long M, int N, int K, void* stream) {
  if (!g || !x || !d_w || M <= 0 || N <= 0 || N > 256 || K <= 0 || K > 256) return MATPBR_ERR_INVALID_ARG;
  if ((ldg & 3) || (ldx & 3) || ldg < N || ldx < ((K + 3) & ~3) || ldw < K || !aligned16(g) || !aligned16(x)) return MATPBR_ERR_INVALID_ARG;
  if (M <= kSmallM) {
    const int tiles = ((N + 31) / 32) * ((K + 31) / 32);
    hipLaunchKernelGGL(mlp_small_tn, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, g, ldg, x, ldx, d_w, ldw, (int)M, N, K);
    return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
  }
  if (!workspace || workspace_bytes < matpbr_mlp_bwd_weight_workspace_bytes(M)) return MATPBR_ERR_WORKSPACE;
  const int slabs = wgrad_slabs(M);
  long rows = (M + slabs - 1) / slabs;
  rows = (rows + kWM - 1) / kWM * kWM;
  hipLaunchKernelGGL(mlp_wgrad_tn, dim3(slabs, (N + 127) / 128), dim3(256), 0, (hipStream_t)stream, g, ldg, x, ldx, (float*)workspace, M, rows, K);
  hipLaunchKernelGGL(mlp_wgrad_reduce, dim3(256), dim3(1024), 0, (hipStream_t)stream, (const float*)workspace, slabs, d_w, N, K, ldw);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

static int mlp_layer_bwd_weight_bx_impl(const float* g, int ldg, const float* x, int ldx, float* d_w, int ldw, void* workspace,
                                        size_t workspace_bytes, long M, int N, int K, int nprod, const unsigned* g_tile_max, void* stream,
                                        MatpbrReduceJob* defer = nullptr) {
  if (!g || !x || !d_w || M <= 0 || N <= 0 || N > 256 || K <= 0 || K > 256) return MATPBR_ERR_INVALID_ARG;
  if (nprod == 3 && (!g_tile_max || (M % kBM))) return MATPBR_ERR_INVALID_ARG;
  if ((nprod != 3 && nprod != 6 && nprod != 9) || (M & 15) || ldg < 256 || ldx < 256 || (ldg & 3) || (ldx & 3) || ldw < K || !aligned16(g) || !aligned16(x))
    return MATPBR_ERR_UNSUPPORTED;   // all 256 columns of both are read
  if (!workspace || workspace_bytes < matpbr_mlp_bwd_weight_workspace_bytes(M)) return MATPBR_ERR_WORKSPACE;
  int slabs = wgrad_slabs(M);
  long rows = ((M + slabs - 1) / slabs + 15) / 16 * 16;
  slabs = (int)((M + rows - 1) / rows);
  if (nprod == 3 && MATPBR_WG_HX && (long)ldg * 4 >= 1024 && (long)ldx * 4 >= 1024 && (long)3 * ldg * 4 + 1024 < (1l << 31) && (long)3 * ldx * 4 + 1024 < (1l << 31)) {
    if (!lds_opt_in<&mlp_wgrad_hx>(kHxSmem)) return MATPBR_ERR_LAUNCH;
    hipLaunchKernelGGL(mlp_wgrad_hx, dim3(slabs), dim3(kWgThreads), kHxSmem, (hipStream_t)stream, g, ldg, x, ldx, (float*)workspace, M, rows, g_tile_max);
  } else if (nprod == 3) {
    if (!lds_opt_in<&mlp_wgrad_bx<3>>(kWgSmem)) return MATPBR_ERR_LAUNCH;
    hipLaunchKernelGGL(mlp_wgrad_bx<3>, dim3(slabs), dim3(kWgThreads), kWgSmem, (hipStream_t)stream, g, ldg, x, ldx, (float*)workspace, M, rows, g_tile_max);
  } else if (nprod == 6) {
    if (!lds_opt_in<&mlp_wgrad_bx<6>>(kWgSmem)) return MATPBR_ERR_LAUNCH;
    hipLaunchKernelGGL(mlp_wgrad_bx<6>, dim3(slabs), dim3(kWgThreads), kWgSmem, (hipStream_t)stream, g, ldg, x, ldx, (float*)workspace, M, rows, (const unsigned*)nullptr);
  } else {
    if (!lds_opt_in<&mlp_wgrad_bx<9>>(kWgSmem)) return MATPBR_ERR_LAUNCH;
    hipLaunchKernelGGL(mlp_wgrad_bx<9>, dim3(slabs), dim3(kWgThreads), kWgSmem, (hipStream_t)stream, g, ldg, x, ldx, (float*)workspace, M, rows, (const unsigned*)nullptr);
  }
  MatpbrReduceJob q{};
  q.kind = MATPBR_REDUCE_WGRAD; q.groups = slabs; q.src = (const float*)workspace; q.dst = d_w; q.n0 = N; q.n1 = K; q.n2 = ldw;
  reduce_or_defer(q, defer, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}
int matpbr_mlp_layer_bwd_weight_bx(const float* g, int ldg, const float* x, int ldx, float* d_w, int ldw, void* workspace,
                                   size_t workspace_bytes, long M, int N, int K, int nprod, void* stream) {
  if (nprod == 3) return MATPBR_ERR_INVALID_ARG;           // matpbr_mlp_layer_bwd_weight_blk
  return mlp_layer_bwd_weight_bx_impl(g, ldg, x, ldx, d_w, ldw, workspace, workspace_bytes, M, N, K, nprod, nullptr, stream);
}
int matpbr_mlp_layer_bwd_weight_blk(const float* g, int ldg, const void* g_tile_max, const float* x, int ldx, float* d_w, int ldw, void* workspace,
                                    size_t workspace_bytes, long M, int N, int K, MatpbrReduceJob* defer, void* stream) {
  return mlp_layer_bwd_weight_bx_impl(g, ldg, x, ldx, d_w, ldw, workspace, workspace_bytes, M, N, K, 3, (const unsigned*)g_tile_max, stream, defer);
}

int matpbr_mlp_skinny_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias, float* out, int ldo, long M, int J, int K,
                          void* stream) {
  if (!x || !w || !bias || !out || M <= 0 || (J != 3 && J != 5 && J != 8) || K <= 0 || K > 256 || (K & 3)) return MATPBR_ERR_INVALID_ARG;
  if ((ldx & 3) || ldx < K || (ldw & 3) || ldw < K || ldo < J || !aligned16(x) || !aligned16(w)) return MATPBR_ERR_INVALID_ARG;
  const ArmHead none{nullptr, 0, nullptr, nullptr, nullptr, nullptr};
  if (J == 3) launch_skinny_nt<3, false>(x, ldx, w, ldw, bias, out, ldo, M, K, none, (hipStream_t)stream);
  else if (J == 5) launch_skinny_nt<5, false>(x, ldx, w, ldw, bias, out, ldo, M, K, none, (hipStream_t)stream);
  else launch_skinny_nt<8, false>(x, ldx, w, ldw, bias, out, ldo, M, K, none, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

int matpbr_mlp_arm_head_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias, const float* start, int lds, float* th,
                            float* map_a, float* map_r, float* map_m, long M, int K, void* stream) {
  if (!x || !w || !bias || !start || !th || M <= 0 || K <= 0 || K > 256 || (K & 3) || lds < 5) return MATPBR_ERR_INVALID_ARG;
  if ((ldx & 3) || ldx < K || (ldw & 3) || ldw < K || !aligned16(x) || !aligned16(w) || !aligned16(th)) return MATPBR_ERR_INVALID_ARG;
  const ArmHead h{start, lds, th, map_a, map_r, map_m};
  launch_skinny_nt<5, true>(x, ldx, w, ldw, bias, nullptr, 0, M, K, h, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

int matpbr_mlp_arm_head_bwd(const float* g_a, const float* g_r, const float* g_m, const float* th, float* d_x, long M, void* stream) {
  if (!th || !d_x || M <= 0 || !aligned16(th) || !aligned16(d_x)) return MATPBR_ERR_INVALID_ARG;
  const unsigned grid = (unsigned)((M + 255) / 256 < 4096 ? (M + 255) / 256 : 4096);
  hipLaunchKernelGGL(arm_head_bwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, g_a, g_r, g_m, th, d_x, M);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

size_t matpbr_mlp_skinny_workspace_bytes(int J) {   // slab partials [slabs][JP][256], bias partials [slabs][JP], column sums of the fused input gradient [slabs][256]
  return J > 0 && J <= 16 ? (size_t)kSkinnySlabs * (((J + 7) / 8 * 8) * 257 + 256) * sizeof(float) : 0;
}

int matpbr_mlp_skinny_bwd_weight(const float* s, int lds, const float* b, int ldb, float* d_w, long ld_j, long ld_c, float* d_bias, void* workspace,
                                 size_t workspace_bytes, long M, int J, int C, void* stream) {
  if (!s || !b || !d_w || M <= 0 || J <= 0 || J > 16 || C <= 0 || C > 256) return MATPBR_ERR_INVALID_ARG;
  const int JP = (J + 7) / 8 * 8;                        // columns of S that are read: the caller pads S to a multiple of 8 columns
  if (lds < JP || ldb < 256 || (ldb & 3) || !aligned16(b)) return MATPBR_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < matpbr_mlp_skinny_workspace_bytes(J)) return MATPBR_ERR_WORKSPACE;
  long rows = (M + kSkinnySlabs - 1) / kSkinnySlabs;
  rows = (rows + 3) / 4 * 4;
  const int slabs = (int)((M + rows - 1) / rows);
  float* partial = (float*)workspace;
  float* bpart = partial + (size_t)kSkinnySlabs * JP * 256;
  if (JP == 8) hipLaunchKernelGGL((mlp_skinny_tn_kernel<8>), dim3(slabs), dim3(256), 0, (hipStream_t)stream, s, lds, b, ldb, partial, bpart, M, rows, SkinnyDgrad{});
  else hipLaunchKernelGGL((mlp_skinny_tn_kernel<16>), dim3(slabs), dim3(256), 0, (hipStream_t)stream, s, lds, b, ldb, partial, bpart, M, rows, SkinnyDgrad{});
  hipLaunchKernelGGL(mlp_skinny_tn_reduce, dim3(JP * 4), dim3(1024), 0, (hipStream_t)stream, (const float*)partial, (const float*)bpart, slabs, JP, J, C, d_w,
                     ld_j, ld_c, d_bias, (const float*)nullptr, 0, (float*)nullptr);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

static int mlp_out_layer_bwd_impl(const float* d_x, int ldd, const float* s_prev, const float* c_prev, int lds, const float* w_out, int ldw, float* g_prev,
                                  int ldg, float* d_w, long ld_j, long ld_c, float* d_bias, float* d_bias_prev, void* workspace, size_t workspace_bytes,
                                  long M, int J, int n_prev, unsigned* g_tile_max, void* stream, MatpbrReduceJob* defer = nullptr) {
  if (!d_x || !s_prev || !w_out || !g_prev || !d_w || M <= 0 || J <= 0 || J > 8 || n_prev <= 0 || n_prev > 256) return MATPBR_ERR_INVALID_ARG;
  if (ldd < 8 || lds < 256 || (lds & 3) || ldg < 256 || (ldg & 3) || ldw < 256 || (ldw & 3) || !aligned16(s_prev) || !aligned16(g_prev) || !aligned16(w_out) ||
      (c_prev && !aligned16(c_prev)))
    return MATPBR_ERR_INVALID_ARG;
  if (!workspace || workspace_bytes < matpbr_mlp_skinny_workspace_bytes(J)) return MATPBR_ERR_WORKSPACE;
  long rows = (M + kSkinnySlabs - 1) / kSkinnySlabs;
  rows = (rows + 3) / 4 * 4;
  const int slabs = (int)((M + rows - 1) / rows);
  float* partial = (float*)workspace;
  float* bpart = partial + (size_t)kSkinnySlabs * 8 * 256;
  float* gsum_part = bpart + (size_t)kSkinnySlabs * 8;
  SkinnyDgrad dg{w_out, ldw, J, c_prev, g_prev, ldg, gsum_part, g_tile_max};
  if (J <= kOutJ) hipLaunchKernelGGL((mlp_skinny_tn_kernel<8, true>), dim3(slabs), dim3(256), 0, (hipStream_t)stream, d_x, ldd, s_prev, lds, partial, bpart, M, rows, dg);
  else hipLaunchKernelGGL((mlp_skinny_tn_kernel<8, true, 8>), dim3(slabs), dim3(256), 0, (hipStream_t)stream, d_x, ldd, s_prev, lds, partial, bpart, M, rows, dg);
  MatpbrReduceJob q{};
  q.kind = MATPBR_REDUCE_SKINNY; q.groups = slabs; q.src = partial; q.src_b = bpart; q.src_g = gsum_part; q.dst = d_w; q.dst_b = d_bias; q.dst_g = d_bias_prev;
  q.n0 = 8; q.n1 = J; q.n2 = 256; q.n3 = n_prev; q.ld_j = ld_j; q.ld_c = ld_c;
  reduce_or_defer(q, defer, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

int matpbr_mlp_out_layer_bwd(const float* d_x, int ldd, const float* s_prev, const float* c_prev, int lds, const float* w_out, int ldw, float* g_prev,
                             int ldg, float* d_w, long ld_j, long ld_c, float* d_bias, float* d_bias_prev, void* workspace, size_t workspace_bytes,
                             long M, int J, int n_prev, void* stream) {
  return mlp_out_layer_bwd_impl(d_x, ldd, s_prev, c_prev, lds, w_out, ldw, g_prev, ldg, d_w, ld_j, ld_c, d_bias, d_bias_prev, workspace, workspace_bytes, M, J,
                                n_prev, nullptr, stream);
}
int matpbr_mlp_out_layer_bwd_tmax(const float* d_x, int ldd, const float* s_prev, const float* c_prev, int lds, const float* w_out, int ldw, float* g_prev,
                                  int ldg, void* g_tile_max, float* d_w, long ld_j, long ld_c, float* d_bias, float* d_bias_prev, void* workspace,
                                  size_t workspace_bytes, long M, int J, int n_prev, MatpbrReduceJob* defer, void* stream) {
  if (!g_tile_max) return MATPBR_ERR_INVALID_ARG;
  return mlp_out_layer_bwd_impl(d_x, ldd, s_prev, c_prev, lds, w_out, ldw, g_prev, ldg, d_w, ld_j, ld_c, d_bias, d_bias_prev, workspace, workspace_bytes, M, J,
                                n_prev, (unsigned*)g_tile_max, stream, defer);
}

int matpbr_mlp_reduce_jobs(const MatpbrReduceJob* jobs, int n_jobs, void* stream) {
  if (!jobs || n_jobs <= 0 || n_jobs > kMaxReduceJobs) return MATPBR_ERR_INVALID_ARG;
  ReduceJobs js{};
  int blocks = 0;
  for (int i = 0; i < n_jobs; ++i) {
    const MatpbrReduceJob& q = jobs[i];
    if (q.kind == MATPBR_REDUCE_NONE) continue;
    if ((q.kind != MATPBR_REDUCE_WGRAD && q.kind != MATPBR_REDUCE_COLSUM && q.kind != MATPBR_REDUCE_SKINNY) || !q.src || !q.dst || q.groups <= 0 || q.n0 <= 0)
      return MATPBR_ERR_INVALID_ARG;
    js.j[js.n] = q;
    js.first[js.n] = blocks;
    blocks += reduce_job_blocks(q);
    ++js.n;
  }
  if (js.n == 0) return MATPBR_OK;
  js.first[js.n] = blocks;
  hipLaunchKernelGGL(mlp_reduce_jobs_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, js);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

}  // extern "C"
