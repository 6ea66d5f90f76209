// The forward pass of the 'arm' coordinate MLP (mymodels/mlps.py:211-236: 15 -> 241 -> 256 -> 241 -> 256 -> 5 with the skip concatenations
// of :214-217 and the tanh head of :232-234, as inverse_img_w_mi.py:493-496 drives it) as ONE launch: the activations of a row never
// leave the registers of the wave that owns it between two layers (round 5).
//
// Layer by layer (posmlp_kernels.hip) every 256-wide layer reads the [M,256] sines of the layer below from HBM (268 MB at 512 x 512) and
// writes its own; the backward pass needs every layer's sines once, so the writes stay -- the reads do not have to exist.  Here the product
// is formed TRANSPOSED, D[f][b] = sum_k W[f][k] X[b][k]: the weights are the A operand of v_mfma_f32_32x32x16_f16, the rows the B operand.
// A lane of the 32 x 32 result holds column b (one row of the batch) and the features f = 8 q + 4 h + t (q, t < 4; h = lane half) of each
// 32-feature block: after the sine these ARE the lane's share of the next layer's B operand -- eight consecutive slots of a 16-deep k-step
// per (block, half) -- once the weight image orders its k accordingly (MATPBR_WSPLIT_CHAIN: slot (g, i) of k-step S is feature
// 16 S + 8 (i / 4) + 4 g + i % 4).  No transposition, no LDS round trip, no HBM read of activations.
//
// A wave owns 32 rows and all 256 features: 8 accumulator blocks (128 registers).  Block t of a finished layer yields k-steps 2t, 2t+1 of
// the next one, and the next layer's products of those two steps (48 MFMAs over its 8 blocks) do not depend on the blocks still to be
// drained: the epilogue of block t + 1 (sine, packed-sine store, two f16 pieces per value: ~20 VALU per output) is interleaved, chunk by
// chunk, with the products that block t released -- two accumulator sets, the old one draining while the new one fills, and only two
// k-steps' worth of B pieces alive at any time.  One wave per SIMD (~350 registers), four per workgroup, one workgroup per CU.
// Weights: two f16 pieces of 256 w (posmlp_device.hpp, split2h), streamed from L2 through a ring of three 32 KB LDS stages by LDS-DMA, one
// stage (two k-steps) ahead of the products; the first layer (K = 15, arguments of hundreds of radians) on v_mfma_f32_32x32x2_f32 from an
// f32 image of its weights; the output layer as a one-block product on the same two-piece scheme, then the 'arm' head per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "../../include/matpbr.h"
#include "posmlp_device.hpp"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
constexpr int kChStage = 32768;                         // bytes of a weight stage: [2 k-steps][2 pieces][2 lane halves][256 features] x 16 B
constexpr int kChLayer = 8 * kChStage;                  // a layer's image: 8 stages
constexpr int kChW0 = 8 * 2 * 32 * 8 * 4;               // [8 blocks][2 g][32 i][8 s] f32: W0[32 fb + i][2 s + g]
constexpr int kChHead = 16 * 2 * 2 * 32 * 16;           // [16 k-steps][2 pieces][2 g][32 outputs] x 16 B
constexpr int kChBias = (4 * 256 + 8) * 4;              // the four sine layers' biases, the output layer's (8 floats)
constexpr int kChOffW0 = 3 * kChLayer, kChOffHead = kChOffW0 + kChW0, kChOffBias = kChOffHead + kChHead;
constexpr size_t kChImages = (size_t)kChOffBias + kChBias;
constexpr size_t kChSmem = 3 * kChStage + kChW0 + kChHead + kChBias;   // 151 KB

struct ChainPrep {
  const float* w[5];      // W0 [n0, ldw >= d0], W1..W3 [n, ldw >= 256], W_out [n_head, ldw >= 256]
  const float* b[5];
  int ldw[5], n[5], d0;
  unsigned char* images;
  // what else changes once per optimiser step, in the same launch (jobs 6..): the input-gradient products' operands -- MATPBR_WSPLIT_F16X2 |
  // MATPBR_WSPLIT_TRANSPOSED images of W1..W3 (element (n, k) = W[k][n], n < bn, k < bk: mlp_split_weights_kernel's item, the same bits) --
  // and the reset of the gradient tiles' maxima
  uint4* bwd[3];
  int bn[3], bk[3], n_bwd;
  unsigned* zero;
  int zero_words;
};

// blockIdx.y: 0..2 the chain images of layers 1..3, 3 the f32 image of the first layer, 4 the output layer's image, 5 the biases,
// 6.. the backward images, then the zeros
__global__ __launch_bounds__(256) void mlp_chain_prep_kernel(const ChainPrep a) {
  const int job = blockIdx.y, idx = blockIdx.x * 256 + threadIdx.x;
  if (job >= 6) {
    const int jb = job - 6;
    if (jb < a.n_bwd) {
      const int N = a.bn[jb], K = a.bk[jb], nks = (K + 31) / 32;
      if (idx >= nks * 2 * 256 * 2) return;
      const int g = idx & 1, n = (idx >> 1) & 255, s2 = (idx >> 9) & 1, ks = idx >> 10;
      const float* B = a.w[1 + jb];
      const int ldb = a.ldw[1 + jb];
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = 32 * ks + 16 * g + 8 * s2 + q;
        v[q] = (n < N && k < K) ? B[(size_t)k * ldb + n] : 0.f;
      }
      unsigned p[2][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) split2h(v[2 * q] * kF16WScale, v[2 * q + 1] * kF16WScale, p[0][q], p[1][q]);
#pragma unroll
      for (int piece = 0; piece < 2; ++piece)
        a.bwd[jb][((((size_t)ks * 2 + s2) * 2 + piece) * 2 + g) * 256 + n] = make_uint4(p[piece][0], p[piece][1], p[piece][2], p[piece][3]);
    } else {
      for (int i = idx; i < a.zero_words; i += 32 * 256) a.zero[i] = 0u;
    }
    return;
  }
  if (job < 3) {                                               // thread = (stage ks, k-step s, feature n, lane half g): 8 slots of one k-step
    if (idx >= 8 * 2 * 256 * 2) return;
    const int g = idx & 1, n = (idx >> 1) & 255, s = (idx >> 9) & 1, ks = idx >> 10;
    const float* W = a.w[1 + job];
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = 16 * (2 * ks + s) + 8 * (i >> 2) + 4 * g + (i & 3);
      v[i] = n < a.n[1 + job] ? W[(size_t)n * a.ldw[1 + job] + k] * kF16WScale : 0.f;
    }
    unsigned p[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split2h(v[2 * q], v[2 * q + 1], p[0][q], p[1][q]);
    uint4* out = reinterpret_cast<uint4*>(a.images + (size_t)job * kChLayer);
#pragma unroll
    for (int piece = 0; piece < 2; ++piece)
      out[((((size_t)ks * 2 + s) * 2 + piece) * 2 + g) * 256 + n] = make_uint4(p[piece][0], p[piece][1], p[piece][2], p[piece][3]);
  } else if (job == 3) {
    if (idx >= 8 * 2 * 32 * 8) return;
    const int s = idx & 7, i = (idx >> 3) & 31, g = (idx >> 8) & 1, fb = idx >> 9;
    const int f = 32 * fb + i, k = 2 * s + g;
    reinterpret_cast<float*>(a.images + kChOffW0)[idx] = (f < a.n[0] && k < a.d0) ? a.w[0][(size_t)f * a.ldw[0] + k] : 0.f;
  } else if (job == 4) {
    if (idx >= 16 * 2 * 32) return;
    const int i = idx & 31, g = (idx >> 5) & 1, S = idx >> 6;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = 16 * S + 8 * (q >> 2) + 4 * g + (q & 3);
      v[q] = i < a.n[4] ? a.w[4][(size_t)i * a.ldw[4] + k] * kF16WScale : 0.f;
    }
    unsigned p[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) split2h(v[2 * q], v[2 * q + 1], p[0][q], p[1][q]);
    uint4* out = reinterpret_cast<uint4*>(a.images + kChOffHead);
#pragma unroll
    for (int piece = 0; piece < 2; ++piece) out[(((size_t)S * 2 + piece) * 2 + g) * 32 + i] = make_uint4(p[piece][0], p[piece][1], p[piece][2], p[piece][3]);
  } else {
    if (idx >= 4 * 256 + 8) return;
    const int l = idx >> 8, f = idx & 255;
    reinterpret_cast<float*>(a.images + kChOffBias)[idx] = f < a.n[l] ? a.b[l][f] : 0.f;
  }
}

struct ChainArgs {
  const float* x0;          // [M, ldx0 >= 16]: the network's input rows, zero beyond d0
  int ldx0;
  const unsigned char* images;
  float* out[4];            // the four sine layers' buffers [M, ldo]: sign-carrying sines, the x0 tail behind a skip layer's outputs
  int ldo;
  int tail[4];              // layer l has 241 outputs: columns 241.. of its buffer (and of the next layer's input) are x0
  ArmHead head;
  int n_head;
  int M;
};

template <bool TAIL>
__device__ __forceinline__ float tail_pick(float v, const float (&xr)[16], int h, int q, int t) {   // feature 224 + 8 q + 4 h + t of a 241-wide layer's last block
  if (!TAIL) return v;
  const int i0 = 8 * q + t - 17, i1 = 8 * q + 4 + t - 17;                         // x0 column for lane half 0 / 1 (negative: a sine)
  const float a = i0 >= 0 ? xr[i0 < 0 ? 0 : i0] : v, b = i1 >= 0 ? xr[i1 < 0 ? 0 : i1] : v;
  return h ? b : a;
}

__device__ __forceinline__ void set_comp(uint4& u, int comp, unsigned v) {      // comp is a constant after unrolling: no address is taken
  if (comp == 0) u.x = v;
  else if (comp == 1) u.y = v;
  else if (comp == 2) u.z = v;
  else u.w = v;
}

__global__ __launch_bounds__(256, 1) void mlp_chain_fwd_kernel(const ChainArgs p) {
  extern __shared__ __align__(16) unsigned char ch_smem[];
  unsigned char* sRing = ch_smem;
  float* sW0 = reinterpret_cast<float*>(ch_smem + 3 * kChStage);
  const unsigned char* sHead = ch_smem + 3 * kChStage + kChW0;
  const float* sBias = reinterpret_cast<const float*>(ch_smem + 3 * kChStage + kChW0 + kChHead);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 31, h = lane >> 5;
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned lds0 = lds_byte_address(ch_smem);
  const unsigned dma_voff = (unsigned)(wave * 8192 + lane * 16);
  const int tiles = p.M / 128;

  auto issue_stage = [&](int P) {                              // P: this workgroup's running stage count; layer (P / 8) % 3, stage P % 8
    const unsigned char* src = p.images + (size_t)((P >> 3) % 3) * kChLayer + (size_t)(P & 7) * kChStage;
    const unsigned dst = lds0 + (unsigned)(P % 3) * (unsigned)kChStage + (unsigned)wave_u * 8192u;
    glds16_x4(src, dma_voff, dst);
    glds16_x4(src + 4096, dma_voff, dst + 4096u);
  };
  auto issue_piece = [&](int P, int i) {                       // piece i (0..7) of this wave's share of stage P
    const unsigned char* src = p.images + (size_t)((P >> 3) % 3) * kChLayer + (size_t)(P & 7) * kChStage + i * 1024;
    glds16(src, dma_voff, lds0 + (unsigned)(P % 3) * (unsigned)kChStage + (unsigned)wave_u * 8192u + (unsigned)i * 1024u);
  };
  issue_stage(0);
  issue_stage(1);
  {                                                            // the small images: first layer, output layer, biases (they change every step)
    const uint4* src = reinterpret_cast<const uint4*>(p.images + kChOffW0);
    uint4* dst = reinterpret_cast<uint4*>(ch_smem + 3 * kChStage);
    for (int i = tid; i < (kChW0 + kChHead + kChBias) / 16; i += 256) dst[i] = src[i];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the first two stages have landed (the counted waits below start from a clean slate)
  __syncthreads();
  int P = 0;

  for (int tile = blockIdx.x; tile < tiles; tile += gridDim.x) {
    const long row = (long)tile * 128 + wave * 32 + j;         // this lane's row of the batch
    float xr[16];
    {
      const float4* xs = reinterpret_cast<const float4*>(p.x0 + row * p.ldx0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = xs[q];
        xr[4 * q] = v.x; xr[4 * q + 1] = v.y; xr[4 * q + 2] = v.z; xr[4 * q + 3] = v.w;
      }
    }
    // ---- first layer: D[f][b] = sum_k W0[f][k] x0[b][k] on the f32 matrix instruction (A: lane (i, g) = W0[32 fb + i][2 s + g]; B: lane (j, g) = x0[j][2 s + g])
    f32x16 acc[8];
    {
      float xb[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        float even = xr[2 * s], odd = xr[2 * s + 1];
        asm volatile("" : "+v"(even), "+v"(odd));              // opaque: the select must not become an indexed read of xr (scratch memory)
        xb[s] = h ? odd : even;
      }
#pragma unroll
      for (int fb = 0; fb < 8; ++fb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[fb][r] = 0.f;
        const float4 w_lo = *reinterpret_cast<const float4*>(sW0 + ((fb * 2 + h) * 32 + j) * 8);
        const float4 w_hi = *reinterpret_cast<const float4*>(sW0 + ((fb * 2 + h) * 32 + j) * 8 + 4);
        const float wa[8] = {w_lo.x, w_lo.y, w_lo.z, w_lo.w, w_hi.x, w_hi.y, w_hi.z, w_hi.w};
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[fb] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], xb[s], acc[fb], 0, 0, 0);
      }
    }
    // ---- four transitions: layer L - 1 (in `acc`) through its epilogue into layer L (L = 4: the output layer, one block).  Compile-time
    // forms (no branch inside a slot: a slot is ONE scheduling region, its products and its epilogue interleave): HEAD, and FIRST (the
    // finished layer is the first one: unit scale, arguments of hundreds of radians -> the robust cosine sign)
    auto transition = [&](auto head_tag, auto first_tag, int L) {
      constexpr bool HEAD = decltype(head_tag)::value, FIRST = decltype(first_tag)::value;
      const float unscale = FIRST ? 1.0f : kF16WUnscale;
      // (selects, not indexed kernel arguments: an indexed read would put the argument block into scratch memory)
      const bool tail = (L == 1 ? p.tail[0] : L == 2 ? p.tail[1] : L == 3 ? p.tail[2] : p.tail[3]) != 0;
      float* const outp = (L == 1 ? p.out[0] : L == 2 ? p.out[1] : L == 3 ? p.out[2] : p.out[3]) + row * p.ldo + 4 * h;
      const float* const bias = sBias + (L - 1) * 256 + 4 * h;
      f32x16 accN[8];
#pragma unroll
      for (int fb = 0; fb < (HEAD ? 1 : 8); ++fb)
#pragma unroll
        for (int r = 0; r < 16; ++r) accN[fb][r] = 0.f;
      uint4 bp[2][2][2];                                       // [slot parity][k-step of the stage][piece]: the B operand of the stage after this slot
#pragma unroll
      for (int t = 0; t <= 8; ++t) {
        const uint4* stage = reinterpret_cast<const uint4*>(sRing + (size_t)(P % 3) * kChStage) + h * 256 + j;
        if (t >= 1 && !HEAD) {
          // the weights of stage P (requested two stages ago): everything but the youngest request (8 pieces) and the stores behind it (4)
          // the weights of stage P: its last piece went out with chunk 7 two slots ago; behind it one store, then last slot's 8 pieces and 4 stores
          asm volatile("s_waitcnt vmcnt(13) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        // A fragments (weights) of block c, stage t - 1: [k-step][piece]; requested one chunk ahead of their products
        uint4 af[2][2][2];
        auto read_frags = [&](int c, uint4 (&dst)[2][2]) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int piece = 0; piece < 2; ++piece)
              dst[ks][piece] = HEAD ? reinterpret_cast<const uint4*>(sHead)[(((2 * (t - 1) + ks) * 2 + piece) * 2 + h) * 32 + j]
                                    : stage[((ks * 2 + piece) * 2) * 256 + 32 * c];
        };
        float4 bq[4];
        if (t < 8) {
#pragma unroll
          for (int q = 0; q < 4; ++q) bq[q] = *reinterpret_cast<const float4*>(bias + 32 * t + 8 * q);
        }
        if (t >= 1) read_frags(0, af[0]);
        float vv[16];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const bool mm = t >= 1 && (!HEAD || c == 0);          // the products block t - 1 released: 32 features (block c) x this stage's two k-steps
          if (t >= 1 && !HEAD) issue_piece(P + 2, c);           // one 1 KB piece of the stage after next per chunk: its issue rides in a product's shadow
          if (t >= 1 && !HEAD && c + 1 < 8) read_frags(c + 1, af[(c + 1) & 1]);
          if (mm) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              const uint4 a_hi = af[c & 1][ks][0], a_lo = af[c & 1][ks][1];
              const uint4 b_hi = bp[(t - 1) & 1][ks][0], b_lo = bp[(t - 1) & 1][ks][1];
              accN[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_lo), __builtin_bit_cast(f16x8, b_hi), accN[c], 0, 0, 0);
              accN[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_hi), __builtin_bit_cast(f16x8, b_lo), accN[c], 0, 0, 0);
              accN[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_hi), __builtin_bit_cast(f16x8, b_hi), accN[c], 0, 0, 0);
            }
          }
          if (t < 8) {                                         // the epilogue of block t of the finished layer: outputs 2 c, 2 c + 1 (features 32 t + 8 q + 4 h + t0, + 1)
            const int q = c >> 1, t0 = 2 * (c & 1);
            const float b0 = t0 ? bq[q].z : bq[q].x, b1 = t0 ? bq[q].w : bq[q].y;
            float v0 = sin_packed<FIRST>(__builtin_fmaf(acc[t][2 * c], unscale, b0));
            float v1 = sin_packed<FIRST>(__builtin_fmaf(acc[t][2 * c + 1], unscale, b1));
            if (t == 7 && tail) {                              // uniform: a 241-wide layer's last block ends in x0 (mymodels/mlps.py:214-217)
              v0 = tail_pick<true>(v0, xr, h, q, t0);
              v1 = tail_pick<true>(v1, xr, h, q, t0 + 1);
            }
            vv[2 * c] = v0;
            vv[2 * c + 1] = v1;
            unsigned p_hi, p_lo;
            split2h(v0, v1, p_hi, p_lo);
            set_comp(bp[t & 1][q >> 1][0], 2 * (q & 1) + (c & 1), p_hi);
            set_comp(bp[t & 1][q >> 1][1], 2 * (q & 1) + (c & 1), p_lo);
            if (c & 1) *reinterpret_cast<float4*>(outp + 32 * t + 8 * q) = make_float4(vv[4 * q], vv[4 * q + 1], vv[4 * q + 2], vv[4 * q + 3]);
          }
          // one chunk = one scheduling pattern: the next chunk's four fragment reads first, then each product followed by its share of
          // the epilogue's vector instructions (an in-order wave issues them in the product's shadow only if they FOLLOW it in the stream)
          if (mm && t < 8) {
            if (!HEAD && c + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int u = 0; u < 6; ++u) {
              __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (t >= 1 && !HEAD) ++P;
      }
#pragma unroll
      for (int fb = 0; fb < (HEAD ? 1 : 8); ++fb) acc[fb] = accN[fb];
    };
    transition(std::false_type{}, std::true_type{}, 1);
    for (int L = 2; L <= 3; ++L) transition(std::false_type{}, std::false_type{}, L);
    transition(std::true_type{}, std::false_type{}, 4);
    // ---- the 'arm' head on the output layer's sums: lane (row, h) holds outputs 4 h + r (r < 4) in acc[0][r]
    {
      const float* b_out = sBias + 4 * 256;
      const ArmHead& hd = p.head;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jo = 4 * h + r;
        if (jo < p.n_head) {
          const float x = __builtin_fmaf(acc[0][r], kF16WUnscale, b_out[jo]);
          const float th = tanhf(x);
          hd.th[row * 8 + jo] = th;
          const float u = __fadd_rn(__fmul_rn(1.3f, th), hd.start[row * hd.lds + jo]);
          const float y = __fsub_rn(__fadd_rn(fminf(fmaxf(u, 0.f), 1.f), u), u);
          if (jo < 3) {
            if (hd.map_a) hd.map_a[row * 3 + jo] = y;
          } else if (jo == 3) {
            if (hd.map_r) hd.map_r[row] = __fadd_rn(__fmul_rn(y, 0.93f), 0.07f);
          } else if (hd.map_m) {
            hd.map_m[row] = y;
          }
        }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the look-ahead stages: nothing may land after the workgroup ends
}

bool chain_lds_opt_in() {
  static int done[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (done[dev & 63]) return true;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_chain_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChSmem) != hipSuccess) return false;
  done[dev & 63] = 1;
  return true;
}

}  // namespace

extern "C" {

size_t matpbr_mlp_chain_images_bytes(void) { return kChImages; }

int matpbr_mlp_chain_prep(const float* const* w, const int* ldw, const int* n, const float* const* bias, int d0, void* images, void* const* bwd_images,
                          void* zero, long zero_words, void* stream) {
  if (!w || !ldw || !n || !bias || !images || d0 <= 0 || d0 > 16 || zero_words < 0 || zero_words > 0x7fffffffL || (zero_words > 0 && !zero))
    return MATPBR_ERR_INVALID_ARG;
  ChainPrep a{};
  for (int l = 0; l < 5; ++l) {
    if (!w[l] || !bias[l] || n[l] <= 0 || n[l] > 256 || ldw[l] < (l == 0 ? d0 : 256)) return MATPBR_ERR_INVALID_ARG;
    a.w[l] = w[l]; a.b[l] = bias[l]; a.ldw[l] = ldw[l]; a.n[l] = n[l];
  }
  if (n[4] > 8) return MATPBR_ERR_INVALID_ARG;
  a.d0 = d0;
  a.images = (unsigned char*)images;
  if (bwd_images) {                                         // (W_l[:, :n_{l-1}])^T for l = 1..3: n_{l-1} outputs over n_l reduction columns
    for (int l = 1; l <= 3; ++l) {
      if (!bwd_images[l - 1]) return MATPBR_ERR_INVALID_ARG;
      a.bwd[l - 1] = (uint4*)bwd_images[l - 1];
      a.bn[l - 1] = n[l - 1]; a.bk[l - 1] = n[l];
    }
    a.n_bwd = 3;
  }
  a.zero = (unsigned*)zero;
  a.zero_words = (int)zero_words;
  const unsigned jobs = 6u + (unsigned)a.n_bwd + (zero_words > 0 ? 1u : 0u);
  hipLaunchKernelGGL(mlp_chain_prep_kernel, dim3(32, jobs), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

int matpbr_mlp_chain_fwd(const float* x0, int ldx0, const void* images, float* const* s_out, int ldo, const int* n, const float* start, int lds, float* th,
                         float* map_a, float* map_r, float* map_m, int n_head, long M, void* stream) {
  if (!x0 || !images || !s_out || !n || !start || !th || M <= 0 || n_head != 5 || lds < n_head) return MATPBR_ERR_INVALID_ARG;      // (the head epilogue is the five-output 'arm' head)
  if ((M % 128) || M > 0x7fffff00L || ldx0 < 16 || (ldx0 & 3) || ldo < 256 || (ldo & 3) || (reinterpret_cast<uintptr_t>(x0) & 15)) return MATPBR_ERR_UNSUPPORTED;
  ChainArgs p{};
  p.x0 = x0; p.ldx0 = ldx0; p.images = (const unsigned char*)images; p.ldo = ldo; p.M = (int)M; p.n_head = n_head;
  for (int l = 0; l < 4; ++l) {
    if (!s_out[l] || (reinterpret_cast<uintptr_t>(s_out[l]) & 15) || (n[l] != 256 && n[l] != 241)) return MATPBR_ERR_UNSUPPORTED;
    p.out[l] = s_out[l];
    p.tail[l] = n[l] == 241;
  }
  p.head = ArmHead{start, lds, th, map_a, map_r, map_m};
  if (!chain_lds_opt_in()) return MATPBR_ERR_LAUNCH;
  const int tiles = (int)(M / 128);
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const unsigned grid = (unsigned)(tiles < cus ? tiles : cus);
  hipLaunchKernelGGL(mlp_chain_fwd_kernel, dim3(grid), dim3(256), kChSmem, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? MATPBR_OK : MATPBR_ERR_LAUNCH;
}

}  // extern "C"
