// Device helpers shared by the PosMLP kernels (posmlp_kernels.hip: the layer-by-layer kernels; posmlp_chain.hip: the forward chain):
// sines, packed sines, operand splits, LDS-DMA, wave reductions, the 'arm' head.  Every translation unit gets its own copy (anonymous namespace).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// sin and cos of x in f32 to 1.5 ulp (|error| <= 9e-8) for |x| < 1e5: three-constant Cody-Waite reduction to [-pi/4, pi/4] by
// fma, degree-7 / degree-8 polynomials, quadrant fix-up by sign bits -- 22 VALU instructions and no scratch (the library sincosf
// carries a Payne-Hanek path with a stack array).  Pre-activations here are O(10): pixel coordinates <= 4096 times O(0.1) weights.
__device__ __forceinline__ void sincos_cw(float x, float& s_out, float& c_out) {
  const float k = __builtin_rintf(x * 0.6366197466850281f);
  float r = __builtin_fmaf(k, -1.5707963705062866f, x);
  r = __builtin_fmaf(k, 4.371138828673793e-08f, r);
  r = __builtin_fmaf(k, 1.7151245100058819e-15f, r);
  const float r2 = r * r;
  const float ps = __builtin_fmaf(__builtin_fmaf(-0.00019587950373534113f, r2, 0.008332748897373676f), r2, -0.166666641831398f);
  const float pc = __builtin_fmaf(__builtin_fmaf(2.4547991415602155e-05f, r2, -0.001388830365613103f), r2, 0.0416666641831398f);
  const float s = __builtin_fmaf(r * r2, ps, r);
  const float c = __builtin_fmaf(r2 * r2, pc, __builtin_fmaf(r2, -0.5f, 1.0f));
  const unsigned q = (unsigned)(int)k;
  const bool swap = q & 1u;
  const float ss = swap ? c : s, cc = swap ? s : c;
  s_out = __uint_as_float(__float_as_uint(ss) ^ ((q & 2u) << 30));
  c_out = __uint_as_float(__float_as_uint(cc) ^ (((q + 1u) & 2u) << 30));
}

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_add_f(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, BANK_MASK, true);
  return v + __builtin_bit_cast(float, moved);
}
__device__ __forceinline__ float wave_sum_lane63(float v) {   // fixed-order tree; the total is in lane 63
  v = dpp_add_f<0x111>(v);        // row_shr:1
  v = dpp_add_f<0x112>(v);        // row_shr:2
  v = dpp_add_f<0x114>(v);        // row_shr:4
  v = dpp_add_f<0x118>(v);        // row_shr:8
  v = dpp_add_f<0x142, 0xa>(v);   // row_bcast:15 into rows 1, 3
  v = dpp_add_f<0x143, 0xc>(v);   // row_bcast:31 into rows 2, 3
  return v;
}

struct ArmHead {       // mymodels/mlps.py:233-236 ('arm') and inverse_img_w_mi.py:493-496
  const float* start;  // [M, lds]: start_arm, the network's colour input (columns 0..4)
  int lds;
  float* th;           // [M, 8]: tanh(x), kept for the backward
  float* map_a;        // [M, 3] | null: clamp(1.3 tanh(x) + start, 0, 1)[0:3]
  float* map_r;        // [M]    | null: clamp(...)[3] * 0.93 + 0.07
  float* map_m;        // [M]    | null: clamp(...)[4]
};

// the 'arm' head on the five outputs v of row m (mymodels/mlps.py:231-233, inverse_img_w_mi.py:493-496)
__device__ __forceinline__ void arm_head_store(const ArmHead& h, long m, const float v[5]) {
  float y[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const float t = tanhf(v[j]);
    h.th[m * 8 + j] = t;
    // x = 1.3 tanh(x) + img; x = x.clamp(0,1).detach() + x - x.detach() (mlps.py:232-233): the value of the straight-through
    // clamp is (clamp(x) + x) - x as rounded in fp32, not clamp(x) -- it can land one ulp outside [0,1], which the clamp
    // of :494-496 then gates.  Separate roundings as torch's (no fma).
    const float u = __fadd_rn(__fmul_rn(1.3f, t), h.start[m * h.lds + j]);
    y[j] = __fsub_rn(__fadd_rn(fminf(fmaxf(u, 0.f), 1.f), u), u);
  }
  if (h.map_a) { h.map_a[m * 3 + 0] = y[0]; h.map_a[m * 3 + 1] = y[1]; h.map_a[m * 3 + 2] = y[2]; }
  if (h.map_r) h.map_r[m] = __fadd_rn(__fmul_rn(y[3], 0.93f), 0.07f);
  if (h.map_m) h.map_m[m] = y[4];
}

// Gradients on two f16 pieces (round 5).  A loss gradient has no natural size, so a 128-row tile of G travels as 2^-e (p1 + p2) with ONE
// exponent e per tile -- the tile's largest |g| is brought to [2^13, 2^14) (f16 overflows at 2^16) -- taken from `tile_max`, which the
// kernel that produced G filled (an atomic max over bit patterns: order-free).  Elements within 2^-16 of their tile's largest keep the
// 2^-24 relative accuracy of two pieces; smaller ones are carried to an absolute 2^-39 of it.  Rows of one tile are 128 consecutive pixels.
__device__ __forceinline__ void block_scale(unsigned max_bits, float& scale, float& unscale) {
  const float mx = __uint_as_float(max_bits);
  int e = 0;
  if (mx > 0.f && mx < 3.0e38f) (void)__builtin_frexpf(mx, &e);        // mx = m 2^e, m in [0.5, 1)
  else e = 14;                                                       // no gradient at all (or not finite: it propagates as it is)
  scale = __builtin_ldexpf(1.0f, 14 - e);
  unscale = __builtin_ldexpf(1.0f, e - 14);
}
__device__ __forceinline__ float wave_max_lane63(float v) {
#define MATPBR_DPP_MAX(ctrl, rowmask) v = __builtin_fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, rowmask, 0xf, true)))
  MATPBR_DPP_MAX(0x111, 0xf);     // row_shr:1 ... (zeros shifted in: the operands are magnitudes)
  MATPBR_DPP_MAX(0x112, 0xf);
  MATPBR_DPP_MAX(0x114, 0xf);
  MATPBR_DPP_MAX(0x118, 0xf);
  MATPBR_DPP_MAX(0x142, 0xa);     // row_bcast:15 into rows 1, 3
  MATPBR_DPP_MAX(0x143, 0xc);     // row_bcast:31 into rows 2, 3
#undef MATPBR_DPP_MAX
  return v;
}


// One float per sine activation instead of two.  The backward pass needs cos(pre) of every hidden unit; sin and cos lie on the unit circle,
// so |cos| = sqrt(1 - sin^2) and only its sign is missing: the forward epilogue writes it into the LAST MANTISSA BIT of the sine it
// stores (the stored sine moves by at most one ulp, the size of the rounding it already carries) and stops writing the cosines
// (268 MB per 256-wide layer at 512 x 512, a third of the layer's traffic).  The backward epilogue rebuilds the cosine from the stored
// sine.  Cost: where |cos| is small the square root amplifies the sine's rounding, |error| ~ 1.2e-7 / |cos| (1e-5 at |cos| = 0.01, 1.3 %
// of the units; rms relative error of a layer's cosines ~ 1e-5): the PRODUCTS stay f32-accurate, the cos factor of the backward pass is
// good to five digits (tests/test_gpu_parity.py::test_sines_that_carry_the_sign_of_their_cosine).
__device__ __forceinline__ float pack_cos_sign(float s, float c) {
  return __uint_as_float((__float_as_uint(s) & ~1u) | (__float_as_uint(c) >> 31));
}
// The packed sine in one go (round 5): a forward that keeps no cosines needs sin(x) and the SIGN of cos(x) only.  Reduction by whole
// multiples of pi to r in [-pi/2, pi/2] (k = rint(x / pi) by the magic-number add: the integer sits in the low mantissa bits of t):
// sin(x) = (-1)^k sin(r), sign(cos(x)) = (-1)^k -- one odd polynomial (degree 9, |error| 4.7e-9 on the interval), no quadrant select,
// no second polynomial.  14 VALU instructions against 22 + 3 for sincos_cw + pack_cos_sign; |sin error| <= 1.2e-7 for |x| < 1e5.
// k is the rounding of an f32 product: at |x| of a few hundred radians an argument within ~|x| 1e-7 of an odd multiple of pi/2 can be
// reduced to |r| slightly beyond pi/2, where cos(r) < 0 -- ROBUST (the first layer, whose arguments are pixel coordinates times
// weights) fixes the sign there with a compare and an add-with-carry; in the 256-wide layers (|x| of order 10) the affected |cos| is
// below 2e-6, inside the error of the rebuilt cosine (cos_from_packed_sin).
template <bool ROBUST = false>
__device__ __forceinline__ float sin_packed(float x) {
  const float t = __builtin_fmaf(x, 0.3183098861837907f, 12582912.0f);
  const float kf = t - 12582912.0f;
  float r = __builtin_fmaf(kf, -3.1415927410125732f, x);
  r = __builtin_fmaf(kf, 8.742277657347586e-08f, r);
  const float r2 = r * r;
  float p = __builtin_fmaf(2.5997510419983882e-06f, r2, -0.0001980647793971002f);
  p = __builtin_fmaf(p, r2, 0.008333015255630016f);
  p = __builtin_fmaf(p, r2, -0.16666656732559204f);
  const float s = __builtin_fmaf(r * r2, p, r);
  unsigned kb = __float_as_uint(t);
  const unsigned sb = __float_as_uint(s) ^ (kb << 31);
  if (ROBUST) kb ^= __builtin_fabsf(r) > 1.5707963705062866f ? 1u : 0u;
  return __uint_as_float((kb & 1u) | (sb & ~1u));                    // v_bfi_b32
}
__device__ __forceinline__ float cos_from_packed_sin(float sp) {
  const float c = __builtin_amdgcn_sqrtf(__builtin_fmaxf(__builtin_fmaf(-sp, sp, 1.0f), 0.0f));
  return __uint_as_float(__float_as_uint(c) | (__float_as_uint(sp) << 31));
}
__device__ __forceinline__ float4 cos_from_packed_sin(float4 v) {
  return make_float4(cos_from_packed_sin(v.x), cos_from_packed_sin(v.y), cos_from_packed_sin(v.z), cos_from_packed_sin(v.w));
}


typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {   // v_cvt_pk_bf16_f32: round-to-nearest-even, lo -> bits 15:0
  const f32x2v v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
// (x0, x1) -> three packed bf16 pairs with x = p1 + p2 + p3 exactly (up to the last piece's rounding, 2^-25 |x|)
__device__ __forceinline__ void split3(float x0, float x1, unsigned& p1, unsigned& p2, unsigned& p3) {
  p1 = cvt_pk_bf16(x0, x1);
  const float r0 = x0 - __uint_as_float(p1 << 16), r1 = x1 - __uint_as_float(p1 & 0xffff0000u);
  p2 = cvt_pk_bf16(r0, r1);
  const float s0 = r0 - __uint_as_float(p2 << 16), s1 = r1 - __uint_as_float(p2 & 0xffff0000u);
  p3 = cvt_pk_bf16(s0, s1);
}

// Two f16 pieces per f32 (round 5, the FORWARD layers: NPROD == 3).  f16 keeps 11 significant bits, and a rounded piece leaves a SIGNED
// remainder of at most half its last place: x - p1 is below 2^-12 |x|, its own rounding below 2^-24 |x| -- two f16 pieces carry an f32
// operand to the size of its own rounding, where bf16 (8 bits) needs three.  Three products (p1 q1 + p1 q2 + p2 q1; the dropped p2 q2
// is 2^-24 of the product) then do the work of six: half the matrix time, two thirds of the weight bytes, 6 instead of 11 split
// instructions per pair.  What f16 lacks is exponent range (normal down to 6.1e-5, subnormal spacing 6e-8): the operands must be of
// order one.  Activations are sines and the x0 tail (coordinates, colours): |x| <= 65504 is the documented limit, and below 2^-14 a
// second piece is subnormal, i.e. carried to an ABSOLUTE 3e-8 -- the size of an f32 rounding at 0.5.  Weights (~ +-1/16) are cut as
// 256 w (exact) and the accumulator is scaled back by 2^-8 in the epilogue (one fma with the bias): |w| < 255.  Measured against fp64
// (tests/test_gpu_parity.py::test_two_piece_f16_forward_layers_are_f32_accurate): representation + dropped term 2e-8 rms at K = 256
// against 1.2e-7 rms of the f32 accumulation every f32 kernel carries -- the error of the layer is that of the exact-f32 kernels.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
constexpr float kF16WScale = 256.0f, kF16WUnscale = 0.00390625f;
__device__ __forceinline__ unsigned cvt_pk_f16(float lo, float hi) {     // round-to-nearest-even (v_cvt_pk_f16_f32 / two v_cvt_f16_f32)
  const f32x2v v = {lo, hi};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2v));
}
// The remainders x - (float)p1 by v_fma_mix_f32 (an f16 half as an operand of an f32 fma: exact, as the conversion and the subtraction it
// replaces): four instructions per pair where the compiler's form has five, one of them a packed f32 add (twice the issue cost beside MFMAs)
__device__ __forceinline__ void split2h(float x0, float x1, unsigned& p1, unsigned& p2) {
  p1 = cvt_pk_f16(x0, x1);
  float r0, r1;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(p1), "v"(x0));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(p1), "v"(x1));
  p2 = cvt_pk_f16(r0, r1);
}


__device__ __forceinline__ unsigned lds_byte_address(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
// three / four consecutive 1 KB pieces under one M0: the instruction offset advances the global AND the LDS address
__device__ __forceinline__ void glds16_x3(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:2048\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16_x4(const void* sbase, unsigned voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
               "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds16_x4v(const void* sbase, unsigned v0, unsigned v1, unsigned v2, unsigned v3, unsigned lds_dst) {   // v_j: the lane's offset of piece j MINUS 1024 j
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %5\n\tglobal_load_lds_dwordx4 %2, %5 offset:1024\n\t"
               "global_load_lds_dwordx4 %3, %5 offset:2048\n\tglobal_load_lds_dwordx4 %4, %5 offset:3072\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase), "s"(lds_dst) : "memory");
}

}  // namespace
