// Shared device/host helpers for libwanq_hip (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/wanq_hip.h"

namespace wanq {

// ---------------------------------------------------------------- host-side error plumbing
void set_error(const char* fmt, ...);  // runtime.hip

#define WANQ_REQUIRE(cond, code, ...)   \
  do {                                  \
    if (!(cond)) {                      \
      ::wanq::set_error(__VA_ARGS__);   \
      return (code);                    \
    }                                   \
  } while (0)

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return WANQ_E_LAUNCH;
  }
  return WANQ_OK;
}

// rowwise.hip: scale-only form of the transform entry points (had_k == 0)
int premul_quant_rows(bool ln, const void* x, int x_dtype, const void* gamma, const void* mshift, const void* mscale,
                      int64_t mod_stride, int64_t rows_per_batch, float eps, const float* premul, void* out_fp, int out_dtype,
                      int8_t* q, void* scale, void* sum, int vec_dtype, int64_t rows, int cols, hipStream_t st, const char* what);

// rotate140.hip: the n = 8960 = 140 x 64 transform (+ per-token quantiser)
int rotate140_rows(const void* x, int x_dtype, const float* premul, void* out_fp, int out_dtype, int8_t* q, void* scale, void* sum,
                   int vec_dtype, int64_t rows, hipStream_t st, const char* what);

// rotate108.hip: the n = 13824 = 108 x 128 transform (+ per-token quantiser); repo-defined, see the file's header
int rotate108_rows(const void* x, int x_dtype, const float* premul, void* out_fp, int out_dtype, int8_t* q, void* scale, void* sum,
                   int vec_dtype, int64_t rows, hipStream_t st, const char* what);

inline bool is_fp(int dt) { return dt == WANQ_F16 || dt == WANQ_BF16 || dt == WANQ_F32; }
inline bool is_vec(int dt) { return dt == WANQ_F16 || dt == WANQ_F32; }

// ---------------------------------------------------------------- device: wave64 lane exchanges and reductions
// Value of lane ^ MASK for MASK in {1, 2, 4, 8}: DPP operand modifiers of the vector ALU (hipcc folds them into the consuming
// add / max: v_add_f32_dpp ...).  The ds_swizzle / ds_bpermute forms they replace go through the LDS crossbar: 100-300 cycles of
// latency per step under load, which a dependent reduction chain (six steps) or a butterfly stage pays in full; SQ counters of
// the transform kernel had 46 % of its wave cycles parked behind them (profiles/r03_h_rowwise_sq.csv).
//   xor 1, 2: quad_perm;  xor 8: row_ror:8 (a rotation by half a 16-lane row);  xor 4: row_shl:4 for the lanes whose bit 2 is
//   clear (banks 0, 2) + row_shr:4 for the others (banks 1, 3), two moves into one register.
template <int MASK>
__device__ __forceinline__ int lane_xor_dpp(int v) {
  static_assert(MASK == 1 || MASK == 2 || MASK == 4 || MASK == 8, "DPP lane exchange: masks inside a 16-lane row");
  if constexpr (MASK == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true);        // quad_perm:[1,0,3,2]
  else if constexpr (MASK == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true);   // quad_perm:[2,3,0,1]
  else if constexpr (MASK == 8) return __builtin_amdgcn_update_dpp(0, v, 0x128, 0xf, 0xf, true);  // row_ror:8
  else {
    const int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xf, 0x5, false);                      // row_shl:4, banks 0 and 2
    return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xf, 0xA, false);                             // row_shr:4, banks 1 and 3
  }
}
template <int MASK>
__device__ __forceinline__ float lane_xor_dpp(float v) { return __int_as_float(lane_xor_dpp<MASK>(__float_as_int(v))); }

// v_permlane16_swap / v_permlane32_swap of a register with itself: .x = the value of the pair's LOWER member (even 16-lane row /
// lanes 0-31), .y = of the UPPER member, in both lanes of every (lane, lane ^ 16) / (lane, lane ^ 32) pair
__device__ __forceinline__ uint2 pair16(int v) {
  const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);
  return make_uint2(r[0], r[1]);
}
__device__ __forceinline__ uint2 pair32(int v) {
  const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);
  return make_uint2(r[0], r[1]);
}

// Wave reductions in the order 32, 16, 8, 4, 2, 1 (the operand pairs, and with them every rounding, are those of the
// __shfl_xor butterflies these replace: bit-identical results)
struct OpSum { template <typename T> static __device__ __forceinline__ T f(T a, T b) { return a + b; } };
struct OpMax { static __device__ __forceinline__ float f(float a, float b) { return fmaxf(a, b); } };
struct OpMin { static __device__ __forceinline__ float f(float a, float b) { return fminf(a, b); } };
template <typename OP>
__device__ __forceinline__ float wave_reduce(float v) {
  uint2 r = pair32(__float_as_int(v));
  v = OP::f(__uint_as_float(r.x), __uint_as_float(r.y));
  r = pair16(__float_as_int(v));
  v = OP::f(__uint_as_float(r.x), __uint_as_float(r.y));
  v = OP::f(v, lane_xor_dpp<8>(v));
  v = OP::f(v, lane_xor_dpp<4>(v));
  v = OP::f(v, lane_xor_dpp<2>(v));
  v = OP::f(v, lane_xor_dpp<1>(v));
  return v;
}
__device__ __forceinline__ float wave_max(float v) { return wave_reduce<OpMax>(v); }
__device__ __forceinline__ float wave_min(float v) { return wave_reduce<OpMin>(v); }
__device__ __forceinline__ float wave_sum(float v) { return wave_reduce<OpSum>(v); }
__device__ __forceinline__ int wave_sum(int v) {
  uint2 r = pair32(v);
  v = (int)r.x + (int)r.y;
  r = pair16(v);
  v = (int)r.x + (int)r.y;
  v += lane_xor_dpp<8>(v);
  v += lane_xor_dpp<4>(v);
  v += lane_xor_dpp<2>(v);
  v += lane_xor_dpp<1>(v);
  return v;
}

// ---------------------------------------------------------------- device: typed 8-element access
struct F16 {};
struct BF16 {};
struct F32 {};

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t b) { return __uint_as_float(b << 16); }

template <typename T>
struct Io;

template <>
struct Io<F16> {
  static constexpr int kBytes = 2;
  __device__ static __forceinline__ void load8(const void* base, int64_t elem, float (&v)[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(static_cast<const char*>(base) + elem * 2);
    const __half2* h = reinterpret_cast<const __half2*>(&r);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float2 f = __half22float2(h[i]);
      v[2 * i] = f.x;
      v[2 * i + 1] = f.y;
    }
  }
  __device__ static __forceinline__ void store8(void* base, int64_t elem, const float (&v)[8]) {
    uint4 r;
    __half2* h = reinterpret_cast<__half2*>(&r);
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = __floats2half2_rn(v[2 * i], v[2 * i + 1]);
    *reinterpret_cast<uint4*>(static_cast<char*>(base) + elem * 2) = r;
  }
};

template <>
struct Io<BF16> {
  static constexpr int kBytes = 2;
  __device__ static __forceinline__ void load8(const void* base, int64_t elem, float (&v)[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(static_cast<const char*>(base) + elem * 2);
    const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[2 * i] = __uint_as_float(w[i] << 16);
      v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  __device__ static __forceinline__ void store8(void* base, int64_t elem, const float (&v)[8]) {
    uint32_t w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __hip_bfloat16 lo = __float2bfloat16(v[2 * i]);      // plain casts: RNE, NaN-preserving
      const __hip_bfloat16 hi = __float2bfloat16(v[2 * i + 1]);
      w[i] = (uint32_t)(*reinterpret_cast<const uint16_t*>(&lo)) |
             ((uint32_t)(*reinterpret_cast<const uint16_t*>(&hi)) << 16);
    }
    *reinterpret_cast<uint4*>(static_cast<char*>(base) + elem * 2) = make_uint4(w[0], w[1], w[2], w[3]);
  }
};

template <>
struct Io<F32> {
  static constexpr int kBytes = 4;
  __device__ static __forceinline__ void load8(const void* base, int64_t elem, float (&v)[8]) {
    const float4* p = reinterpret_cast<const float4*>(static_cast<const char*>(base) + elem * 4);
    const float4 a = p[0], b = p[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
  __device__ static __forceinline__ void store8(void* base, int64_t elem, const float (&v)[8]) {
    float4* p = reinterpret_cast<float4*>(static_cast<char*>(base) + elem * 4);
    p[0] = make_float4(v[0], v[1], v[2], v[3]);
    p[1] = make_float4(v[4], v[5], v[6], v[7]);
  }
};

// per-token / per-channel vectors: dtype chosen at run time (wave-uniform branch)
__device__ __forceinline__ float vec_load(const void* p, int dt, int64_t i) {
  return dt == WANQ_F32 ? static_cast<const float*>(p)[i] : __half2float(static_cast<const __half*>(p)[i]);
}
__device__ __forceinline__ void vec_store(void* p, int dt, int64_t i, float v) {
  if (dt == WANQ_F32) static_cast<float*>(p)[i] = v;
  else static_cast<__half*>(p)[i] = __float2half_rn(v);
}

__device__ __forceinline__ float gelu_tanh_f32(float x) {
  // 0.5 x (1 + tanh(0.79788456 (x + 0.044715 x^3)))   (reference fused.cu:22-26, in fp32)
  const float inner = 0.79788456f * (x + 0.044715f * x * x * x);
  return 0.5f * x * (1.0f + tanhf(inner));
}

// Same function through the identity 0.5 (1 + tanh u) = 1 / (1 + e^{-2u}): two transcendental instructions (v_exp_f32,
// v_rcp_f32, 1 ulp each) and five plain ones instead of libm's branchy tanhf (~35).  Relative error < 3e-6 against the
// form above, i.e. invisible after the fp16/bf16 rounding of a GEMM output; saturates correctly (e^{+inf} -> x/inf = -0,
// e^{-inf} -> x).  Used in the GEMM epilogue, where 293 M evaluations per FFN cost more than a tenth of the kernel.
__device__ __forceinline__ float gelu_tanh_fast_f32(float x) {
  // -2 u log2(e) = x (c1 + c3 x^2),  c1 = -2*0.79788456*log2(e),  c3 = c1*0.044715
  const float t = x * fmaf(x * x, -0.10294324f, -2.3022082f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(t));
}

// q = clamp(rne(x / s)) with IEEE fp32 division semantics at multiply cost.  t = x*inv is within 1.8e-7*|t| of
// fl(x/s); rint(t) can differ from rint(fl(x/s)) only when t sits that close to a .5 boundary, and then the true
// division is evaluated (rare: ~1e-5 of the elements).  Eight elements at a time so that the fallback costs one
// wave-level branch per chunk: per element the common path is mul, rndne, sub, fma, cmp, med3, cvt.
__device__ __forceinline__ void quant8_div_rne(const float (&x)[8], float s, float inv, int (&q)[8]) {
  float r[8];
  bool near = false;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float t = x[j] * inv;
    r[j] = rintf(t);
    near |= fabsf(t - r[j]) >= fmaf(-4e-7f, fabsf(t), 0.5f);
  }
  if (near) {
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = rintf(x[j] / s);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) q[j] = (int)__builtin_amdgcn_fmed3f(r[j], -128.f, 127.f);
}

// The same quantiser for values KNOWN to satisfy |x / s| <= 127.5 (a dynamic per-row scale s = amax / 127, or its eps floor),
// producing the packed bytes directly, in 4 vector instructions per element + 3 per four for the pack (the form above: 7 + 3):
//   u = fma(x, inv, M), M = 1.5 * 2^23: the EXACT product x * inv rounded once to the integer grid of [2^23, 2^24) -- the low byte
//       of u's bit pattern is the two's-complement code (rounding and float -> int in one instruction, no clamp needed);
//   r = u - M (exact), d = fma(x, inv, -r): the exact residual, |d| <= 0.5;
//   near: |d| >= 0.5 - 5.1e-5.  RN(x * inv exact) can differ from rint(fl(x / s)) only when x / s lies within
//       127 * 2^-24 (inv's rounding) + 2^-18 (the quotient's) = 1.2e-5 of a .5 boundary: the threshold keeps a 4x margin, and
//       the flagged chunk (about 1e-4 of the elements) takes the true division, as above.
// Bit-identical codes to quant8_div_rne on that domain (tests/test_gpu_rowwise.py: ties, near-ties, eps rows, golden vectors).
constexpr float WANQ_QMAGIC = 12582912.0f;
template <int N>
__device__ __forceinline__ void quantN_pack_rne(const float (&x)[N], float s, float inv, uint32_t (&packed)[N / 4]) {
  static_assert(N % 4 == 0, "four codes per dword");
  float u[N];
  bool near = false;
#ifndef WANQ_QUANT_SCALAR
  // on element PAIRS (v_pk_fma_f32 / v_pk_add_f32: the same IEEE operations per half, bit-identical codes, 2.5 instead of 4
  // vector instructions per element in front of the byte packing)
  typedef float v2f __attribute__((ext_vector_type(2)));
  const v2f inv2 = {inv, inv}, magic2 = {WANQ_QMAGIC, WANQ_QMAGIC};
#pragma unroll
  for (int j = 0; j < N; j += 2) {
    const v2f x2 = {x[j], x[j + 1]};
    const v2f u2 = __builtin_elementwise_fma(x2, inv2, magic2);
    const v2f d2 = __builtin_elementwise_fma(x2, inv2, -(u2 - magic2));
    u[j] = u2.x;
    u[j + 1] = u2.y;
    near |= fabsf(d2.x) >= 0.4999488f;
    near |= fabsf(d2.y) >= 0.4999488f;
  }
#else
#pragma unroll
  for (int j = 0; j < N; ++j) {
    u[j] = fmaf(x[j], inv, WANQ_QMAGIC);
    const float r = u[j] - WANQ_QMAGIC;
    near |= fabsf(fmaf(x[j], inv, -r)) >= 0.4999488f;
  }
#endif
  if (near) {
#pragma unroll
    for (int j = 0; j < N; ++j) u[j] = rintf(x[j] / s) + WANQ_QMAGIC;
  }
#pragma unroll
  for (int g = 0; g < N / 4; ++g) {
    const uint32_t b0 = __float_as_uint(u[4 * g]), b1 = __float_as_uint(u[4 * g + 1]), b2 = __float_as_uint(u[4 * g + 2]), b3 = __float_as_uint(u[4 * g + 3]);
    // bytes (b0.0, b1.0, 0, 0) | (0, 0, b2.0, b3.0)
    packed[g] = __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u) | __builtin_amdgcn_perm(b3, b2, 0x04000c0cu);
  }
}

// quantN_pack_rne for x = fl(y * c) WITHOUT forming x: the codes of rne(fl(y * c) / s), c > 0 a per-launch constant (the
// 1 / sqrt(n) of the Hadamard transform), from u = fma(y, fl(c * inv), M).  The exact y * fl(c * inv) lies within
// 127 * 3 * 2^-24 + 2^-18 = 2.7e-5 of fl(fl(y * c) / s), inside the same threshold; the flagged chunk evaluates the
// reference's two operations literally.  (The row maximum commutes with the scaling: fl(max|y| * c) == max fl(|y| * c).)
template <int N>
__device__ __forceinline__ void quantN_pack_rne_pre(const float (&y)[N], float c, float s, float cinv, uint32_t (&packed)[N / 4]) {
  static_assert(N % 4 == 0, "four codes per dword");
  float u[N];
  bool near = false;
#ifndef WANQ_QUANT_SCALAR
  typedef float v2f __attribute__((ext_vector_type(2)));  // on element pairs, as in quantN_pack_rne
  const v2f cinv2 = {cinv, cinv}, magic2 = {WANQ_QMAGIC, WANQ_QMAGIC};
#pragma unroll
  for (int j = 0; j < N; j += 2) {
    const v2f y2 = {y[j], y[j + 1]};
    const v2f u2 = __builtin_elementwise_fma(y2, cinv2, magic2);
    const v2f d2 = __builtin_elementwise_fma(y2, cinv2, -(u2 - magic2));
    u[j] = u2.x;
    u[j + 1] = u2.y;
    near |= fabsf(d2.x) >= 0.4999488f;
    near |= fabsf(d2.y) >= 0.4999488f;
  }
#else
#pragma unroll
  for (int j = 0; j < N; ++j) {
    u[j] = fmaf(y[j], cinv, WANQ_QMAGIC);
    const float r = u[j] - WANQ_QMAGIC;
    near |= fabsf(fmaf(y[j], cinv, -r)) >= 0.4999488f;
  }
#endif
  if (near) {
#pragma unroll
    for (int j = 0; j < N; ++j) u[j] = rintf((y[j] * c) / s) + WANQ_QMAGIC;
  }
#pragma unroll
  for (int g = 0; g < N / 4; ++g) {
    const uint32_t b0 = __float_as_uint(u[4 * g]), b1 = __float_as_uint(u[4 * g + 1]), b2 = __float_as_uint(u[4 * g + 2]), b3 = __float_as_uint(u[4 * g + 3]);
    packed[g] = __builtin_amdgcn_perm(b1, b0, 0x0c0c0400u) | __builtin_amdgcn_perm(b3, b2, 0x04000c0cu);
  }
}

// 4 ints in [-128,127] -> packed bytes: two saturating i32->i16 packs + one byte permute
__device__ __forceinline__ uint32_t pack4_i8_fast(int a, int b, int c, int d) {
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const s16x2 lo = __builtin_amdgcn_cvt_pk_i16(a, b), hi = __builtin_amdgcn_cvt_pk_i16(c, d);
  return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, hi), __builtin_bit_cast(uint32_t, lo), 0x06040200u);
}

__device__ __forceinline__ int quant_div_rne(float x, float s, float inv) {
  float t = x * inv;
  float r = rintf(t);
  if (fabsf(t - r) >= fmaf(-4e-7f, fabsf(t), 0.5f)) r = rintf(x / s);
  return (int)__builtin_amdgcn_fmed3f(r, -128.f, 127.f);
}

__device__ __forceinline__ uint32_t pack4_i8(int a, int b, int c, int d) {
  return (uint32_t)(a & 0xff) | ((uint32_t)(b & 0xff) << 8) | ((uint32_t)(c & 0xff) << 16) | ((uint32_t)(d & 0xff) << 24);
}

}  // namespace wanq
