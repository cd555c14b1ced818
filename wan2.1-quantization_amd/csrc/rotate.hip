// ViDiT / QuaRot activation transform fused with (LayerNorm + modulate and) the per-token int8 quantiser:
//     y = hadU(x * premul),   hadU = (H_K' (x) H_128) / fp32-sqrt(n),   n = K' * 128
// (ViDiT-Q/quant_utils/qdiff/viditq/viditq_quant_layer.py:60-73 does x*mask -> x.double() @ R with a dense fp64 matrix;
// row i of R is sign_i * hadU(e_i), quarot_utils.py:158-192, so premul = channel_mask * signs.)
//
// Register-resident: NO LDS memory and NO barriers.  A 128-wide block is spread over LB = 128 / EPL lanes (EPL = 8 or 4
// consecutive elements per lane); a row is owned by LPR = LB * Q lanes of ONE wave, and lane (q, c) holds, for each of its KIN
// blocks r, the elements  e = (q*KIN + r)*128 + EPL*c + j :
//   * the 128-point Walsh-Hadamard transform of a block = log2(EPL) in-register stages (bits of j) + log2(LB) lane-exchange
//     stages (bits of c: ds_swizzle / ds_bpermute move data through the LDS crossbar but touch no LDS memory);
//   * the +-1 mixing ACROSS blocks, H_K' = S_Q (x) M_KIN, is lane-local for M_KIN -- every block a lane needs is already in
//     its registers -- plus one or two more lane-exchange stages for the Sylvester factor S_Q.
// M_KIN is the reference's get_hadK table for the width (quarot_utils.py:100-155), a fixed function of n: Sylvester for powers
// of two, the Paley-I matrix of order 12 (n = 1536) or 20 (n = 5120 = (H_2 (x) P_20) (x) H_128); both are generated at compile
// time from quadratic residues, so the mix is straight-line adds with no sign loads.
//
// This replaces the round-1 form that staged the row in an LDS buffer written by one lane and read back by other lanes of
// the same wave: that hand-off was the one place to which a run-to-run deviation under two processes per GPU had been
// traced; with the row in registers there is no such hand-off left.
#include "wanq_common.h"

namespace wanq {

struct RotParams {
  const void* x;
  int x_dtype;
  const void* gamma;
  const void* mshift;
  const void* mscale;
  int64_t mod_stride;
  int64_t rows_per_batch;
  float eps;
  void* out_fp;
  int out_dtype;
  int vec_dtype;
  int64_t rows;
  int cols;
  float inv_div;  // 1 / fp32 sqrt(cols): the reference divides by torch.tensor(n).sqrt()
  int nsets;      // 1..3 (premul, q, scale, sum) sets computed from the same (normalised) row
  const float* premul[3];
  int8_t* q[3];
  void* scale[3];
  void* sum[3];
};

// ---- EPL (8 or 4) consecutive elements of one dtype <-> fp32 registers
template <typename T, int EPL>
struct IoN;
template <typename T>
struct IoN<T, 8> {
  __device__ static __forceinline__ void load(const void* b, int64_t e, float (&v)[8]) { Io<T>::load8(b, e, v); }
  __device__ static __forceinline__ void store(void* b, int64_t e, const float (&v)[8]) { Io<T>::store8(b, e, v); }
};
template <>
struct IoN<F32, 4> {
  __device__ static __forceinline__ void load(const void* b, int64_t e, float (&v)[4]) {
    const float4 a = *reinterpret_cast<const float4*>(static_cast<const float*>(b) + e);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  }
  __device__ static __forceinline__ void store(void* b, int64_t e, const float (&v)[4]) {
    *reinterpret_cast<float4*>(static_cast<float*>(b) + e) = make_float4(v[0], v[1], v[2], v[3]);
  }
};
template <>
struct IoN<BF16, 4> {
  __device__ static __forceinline__ void load(const void* b, int64_t e, float (&v)[4]) {
    const uint2 r = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(b) + e);
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
  }
  __device__ static __forceinline__ void store(void* b, int64_t e, const float (&v)[4]) {
    uint16_t h[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __hip_bfloat16 t = __float2bfloat16(v[i]);
      h[i] = *reinterpret_cast<const uint16_t*>(&t);
    }
    *reinterpret_cast<uint2*>(static_cast<uint16_t*>(b) + e) =
        make_uint2((uint32_t)h[0] | ((uint32_t)h[1] << 16), (uint32_t)h[2] | ((uint32_t)h[3] << 16));
  }
};
template <>
struct IoN<F16, 4> {
  __device__ static __forceinline__ void load(const void* b, int64_t e, float (&v)[4]) {
    const uint2 r = *reinterpret_cast<const uint2*>(static_cast<const __half*>(b) + e);
    const __half2* h = reinterpret_cast<const __half2*>(&r);
    const float2 a = __half22float2(h[0]), c = __half22float2(h[1]);
    v[0] = a.x; v[1] = a.y; v[2] = c.x; v[3] = c.y;
  }
  __device__ static __forceinline__ void store(void* b, int64_t e, const float (&v)[4]) {
    uint2 r;
    __half2* h = reinterpret_cast<__half2*>(&r);
    h[0] = __floats2half2_rn(v[0], v[1]);
    h[1] = __floats2half2_rn(v[2], v[3]);
    *reinterpret_cast<uint2*>(static_cast<__half*>(b) + e) = r;
  }
};

// value of lane ^ MASK: DPP for 1, 2, 4, 8 (wanq_common.h); 16: ds_swizzle bit mode; 32: ds_bpermute.  The butterfly stages and
// the reductions below do not use the 16 / 32 forms: they take both members of a pair from one v_permlane16/32_swap.
template <int MASK>
__device__ __forceinline__ float lane_xor(float v) {
  if constexpr (MASK < 16) return lane_xor_dpp<MASK>(v);
  else if constexpr (MASK < 32) return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (MASK << 10) | 0x1f));
  else return __shfl_xor(v, 32, 64);
}

// reductions over the LPR = 16, 32 or 64 lanes that own a row (order 1, 2, 4, 8, 16, 32: the operand pairs of the lane_xor
// butterflies this replaces)
template <int LPR, typename OP>
__device__ __forceinline__ float group_reduce(float v) {
  v = OP::f(v, lane_xor_dpp<1>(v)); v = OP::f(v, lane_xor_dpp<2>(v)); v = OP::f(v, lane_xor_dpp<4>(v)); v = OP::f(v, lane_xor_dpp<8>(v));
  if constexpr (LPR >= 32) {
    const uint2 r = pair16(__float_as_int(v));
    v = OP::f(__uint_as_float(r.x), __uint_as_float(r.y));
  }
  if constexpr (LPR >= 64) {
    const uint2 r = pair32(__float_as_int(v));
    v = OP::f(__uint_as_float(r.x), __uint_as_float(r.y));
  }
  return v;
}
template <int LPR>
__device__ __forceinline__ float group_sum(float v) { return group_reduce<LPR, OpSum>(v); }
template <int LPR>
__device__ __forceinline__ float group_max(float v) { return group_reduce<LPR, OpMax>(v); }
template <int LPR>
__device__ __forceinline__ int group_isum(int v) {
  v += lane_xor_dpp<1>(v); v += lane_xor_dpp<2>(v); v += lane_xor_dpp<4>(v); v += lane_xor_dpp<8>(v);
  if constexpr (LPR >= 32) {
    const uint2 r = pair16(v);
    v = (int)r.x + (int)r.y;
  }
  if constexpr (LPR >= 64) {
    const uint2 r = pair32(v);
    v = (int)r.x + (int)r.y;
  }
  return v;
}

// ---- the reference's Hadamard tables, generated (qdiff/quarot/quarot_utils.py paley_hadamard: first column +1, first row
// (+1, -1, ..., -1), core chi(a - b) off the diagonal and +1 on it; chi = quadratic character mod q)
constexpr int paley_chi(int q, int a) {
  a = ((a % q) + q) % q;
  if (a == 0) return 0;
  for (int t = 1; t < q; ++t)
    if ((t * t) % q == a) return 1;
  return -1;
}
template <int KIN>
struct MixTable {  // entry [a][b] of M_KIN for the non-Sylvester orders (KIN - 1 prime, = 3 mod 4)
  bool plus[KIN][KIN];
  constexpr MixTable() : plus() {
    for (int a = 0; a < KIN; ++a)
      for (int b = 0; b < KIN; ++b)
        plus[a][b] = (b == 0) ? true : (a == 0) ? false : (a == b) ? true : paley_chi(KIN - 1, (a - 1) - (b - 1)) > 0;
  }
};
constexpr bool is_pow2_c(int n) { return n > 0 && (n & (n - 1)) == 0; }

// hipcc's scheduler otherwise hoists every load and lane exchange of a phase to its start and keeps several copies of the row
// live (spills); a scheduling fence every two blocks bounds the working set, the other waves of the SIMD cover the latency.
#define ROT_FENCE(r) do { if (((r) & 1) == 1) __builtin_amdgcn_sched_barrier(0); } while (0)

// one lane-exchange butterfly stage on every element: v <- other + sgn * v, sgn = -1 on the lane whose bit is set
// (bit clear: v + other; bit set: other - v): one rounding, identical to the add / subtract form
// One lane-exchange butterfly stage at two vector instructions per element and no LDS traffic (bit 'MASK' of the lane
// clear: v + other; set: other - v -- the operands and the single rounding of the ds_swizzle + fma form: bit-identical):
//   MASK 1, 2  (inside a quad: bank masks cannot tell the two lanes of a pair apart)  t = +-v by v_cndmask, then
//              v_add_f32_dpp quad_perm.  The four t of a block are made before the four adds: a DPP instruction must not
//              read a register written by one of the two instructions in front of it (hipcc pads with s_nop 1 otherwise --
//              899 of them in the first DPP build of the q / k / v kernel, which reused one temporary);
//   MASK 4, 8  v_add_f32_dpp row_shl:MASK on the banks whose lanes have the bit clear + v_sub_f32_dpp row_shr:MASK on the
//              others, into one register (inline asm: the masked forms have no builtin; leading s_nop 1 = the hazard above
//              for inputs written just in front of the statement);
//   MASK 16, 32  two elements p, q at a time through v_permlane16/32_swap: swap(p, q) leaves (p_lo, q_lo) in one register
//              and (p_hi, q_hi) in the other (lo / hi = the pair's lower / upper member), their sum and difference are the
//              outputs (P_lo, Q_lo) and (P_hi, Q_hi), and a second swap sorts them back into P and Q.
template <int MASK, int KIN, int EPL>
__device__ __forceinline__ void lane_stage(float (&v)[KIN][EPL], int lane) {
  if constexpr (MASK <= 2) {
    const bool neg = (lane & MASK) != 0;
#pragma unroll
    for (int r = 0; r < KIN; ++r) {
      float t[EPL];
#pragma unroll
      for (int j = 0; j < EPL; ++j) t[j] = neg ? -v[r][j] : v[r][j];
      if constexpr (EPL == 4) asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
      else asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
#pragma unroll
      for (int j = 0; j < EPL; ++j) v[r][j] = t[j] + lane_xor_dpp<MASK>(v[r][j]);
      ROT_FENCE(r);
    }
  } else if constexpr (MASK <= 8) {
#pragma unroll
    for (int r = 0; r < KIN; ++r) {
#pragma unroll
      for (int j = 0; j < EPL; j += 4) {
        float o0, o1, o2, o3;
        if constexpr (MASK == 4)
          asm volatile("s_nop 1\n\t"
                       "v_add_f32_dpp %0, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_add_f32_dpp %1, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                       "v_add_f32_dpp %2, %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\tv_add_f32_dpp %3, %7, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                       "v_sub_f32_dpp %0, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\tv_sub_f32_dpp %1, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                       "v_sub_f32_dpp %2, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\tv_sub_f32_dpp %3, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa"
                       : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
                       : "v"(v[r][j]), "v"(v[r][j + 1]), "v"(v[r][j + 2]), "v"(v[r][j + 3]));
        else
          asm volatile("s_nop 1\n\t"
                       "v_add_f32_dpp %0, %4, %4 row_shl:8 row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %1, %5, %5 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                       "v_add_f32_dpp %2, %6, %6 row_shl:8 row_mask:0xf bank_mask:0x3\n\tv_add_f32_dpp %3, %7, %7 row_shl:8 row_mask:0xf bank_mask:0x3\n\t"
                       "v_sub_f32_dpp %0, %4, %4 row_shr:8 row_mask:0xf bank_mask:0xc\n\tv_sub_f32_dpp %1, %5, %5 row_shr:8 row_mask:0xf bank_mask:0xc\n\t"
                       "v_sub_f32_dpp %2, %6, %6 row_shr:8 row_mask:0xf bank_mask:0xc\n\tv_sub_f32_dpp %3, %7, %7 row_shr:8 row_mask:0xf bank_mask:0xc"
                       : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3)
                       : "v"(v[r][j]), "v"(v[r][j + 1]), "v"(v[r][j + 2]), "v"(v[r][j + 3]));
        v[r][j] = o0; v[r][j + 1] = o1; v[r][j + 2] = o2; v[r][j + 3] = o3;
      }
      ROT_FENCE(r);
    }
  } else {
#pragma unroll
    for (int r = 0; r < KIN; ++r) {
#pragma unroll
      for (int j = 0; j < EPL; j += 2) {
        const unsigned pb = __float_as_uint(v[r][j]), qb = __float_as_uint(v[r][j + 1]);
        const auto lohi = MASK == 16 ? __builtin_amdgcn_permlane16_swap(pb, qb, false, false) : __builtin_amdgcn_permlane32_swap(pb, qb, false, false);
        const float lo = __uint_as_float(lohi[0]), hi = __uint_as_float(lohi[1]);  // (p_lo, q_lo), (p_hi, q_hi)
        const float su = lo + hi, di = lo - hi;                                        // (P_lo, Q_lo), (P_hi, Q_hi)
        const auto pq = MASK == 16 ? __builtin_amdgcn_permlane16_swap(__float_as_uint(su), __float_as_uint(di), false, false)
                                   : __builtin_amdgcn_permlane32_swap(__float_as_uint(su), __float_as_uint(di), false, false);
        v[r][j] = __uint_as_float(pq[0]);
        v[r][j + 1] = __uint_as_float(pq[1]);
      }
      ROT_FENCE(r);
    }
  }
}

template <int KIN, int Q, int EPL>
__device__ __forceinline__ void hadamard_regs(float (&v)[KIN][EPL], int lane) {  // UNSCALED: the caller owns the 1 / sqrt(n)
  constexpr int LB = 128 / EPL;
  // H_128 inside every block: bits of j ...
#pragma unroll
  for (int r = 0; r < KIN; ++r)
#pragma unroll
    for (int h = 1; h < EPL; h <<= 1)
#pragma unroll
      for (int j = 0; j < EPL; ++j)
        if (!(j & h)) {
          const float a = v[r][j], b = v[r][j | h];
          v[r][j] = a + b;
          v[r][j | h] = a - b;
        }
  __builtin_amdgcn_sched_barrier(0);
  // ... then bits of c (lane bits 0 .. log2(LB)-1)
#if (defined(WANQ_ROT_ABLATE_LANE) || defined(WANQ_ROT_ABLATE_MIX) || defined(WANQ_ROT_ABLATE_QUANT)) && !defined(WANQ_ALLOW_ABLATIONS)
#error "WANQ_ROT_ABLATE_* build deliberately wrong kernels (timing ablations): add -DWANQ_ALLOW_ABLATIONS, never in build.py's library"
#endif
#ifndef WANQ_ROT_ABLATE_LANE  // (WANQ_ROT_ABLATE_*: timing probes only, tools/probes/rotate_ablate.py; never defined in the product build)
  lane_stage<1, KIN, EPL>(v, lane);
  lane_stage<2, KIN, EPL>(v, lane);
  lane_stage<4, KIN, EPL>(v, lane);
  lane_stage<8, KIN, EPL>(v, lane);
  if constexpr (LB >= 32) lane_stage<16, KIN, EPL>(v, lane);
#endif
  __builtin_amdgcn_sched_barrier(0);
  // M_KIN across this lane's blocks
  if constexpr (is_pow2_c(KIN)) {
#pragma unroll
    for (int h = 1; h < KIN; h <<= 1)
#pragma unroll
      for (int r = 0; r < KIN; ++r)
        if (!(r & h)) {
#pragma unroll
          for (int j = 0; j < EPL; ++j) {
            const float a = v[r][j], b = v[r | h][j];
            v[r][j] = a + b;
            v[r | h][j] = a - b;
          }
        }
  } else {
#ifndef WANQ_ROT_ABLATE_MIX
    // Paley-I structure (column 0 all +1, row 0 = (+1, -1, ..., -1), core +1 on the diagonal and chi(a - b) off it): with
    // S = x_1 + ... + x_{K-1} and P_a = the sum of the x_b, b >= 1, b != a, with chi(a - b) = +1 ((K - 2) / 2 terms),
    //     y_0 = x_0 - S,     y_a = x_0 + x_a + (2 P_a - (S - x_a)) = (x_0 - S) + 2 (x_a + P_a)
    // -- 77 additions per column at K = 12 (209 at K = 20) instead of K (K - 1) = 132 (380).
    constexpr MixTable<KIN> tab{};
#pragma unroll
    for (int j = 0; j < EPL; ++j) {
      float t[KIN];
      float S = v[1][j];
#pragma unroll
      for (int b = 2; b < KIN; ++b) S += v[b][j];
      const float base = v[0][j] - S;
      t[0] = base;
#pragma unroll
      for (int a = 1; a < KIN; ++a) {
        float P = v[a][j];
#pragma unroll
        for (int b = 1; b < KIN; ++b)
          if (b != a && tab.plus[a][b]) P += v[b][j];
        t[a] = fmaf(2.0f, P, base);
      }
#pragma unroll
      for (int a = 0; a < KIN; ++a) v[a][j] = t[a];
      __builtin_amdgcn_sched_barrier(0);
    }
#endif
  }
  // S_Q (Sylvester) across the lane groups of the row
  if constexpr (Q >= 2) lane_stage<LB, KIN, EPL>(v, lane);
  if constexpr (Q >= 4) lane_stage<2 * LB, KIN, EPL>(v, lane);
}

// One dtype branch per ROW, not per chunk: inside a branch the KIN loads issue back to back (a branch per chunk makes hipcc
// wait for each 16-bit load before it issues the next).
template <int KIN, int EPL>
__device__ __forceinline__ void rot_load_row(const void* x, int dt, int64_t elem0, float (&v)[KIN][EPL]) {
  if (dt == WANQ_F32) {
#pragma unroll
    for (int r = 0; r < KIN; ++r) IoN<F32, EPL>::load(x, elem0 + 128 * r, v[r]);
  } else if (dt == WANQ_BF16) {
#pragma unroll
    for (int r = 0; r < KIN; ++r) IoN<BF16, EPL>::load(x, elem0 + 128 * r, v[r]);
  } else {
#pragma unroll
    for (int r = 0; r < KIN; ++r) IoN<F16, EPL>::load(x, elem0 + 128 * r, v[r]);
  }
}
template <int KIN, int EPL>
__device__ __forceinline__ void rot_store_row(void* out, int dt, int64_t elem0, const float (&v)[KIN][EPL]) {
  if (dt == WANQ_F32) {
#pragma unroll
    for (int r = 0; r < KIN; ++r) IoN<F32, EPL>::store(out, elem0 + 128 * r, v[r]);
  } else if (dt == WANQ_BF16) {
#pragma unroll
    for (int r = 0; r < KIN; ++r) IoN<BF16, EPL>::store(out, elem0 + 128 * r, v[r]);
  } else {
#pragma unroll
    for (int r = 0; r < KIN; ++r) IoN<F16, EPL>::store(out, elem0 + 128 * r, v[r]);
  }
}

// (x - mean) * rstd [* gamma] [* (1 + scale)] [+ shift]; gamma / scale / shift are fp32 (checked on the host: the
// modulation of the simulation path is fp32, wan/modules/model.py:327).  One uniform branch per optional operand around a
// loop over the blocks -- a branch per block and operand gives hipcc a control-flow graph it answers with spills.
template <int KIN, int EPL>
__device__ __forceinline__ void rot_normalise(const RotParams& p, int col0, int64_t mb, float mean, float rstd, float (&v)[KIN][EPL]) {
#pragma unroll
  for (int r = 0; r < KIN; ++r)
#pragma unroll
    for (int j = 0; j < EPL; ++j) v[r][j] = (v[r][j] - mean) * rstd;
  if (p.gamma) {
#pragma unroll
    for (int r = 0; r < KIN; ++r) {
      float g[EPL];
      IoN<F32, EPL>::load(p.gamma, col0 + 128 * r, g);
#pragma unroll
      for (int j = 0; j < EPL; ++j) v[r][j] *= g[j];
      ROT_FENCE(r);
    }
  }
  if (p.mscale) {
#pragma unroll
    for (int r = 0; r < KIN; ++r) {
      float sc[EPL];
      IoN<F32, EPL>::load(p.mscale, mb + col0 + 128 * r, sc);
#pragma unroll
      for (int j = 0; j < EPL; ++j) v[r][j] *= (1.0f + sc[j]);
      ROT_FENCE(r);
    }
  }
  if (p.mshift) {
#pragma unroll
    for (int r = 0; r < KIN; ++r) {
      float sh[EPL];
      IoN<F32, EPL>::load(p.mshift, mb + col0 + 128 * r, sh);
#pragma unroll
      for (int j = 0; j < EPL; ++j) v[r][j] += sh[j];
      ROT_FENCE(r);
    }
  }
}

// Waves per SIMD the register allocator must leave room for: the row (EPL * KIN floats) plus ~100 working registers.
constexpr int rot_waves_per_simd(int row_regs) { return row_regs <= 8 ? 4 : row_regs <= 48 ? 3 : 2; }  // (64 floats per lane + the DPP forms' temporaries spill at 168 registers)

// KIN blocks per lane, Q lane groups per row (K' = Q * KIN), EPL elements of a block per lane.  A MULTI launch produces the
// normalised row again from x for every set after the first (the wave has just read it: L1 / L2 hits) instead of keeping
// a second copy in registers.
template <int KIN, int Q, int EPL, bool LN, bool MULTI>
__global__ __launch_bounds__(256, rot_waves_per_simd(KIN* EPL * ((MULTI && KIN * EPL <= 48 && Q == 1) ? 2 : 1))) void rotate_kernel(const RotParams p) {
  constexpr int LB = 128 / EPL, LPR = LB * Q, RPW = 64 / LPR;
  static_assert(LPR <= 64, "a row must fit one wave");
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int c = lane & (LB - 1), q = (lane / LB) & (Q - 1), rw = lane / LPR;
  const int64_t row_raw = ((int64_t)blockIdx.x * 4 + wave) * RPW + rw;
  const bool live = row_raw < p.rows;
  const int64_t row = live ? row_raw : p.rows - 1;  // surplus lane groups recompute the last row and store nothing
  const int C = p.cols;
  const int64_t rbase_k = row * (int64_t)C;
  const int col0_k = q * KIN * 128 + c * EPL;  // + 128 r

  float v[KIN][EPL];
  float mean = 0.f, rstd = 1.f;
  const int64_t mb = LN ? (row / p.rows_per_batch) * p.mod_stride : 0;

  rot_load_row<KIN, EPL>(p.x, p.x_dtype, rbase_k + col0_k, v);
  if (LN) {
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < KIN; ++r)
#pragma unroll
      for (int j = 0; j < EPL; ++j) s += v[r][j];
    mean = group_sum<LPR>(s) / (float)C;
    float s2 = 0.f;
#pragma unroll
    for (int r = 0; r < KIN; ++r)
#pragma unroll
      for (int j = 0; j < EPL; ++j) {
        const float d = v[r][j] - mean;
        s2 += d * d;
      }
    rstd = 1.0f / sqrtf(group_sum<LPR>(s2) / (float)C + p.eps);
    rot_normalise<KIN, EPL>(p, col0_k, mb, mean, rstd, v);
  }

  // MULTI with few floats per lane (KIN * EPL <= 48): the normalised row is KEPT in a second register copy for outputs 2 and 3
  // instead of being re-read and re-normalised (ablation: 147 of the call's 236 us are load / LayerNorm / re-read / stores)
  constexpr bool KEEP = MULTI && KIN * EPL <= 48 && Q == 1;
  float vn[KEEP ? KIN : 1][KEEP ? EPL : 1];
  if constexpr (KEEP) {
#pragma unroll
    for (int r = 0; r < KIN; ++r)
#pragma unroll
      for (int j = 0; j < EPL; ++j) vn[r][j] = v[r][j];
  }
  const int nsets = MULTI ? p.nsets : 1;
#pragma unroll
  for (int set = 0; set < (MULTI ? 3 : 1); ++set) {
    if (set >= nsets) break;
    // (MULTI) keep the per-block addresses out of common-subexpression / loop-invariant motion across the sets: hoisted,
    // the 64-bit addresses of every block for every pointer stay live through all sets and push the row out of the registers
    int col0 = col0_k;
    int64_t rbase = rbase_k;
    if (MULTI) asm volatile("" : "+v"(col0), "+v"(rbase));
    if (MULTI && set > 0) {
      if constexpr (KEEP) {
#pragma unroll
        for (int r = 0; r < KIN; ++r)
#pragma unroll
          for (int j = 0; j < EPL; ++j) v[r][j] = vn[r][j];
      } else {
        rot_load_row<KIN, EPL>(p.x, p.x_dtype, rbase + col0, v);
        rot_normalise<KIN, EPL>(p, col0, mb, mean, rstd, v);
      }
    }
    const float* premul = p.premul[set];
    if (premul) {
#pragma unroll
      for (int r = 0; r < KIN; ++r) {
        float pm[EPL];
        IoN<F32, EPL>::load(premul, col0 + 128 * r, pm);
#pragma unroll
        for (int j = 0; j < EPL; ++j) v[r][j] *= pm[j];
        ROT_FENCE(r);
      }
    }
    hadamard_regs<KIN, Q, EPL>(v, lane);
    // the 1 / sqrt(n): applied to the values only where they leave as floating point; the quantiser takes it as a factor
    float cdiv = p.inv_div;
    if (!MULTI && p.out_fp) {
#pragma unroll
      for (int r = 0; r < KIN; ++r)
#pragma unroll
        for (int j = 0; j < EPL; ++j) v[r][j] *= cdiv;
      cdiv = 1.0f;
      if (live) rot_store_row<KIN, EPL>(p.out_fp, p.out_dtype, rbase + col0, v);
    }
    int8_t* q8 = p.q[set];
    if (!q8) return;
    void* scale_t = p.scale[set];
    void* sum_t = p.sum[set];

    float m = 0.f;
#pragma unroll
    for (int r = 0; r < KIN; ++r)
#pragma unroll
      for (int j = 0; j < EPL; ++j) m = fmaxf(m, fabsf(v[r][j]));
    const float amax = group_max<LPR>(m) * cdiv;  // == the maximum of the scaled values (rounding is monotone)
    float scale = amax / 127.0f;
    if (scale < 1e-6f) scale = 1e-6f;  // qdiff eps rule (base_quantizer.py:122-127)
    const float cinv = cdiv * (1.0f / scale);
    int isum = 0;
#pragma unroll
    for (int r = 0; r < KIN; ++r) {
      uint32_t pk[EPL / 4];
      quantN_pack_rne_pre<EPL>(v[r], cdiv, scale, cinv, pk);  // (dynamic scale: |v * cdiv / scale| <= 127.5)
      int8_t* dst = q8 + rbase + col0 + 128 * r;
      const uint32_t lo = pk[0];
      isum = __builtin_amdgcn_sdot4((int)lo, 0x01010101, isum, false);  // sum of the four signed bytes
      if constexpr (EPL == 8) {
        const uint32_t hi = pk[EPL / 4 - 1];
        isum = __builtin_amdgcn_sdot4((int)hi, 0x01010101, isum, false);
        if (live) *reinterpret_cast<uint2*>(dst) = make_uint2(lo, hi);
      } else {
        if (live) *reinterpret_cast<uint32_t*>(dst) = lo;
      }
      ROT_FENCE(r);
    }
    const bool writer = live && (lane & (LPR - 1)) == 0;
    if (sum_t) {
      const int tot = group_isum<LPR>(isum);
      if (writer) vec_store(sum_t, p.vec_dtype, row, (float)tot * scale);
    }
    if (writer) vec_store(scale_t, p.vec_dtype, row, scale);
  }
}

// K' = cols / 128 -> (KIN, Q, EPL): powers of two up to 32, 12 (n = 1536) and 40 = 2 x 20 (n = 5120).  (8 blocks as 2 x 4 lane
// groups: the 8 x 1 and 4 x 2 forms of the LayerNorm variants are ones hipcc fills with spills.)
template <bool LN, bool MULTI>
static int launch_rotate(const RotParams& p, int had_k, hipStream_t st, const char* what) {
  const int64_t rows = p.rows;
#define WANQ_ROT(KIN, Q, EPL)                                                                                     \
  do {                                                                                                            \
    constexpr int rpb = 4 * (64 / ((128 / (EPL)) * (Q)));                                                         \
    hipLaunchKernelGGL((rotate_kernel<KIN, Q, EPL, LN, MULTI>), dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, st, p); \
  } while (0)
  switch (had_k) {
    case 1: WANQ_ROT(1, 1, 8); break;
    case 2: WANQ_ROT(2, 1, 8); break;
    case 4: WANQ_ROT(4, 1, 8); break;
    case 8: WANQ_ROT(2, 4, 8); break;
    case 16: WANQ_ROT(8, 2, 8); break;
    case 32: WANQ_ROT(8, 4, 8); break;
    case 12: WANQ_ROT(12, 1, 4); break;
    case 40: WANQ_ROT(20, 2, 4); break;
    default:
      set_error("%s: no transform kernel for had_k=%d", what, had_k);
      return WANQ_E_SHAPE;
  }
#undef WANQ_ROT
  return check_launch(what);
}

static int check_rotation(const char* what, int had_k, int cols) {
  WANQ_REQUIRE(cols >= 128 && cols % 128 == 0 && had_k == cols / 128, WANQ_E_SHAPE,
               "%s: the transform is (H_K (x) H_128): cols=%d must be had_k * 128 (had_k=%d)", what, cols, had_k);
  const bool pow2 = (had_k & (had_k - 1)) == 0;
  WANQ_REQUIRE((pow2 && had_k <= 32) || had_k == 12 || had_k == 40, WANQ_E_SHAPE,
               "%s: cols=%d has no fused Hadamard transform here (supported: 2^p in [128, 4096], 1536, 5120; 8960 without LayerNorm)", what, cols);
  return WANQ_OK;
}

static int check_rows(const char* what, int64_t rows) {
  WANQ_REQUIRE(rows >= 0 && rows < (1ll << 31), WANQ_E_SHAPE, "%s: rows=%lld out of range", what, (long long)rows);
  return WANQ_OK;
}

}  // namespace wanq

using namespace wanq;

extern "C" int wanq_rotate_quant_rows(const void* x, int x_dtype, const float* premul, int had_k, void* out_fp, int out_dtype,
                                      int8_t* q, void* scale, void* sum, int vec_dtype, int64_t rows, int cols, void* stream) {
  const char* what = "wanq_rotate_quant_rows";
  WANQ_REQUIRE(x && (q || out_fp), WANQ_E_ARG, "%s: need x and at least one of out_fp / q", what);
  WANQ_REQUIRE(is_fp(x_dtype) && (!out_fp || is_fp(out_dtype)), WANQ_E_ARG, "%s: bad dtype code", what);
  WANQ_REQUIRE(!q || (scale && is_vec(vec_dtype)), WANQ_E_ARG, "%s: q needs scale and a valid vec dtype", what);
  if (had_k == 0)  // channel scale only
    return premul_quant_rows(false, x, x_dtype, nullptr, nullptr, nullptr, 0, 1, 0.f, premul, out_fp, out_dtype, q, scale, sum,
                             vec_dtype, rows, cols, (hipStream_t)stream, what);
  if (int e = check_rows(what, rows)) return e;
  if (had_k == 140) {  // 8960 = 140 x 64: the Paley-140 mix runs on the matrix cores (rotate140.hip)
    WANQ_REQUIRE(cols == 8960, WANQ_E_SHAPE, "%s: had_k=140 is the transform of cols=8960 (got %d)", what, cols);
    return rotate140_rows(x, x_dtype, premul, out_fp, out_dtype, q, scale, sum, vec_dtype, rows, (hipStream_t)stream, what);
  }
  if (had_k == 108) {  // 13824 = 108 x 128 (14B ffn.2): Paley-108 on the matrix cores (rotate108.hip); repo-defined, the reference asserts
    WANQ_REQUIRE(cols == 13824, WANQ_E_SHAPE, "%s: had_k=108 is the transform of cols=13824 (got %d)", what, cols);
    return rotate108_rows(x, x_dtype, premul, out_fp, out_dtype, q, scale, sum, vec_dtype, rows, (hipStream_t)stream, what);
  }
  if (int e = check_rotation(what, had_k, cols)) return e;
  if (rows == 0) return WANQ_OK;
  RotParams p{};
  p.x = x; p.x_dtype = x_dtype; p.out_fp = out_fp; p.out_dtype = out_dtype; p.vec_dtype = vec_dtype; p.rows = rows; p.cols = cols;
  p.rows_per_batch = 1; p.inv_div = 1.0f / sqrtf((float)cols); p.nsets = 1;
  p.premul[0] = premul; p.q[0] = q; p.scale[0] = scale; p.sum[0] = sum;
  return launch_rotate<false, false>(p, had_k, (hipStream_t)stream, what);
}

static int fill_ln(RotParams& p, const char* what, const void* x, int x_dtype, const void* gamma, const void* mshift,
                   const void* mscale, int mod_dtype, int64_t mod_stride, int64_t rows_per_batch, float eps, int vec_dtype,
                   int64_t rows, int cols) {
  WANQ_REQUIRE(x, WANQ_E_ARG, "%s: NULL pointer", what);
  WANQ_REQUIRE(is_fp(x_dtype) && is_vec(vec_dtype), WANQ_E_ARG, "%s: bad dtype code", what);
  WANQ_REQUIRE(!(gamma || mshift || mscale) || mod_dtype == WANQ_F32, WANQ_E_ARG,
               "%s: gamma / shift / scale must be fp32 here (mod dtype %d)", what, mod_dtype);
  WANQ_REQUIRE(rows_per_batch >= 1, WANQ_E_ARG, "%s: rows_per_batch must be >= 1", what);
  p.x = x; p.x_dtype = x_dtype; p.gamma = gamma; p.mshift = mshift; p.mscale = mscale;
  p.mod_stride = mod_stride; p.rows_per_batch = rows_per_batch; p.eps = eps; p.vec_dtype = vec_dtype; p.rows = rows; p.cols = cols;
  p.inv_div = 1.0f / sqrtf((float)cols);
  return WANQ_OK;
}

extern "C" int wanq_layernorm_rotate_quant_rows(const void* x, int x_dtype, const void* gamma, const void* mshift,
                                                const void* mscale, int mod_dtype, int64_t mod_stride,
                                                int64_t rows_per_batch, float eps, const float* premul, int had_k, int8_t* q,
                                                void* scale, void* sum, int vec_dtype, int64_t rows, int cols, void* stream) {
  const char* what = "wanq_layernorm_rotate_quant_rows";
  WANQ_REQUIRE(q && scale, WANQ_E_ARG, "%s: NULL pointer", what);
  if (had_k == 0) {  // channel scale only
    RotParams chk{};
    if (int e = fill_ln(chk, what, x, x_dtype, gamma, mshift, mscale, mod_dtype, mod_stride, rows_per_batch, eps, vec_dtype, rows, cols))
      return e;
    return premul_quant_rows(true, x, x_dtype, gamma, mshift, mscale, mod_stride, rows_per_batch, eps, premul, nullptr, WANQ_F32, q,
                             scale, sum, vec_dtype, rows, cols, (hipStream_t)stream, what);
  }
  if (int e = check_rows(what, rows)) return e;
  if (int e = check_rotation(what, had_k, cols)) return e;
  RotParams p{};
  if (int e = fill_ln(p, what, x, x_dtype, gamma, mshift, mscale, mod_dtype, mod_stride, rows_per_batch, eps, vec_dtype, rows, cols))
    return e;
  if (rows == 0) return WANQ_OK;
  p.nsets = 1;
  p.premul[0] = premul; p.q[0] = q; p.scale[0] = scale; p.sum[0] = sum;
  return launch_rotate<true, false>(p, had_k, (hipStream_t)stream, what);
}

extern "C" int wanq_layernorm_rotate_quant_rows_multi(const void* x, int x_dtype, const void* gamma, const void* mshift,
                                                      const void* mscale, int mod_dtype, int64_t mod_stride,
                                                      int64_t rows_per_batch, float eps, int nsets,
                                                      const float* const* premul, int had_k, int8_t* const* q,
                                                      void* const* scale, void* const* sum, int vec_dtype, int64_t rows,
                                                      int cols, void* stream) {
  const char* what = "wanq_layernorm_rotate_quant_rows_multi";
  WANQ_REQUIRE(nsets >= 1 && nsets <= 3, WANQ_E_ARG, "%s: nsets=%d must be 1..3", what, nsets);
  WANQ_REQUIRE(q && scale && sum && premul, WANQ_E_ARG, "%s: NULL pointer", what);
  if (had_k == 0) {  // channel scale only: one pass per set
    for (int t = 0; t < nsets; ++t) {
      WANQ_REQUIRE(q[t] && scale[t], WANQ_E_ARG, "%s: set %d: q and scale are required", what, t);
      if (int e = wanq_layernorm_rotate_quant_rows(x, x_dtype, gamma, mshift, mscale, mod_dtype, mod_stride, rows_per_batch, eps,
                                                   premul[t], 0, q[t], scale[t], sum[t], vec_dtype, rows, cols, stream))
        return e;
    }
    return WANQ_OK;
  }
  if (int e = check_rows(what, rows)) return e;
  if (int e = check_rotation(what, had_k, cols)) return e;
  RotParams p{};
  if (int e = fill_ln(p, what, x, x_dtype, gamma, mshift, mscale, mod_dtype, mod_stride, rows_per_batch, eps, vec_dtype, rows, cols))
    return e;
  for (int t = 0; t < nsets; ++t) {
    WANQ_REQUIRE(q[t] && scale[t], WANQ_E_ARG, "%s: set %d: q and scale are required", what, t);
    p.premul[t] = premul[t]; p.q[t] = q[t]; p.scale[t] = scale[t]; p.sum[t] = sum[t];
  }
  if (rows == 0) return WANQ_OK;
  p.nsets = nsets;
  return launch_rotate<true, true>(p, had_k, (hipStream_t)stream, what);
}
