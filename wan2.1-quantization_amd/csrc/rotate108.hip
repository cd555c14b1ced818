// The transform for n = 13824 = 108 x 128 (Wan2.1-14B ffn.2 input):  y = hadU(x * premul),  hadU = (P_108 (x) H_128) / fp32-sqrt(n).
//
// REPO-DEFINED behaviour: the reference cannot rotate 13824 columns at all -- get_hadK reaches `n % 144 == 0` first and
// asserts is_pow2(96) (quarot_utils.py:110-112) before its K = 108 branch (:118-121), which would fit.  This is that branch:
// the reference's own get_had108 table (= the Paley-I matrix of order 108, quadratic character mod 107), its butterfly over the 128
// columns of every block and its fp32 sqrt (matmul_hadU, :158-179).  Pinned by tests/golden/a5_hadamard_13824.npz, which is made from
// the reference's table and loop.
//
// Structure = rotate140.hip's (8960 = 140 x 64), re-cut for 128-wide blocks:
//   * one workgroup (4 waves) per row at a time, rows round-robin; 16 lanes x 8 elements hold a block; the 128-point Walsh-Hadamard
//     transform of every block (3 in-register + 4 lane-exchange stages) and the 1 / sqrt(n) run on the vector ALUs in fp32;
//   * the P_108 mix across the 108 blocks is a dense +-1 product per column -- 108 x 108 x 128 adds per row, too many for the vector
//     ALUs -- and runs on the matrix cores EXACTLY: every fp32 value is split into three bf16 terms v = hi + mid + lo (3 x 8
//     significand bits), stored as three [112][64] bf16 planes in LDS, and Y = P_108 . V is v_mfma_f32_32x32x16_bf16 with
//     A = P_108 (+-1: exact in bf16; generated from the quadratic character mod 107 and held in registers for the whole kernel)
//     and B = the three planes accumulated into one fp32 accumulator (products exact, sums fp32);
//   * the 128 columns of a block go through the matrix cores in two halves of 64, so that the planes are 42 KiB and two workgroups
//     share a CU (one's load / butterfly phase beside the other's MFMA phase); wave w owns row tile w (rows 32 w .. 32 w + 31 of
//     the 108 -> 128 rows of Y) for both 32-column sub-tiles of the half.
#include "wanq_common.h"

namespace wanq {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Rot108Params {
  const void* x;
  int x_dtype;
  const float* premul;
  void* out_fp;
  int out_dtype;
  int8_t* q;
  void* scale;
  void* sum;
  int vec_dtype;
  int64_t rows;
  float inv_div;
};

constexpr int R108_K = 108;            // blocks per row = order of the Paley matrix
constexpr int R108_Q = 107;            // its prime
constexpr int R108_M = 128;            // columns per block
constexpr int R108_N = R108_K * R108_M;  // 13824
constexpr int R108_KP = 112;           // K padded to 7 x 16 (the MFMA's k-steps)
constexpr int R108_PLANE = R108_KP * 128;  // one bf16 plane: 112 rows of 64 columns
constexpr int R108_PASSES = 7;         // 112 block slots over 16 lane groups of 16

// byte offset of (row k', column j) in a plane: the row's two 64-B halves swap on rows 2, 3 (mod 4), so that the four rows a
// half-wave gathers with one transposed read cover all 64 banks (rotate140.hip's image)
__device__ __forceinline__ int r108_off(int row, int col) { return row * 128 + ((col * 2) ^ (((row >> 1) & 1) << 6)); }

__device__ __forceinline__ uint32_t r108_bf16_pair(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 t;
  t[0] = (__bf16)a;
  t[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, t);
}

// (one dtype branch per ROW, not per chunk: with a branch per chunk hipcc waits for each load before it issues the next one)
template <typename T>
__device__ __forceinline__ void r108_load_row(const void* x, int64_t rbase, int bgrp, int c8, float (&v)[R108_PASSES][8]) {
#pragma unroll
  for (int ps = 0; ps < R108_PASSES; ++ps) {
    const int b = bgrp + 16 * ps;
    if (b < R108_K) Io<T>::load8(x, rbase + b * R108_M + c8, v[ps]);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[ps][j] = 0.f;
    }
  }
}
template <typename T>
__device__ __forceinline__ void r108_store_row(void* out, int64_t rbase, int bgrp, int c8, const float (&v)[R108_PASSES][8]) {
#pragma unroll
  for (int ps = 0; ps < R108_PASSES; ++ps) {
    const int b = bgrp + 16 * ps;
    if (b < R108_K) Io<T>::store8(out, rbase + b * R108_M + c8, v[ps]);
  }
}

__global__ __launch_bounds__(256, 2) void rotate108_kernel(const Rot108Params p) {
  __shared__ __attribute__((aligned(16))) char smem[3 * R108_PLANE + 64];
  float* red = reinterpret_cast<float*>(smem + 3 * R108_PLANE);      // [4] wave absmax
  int* red_i = reinterpret_cast<int*>(smem + 3 * R108_PLANE + 32);   // [4] wave code sums
  int8_t* chi = reinterpret_cast<int8_t*>(smem);                     // start-up only: quadratic character mod 107
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- A operand: this wave's row tile of P_108 (first column +1, first row -1, diagonal +1, chi(m - k) elsewhere), generated once
  if (tid < R108_Q) chi[tid] = -1;
  __syncthreads();
  if (tid >= 1 && tid < R108_Q) chi[(tid * tid) % R108_Q] = 1;
  __syncthreads();
  bf16x8 af[7];
#pragma unroll
  for (int s = 0; s < 7; ++s) {
    const int m = 32 * wave + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
    const int idx0 = (m - k0 + 2 * R108_Q) % R108_Q;  // chi index of element 0; one step down per element
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = k0 + i;
      int idx = idx0 - i;
      idx += idx < 0 ? R108_Q : 0;
      float e = (float)chi[idx];
      e = (m == k) ? 1.f : e;
      e = (m == 0) ? -1.f : e;
      e = (k == 0) ? 1.f : e;
      e = (m >= R108_K || k >= R108_K) ? 0.f : e;
      af[s][i] = (__bf16)e;
    }
  }
  __syncthreads();  // chi shares the plane area

  // transposed-read addresses (bytes within a plane, k-step 0) for the two 32-column sub-tiles: lane 4q+p of a 16-lane group gives
  // row q, columns 4p..4p+3
  const int grp = lane >> 4, l16 = lane & 15, tq = l16 >> 2, tp = l16 & 3;
  const int trow = 8 * (grp >> 1) + tq, tcol = 16 * (grp & 1) + 4 * tp;
  const int a_lo0 = r108_off(trow, tcol), a_hi0 = r108_off(trow + 4, tcol);
  const int a_lo1 = r108_off(trow, 32 + tcol), a_hi1 = r108_off(trow + 4, 32 + tcol);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

  const int tid_k = tid;
  for (int64_t row = blockIdx.x; row < p.rows; row += gridDim.x) {
    const int64_t rbase = row * (int64_t)R108_N;
    int t = tid_k;  // (kept out of loop-invariant motion, as in rotate140.hip: hoisted per-pass addresses spill)
    asm volatile("" : "+v"(t));
    const int li = t & 15, bgrp = t >> 4;  // lane within its block's 16, block slot within a pass
    const int c8 = li * 8;
    // ---- phase 1: load, premultiply, H_128 per block, scale
    float v[R108_PASSES][8];
    if (p.x_dtype == WANQ_BF16) r108_load_row<BF16>(p.x, rbase, bgrp, c8, v);
    else if (p.x_dtype == WANQ_F16) r108_load_row<F16>(p.x, rbase, bgrp, c8, v);
    else r108_load_row<F32>(p.x, rbase, bgrp, c8, v);
    if (p.premul) {
#pragma unroll
      for (int ps = 0; ps < R108_PASSES; ++ps) {
        const int b = bgrp + 16 * ps;
        if (b < R108_K) {
          float pm[8];
          Io<F32>::load8(p.premul, b * R108_M + c8, pm);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[ps][j] *= pm[j];
        }
      }
    }
#pragma unroll
    for (int ps = 0; ps < R108_PASSES; ++ps) {
#pragma unroll
      for (int h = 1; h < 8; h <<= 1)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          if (!(j & h)) {
            const float a = v[ps][j], b2 = v[ps][j | h];
            v[ps][j] = a + b2;
            v[ps][j | h] = a - b2;
          }
    }
#define R108_LANE_STAGE(MASK)                                                                                       \
  {                                                                                                                  \
    const float sgn = (lane & MASK) ? -1.f : 1.f;                                                                    \
    _Pragma("unroll") for (int ps = 0; ps < R108_PASSES; ++ps) _Pragma("unroll") for (int j = 0; j < 8; ++j)           \
        v[ps][j] = fmaf(sgn, v[ps][j], __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v[ps][j]), ((MASK) << 10) | 0x1f))); \
  }
    R108_LANE_STAGE(1)
    R108_LANE_STAGE(2)
    R108_LANE_STAGE(4)
    R108_LANE_STAGE(8)
#undef R108_LANE_STAGE
#pragma unroll
    for (int ps = 0; ps < R108_PASSES; ++ps)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[ps][j] *= p.inv_div;

    float am = 0.f;
    float* ylds = reinterpret_cast<float*>(smem);  // Y of a half back in the natural layout: fp32 [128][64] over the plane area
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const bool mine = (li >> 3) == half;  // this lane's 8 columns belong to the half
      // ---- the three bf16 planes of this half's 64 columns (rows 108..111 are written as zeros by the padding slots)
      if (mine) {
#pragma unroll
        for (int ps = 0; ps < R108_PASSES; ++ps) {
          const int b = bgrp + 16 * ps;
          uint32_t hi[4], mid[4], lo[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float x0 = v[ps][2 * j], x1 = v[ps][2 * j + 1];
            hi[j] = r108_bf16_pair(x0, x1);
            const float r0 = x0 - __uint_as_float(hi[j] << 16), r1 = x1 - __uint_as_float(hi[j] & 0xffff0000u);
            mid[j] = r108_bf16_pair(r0, r1);
            const float s0 = r0 - __uint_as_float(mid[j] << 16), s1 = r1 - __uint_as_float(mid[j] & 0xffff0000u);
            lo[j] = r108_bf16_pair(s0, s1);
          }
          const int off = r108_off(b, (li & 7) * 8);
          *reinterpret_cast<uint4*>(smem + off) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
          *reinterpret_cast<uint4*>(smem + R108_PLANE + off) = make_uint4(mid[0], mid[1], mid[2], mid[3]);
          *reinterpret_cast<uint4*>(smem + 2 * R108_PLANE + off) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
      }
      __syncthreads();

      // ---- Y = P_108 . (hi + mid + lo) on the matrix cores: row tile `wave`, both 32-column sub-tiles
      f32x16 acc0, acc1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        bf16x8 b0[3], b1[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const char* base = smem + pl * R108_PLANE + s * 2048;
          const s16x4 l0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + a_lo0));
          const s16x4 h0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + a_hi0));
          const s16x4 l1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + a_lo1));
          const s16x4 h1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + a_hi1));
          b0[pl] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(l0, h0, 0, 1, 2, 3, 4, 5, 6, 7));
          b1[pl] = __builtin_bit_cast(bf16x8, (s16x8)__builtin_shufflevector(l1, h1, 0, 1, 2, 3, 4, 5, 6, 7));
        }
        // smallest terms first
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], b0[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], b1[2], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], b0[1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], b1[1], acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], b0[0], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], b1[0], acc1, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) am = fmaxf(am, fmaxf(fabsf(acc0[r]), fabsf(acc1[r])));  // rows 108..127 of the last tile are zeros
      __syncthreads();  // every wave is done reading the planes

      // ---- Y back to the natural layout through LDS: accumulator register r of lane (n, hf) is row 8 (r >> 2) + (r & 3) + 4 hf
      {
        float* yb = ylds + (32 * wave + 4 * (lane >> 5)) * 64 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          yb[(8 * (r >> 2) + (r & 3)) * 64] = acc0[r];
          yb[(8 * (r >> 2) + (r & 3)) * 64 + 32] = acc1[r];
        }
      }
      __syncthreads();
      if (mine) {
#pragma unroll
        for (int ps = 0; ps < R108_PASSES; ++ps) {
          const int b = bgrp + 16 * ps;
          if (b < R108_K) {
            const float* yr = ylds + b * 64 + (li & 7) * 8;
            const float4 y0 = *reinterpret_cast<const float4*>(yr), y1 = *reinterpret_cast<const float4*>(yr + 4);
            v[ps][0] = y0.x; v[ps][1] = y0.y; v[ps][2] = y0.z; v[ps][3] = y0.w;
            v[ps][4] = y1.x; v[ps][5] = y1.y; v[ps][6] = y1.z; v[ps][7] = y1.w;
          }
        }
      }
      __syncthreads();  // Y is consumed before the next half's planes (or the next row's) overwrite it
    }
    am = wave_max(am);
    if (lane == 0) red[wave] = am;
    __syncthreads();
    const float amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));

    // ---- optional fp output, per-token int8 quantisation (qdiff DynamicQuantizer: base_quantizer.py:101-162)
    if (p.out_fp) {
      if (p.out_dtype == WANQ_BF16) r108_store_row<BF16>(p.out_fp, rbase, bgrp, c8, v);
      else if (p.out_dtype == WANQ_F16) r108_store_row<F16>(p.out_fp, rbase, bgrp, c8, v);
      else r108_store_row<F32>(p.out_fp, rbase, bgrp, c8, v);
    }
    if (p.q) {
      float scale = amax / 127.0f;
      if (scale < 1e-6f) scale = 1e-6f;  // qdiff eps rule (base_quantizer.py:122-127)
      const float inv = 1.0f / scale;
      int isum = 0;
#pragma unroll
      for (int ps = 0; ps < R108_PASSES; ++ps) {
        const int b = bgrp + 16 * ps;
        if (b < R108_K) {
          uint32_t pk[2];
          quantN_pack_rne<8>(v[ps], scale, inv, pk);
          isum = __builtin_amdgcn_sdot4((int)pk[0], 0x01010101, isum, false);
          isum = __builtin_amdgcn_sdot4((int)pk[1], 0x01010101, isum, false);
          *reinterpret_cast<uint2*>(p.q + rbase + b * R108_M + c8) = make_uint2(pk[0], pk[1]);
        }
      }
      isum = wave_sum(isum);
      if (lane == 0) red_i[wave] = isum;
      __syncthreads();
      if (tid == 0) {
        vec_store(p.scale, p.vec_dtype, row, scale);
        if (p.sum) vec_store(p.sum, p.vec_dtype, row, (float)(red_i[0] + red_i[1] + red_i[2] + red_i[3]) * scale);
      }
    }
    __syncthreads();  // red / red_i are free for the next row
  }
}

int rotate108_rows(const void* x, int x_dtype, const float* premul, void* out_fp, int out_dtype, int8_t* q, void* scale, void* sum,
                   int vec_dtype, int64_t rows, hipStream_t st, const char* what) {
  if (rows == 0) return WANQ_OK;
  Rot108Params p{};
  p.x = x; p.x_dtype = x_dtype; p.premul = premul; p.out_fp = out_fp; p.out_dtype = out_dtype; p.q = q; p.scale = scale; p.sum = sum;
  p.vec_dtype = vec_dtype; p.rows = rows; p.inv_div = 1.0f / sqrtf((float)R108_N);
  const unsigned grid = (unsigned)(rows < 512 ? rows : 512);  // 2 resident workgroups per CU, rows round-robin
  hipLaunchKernelGGL(rotate108_kernel, dim3(grid), dim3(256), 0, st, p);
  return check_launch(what);
}

}  // namespace wanq
