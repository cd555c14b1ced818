// The transform for n = 8960 = 140 x 64 (Wan2.1-1.3B ffn.2 input):  y = hadU(x * premul),  hadU = (P_140 (x) H_64) / fp32-sqrt(n)
// (get_hadK picks K = 140 for 8960, quarot_utils.py:100-155; matmul_hadU :158-179; the reference multiplies by the dense
// 8960 x 8960 fp64 matrix, viditq_quant_layer.py:62-63).  P_140 is the Paley-I matrix of order 140 -- no butterfly, a dense
// +-1 mix of 140 x 140 per column: 64 x 140 x 140 = 1.25 M adds per row, 4.1e10 per [32760, 8960] call, too many for the
// vector ALUs (> 0.5 ms at their peak).  So the mix runs on the matrix cores, exactly:
//   * the 64-point Walsh-Hadamard transform of every block and the 1/sqrt(n) run on the vector ALUs in fp32 (8 lanes per
//     block, 3 in-register + 3 lane-exchange stages), in the natural layout the row is loaded in;
//   * every fp32 value is then split into three bf16 terms  v = hi + mid + lo  (each the RNE bf16 of the remainder: 3 x 8
//     significand bits = all 24), stored as three [144][64] bf16 planes in LDS;
//   * Y = P_140 . V as v_mfma_f32_32x32x16_bf16 with A = P_140 (entries +-1, exact in bf16, held in REGISTERS for the whole
//     kernel: generated from the quadratic character mod 139, never loaded) and B = the three planes accumulated into the same
//     fp32 accumulator (read with ds_read_b64_tr_b16: the planes are row-major [k'][j] as written, the B operand wants k'
//     along the lane's elements).  Products are exact, sums are fp32: the result is an fp32 evaluation of the transform.
// One workgroup (4 waves) per row at a time, 2 workgroups per CU (54 KiB of LDS each) so that one's load / transform phase
// overlaps the other's MFMA phase; a wave owns one half of the 64 columns and 3 or 2 of the 5 row tiles of Y (140 -> 160), the
// 3 : 2 split alternating with the workgroup's parity so that the two waves a SIMD hosts add up to 5.
#include "wanq_common.h"

namespace wanq {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Rot140Params {
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

constexpr int R140_K = 140;             // blocks per row = order of the Paley matrix
constexpr int R140_N = 140 * 64;        // 8960
constexpr int R140_PLANE = 144 * 128;   // one bf16 plane: 144 rows (K padded to 9 x 16) of 64 columns
constexpr int R140_PASSES = 5;          // 140 blocks over 32 lane groups of 8

// byte offset of (row k', column j) in a plane: the row's two 64-B halves swap on rows 2, 3 (mod 4), so that the four rows a
// half-wave gathers with one transposed read cover all 64 banks (rows are 128 B = 32 banks: rows q and q + 2 would collide)
__device__ __forceinline__ int r140_off(int row, int col) { return row * 128 + ((col * 2) ^ (((row >> 1) & 1) << 6)); }

__device__ __forceinline__ uint32_t bf16_pair_bits(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 t;
  t[0] = (__bf16)a;
  t[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, t);
}

template <typename T>
__device__ __forceinline__ void r140_load_row(const void* x, int64_t rbase, int tid, float (&v)[R140_PASSES][8]) {
#pragma unroll
  for (int ps = 0; ps < R140_PASSES; ++ps) {
    const int b = (tid >> 3) + 32 * ps;
    if (b < R140_K) Io<T>::load8(x, rbase + b * 64 + (tid & 7) * 8, v[ps]);
    else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[ps][j] = 0.f;
    }
  }
}
// 16-bit inputs: the row's raw 16-B chunks.  (Requesting the NEXT row's chunks a row ahead was tried: the 20 registers they pin beside
// the 108 of A make hipcc spill, and a spill reload waits vmcnt(0), i.e. for that very prefetch -- 717 us with it, 664 us without.)
__device__ __forceinline__ void r140_load_raw16(const void* x, int64_t rbase, int tid, uint4 (&raw)[R140_PASSES]) {
#pragma unroll
  for (int ps = 0; ps < R140_PASSES; ++ps) {
    const int b = (tid >> 3) + 32 * ps;
    raw[ps] = b < R140_K ? *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(x) + rbase + b * 64 + (tid & 7) * 8)
                         : make_uint4(0, 0, 0, 0);
  }
}
template <bool BF>
__device__ __forceinline__ void r140_unpack16(const uint4 (&raw)[R140_PASSES], float (&v)[R140_PASSES][8]) {
#pragma unroll
  for (int ps = 0; ps < R140_PASSES; ++ps) {
    const uint32_t w[4] = {raw[ps].x, raw[ps].y, raw[ps].z, raw[ps].w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (BF) {
        v[ps][2 * i] = __uint_as_float(w[i] << 16);
        v[ps][2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
      } else {
        const float2 f = __half22float2(*reinterpret_cast<const __half2*>(&w[i]));
        v[ps][2 * i] = f.x;
        v[ps][2 * i + 1] = f.y;
      }
    }
  }
}

template <typename T>
__device__ __forceinline__ void r140_store_row(void* out, int64_t rbase, int tid, const float (&v)[R140_PASSES][8]) {
#pragma unroll
  for (int ps = 0; ps < R140_PASSES; ++ps) {
    const int b = (tid >> 3) + 32 * ps;
    if (b < R140_K) Io<T>::store8(out, rbase + b * 64 + (tid & 7) * 8, v[ps]);
  }
}

__global__ __launch_bounds__(256, 2) void rotate140_kernel(const Rot140Params p) {
  __shared__ __attribute__((aligned(16))) char smem[3 * R140_PLANE + 256];
  float* red = reinterpret_cast<float*>(smem + 3 * R140_PLANE);        // [4] wave absmax
  int* red_i = reinterpret_cast<int*>(smem + 3 * R140_PLANE + 32);     // [4] wave code sums
  int8_t* chi = reinterpret_cast<int8_t*>(smem);                       // start-up only: quadratic character mod 139
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- A operand: this wave's row tiles of P_140, generated once (first column +1, first row -1, diagonal +1, chi(a - b))
  if (tid < 139) chi[tid] = -1;
  __syncthreads();
  if (tid >= 1 && tid < 139) chi[(tid * tid) % 139] = 1;
  __syncthreads();
  const int nh = wave & 1;                                     // which 32 of the 64 columns
  const bool heavy = (((wave >> 1) ^ (int)(blockIdx.x & 1)) == 0);  // row tiles 0,1,2 or 3,4
  const int mt0 = heavy ? 0 : 3;
  bf16x8 af[3][9];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      const int m = 32 * (mt0 + t) + (lane & 31), k0 = 16 * s + 8 * (lane >> 5);
      const int idx0 = (m - k0 + 278) % 139;  // chi index of element 0; one step down per element
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = k0 + i;
        int idx = idx0 - i;
        idx += idx < 0 ? 139 : 0;
        float e = (float)chi[idx];
        e = (m == k) ? 1.f : e;
        e = (m == 0) ? -1.f : e;
        e = (k == 0) ? 1.f : e;
        e = (m >= R140_K || k >= R140_K) ? 0.f : e;
        af[t][s][i] = (__bf16)e;
      }
    }
  __syncthreads();  // chi shares the plane area

  // transposed-read addresses (bytes within a plane, k-step 0): lane 4q+p of a 16-lane group gives row q, columns 4p..4p+3
  const int grp = lane >> 4, li = lane & 15, tq = li >> 2, tp = li & 3;
  const int trow = 8 * (grp >> 1) + tq, tcol = 32 * nh + 16 * (grp & 1) + 4 * tp;
  const int a_lo = r140_off(trow, tcol), a_hi = r140_off(trow + 4, tcol);
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

  const int tid_k = tid;
  const bool x16 = p.x_dtype != WANQ_F32;
  for (int64_t row = blockIdx.x; row < p.rows; row += gridDim.x) {
    const int64_t rbase = row * (int64_t)R140_N;
    // keep the per-pass addresses of x / premul / out / q out of loop-invariant motion: hoisted, their 64-bit copies for every
    // pass and pointer stay live through the MFMA phase, next to the 108 registers of A, and spill
    int tid = tid_k;
    asm volatile("" : "+v"(tid));
    const int c8 = (tid & 7) * 8;
    // ---- phase 1: load, premultiply, H_64 per block, scale, split into the three planes
    if (tid < 96) *reinterpret_cast<uint4*>(smem + (tid >> 5) * R140_PLANE + 140 * 128 + (tid & 31) * 16) = make_uint4(0, 0, 0, 0);
    float v[R140_PASSES][8];
    if (x16) {
      uint4 raw[R140_PASSES];
      r140_load_raw16(p.x, rbase, tid, raw);
      if (p.x_dtype == WANQ_BF16) r140_unpack16<true>(raw, v);
      else r140_unpack16<false>(raw, v);
    } else {
      r140_load_row<F32>(p.x, rbase, tid, v);
    }
    if (p.premul) {
#pragma unroll
      for (int ps = 0; ps < R140_PASSES; ++ps) {
        const int b = (tid >> 3) + 32 * ps;
        if (b < R140_K) {
          float pm[8];
          Io<F32>::load8(p.premul, b * 64 + c8, pm);
#pragma unroll
          for (int j = 0; j < 8; ++j) v[ps][j] *= pm[j];
        }
      }
    }
#pragma unroll
    for (int ps = 0; ps < R140_PASSES; ++ps) {
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
#define R140_LANE_STAGE(MASK)                                                                                       \
  {                                                                                                                  \
    const float sgn = (lane & MASK) ? -1.f : 1.f;                                                                    \
    _Pragma("unroll") for (int ps = 0; ps < R140_PASSES; ++ps) _Pragma("unroll") for (int j = 0; j < 8; ++j)           \
        v[ps][j] = fmaf(sgn, v[ps][j], __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v[ps][j]), ((MASK) << 10) | 0x1f))); \
  }
    R140_LANE_STAGE(1)
    R140_LANE_STAGE(2)
    R140_LANE_STAGE(4)
#undef R140_LANE_STAGE
#pragma unroll
    for (int ps = 0; ps < R140_PASSES; ++ps) {
      const int b = (tid >> 3) + 32 * ps;
      if (b < R140_K) {
        uint32_t hi[4], mid[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float x0 = v[ps][2 * j] * p.inv_div, x1 = v[ps][2 * j + 1] * p.inv_div;
          hi[j] = bf16_pair_bits(x0, x1);
          const float r0 = x0 - __uint_as_float(hi[j] << 16), r1 = x1 - __uint_as_float(hi[j] & 0xffff0000u);
          mid[j] = bf16_pair_bits(r0, r1);
          const float s0 = r0 - __uint_as_float(mid[j] << 16), s1 = r1 - __uint_as_float(mid[j] & 0xffff0000u);
          lo[j] = bf16_pair_bits(s0, s1);
        }
        const int off = r140_off(b, c8);
        *reinterpret_cast<uint4*>(smem + off) = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        *reinterpret_cast<uint4*>(smem + R140_PLANE + off) = make_uint4(mid[0], mid[1], mid[2], mid[3]);
        *reinterpret_cast<uint4*>(smem + 2 * R140_PLANE + off) = make_uint4(lo[0], lo[1], lo[2], lo[3]);
      }
    }
    __syncthreads();

    // ---- phase 2: Y = P_140 . (hi + mid + lo) on the matrix cores
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
      bf16x8 bf[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        const s16x4 l4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + pl * R140_PLANE + s * 2048 + a_lo));
        const s16x4 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(smem + pl * R140_PLANE + s * 2048 + a_hi));
        const s16x8 j8 = __builtin_shufflevector(l4, h4, 0, 1, 2, 3, 4, 5, 6, 7);
        bf[pl] = __builtin_bit_cast(bf16x8, j8);
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (t == 2 && !heavy) continue;
        // smallest terms first
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][s], bf[2], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][s], bf[1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[t][s], bf[0], acc[t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);  // one k-step of B fragments in flight at a time (108 registers hold A)
    }
    float am = 0.f;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) am = fmaxf(am, fabsf(acc[t][r]));  // padded rows and the light wave's third tile are zeros
    am = wave_max(am);
    if (lane == 0) red[wave] = am;
    __syncthreads();  // every wave is done reading the planes

    // ---- Y back to the natural layout through LDS (fp32 [140][64] over the plane area)
    float* ylds = reinterpret_cast<float*>(smem);
    {
      float* yb = ylds + (4 * (lane >> 5)) * 64 + 32 * nh + (lane & 31);  // row 8(r>>2) + (r&3) of tile t: + a constant
#define R140_YW(t, mt, r) yb[(32 * (mt) + 8 * ((r) >> 2) + ((r) & 3)) * 64] = acc[t][r]
      if (heavy) {
#pragma unroll
        for (int r = 0; r < 16; ++r) { R140_YW(0, 0, r); R140_YW(1, 1, r); R140_YW(2, 2, r); }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) R140_YW(0, 3, r);
#pragma unroll
        for (int r = 0; r < 4; ++r) R140_YW(1, 4, r);  // rows 128..135
        if (lane < 32) {                               // rows 136..139 (the upper half-wave holds 140..143: padding)
#pragma unroll
          for (int r = 4; r < 8; ++r) R140_YW(1, 4, r);
        }
      }
#undef R140_YW
    }
    const float amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();

    // ---- phase 3: optional fp output, per-token int8 quantisation (qdiff DynamicQuantizer: base_quantizer.py:101-162)
#pragma unroll
    for (int ps = 0; ps < R140_PASSES; ++ps) {
      const int b = (tid >> 3) + 32 * ps;
      if (b < R140_K) {
        const float4 y0 = *reinterpret_cast<const float4*>(ylds + b * 64 + c8), y1 = *reinterpret_cast<const float4*>(ylds + b * 64 + c8 + 4);
        v[ps][0] = y0.x; v[ps][1] = y0.y; v[ps][2] = y0.z; v[ps][3] = y0.w;
        v[ps][4] = y1.x; v[ps][5] = y1.y; v[ps][6] = y1.z; v[ps][7] = y1.w;
      }
    }
    if (p.out_fp) {
      if (p.out_dtype == WANQ_BF16) r140_store_row<BF16>(p.out_fp, rbase, tid, v);
      else if (p.out_dtype == WANQ_F16) r140_store_row<F16>(p.out_fp, rbase, tid, v);
      else r140_store_row<F32>(p.out_fp, rbase, tid, v);
    }
    if (p.q) {
      float scale = amax / 127.0f;
      if (scale < 1e-6f) scale = 1e-6f;  // qdiff eps rule (base_quantizer.py:122-127)
      const float inv = 1.0f / scale;
      int isum = 0;
#pragma unroll
      for (int ps = 0; ps < R140_PASSES; ++ps) {
        const int b = (tid >> 3) + 32 * ps;
        if (b < R140_K) {
          uint32_t pk[2];
          quantN_pack_rne<8>(v[ps], scale, inv, pk);
          const uint32_t lo = pk[0], hi = pk[1];
          isum = __builtin_amdgcn_sdot4((int)lo, 0x01010101, isum, false);
          isum = __builtin_amdgcn_sdot4((int)hi, 0x01010101, isum, false);
          *reinterpret_cast<uint2*>(p.q + rbase + b * 64 + c8) = make_uint2(lo, hi);
        }
      }
      isum = wave_sum(isum);
      if (lane == 0) red_i[wave] = isum;
      __syncthreads();
      if (tid == 0) {
        vec_store(p.scale, p.vec_dtype, row, scale);
        if (p.sum) vec_store(p.sum, p.vec_dtype, row, (float)(red_i[0] + red_i[1] + red_i[2] + red_i[3]) * scale);
      }
    } else {
      __syncthreads();
    }
  }
}

int rotate140_rows(const void* x, int x_dtype, const float* premul, void* out_fp, int out_dtype, int8_t* q, void* scale, void* sum,
                   int vec_dtype, int64_t rows, hipStream_t st, const char* what) {
  if (rows == 0) return WANQ_OK;
  Rot140Params p{};
  p.x = x; p.x_dtype = x_dtype; p.premul = premul; p.out_fp = out_fp; p.out_dtype = out_dtype; p.q = q; p.scale = scale; p.sum = sum;
  p.vec_dtype = vec_dtype; p.rows = rows; p.inv_div = 1.0f / sqrtf((float)R140_N);
  const unsigned grid = (unsigned)(rows < 512 ? rows : 512);  // 2 resident workgroups per CU, rows round-robin
  hipLaunchKernelGGL(rotate140_kernel, dim3(grid), dim3(256), 0, st, p);
  return check_launch(what);
}

}  // namespace wanq
