// Row-wise HBM-bound kernels: per-token int8 quantisation, LayerNorm+modulate(+quant), gate-residual,
// calibration absmax, weight row statistics and static weight quantisation.
//
// Data layout: every tensor is row-major [rows, cols]; a row is split into 16-byte-aligned chunks of
// 8 elements; chunk c of a row goes to lane (c % (64*WPR)), so one wave instruction reads
// 64 consecutive chunks (1 KiB for 16-bit inputs, 2 KiB for fp32) -- fully coalesced.
// A row lives in registers between its reductions and its store: each element is read from HBM once
// and written once (algorithmic bytes == traffic).
//   WPR = 1: one wave per row (cols <= 2048), 4 rows per 256-thread workgroup, no barriers, no LDS.
//   WPR = 4: four waves per row (cols <= 16384), reductions finished through 64 B of LDS.
#include "wanq_common.h"
#include <stdlib.h>

namespace wanq {

struct RowParams {
  const void* x;
  int x_dtype;
  // layernorm
  const void* gamma;
  const void* mshift;
  const void* mscale;
  int mod_dtype;
  int64_t mod_stride;
  int64_t rows_per_batch;
  float eps;
  // outputs
  void* out_fp;
  int out_dtype;
  int8_t* q;
  void* scale;
  void* sum;
  int vec_dtype;
  int64_t rows;
  int cols;
  int act;
  int static_amax;
  const float* premul;  // [cols] fp32 multiplier applied before quantisation (SmoothQuant channel mask), or NULL
  // dynamic quantiser range: scale = absmax / levels floored at `floor` (127 and 1e-6 for every 8-bit entry point;
  // wanq_quant_rows_levels: 2^(b-1) - 1 and the caller's floor, 0 = none: a zero row then gets scale 0 and codes 0)
  float levels, floor;
};

__device__ __forceinline__ void load8_rt(const void* base, int dt, int64_t elem, float (&v)[8]) {
  if (dt == WANQ_F16) Io<F16>::load8(base, elem, v);
  else if (dt == WANQ_BF16) Io<BF16>::load8(base, elem, v);
  else Io<F32>::load8(base, elem, v);
}
__device__ __forceinline__ void store8_rt(void* base, int dt, int64_t elem, const float (&v)[8]) {
  if (dt == WANQ_F16) Io<F16>::store8(base, elem, v);
  else if (dt == WANQ_BF16) Io<BF16>::store8(base, elem, v);
  else Io<F32>::store8(base, elem, v);
}

// One dtype branch per ROW, not per chunk: with a branch per chunk hipcc waits for each load (s_waitcnt vmcnt(0) at the
// branch's end) before it issues the next one, and a row's NCH loads run back to back in latency instead of in parallel.
template <typename T, int WPR, int NCH>
__device__ __forceinline__ void load_row_t(const void* x, int64_t rbase, int sub, int lane, int C, float (&v)[NCH][8], bool (&ok)[NCH]) {
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c0 = (sub * 64 + lane + i * 64 * WPR) * 8;
    ok[i] = c0 < C;
    if (ok[i]) {
      Io<T>::load8(x, rbase + c0, v[i]);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    }
  }
}
template <int WPR, int NCH>
__device__ __forceinline__ void load_row(const void* x, int dt, int64_t rbase, int sub, int lane, int C, float (&v)[NCH][8], bool (&ok)[NCH]) {
  if (dt == WANQ_F32) load_row_t<F32, WPR, NCH>(x, rbase, sub, lane, C, v, ok);
  else if (dt == WANQ_BF16) load_row_t<BF16, WPR, NCH>(x, rbase, sub, lane, C, v, ok);
  else load_row_t<F16, WPR, NCH>(x, rbase, sub, lane, C, v, ok);
}
// v <- v * m (MUL) / v * (1 + m) (MUL1P) / v + m (ADD) with a per-column vector m, same dispatch
enum { MOD_MUL = 0, MOD_MUL1P = 1, MOD_ADD = 2 };
template <typename T, int OP, int WPR, int NCH>
__device__ __forceinline__ void mod_row_t(const void* m, int64_t mbase, int sub, int lane, float (&v)[NCH][8], const bool (&ok)[NCH]) {
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (ok[i]) {
      float t[8];
      Io<T>::load8(m, mbase + (sub * 64 + lane + i * 64 * WPR) * 8, t);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = OP == MOD_MUL ? v[i][j] * t[j] : OP == MOD_MUL1P ? v[i][j] * (1.0f + t[j]) : v[i][j] + t[j];
    }
}
template <int OP, int WPR, int NCH>
__device__ __forceinline__ void mod_row(const void* m, int dt, int64_t mbase, int sub, int lane, float (&v)[NCH][8], const bool (&ok)[NCH]) {
  if (dt == WANQ_F32) mod_row_t<F32, OP, WPR, NCH>(m, mbase, sub, lane, v, ok);
  else if (dt == WANQ_BF16) mod_row_t<BF16, OP, WPR, NCH>(m, mbase, sub, lane, v, ok);
  else mod_row_t<F16, OP, WPR, NCH>(m, mbase, sub, lane, v, ok);
}
template <typename T, int WPR, int NCH>
__device__ __forceinline__ void store_row_t(void* out, int64_t rbase, int sub, int lane, const float (&v)[NCH][8], const bool (&ok)[NCH]) {
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (ok[i]) Io<T>::store8(out, rbase + (sub * 64 + lane + i * 64 * WPR) * 8, v[i]);
}

template <int WPR>
struct RowReduce {
  // LDS slots: one float per wave per reduction id; every reduction id is used once per kernel, so a
  // single barrier per reduction suffices.
  float* slots;
  int wave;
  __device__ __forceinline__ float sum(float v, int id) {
    v = wave_sum(v);
    if (WPR == 1) return v;
    if ((threadIdx.x & 63) == 0) slots[id * WPR + wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WPR; ++w) t += slots[id * WPR + w];
    return t;
  }
  __device__ __forceinline__ float max(float v, int id) {
    v = wave_max(v);
    if (WPR == 1) return v;
    if ((threadIdx.x & 63) == 0) slots[id * WPR + wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < WPR; ++w) t = fmaxf(t, slots[id * WPR + w]);
    return t;
  }
  __device__ __forceinline__ int isum(int v, int id) {
    v = wave_sum(v);
    if (WPR == 1) return v;
    if ((threadIdx.x & 63) == 0) reinterpret_cast<int*>(slots)[id * WPR + wave] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int w = 0; w < WPR; ++w) t += reinterpret_cast<int*>(slots)[id * WPR + w];
    return t;
  }
};

template <int WPR, int NCH, bool LN>
__global__ __launch_bounds__(256) void rowwise_kernel(const RowParams p) {
  __shared__ float red_slots[4 * 4];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t row = (WPR == 1) ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x;
  if (WPR == 1 && row >= p.rows) return;  // surplus wave of the last workgroup (wave-per-row variants have no barriers)
  RowReduce<WPR> red{red_slots, wave};
  const int sub = (WPR == 1) ? 0 : wave;
  const int C = p.cols;
  const int64_t rbase = row * (int64_t)C;

  float v[NCH][8];
  bool ok[NCH];
  load_row<WPR, NCH>(p.x, p.x_dtype, rbase, sub, lane, C, v, ok);

  if (LN) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[i][j];
    const float mean = red.sum(s, 0) / (float)C;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (ok[i]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = v[i][j] - mean;
          s2 += d * d;
        }
      }
    const float var = red.sum(s2, 1) / (float)C;
    const float rstd = 1.0f / sqrtf(var + p.eps);
    const int64_t mb = (row / p.rows_per_batch) * p.mod_stride;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (ok[i]) {  // chunks past the row end stay zero: they take part in the row max and sum
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] = (v[i][j] - mean) * rstd;
      }
    if (p.gamma) mod_row<MOD_MUL, WPR, NCH>(p.gamma, p.mod_dtype, 0, sub, lane, v, ok);
    if (p.mscale) mod_row<MOD_MUL1P, WPR, NCH>(p.mscale, p.mod_dtype, mb, sub, lane, v, ok);
    if (p.mshift) mod_row<MOD_ADD, WPR, NCH>(p.mshift, p.mod_dtype, mb, sub, lane, v, ok);
  } else if (p.act == 1) {
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = gelu_tanh_fast_f32(v[i][j]);  // branch-free form (3e-6 rel): libm tanhf x 8 NCH bloats every variant
  }

  if (p.premul) {
#pragma unroll
    for (int i = 0; i < NCH; ++i)
      if (ok[i]) {
        float pm[8];
        Io<F32>::load8(p.premul, (sub * 64 + lane + i * 64 * WPR) * 8, pm);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[i][j] *= pm[j];
      }
  }
  if (p.out_fp) {
    if (p.out_dtype == WANQ_F32) store_row_t<F32, WPR, NCH>(p.out_fp, rbase, sub, lane, v, ok);
    else if (p.out_dtype == WANQ_BF16) store_row_t<BF16, WPR, NCH>(p.out_fp, rbase, sub, lane, v, ok);
    else store_row_t<F16, WPR, NCH>(p.out_fp, rbase, sub, lane, v, ok);
  }
  if (!p.q) return;

  float amax;
  if (p.static_amax) {
    amax = vec_load(p.scale, p.vec_dtype, row);
  } else {
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[i][j]));
    amax = red.max(m, 2);
  }
  float scale = amax / p.levels;
  if (scale < p.floor) scale = p.floor;  // qdiff eps rule (base_quantizer.py:122-127)
  const float inv = scale > 0.f ? 1.0f / scale : 0.f;  // (no floor and an all-zero row: every code is 0)
  int isum = 0;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    uint32_t lo, hi;
    if (p.static_amax) {  // a given scale: values may exceed the code range and are clamped
      int qi[8];
      quant8_div_rne(v[i], scale, inv, qi);
      lo = pack4_i8_fast(qi[0], qi[1], qi[2], qi[3]), hi = pack4_i8_fast(qi[4], qi[5], qi[6], qi[7]);
    } else {              // the row's own scale: |v / scale| <= 127.5 by construction
      uint32_t pk[2];
      quantN_pack_rne<8>(v[i], scale, inv, pk);
      lo = pk[0], hi = pk[1];
    }
    isum = __builtin_amdgcn_sdot4((int)lo, 0x01010101, isum, false);  // sum of the four signed bytes
    isum = __builtin_amdgcn_sdot4((int)hi, 0x01010101, isum, false);
    if (ok[i]) *reinterpret_cast<uint2*>(p.q + rbase + (sub * 64 + lane + i * 64 * WPR) * 8) = make_uint2(lo, hi);
  }
  if (p.sum) {
    const int tot = red.isum(isum, 3);
    if (lane == 0 && sub == 0) vec_store(p.sum, p.vec_dtype, row, (float)tot * scale);
  }
  if (!p.static_amax && lane == 0 && sub == 0) vec_store(p.scale, p.vec_dtype, row, scale);
}

template <bool LN>
static int launch_rowwise(const RowParams& p, hipStream_t st, const char* what) {
  const int chunks = p.cols / 8;
  const int64_t rows = p.rows;
#define WANQ_RW(WPR, NCH) \
  hipLaunchKernelGGL((rowwise_kernel<WPR, NCH, LN>), dim3((unsigned)((WPR) == 1 ? (rows + 3) / 4 : rows)), dim3(256), 0, st, p)
  if (chunks <= 64) WANQ_RW(1, 1);
  else if (chunks <= 128) WANQ_RW(1, 2);
  else if (chunks <= 192) WANQ_RW(1, 3);
  else if (chunks <= 256) WANQ_RW(1, 4);
  else if (chunks <= 512) WANQ_RW(4, 2);
  else if (chunks <= 768) WANQ_RW(4, 3);
  else if (chunks <= 1024) WANQ_RW(4, 4);
  else if (chunks <= 1280) WANQ_RW(4, 5);
  else if (chunks <= 1536) WANQ_RW(4, 6);
  else if (chunks <= 1792) WANQ_RW(4, 7);
  else WANQ_RW(4, 8);
#undef WANQ_RW
  return check_launch(what);
}

static int check_rows_cols(const char* what, int64_t rows, int cols) {
  WANQ_REQUIRE(rows >= 0 && rows < (1ll << 31), WANQ_E_SHAPE, "%s: rows=%lld out of range", what, (long long)rows);
  WANQ_REQUIRE(cols >= 8 && cols % 8 == 0 && cols <= 16384, WANQ_E_SHAPE,
               "%s: cols=%d must be a multiple of 8 in [8, 16384]", what, cols);
  return WANQ_OK;
}

// ------------------------------------------------------------------------------ gate * y + residual
struct GateParams {
  const void* y;
  const void* gate;
  const void* res;
  void* out;
  int y_dtype, gate_dtype, res_dtype, out_dtype;
  int64_t gate_stride, rows_per_batch, rows;
  int cols;
};

__global__ __launch_bounds__(256) void gate_residual_kernel(const GateParams p) {
  const int cpr = p.cols / 8;  // chunks per row
  const int64_t total = p.rows * (int64_t)cpr;
  for (int64_t ch = (int64_t)blockIdx.x * 256 + threadIdx.x; ch < total; ch += (int64_t)gridDim.x * 256) {
    const int64_t row = ch / cpr;
    const int c0 = (int)(ch - row * cpr) * 8;
    float y[8], g[8], r[8];
    load8_rt(p.y, p.y_dtype, row * p.cols + c0, y);
    load8_rt(p.gate, p.gate_dtype, (row / p.rows_per_batch) * p.gate_stride + c0, g);
    load8_rt(p.res, p.res_dtype, row * p.cols + c0, r);
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = y[j] * g[j] + r[j];
    store8_rt(p.out, p.out_dtype, row * p.cols + c0, y);
  }
}

// ------------------------------------------------------------------------------ calibration: column absmax
// Each workgroup owns a 512-column panel (64 lanes x 8 columns) and a slab of rows; a lane keeps 8
// running maxima in registers, the 4 waves of a workgroup take rows round-robin, partials meet in LDS
// and one atomicMax per column per workgroup goes to HBM (non-negative floats order like uints).
template <typename T>
__global__ __launch_bounds__(256) void col_absmax_kernel(const void* x, float* colmax, int64_t rows, int cols,
                                                         int rows_per_block) {
  __shared__ float part[4][512];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = blockIdx.x * 512 + lane * 8;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < rows) ? r0 + rows_per_block : rows;
  float m[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) m[j] = 0.f;
  if (c0 < cols) {
    int64_t r = r0 + wave;
    for (; r + 12 < r1; r += 16) {  // 4 independent loads in flight per lane
      float a[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) Io<T>::load8(x, (r + 4 * u) * cols + c0, a[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], fabsf(a[u][j]));
    }
    for (; r < r1; r += 4) {
      float a[8];
      Io<T>::load8(x, r * cols + c0, a);
#pragma unroll
      for (int j = 0; j < 8; ++j) m[j] = fmaxf(m[j], fabsf(a[j]));
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[wave][lane * 8 + j] = m[j];
  __syncthreads();
  for (int c = threadIdx.x; c < 512; c += 256) {
    const int col = blockIdx.x * 512 + c;
    if (col < cols) {
      const float v = fmaxf(fmaxf(part[0][c], part[1][c]), fmaxf(part[2][c], part[3][c]));
      atomicMax(reinterpret_cast<unsigned int*>(colmax + col), __float_as_uint(v));
    }
  }
}

// ------------------------------------------------------------------------------ weight row statistics
template <int WPR, int NCH>
__global__ __launch_bounds__(256) void row_minmax_kernel(const void* w, int dt, float* rmin, float* rmax,
                                                         float* rabs, int64_t rows, int cols) {
  __shared__ float slots[2][4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (WPR == 1) ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x;
  if (WPR == 1 && row >= rows) return;
  const int sub = (WPR == 1) ? 0 : wave;
  float lo = INFINITY, hi = -INFINITY;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int c0 = (sub * 64 + lane + i * 64 * WPR) * 8;
    if (c0 < cols) {
      float v[8];
      load8_rt(w, dt, row * cols + c0, v);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        lo = fminf(lo, v[j]);
        hi = fmaxf(hi, v[j]);
      }
    }
  }
  lo = wave_min(lo);
  hi = wave_max(hi);
  if (WPR > 1) {
    if (lane == 0) {
      slots[0][wave] = lo;
      slots[1][wave] = hi;
    }
    __syncthreads();
    lo = fminf(fminf(slots[0][0], slots[0][1]), fminf(slots[0][2], slots[0][3]));
    hi = fmaxf(fmaxf(slots[1][0], slots[1][1]), fmaxf(slots[1][2], slots[1][3]));
  }
  if (lane == 0 && sub == 0) {
    if (rmin) rmin[row] = lo;
    if (rmax) rmax[row] = hi;
    if (rabs) rabs[row] = fmaxf(fabsf(lo), fabsf(hi));
  }
}

// ------------------------------------------------------------------------------ static weight quantisation
__global__ __launch_bounds__(256) void weight_quant_kernel(const void* w, int dt, const float* delta, const float* zp,
                                                           int qmin, int qmax, int8_t* q8, float* deq, int64_t rows,
                                                           int cols) {
  const int cpr = cols / 8;
  const int64_t total = rows * (int64_t)cpr;
  for (int64_t ch = (int64_t)blockIdx.x * 256 + threadIdx.x; ch < total; ch += (int64_t)gridDim.x * 256) {
    const int64_t row = ch / cpr;
    const int c0 = (int)(ch - row * cpr) * 8;
    float v[8];
    load8_rt(w, dt, row * cols + c0, v);
    const float d = delta[row], z = zp[row];
    int qi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      // rne(w/delta) - zp, clamp   (base_quantizer.py:64-67); true division: runs once per model
      float t = rintf(v[j] / d) - z;
      t = fminf(fmaxf(t, (float)qmin), (float)qmax);  // the reference's (loose) clamp for the fake-quant value
      qi[j] = (int)fminf(fmaxf(t, -128.f), 127.f);     // int8 storage saturates on top of it
      v[j] = (t + z) * d;
    }
    if (q8)
      *reinterpret_cast<uint2*>(q8 + row * cols + c0) =
          make_uint2(pack4_i8(qi[0], qi[1], qi[2], qi[3]), pack4_i8(qi[4], qi[5], qi[6], qi[7]));
    if (deq) Io<F32>::store8(deq, row * cols + c0, v);
  }
}

// x * premul (-> LayerNorm + modulate first when `ln`) -> fp output and / or int8 quantise: the transform entry points of
// rotate.hip with had_k == 0 (channel scale without rotation: SmoothQuant, Q/smooth_quant/sq_quant_layer.py:52-60).
// ------------------------------------------------------------------------------ wide 16-bit rows: one WAVE per row
// The FFN hidden of the 1.3B model ([L, 8960] bf16 -> int8, with or without the tanh-GELU: quant_sum / gelu_quant_sum) through
// the general kernel is four waves per row, two workgroup barriers per row (row maximum, integer sum), five chunk slots per
// lane of which the row fills 4.4, and a control-flow graph that carries every option of the general entry: 26 vector
// instructions per element (profiles/r03_h_rowwise_sq.csv) for a job that needs about ten.  Here a wave owns a row: NCH
// 16-byte chunks per lane (chunk lane + 64 i), all requested up front, the row kept in registers, both reductions inside
// the wave (DPP / permlane swaps: no LDS, no barrier), one dtype per instantiation, no other options.  Bit-identical
// outputs (same operations per element, same reduction tree per wave; the cross-wave LDS step of the general kernel is a
// max / an integer sum).
template <typename T, int NCH, bool GELU>
__global__ __launch_bounds__(256, 2) void quant_rows_wave_kernel(const void* x, int8_t* q, void* scale_out, void* sum_out, int vec_dtype,
                                                                 int64_t rows, int cols) {
  static_assert(NCH % 2 == 0, "chunks are owned in pairs");
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;  // whole wave (no barriers in this kernel)
  const int64_t rbase = row * (int64_t)cols;
  const int last = cols / 8 - 1;  // last valid chunk
  // a lane owns PAIRS of adjacent chunks (pair lane + 64 h = chunks 2 pair, 2 pair + 1): 32 contiguous input bytes and, for the
  // codes, ONE 16-byte store per pair (8-byte stores run at about half the rate per byte)
  float v[NCH][8];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int ch = 2 * (lane + 64 * (i >> 1)) + (i & 1);
    Io<T>::load8(x, rbase + (int64_t)(ch <= last ? ch : last) * 8, v[i]);  // (a lane past the row end re-reads the last chunk)
  }
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const bool ok = 2 * (lane + 64 * (i >> 1)) + (i & 1) <= last;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
      float t0 = v[i][j], t1 = v[i][j + 1];
      if (GELU) {
#ifndef WANQ_QW_SCALAR_GELU
        // gelu_tanh_fast_f32 on element PAIRS: the same operations, the plain ones as packed fp32 (bit-identical)
        typedef float v2f __attribute__((ext_vector_type(2)));
        const v2f x = {t0, t1}, c3 = {-0.10294324f, -0.10294324f}, c1 = {-2.3022082f, -2.3022082f}, one = {1.0f, 1.0f};
        const v2f u = x * __builtin_elementwise_fma(x * x, c3, c1);
        const v2f d = one + (v2f){__builtin_amdgcn_exp2f(u.x), __builtin_amdgcn_exp2f(u.y)};
        const v2f r = x * (v2f){__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        t0 = r.x;
        t1 = r.y;
#else
        t0 = gelu_tanh_fast_f32(t0);
        t1 = gelu_tanh_fast_f32(t1);
#endif
      }
      t0 = ok ? t0 : 0.f;
      t1 = ok ? t1 : 0.f;
      v[i][j] = t0;
      v[i][j + 1] = t1;
      m = fmaxf(m, fmaxf(fabsf(t0), fabsf(t1)));
    }
  }
  const float amax = wave_max(m);
  float scale = amax / 127.0f;
  if (scale < 1e-6f) scale = 1e-6f;  // qdiff eps rule (base_quantizer.py:122-127)
  const float inv = 1.0f / scale;
  int isum = 0;
#pragma unroll
  for (int h = 0; h < NCH / 2; ++h) {
    uint32_t pa[2], pb[2];
    quantN_pack_rne<8>(v[2 * h], scale, inv, pa);
    quantN_pack_rne<8>(v[2 * h + 1], scale, inv, pb);
    isum = __builtin_amdgcn_sdot4((int)pa[0], 0x01010101, isum, false);
    isum = __builtin_amdgcn_sdot4((int)pa[1], 0x01010101, isum, false);
    isum = __builtin_amdgcn_sdot4((int)pb[0], 0x01010101, isum, false);
    isum = __builtin_amdgcn_sdot4((int)pb[1], 0x01010101, isum, false);
    const int ch = 2 * (lane + 64 * h);
    int8_t* dst = q + rbase + (int64_t)ch * 8;
    if (ch + 1 <= last) *reinterpret_cast<uint4*>(dst) = make_uint4(pa[0], pa[1], pb[0], pb[1]);
    else if (ch <= last) *reinterpret_cast<uint2*>(dst) = make_uint2(pa[0], pa[1]);
  }
  if (sum_out) {
    const int tot = wave_sum(isum);
    if (lane == 0) vec_store(sum_out, vec_dtype, row, (float)tot * scale);
  }
  if (lane == 0) vec_store(scale_out, vec_dtype, row, scale);
}

// cols in (8704, 9216] (the 1.3B FFN width 8960 = 17.5 chunks per lane): NCH = 18
static bool launch_quant_rows_wave(const void* x, int x_dtype, int8_t* q, void* scale, void* sum, int vec_dtype, int64_t rows, int cols, int act,
                                   hipStream_t st) {
  static const bool off = getenv("WANQ_QUANT_WAVE_OFF") != nullptr;
  if (off || cols <= 8704 || cols > 9216 || (x_dtype != WANQ_BF16 && x_dtype != WANQ_F16)) return false;
  const dim3 grid((unsigned)((rows + 3) / 4));
#define WANQ_QW(T, G) hipLaunchKernelGGL((quant_rows_wave_kernel<T, 18, G>), grid, dim3(256), 0, st, x, q, scale, sum, vec_dtype, rows, cols)
  if (x_dtype == WANQ_BF16) { if (act) WANQ_QW(BF16, true); else WANQ_QW(BF16, false); }
  else { if (act) WANQ_QW(F16, true); else WANQ_QW(F16, false); }
#undef WANQ_QW
  return true;
}

int premul_quant_rows(bool ln, const void* x, int x_dtype, const void* gamma, const void* mshift, const void* mscale,
                      int64_t mod_stride, int64_t rows_per_batch, float eps, const float* premul, void* out_fp, int out_dtype,
                      int8_t* q, void* scale, void* sum, int vec_dtype, int64_t rows, int cols, hipStream_t st, const char* what) {
  if (int e = check_rows_cols(what, rows, cols)) return e;
  if (rows == 0) return WANQ_OK;
  RowParams p{};
  p.levels = 127.0f; p.floor = 1e-6f;
  p.x = x; p.x_dtype = x_dtype; p.gamma = gamma; p.mshift = mshift; p.mscale = mscale; p.mod_dtype = WANQ_F32;
  p.mod_stride = mod_stride; p.rows_per_batch = rows_per_batch; p.eps = eps; p.out_fp = out_fp; p.out_dtype = out_dtype;
  p.q = q; p.scale = scale; p.sum = sum; p.vec_dtype = vec_dtype; p.rows = rows; p.cols = cols; p.premul = premul;
  return ln ? launch_rowwise<true>(p, st, what) : launch_rowwise<false>(p, st, what);
}

// ------------------------------------------------------------------------------ reference-format int8 export
// quantize_and_save_weight_ (W/wan/quant_wanx_cuda.py:39-53): everything in HALF precision --
//   int8 = clamp( round( f16(w) / f16(delta) ) - f16(zp), -128, 127 )
// torch evaluates the fp16 quotient as fl16(fl32(a / b)); round() and the subtraction are exact on these magnitudes.
__global__ __launch_bounds__(256) void weight_export_f16_kernel(const void* w, int dt, const __half* delta, const __half* zp,
                                                                int8_t* q8, int64_t rows, int cols) {
  const int cpr = cols / 8;
  const int64_t total = rows * (int64_t)cpr;
  for (int64_t ch = (int64_t)blockIdx.x * 256 + threadIdx.x; ch < total; ch += (int64_t)gridDim.x * 256) {
    const int64_t row = ch / cpr;
    const int c0 = (int)(ch - row * cpr) * 8;
    float v[8];
    load8_rt(w, dt, row * cols + c0, v);
    const float d = __half2float(delta[row]), z = __half2float(zp[row]);
    int qi[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float w16 = __half2float(__float2half_rn(v[j]));        // fp_module.weight.to(torch.float16)
      const float quo = __half2float(__float2half_rn(w16 / d));     // fp16 division
      const float t = __half2float(__float2half_rn(rintf(quo) - z));
      qi[j] = (int)fminf(fmaxf(t, -128.f), 127.f);
    }
    *reinterpret_cast<uint2*>(q8 + row * cols + c0) =
        make_uint2(pack4_i8(qi[0], qi[1], qi[2], qi[3]), pack4_i8(qi[4], qi[5], qi[6], qi[7]));
  }
}

}  // namespace wanq

using namespace wanq;

extern "C" int wanq_weight_export_f16(const void* w, int w_dtype, const void* delta_f16, const void* zp_f16, int8_t* q8,
                                      int64_t rows, int cols, void* stream) {
  WANQ_REQUIRE(w && delta_f16 && zp_f16 && q8, WANQ_E_ARG, "wanq_weight_export_f16: NULL pointer");
  WANQ_REQUIRE(is_fp(w_dtype), WANQ_E_ARG, "wanq_weight_export_f16: bad dtype %d", w_dtype);
  if (int e = check_rows_cols("wanq_weight_export_f16", rows, cols)) return e;
  if (rows == 0) return WANQ_OK;
  const int64_t total = rows * (cols / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(weight_export_f16_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, w_dtype,
                     static_cast<const __half*>(delta_f16), static_cast<const __half*>(zp_f16), q8, rows, cols);
  return check_launch("wanq_weight_export_f16");
}

extern "C" int wanq_quant_rows(const void* x, int x_dtype, int8_t* q, void* scale, void* sum, int vec_dtype,
                               int64_t rows, int cols, int act, int static_amax, void* stream) {
  WANQ_REQUIRE(x && q && scale, WANQ_E_ARG, "wanq_quant_rows: x, q and scale must be non-NULL");
  WANQ_REQUIRE(is_fp(x_dtype) && is_vec(vec_dtype), WANQ_E_ARG, "wanq_quant_rows: bad dtype code (x=%d vec=%d)", x_dtype, vec_dtype);
  WANQ_REQUIRE(act == 0 || act == 1, WANQ_E_ARG, "wanq_quant_rows: act must be 0 or 1");
  if (int e = check_rows_cols("wanq_quant_rows", rows, cols)) return e;
  if (rows == 0) return WANQ_OK;
  if (!static_amax && launch_quant_rows_wave(x, x_dtype, q, scale, sum, vec_dtype, rows, cols, act, (hipStream_t)stream))
    return check_launch("wanq_quant_rows");
  RowParams p{};
  p.levels = 127.0f; p.floor = 1e-6f;
  p.x = x; p.x_dtype = x_dtype; p.q = q; p.scale = scale; p.sum = sum; p.vec_dtype = vec_dtype;
  p.rows = rows; p.cols = cols; p.act = act; p.static_amax = static_amax; p.rows_per_batch = 1;
  return launch_rowwise<false>(p, (hipStream_t)stream, "wanq_quant_rows");
}

extern "C" int wanq_quant_rows_levels(const void* x, int x_dtype, int8_t* q, void* scale, void* sum, int vec_dtype, int64_t rows,
                                      int cols, int n_levels, float floor, void* stream) {
  WANQ_REQUIRE(x && q && scale, WANQ_E_ARG, "wanq_quant_rows_levels: x, q and scale must be non-NULL");
  WANQ_REQUIRE(is_fp(x_dtype) && is_vec(vec_dtype), WANQ_E_ARG, "wanq_quant_rows_levels: bad dtype code (x=%d vec=%d)", x_dtype, vec_dtype);
  WANQ_REQUIRE(n_levels >= 1 && n_levels <= 127, WANQ_E_ARG, "wanq_quant_rows_levels: n_levels=%d must be in [1, 127] (int8 codes)", n_levels);
  WANQ_REQUIRE(floor >= 0.f && floor == floor, WANQ_E_ARG, "wanq_quant_rows_levels: floor must be >= 0");
  if (int e = check_rows_cols("wanq_quant_rows_levels", rows, cols)) return e;
  if (rows == 0) return WANQ_OK;
  RowParams p{};
  p.x = x; p.x_dtype = x_dtype; p.q = q; p.scale = scale; p.sum = sum; p.vec_dtype = vec_dtype;
  p.rows = rows; p.cols = cols; p.rows_per_batch = 1; p.levels = (float)n_levels; p.floor = floor;
  return launch_rowwise<false>(p, (hipStream_t)stream, "wanq_quant_rows_levels");
}

extern "C" int wanq_layernorm_rows(const void* x, int x_dtype, const void* gamma, const void* mshift,
                                   const void* mscale, int mod_dtype, int64_t mod_stride, int64_t rows_per_batch,
                                   float eps, void* out_fp, int out_dtype, int8_t* q, void* scale, void* sum,
                                   int vec_dtype, int64_t rows, int cols, void* stream) {
  WANQ_REQUIRE(x && (out_fp || q), WANQ_E_ARG, "wanq_layernorm_rows: need x and at least one of out_fp / q");
  WANQ_REQUIRE(is_fp(x_dtype), WANQ_E_ARG, "wanq_layernorm_rows: bad x dtype %d", x_dtype);
  WANQ_REQUIRE(!(gamma || mshift || mscale) || is_fp(mod_dtype), WANQ_E_ARG, "wanq_layernorm_rows: bad mod dtype %d", mod_dtype);
  WANQ_REQUIRE(!out_fp || is_fp(out_dtype), WANQ_E_ARG, "wanq_layernorm_rows: bad out dtype %d", out_dtype);
  WANQ_REQUIRE(!q || (scale && is_vec(vec_dtype)), WANQ_E_ARG, "wanq_layernorm_rows: q needs scale and a valid vec dtype");
  WANQ_REQUIRE(rows_per_batch >= 1, WANQ_E_ARG, "wanq_layernorm_rows: rows_per_batch must be >= 1");
  if (int e = check_rows_cols("wanq_layernorm_rows", rows, cols)) return e;
  if (rows == 0) return WANQ_OK;
  RowParams p{};
  p.levels = 127.0f; p.floor = 1e-6f;
  p.x = x; p.x_dtype = x_dtype; p.gamma = gamma; p.mshift = mshift; p.mscale = mscale; p.mod_dtype = mod_dtype;
  p.mod_stride = mod_stride; p.rows_per_batch = rows_per_batch; p.eps = eps; p.out_fp = out_fp; p.out_dtype = out_dtype;
  p.q = q; p.scale = scale; p.sum = sum; p.vec_dtype = vec_dtype; p.rows = rows; p.cols = cols;
  return launch_rowwise<true>(p, (hipStream_t)stream, "wanq_layernorm_rows");
}

extern "C" int wanq_gate_residual(const void* y, int y_dtype, const void* gate, int gate_dtype, int64_t gate_stride,
                                  const void* residual, int res_dtype, void* out, int out_dtype, int64_t rows,
                                  int cols, int64_t rows_per_batch, void* stream) {
  WANQ_REQUIRE(y && gate && residual && out, WANQ_E_ARG, "wanq_gate_residual: NULL pointer");
  WANQ_REQUIRE(is_fp(y_dtype) && is_fp(gate_dtype) && is_fp(res_dtype) && is_fp(out_dtype), WANQ_E_ARG,
               "wanq_gate_residual: bad dtype code");
  WANQ_REQUIRE(rows_per_batch >= 1, WANQ_E_ARG, "wanq_gate_residual: rows_per_batch must be >= 1");
  if (int e = check_rows_cols("wanq_gate_residual", rows, cols)) return e;
  if (rows == 0) return WANQ_OK;
  GateParams p{y, gate, residual, out, y_dtype, gate_dtype, res_dtype, out_dtype, gate_stride, rows_per_batch, rows, cols};
  const int64_t total = rows * (cols / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(gate_residual_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("wanq_gate_residual");
}

extern "C" int wanq_col_absmax(const void* x, int x_dtype, float* colmax, int64_t rows, int cols, void* stream) {
  WANQ_REQUIRE(x && colmax, WANQ_E_ARG, "wanq_col_absmax: NULL pointer");
  WANQ_REQUIRE(is_fp(x_dtype), WANQ_E_ARG, "wanq_col_absmax: bad dtype %d", x_dtype);
  WANQ_REQUIRE(cols >= 8 && cols % 8 == 0, WANQ_E_SHAPE, "wanq_col_absmax: cols=%d must be a multiple of 8", cols);
  WANQ_REQUIRE(rows >= 0 && rows < (1ll << 40), WANQ_E_SHAPE, "wanq_col_absmax: rows out of range");
  if (rows == 0) return WANQ_OK;
  const unsigned panels = (unsigned)((cols + 511) / 512);
  // ~2048 workgroups in total, at least 64 rows each
  int64_t slabs = 2048 / panels;
  if (slabs < 1) slabs = 1;
  int64_t rpb = (rows + slabs - 1) / slabs;
  if (rpb < 64) rpb = 64;
  slabs = (rows + rpb - 1) / rpb;
  WANQ_REQUIRE(slabs <= 65535, WANQ_E_SHAPE, "wanq_col_absmax: too many row slabs");
  dim3 grid(panels, (unsigned)slabs);
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype == WANQ_F16) hipLaunchKernelGGL(col_absmax_kernel<F16>, grid, dim3(256), 0, st, x, colmax, rows, cols, (int)rpb);
  else if (x_dtype == WANQ_BF16) hipLaunchKernelGGL(col_absmax_kernel<BF16>, grid, dim3(256), 0, st, x, colmax, rows, cols, (int)rpb);
  else hipLaunchKernelGGL(col_absmax_kernel<F32>, grid, dim3(256), 0, st, x, colmax, rows, cols, (int)rpb);
  return check_launch("wanq_col_absmax");
}

// v fake-quantisation of the reference's quantized attention: DynamicQuantizer over ALL TOKENS for every (head, channel) --
// `self.v_quantizer(v.permute([0,1,3,2]).reshape([-1, N_token]))`, ViDiT-Q/examples/Wan2.1/models/quant_opensora.py:438-440 --
// i.e. per COLUMN of the token-major [tokens, heads*head_dim] tensor: delta_c = max(absmax_c / n, 1e-6), n = 2^(b-1) - 1,
// y = clamp(rne(x / delta_c), -n-1, n) * delta_c.  colmax comes from wanq_col_absmax over the same rows.
namespace wanq {
__global__ __launch_bounds__(256) void fake_quant_cols_kernel(const void* x, int x_dt, const float* colmax, void* out, int out_dt,
                                                              float nlev, int64_t rows, int cols) {
  const int cpr = cols / 8;
  const int64_t total = rows * (int64_t)cpr;
  for (int64_t ch = (int64_t)blockIdx.x * 256 + threadIdx.x; ch < total; ch += (int64_t)gridDim.x * 256) {
    const int64_t row = ch / cpr;
    const int c0 = (int)(ch - row * cpr) * 8;
    float v[8], m[8];
    load8_rt(x, x_dt, row * cols + c0, v);
    Io<F32>::load8(colmax, c0, m);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float d = m[j] / nlev;
      if (d < 1e-6f) d = 1e-6f;
      v[j] = __builtin_amdgcn_fmed3f(rintf(v[j] / d), -nlev - 1.f, nlev) * d;
    }
    store8_rt(out, out_dt, row * cols + c0, v);
  }
}
}  // namespace wanq

extern "C" int wanq_fake_quant_cols(const void* x, int x_dtype, const float* colmax, void* out, int out_dtype, int n_bits,
                                    int64_t rows, int cols, void* stream) {
  WANQ_REQUIRE(x && colmax && out, WANQ_E_ARG, "wanq_fake_quant_cols: NULL pointer");
  WANQ_REQUIRE(is_fp(x_dtype) && is_fp(out_dtype), WANQ_E_ARG, "wanq_fake_quant_cols: bad dtype code");
  WANQ_REQUIRE(n_bits >= 2 && n_bits <= 8, WANQ_E_ARG, "wanq_fake_quant_cols: n_bits=%d must be in [2, 8]", n_bits);
  WANQ_REQUIRE(cols >= 8 && cols % 8 == 0, WANQ_E_SHAPE, "wanq_fake_quant_cols: cols=%d must be a multiple of 8", cols);
  WANQ_REQUIRE(rows >= 0 && rows < (1ll << 40), WANQ_E_SHAPE, "wanq_fake_quant_cols: rows out of range");
  if (rows == 0) return WANQ_OK;
  const int64_t total = rows * (cols / 8);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(fake_quant_cols_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, x_dtype, colmax, out,
                     out_dtype, (float)((1 << (n_bits - 1)) - 1), rows, cols);
  return check_launch("wanq_fake_quant_cols");
}

// ------------------------------------------------------------------------------ fake-quant with a precomputed delta
// DynamicQuantizer.forward_with_quant_params (Q/base/base_quantizer.py:164-206): elementwise, delta has x's shape; optional per-element
// bit-widths (`mixed_precision`).  IEEE divisions, as the reference's torch ops (an HBM-bound pass: 12-16 B per element).
namespace wanq {
__global__ __launch_bounds__(256) void fake_quant_delta_kernel(const void* x, int x_dt, const float* delta, const int32_t* bits, void* out,
                                                               int out_dt, float levels, int64_t chunks) {
  for (int64_t ch = (int64_t)blockIdx.x * 256 + threadIdx.x; ch < chunks; ch += (int64_t)gridDim.x * 256) {
    float v[8], d[8];
    load8_rt(x, x_dt, ch * 8, v);
    Io<F32>::load8(delta, ch * 8, d);
    int b[8];
    if (bits) {
      const int4 b0 = *reinterpret_cast<const int4*>(bits + ch * 8), b1 = *reinterpret_cast<const int4*>(bits + ch * 8 + 4);
      b[0] = b0.x; b[1] = b0.y; b[2] = b0.z; b[3] = b0.w; b[4] = b1.x; b[5] = b1.y; b[6] = b1.z; b[7] = b1.w;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float dj = d[j] < 1e-6f ? 1e-6f : d[j];  // :181-189
      if (bits) {  // levels 2^bits - 1; 0 bits: computed as 8 bits, then masked (:174-178, :203-204); clipped from above only (:194)
        const bool zero = b[j] == 0;
        const float nl = zero ? 255.f : (float)((1u << b[j]) - 1u);
        dj = dj / nl;
        const float xi = rintf(v[j] / dj);
        v[j] = zero ? 0.f : (xi > nl ? nl : xi) * dj;
      } else {     // :196-199
        dj = dj / levels;
        v[j] = __builtin_amdgcn_fmed3f(rintf(v[j] / dj), 0.f, levels) * dj;
      }
    }
    store8_rt(out, out_dt, ch * 8, v);
  }
}
}  // namespace wanq

extern "C" int wanq_fake_quant_with_delta(const void* x, int x_dtype, const float* delta, const int32_t* bits, void* out, int out_dtype,
                                          int n_bits, int64_t n, void* stream) {
  WANQ_REQUIRE(x && delta && out, WANQ_E_ARG, "wanq_fake_quant_with_delta: NULL pointer");
  WANQ_REQUIRE(is_fp(x_dtype) && is_fp(out_dtype), WANQ_E_ARG, "wanq_fake_quant_with_delta: bad dtype code");
  WANQ_REQUIRE(n_bits >= 2 && n_bits <= 16, WANQ_E_ARG, "wanq_fake_quant_with_delta: n_bits=%d must be in [2, 16]", n_bits);
  WANQ_REQUIRE(n >= 0 && n % 8 == 0 && n < (1ll << 40), WANQ_E_SHAPE, "wanq_fake_quant_with_delta: n=%lld must be a multiple of 8", (long long)n);
  if (n == 0) return WANQ_OK;
  int64_t blocks = (n / 8 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  // symmetric quantiser: n_levels = 2^(b-1) - 1, the unsigned range of this method is 2 n_levels + 1 = 2^b - 1
  hipLaunchKernelGGL(fake_quant_delta_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, x_dtype, delta, bits, out, out_dtype,
                     (float)((1u << n_bits) - 1u), n / 8);
  return check_launch("wanq_fake_quant_with_delta");
}

extern "C" int wanq_row_minmax(const void* w, int w_dtype, float* row_min, float* row_max, float* row_absmax,
                               int64_t rows, int cols, void* stream) {
  WANQ_REQUIRE(w && (row_min || row_max || row_absmax), WANQ_E_ARG, "wanq_row_minmax: NULL pointer");
  WANQ_REQUIRE(is_fp(w_dtype), WANQ_E_ARG, "wanq_row_minmax: bad dtype %d", w_dtype);
  if (int e = check_rows_cols("wanq_row_minmax", rows, cols)) return e;
  if (rows == 0) return WANQ_OK;
  hipStream_t st = (hipStream_t)stream;
  const int chunks = cols / 8;
  if (chunks <= 256)
    hipLaunchKernelGGL((row_minmax_kernel<1, 4>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, w, w_dtype, row_min, row_max, row_absmax, rows, cols);
  else
    hipLaunchKernelGGL((row_minmax_kernel<4, 8>), dim3((unsigned)rows), dim3(256), 0, st, w, w_dtype, row_min, row_max, row_absmax, rows, cols);
  return check_launch("wanq_row_minmax");
}

extern "C" int wanq_weight_quant(const void* w, int w_dtype, const float* delta, const float* zp, int qmin, int qmax,
                                 int8_t* q8, float* deq, int64_t rows, int cols, void* stream) {
  WANQ_REQUIRE(w && delta && zp && (q8 || deq), WANQ_E_ARG, "wanq_weight_quant: NULL pointer");
  WANQ_REQUIRE(is_fp(w_dtype), WANQ_E_ARG, "wanq_weight_quant: bad dtype %d", w_dtype);
  WANQ_REQUIRE(qmin < qmax, WANQ_E_ARG, "wanq_weight_quant: bad clamp range [%d,%d]", qmin, qmax);
  if (int e = check_rows_cols("wanq_weight_quant", rows, cols)) return e;
  if (rows == 0) return WANQ_OK;
  const int64_t total = rows * (cols / 8);
  const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(weight_quant_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, w_dtype, delta, zp, qmin, qmax, q8, deq, rows, cols);
  return check_launch("wanq_weight_quant");
}

// ------------------------------------------------------------------------------ 4-bit weight storage
// Packed layout (ours; the reference ships no packer and its QServe layout is an NVIDIA ldmatrix interleave): row-major
// [N, K/2] bytes, K % 32 == 0.  Each group of 32 consecutive codes takes 16 bytes = 4 dwords (P0a, P1a, P0b, P1b); for a
// 16-code half e[0..15] (a = codes 0-15 of the group, b = codes 16-31), codes biased to unsigned nibbles u = e + bias:
//     P0 byte i = u[i] | u[4+i] << 4,    P1 byte i = u[8+i] | u[12+i] << 4        (i = 0..3)
// so that `P & 0x0f0f0f0f` and `(P >> 4) & 0x0f0f0f0f` ARE the four dwords of an int8 MFMA operand (16 consecutive k):
// wanq_gemm_w4a8 reads 16 packed bytes per lane from LDS and gets two MFMA operands for six VALU instructions.
__device__ __forceinline__ uint32_t w4_nibbles(uint32_t d, uint32_t flip) { return (d ^ flip) & 0x0f0f0f0fu; }

__global__ __launch_bounds__(256) void pack_w4_kernel(const int8_t* q, uint8_t* packed, int bias, int64_t total32) {
  const uint32_t flip = bias ? 0x08080808u : 0u;  // low nibble of (e + 8) = low nibble of e with bit 3 flipped
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total32; i += (int64_t)gridDim.x * 256) {
    const uint4 a = *reinterpret_cast<const uint4*>(q + i * 32), b = *reinterpret_cast<const uint4*>(q + i * 32 + 16);
    uint4 o;
    o.x = w4_nibbles(a.x, flip) | (w4_nibbles(a.y, flip) << 4);
    o.y = w4_nibbles(a.z, flip) | (w4_nibbles(a.w, flip) << 4);
    o.z = w4_nibbles(b.x, flip) | (w4_nibbles(b.y, flip) << 4);
    o.w = w4_nibbles(b.z, flip) | (w4_nibbles(b.w, flip) << 4);
    *reinterpret_cast<uint4*>(packed + i * 16) = o;
  }
}

__device__ __forceinline__ uint32_t w4_to_i8(uint32_t u4, int bias) {  // four unsigned nibbles (one per byte) -> int8 codes u - bias
  uint32_t r = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) r |= (uint32_t)(((int)((u4 >> (8 * b)) & 0xf) - bias) & 0xff) << (8 * b);
  return r;
}

__global__ __launch_bounds__(256) void unpack_w4_kernel(const uint8_t* packed, int8_t* q, int bias, int64_t total32) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total32; i += (int64_t)gridDim.x * 256) {
    const uint4 p = *reinterpret_cast<const uint4*>(packed + i * 16);
    const uint32_t m = 0x0f0f0f0fu;
    *reinterpret_cast<uint4*>(q + i * 32) =
        make_uint4(w4_to_i8(p.x & m, bias), w4_to_i8((p.x >> 4) & m, bias), w4_to_i8(p.y & m, bias), w4_to_i8((p.y >> 4) & m, bias));
    *reinterpret_cast<uint4*>(q + i * 32 + 16) =
        make_uint4(w4_to_i8(p.z & m, bias), w4_to_i8((p.z >> 4) & m, bias), w4_to_i8(p.w & m, bias), w4_to_i8((p.w >> 4) & m, bias));
  }
}

extern "C" int wanq_pack_w4(const int8_t* q, uint8_t* packed, int bias, int64_t rows, int cols, void* stream) {
  WANQ_REQUIRE(q && packed, WANQ_E_ARG, "wanq_pack_w4: NULL pointer");
  WANQ_REQUIRE(cols >= 32 && cols % 32 == 0, WANQ_E_SHAPE, "wanq_pack_w4: cols=%d must be a multiple of 32", cols);
  WANQ_REQUIRE(rows >= 0 && (bias == 0 || bias == 8), WANQ_E_ARG, "wanq_pack_w4: bias must be 0 (unsigned codes) or 8 (signed codes)");
  const int64_t total32 = rows * (cols / 32);
  if (total32 == 0) return WANQ_OK;
  const unsigned grid = (unsigned)((total32 + 255) / 256 < 4096 ? (total32 + 255) / 256 : 4096);
  hipLaunchKernelGGL(pack_w4_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, q, packed, bias, total32);
  return check_launch("wanq_pack_w4");
}

extern "C" int wanq_unpack_w4(const uint8_t* packed, int8_t* q, int bias, int64_t rows, int cols, void* stream) {
  WANQ_REQUIRE(q && packed, WANQ_E_ARG, "wanq_unpack_w4: NULL pointer");
  WANQ_REQUIRE(cols >= 32 && cols % 32 == 0, WANQ_E_SHAPE, "wanq_unpack_w4: cols=%d must be a multiple of 32", cols);
  WANQ_REQUIRE(rows >= 0 && (bias == 0 || bias == 8), WANQ_E_ARG, "wanq_unpack_w4: bias must be 0 (unsigned codes) or 8 (signed codes)");
  const int64_t total32 = rows * (cols / 32);
  if (total32 == 0) return WANQ_OK;
  const unsigned grid = (unsigned)((total32 + 255) / 256 < 4096 ? (total32 + 255) / 256 : 4096);
  hipLaunchKernelGGL(unpack_w4_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, packed, q, bias, total32);
  return check_launch("wanq_unpack_w4");
}
