// Attention front-end: RMSNorm over the full model dim (WanRMSNorm, wan/modules/model.py:73-89) fused with
// the 3-axis rotary embedding (rope_apply, model.py:43-70) for q and k.  Row-wise, HBM-bound: a row is
// read once, kept in registers across the sum-of-squares reduction, rotated, written once.
//
// rope table: fp32 [positions, head_dim/2, 2] = (cos, sin) per token position and complex pair; the host
// builds it once per (F,H,W) grid in float64 (as the reference does its whole rotation) and rounds to fp32.
// Rows at or beyond `positions` (sequence padding) are normalised but not rotated, as in the reference.
#include "wanq_common.h"

namespace wanq {

struct PrepParams {
  const void* x;
  void* out;
  const float* weight;
  const float* rope;
  int x_dtype, out_dtype;
  int64_t rows, rows_per_batch, positions;
  int cols, head_dim;
  float eps;
  // optional int8 form for the int8 Q.K^T attention (wanq_attention_qk8_fwd): per-(token, head) symmetric 8-bit codes of the
  // normalised + rotated row (DynamicQuantizer semantics on [tokens*heads, head_dim] rows, Q/base/quant_attn.py:168-174 /
  // W/models/quant_opensora.py:431-436) and two fp32 planes [heads][scale_stride]: delta, and -12582912 * delta (the
  // constant the attention kernel's dequantising fma takes, see attention.hip)
  int8_t* q8;
  float* qscale;
  int64_t scale_stride;
  // optional scattered store (SC): head h of row r goes to out + head_map[2h] + r * head_map[2h+1] (elements) instead of the
  // row-major [rows, cols] image -- the kernel writes the Ulysses all-to-all send buffers ([P][rows][w] per head chunk) directly
  const int64_t* head_map;
};

__device__ __forceinline__ void prep_load8(const void* base, int dt, int64_t elem, float (&v)[8]) {
  if (dt == WANQ_F16) Io<F16>::load8(base, elem, v);
  else if (dt == WANQ_BF16) Io<BF16>::load8(base, elem, v);
  else Io<F32>::load8(base, elem, v);
}
__device__ __forceinline__ void prep_store8(void* base, int dt, int64_t elem, const float (&v)[8]) {
  if (dt == WANQ_F16) Io<F16>::store8(base, elem, v);
  else if (dt == WANQ_BF16) Io<BF16>::store8(base, elem, v);
  else Io<F32>::store8(base, elem, v);
}

// Q8: also emit the per-(token, head) int8 form (a template flag so that the plain kernel keeps its register count)
template <int WPR, int NCH, bool Q8, bool SC = false>
__global__ __launch_bounds__(256) void rmsnorm_rope_kernel(const PrepParams p) {
  __shared__ float slots[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t row = (WPR == 1) ? (int64_t)blockIdx.x * 4 + wave : (int64_t)blockIdx.x;
  if (WPR == 1 && row >= p.rows) return;
  const int sub = (WPR == 1) ? 0 : wave;
  const int C = p.cols;
  const int64_t rbase = row * (int64_t)C;
  float v[NCH][8];
  bool ok[NCH];
  float ss = 0.f;
  // one dtype branch per row (a branch per chunk serialises the row's loads: see rowwise.hip load_row)
#define PREP_LOAD_ROW(T)                                                       \
  _Pragma("unroll") for (int i = 0; i < NCH; ++i) {                            \
    const int c0 = (sub * 64 + lane + i * 64 * WPR) * 8;                       \
    ok[i] = c0 < C;                                                            \
    if (ok[i]) Io<T>::load8(p.x, rbase + c0, v[i]);                            \
  }
  if (p.x_dtype == WANQ_F32) { PREP_LOAD_ROW(F32) } else if (p.x_dtype == WANQ_BF16) { PREP_LOAD_ROW(BF16) } else { PREP_LOAD_ROW(F16) }
#undef PREP_LOAD_ROW
#pragma unroll
  for (int i = 0; i < NCH; ++i)
    if (ok[i]) {
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[i][j] * v[i][j];
    }
  ss = wave_sum(ss);
  if (WPR > 1) {
    if (lane == 0) slots[wave] = ss;
    __syncthreads();
    ss = slots[0] + slots[1] + slots[2] + slots[3];
  }
  const float rinv = p.weight ? 1.0f / sqrtf(ss / (float)C + p.eps) : 1.0f;  // weight == NULL: rope only
  const int64_t pos = row % p.rows_per_batch;
  const bool rot = p.rope && pos < p.positions;
  const int half = p.head_dim >> 1;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    if (!ok[i]) continue;
    const int c0 = (sub * 64 + lane + i * 64 * WPR) * 8;
    if (p.weight) {
      float w[8];
      Io<F32>::load8(p.weight, c0, w);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = v[i][j] * rinv * w[j];
    }
    if (rot) {
      const int pair0 = (c0 % p.head_dim) >> 1;  // 4 consecutive complex pairs of one head
      float cs[8];
      Io<F32>::load8(p.rope, (pos * half + pair0) * 2, cs);
      // BOTH 16-byte pieces of the table row have landed before the first packed fp32 op.  hipcc otherwise waits for the first
      // piece only (s_waitcnt vmcnt(1)); with the attention-map kernels running beside this kernel (another stream or another
      // process on the GPU) a v_pk_mul_f32 reading the high register of a source pair through op_sel then returned 0 for lanes
      // 48-63 in ~0.4 % of launches while the second piece was still returning: one element per 16-byte chunk came out as a*cos
      // instead of a*cos - b*sin (profiles/r04_z_corun_corruption.txt; micro-victim tools/probes/late_beat.py; 0 of 30000 with
      // this wait).
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(cs[0]), "+v"(cs[1]), "+v"(cs[2]), "+v"(cs[3]), "+v"(cs[4]), "+v"(cs[5]), "+v"(cs[6]), "+v"(cs[7]));
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // roundings fixed in the source (a product, then one fused multiply-add), so that every instantiation of this kernel and
        // every compiler mood gives the same bits: left to -ffp-contract the plain and the int8 form once disagreed
        const float a = v[i][2 * k], b = v[i][2 * k + 1];
        v[i][2 * k] = fmaf(a, cs[2 * k], -__fmul_rn(b, cs[2 * k + 1]));
        v[i][2 * k + 1] = fmaf(a, cs[2 * k + 1], __fmul_rn(b, cs[2 * k]));
      }
    }
    if (SC) {
      const int hd = c0 / p.head_dim;
      prep_store8(p.out, p.out_dtype, p.head_map[2 * hd] + row * p.head_map[2 * hd + 1] + (c0 - hd * p.head_dim), v[i]);
    } else if (p.out) {
      prep_store8(p.out, p.out_dtype, rbase + c0, v[i]);
    }
    if (Q8) {  // head_dim == 128: a head is the 16 chunks of 16 consecutive lanes
      float m = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[i][j]));
      // lane ^ o inside the head's 16 lanes (DPP: wanq_common.h)
      m = fmaxf(m, lane_xor_dpp<1>(m)); m = fmaxf(m, lane_xor_dpp<2>(m)); m = fmaxf(m, lane_xor_dpp<4>(m)); m = fmaxf(m, lane_xor_dpp<8>(m));
      float scale = m / 127.0f;
      if (scale < 1e-6f) scale = 1e-6f;  // qdiff eps rule (base_quantizer.py:122-127)
      uint32_t pk[2];
      quantN_pack_rne<8>(v[i], scale, 1.0f / scale, pk);
      *reinterpret_cast<uint2*>(p.q8 + rbase + c0) = make_uint2(pk[0], pk[1]);
      if ((lane & 15) == 0) {
        const int64_t so = (int64_t)(c0 >> 7) * p.scale_stride + row;
        p.qscale[so] = scale;
        p.qscale[(int64_t)(C >> 7) * p.scale_stride + so] = -12582912.0f * scale;
      }
    }
  }
}

}  // namespace wanq

using namespace wanq;

static int rmsnorm_rope_impl(const void* x, int x_dtype, const float* weight, const float* rope, void* out, int out_dtype,
                             int8_t* q8, float* qscale, int64_t scale_stride, int64_t rows, int cols, int head_dim,
                             int64_t rows_per_batch, int64_t positions, float eps, void* stream, const int64_t* head_map = nullptr) {
  WANQ_REQUIRE(x && (out || q8), WANQ_E_ARG, "wanq_rmsnorm_rope: NULL pointer");
  WANQ_REQUIRE(!q8 || (qscale && head_dim == 128 && cols % 128 == 0 && scale_stride >= rows), WANQ_E_ARG,
               "wanq_rmsnorm_rope_q8: needs qscale, head_dim == 128, cols %% 128 == 0 and scale_stride >= rows");
  WANQ_REQUIRE(is_fp(x_dtype) && (!out || is_fp(out_dtype)), WANQ_E_ARG, "wanq_rmsnorm_rope: bad dtype code");
  WANQ_REQUIRE(weight || rope, WANQ_E_ARG, "wanq_rmsnorm_rope: nothing to do (weight and rope both NULL)");
  WANQ_REQUIRE(cols >= 8 && cols % 8 == 0 && cols <= 16384, WANQ_E_SHAPE, "wanq_rmsnorm_rope: cols=%d must be a multiple of 8 in [8,16384]", cols);
  WANQ_REQUIRE(!rope || (head_dim >= 8 && head_dim % 8 == 0 && cols % head_dim == 0), WANQ_E_SHAPE,
               "wanq_rmsnorm_rope: head_dim=%d must be a multiple of 8 dividing cols=%d", head_dim, cols);
  WANQ_REQUIRE(rows >= 0 && rows < (1ll << 31) && rows_per_batch >= 1, WANQ_E_SHAPE, "wanq_rmsnorm_rope: bad rows");
  if (rows == 0) return WANQ_OK;
  PrepParams p{x, out, weight, rope, x_dtype, out_dtype, rows, rows_per_batch, positions, cols, head_dim > 0 ? head_dim : 8, eps,
               q8, qscale, scale_stride, head_map};
  hipStream_t st = (hipStream_t)stream;
  const int chunks = cols / 8;
#define WANQ_PR(WPR, NCH)                                                                                                  \
  do {                                                                                                                     \
    if (q8) hipLaunchKernelGGL((rmsnorm_rope_kernel<WPR, NCH, true>), dim3((unsigned)((WPR) == 1 ? (rows + 3) / 4 : rows)), dim3(256), 0, st, p); \
    else if (head_map) hipLaunchKernelGGL((rmsnorm_rope_kernel<WPR, NCH, false, true>), dim3((unsigned)((WPR) == 1 ? (rows + 3) / 4 : rows)), dim3(256), 0, st, p); \
    else hipLaunchKernelGGL((rmsnorm_rope_kernel<WPR, NCH, false>), dim3((unsigned)((WPR) == 1 ? (rows + 3) / 4 : rows)), dim3(256), 0, st, p);   \
  } while (0)
  if (chunks <= 64) WANQ_PR(1, 1);
  else if (chunks <= 128) WANQ_PR(1, 2);
  else if (chunks <= 192) WANQ_PR(1, 3);
  else if (chunks <= 256) WANQ_PR(1, 4);
  else if (chunks <= 512) WANQ_PR(4, 2);
  else if (chunks <= 768) WANQ_PR(4, 3);
  else if (chunks <= 1024) WANQ_PR(4, 4);
  else if (chunks <= 1536) WANQ_PR(4, 6);
  else WANQ_PR(4, 8);
#undef WANQ_PR
  return check_launch("wanq_rmsnorm_rope");
}

extern "C" int wanq_rmsnorm_rope(const void* x, int x_dtype, const float* weight, const float* rope, void* out,
                                 int out_dtype, int64_t rows, int cols, int head_dim, int64_t rows_per_batch,
                                 int64_t positions, float eps, void* stream) {
  WANQ_REQUIRE(out, WANQ_E_ARG, "wanq_rmsnorm_rope: NULL pointer");
  return rmsnorm_rope_impl(x, x_dtype, weight, rope, out, out_dtype, nullptr, nullptr, 0, rows, cols, head_dim, rows_per_batch,
                           positions, eps, stream);
}

extern "C" int wanq_rmsnorm_rope_q8(const void* x, int x_dtype, const float* weight, const float* rope, void* out,
                                    int out_dtype, int8_t* q8, float* qscale, int64_t scale_stride, int64_t rows, int cols,
                                    int head_dim, int64_t rows_per_batch, int64_t positions, float eps, void* stream) {
  WANQ_REQUIRE(q8, WANQ_E_ARG, "wanq_rmsnorm_rope_q8: q8 is required");
  return rmsnorm_rope_impl(x, x_dtype, weight, rope, out, out_dtype, q8, qscale, scale_stride, rows, cols, head_dim,
                           rows_per_batch, positions, eps, stream);
}

// RMSNorm + RoPE with the store scattered per head: head h (head_dim columns) of row r lands at
// out + head_map[2h] + r * head_map[2h+1] (elements of out_dtype; head_map is a DEVICE array of 2 * cols/head_dim int64).
// Writes the Ulysses head-scatter send buffers in place of the torch transpose().contiguous() pack (the reference's
// all_to_all_4D, ViDiT-Q/examples/Wan2.1/wan/distributed/xdit_context_parallel.py:147-192 via yunchang, does the same permute
// + contiguous on the host side of the collective).
extern "C" int wanq_rmsnorm_rope_scatter(const void* x, int x_dtype, const float* weight, const float* rope, void* out,
                                         int out_dtype, const int64_t* head_map, int64_t rows, int cols, int head_dim,
                                         int64_t rows_per_batch, int64_t positions, float eps, void* stream) {
  WANQ_REQUIRE(out && head_map, WANQ_E_ARG, "wanq_rmsnorm_rope_scatter: NULL pointer");
  WANQ_REQUIRE(head_dim >= 8 && head_dim % 8 == 0 && cols % head_dim == 0, WANQ_E_SHAPE,
               "wanq_rmsnorm_rope_scatter: head_dim=%d must be a multiple of 8 dividing cols=%d", head_dim, cols);
  return rmsnorm_rope_impl(x, x_dtype, weight, rope, out, out_dtype, nullptr, nullptr, 0, rows, cols, head_dim, rows_per_batch,
                           positions, eps, stream, head_map);
}
