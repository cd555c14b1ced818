// The fp32 ends of a DiT pass: what sits before the first block and after the last one
// (ViDiT-Q/examples/Wan2.1/wan/modules/model.py:580-610 patch / time / text embeddings, :372-400 Head, :633-656 unpatchify;
// SURVEY 8(f)3).  One fp32 tile kernel on the fp32 matrix cores (v_mfma_f32_16x16x4_f32: IEEE fp32 products and sums, no
// reduced-precision operand format), with the data movement of each user folded into its operand loader / its store:
//   * Linear (time_embedding, time_projection, text_embedding):   out = act_out(act_in(x) w^T + b)
//   * patch embedding: Conv3d with kernel == stride is a Linear over non-overlapping patches; the loader gathers a token's patch
//     straight from the latent [C, F, H, W], no patch matrix is ever written
//   * head: LayerNorm (no affine) + (1 + scale) * . + shift with scale / shift = modulation + e, applied while the operand tile
//     is staged (row statistics in a prologue of the same workgroup, two passes: mean, then centred squares), Linear, and the
//     store scattered to the latent layout [C_out, F, H, W] (unpatchify) through LDS so that runs along W stay contiguous.
// All of it is 0.3 % of a denoising step; the point of the kernel is that a pass has no torch / hipBLASLt launches left, not speed:
// 64 x 64 output tiles, K in steps of 16, operands double-buffered through registers, 16 MFMAs per wave and K-step.
#include "wanq_common.h"

namespace wanq {
namespace {

enum { ACT_NONE = 0, ACT_GELU_TANH = 1, ACT_SILU = 2 };
enum { A_PLAIN = 0, A_PATCH = 1, A_LNMOD = 2 };
enum { C_PLAIN = 0, C_UNPATCH = 1 };

constexpr int BM = 64, BN = 64, BK = 16;
// Operand tiles in LDS are row-major [row][k] with the row stride padded to 20 floats (tile kernel, 16 k per step) / 68 floats
// (patch kernel, 64 k): the 16 rows a 16-lane group reads then start in banks 20 r (resp. 4 r) mod 64, four banks each, all
// disjoint.  A lane reads FOUR CONSECUTIVE k of its row with one ds_read_b128 and uses them in four consecutive MFMAs: MFMA step
// kk takes, on the lanes of k-group fk = lane / 16, element k = 4 fk + kk of the 16 -- a permutation of the k order that both
// operands share, so the product is the same sum.
constexpr int LDK = 20;

struct LinParams {
  const float* x;
  const float* w;
  const float* bias;
  float* out;
  int64_t x_rows;    // rows that exist in x; rows [x_rows, M) enter the product as zeros (text_embedding pads its INPUT: model.py:600-605)
  int64_t M;         // rows computed
  int64_t out_rows;  // rows written (>= M; rows [M, out_rows) are zero: the patch embedding's OUTPUT is padded to seq_len, model.py:586-590)
  int N, K;
  int64_t ldx, ldo;
  int in_act, out_act;
  // patch gather (A_PATCH) / unpatchify scatter (C_UNPATCH): latent [C, F, H, W], patch (pt, ph, pw), token grid (gf, gh, gw)
  int C, F, H, W, pt, ph, pw, gh, gw;
  // A_LNMOD
  const float* mod;  // [2, K]: shift row, scale row (Head.modulation)
  const float* e;    // [K]
  float eps;
};

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act(float v, int kind) {
  if (kind == ACT_SILU) return v / (1.0f + expf(-v));
  if (kind == ACT_GELU_TANH) {
    // torch's tanh form: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3)))
    const float u = 0.7978845608028654f * (v + 0.044715f * v * v * v);
    return 0.5f * v * (1.0f + tanhf(u));
  }
  return v;
}

template <int AMODE, int CMODE>
__global__ __launch_bounds__(256) void linear_f32_kernel(const LinParams p) {
  __shared__ __attribute__((aligned(16))) float sA[2][BM][LDK];
  __shared__ __attribute__((aligned(16))) float sW[2][BN][LDK];
  __shared__ float s_mean[BM], s_rstd[BM];
  __shared__ float sC[CMODE == C_UNPATCH ? BM * (BN + 1) : 1];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  if constexpr (AMODE == A_LNMOD) {
    // row statistics of this tile's 64 rows, 16 per wave, as F.layer_norm computes them (biased variance of the centred row)
    for (int r = 0; r < 16; ++r) {
      const int64_t row = m0 + wave * 16 + r;
      float mean = 0.f, rstd = 0.f;
      if (row < p.x_rows) {
        const float* xr = p.x + row * p.ldx;
        float s = 0.f;
        for (int c = lane * 4; c < p.K; c += 256) {
          const float4 v = *reinterpret_cast<const float4*>(xr + c);
          s += (v.x + v.y) + (v.z + v.w);
        }
        mean = wave_sum(s) / (float)p.K;
        float q = 0.f;
        for (int c = lane * 4; c < p.K; c += 256) {
          const float4 v = *reinterpret_cast<const float4*>(xr + c);
          const float a = v.x - mean, b = v.y - mean, cc = v.z - mean, d = v.w - mean;
          q += (a * a + b * b) + (cc * cc + d * d);
        }
        rstd = rsqrtf(wave_sum(q) / (float)p.K + p.eps);
      }
      if (lane == 0) { s_mean[wave * 16 + r] = mean; s_rstd[wave * 16 + r] = rstd; }
    }
    __syncthreads();
  }

  // operand staging: thread -> (row, four consecutive k).  Row-major sources: row = tid / 4 (a wave reads 64-byte row segments);
  // the patch gather: row = tid % 64 (a wave walks 64 consecutive tokens, i.e. contiguous latent runs along W).
  const int a_row = AMODE == A_PATCH ? (tid & 63) : (tid >> 2);
  const int a_kq = AMODE == A_PATCH ? (tid >> 6) * 4 : (tid & 3) * 4;
  const int w_row = tid >> 2, w_kq = (tid & 3) * 4;
  const int64_t arow = m0 + a_row;
  const bool a_ok = arow < p.x_rows;
  const bool w_ok = n0 + w_row < p.N;
  int tf = 0, th = 0, tw = 0;
  if constexpr (AMODE == A_PATCH) {
    tw = (int)(arow % p.gw);
    th = (int)((arow / p.gw) % p.gh);
    tf = (int)(arow / ((int64_t)p.gw * p.gh));
  }
  float a_mean = 0.f, a_rstd = 0.f;
  if constexpr (AMODE == A_LNMOD) { a_mean = s_mean[a_row]; a_rstd = s_rstd[a_row]; }

  auto load_a = [&](int k0) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!a_ok) return v;
    const int k = k0 + a_kq;
    if constexpr (AMODE == A_PATCH) {
      float e[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int kk = k + j;  // (c, dt, dh, dw) as Conv3d's weight.flatten(1) orders them
        const int dw = kk % p.pw; kk /= p.pw;
        const int dh = kk % p.ph; kk /= p.ph;
        const int dt = kk % p.pt;
        const int c = kk / p.pt;
        e[j] = p.x[(((int64_t)c * p.F + tf * p.pt + dt) * p.H + th * p.ph + dh) * p.W + tw * p.pw + dw];
      }
      v = make_float4(e[0], e[1], e[2], e[3]);
    } else {
      v = *reinterpret_cast<const float4*>(p.x + arow * p.ldx + k);
      if constexpr (AMODE == A_LNMOD) {
        const float4 sh = *reinterpret_cast<const float4*>(p.mod + k);
        const float4 sc = *reinterpret_cast<const float4*>(p.mod + p.K + k);
        const float4 ee = *reinterpret_cast<const float4*>(p.e + k);
        // norm(x) * (1 + e[1]) + e[0],  e = modulation + e   (model.py:396-398)
        v.x = (v.x - a_mean) * a_rstd * (1.0f + (sc.x + ee.x)) + (sh.x + ee.x);
        v.y = (v.y - a_mean) * a_rstd * (1.0f + (sc.y + ee.y)) + (sh.y + ee.y);
        v.z = (v.z - a_mean) * a_rstd * (1.0f + (sc.z + ee.z)) + (sh.z + ee.z);
        v.w = (v.w - a_mean) * a_rstd * (1.0f + (sc.w + ee.w)) + (sh.w + ee.w);
      } else if (p.in_act != ACT_NONE) {
        v.x = act(v.x, p.in_act); v.y = act(v.y, p.in_act); v.z = act(v.z, p.in_act); v.w = act(v.w, p.in_act);
      }
    }
    return v;
  };
  auto load_w = [&](int k0) -> float4 {
    if (!w_ok) return make_float4(0.f, 0.f, 0.f, 0.f);
    return *reinterpret_cast<const float4*>(p.w + (int64_t)(n0 + w_row) * p.K + k0 + w_kq);
  };
  auto stage = [&](int buf, const float4& a, const float4& w) {
    *reinterpret_cast<float4*>(&sA[buf][a_row][a_kq]) = a;
    *reinterpret_cast<float4*>(&sW[buf][w_row][w_kq]) = w;
  };

  v4f acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  float4 ra = load_a(0), rw = load_w(0);
  stage(0, ra, rw);
  __syncthreads();
  const int fr = lane & 15, fk = lane >> 4;  // fragment lane: row / column fr, k index fk
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) { ra = load_a((kt + 1) * BK); rw = load_w((kt + 1) * BK); }
    const v4f a = *reinterpret_cast<const v4f*>(&sA[buf][wave * 16 + fr][fk * 4]);
    v4f b[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) b[t] = *reinterpret_cast<const v4f*>(&sW[buf][t * 16 + fr][fk * 4]);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[kk], b[t][kk], acc[t], 0, 0, 0);
    }
    if (kt + 1 < nk) stage(buf ^ 1, ra, rw);  // the other buffer: its last readers passed the barrier of iteration kt - 1
    __syncthreads();
  }

  // accumulator layout: lane holds rows 4 * (lane / 16) + r (r = 0..3), column lane % 16 of each 16 x 16 tile
  if constexpr (CMODE == C_PLAIN) {
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int col = n0 + t * 16 + fr;
      if (col >= p.N) continue;
      const float b = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t row = m0 + wave * 16 + fk * 4 + r;
        if (row < p.out_rows) p.out[row * p.ldo + col] = row < p.M ? act(acc[t][r] + b, p.out_act) : 0.f;
      }
    }
  } else {
    // unpatchify (model.py:633-656): column n = ((p * ph + q) * pw + r) * C_out + c of token (f, h, w) goes to
    // out[c][f * pt + p][h * ph + q][w * pw + r].  The tile is turned in LDS so that consecutive lanes write consecutive
    // (token, r) pairs of one (c, p, q): whole runs along W.
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int col = t * 16 + fr;
      const float b = (p.bias && col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) sC[(wave * 16 + fk * 4 + r) * (BN + 1) + col] = acc[t][r] + b;
    }
    __syncthreads();
    const int per_c = p.pt * p.ph * BM * p.pw;  // elements of one output channel in this tile
    for (int idx = tid; idx < p.C * per_c; idx += 256) {
      int i = idx;
      const int rr = i % p.pw; i /= p.pw;
      const int tok = i % BM; i /= BM;
      const int q = i % p.ph; i /= p.ph;
      const int pp = i % p.pt;
      const int c = i / p.pt;
      const int64_t row = m0 + tok;
      if (row >= p.M) continue;
      const int w_ = (int)(row % p.gw), h_ = (int)((row / p.gw) % p.gh), f_ = (int)(row / ((int64_t)p.gw * p.gh));
      const int n = ((pp * p.ph + q) * p.pw + rr) * p.C + c;
      p.out[(((int64_t)c * p.F + f_ * p.pt + pp) * p.H + h_ * p.ph + q) * p.W + w_ * p.pw + rr] = sC[tok * (BN + 1) + n];
    }
  }
}

// sinusoidal_embedding_1d (model.py:18-28): float64 angles position * 10000^(-i / half), cos half first, cast to fp32
__global__ void sinusoid_kernel(const void* t, int t_kind, float* out, int n, int dim) {
  const int half = dim / 2;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n * half) return;
  const int r = idx / half, i = idx % half;
  double pos;
  if (t_kind == 0) pos = (double)static_cast<const float*>(t)[r];
  else if (t_kind == 1) pos = (double)static_cast<const long long*>(t)[r];
  else if (t_kind == 2) pos = static_cast<const double*>(t)[r];
  else pos = (double)static_cast<const int*>(t)[r];
  const double ang = pos * pow(10000.0, -(double)i / (double)half);
  out[(int64_t)r * dim + i] = (float)cos(ang);
  out[(int64_t)r * dim + half + i] = (float)sin(ang);
}

// Patch embedding when the whole patch fits one operand tile (K = C * pt * ph * pw <= 64; Wan: 16 * 1 * 2 * 2): the 64 tokens'
// patches are gathered ONCE (the fragments then live in 16 registers per lane), the workgroup walks all N output channels in
// chunks of 64 with the weight chunk double-buffered through registers.  The tile kernel above would gather the patches again
// for each of its N / 64 column tiles.
template <int NK4>  // K / 4, a compile-time constant: the MFMA chain of a chunk is straight-line code, its LDS reads issued in a batch
__global__ __launch_bounds__(256) void patch_embed_kernel(const LinParams p) {
  constexpr int LDP = 68;
  __shared__ __attribute__((aligned(16))) float sA[64][LDP];
  __shared__ __attribute__((aligned(16))) float sW[2][64][LDP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fk = lane >> 4;
  const int64_t m0 = (int64_t)blockIdx.x * BM;
  {
    const int tok = tid & 63, kg = (tid >> 6) * 16;
    const int64_t row = m0 + tok;
    const bool ok = row < p.M;
    const int tw = (int)(row % p.gw), th = (int)((row / p.gw) % p.gh), tf = (int)(row / ((int64_t)p.gw * p.gh));
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      int kk = kg + j;
      float v = 0.f;
      if (ok && kk < p.K) {
        const int dw = kk % p.pw; kk /= p.pw;
        const int dh = kk % p.ph; kk /= p.ph;
        const int dt = kk % p.pt;
        const int c = kk / p.pt;
        v = p.x[(((int64_t)c * p.F + tf * p.pt + dt) * p.H + th * p.ph + dh) * p.W + tw * p.pw + dw];
      }
      sA[tok][kg + j] = v;
    }
  }
  const int w_row = tid >> 2, w_k = (tid & 3) * 16;
  auto load_w = [&](int chunk, float4 (&r)[4]) {
    const int n = chunk * BN + w_row;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int k = w_k + j * 4;
      r[j] = (n < p.N && k < p.K) ? *reinterpret_cast<const float4*>(p.w + (int64_t)n * p.K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto stage_w = [&](int buf, const float4 (&r)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      *reinterpret_cast<float4*>(&sW[buf][w_row][w_k + j * 4]) = r[j];
    }
  };
  float4 rw[4];
  load_w(0, rw);
  stage_w(0, rw);
  __syncthreads();
  v4f a[NK4 / 4];  // this wave's 16 tokens: k = 16 j + 4 fk + (0..3) of every 16-k block j
#pragma unroll
  for (int j = 0; j < NK4 / 4; ++j) a[j] = *reinterpret_cast<const v4f*>(&sA[wave * 16 + fr][j * 16 + fk * 4]);
  const int chunks = (p.N + BN - 1) / BN;
  for (int c = 0; c < chunks; ++c) {
    const int buf = c & 1;
    if (c + 1 < chunks) load_w(c + 1, rw);
    v4f acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (v4f){0.f, 0.f, 0.f, 0.f};
    // operands swapped (weight rows as the MFMA's A, tokens as its B): a lane then holds FOUR CONSECUTIVE output channels
    // 4 * fk .. + 3 of token fr in each 16 x 16 tile -- one 16-byte store per tile instead of four scalar ones
    v4f b[NK4 / 4][4];
#pragma unroll
    for (int j = 0; j < NK4 / 4; ++j) {
#pragma unroll
      for (int t = 0; t < 4; ++t) b[j][t] = *reinterpret_cast<const v4f*>(&sW[buf][t * 16 + fr][j * 16 + fk * 4]);
    }
#pragma unroll
    for (int j = 0; j < NK4 / 4; ++j) {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][t][kk], a[j][kk], acc[t], 0, 0, 0);
      }
    }
    // The 16 x 64 tile of this wave is turned through LDS -- its own 16 rows of sA, free since the token fragments went to
    // registers -- so that a store instruction writes 4 rows x 256 contiguous bytes.  (Stored straight from the accumulators,
    // 16 rows x 64 B per instruction, the kernel ran at 1.2 TB/s of writes: 166 us against 66 us without the stores.)
    float* turn = &sA[wave * 16][0];
#pragma unroll
    for (int t = 0; t < 4; ++t) *reinterpret_cast<v4f*>(turn + fr * LDP + t * 16 + fk * 4) = acc[t];
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): LDS operations of one wave complete in order, no barrier needed
    const int tc = (lane & 15) * 4;      // this lane's four columns of the chunk
    const int col = c * BN + tc;
    if (col < p.N) {
      const float4 b = p.bias ? *reinterpret_cast<const float4*>(p.bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int tr = i * 4 + (lane >> 4);
        const int64_t row = m0 + wave * 16 + tr;
        const float4 v = *reinterpret_cast<const float4*>(turn + tr * LDP + tc);
        if (row < p.out_rows)
          *reinterpret_cast<float4*>(p.out + row * p.ldo + col) =
              row < p.M ? make_float4(v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
    if (c + 1 < chunks) stage_w(buf ^ 1, rw);
    __syncthreads();
  }
}

// Linear on a handful of rows (the time MLPs: one timestep): a wave per output channel, the weight row read once in 16-byte
// pieces, all rows' dot products from it, wave reduction.  Bound by the weight read (time_projection of the 14B model: 629 MB).
template <int MR>
__global__ __launch_bounds__(256) void gemv_f32_kernel(const LinParams p) {
  const int lane = threadIdx.x & 63;
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (n >= p.N) return;
  const float* wr = p.w + (int64_t)n * p.K;
  float acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = 0.f;
  for (int k = lane * 4; k < p.K; k += 256) {
    const float4 w = *reinterpret_cast<const float4*>(wr + k);
#pragma unroll
    for (int m = 0; m < MR; ++m) {
      if (m < p.M) {
        float4 x = m < p.x_rows ? *reinterpret_cast<const float4*>(p.x + (int64_t)m * p.ldx + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.in_act != ACT_NONE) { x.x = act(x.x, p.in_act); x.y = act(x.y, p.in_act); x.z = act(x.z, p.in_act); x.w = act(x.w, p.in_act); }
        acc[m] = fmaf(w.x, x.x, fmaf(w.y, x.y, fmaf(w.z, x.z, fmaf(w.w, x.w, acc[m]))));
      }
    }
  }
  const float b = p.bias ? p.bias[n] : 0.f;
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    if (m < p.M) {
      const float v = wave_sum(acc[m]);
      if (lane == 0) p.out[(int64_t)m * p.ldo + n] = act(v + b, p.out_act);
    }
  }
}

template <int AMODE, int CMODE>
int launch(const LinParams& p, hipStream_t st, const char* what) {
  const int64_t rows = p.out_rows > p.M ? p.out_rows : p.M;
  const int64_t mt = (rows + BM - 1) / BM;
  const int nt = (p.N + BN - 1) / BN;
  if (mt == 0) return WANQ_OK;
  hipLaunchKernelGGL((linear_f32_kernel<AMODE, CMODE>), dim3((unsigned)mt, (unsigned)nt), dim3(256), 0, st, p);
  return check_launch(what);
}

inline bool al16(const void* q) { return ((uintptr_t)q & 15) == 0; }

}  // namespace
}  // namespace wanq

using namespace wanq;

extern "C" int wanq_linear_f32(const float* x, int64_t x_rows, const float* w, const float* bias, float* out, int64_t M, int N, int K,
                               int in_act, int out_act, void* stream) {
  WANQ_REQUIRE(x && w && out, WANQ_E_ARG, "wanq_linear_f32: NULL pointer");
  WANQ_REQUIRE(x_rows >= 0 && x_rows <= M, WANQ_E_SHAPE, "wanq_linear_f32: x_rows=%lld must lie in [0, M=%lld]", (long long)x_rows, (long long)M);
  WANQ_REQUIRE(M >= 0 && M < (1ll << 37) && N > 0 && K > 0 && K % BK == 0, WANQ_E_SHAPE,
               "wanq_linear_f32: M=%lld N=%d K=%d (K must be a multiple of %d)", (long long)M, N, K, BK);
  WANQ_REQUIRE(al16(x) && al16(w), WANQ_E_ARG, "wanq_linear_f32: x and w must be 16-byte aligned");
  WANQ_REQUIRE(in_act >= 0 && in_act <= 2 && out_act >= 0 && out_act <= 2, WANQ_E_ARG,
               "wanq_linear_f32: activations are 0 (none), 1 (gelu-tanh), 2 (silu); got %d, %d", in_act, out_act);
  LinParams p{};
  p.x = x; p.w = w; p.bias = bias; p.out = out;
  p.x_rows = x_rows; p.M = M; p.out_rows = M; p.N = N; p.K = K; p.ldx = K; p.ldo = N;
  p.in_act = in_act; p.out_act = out_act;
  if (M >= 1 && M <= 4) {
    hipLaunchKernelGGL(gemv_f32_kernel<4>, dim3((unsigned)((N + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p);
    return check_launch("wanq_linear_f32");
  }
  return launch<A_PLAIN, C_PLAIN>(p, (hipStream_t)stream, "wanq_linear_f32");
}

extern "C" int wanq_time_sinusoid(const void* t, int t_kind, float* out, int n, int dim, void* stream) {
  WANQ_REQUIRE(t && out, WANQ_E_ARG, "wanq_time_sinusoid: NULL pointer");
  WANQ_REQUIRE(t_kind >= 0 && t_kind <= 3, WANQ_E_ARG, "wanq_time_sinusoid: t_kind=%d (0 f32, 1 i64, 2 f64, 3 i32)", t_kind);
  WANQ_REQUIRE(n >= 0 && dim > 0 && dim % 2 == 0, WANQ_E_SHAPE, "wanq_time_sinusoid: n=%d dim=%d (dim must be even)", n, dim);
  if (n == 0) return WANQ_OK;
  const int total = n * (dim / 2);
  hipLaunchKernelGGL(sinusoid_kernel, dim3((total + 127) / 128), dim3(128), 0, (hipStream_t)stream, t, t_kind, out, n, dim);
  return check_launch("wanq_time_sinusoid");
}

extern "C" int wanq_patch_embed(const float* latent, const float* w, const float* bias, float* out, int C, int F, int H, int W, int pt,
                                int ph, int pw, int N, int64_t out_rows, void* stream) {
  WANQ_REQUIRE(latent && w && out, WANQ_E_ARG, "wanq_patch_embed: NULL pointer");
  WANQ_REQUIRE(C > 0 && F > 0 && H > 0 && W > 0 && pt > 0 && ph > 0 && pw > 0 && N > 0, WANQ_E_SHAPE, "wanq_patch_embed: empty dimension");
  WANQ_REQUIRE(F % pt == 0 && H % ph == 0 && W % pw == 0, WANQ_E_SHAPE,
               "wanq_patch_embed: latent %dx%dx%d is not a whole number of %dx%dx%d patches", F, H, W, pt, ph, pw);
  const int64_t K = (int64_t)C * pt * ph * pw;
  WANQ_REQUIRE(K % BK == 0 && K < (1 << 24), WANQ_E_SHAPE, "wanq_patch_embed: C * pt * ph * pw = %lld must be a multiple of %d", (long long)K, BK);
  const int64_t L = (int64_t)(F / pt) * (H / ph) * (W / pw);
  WANQ_REQUIRE(out_rows >= L, WANQ_E_SHAPE, "wanq_patch_embed: out_rows=%lld < %lld tokens", (long long)out_rows, (long long)L);
  WANQ_REQUIRE(N % 4 == 0, WANQ_E_SHAPE, "wanq_patch_embed: N=%d must be a multiple of 4", N);
  WANQ_REQUIRE(al16(w) && al16(out) && al16(bias), WANQ_E_ARG, "wanq_patch_embed: w, bias and out must be 16-byte aligned");
  LinParams p{};
  p.x = latent; p.w = w; p.bias = bias; p.out = out;
  p.x_rows = L; p.M = L; p.out_rows = out_rows; p.N = N; p.K = (int)K; p.ldo = N;
  p.C = C; p.F = F; p.H = H; p.W = W; p.pt = pt; p.ph = ph; p.pw = pw; p.gh = H / ph; p.gw = W / pw;
  if (K <= 64) {
    const dim3 grid((unsigned)((out_rows + BM - 1) / BM));
    hipStream_t st = (hipStream_t)stream;
    switch (K / 16) {
      case 1: hipLaunchKernelGGL(patch_embed_kernel<4>, grid, dim3(256), 0, st, p); break;
      case 2: hipLaunchKernelGGL(patch_embed_kernel<8>, grid, dim3(256), 0, st, p); break;
      case 3: hipLaunchKernelGGL(patch_embed_kernel<12>, grid, dim3(256), 0, st, p); break;
      default: hipLaunchKernelGGL(patch_embed_kernel<16>, grid, dim3(256), 0, st, p); break;
    }
    return check_launch("wanq_patch_embed");
  }
  return launch<A_PATCH, C_PLAIN>(p, (hipStream_t)stream, "wanq_patch_embed");
}

extern "C" int wanq_head_fwd(const float* x, const float* modulation, const float* e, const float* w, const float* bias, float* out,
                             int64_t rows, int K, int N, float eps, int unpatchify, int C, int F, int H, int W, int pt, int ph, int pw,
                             void* stream) {
  WANQ_REQUIRE(x && modulation && e && w && out, WANQ_E_ARG, "wanq_head_fwd: NULL pointer");
  WANQ_REQUIRE(rows >= 0 && K > 0 && N > 0 && K % BK == 0, WANQ_E_SHAPE, "wanq_head_fwd: rows=%lld K=%d N=%d (K must be a multiple of %d)",
               (long long)rows, K, N, BK);
  WANQ_REQUIRE(al16(x) && al16(modulation) && al16(e) && al16(w), WANQ_E_ARG, "wanq_head_fwd: x, modulation, e and w must be 16-byte aligned");
  LinParams p{};
  p.x = x; p.w = w; p.bias = bias; p.out = out; p.mod = modulation; p.e = e; p.eps = eps;
  p.x_rows = rows; p.M = rows; p.out_rows = rows; p.N = N; p.K = K; p.ldx = K; p.ldo = N;
  if (!unpatchify) return launch<A_LNMOD, C_PLAIN>(p, (hipStream_t)stream, "wanq_head_fwd");
  WANQ_REQUIRE(C > 0 && F > 0 && H > 0 && W > 0 && pt > 0 && ph > 0 && pw > 0 && F % pt == 0 && H % ph == 0 && W % pw == 0, WANQ_E_SHAPE,
               "wanq_head_fwd: latent %dx%dx%dx%d / patch %dx%dx%d", C, F, H, W, pt, ph, pw);
  WANQ_REQUIRE(N == C * pt * ph * pw && N <= BN, WANQ_E_SHAPE, "wanq_head_fwd: N=%d must be C * pt * ph * pw = %d and at most %d", N,
               C * pt * ph * pw, BN);
  WANQ_REQUIRE(rows == (int64_t)(F / pt) * (H / ph) * (W / pw), WANQ_E_SHAPE, "wanq_head_fwd: rows=%lld is not the token count of the grid",
               (long long)rows);
  p.C = C; p.F = F; p.H = H; p.W = W; p.pt = pt; p.ph = ph; p.pw = pw; p.gh = H / ph; p.gw = W / pw;
  return launch<A_LNMOD, C_UNPATCH>(p, (hipStream_t)stream, "wanq_head_fwd");
}
