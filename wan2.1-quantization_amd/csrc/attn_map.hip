// Attention with a QUANTISED ATTENTION MAP, streaming (no N x N map in memory).
//
// The reference quantises the materialised post-softmax map (QuantizedAttentionMapOpenSORA, Q/base/quant_attn.py:118-173,
// wired at W/models/quant_opensora.py:459-476: attn = softmax(q k^T * scale); attn = attn_map_quantizer(attn); x = attn @ v)
// and asserts against flash attention.  Its group 'row' permutes the map so that every KEY COLUMN is one quantisation group
// (all queries share its parameters) and applies the DynamicQuantizer (Q/base/base_quantizer.py:101-162).  For a map in
// [0, 1] both of that quantiser's forms reduce to
//     P~[q,k] = rne(P[q,k] / delta_k) * delta_k,   delta_k = max_q P[q,k] / L,   L = 2^(n-1) - 1 (sym) or 2^n - 1 (asym)
// (asym: x_min is clipped to <= 0, i.e. 0, so delta = x_max / (n_levels - 1) and the zero point cancels in the dequantised value;
// eps floors 1e-6 / 1e-8 as there).  That needs the column maxima of the NORMALISED map, hence three streaming passes over the
// keys, each recomputing S = K . Q^T on the matrix cores:
//   pass 0  row statistics (m_q, l_q)                       -> workspace [H][Lq] x 2
//   pass 1  column maxima  c_k = max_q exp2(s - m_q) / l_q   -> workspace [H][Lk]   (atomic max on the bits of a float >= 0)
//   pass 2  O = sum_k P~[q,k] V[k]                           (P~ rounded to bf16 for the P.V MFMA, fp32 accumulation)
// A study feature (2.5x the attention work, and the reference cannot run it at all beyond toy lengths): written for
// clarity, not tuned -- one wave per 32 queries, v_mfma_f32_16x16x32_bf16 with operands straight from global memory.
#include "wanq_common.h"

namespace wanq {

typedef __bf16 mbf16x8 __attribute__((ext_vector_type(8)));
typedef float mf32x4 __attribute__((ext_vector_type(4)));

struct AttnMapParams {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  uint16_t* o;
  int64_t q_stride, k_stride, v_stride, o_stride;  // elements between consecutive tokens
  int Lq, Lk, H;
  float c;       // softmax scale * log2(e)
  float levels;  // 2^(n-1) - 1 (sym) or 2^n - 1 (asym)
  float eps;     // 1e-6 (sym) / 1e-8 (asym)
  float* m;      // [H][Lq] row maxima (log2 domain)
  float* l;      // [H][Lq] row sums
  float* cmax;   // [H][Lk] column maxima of the normalised map
};

// One wave = 32 queries (two blocks nq of 16): a K fragment (and, in pass 2, a gathered V^T fragment) feeds the MFMAs of both.
template <int PASS>
__global__ __launch_bounds__(256) void attn_map_kernel(const AttnMapParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n16 = lane & 15, g4 = lane >> 4;
  const int head = blockIdx.y;
  const int q0 = (blockIdx.x * 4 + wave) * 32;
  if (q0 >= p.Lq) return;  // whole wave (no barriers in this kernel)
  int qr[2];
  bool q_ok[2];
  mbf16x8 qf[2][4];  // pre-scaled by softmax scale * log2(e): query qr[nq], d = 32 s + 8 g4 + [0, 8)
#pragma unroll
  for (int nq = 0; nq < 2; ++nq) {
    qr[nq] = q0 + 16 * nq + n16;
    q_ok[nq] = qr[nq] < p.Lq;
    qr[nq] = q_ok[nq] ? qr[nq] : p.Lq - 1;
    const uint16_t* qp = p.q + (int64_t)qr[nq] * p.q_stride + head * 128 + 8 * g4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qf[nq][s] = *reinterpret_cast<const mbf16x8*>(qp + 32 * s);
#pragma unroll
      for (int e = 0; e < 8; ++e) qf[nq][s][e] = (__bf16)((float)qf[nq][s][e] * p.c);
    }
  }
  const uint16_t* kbase = p.k + head * 128;
  const int64_t stat[2] = {(int64_t)head * p.Lq + qr[0], (int64_t)head * p.Lq + qr[1]};

  // S^T blocks of keys kb .. kb+15 against both query blocks: A = K rows (lane (r, g): row kb + r, d = 32 s + 8 g + [0, 8)),
  // B = Q fragments; accumulator element e of lane (n, g): key kb + 4 g + e, query n.  Keys >= Lk come out as -inf.
  auto s_blocks = [&](int kb, mf32x4 (&s)[2]) {
    int kr = kb + n16;
    kr = kr < p.Lk ? kr : p.Lk - 1;
    const uint16_t* kp = kbase + (int64_t)kr * p.k_stride + 8 * g4;
    s[0] = mf32x4{0.f, 0.f, 0.f, 0.f};
    s[1] = mf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const mbf16x8 kf = *reinterpret_cast<const mbf16x8*>(kp + 32 * sl);
      s[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][sl], s[0], 0, 0, 0);
      s[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][sl], s[1], 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (kb + 4 * g4 + e >= p.Lk) { s[0][e] = -INFINITY; s[1][e] = -INFINITY; }
  };

  if (PASS == 0) {
    // ---- row statistics.  The four lanes of a query (n, n+16, n+32, n+48) see different keys: the maximum is shared every
    // block, the partial sums are added at the end.
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    for (int kb = 0; kb < p.Lk; kb += 16) {
      mf32x4 s[2];
      s_blocks(kb, s);
#pragma unroll
      for (int nq = 0; nq < 2; ++nq) {
        float mx = fmaxf(fmaxf(s[nq][0], s[nq][1]), fmaxf(s[nq][2], s[nq][3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float mn = fmaxf(m[nq], mx);  // finite from the first block on (its key 0 is never masked)
        l[nq] = l[nq] * __builtin_amdgcn_exp2f(m[nq] - mn) + __builtin_amdgcn_exp2f(s[nq][0] - mn) + __builtin_amdgcn_exp2f(s[nq][1] - mn) +
                __builtin_amdgcn_exp2f(s[nq][2] - mn) + __builtin_amdgcn_exp2f(s[nq][3] - mn);
        m[nq] = mn;
      }
    }
#pragma unroll
    for (int nq = 0; nq < 2; ++nq) {
      float lt = l[nq];
      lt += __shfl_xor(lt, 16, 64);
      lt += __shfl_xor(lt, 32, 64);
      if (q_ok[nq] && g4 == 0) {
        p.m[stat[nq]] = m[nq];
        p.l[stat[nq]] = lt;
      }
    }
    return;
  }

  const float m[2] = {p.m[stat[0]], p.m[stat[1]]}, inv_l[2] = {1.0f / p.l[stat[0]], 1.0f / p.l[stat[1]]};

  if (PASS == 1) {
    // ---- column maxima of the normalised map: max over the wave's 32 queries (two blocks, then the 16 lanes of a lane group),
    // then one atomic per key and wave.  Duplicated (clamped) queries of a ragged last block repeat a real query: harmless.
    unsigned int* cm = reinterpret_cast<unsigned int*>(p.cmax) + (int64_t)head * p.Lk;
    for (int kb = 0; kb < p.Lk; kb += 16) {
      mf32x4 s[2];
      s_blocks(kb, s);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pv = fmaxf(__builtin_amdgcn_exp2f(s[0][e] - m[0]) * inv_l[0], __builtin_amdgcn_exp2f(s[1][e] - m[1]) * inv_l[1]);  // masked: 0
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) pv = fmaxf(pv, __shfl_xor(pv, o, 64));
        const int key = kb + 4 * g4 + e;
        if (n16 == 0 && key < p.Lk) atomicMax(cm + key, __float_as_uint(pv));  // pv >= 0: the bit patterns order like the values
      }
    }
    return;
  }

  // ---- PASS 2: O^T += V^T . P~^T over key slices of 32 (two S blocks).  k index 8 g + j of the P.V MFMA <-> key
  // kb + 16 (j >> 2) + 4 g + (j & 3); the A operand V^T[d][that key] is gathered with 2-byte loads (d = 16 db + n16).
  mf32x4 o[8][2];
#pragma unroll
  for (int db = 0; db < 8; ++db) {
    o[db][0] = mf32x4{0.f, 0.f, 0.f, 0.f};
    o[db][1] = mf32x4{0.f, 0.f, 0.f, 0.f};
  }
  const float* cm = p.cmax + (int64_t)head * p.Lk;
  const uint16_t* vbase = p.v + head * 128 + n16;
  for (int kb = 0; kb < p.Lk; kb += 32) {
    mbf16x8 pf[2];
    int64_t vrow[8];
#pragma unroll
    for (int jh = 0; jh < 2; ++jh) {
      mf32x4 s[2];
      s_blocks(kb + 16 * jh, s);  // a block wholly past Lk gives -inf -> P~ = 0
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int key = kb + 16 * jh + 4 * g4 + e;
        const int kc = key < p.Lk ? key : p.Lk - 1;
        float delta = cm[kc] / p.levels;
        delta = delta < p.eps ? p.eps : delta;
#pragma unroll
        for (int nq = 0; nq < 2; ++nq) {
          const float pr = __builtin_amdgcn_exp2f(s[nq][e] - m[nq]) * inv_l[nq];
          pf[nq][4 * jh + e] = (__bf16)(__builtin_rintf(pr / delta) * delta);
        }
        vrow[4 * jh + e] = (int64_t)kc * p.v_stride;
      }
    }
#pragma unroll
    for (int db = 0; db < 8; ++db) {
      mbf16x8 vf;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint16_t bits = vbase[vrow[j] + 16 * db];
        vf[j] = __builtin_bit_cast(__bf16, bits);
      }
      o[db][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[0], o[db][0], 0, 0, 0);
      o[db][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[1], o[db][1], 0, 0, 0);
    }
  }
#pragma unroll
  for (int nq = 0; nq < 2; ++nq) {
    if (!q_ok[nq]) continue;  // lane (n, g) holds O[query n][d = 16 db + 4 g + e]
    uint16_t* op = p.o + (int64_t)qr[nq] * p.o_stride + head * 128 + 4 * g4;
#pragma unroll
    for (int db = 0; db < 8; ++db) {
      typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
      bf16x4 b;
#pragma unroll
      for (int e = 0; e < 4; ++e) b[e] = (__bf16)o[db][nq][e];
      *reinterpret_cast<bf16x4*>(op + 16 * db) = b;
    }
  }
}

}  // namespace wanq

using namespace wanq;

extern "C" int64_t wanq_attention_map_workspace(int64_t Lq, int64_t Lk, int heads) {
  return (2 * Lq + Lk) * (int64_t)heads * (int64_t)sizeof(float);
}

extern "C" int wanq_attention_map_quant_fwd(const void* q, const void* k, const void* v, void* o, int dtype, int64_t Lq, int64_t Lk, int heads,
                                            int head_dim, int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride, float scale,
                                            int n_bits, int sym, void* workspace, int64_t workspace_bytes, void* stream) {
  const char* what = "wanq_attention_map_quant_fwd";
  WANQ_REQUIRE(q && k && v && o, WANQ_E_ARG, "%s: NULL pointer", what);
  WANQ_REQUIRE(dtype == WANQ_BF16, WANQ_E_ARG, "%s: only bf16 is implemented (dtype code %d)", what, dtype);
  WANQ_REQUIRE(head_dim == 128, WANQ_E_SHAPE, "%s: head_dim=%d, only 128 is implemented", what, head_dim);
  WANQ_REQUIRE(heads >= 1 && heads <= 65535, WANQ_E_SHAPE, "%s: heads=%d out of range", what, heads);
  WANQ_REQUIRE(Lq >= 0 && Lk >= 1 && Lq < (1ll << 30) && Lk < (1ll << 30), WANQ_E_SHAPE, "%s: bad lengths", what);
  WANQ_REQUIRE(n_bits >= 2 && n_bits <= 8, WANQ_E_ARG, "%s: n_bits=%d must be 2..8", what, n_bits);
  const int64_t need = (int64_t)heads * head_dim;
  WANQ_REQUIRE(q_stride >= need && k_stride >= need && v_stride >= need && o_stride >= need, WANQ_E_SHAPE,
               "%s: token stride smaller than heads*head_dim", what);
  WANQ_REQUIRE((q_stride | k_stride) % 8 == 0 && o_stride % 4 == 0, WANQ_E_SHAPE, "%s: q / k strides must be multiples of 8, o of 4 elements", what);
  if (Lq == 0) return WANQ_OK;
  const int64_t need_ws = wanq_attention_map_workspace(Lq, Lk, heads);
  WANQ_REQUIRE(workspace && workspace_bytes >= need_ws, WANQ_E_ARG, "%s: workspace of %lld bytes needed, %lld given", what, (long long)need_ws,
               (long long)workspace_bytes);
  WANQ_REQUIRE(((uintptr_t)workspace & 15) == 0, WANQ_E_ARG, "%s: workspace must be 16-byte aligned", what);
  AttnMapParams p{};
  p.q = (const uint16_t*)q; p.k = (const uint16_t*)k; p.v = (const uint16_t*)v; p.o = (uint16_t*)o;
  p.q_stride = q_stride; p.k_stride = k_stride; p.v_stride = v_stride; p.o_stride = o_stride;
  p.Lq = (int)Lq; p.Lk = (int)Lk; p.H = heads;
  p.c = scale * 1.4426950408889634f;
  p.levels = sym ? (float)((1 << (n_bits - 1)) - 1) : (float)((1 << n_bits) - 1);
  p.eps = sym ? 1e-6f : 1e-8f;
  p.m = static_cast<float*>(workspace);
  p.l = p.m + Lq * heads;
  p.cmax = p.l + Lq * heads;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(p.cmax, 0, (size_t)Lk * heads * sizeof(float), st) != hipSuccess) {
    set_error("%s: hipMemsetAsync failed", what);
    return WANQ_E_LAUNCH;
  }
  dim3 grid((unsigned)((Lq + 127) / 128), (unsigned)heads);  // 4 waves x 32 queries
  hipLaunchKernelGGL(attn_map_kernel<0>, grid, dim3(256), 0, st, p);
  hipLaunchKernelGGL(attn_map_kernel<1>, grid, dim3(256), 0, st, p);
  hipLaunchKernelGGL(attn_map_kernel<2>, grid, dim3(256), 0, st, p);
  return check_launch(what);
}
