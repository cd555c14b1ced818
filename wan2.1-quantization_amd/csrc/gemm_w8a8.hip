// W8A8 GEMM on gfx950 int8 MFMA (v_mfma_i32_32x32x32_i8) with the dequant epilogue fused.
//
//   acc[m,n] = sum_k A[m,k] * W[n,k]          A: int8 [M,K] activations, W: int8 [N,K] weights
//
// Operand roles: the MFMA "A" operand carries W rows (n), the "B" operand carries activation rows (m),
// so the 32x32 accumulator has its token m on the lane (lane & 31) and 16 output channels n in
// registers, 4 consecutive n per register quad: the epilogue's per-token scalars (sA, sumA) are one
// load per lane and a lane stores 4 consecutive channels at once.
//
// Tiling (v1 structure): 128(M) x 128(N) output tile per 256-thread workgroup, 4 waves as 2(M) x 2(N),
// each wave 64x64 = 2x2 MFMA tiles; K step 128 bytes (whole 128-B lines from HBM/L2).  Global->register
// prefetch of tile k+1 is issued before the MFMAs of tile k and written to the other LDS stage after
// them (one barrier per K tile).  LDS rows are 128 B with the 16-B chunk index XORed by (row>>1)&7:
// ds_read_b128 fragment reads and ds_write_b128 staging writes are both bank-conflict free.
// Workgroup ids are remapped so that each XCD's L2 sees a compact (8 m-tiles x n) panel.
#include "wanq_common.h"

namespace wanq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

struct GemmParams {
  const int8_t* a;
  const int8_t* w;
  void* out;
  const void* sa;
  const void* asum;
  const void* sw;
  const void* bias;
  const void* zp;
  const float* gate;
  const void* residual;
  int tok_dtype, ch_dtype, zp_dtype, epi;
  int M, N, K;
  int mt, nt;
};

constexpr int BM = 128, BN = 128, BK = 128;
constexpr int STAGE_BYTES = (BM + BN) * BK;  // 32 KiB
constexpr int GROUP_M = 8;

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ void load4_ch(const void* p, int dt, int idx, float (&o)[4]) {
  if (dt == WANQ_F32) {
    const float4 v = *reinterpret_cast<const float4*>(static_cast<const float*>(p) + idx);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  } else if (dt == WANQ_F16) {
    const uint2 v = *reinterpret_cast<const uint2*>(static_cast<const __half*>(p) + idx);
    const __half2* h = reinterpret_cast<const __half2*>(&v);
    const float2 a = __half22float2(h[0]), b = __half22float2(h[1]);
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
  } else {  // WANQ_I16
    const short4 v = *reinterpret_cast<const short4*>(static_cast<const short*>(p) + idx);
    o[0] = (float)v.x; o[1] = (float)v.y; o[2] = (float)v.z; o[3] = (float)v.w;
  }
}

template <int OUT>
struct OutIo;
template <>
struct OutIo<WANQ_F16> {
  __device__ static void load4(const void* p, int64_t i, float (&o)[4]) {
    const uint2 v = *reinterpret_cast<const uint2*>(static_cast<const __half*>(p) + i);
    const __half2* h = reinterpret_cast<const __half2*>(&v);
    const float2 a = __half22float2(h[0]), b = __half22float2(h[1]);
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y;
  }
  __device__ static void store4(void* p, int64_t i, const float (&y)[4]) {
    uint2 v;
    __half2* h = reinterpret_cast<__half2*>(&v);
    h[0] = __floats2half2_rn(y[0], y[1]);
    h[1] = __floats2half2_rn(y[2], y[3]);
    *reinterpret_cast<uint2*>(static_cast<__half*>(p) + i) = v;
  }
};
template <>
struct OutIo<WANQ_BF16> {
  __device__ static void load4(const void* p, int64_t i, float (&o)[4]) {
    const uint2 v = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(p) + i);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xffff0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xffff0000u);
  }
  __device__ static void store4(void* p, int64_t i, const float (&y)[4]) {
    uint16_t b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const __hip_bfloat16 t = __float2bfloat16(y[j]);
      b[j] = *reinterpret_cast<const uint16_t*>(&t);
    }
    *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p) + i) =
        make_uint2((uint32_t)b[0] | ((uint32_t)b[1] << 16), (uint32_t)b[2] | ((uint32_t)b[3] << 16));
  }
};
template <>
struct OutIo<WANQ_F32> {
  __device__ static void load4(const void* p, int64_t i, float (&o)[4]) {
    const float4 v = *reinterpret_cast<const float4*>(static_cast<const float*>(p) + i);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
  }
  __device__ static void store4(void* p, int64_t i, const float (&y)[4]) {
    *reinterpret_cast<float4*>(static_cast<float*>(p) + i) = make_float4(y[0], y[1], y[2], y[3]);
  }
};

template <int OUT>
__global__ __launch_bounds__(256, 2) void gemm_w8a8_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;
  const int fr = lane & 31, fh = lane >> 5;

  // ---- workgroup -> tile: XCD-contiguous ids (bijective remap), then groups of GROUP_M m-tiles
  const int nwg = gridDim.x, bid = blockIdx.x;
  const int xq = nwg >> 3, xr = nwg & 7, xcd = bid & 7;
  const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  const int per_group = GROUP_M * p.nt;
  const int group = wg / per_group;
  const int first_m = group * GROUP_M;
  const int gsz = (p.mt - first_m < GROUP_M) ? (p.mt - first_m) : GROUP_M;
  const int in_g = wg - group * per_group;
  const int m0 = (first_m + in_g % gsz) * BM;
  const int n0 = (in_g / gsz) * BN;

  const int K = p.K;
  const int nk = (K + BK - 1) / BK;

  // ---- staging assignment: 4 x 16-B chunks of the activation tile and 4 of the weight tile per thread
  const int8_t* ga[4];
  const int8_t* gw[4];
  int soff[4], kc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int id = tid + 256 * i;
    const int row = id >> 3, c = id & 7;
    const int gm = (m0 + row < p.M) ? (m0 + row) : (p.M - 1);
    const int gn = (n0 + row < p.N) ? (n0 + row) : (p.N - 1);
    ga[i] = p.a + (int64_t)gm * K + c * 16;
    gw[i] = p.w + (int64_t)gn * K + c * 16;
    soff[i] = lds_off(row, c);
    kc[i] = c * 16;
  }
  uint4 ra[4], rw[4];
  auto gload = [&](int kt) {
    const int kb = kt * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (kb + kc[i] < K) {
        ra[i] = *reinterpret_cast<const uint4*>(ga[i] + kb);
        rw[i] = *reinterpret_cast<const uint4*>(gw[i] + kb);
      } else {
        ra[i] = make_uint4(0, 0, 0, 0);
        rw[i] = make_uint4(0, 0, 0, 0);
      }
    }
  };
  auto lstore = [&](int stage) {
    char* sx = smem + stage * STAGE_BYTES;
    char* sw = sx + BM * BK;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      *reinterpret_cast<uint4*>(sx + soff[i]) = ra[i];
      *reinterpret_cast<uint4*>(sw + soff[i]) = rw[i];
    }
  };

  v16i acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

  gload(0);
  lstore(0);
  __syncthreads();

  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) gload(kt + 1);
    const char* sx = smem + cur * STAGE_BYTES;
    const char* sw = sx + BM * BK;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int chunk = 2 * ks + fh;
      v4i wf[2], xf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = *reinterpret_cast<const v4i*>(sw + lds_off(wn * 64 + i * 32 + fr, chunk));
#pragma unroll
      for (int j = 0; j < 2; ++j) xf[j] = *reinterpret_cast<const v4i*>(sx + lds_off(wm * 64 + j * 32 + fr, chunk));
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wf[i], xf[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) lstore(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: acc[i][j][4g+e] is (n = n0 + wn*64 + i*32 + 8g + 4*fh + e, m = m0 + wm*64 + j*32 + fr)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + wm * 64 + j * 32 + fr;
    if (m >= p.M) continue;
    float sa_m = 1.f, asum_m = 0.f;
    if (OUT != WANQ_I32) {
      sa_m = vec_load(p.sa, p.tok_dtype, m);
      if (p.zp) asum_m = vec_load(p.asum, p.tok_dtype, m);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int n = n0 + wn * 64 + i * 32 + 8 * g + 4 * fh;
        if (n >= p.N) continue;
        const int64_t o = (int64_t)m * p.N + n;
        if (OUT == WANQ_I32) {
          *reinterpret_cast<int4*>(static_cast<int*>(p.out) + o) =
              make_int4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        } else {
          float y[4], sw4[4];
          load4_ch(p.sw, p.ch_dtype, n, sw4);
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = (float)acc[i][j][4 * g + e] * sa_m * sw4[e];
          if (p.zp) {
            float z4[4];
            load4_ch(p.zp, p.zp_dtype, n, z4);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] += asum_m * z4[e] * sw4[e];
          }
          if (p.bias) {
            float b4[4];
            load4_ch(p.bias, p.ch_dtype, n, b4);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] += b4[e];
          }
          if (p.epi & WANQ_EPI_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = gelu_tanh_f32(y[e]);
          }
          if (p.epi & WANQ_EPI_GATE_RES) {
            float g4[4], r4[4];
            load4_ch(p.gate, WANQ_F32, n, g4);
            OutIo<OUT == WANQ_I32 ? WANQ_F32 : OUT>::load4(p.residual, o, r4);
#pragma unroll
            for (int e = 0; e < 4; ++e) y[e] = r4[e] + y[e] * g4[e];
          }
          OutIo<OUT == WANQ_I32 ? WANQ_F32 : OUT>::store4(p.out, o, y);
        }
      }
    }
  }
}

template <int OUT>
static int launch_gemm(const GemmParams& p, hipStream_t st) {
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w8a8_kernel<OUT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        2 * STAGE_BYTES);
    attr_set = true;
  }
  hipLaunchKernelGGL(gemm_w8a8_kernel<OUT>, dim3((unsigned)(p.mt * p.nt)), dim3(256), 2 * STAGE_BYTES, st, p);
  return check_launch("wanq_gemm_w8a8");
}

}  // namespace wanq

using namespace wanq;

extern "C" int wanq_gemm_w8a8(const int8_t* a, const int8_t* w, void* out, int out_dtype, const void* sa,
                              const void* asum, int tok_dtype, const void* sw, const void* bias, int ch_dtype,
                              const void* zp, int zp_dtype, const float* gate, const void* residual, int epi_flags,
                              int64_t M, int N, int K, void* stream) {
  WANQ_REQUIRE(a && w && out, WANQ_E_ARG, "wanq_gemm_w8a8: a, w and out must be non-NULL");
  WANQ_REQUIRE(out_dtype == WANQ_F16 || out_dtype == WANQ_BF16 || out_dtype == WANQ_F32 || out_dtype == WANQ_I32,
               WANQ_E_ARG, "wanq_gemm_w8a8: bad out dtype %d", out_dtype);
  WANQ_REQUIRE(M >= 0 && M < (1ll << 31) / 2, WANQ_E_SHAPE, "wanq_gemm_w8a8: M=%lld out of range", (long long)M);
  WANQ_REQUIRE(N >= 8 && N % 8 == 0, WANQ_E_SHAPE, "wanq_gemm_w8a8: N=%d must be a positive multiple of 8", N);
  WANQ_REQUIRE(K >= 16 && K % 16 == 0, WANQ_E_SHAPE, "wanq_gemm_w8a8: K=%d must be a positive multiple of 16", K);
  if (out_dtype != WANQ_I32) {
    WANQ_REQUIRE(sa && sw, WANQ_E_ARG, "wanq_gemm_w8a8: sa and sw are required for a floating output");
    WANQ_REQUIRE(is_vec(tok_dtype) && is_vec(ch_dtype), WANQ_E_ARG, "wanq_gemm_w8a8: tok/ch dtype must be F16 or F32");
    WANQ_REQUIRE(!zp || (asum && (zp_dtype == WANQ_I16 || zp_dtype == WANQ_F32)), WANQ_E_ARG,
                 "wanq_gemm_w8a8: zp needs asum and dtype I16 or F32");
    WANQ_REQUIRE(!(epi_flags & WANQ_EPI_GATE_RES) || (gate && residual), WANQ_E_ARG,
                 "wanq_gemm_w8a8: WANQ_EPI_GATE_RES needs gate and residual");
  } else {
    WANQ_REQUIRE(epi_flags == 0, WANQ_E_ARG, "wanq_gemm_w8a8: int32 output takes no epilogue flags");
  }
  WANQ_REQUIRE((epi_flags & ~(WANQ_EPI_GELU | WANQ_EPI_GATE_RES)) == 0, WANQ_E_ARG, "wanq_gemm_w8a8: unknown epilogue flag");
  if (M == 0) return WANQ_OK;
  GemmParams p{};
  p.a = a; p.w = w; p.out = out; p.sa = sa; p.asum = asum; p.sw = sw; p.bias = bias; p.zp = zp; p.gate = gate;
  p.residual = residual; p.tok_dtype = tok_dtype; p.ch_dtype = ch_dtype; p.zp_dtype = zp_dtype; p.epi = epi_flags;
  p.M = (int)M; p.N = N; p.K = K;
  p.mt = (int)((M + BM - 1) / BM);
  p.nt = (N + BN - 1) / BN;
  WANQ_REQUIRE((int64_t)p.mt * p.nt < (1ll << 31), WANQ_E_SHAPE, "wanq_gemm_w8a8: too many tiles");
  hipStream_t st = (hipStream_t)stream;
  switch (out_dtype) {
    case WANQ_F16: return launch_gemm<WANQ_F16>(p, st);
    case WANQ_BF16: return launch_gemm<WANQ_BF16>(p, st);
    case WANQ_F32: return launch_gemm<WANQ_F32>(p, st);
    default: return launch_gemm<WANQ_I32>(p, st);
  }
}
