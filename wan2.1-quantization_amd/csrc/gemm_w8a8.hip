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
#include "gemm_params.h"
#include <stdlib.h>

namespace wanq {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int OUT>
__device__ __forceinline__ uint2 pack16x4(const float (&y)[4]) {
  uint2 v;
  if (OUT == WANQ_F16) {
    // the fp32 value first, then its cast (the reference's order, w8a8_gemm_cuda.cu:416-442): without the opaque copies hipcc may
    // contract the last fma and the cast into v_fma_mixlo_f16 -- one rounding instead of two, a different half in rare cases, and
    // which of the two a kernel gets depends on the code around it
    float z[4] = {y[0], y[1], y[2], y[3]};
    asm volatile("" : "+v"(z[0]), "+v"(z[1]), "+v"(z[2]), "+v"(z[3]));
    __half2* h = reinterpret_cast<__half2*>(&v);
    h[0] = __floats2half2_rn(z[0], z[1]);
    h[1] = __floats2half2_rn(z[2], z[3]);
  } else {
    uint16_t b[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const __hip_bfloat16 t = __float2bfloat16(y[j]);
      b[j] = *reinterpret_cast<const uint16_t*>(&t);
    }
    v = make_uint2((uint32_t)b[0] | ((uint32_t)b[1] << 16), (uint32_t)b[2] | ((uint32_t)b[3] << 16));
  }
  return v;
}

constexpr int BM = 128, BN = 128, BK = 128;
constexpr int STAGE_BYTES = (BM + BN) * BK;  // 32 KiB
constexpr int GROUP_M = 4;  // m-tiles per L2 panel (sweep 2..32 on cfg-B: 4 best, 8 within 1-3 %; WANQ_GEMM_GROUP_M overrides)
// kernel selection (wanq_gemm_select_kernel; environment at start-up: WANQ_GEMM_V1=1 -> 1, WANQ_GEMM_PP=0 -> 2)
static int g_kernel_sel = [] {
  const char* v1 = getenv("WANQ_GEMM_V1");
  const char* pp = getenv("WANQ_GEMM_PP");
  return (v1 && v1[0] == '1') ? 1 : (pp && pp[0] == '0') ? 2 : 0;
}();

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

// W4: `w` holds 4-bit codes in the packed layout of wanq_pack_w4 (bias 0 nibbles, K/2 bytes per row); the staging pass expands
// a thread's 8 packed bytes to the 16 int8 of its chunk, so LDS and the main loop are those of the W8 form.
template <int OUT, bool W4>
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
    gw[i] = W4 ? p.w + (int64_t)gn * (K / 2) + c * 8 : p.w + (int64_t)gn * K + c * 16;
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
        if (W4) {
          const uint2 t = *reinterpret_cast<const uint2*>(gw[i] + kb / 2);
          rw[i] = make_uint4(t.x & 0x0f0f0f0fu, (t.x >> 4) & 0x0f0f0f0fu, t.y & 0x0f0f0f0fu, (t.y >> 4) & 0x0f0f0f0fu);
        } else {
          rw[i] = *reinterpret_cast<const uint4*>(gw[i] + kb);
        }
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
            for (int e = 0; e < 4; ++e) y[e] = gelu_tanh_fast_f32(y[e]);
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

// =====================================================================================================
// v2 (large M): persistent kernel, 256(M) x 256(N) tile, 8 waves as 2(M) x 4(N), each wave 128 tokens x 64
// channels = 8 x 4 blocks of v_mfma_i32_16x16x64_i8 (12 ds_read_b128 per 32 MFMAs).
//  * MFMA shape: the 16x16x64 form does the same work per cycle as 32x32x32 but the part holds a higher clock under it (the
//    chip is clock/power-limited in this kernel: 1.45-1.85 GHz measured inside it, tools/probes/clock_probe_run.py; bare
//    streams: 4.25 vs 3.74 POP/s, tools/probes/mfma_shape_clock.hip).  Same LDS bytes per MFMA cycle, same LDS image.
//  * Both operands go global -> LDS directly (global_load_lds_dwordx4, 16 B per lane, one 1-KiB wave
//    instruction = 8 rows x 128 B); the LDS image stays lane-linear and the bank swizzle is applied on the
//    per-lane SOURCE address (and on the fragment reads).
//  * Two 64-KiB stages: the loads of K-tile t+1 are issued right after the barrier that publishes tile t and
//    fly under its 64 MFMAs per wave (one barrier per K tile); inside a K tile the fragments of MFMA block b+1 are
//    read while the MFMAs of block b issue (register double buffer).
//  * One workgroup per CU walks its tiles: the first K-tile of the NEXT output tile is requested before the
//    epilogue of the current one, and the wait that publishes it is a COUNTED s_waitcnt vmcnt(#epilogue
//    stores), so the epilogue's stores drain to HBM underneath the next tile's main loop instead of stalling
//    every CU at the same time.
// Used when M >= 512 and K % 128 == 0; everything else takes the v1 kernel.
#ifdef WANQ_CLOCK_PROBE  // diagnostic build only: shader clock held by one workgroup of the persistent kernel
__device__ unsigned long long g_gemm_clk[2];
#endif
constexpr int B2M = 256, B2N = 256, B2K = 128;
constexpr int B2_STAGE = (B2M + B2N) * B2K;  // 64 KiB

// W4 (packed 4-bit weights, wanq_pack_w4 layout): a K-tile of the weight panel is 256 rows x 64 B = 16 KiB instead of 32 (two
// LDS-DMA instructions per wave instead of four: the weight half of the ingest stream halves); a lane reads the 8 packed bytes
// = 16 codes of its operand with ONE ds_read_b64 and expands them in registers (and / shift+and).  LDS image of the packed
// panel: 64-B rows, 16-B pieces XORed with (row>>2)&3 (conflict-free for the b64 lane groups, and a whole 16-B piece moves, so
// the DMA source stays 16 B contiguous).
// Store loop of the fp32 + gate + residual epilogue (the o / cross_attn.o / ffn.2 projections) with the residual lines PREFETCHED
// ONE CHUNK AHEAD BY LDS-DMA.  Read where they are used, the residual lines cost every chunk a full memory latency (the C x C
// GEMM ran 154 us against 91 us without that load; with the prefetch 129 us, K = 8960 447 -> 420 us; tools/ab_gemm_variants.py).
// Chunk c = 2 J + ih = 32 tokens x 32 channels; its four 1-KiB residual pieces go to rbuf + (c & 1) * 4096, lane-linear (the lane
// that fetched a 16-B piece reads it back): no registers, counted vmcnt waits (the four stores of chunk c-1 and the four pieces
// of chunk c+1 may stay in flight).  rbuf lives in stage 0: the CALLER puts a workgroup barrier between this loop and the next
// tile's first LDS-DMA, which lands in the other waves' buffers (tests/test_gpu_gemm.py::test_fp32_gate_residual_in_place_many_tiles
// fails on every shape without it).  Requesting chunks 0 and 1 already behind the main loop's last barrier was measured too:
// +3 % at K = 1536, -2 % at K = 13824, two spilled registers in the W4 form: not kept.
// A function of its own for its __restrict__ parameters: hipcc puts s_waitcnt vmcnt(0) in front of every LDS access that may
// alias an LDS-DMA in flight, i.e. in front of the turn-buffer writes right behind the prefetch (seen in the ISA); with the three
// LDS regions as distinct restrict pointers it knows they do not.
typedef int gemm_v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void b2_store_f32_res(const GemmParams& p, gemm_v4i (&acc)[4][8], const float (&sa_m)[8], const float (&asum_m)[8],
                                                 const float* __restrict__ chan, char* __restrict__ tb, char* __restrict__ rbuf, int cur_n0,
                                                 int tok_base, int wn, int e16, int eq4, int rd_row, int rd_c, int lane_e, bool full_tile) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  // rows past M / channels past N are clamped: fetched, never stored
#define B2_RES_DMA(c)                                                                                          \
  _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                           \
    int tok_ = tok_base + ((c) >> 1) * 32 + rd_row + 8 * ps, n_ = cur_n0 + wn * 64 + ((c) & 1) * 32 + rd_c * 4; \
    tok_ = tok_ < p.M ? tok_ : p.M - 1;                                                                        \
    n_ = n_ + 4 <= p.N ? n_ : p.N - 4;                                                                         \
    __builtin_amdgcn_global_load_lds((glb_void*)(static_cast<const float*>(p.residual) + (int64_t)tok_ * p.N + n_), \
                                     (lds_void*)(rbuf + ((c) & 1) * 4096 + ps * 1024), 16, 0, 0);              \
  }
  B2_RES_DMA(0)
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int J = c >> 1, ih = c & 1;
    if (c + 1 < 8) { B2_RES_DMA(c + 1) }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * J + jj, tr = jj * 16 + e16;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * ih + ii;
        char* dst = tb + tr * 128 + (((4 * ii + eq4) ^ (tr & 7)) << 4);
        const int nl = wn * 64 + i * 16 + 4 * eq4;
        const float4 sw4 = *reinterpret_cast<const float4*>(chan + nl);
        const float4 zs4 = *reinterpret_cast<const float4*>(chan + 256 + nl);
        const float4 b4 = *reinterpret_cast<const float4*>(chan + 512 + nl);
        const float swa[4] = {sw4.x, sw4.y, sw4.z, sw4.w}, zsa[4] = {zs4.x, zs4.y, zs4.z, zs4.w};
        const float ba[4] = {b4.x, b4.y, b4.z, b4.w};
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) y[e] = fmaf((float)acc[i][j][e] * sa_m[j], swa[e], fmaf(asum_m[j], zsa[e], ba[e]));
        if (p.epi & WANQ_EPI_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = gelu_tanh_fast_f32(y[e]);
        }
        *reinterpret_cast<float4*>(dst) = make_float4(y[0], y[1], y[2], y[3]);
      }
    }
    // chunk c's pieces have landed; the stores of chunk c-1 and the pieces of chunk c+1 may still fly.  The counts hold for a
    // full tile only: in a ragged one a wave skips whole store instructions (rows past M), so it drains instead.
    if (!full_tile) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (c == 0 || c == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int row = rd_row + 8 * ps;
      const int tok = tok_base + J * 32 + row;
      const int nl = wn * 64 + ih * 32 + rd_c * 4;
      const int n = cur_n0 + nl;
      if (!(full_tile || (tok < p.M && n < p.N))) continue;
      const float4 v = *reinterpret_cast<const float4*>(tb + row * 128 + ((rd_c ^ (row & 7)) << 4));
      const float4 rv = *reinterpret_cast<const float4*>(rbuf + (c & 1) * 4096 + ps * 1024 + lane_e * 16);
      const float4 g4 = *reinterpret_cast<const float4*>(chan + 768 + nl);
      *reinterpret_cast<float4*>(static_cast<float*>(p.out) + (int64_t)tok * p.N + n) =
          make_float4(fmaf(v.x, g4.x, rv.x), fmaf(v.y, g4.y, rv.y), fmaf(v.z, g4.z, rv.z), fmaf(v.w, g4.w, rv.w));
    }
  }
#undef B2_RES_DMA
}

// 16-bit store loop (no residual) TOGETHER WITH the request of the next tile's first K-tile, in one function with the LDS regions
// as __restrict__ parameters.  In the kernel body hipcc put s_waitcnt vmcnt(0) in front of the first LDS read behind that request
// (the per-channel values; it cannot tell stage 0 from stage 1), so the prefetch that was meant to fly under the store loop was
// waited for on the spot, once per tile (seen in the ISA).  Inside this function the request carries the noalias scopes of
// `chan` and `tb` and no wait is inserted: the stores queue up behind the LDS-DMA and the next tile's first barrier counts them.
template <int OUT, bool W4>
__device__ __forceinline__ void b2_store16_and_request(const GemmParams& p, gemm_v4i (&acc)[4][8], const float (&sa_m)[8], const float (&asum_m)[8],
                                                       const float* __restrict__ chan, char* __restrict__ tb, char* __restrict__ stage0, bool request,
                                                       const uint32_t (&srcx)[4], const uint32_t (&srcw)[4], int wave, int cur_n0, int tok_base, int wn,
                                                       int e16, int eq4, int rd_row, int rd_c, bool full_tile) {
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  if (request) {  // K-tile 0 of the next tile -> stage 0 (the pieces of B2_ISSUE(0, 0))
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      __builtin_amdgcn_global_load_lds((glb_void*)(p.a + srcx[g]), (lds_void*)(stage0 + wave * 1024 + g * 8192), 16, 0, 0);
      if (!W4 || g < 2)
        __builtin_amdgcn_global_load_lds((glb_void*)(p.w + srcw[g]), (lds_void*)(stage0 + B2M * B2K + wave * 1024 + g * 8192), 16, 0, 0);
    }
  }
  // chunk = 32 tokens (token blocks 2J, 2J+1) x 64 channels in the output type.  The lane's 16 channels (4 per channel block i)
  // keep their three per-channel values in registers for the whole loop (48 registers; read per use they were 96 ds_read_b128
  // per wave and tile).
  float swa[4][4], zsa[4][4], ba[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int nl = wn * 64 + i * 16 + 4 * eq4;
    const float4 sw4 = *reinterpret_cast<const float4*>(chan + nl);
    const float4 zs4 = *reinterpret_cast<const float4*>(chan + 256 + nl);
    const float4 b4 = *reinterpret_cast<const float4*>(chan + 512 + nl);
    swa[i][0] = sw4.x; swa[i][1] = sw4.y; swa[i][2] = sw4.z; swa[i][3] = sw4.w;
    zsa[i][0] = zs4.x; zsa[i][1] = zs4.y; zsa[i][2] = zs4.z; zsa[i][3] = zs4.w;
    ba[i][0] = b4.x; ba[i][1] = b4.y; ba[i][2] = b4.z; ba[i][3] = b4.w;
  }
#pragma unroll
  for (int J = 0; J < 4; ++J) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * J + jj, tr = jj * 16 + e16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float y[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)  // acc*sA*sW + (sumA*(zp*sW) + bias)
          y[e] = fmaf((float)acc[i][j][e] * sa_m[j], swa[i][e], fmaf(asum_m[j], zsa[i][e], ba[i][e]));
        if (p.epi & WANQ_EPI_GELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) y[e] = gelu_tanh_fast_f32(y[e]);
        }
        const int cb = (i * 16 + 4 * eq4) * 2;  // byte column inside the 128-B row
        *reinterpret_cast<uint2*>(tb + tr * 128 + ((((cb >> 4) ^ (tr & 7)) << 4) | (cb & 15))) = pack16x4<OUT>(y);
      }
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int row = rd_row + 8 * ps;
      const uint4 v = *reinterpret_cast<const uint4*>(tb + row * 128 + ((rd_c ^ (row & 7)) << 4));
      const int tok = tok_base + J * 32 + row;
      const int n = cur_n0 + wn * 64 + rd_c * 8;
      if (full_tile || (tok < p.M && n < p.N))
        *reinterpret_cast<uint4*>(static_cast<uint16_t*>(p.out) + (int64_t)tok * p.N + n) = v;
    }
  }
}

template <int OUT, bool W4>
__global__ __launch_bounds__(512, 2) void gemm_w8a8_big_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 1, wn = wave >> 1;
#ifdef WANQ_CLOCK_PROBE
  const unsigned long long clk_c0 = clock64(), clk_w0 = wall_clock64();
#endif
  const int fr = lane & 31, fh = lane >> 5;
  const int K = p.K;
  const int nk = K / B2K;
  const int ntiles = p.mt * p.nt;

  // tile id -> (m0, n0): XCD-contiguous ids (bijective remap; gridDim.x % 8 == 0 so a workgroup's tiles all
  // sit in its own XCD's range), then groups of GROUP_M m-tiles
  auto tile_origin = [&](int t, int& m0, int& n0) {
    const int xq = ntiles >> 3, xr = ntiles & 7, xcd = t & 7;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (t >> 3);
    const int per_group = p.group_m * p.nt;
    const int group = wg / per_group;
    const int first_m = group * p.group_m;
    const int gsz = (p.mt - first_m < p.group_m) ? (p.mt - first_m) : p.group_m;
    const int in_g = wg - group * per_group;
    m0 = (first_m + in_g % gsz) * B2M;
    n0 = (in_g / gsz) * B2N;
  };

  // LDS-DMA: instruction g of this wave fills rows 8*(wave+8g) .. +7 of a stage (1 KiB, lane-linear)
  const int drow = wave * 8 + (lane >> 3);  // + 64 g
  uint32_t srcx[4], srcw[4];  // byte offsets from p.a / p.w (M*K and N*K < 4 GiB is checked on the host)
  auto set_sources = [&](int m0, int n0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = drow + 64 * g;
      const int lc = (lane & 7) ^ ((row >> 1) & 7);  // logical chunk that belongs at this physical slot
      const int gm = (m0 + row < p.M) ? (m0 + row) : (p.M - 1);
      srcx[g] = (uint32_t)gm * (uint32_t)K + lc * 16;
      if (!W4) {
        const int gn = (n0 + row < p.N) ? (n0 + row) : (p.N - 1);
        srcw[g] = (uint32_t)gn * (uint32_t)K + lc * 16;
      } else if (g < 2) {  // instruction g fills packed rows 16*(wave + 8g) .. +15: 4 lanes (16-B pieces) per 64-B row
        const int rw_ = 16 * (wave + 8 * g) + (lane >> 2);
        const int lp = (lane & 3) ^ ((rw_ >> 2) & 3);
        const int gn = (n0 + rw_ < p.N) ? (n0 + rw_) : (p.N - 1);
        srcw[g] = (uint32_t)gn * (uint32_t)(K / 2) + lp * 16;
      }
    }
  };
  // piece idx = 2 g + isW: instruction g of the activation tile / of the weight tile (W4: the packed panel has pieces g < 2 only)
#define B2_PIECE(kt, stage, idx)                                                                              \
  do {                                                                                                        \
    char* sx_ = smem + (stage) * B2_STAGE + wave * 1024 + ((idx) >> 1) * 8192;                                \
    if (((idx) & 1) == 0)                                                                                     \
      __builtin_amdgcn_global_load_lds((glb_void*)(p.a + (kt) * B2K + srcx[(idx) >> 1]), (lds_void*)sx_, 16, 0, 0); \
    else if (!W4 || (idx) < 4)                                                                                \
      __builtin_amdgcn_global_load_lds((glb_void*)(p.w + (kt) * (W4 ? B2K / 2 : B2K) + srcw[(idx) >> 1]),    \
                                       (lds_void*)(sx_ + B2M * B2K), 16, 0, 0);                               \
  } while (0)
#define B2_ISSUE(kt, stage)                                                   \
  do {                                                                        \
    _Pragma("unroll") for (int x_ = 0; x_ < 8; ++x_) B2_PIECE(kt, stage, x_); \
  } while (0)

  // fragment reads for v_mfma_i32_16x16x64_i8: lane (r16 = lane & 15, q4 = lane >> 4) holds 16 consecutive k of row r16 of a
  // 16-row block: W rows wn*64 + 16 i + r16 (A operand), X rows wm*128 + 16 j + r16 (B operand), 16-B chunk 4 ks + q4 of the
  // 128-B K-tile row.  The row swizzle (row>>1)&7 equals (r16>>1)&7 for every i / j, so one chunk offset per k-step serves all
  // twelve reads (i, j become immediates); a b128 lane group {0-3, 12-15 | 20-27} then covers all 16 slots of a bank row.
  // W4: the 16 codes of chunk c are the 8 packed bytes at half (c & 1) of 16-B piece c >> 1 of the 64-B packed row (one
  // ds_read_b64 per operand, conflict-free: a 32-lane half reads 16 rows x one whole piece).
  const int r16 = lane & 15, q4 = lane >> 4;
  const int fsw = (r16 >> 1) & 7;
  int ck[2], cw4[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    ck[ks] = ((4 * ks + q4) ^ fsw) << 4;
    cw4[ks] = (((2 * ks + (q4 >> 1)) ^ ((r16 >> 2) & 3)) << 4) + 8 * (q4 & 1);
  }
  const int rowx = (wm * 128 + r16) * B2K;
  const int roww = B2M * B2K + (wn * 64 + r16) * (W4 ? B2K / 2 : B2K);

  const bool dma_top = !W4 && nk >= 32;
  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  int m0, n0;
  tile_origin(tile, m0, n0);
  set_sources(m0, n0);
  B2_ISSUE(0, 0);
  int pending_stores = 0;  // epilogue store instructions issued after the loads now in flight

  for (;;) {
    // acc[i][j][e]: channel n0 + wn*64 + 16 i + 4 q4 + e, token m0 + wm*128 + 16 j + r16
    v4i acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0;

    for (int kt = 0; kt < nk; ++kt) {
      // publish K-tile kt: my LDS-DMA for it has landed (everything older than the last `pending_stores`
      // vector-memory ops), then the barrier makes that true for every wave and also says every wave has
      // finished reading the other stage.
      if (pending_stores == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (pending_stores == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
      pending_stores = 0;
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const bool more = kt + 1 < nk;
      if (more && dma_top) B2_ISSUE(kt + 1, (kt + 1) & 1);
      const bool more_spread = more && !dma_top;
      const char* st = smem + (kt & 1) * B2_STAGE;
      // A K-tile is sixteen groups of 4 MFMAs: group g = 8 ks + j multiplies the four weight fragments of k-step ks (64 deep)
      // with the activation fragment of token block j.  Activation fragments live in a ring of four registers quads, read
      // three groups (192 MFMA cycles) ahead of their use; the weight fragments of k-step 1 are read during k-step 0.
      // 48 fragment registers: a block-wise double buffer (64) made hipcc spill a fragment address and reload it -- behind
      // s_waitcnt vmcnt(0), i.e. after the prefetch had drained -- on every K-tile.
      // The eight LDS-DMA pieces of K-tile kt+1 are either issued one per group in front of the first eight groups (short K:
      // +2 % at K = 1536) or all at the top of the tile (long K: +4 % at K = 8960 / 13824); A/B in one process,
      // tools/ab_gemm_variants.py.  Either way they have at least the second half of the tile plus the barrier to land.
      v4i wf[2][4], xr[4];
      const v4i m4 = {0x0f0f0f0f, 0x0f0f0f0f, 0x0f0f0f0f, 0x0f0f0f0f};
#define B2_LDW(ks)                                                                                          \
  _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                           \
    if (!W4) wf[ks][i] = *reinterpret_cast<const v4i*>(st + roww + ck[ks] + i * 16 * B2K);                  \
    else wraw[i] = *reinterpret_cast<const uint2*>(st + roww + cw4[ks] + i * 16 * (B2K / 2));              \
  }
  // W4: 8 packed bytes -> the 16 codes of the operand (byte b of dword d: code 8 d + b low nibble, code 8 d + 4 + b high nibble);
  // placed a few groups AFTER the reads so that the conversion does not wait for them
#define B2_UNPACK(ks)                                                                                       \
  if (W4) {                                                                                                 \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                         \
      const v4i u_ = {(int)wraw[i].x, (int)(wraw[i].x >> 4), (int)wraw[i].y, (int)(wraw[i].y >> 4)};        \
      wf[ks][i] = u_ & m4;                                                                                  \
    }                                                                                                       \
  }
#define B2_LDX(g) xr[(g) & 3] = *reinterpret_cast<const v4i*>(st + rowx + ck[(g) >> 3] + ((g) & 7) * 16 * B2K);
      uint2 wraw[4];
      B2_LDW(0)
      B2_LDX(0) B2_LDX(1) B2_LDX(2)
      B2_UNPACK(0)
#pragma unroll
      for (int g = 0; g < 16; ++g) {
        if (g + 3 < 16) { B2_LDX(g + 3) }
        if (g == 3) { B2_LDW(1) }
        if (g == 6) { B2_UNPACK(1) }
        if (W4) {  // six pieces over the first twelve groups (idx 5 and 7 are empty for the packed panel)
          if (g < 12 && g % 3 != 2 && more_spread) B2_PIECE(kt + 1, (kt + 1) & 1, g - g / 3);
        } else {
          if (g < 8 && more_spread) B2_PIECE(kt + 1, (kt + 1) & 1, g);
        }
        __builtin_amdgcn_sched_barrier(0);  // keep the reads (and the DMA piece) AHEAD of the MFMAs of group g
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i][g & 7] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[g >> 3][i], xr[g & 3], acc[i][g & 7], 0, 0, 0);
      }
#undef B2_UNPACK
#undef B2_LDW
#undef B2_LDX
    }

    // ---- all waves are done with LDS.  Epilogue, in this order:
    //   1. ONE round trip: this tile's 256 per-channel values (sW, zp*sW, bias, gate) -> LDS (stage 1, free until
    //      the next tile's second K-tile), per-token sA / sumA -> registers;
    //   2. request the next tile's first K-tile (stage 0) so that it flies under the store loop;
    //   3. store loop: reads LDS only (no vector-memory loads, hence no vmcnt waits: stores just queue up).
    //      With a residual the loop must also LOAD; then step 2 moves behind the loop (hipcc would otherwise
    //      wait vmcnt(0) on every residual load while LDS-DMA is in flight).
    const int cur_m0 = m0, cur_n0 = n0;
    const int next = tile + gridDim.x;
    const bool has_res = (p.epi & WANQ_EPI_GATE_RES) != 0;
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    float* chan = reinterpret_cast<float*>(smem + B2_STAGE);  // [4][256]: sW, zp*sW, bias, gate
    float sa_m[8], asum_m[8];
    int mcl[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sa_m[j] = 1.f;
      asum_m[j] = 0.f;
      const int mr = cur_m0 + wm * 128 + j * 16 + r16;
      mcl[j] = mr < p.M ? mr : p.M - 1;
    }
    if (OUT != WANQ_I32) {  // one uniform branch per dtype so that the eight loads of a kind issue together
      if (p.tok_dtype == WANQ_F32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) sa_m[j] = static_cast<const float*>(p.sa)[mcl[j]];
        if (p.zp) {
#pragma unroll
          for (int j = 0; j < 8; ++j) asum_m[j] = static_cast<const float*>(p.asum)[mcl[j]];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) sa_m[j] = __half2float(static_cast<const __half*>(p.sa)[mcl[j]]);
        if (p.zp) {
#pragma unroll
          for (int j = 0; j < 8; ++j) asum_m[j] = __half2float(static_cast<const __half*>(p.asum)[mcl[j]]);
        }
      }
    }
    if (OUT != WANQ_I32) {
      if (tid < 256) {
        const int nc = (cur_n0 + tid < p.N) ? cur_n0 + tid : p.N - 1;
        const float swv = vec_load(p.sw, p.ch_dtype, nc);
        float zv = 0.f;
        if (p.zp) zv = (p.zp_dtype == WANQ_I16) ? (float)static_cast<const short*>(p.zp)[nc] : static_cast<const float*>(p.zp)[nc];
        chan[tid] = swv;
        chan[256 + tid] = zv * swv;
        chan[512 + tid] = p.bias ? vec_load(p.bias, p.ch_dtype, nc) : 0.f;
        chan[768 + tid] = has_res ? p.gate[nc] : 0.f;
      }
      __syncthreads();
    }
    constexpr bool OUT16 = (OUT == WANQ_F16 || OUT == WANQ_BF16);
    if (next < ntiles) {
      tile_origin(next, m0, n0);
      set_sources(m0, n0);
      // the 16-bit no-residual path requests the next tile's first K-tile inside its store function (b2_store16_and_request)
      if (!has_res && !OUT16) B2_ISSUE(0, 0);
    }
    const bool full_tile = (cur_m0 + B2M <= p.M) && (cur_n0 + B2N <= p.N);
    // ---- store loop.  An accumulator has one token per lane and 4 channels per register quad, so storing it directly
    // writes 8-16 B per lane at a row stride: 32 partial lines per instruction (PMC: WRITE_SIZE 2.6-2.7x the output
    // bytes).  Each wave therefore turns its results through a private 4-KiB LDS buffer (32 tokens x 128 B, 16-B chunks
    // XORed with row&7; stage 1 is free until the next tile's second K-tile) and stores -- and reads the residual --
    // as whole 128-B lines, 8 lanes per line.  Same-wave LDS traffic only: no barrier.
    // (the lane-derived constants of the store loop are taken from an OPAQUE copy of the lane id: derived from `lane` itself they
    // are loop-invariant, hipcc keeps them live across the main loop -- which sits at the 256-register cap -- and spills them; every
    // reload in the store loop then carries s_waitcnt vmcnt(0), i.e. waits for the next tile's LDS-DMA and for the acknowledgement
    // of the stores issued so far)
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    char* tb = smem + B2_STAGE + 4096 + wave * 4096;
    const int rd_row = lane_e >> 3, rd_c = lane_e & 7;  // read-back: row rd_row + 8*pass, 16-B chunk rd_c
    const int e16 = lane_e & 15, eq4 = lane_e >> 4;     // = r16, q4 (opaque copies)
    const int tok_base = cur_m0 + wm * 128;
    if (OUT16 && !has_res) {
      if (OUT16)  // (instantiated for the 16-bit output types only)
        b2_store16_and_request<OUT16 ? OUT : WANQ_BF16, W4>(p, acc, sa_m, asum_m, chan, tb, smem, next < ntiles, srcx, srcw, wave, cur_n0, tok_base,
                                                            wn, e16, eq4, rd_row, rd_c, full_tile);
    } else if (OUT == WANQ_F32 && has_res) {
      // stage 0 is free until the next tile's first K-tile is requested, which this path does after the store loop
      b2_store_f32_res(p, acc, sa_m, asum_m, chan, tb, smem + wave * 8192, cur_n0, tok_base, wn, e16, eq4, rd_row, rd_c, lane_e, full_tile);
    } else {
      // chunk = 32 tokens x 32 channels (channel blocks 2 ih, 2 ih + 1) of fp32 / int32; gate*y + residual is applied after
      // the turn, on whole lines
#pragma unroll
      for (int J = 0; J < 4; ++J) {
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            const int j = 2 * J + jj, tr = jj * 16 + e16;
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
              const int i = 2 * ih + ii;
              char* dst = tb + tr * 128 + (((4 * ii + eq4) ^ (tr & 7)) << 4);
              if (OUT == WANQ_I32) {
                *reinterpret_cast<int4*>(dst) = make_int4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
              } else {
                const int nl = wn * 64 + i * 16 + 4 * eq4;
                const float4 sw4 = *reinterpret_cast<const float4*>(chan + nl);
                const float4 zs4 = *reinterpret_cast<const float4*>(chan + 256 + nl);
                const float4 b4 = *reinterpret_cast<const float4*>(chan + 512 + nl);
                const float swa[4] = {sw4.x, sw4.y, sw4.z, sw4.w}, zsa[4] = {zs4.x, zs4.y, zs4.z, zs4.w};
                const float ba[4] = {b4.x, b4.y, b4.z, b4.w};
                float y[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                  y[e] = fmaf((float)acc[i][j][e] * sa_m[j], swa[e], fmaf(asum_m[j], zsa[e], ba[e]));
                if (p.epi & WANQ_EPI_GELU) {
#pragma unroll
                  for (int e = 0; e < 4; ++e) y[e] = gelu_tanh_fast_f32(y[e]);
                }
                *reinterpret_cast<float4*>(dst) = make_float4(y[0], y[1], y[2], y[3]);
              }
            }
          }
#pragma unroll
          for (int ps = 0; ps < 4; ++ps) {
            const int row = rd_row + 8 * ps;
            const char* src = tb + row * 128 + ((rd_c ^ (row & 7)) << 4);
            const int tok = tok_base + J * 32 + row;
            const int nl = wn * 64 + ih * 32 + rd_c * 4;
            const int n = cur_n0 + nl;
            if (!(full_tile || (tok < p.M && n < p.N))) continue;
            const int64_t o = (int64_t)tok * p.N + n;
            if (OUT == WANQ_I32) {
              *reinterpret_cast<int4*>(static_cast<int*>(p.out) + o) = *reinterpret_cast<const int4*>(src);
            } else {
              const float4 v = *reinterpret_cast<const float4*>(src);
              float y[4] = {v.x, v.y, v.z, v.w};
              if (has_res) {
                float r4[4];
                OutIo<OUT == WANQ_I32 ? WANQ_F32 : OUT>::load4(p.residual, o, r4);
                const float4 g4 = *reinterpret_cast<const float4*>(chan + 768 + nl);
                y[0] = fmaf(y[0], g4.x, r4[0]);
                y[1] = fmaf(y[1], g4.y, r4[1]);
                y[2] = fmaf(y[2], g4.z, r4[2]);
                y[3] = fmaf(y[3], g4.w, r4[3]);
              }
              OutIo<OUT == WANQ_I32 ? WANQ_F32 : OUT>::store4(p.out, o, y);
            }
          }
        }
      }
    }
#ifdef WANQ_CLOCK_PROBE
    if (next >= ntiles && blockIdx.x == 77 && tid == 0) { g_gemm_clk[0] = clock64() - clk_c0; g_gemm_clk[1] = wall_clock64() - clk_w0; }
#endif
    if (next >= ntiles) break;
    // A full tile issues exactly 16 (16-bit output) or 32 store instructions per wave after the LDS-DMA above; a ragged
    // tile may issue fewer (whole-wave skips), so it falls back to a full drain.
    if (has_res) {
      // fp32 + residual: every wave's residual buffers live in stage 0, which the next tile's first K-tile is about to fill
      // (wave w's pieces land in the buffers of waves g and 4 + g): all waves must be out of the store loop first
      if (OUT == WANQ_F32) __builtin_amdgcn_s_barrier();
      B2_ISSUE(0, 0);  // behind the stores: the first barrier of the next tile drains them (vmcnt(0))
      pending_stores = 0;
    } else {
      pending_stores = full_tile ? (OUT16 ? 16 : 32) : 0;
    }
    tile = next;
  }
#undef B2_ISSUE
}

template <int OUT, bool W4>
static int launch_gemm(GemmParams p, hipStream_t st) {
  // function-local static with an initialiser: set once, thread-safe (C++11)
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w8a8_kernel<OUT, W4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              2 * STAGE_BYTES);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w8a8_big_kernel<OUT, W4>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 2 * B2_STAGE);
    return true;
  }();
  (void)attr_set;
  if (p.M >= 512 && p.K % B2K == 0 && g_kernel_sel != 1 && (int64_t)p.M * p.K < (1ll << 32) && (int64_t)p.N * p.K < (1ll << 32)) {
    p.mt = (p.M + B2M - 1) / B2M;
    p.nt = (p.N + B2N - 1) / B2N;
    const int tiles = p.mt * p.nt;
    const int grid = tiles < 256 ? ((tiles + 7) & ~7) : 256;  // one workgroup per CU; % 8 == 0 for the XCD ranges
    hipLaunchKernelGGL((gemm_w8a8_big_kernel<OUT, W4>), dim3((unsigned)grid), dim3(512), 2 * B2_STAGE, st, p);
#ifdef WANQ_CLOCK_PROBE
    {
      (void)hipStreamSynchronize(st);
      unsigned long long h[2];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_clk), sizeof(h));
      printf("[clock] gemm M=%d N=%d K=%d: %llu cycles in %.1f us -> %.0f MHz\n", p.M, p.N, p.K, h[0], h[1] / 100.0, h[0] / (h[1] / 100.0));
    }
#endif
  } else {
    hipLaunchKernelGGL((gemm_w8a8_kernel<OUT, W4>), dim3((unsigned)(p.mt * p.nt)), dim3(256), 2 * STAGE_BYTES, st, p);
  }
  return check_launch(W4 ? "wanq_gemm_w4a8" : "wanq_gemm_w8a8");
}

static int gemm_entry(bool w4, const int8_t* a, const void* w, void* out, int out_dtype, const void* sa, const void* asum,
                      int tok_dtype, const void* sw, const void* bias, int ch_dtype, const void* zp, int zp_dtype,
                      const float* gate, const void* residual, int epi_flags, int64_t M, int N, int K, void* stream) {
  const char* what = w4 ? "wanq_gemm_w4a8" : "wanq_gemm_w8a8";
  WANQ_REQUIRE(a && w && out, WANQ_E_ARG, "%s: a, w and out must be non-NULL", what);
  WANQ_REQUIRE(out_dtype == WANQ_F16 || out_dtype == WANQ_BF16 || out_dtype == WANQ_F32 || out_dtype == WANQ_I32,
               WANQ_E_ARG, "%s: bad out dtype %d", what, out_dtype);
  WANQ_REQUIRE(M >= 0 && M < (1ll << 31) / 2, WANQ_E_SHAPE, "%s: M=%lld out of range", what, (long long)M);
  WANQ_REQUIRE(N >= 8 && N % 8 == 0, WANQ_E_SHAPE, "%s: N=%d must be a positive multiple of 8", what, N);
  if (w4) WANQ_REQUIRE(K >= 32 && K % 32 == 0, WANQ_E_SHAPE, "%s: K=%d must be a positive multiple of 32 (packed 4-bit groups)", what, K);
  else WANQ_REQUIRE(K >= 16 && K % 16 == 0, WANQ_E_SHAPE, "%s: K=%d must be a positive multiple of 16", what, K);
  if (out_dtype != WANQ_I32) {
    WANQ_REQUIRE(sa && sw, WANQ_E_ARG, "%s: sa and sw are required for a floating output", what);
    WANQ_REQUIRE(is_vec(tok_dtype) && is_vec(ch_dtype), WANQ_E_ARG, "%s: tok/ch dtype must be F16 or F32", what);
    WANQ_REQUIRE(!zp || (asum && (zp_dtype == WANQ_I16 || zp_dtype == WANQ_F32)), WANQ_E_ARG,
                 "%s: zp needs asum and dtype I16 or F32", what);
    WANQ_REQUIRE(!(epi_flags & WANQ_EPI_GATE_RES) || (gate && residual), WANQ_E_ARG, "%s: WANQ_EPI_GATE_RES needs gate and residual", what);
  } else {
    WANQ_REQUIRE(epi_flags == 0, WANQ_E_ARG, "%s: int32 output takes no epilogue flags", what);
  }
  WANQ_REQUIRE((epi_flags & ~(WANQ_EPI_GELU | WANQ_EPI_GATE_RES)) == 0, WANQ_E_ARG, "%s: unknown epilogue flag", what);
  if (M == 0) return WANQ_OK;
  static const int group_m = [] { const char* e = getenv("WANQ_GEMM_GROUP_M"); const int v = e ? atoi(e) : 0; return v > 0 ? v : GROUP_M; }();
  GemmParams p{};
  p.a = a; p.w = static_cast<const int8_t*>(w); p.out = out; p.sa = sa; p.asum = asum; p.sw = sw; p.bias = bias; p.zp = zp; p.gate = gate;
  p.residual = residual; p.tok_dtype = tok_dtype; p.ch_dtype = ch_dtype; p.zp_dtype = zp_dtype; p.epi = epi_flags;
  p.M = (int)M; p.N = N; p.K = K; p.group_m = group_m;
  p.mt = (int)((M + BM - 1) / BM);
  p.nt = (N + BN - 1) / BN;
  WANQ_REQUIRE((int64_t)p.mt * p.nt < (1ll << 31), WANQ_E_SHAPE, "%s: too many tiles", what);
  hipStream_t st = (hipStream_t)stream;
  if (g_kernel_sel == 0 && gemm_pp_eligible(p, out_dtype, w4)) return launch_gemm_pp(p, out_dtype, st);  // ping-pong persistent kernel
  if (w4) {
    switch (out_dtype) {
      case WANQ_F16: return launch_gemm<WANQ_F16, true>(p, st);
      case WANQ_BF16: return launch_gemm<WANQ_BF16, true>(p, st);
      case WANQ_F32: return launch_gemm<WANQ_F32, true>(p, st);
      default: return launch_gemm<WANQ_I32, true>(p, st);
    }
  }
  switch (out_dtype) {
    case WANQ_F16: return launch_gemm<WANQ_F16, false>(p, st);
    case WANQ_BF16: return launch_gemm<WANQ_BF16, false>(p, st);
    case WANQ_F32: return launch_gemm<WANQ_F32, false>(p, st);
    default: return launch_gemm<WANQ_I32, false>(p, st);
  }
}

}  // namespace wanq

using namespace wanq;

extern "C" int wanq_gemm_w8a8(const int8_t* a, const int8_t* w, void* out, int out_dtype, const void* sa,
                              const void* asum, int tok_dtype, const void* sw, const void* bias, int ch_dtype,
                              const void* zp, int zp_dtype, const float* gate, const void* residual, int epi_flags,
                              int64_t M, int N, int K, void* stream) {
  return gemm_entry(false, a, w, out, out_dtype, sa, asum, tok_dtype, sw, bias, ch_dtype, zp, zp_dtype, gate, residual, epi_flags,
                    M, N, K, stream);
}

extern "C" int wanq_gemm_select_kernel(int which) {
  if (which < 0 || which > 2) return -1;
  const int prev = g_kernel_sel;
  g_kernel_sel = which;
  return prev;
}

extern "C" int wanq_gemm_w4a8(const int8_t* a, const uint8_t* w_packed, void* out, int out_dtype, const void* sa,
                              const void* asum, int tok_dtype, const void* sw, const void* bias, int ch_dtype,
                              const void* zp, int zp_dtype, const float* gate, const void* residual, int epi_flags,
                              int64_t M, int N, int K, void* stream) {
  return gemm_entry(true, a, w_packed, out, out_dtype, sa, asum, tok_dtype, sw, bias, ch_dtype, zp, zp_dtype, gate, residual,
                    epi_flags, M, N, K, stream);
}
