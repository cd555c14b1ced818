// W8A8 GEMM, persistent PING-PONG kernel (v3) for gfx950: v_mfma_i32_16x16x64_i8, 256(M) x 256(N) tile, 8 waves.
//
// Why a third kernel.  The v2 persistent kernel (gemm_w8a8.hip) lets all eight waves run the same K-tile body between two
// barriers: every wave issues its LDS-DMA pieces, its fragment reads and its MFMAs in the same window, and the probe of that
// loop (tools/probes/mfma_lds_probe.hip) shows the 64 KiB of LDS-DMA per K-tile adding their whole cost to the MFMA time
// (4002 cycles per K-tile against 2048 of MFMA).  Here the two waves of a SIMD never do the same thing at the same time:
//
//   * waves 0-3 (group 0, one per SIMD) and waves 4-7 (group 1, their SIMD partners) run the SAME program one barrier
//     interval apart (group 1 executes one extra s_barrier in front of a tile, group 0 one behind it).  A K-tile is TWO
//     phases per group, each phase = LOAD (16 or 8 ds_read_b128 of the next fragments, 4 LDS-DMA pieces, lgkmcnt(0)) |
//     barrier | BURST (32 back-to-back MFMAs = half of the wave's 128 x 64 output over the whole 128-deep K-tile,
//     s_setprio 1, then one counted vmcnt wait) | barrier.  While one wave of a SIMD bursts, its partner loads: the matrix
//     pipe sees one uninterrupted MFMA stream per slot and the partner's vector-memory / LDS issue never sits in front of an MFMA.
//   * The LDS-DMA stream is spread and counted, never drained inside a tile (schedule below).
//   * Two 64-KiB K-tile buffers.  X rows = tokens in tile order (group g owns rows 128 g .. +127, halves a / b of 64);
//     W rows = [half a of waves 0-3 | half b of waves 0-3], 32 channels each (wave c's channels 64 c + 32 s + r sit at
//     physical row 128 s + 32 c + r), so that a load phase's reads are contiguous sub-tiles.  Rows are 128 B, 16-B chunk
//     index XORed with (row >> 1) & 7 on the DMA SOURCE address and on the fragment reads (conflict-free ds_read_b128 for
//     the 16x16x64 operand: gemm_w8a8.hip).
//   * The issue stream runs two K-tiles ahead of the compute stream and does not know about tile boundaries: the first
//     K-tiles of the workgroup's NEXT output tile arrive under the last K-tiles of this one.
//
// Schedule of one K-tile t, WANQ_PP_BURST = 32 (shipped; slots relative to group 0's first burst, G0 bursts in even slots):
//   slot  group  phase  fragment reads                 LDS-DMA issued (4 pieces per wave)
//   -1    G0     LA     X0a(t), Wa(t), Wb(t)  (16)     X0a, X0b (t+1)
//    0    G1     LA     X1a(t), Wa(t), Wb(t)           X1a, X1b (t+1)
//    1    G0     LB     X0b(t)                (8)      G0's W rows (0-63, 128-191) of (t+2)
//    2    G1     LB     X1b(t)                         G1's W rows (64-127, 192-255) of (t+2)
// Bursts QA = Xa x (Wa, Wb), QB = Xb x (Wb, Wa): 64 fragment registers.  Per wave the vector-memory queue reads
// ... W(t+1) | X(t+1) | W(t+2) | X(t+2) ...; the wait at the END OF A BURST leaves the youngest group in flight and retires the
// one before it (vmcnt(4)): W(t+1) behind QA(t), X(t+1) behind QB(t) -- 3.5 slots after its issue, one barrier before its first
// reader (the issuing waves' wait, then the barrier every reader passes).  Write-after-read: W is re-filled ONE slot after its
// last reader (G1's LA), so load phases finish their reads (lgkmcnt(0)) in front of their barrier; X two slots after.
// WANQ_PP_BURST = 16 builds the first form (four phases per K-tile, 16-MFMA bursts, one 8-KiB chunk per slot, vmcnt(10) at the
// end of every load phase: 2775 cycles per K-tile against 2500-2600; kept for the A/B of profiles/r04_l_*, r04_m_*).
//
// Epilogue: the v2 kernel's store path (per-wave 4-KiB LDS turn buffers -> whole 128-B lines); the tile's per-token and
// per-channel values are prefetched by LDS-DMA into the wave's idle turn buffer two K-tiles ahead (all-fp32 parameter sets;
// otherwise loaded straight into registers) -- no LDS staging area, no workgroup barrier: the ring is full of the next tile's
// K-tiles.  Store instructions share the vmcnt queue with the LDS-DMA pieces, so the first burst behind an epilogue waits with
// the count raised by the stores a FULL tile issues (16 / 32); a ragged tile drains after its store loop instead.  The fp32 +
// gate + residual epilogue needs 64 KiB for its residual prefetch ring: its kernels do not request the next tile's second
// K-tile under the last K-tile, use that buffer for the ring and request it behind the store loop.
#include "gemm_params.h"
#include <stdlib.h>
#include <type_traits>

namespace wanq {
namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

constexpr int PM = 256, PN = 256, PK = 128;
constexpr int PBUF = (PM + PN) * PK;  // one K-tile buffer: 64 KiB
constexpr int PXB = PM * PK;          // offset of the W rows inside a buffer
constexpr int PTURN = 2 * PBUF;       // eight 4-KiB turn buffers behind the ring
constexpr int PLDS = PTURN + 8 * 4096;  // 160 KiB
#if (defined(WANQ_PP_ABL_NODMA) || defined(WANQ_PP_ABL_NOREAD)) && !defined(WANQ_ALLOW_ABLATIONS)
#error "WANQ_PP_ABL_* build deliberately wrong kernels (timing ablations): add -DWANQ_ALLOW_ABLATIONS, never in build.py's library"
#endif
#ifndef WANQ_PP_BURST
#define WANQ_PP_BURST 32  // MFMAs per burst: 32 (two phases per K-tile, the shipped schedule) or 16 (four phases, the first form)
#endif

template <int OFF>
__device__ __forceinline__ void dsr(v4i& d, uint32_t addr) {
#ifdef WANQ_PP_ABL_NOREAD  // timing-only diagnostic build: fragments are never read (wrong results)
  asm volatile("" : "=v"(d) : "v"(addr));
#else
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
#endif
}

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

// The tile's dequantisation values of one lane: its 16 channels (4 per channel block i) and its 8 tokens (one per token block j)
struct LaneScales {
  float sw[4][4], zs[4][4], b[4][4];  // sW, zp * sW, bias
  float sa[8], asum[8];
};

template <int OUT>
__device__ __forceinline__ void load_lane_scales(const GemmParams& p, LaneScales& s, int n_base, int tok_base, int e16, int eq4) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s.sa[j] = 1.f;
    s.asum[j] = 0.f;
  }
  if (OUT == WANQ_I32) return;
  int mcl[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int mr = tok_base + j * 16 + e16;
    mcl[j] = mr < p.M ? mr : p.M - 1;
  }
  if (p.tok_dtype == WANQ_F32) {  // one uniform branch per dtype so that the eight loads of a kind issue together
#pragma unroll
    for (int j = 0; j < 8; ++j) s.sa[j] = static_cast<const float*>(p.sa)[mcl[j]];
    if (p.zp) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s.asum[j] = static_cast<const float*>(p.asum)[mcl[j]];
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) s.sa[j] = __half2float(static_cast<const __half*>(p.sa)[mcl[j]]);
    if (p.zp) {
#pragma unroll
      for (int j = 0; j < 8; ++j) s.asum[j] = __half2float(static_cast<const __half*>(p.asum)[mcl[j]]);
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    int n = n_base + i * 16 + 4 * eq4;  // channels past N are clamped: computed, never stored
    n = n + 4 <= p.N ? n : p.N - 4;
    load4_ch(p.sw, p.ch_dtype, n, s.sw[i]);
    if (p.zp) {
      float z[4];
      load4_ch(p.zp, p.zp_dtype, n, z);
#pragma unroll
      for (int e = 0; e < 4; ++e) s.zs[i][e] = z[e] * s.sw[i][e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) s.zs[i][e] = 0.f;
    }
    if (p.bias) {
      load4_ch(p.bias, p.ch_dtype, n, s.b[i]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) s.b[i][e] = 0.f;
    }
  }
}

// The same values for an all-fp32 parameter set (the block's own calls), PREFETCHED: read where they are used (above) the epilogue
// waited 3400-4000 cycles for them (in-kernel stamps, profiles/r04_d_gemm_pp_clock.txt: 40 % of the 16-bit epilogue) -- hipcc waits
// vmcnt(0) for an ordinary load while LDS-DMA is in flight, so the loads also waited for every K-tile piece behind them.  Two
// K-tiles ahead of the epilogue each wave requests its 128 tokens' sA / sumA and its 64 channels' sW / zp / bias / gate by
// LDS-DMA (4 B per lane) into its idle turn buffer; the epilogue reads them back from LDS.  Layout (floats): sA 0, sumA 128,
// sW 256, zp 320, bias 384, gate 448.
__device__ __forceinline__ bool scales_all_f32(const GemmParams& p) {
  return p.tok_dtype == WANQ_F32 && p.ch_dtype == WANQ_F32 && (!p.zp || p.zp_dtype == WANQ_F32);
}
__device__ __forceinline__ void prefetch_scales(const GemmParams& p, char* tb, int n_base, int tok_base, int lane, bool has_res) {
#define PP_SC_DMA(ptr, idx, off) \
  __builtin_amdgcn_global_load_lds((glb_void*)(static_cast<const float*>(ptr) + (idx)), (lds_void*)(tb + (off)), 4, 0, 0)
  int t0 = tok_base + lane, t1 = tok_base + 64 + lane, n = n_base + lane;
  t0 = t0 < p.M ? t0 : p.M - 1;
  t1 = t1 < p.M ? t1 : p.M - 1;
  n = n < p.N ? n : p.N - 1;
  // always EIGHT instructions (absent vectors re-request sW / sA into their unused slots): the schedule's vmcnt counts rely on it
  PP_SC_DMA(p.sa, t0, 0);
  PP_SC_DMA(p.sa, t1, 256);
  PP_SC_DMA(p.zp ? p.asum : p.sa, t0, 512);
  PP_SC_DMA(p.zp ? p.asum : p.sa, t1, 768);
  PP_SC_DMA(p.sw, n, 1024);
  PP_SC_DMA(p.zp ? p.zp : p.sw, n, 1280);
  PP_SC_DMA(p.bias ? p.bias : p.sw, n, 1536);
  PP_SC_DMA(has_res ? static_cast<const void*>(p.gate) : p.sw, n, 1792);
#undef PP_SC_DMA
}
template <int OFF>
__device__ __forceinline__ void dsr32(float& d, uint32_t addr) {
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// (asm reads: hipcc would put s_waitcnt vmcnt(0) in front of ordinary LDS reads that may alias an LDS-DMA in flight)
__device__ __forceinline__ void read_lane_scales(const GemmParams& p, LaneScales& s, float (&gt)[2][4], uint32_t tba, int e16, int eq4,
                                                 int rd_c, bool has_res) {
  const uint32_t a_tok = tba + e16 * 4, a_ch = tba + 1024 + eq4 * 16, a_g = tba + 1792 + rd_c * 16;
  v4i raw[12], rg[2];
  float sa[8], as[8];
#define PP_R8(dst, base) \
  dsr32<(base) + 0>(dst[0], a_tok); dsr32<(base) + 64>(dst[1], a_tok); dsr32<(base) + 128>(dst[2], a_tok); dsr32<(base) + 192>(dst[3], a_tok); \
  dsr32<(base) + 256>(dst[4], a_tok); dsr32<(base) + 320>(dst[5], a_tok); dsr32<(base) + 384>(dst[6], a_tok); dsr32<(base) + 448>(dst[7], a_tok);
  PP_R8(sa, 0)
  if (p.zp) { PP_R8(as, 512) }
#undef PP_R8
  dsr<0>(raw[0], a_ch); dsr<64>(raw[1], a_ch); dsr<128>(raw[2], a_ch); dsr<192>(raw[3], a_ch);
  if (p.zp) { dsr<256>(raw[4], a_ch); dsr<320>(raw[5], a_ch); dsr<384>(raw[6], a_ch); dsr<448>(raw[7], a_ch); }
  if (p.bias) { dsr<512>(raw[8], a_ch); dsr<576>(raw[9], a_ch); dsr<640>(raw[10], a_ch); dsr<704>(raw[11], a_ch); }
  if (has_res) { dsr<0>(rg[0], a_g); dsr<128>(rg[1], a_g); }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s.sa[j] = sa[j];
    s.asum[j] = p.zp ? as[j] : 0.f;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      s.sw[i][e] = __int_as_float(raw[i][e]);
      s.zs[i][e] = p.zp ? __int_as_float(raw[4 + i][e]) * s.sw[i][e] : 0.f;
      s.b[i][e] = p.bias ? __int_as_float(raw[8 + i][e]) : 0.f;
    }
  if (has_res) {
#pragma unroll
    for (int ih = 0; ih < 2; ++ih)
#pragma unroll
      for (int e = 0; e < 4; ++e) gt[ih][e] = __int_as_float(rg[ih][e]);
  }
}

// (GELU is a template parameter of the store loops: tested per call site, the flag put a branch behind every four values --
// 32 per tile -- and no two groups' conversion chains overlapped)
// PK: the same expression on channel PAIRS (v_pk_mul_f32 / v_pk_fma_f32: IEEE per half, bit-identical to the scalar form, 2.25
// instead of 4 vector instructions per value).  Used by the 16-bit and plain 32-bit store loops (bf16 q / k / v GEMM 1.050x, GELU
// ffn.0 1.048x, 14B shapes 1.02x alone); the gate + residual loop keeps the scalar form, which the packed one slowed to 0.953x at
// K = 1536 (profiles/r04_zz_gemm_pk_epilogue_ab.txt).
template <bool GELU, bool PK>
__device__ __forceinline__ void dequant4(const v4i& a, const LaneScales& s, int i, int j, float (&y)[4]) {
  if (PK) {
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f sa2 = {s.sa[j], s.sa[j]}, as2 = {s.asum[j], s.asum[j]};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const v2f af = {(float)a[2 * h], (float)a[2 * h + 1]};
      const v2f sw2 = {s.sw[i][2 * h], s.sw[i][2 * h + 1]}, zs2 = {s.zs[i][2 * h], s.zs[i][2 * h + 1]}, b2 = {s.b[i][2 * h], s.b[i][2 * h + 1]};
      const v2f r = __builtin_elementwise_fma(af * sa2, sw2, __builtin_elementwise_fma(as2, zs2, b2));
      y[2 * h] = r.x;
      y[2 * h + 1] = r.y;
    }
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)  // acc*sA*sW + (sumA*(zp*sW) + bias): the v2 kernel's expression, bit for bit
      y[e] = fmaf((float)a[e] * s.sa[j], s.sw[i][e], fmaf(s.asum[j], s.zs[i][e], s.b[i][e]));
  }
  if (GELU) {
#pragma unroll
    for (int e = 0; e < 4; ++e) y[e] = gelu_tanh_fast_f32(y[e]);
  }
}

// 16-bit store loop: chunk = 32 tokens (token blocks 2J, 2J+1) x 64 channels through the wave's turn buffer (32 rows x 128 B,
// 16-B chunks XORed with row & 7), stored as whole 128-B lines.  Returns nothing; a full tile issues 16 store instructions.
template <int OUT, bool GELU>
__device__ __forceinline__ void store16(const GemmParams& p, v4i (&acc)[4][8], const LaneScales& s, char* __restrict__ tb, int n_base,
                                        int tok_base, int e16, int eq4, int rd_row, int rd_c, bool full_tile) {
#pragma unroll
  for (int J = 0; J < 4; ++J) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * J + jj, tr = jj * 16 + e16;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float y[4];
        dequant4<GELU, true>(acc[i][j], s, i, j, y);
        const int cb = (i * 16 + 4 * eq4) * 2;  // byte column inside the 128-B row
        *reinterpret_cast<uint2*>(tb + tr * 128 + ((((cb >> 4) ^ (tr & 7)) << 4) | (cb & 15))) = pack16x4<OUT>(y);
      }
    }
    uint4 v[4];  // the four line reads together, then the stores (per line, hipcc waited for each read in front of its store)
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int row = rd_row + 8 * ps;
      v[ps] = *reinterpret_cast<const uint4*>(tb + row * 128 + ((rd_c ^ (row & 7)) << 4));
    }
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int tok = tok_base + J * 32 + rd_row + 8 * ps;
      const int n = n_base + rd_c * 8;
      if (full_tile || (tok < p.M && n < p.N))
        *reinterpret_cast<uint4*>(static_cast<uint16_t*>(p.out) + (int64_t)tok * p.N + n) = v[ps];
    }
  }
}

// 32-bit store loop without a residual (fp32 / raw int32): chunk = 32 tokens x 32 channels; a full tile issues 32 stores.
template <int OUT, bool GELU>
__device__ __forceinline__ void store32(const GemmParams& p, v4i (&acc)[4][8], const LaneScales& s, char* __restrict__ tb, int n_base,
                                        int tok_base, int e16, int eq4, int rd_row, int rd_c, bool full_tile) {
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
            float y[4];
            dequant4<GELU, true>(acc[i][j], s, i, j, y);
            *reinterpret_cast<float4*>(dst) = make_float4(y[0], y[1], y[2], y[3]);
          }
        }
      }
      int4 v[4];
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int row = rd_row + 8 * ps;
        v[ps] = *reinterpret_cast<const int4*>(tb + row * 128 + ((rd_c ^ (row & 7)) << 4));
      }
#pragma unroll
      for (int ps = 0; ps < 4; ++ps) {
        const int tok = tok_base + J * 32 + rd_row + 8 * ps;
        const int n = n_base + ih * 32 + rd_c * 4;
        if (!(full_tile || (tok < p.M && n < p.N))) continue;
        *reinterpret_cast<int4*>(static_cast<int*>(p.out) + (int64_t)tok * p.N + n) = v[ps];
      }
    }
  }
}

// fp32 + gate + residual store loop, residual lines prefetched ONE CHUNK AHEAD by LDS-DMA into rbuf (two 4-KiB halves per
// wave, lane-linear; the v2 kernel's b2_store_f32_res).  Chunk c = 2 J + ih = 32 tokens x 32 channels.  Counted waits for a
// full tile (the four stores of chunk c-1 and the four pieces of chunk c+1 stay in flight), a drain for a ragged one.
template <bool GELU>
__device__ __forceinline__ void store32_res(const GemmParams& p, v4i (&acc)[4][8], const LaneScales& s, const float (&gt)[2][4],
                                            char* __restrict__ tb, char* __restrict__ rbuf, int n_base, int tok_base, int e16, int eq4,
                                            int rd_row, int rd_c, int lane_e, bool full_tile) {
#define PP_RES_DMA(c)                                                                                          \
  _Pragma("unroll") for (int ps = 0; ps < 4; ++ps) {                                                           \
    int tok_ = tok_base + ((c) >> 1) * 32 + rd_row + 8 * ps, n_ = n_base + ((c) & 1) * 32 + rd_c * 4;          \
    tok_ = tok_ < p.M ? tok_ : p.M - 1;                                                                        \
    n_ = n_ + 4 <= p.N ? n_ : p.N - 4;                                                                         \
    __builtin_amdgcn_global_load_lds((glb_void*)(static_cast<const float*>(p.residual) + (int64_t)tok_ * p.N + n_), \
                                     (lds_void*)(rbuf + ((c) & 1) * 4096 + ps * 1024), 16, 0, 0);              \
  }
  PP_RES_DMA(0)
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int J = c >> 1, ih = c & 1;
    if (c + 1 < 8) { PP_RES_DMA(c + 1) }
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int j = 2 * J + jj, tr = jj * 16 + e16;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * ih + ii;
        float y[4];
        dequant4<GELU, false>(acc[i][j], s, i, j, y);
        *reinterpret_cast<float4*>(tb + tr * 128 + (((4 * ii + eq4) ^ (tr & 7)) << 4)) = make_float4(y[0], y[1], y[2], y[3]);
      }
    }
    if (!full_tile) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (c == 0 || c == 7) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int row = rd_row + 8 * ps;
      const int tok = tok_base + J * 32 + row;
      const int n = n_base + ih * 32 + rd_c * 4;
      if (!(full_tile || (tok < p.M && n < p.N))) continue;
      const float4 v = *reinterpret_cast<const float4*>(tb + row * 128 + ((rd_c ^ (row & 7)) << 4));
      const float4 rv = *reinterpret_cast<const float4*>(rbuf + (c & 1) * 4096 + ps * 1024 + lane_e * 16);
      *reinterpret_cast<float4*>(static_cast<float*>(p.out) + (int64_t)tok * p.N + n) =
          make_float4(fmaf(v.x, gt[ih][0], rv.x), fmaf(v.y, gt[ih][1], rv.y), fmaf(v.z, gt[ih][2], rv.z), fmaf(v.w, gt[ih][3], rv.w));
    }
  }
#undef PP_RES_DMA
}

#ifdef WANQ_PP_CLOCK  // diagnostic build only: shader clock and cycles of one workgroup's first K loop
__device__ unsigned long long g_pp_clk[4];
#endif
template <int OUT, bool RES>  // RES: gate + residual epilogue (fp32 output only); an instantiation of its own for the register allocator
__global__ __launch_bounds__(512, 2) void gemm_w8a8_pp_kernel(const GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2, c = wave & 3;  // group (token half of the tile), channel quarter
  const int K = p.K;
  const int nk = K / PK;
  const int ntiles = p.mt * p.nt;
  constexpr bool has_res = RES;
  const bool fast_scales = scales_all_f32(p) && p.K >= 4 * PK;  // (>= 4 K-tiles: the prefetch sits two K-tiles in front of the epilogue)
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

  // tile id -> (m0, n0): XCD-contiguous ids (bijective remap; gridDim.x % 8 == 0), then groups of group_m m-tiles (v2's walk)
  auto tile_origin = [&](int t, int& m0, int& n0) {
    const int xq = ntiles >> 3, xr = ntiles & 7, xcd = t & 7;
    const int wg = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (t >> 3);
    const int per_group = p.group_m * p.nt;
    const int group = wg / per_group;
    const int first_m = group * p.group_m;
    const int gsz = (p.mt - first_m < p.group_m) ? (p.mt - first_m) : p.group_m;
    const int in_g = wg - group * per_group;
    m0 = (first_m + in_g % gsz) * PM;
    n0 = (in_g / gsz) * PN;
  };

  // ---- issue side: byte offsets of this lane's 16 B inside the eight pieces a wave moves per K-tile (k = 0)
  //   X half a / b, piece q: tile row 128 g + 64 s + 16 c + 8 q + (lane >> 3);   W half a / b of group g: physical row
  //   128 s + 64 g + 16 c + 8 q + (lane >> 3) = channel 64 (P' >> 5) + 32 s + (P' & 31) with P' = the row inside the half
  uint32_t sxa[2], sxb[2], swa[2], swb[2];
  auto set_sources = [&](int m0, int n0, int lane) {  // (`lane`: see the note at the end of the epilogue)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int ra = g * 128 + c * 16 + q * 8 + (lane >> 3);
      const int lcx = (lane & 7) ^ ((ra >> 1) & 7);  // the same for row ra + 64
      const int ta = (m0 + ra < p.M) ? (m0 + ra) : (p.M - 1);
      const int tb_ = (m0 + ra + 64 < p.M) ? (m0 + ra + 64) : (p.M - 1);
      sxa[q] = (uint32_t)ta * (uint32_t)K + lcx * 16;
      sxb[q] = (uint32_t)tb_ * (uint32_t)K + lcx * 16;
      const int pr = g * 64 + c * 16 + q * 8 + (lane >> 3);  // physical row inside a W half (the same for + 128)
      const int lcw = (lane & 7) ^ ((pr >> 1) & 7);
      const int ca = n0 + (pr >> 5) * 64 + (pr & 31);
      const int na = (ca < p.N) ? ca : (p.N - 1);
      const int nb = (ca + 32 < p.N) ? (ca + 32) : (p.N - 1);
      swa[q] = (uint32_t)na * (uint32_t)K + lcw * 16;
      swb[q] = (uint32_t)nb * (uint32_t)K + lcw * 16;
    }
  };
  const int dstx = (g * 128 + c * 16) * PK;        // + 8192 for half b, + 1024 q
  const int dstw = PXB + (g * 64 + c * 16) * PK;   // + 16384 for half b, + 1024 q
  int tileI, kI = 0;
  uint32_t kIoff = 0, bI = 0;
#ifdef WANQ_PP_ABL_NODMA  // timing-only diagnostic build: no LDS-DMA behind the prologue (wrong results)
#define PP_ISSUE(base, src, dst)                                                                                   \
  do {                                                                                                             \
    if (pp_prologue) {                                                                                             \
      _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_)                                                             \
        __builtin_amdgcn_global_load_lds((glb_void*)((base) + kIoff + src[q_]), (lds_void*)(smem + bI + (dst) + q_ * 1024), 16, 0, 0); \
    }                                                                                                              \
  } while (0)
#else
#define PP_ISSUE(base, src, dst)                                                                                   \
  do {                                                                                                             \
    _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_)                                                               \
      __builtin_amdgcn_global_load_lds((glb_void*)((base) + kIoff + src[q_]), (lds_void*)(smem + bI + (dst) + q_ * 1024), 16, 0, 0); \
  } while (0)
#endif
#define PP_ISSUE_XA() PP_ISSUE(p.a, sxa, dstx)
#define PP_ISSUE_WA() PP_ISSUE(p.w, swa, dstw)
#define PP_ISSUE_WB() PP_ISSUE(p.w, swb, dstw + 16384)
  // X half b closes a K-tile of the issue stream: step to the next K-tile, at the end of a tile to the workgroup's next tile
  // (behind the last tile the stream keeps re-requesting that tile's first K-tiles: bytes nobody reads, L2 hits, and every
  // vmcnt count of the schedule stays what it is)
#define PP_ISSUE_XB_ADVANCE()                         \
  do {                                                \
    PP_ISSUE(p.a, sxb, dstx + 8192);                  \
    ++kI;                                             \
    kIoff += PK;                                      \
    bI ^= PBUF;                                       \
    if (kI == nk) {                                   \
      kI = 0;                                         \
      kIoff = 0;                                      \
      if (tileI + (int)gridDim.x < ntiles) {          \
        tileI += gridDim.x;                           \
        int mi_, ni_;                                 \
        tile_origin(tileI, mi_, ni_);                 \
        set_sources(mi_, ni_, lane);                  \
      }                                               \
    }                                                 \
  } while (0)

  // ---- compute side: fragment read addresses for v_mfma_i32_16x16x64_i8 (lane (r16, q4): row r16 of a 16-row block, 16-B chunk
  // 4 ks + q4 of the 128-B row); the row swizzle (row >> 1) & 7 = (r16 >> 1) & 7 for every block, so blocks are immediates
  uint32_t xrd[2], wrd[2];
  auto set_read_addresses = [&](int lane) {
    const int r16 = lane & 15, q4 = lane >> 4;
    const int fsw = (r16 >> 1) & 7;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int ck = ((4 * ks + q4) ^ fsw) << 4;
      xrd[ks] = lds_base + (g * 128 + r16) * PK + ck;
      wrd[ks] = lds_base + PXB + (c * 32 + r16) * PK + ck;
    }
  };
  set_read_addresses(lane);

  int tile = blockIdx.x;
  if (tile >= ntiles) return;
  int m0, n0;
  tile_origin(tile, m0, n0);
  tileI = tile;
  set_sources(m0, n0, lane);
  bool pp_prologue = true;
  (void)pp_prologue;
#if WANQ_PP_BURST == 32
  // prologue = what the schedule would have issued in front of the first load phase, in its order: W(0), X(0), W(1)
  PP_ISSUE_WA(); PP_ISSUE_WB(); PP_ISSUE_XA(); PP_ISSUE_XB_ADVANCE();
  PP_ISSUE_WA(); PP_ISSUE_WB();
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#else
  // prologue = the seven chunk issues the schedule would have made in front of the first load phase, in its order
  PP_ISSUE_XA(); PP_ISSUE_WA(); PP_ISSUE_WB(); PP_ISSUE_XB_ADVANCE();
  PP_ISSUE_XA(); PP_ISSUE_WA(); PP_ISSUE_WB();
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
#endif
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  pp_prologue = false;
#ifdef WANQ_PP_ABL_NODMA
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif

  // Start-time stagger (experiment hook, off by default: launch_pp): workgroups start in `coh_n` cohorts, `coh_ticks` (10-ns units)
  // apart, so that one cohort's epilogue (an HBM burst: 32 MiB of 16-bit stores or 128 MiB of residual read + write per tile
  // round) falls into the others' K loops.  (The first K-tiles are already requested.)
  if (p.coh_n > 1) {
    const int coh = (blockIdx.x >> 3) % p.coh_n;  // neighbours inside an XCD take different cohorts
    if (coh > 0) {
      const unsigned long long until = wall_clock64() + (unsigned long long)coh * p.coh_ticks;
      while (wall_clock64() < until) __builtin_amdgcn_s_sleep(8);
    }
  }
#ifdef WANQ_PP_JITTER
  unsigned pp_jit = 0x9e3779b9u * (unsigned)(wave + 1) + blockIdx.x * 7919u;
#endif
  uint32_t bC = 0;
  // vector-memory wait at the end of a load phase: all but the pieces of the five youngest load phases (10), plus the store
  // instructions of the epilogue while they are younger than the oldest of those phases (vm_left phases, vm_mode: 1 = 16 stores,
  // 2 = 32 stores, 3 = behind the residual epilogue, whose own waits retired everything older than its last 8 stores)
  int vm_left = 0, vm_mode = 0;
#define PP_WAITVM()                                                              \
  do {                                                                           \
    if (vm_left > 0) {                                                           \
      --vm_left;                                                                 \
      if (vm_mode == 1) asm volatile("s_waitcnt vmcnt(26)" ::: "memory");        \
      else if (vm_mode == 2) asm volatile("s_waitcnt vmcnt(42)" ::: "memory");   \
      else asm volatile("s_waitcnt vmcnt(18)" ::: "memory");                     \
    } else {                                                                     \
      asm volatile("s_waitcnt vmcnt(10)" ::: "memory");                          \
    }                                                                            \
  } while (0)
#ifdef WANQ_PP_JITTER  // race screen (diagnostic build, correct results): every wave idles a pseudo-random 0-1500 cycles in front of every
  // barrier, so that no ordering holds merely because the waves run in lock-step (tools/ab_gemm_variants.py must stay bit-equal)
#define PP_JIT()                                                                                   \
  do {                                                                                             \
    pp_jit = pp_jit * 1664525u + 1013904223u;                                                      \
    const unsigned n_ = __builtin_amdgcn_readfirstlane((pp_jit >> 24) & 15u);                      \
    for (unsigned i_ = 0; i_ < n_; ++i_) __builtin_amdgcn_s_sleep(1);                              \
  } while (0)
#else
#define PP_JIT() do { } while (0)
#endif
#define PP_BAR()                         \
  do {                                   \
    __builtin_amdgcn_sched_barrier(0);   \
    PP_JIT();                            \
    __builtin_amdgcn_s_barrier();        \
    __builtin_amdgcn_sched_barrier(0);   \
  } while (0)
  // one burst: acc[I0 .. I0+1][J0 .. J0+3] += W fragments (2 channel blocks) x X fragments (4 token blocks) over both k-steps
#define PP_BURST(WF, I0, J0)                                                                                       \
  do {                                                                                                             \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                                 \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)                                                            \
      _Pragma("unroll") for (int ii_ = 0; ii_ < 2; ++ii_)                                                          \
        _Pragma("unroll") for (int jj_ = 0; jj_ < 4; ++jj_)                                                        \
          acc[(I0) + ii_][(J0) + jj_] =                                                                            \
              __builtin_amdgcn_mfma_i32_16x16x64_i8(WF[ks_][ii_], xf[ks_][jj_], acc[(I0) + ii_][(J0) + jj_], 0, 0, 0); \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
  } while (0)
  // the 32-MFMA form: all four channel blocks (W halves a and b) x the four token blocks J0 .. J0+3; the reads were waited for in
  // front of the barrier
#define PP_BURST32(J0)                                                                                             \
  do {                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                                 \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_) {                                                          \
      _Pragma("unroll") for (int ii_ = 0; ii_ < 2; ++ii_)                                                          \
        _Pragma("unroll") for (int jj_ = 0; jj_ < 4; ++jj_)                                                        \
          acc[ii_][(J0) + jj_] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wfa[ks_][ii_], xf[ks_][jj_], acc[ii_][(J0) + jj_], 0, 0, 0); \
      _Pragma("unroll") for (int ii_ = 0; ii_ < 2; ++ii_)                                                          \
        _Pragma("unroll") for (int jj_ = 0; jj_ < 4; ++jj_)                                                        \
          acc[2 + ii_][(J0) + jj_] =                                                                               \
              __builtin_amdgcn_mfma_i32_16x16x64_i8(wfb[ks_][ii_], xf[ks_][jj_], acc[2 + ii_][(J0) + jj_], 0, 0, 0); \
    }                                                                                                              \
    __builtin_amdgcn_s_setprio(0);                                                                                 \
  } while (0)
#define PP_READ_X(S)                                                                   \
  do {                                                                                 \
    dsr<(S) * 8192 + 0 * 2048>(xf[0][0], xa0); dsr<(S) * 8192 + 1 * 2048>(xf[0][1], xa0); \
    dsr<(S) * 8192 + 2 * 2048>(xf[0][2], xa0); dsr<(S) * 8192 + 3 * 2048>(xf[0][3], xa0); \
    dsr<(S) * 8192 + 0 * 2048>(xf[1][0], xa1); dsr<(S) * 8192 + 1 * 2048>(xf[1][1], xa1); \
    dsr<(S) * 8192 + 2 * 2048>(xf[1][2], xa1); dsr<(S) * 8192 + 3 * 2048>(xf[1][3], xa1); \
  } while (0)
#define PP_READ_W(WF, S)                                                               \
  do {                                                                                 \
    dsr<(S) * 16384 + 0>(WF[0][0], wa0); dsr<(S) * 16384 + 2048>(WF[0][1], wa0);       \
    dsr<(S) * 16384 + 0>(WF[1][0], wa1); dsr<(S) * 16384 + 2048>(WF[1][1], wa1);       \
  } while (0)

  for (;;) {
    // acc[i][j][e]: channel n0 + 64 c + 16 i + 4 q4 + e, token m0 + 128 g + 16 j + r16
    v4i acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[i][j][r] = 0;
    if (g == 1) PP_BAR();  // group 1 runs one barrier interval behind group 0
#ifdef WANQ_PP_CLOCK
    const unsigned long long clk_c0 = clock64(), clk_w0 = wall_clock64();
#endif

    for (int kt = 0; kt < nk; ++kt) {
      const bool hold = has_res && kt == nk - 1;  // residual epilogue ahead: the next tile's second K-tile is requested behind it
      const uint32_t xa0 = xrd[0] + bC, xa1 = xrd[1] + bC, wa0 = wrd[0] + bC, wa1 = wrd[1] + bC;
      v4i xf[2][4], wfa[2][2], wfb[2][2];
#if WANQ_PP_BURST == 32
      // ---- two phases per K-tile, 32-MFMA bursts (a 16-MFMA burst paid ~90 cycles of barrier / refill per 256: 2775 cycles per
      // K-tile; profiles/r04_b_gemm_pingpong_clock.txt).  Slots relative to G0's first burst of K-tile t, G0 bursts in even slots:
      //   slot -1 G0 LA: reads X0a, Wa, Wb (16)   issues X0a, X0b (t+1)        slot 0 G1 LA: X1a, Wa, Wb     issues X1a, X1b (t+1)
      //   slot  1 G0 LB: reads X0b (8)            issues W rows of G0 (t+2)    slot 2 G1 LB: X1b             issues W rows of G1 (t+2)
      // Bursts QA = Xa x (Wa, Wb), QB = Xb x (Wb, Wa).  Per wave the vector-memory queue is ... W(t+1) | X(t+1) | W(t+2) | X(t+2) ...
      // (four pieces each): the wait at the END OF A BURST leaves the youngest group in flight and retires the one before it --
      // W(t+1) behind QA(t), X(t+1) behind QB(t) -- 3.5 slots after its issue and one barrier before its first reader.  W is
      // re-filled ONE slot after its last reader (G1's LA): load phases therefore finish their reads (lgkmcnt(0)) in front of
      // their barrier, not behind it.
      const bool sc_now = OUT != WANQ_I32 && fast_scales && kt == nk - 2;
      // phase A
      PP_JIT();
      PP_READ_X(0);
      PP_READ_W(wfa, 0);
      PP_READ_W(wfb, 1);
      PP_ISSUE_XA();
      PP_ISSUE_XB_ADVANCE();
      if (sc_now) {
        int lane_s = lane;  // opaque: computed from `lane`, the eight addresses are tile-invariant and hipcc carries them (16 registers)
        asm volatile("" : "+v"(lane_s));  // through the whole K loop, spilling other values
        prefetch_scales(p, smem + PTURN + wave * 4096, n0 + c * 64, m0 + g * 128, lane_s, has_res);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_BAR();
      PP_BURST32(0);
      if (vm_left > 0) {  // first burst behind an epilogue: its stores (16 / 32 of a full tile) are younger than W(t+1)
        vm_left = 0;
        if (vm_mode == 1) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
        else if (vm_mode == 2) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      } else if (sc_now) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");  // X(t+1) and the eight scale pieces stay in flight
      } else {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      PP_BAR();
      // phase B
      PP_JIT();
      PP_READ_X(1);
      if (!hold) {
        PP_ISSUE_WA();
        PP_ISSUE_WB();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_BAR();
      PP_BURST32(4);
      if (hold) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no W(t+2) was issued: X(t+1) is the youngest
      else if (sc_now) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      PP_BAR();
#else
      // phase 1
      PP_READ_X(0);
      PP_READ_W(wfa, 0);
      if (OUT != WANQ_I32 && kt == nk - 2 && fast_scales) prefetch_scales(p, smem + PTURN + wave * 4096, n0 + c * 64, m0 + g * 128, lane, has_res);
      PP_ISSUE_XB_ADVANCE();
      PP_WAITVM();
      PP_BAR();
      PP_BURST(wfa, 0, 0);
      PP_BAR();
      // phase 2
      PP_READ_W(wfb, 1);
      if (!hold) {
        PP_ISSUE_XA();
        PP_WAITVM();
      } else {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      }
      PP_BAR();
      PP_BURST(wfb, 2, 0);
      PP_BAR();
      // phase 3
      PP_READ_X(1);
      if (!hold) {
        PP_ISSUE_WA();
        PP_WAITVM();
      } else {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      }
      PP_BAR();
      PP_BURST(wfb, 2, 4);
      PP_BAR();
      // phase 4
      if (!hold) {
        PP_ISSUE_WB();
        PP_WAITVM();
      } else {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
      PP_BAR();
      PP_BURST(wfa, 0, 4);
      PP_BAR();
#endif
      bC ^= PBUF;
    }
#ifdef WANQ_PP_CLOCK
    if (blockIdx.x == 77 && tid == 0 && tile == blockIdx.x) { g_pp_clk[0] = clock64() - clk_c0; g_pp_clk[1] = wall_clock64() - clk_w0; }
#endif
    if (g == 0) PP_BAR();  // both groups enter the epilogue together

    // ---- epilogue.  The lane-derived constants come from an OPAQUE copy of the lane id: derived from `lane` itself they are
    // loop-invariant and hipcc keeps them live across the main loop (v2 kernel: spills whose reloads wait vmcnt(0)).
    const int next = tile + gridDim.x;
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    char* tb = smem + PTURN + wave * 4096;
    const int rd_row = lane_e >> 3, rd_c = lane_e & 7;  // read-back: row rd_row + 8 * pass, 16-B chunk rd_c
    const int e16 = lane_e & 15, eq4 = lane_e >> 4;     // = r16, q4
    const int tok_base = m0 + g * 128, n_base = n0 + c * 64;
    const bool full_tile = (m0 + PM <= p.M) && (n0 + PN <= p.N);
    constexpr bool OUT16 = (OUT == WANQ_F16 || OUT == WANQ_BF16);
    int stores_mode = OUT16 ? 1 : (OUT == WANQ_F32 && has_res) ? 3 : 2;
    // the store loop of the output type; instantiated once behind each way of getting the scales so that the prefetched path
    // does not inherit the other path's pending loads (hipcc joins them with s_waitcnt vmcnt(0): a drained K-tile prefetch)
    auto store_tile_g = [&](const LaneScales& sc, const float (&gt)[2][4], auto gelu_tag) {
      constexpr bool GELU = decltype(gelu_tag)::value;
      if constexpr (OUT16) {
        store16<OUT16 ? OUT : WANQ_BF16, GELU>(p, acc, sc, tb, n_base, tok_base, e16, eq4, rd_row, rd_c, full_tile);
      } else if constexpr (OUT == WANQ_F32 && has_res) {
        // the ring buffer of the tile's last K-tile (bC has already stepped past it) is free: nothing was requested into it
        store32_res<false>(p, acc, sc, gt, tb, smem + (bC ^ PBUF) + wave * 8192, n_base, tok_base, e16, eq4, rd_row, rd_c, lane_e, full_tile);
      } else {
        store32<OUT, GELU>(p, acc, sc, tb, n_base, tok_base, e16, eq4, rd_row, rd_c, full_tile);
      }
    };
    auto store_tile = [&](const LaneScales& sc, const float (&gt)[2][4]) {
      if (OUT16 && (p.epi & WANQ_EPI_GELU)) store_tile_g(sc, gt, std::true_type{});  // (GELU with a 32-bit output: v2 kernel)
      else store_tile_g(sc, gt, std::false_type{});
    };
#ifdef WANQ_PP_CLOCK
    const unsigned long long clk_e0 = clock64();
    unsigned long long clk_e1 = clk_e0;
#endif
    if (OUT != WANQ_I32 && fast_scales) {
      LaneScales sc;
      float gt[2][4];
      read_lane_scales(p, sc, gt, lds_base + PTURN + wave * 4096, e16, eq4, rd_c, has_res);
#ifdef WANQ_PP_CLOCK
      clk_e1 = clock64();
#endif
      store_tile(sc, gt);
    } else {
      LaneScales sc;
      float gt[2][4];
      load_lane_scales<OUT>(p, sc, n_base, tok_base, e16, eq4);
      if (has_res) {
#pragma unroll
        for (int ih = 0; ih < 2; ++ih) {
          int n = n_base + ih * 32 + rd_c * 4;
          n = n + 4 <= p.N ? n : p.N - 4;
          load4_ch(p.gate, WANQ_F32, n, gt[ih]);
        }
      }
      store_tile(sc, gt);
    }
#ifdef WANQ_PP_CLOCK
    if (blockIdx.x == 77 && tid == 0 && tile == blockIdx.x) { g_pp_clk[2] = clk_e1 - clk_e0; g_pp_clk[3] = clock64() - clk_e0; }
#endif
    if (next >= ntiles) break;
    {
      // The per-lane constants of the K loop (eight DMA source offsets, four fragment read addresses) are RECOMPUTED here from an
      // opaque copy of the lane id instead of being carried through the epilogue, where every register is taken: carried, hipcc
      // spilled them and reloaded them inside the K loop -- each reload behind s_waitcnt vmcnt(0), a drained prefetch per K-tile.
      int lane_k = lane;
      asm volatile("" : "+v"(lane_k));
      int mi_, ni_;
      tile_origin(tileI, mi_, ni_);
      set_sources(mi_, ni_, lane_k);
      set_read_addresses(lane_k);
    }
    if (stores_mode == 3) {
      // every wave is out of its residual ring before the held-back pieces (next tile, second K-tile) land in it
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      PP_BAR();
#if WANQ_PP_BURST == 32
      PP_ISSUE_WA(); PP_ISSUE_WB();  // W(1) of the next tile; its first burst's wait (vmcnt(4): X(1) only) retires it and the last stores
      vm_left = 0;
#else
      PP_ISSUE_XA(); PP_ISSUE_WA(); PP_ISSUE_WB();
      vm_mode = 3;
      vm_left = 3;
#endif
    } else if (full_tile) {
      vm_mode = stores_mode;
      vm_left = 5;
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // a ragged tile skips whole store instructions: no count to rely on
      vm_left = 0;
    }
    tile = next;
    tile_origin(tile, m0, n0);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the issue stream's last (unread) pieces must not outlive the workgroup's LDS
#undef PP_ISSUE
#undef PP_ISSUE_XA
#undef PP_ISSUE_WA
#undef PP_ISSUE_WB
#undef PP_ISSUE_XB_ADVANCE
#undef PP_WAITVM
#undef PP_BAR
#undef PP_BURST
#undef PP_BURST32
#undef PP_READ_X
#undef PP_READ_W
}

template <int OUT, bool RES>
int launch_pp(GemmParams p, hipStream_t st) {
  static const bool attr_set = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_w8a8_pp_kernel<OUT, RES>), hipFuncAttributeMaxDynamicSharedMemorySize, PLDS);
    return true;
  }();
  (void)attr_set;
  p.mt = (p.M + PM - 1) / PM;
  p.nt = (p.N + PN - 1) / PN;
  const int tiles = p.mt * p.nt;
  {
    // Start-time stagger: an experiment hook only (WANQ_GEMM_STAGGER="cohorts:ns").  Measured on the four cfg-B block shapes
    // (tools/gemm_stagger_scan.py, profiles/r04_f_gemm_stagger_scan.txt): alone, the short-K gate + residual GEMM gains 6 % with two
    // cohorts 12 us apart (its epilogue is an HBM burst longer than its K loop), 16-bit outputs lose 1-10 %, K = 8960 loses 1-2 %;
    // inside the denoising step the gain does not show (profiles/r04_j_bench_stagger_ab.txt): off by default.
    static const bool rescan = getenv("WANQ_GEMM_STAGGER_SCAN") != nullptr;  // tools/gemm_stagger_scan.py changes the setting between launches
    static const char* env0 = getenv("WANQ_GEMM_STAGGER");
    const char* env = rescan ? getenv("WANQ_GEMM_STAGGER") : env0;
    p.coh_n = 0;
    p.coh_ticks = 0;
    if (env) {
      int n = 0, ns = 0;
      if (sscanf(env, "%d:%d", &n, &ns) == 2) { p.coh_n = n; p.coh_ticks = ns / 10; }
    }
  }
  const int grid = tiles < 256 ? ((tiles + 7) & ~7) : 256;  // one workgroup per CU; % 8 == 0 for the XCD ranges
  hipLaunchKernelGGL((gemm_w8a8_pp_kernel<OUT, RES>), dim3((unsigned)grid), dim3(512), PLDS, st, p);
#ifdef WANQ_PP_CLOCK
  {
    (void)hipStreamSynchronize(st);
    unsigned long long h[4];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_pp_clk), sizeof(h));
    printf("[clock] pp gemm M=%d N=%d K=%d out=%d epi=%d: K loop %llu cycles = %.0f per K-tile, %.1f us -> %.0f MHz; epilogue: scales %llu, all %llu cycles\n",
           p.M, p.N, p.K, OUT, p.epi, h[0], (double)h[0] / (p.K / PK), h[1] / 100.0, h[0] / (h[1] / 100.0), h[2], h[3]);
  }
#endif
  return check_launch("wanq_gemm_w8a8");
}

}  // namespace

bool gemm_pp_eligible(const GemmParams& p, int out_dtype, bool w4) {
  if (w4) return false;
  if (p.M < 512 || p.K % PK != 0 || p.K < 2 * PK) return false;
  if ((int64_t)p.M * p.K >= (1ll << 32) || (int64_t)p.N * p.K >= (1ll << 32)) return false;
  if ((p.epi & WANQ_EPI_GATE_RES) && out_dtype != WANQ_F32) return false;
  if ((p.epi & WANQ_EPI_GELU) && out_dtype == WANQ_F32) return false;
  return true;
}

int launch_gemm_pp(const GemmParams& p, int out_dtype, hipStream_t st) {
  switch (out_dtype) {
    case WANQ_F16: return launch_pp<WANQ_F16, false>(p, st);
    case WANQ_BF16: return launch_pp<WANQ_BF16, false>(p, st);
    case WANQ_F32: return (p.epi & WANQ_EPI_GATE_RES) ? launch_pp<WANQ_F32, true>(p, st) : launch_pp<WANQ_F32, false>(p, st);
    default: return launch_pp<WANQ_I32, false>(p, st);
  }
}

}  // namespace wanq
