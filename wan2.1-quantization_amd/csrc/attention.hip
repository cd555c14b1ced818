// Flash-attention forward for gfx950, head_dim 128, non-causal, key-length masking.
//
//   O[q, h, :] = softmax_k( Q[q,h,:] . K[k,h,:] / sqrt(d) ) V[k,h,:]
//
// Two kernels in this file: attn_fwd16_kernel (v_mfma_f32_16x16x32_bf16 / v_mfma_i32_16x16x64_i8; the default, described in
// front of it further down) and attn_fwd_kernel (32x32x16 / 32x32x32; WANQ_ATTN_M16=0), described here -- the LDS-DMA ring,
// the barrier protocol, the lazy rescale and the accumulator-initialised softmax are common to both.
//
// Layout: Q/K/V/O are token-major [tokens, heads*128] bf16 (exactly what the q/k/v GEMMs write and what the
// o-projection's quantiser reads), so no head transposes exist anywhere.
//
// Structure: one workgroup = 8 waves = 256 queries of one head; every wave owns 32 queries for the whole
// kernel.  Per 64-key tile:
//   S^T = K . Q^T      A operand = K rows from LDS (ds_read_b128), B operand = Q held in registers;
//                      the 32x32 accumulator then has ONE query per lane (lane&31) and 16 keys in registers,
//                      so the online-softmax row statistics are lane-local (one cross-lane max with lane^32);
//   O^T += V^T . P^T   B operand = the S^T accumulator itself, converted to bf16 in place (the accumulator's
//                      register->key permutation 8(j>>2)+4h+(j&3) is matched by the order in which the A
//                      operand V^T is gathered with ds_read_b64_tr_b16), so P never touches LDS.
// K and V tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4) into a ring of three 32-KiB stages, two
// tiles ahead of the math, published by a counted s_waitcnt vmcnt(N) (N = the pieces this wave issued for the tile after:
// 8 for the issuing waves, 0 for the others; AT_WAIT_TILE_AHEAD) + one bare s_barrier per tile (DMA = true, the
// default; +2.8 % over the register-staged two-stage form, DMA = false, kept behind WANQ_ATTN_V1=1).  LDS rows are
// 256 B with the 16-B chunk index XORed by ((row&3)<<2 | (row>>2)&3): conflict-free for the b128 row reads of K, the
// transposed reads of V and the staging writes; the DMA writes lane-linearly, so it applies the swizzle on the
// source side.
#include "wanq_common.h"
#include <stdlib.h>

namespace wanq {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnParams {
  const uint16_t* q;
  const uint16_t* k;
  const uint16_t* v;
  uint16_t* o;
  int64_t q_stride, k_stride, v_stride, o_stride;  // elements between consecutive tokens
  int Lq, Lk, H;
  float c;  // softmax scale * log2(e)
  // split-KV (gridDim.z > 1): workgroup z covers key tiles [z*tiles_per_split, ...) and writes unnormalised partials
  int tiles_per_split;
  float* part_o;   // [splits, Lq, H*128] fp32: O^T accumulators relative to the split's reference maximum
  float* part_ml;  // [splits, Lq, H, 2]  fp32: (reference maximum m, row sum l)
  // int8 Q.K^T (QK8): per-(token, head) symmetric int8 codes [tokens, H*128] and fp32 scale planes [H][stride]
  const int8_t* q8;
  const int8_t* k8;
  int64_t q8_stride, k8_stride;   // bytes between consecutive tokens
  const float* q_scale;           // delta_q[h][token]
  const float* k_scale;           // delta_k[h][token], followed by the plane -12582912 * delta_k at + H * ks_stride
  int64_t qs_stride, ks_stride;
};

constexpr int AT_D = 128, AT_QW = 32, AT_NW = 8, AT_QB = AT_QW * AT_NW, AT_KB = 64;
constexpr int AT_TILE = AT_KB * AT_D * 2;  // 16 KiB per K or V tile
constexpr int AT_STAGE = 2 * AT_TILE;
// QK8 stage: int8 K tile (64 x 128 B) | bf16 V tile | 64 key scales | 64 dequantisation constants
constexpr int AT_K8 = AT_KB * AT_D;               // 8 KiB
constexpr int AT_SC8 = AT_K8 + AT_TILE;           // scales at 24 KiB
constexpr int AT_STAGE8 = AT_SC8 + 2 * AT_KB * 4;  // 25088 B
constexpr float AT_MAGIC = 12582912.0f;           // 1.5 * 2^23: int32 accumulators start at its bit pattern (see QK8 below)

__device__ __forceinline__ int at_off8(int row, int ch) { return row * 128 + ((ch ^ ((row >> 1) & 7)) << 4); }

__device__ __forceinline__ int at_off(int row, int ch) {
  return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

__device__ __forceinline__ bf16x8 at_join(s16x4 lo, s16x4 hi) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, vv);
}

#ifdef WANQ_CLOCK_PROBE  // diagnostic build only: shader clock held by one workgroup (clock64 ticks per 100-MHz wall tick)
__device__ unsigned long long g_clk[2];
#endif
#ifdef WANQ_ATTN_STAMP
__device__ unsigned long long g_stamp[8 * 16];
#define STAMP(i) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); acc_[i] += t_ - last_; last_ = t_; }
#else
#define STAMP(i)
#endif

// QK8 (int8 Q.K^T, the reference's q / k fake-quant recipe run on the integer matrix cores: Q/base/quant_attn.py:168-174,
// W/models/quant_opensora.py:431-436): q and k arrive as per-(token, head) symmetric int8 codes with fp32 scales.
//   S^T = K8 . Q8^T on v_mfma_i32_32x32x32_i8: 8 MFMAs per 64-key tile instead of 16, and the K tile is 8 KiB instead of 16.
//   The int32 accumulators START at 0x4B400000, the bit pattern of 12582912.0f = 1.5 * 2^23: |dot| <= 128 * 127 * 127 < 2^22,
//   so the accumulator's bits READ AS A FLOAT are exactly 12582912 + dot -- no int->float conversion -- and one fma per
//   score, t = fma(f, delta_k, -12582912 * delta_k), applies the per-key scale (the constant comes precomputed beside the
//   scale).  The per-query scale delta_q is folded into the exp2 coefficient.  P.V stays bf16.
template <bool DMA, bool SPLIT = false, bool QK8 = false>
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(const AttnParams p) {
  static_assert(!QK8 || DMA, "the int8 Q.K^T form exists for the LDS-DMA pipeline only");
  constexpr int STAGE = QK8 ? AT_STAGE8 : AT_STAGE;
  constexpr int VOFF = QK8 ? AT_K8 : AT_TILE;  // byte offset of the V tile inside a stage
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  // Workgroup -> (head, query block).  The dispatcher hands workgroup L = x + gridDim.x * y to XCD L % 8, so with the plain
  // (x, y) = (query block, head) reading every XCD works on every head at once and each of the eight L2s streams every K / V
  // tile (PMC: 1.02 GB per cfg-B launch for 0.30 GB of operands).  The remap gives XCD k the k-th contiguous eighth of the
  // head-major sequence (bijective for any grid): the 32 workgroups an XCD runs at a time share ONE head, whose K / V tiles
  // are then fetched by one L2 per pass of 32 query blocks instead of by all eight.
  int head = blockIdx.y, qblk = blockIdx.x;
#ifndef WANQ_ATTN_NO_XCD_MAP
  if (!SPLIT) {
    const int nqb = gridDim.x, T = nqb * (int)gridDim.y, L = (int)blockIdx.x + nqb * (int)blockIdx.y;
    const int xq = T >> 3, xr = T & 7, xcd = L & 7;
    const int i = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (L >> 3);
    head = i / nqb;
    qblk = i - head * nqb;
  }
#endif
#ifdef WANQ_CLOCK_PROBE
  const unsigned long long clk_c0 = clock64(), clk_w0 = wall_clock64();
#endif
  const int q0 = qblk * AT_QB + wave * AT_QW;
  const int nt = (p.Lk + AT_KB - 1) / AT_KB;
  float c = p.c;
  // key tiles of this workgroup: all of them, or one contiguous share under split-KV (the host makes every share non-empty)
  const int jt0 = SPLIT ? (int)blockIdx.z * p.tiles_per_split : 0;
  const int jt1 = SPLIT ? (jt0 + p.tiles_per_split < nt ? jt0 + p.tiles_per_split : nt) : nt;

  // ---- Q fragments: query (q0+fr), d = 16 s + 8 fh + [0,8)
  typedef int v4i __attribute__((ext_vector_type(4)));
  typedef int v16i __attribute__((ext_vector_type(16)));
  bf16x8 qf[QK8 ? 1 : 8];
  v4i qf8[QK8 ? 4 : 1];  // QK8: query (q0+fr), d = 32 s + 16 fh + [0,16)
  {
    int qr = q0 + fr;
    if (qr >= p.Lq) qr = p.Lq - 1;
    if (QK8) {
      const int8_t* qp = p.q8 + (int64_t)qr * p.q8_stride + head * AT_D + 16 * fh;
#pragma unroll
      for (int s = 0; s < 4; ++s) qf8[s] = *reinterpret_cast<const v4i*>(qp + 32 * s);
      c *= p.q_scale[(int64_t)head * p.qs_stride + qr];  // score = dot * delta_k * delta_q * softmax scale
    } else {
      const uint16_t* qp = p.q + (int64_t)qr * p.q_stride + head * AT_D + 8 * fh;
#pragma unroll
      for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
      // fold softmax scale * log2(e) into Q once: the accumulators then ARE exp2 arguments (minus the running maximum, which
      // rides in as the MFMA chain's initial accumulator, below) and the per-score fma of the usual form disappears
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[s][e] = (__bf16)((float)qf[s][e] * c);
    }
  }

  // ---- staging: thread handles 16-B chunks `tid` and `tid+512` of the 64x16-chunk K tile and V tile
  const int st_row0 = tid >> 4, st_ch = tid & 15;  // second chunk: row + 32
  const uint16_t* kbase = p.k + head * AT_D + st_ch * 8;
  const uint16_t* vbase = p.v + head * AT_D + st_ch * 8;
  // (explicit scalars + macros: arrays captured by a lambda end up in scratch memory)
  uint4 rk0, rk1, rv0, rv1;
  const int st_off0 = at_off(st_row0, st_ch), st_off1 = at_off(st_row0 + 32, st_ch);
#define AT_GLOAD(j)                                                                   \
  do {                                                                                \
    int kr0 = (j) * AT_KB + st_row0, kr1 = kr0 + 32;                                  \
    kr0 = kr0 < p.Lk ? kr0 : p.Lk - 1;                                                \
    kr1 = kr1 < p.Lk ? kr1 : p.Lk - 1;                                                \
    rk0 = *reinterpret_cast<const uint4*>(kbase + (int64_t)kr0 * p.k_stride);         \
    rk1 = *reinterpret_cast<const uint4*>(kbase + (int64_t)kr1 * p.k_stride);         \
    rv0 = *reinterpret_cast<const uint4*>(vbase + (int64_t)kr0 * p.v_stride);         \
    rv1 = *reinterpret_cast<const uint4*>(vbase + (int64_t)kr1 * p.v_stride);         \
  } while (0)
#define AT_LSTORE(stage)                                                              \
  do {                                                                                \
    char* sK_ = smem + (stage) * AT_STAGE;                                            \
    *reinterpret_cast<uint4*>(sK_ + st_off0) = rk0;                                   \
    *reinterpret_cast<uint4*>(sK_ + st_off1) = rk1;                                   \
    *reinterpret_cast<uint4*>(sK_ + AT_TILE + st_off0) = rv0;                         \
    *reinterpret_cast<uint4*>(sK_ + AT_TILE + st_off1) = rv1;                         \
  } while (0)

  // ---- DMA staging (global_load_lds_dwordx4): instruction i of wave w fills LDS rows 4(2w+i)..+3 of a tile, lane-linear
  // (16 B per lane), so the chunk swizzle is applied on the SOURCE side: the lane at physical chunk c fetches logical
  // chunk c ^ swz(row).
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  // The FIRST-dispatched half of the workgroup (waves 0-3) issues the whole tile, 8 pieces per wave (rows 16w+4i..+3 of the K
  // tile and of the V tile, i = 0..3): those waves win the issue arbitration against their SIMD partners and would otherwise
  // wait ~900 cycles per tile at the barrier (phase stamps), and a piece costs less when only four waves issue.
  const bool dma_wave = wave < 4;
  const int d_r = lane >> 4;
#define AT_DOFF(stride, i) ((uint32_t)((16 * (wave & 3) + 4 * (i) + d_r) * (int)(stride) + head * AT_D + ((((lane & 15) ^ (d_r << 2)) ^ (i)) << 3)) * 2u)
  // Per-lane byte offsets inside a tile are constants; a tile's base address is wave-uniform (scalar ALU), so a DMA costs no
  // vector arithmetic: `base + zero-extended 32-bit lane offset` is the instruction's own sgpr + vgpr addressing.  (The
  // 64-bit `row * stride` per lane and instruction it replaces cost ~20 quarter-rate integer multiplies per tile and
  // wave.)  Only a ragged last tile clamps rows, on the slow path.
  uint32_t d_k0 = AT_DOFF(p.k_stride, 0), d_k1 = AT_DOFF(p.k_stride, 1), d_k2 = AT_DOFF(p.k_stride, 2), d_k3 = AT_DOFF(p.k_stride, 3);
  uint32_t d_v0 = AT_DOFF(p.v_stride, 0), d_v1 = AT_DOFF(p.v_stride, 1), d_v2 = AT_DOFF(p.v_stride, 2), d_v3 = AT_DOFF(p.v_stride, 3);
#undef AT_DOFF
#define AT_DMA_F(base, off, tilebyte, i) \
  __builtin_amdgcn_global_load_lds((glb_void*)((base) + (off)), (lds_void*)(sK_ + (tilebyte) + 1024 * (i)), 16, 0, 0);
#define AT_DMA_S(base, stride, tilebyte, i, j)                                                                   \
  {                                                                                                             \
    int kr_ = (j) * AT_KB + 16 * (wave & 3) + 4 * (i) + d_r;                                                    \
    kr_ = kr_ < p.Lk ? kr_ : p.Lk - 1;                                                                          \
    const int col_ = head * AT_D + ((((lane & 15) ^ (d_r << 2)) ^ (i)) << 3);                                   \
    __builtin_amdgcn_global_load_lds((glb_void*)((base) + (int64_t)kr_ * (stride) + col_), (lds_void*)(sK_ + (tilebyte) + 1024 * (i)), 16, 0, 0); \
  }
  // QK8: the K tile is int8 (8 rows x 128 B per 1-KiB piece, 2 pieces per DMA wave: rows 16w + 8i + (lane>>3), physical chunk
  // lane&7 holds logical chunk (lane&7) ^ ((row>>1)&7)); waves 4 and 5 fetch the tile's 64 key scales / constants.
  const int d8_r = lane >> 3;
#define AT_D8OFF(i) ((uint32_t)((16 * (wave & 3) + 8 * (i) + d8_r) * (int)p.k8_stride + head * AT_D + ((((lane & 7) ^ (((8 * (i) + d8_r) >> 1) & 7))) << 4)))
  const uint32_t d8_k0 = QK8 ? AT_D8OFF(0) : 0, d8_k1 = QK8 ? AT_D8OFF(1) : 0;
#undef AT_D8OFF
#define AT_DMA8(j, stage)                                                                                       \
  do {                                                                                                          \
    char* st_ = smem + (stage) * STAGE;                                                                         \
    if (dma_wave) {                                                                                             \
      char* sK_ = st_ + (wave & 3) * 2048;                                                                      \
      char* sV_ = st_ + (wave & 3) * 4096;                                                                      \
      if (((j) + 1) * AT_KB <= p.Lk) {                                                                          \
        const char* kt_ = reinterpret_cast<const char*>(p.k8) + (int64_t)(j) * AT_KB * p.k8_stride;             \
        const char* vt_ = reinterpret_cast<const char*>(p.v) + (int64_t)(j) * (AT_KB * 2) * p.v_stride;         \
        __builtin_amdgcn_global_load_lds((glb_void*)(kt_ + d8_k0), (lds_void*)(sK_), 16, 0, 0);                 \
        __builtin_amdgcn_global_load_lds((glb_void*)(kt_ + d8_k1), (lds_void*)(sK_ + 1024), 16, 0, 0);          \
        { char* sK_ = sV_; AT_DMA_F(vt_, d_v0, AT_K8, 0) AT_DMA_F(vt_, d_v1, AT_K8, 1) AT_DMA_F(vt_, d_v2, AT_K8, 2) AT_DMA_F(vt_, d_v3, AT_K8, 3) } \
      } else {                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                      \
          int kr_ = (j) * AT_KB + 16 * (wave & 3) + 8 * i_ + d8_r;                                              \
          kr_ = kr_ < p.Lk ? kr_ : p.Lk - 1;                                                                    \
          const int col_ = head * AT_D + ((((lane & 7) ^ (((8 * i_ + d8_r) >> 1) & 7))) << 4);                  \
          __builtin_amdgcn_global_load_lds((glb_void*)(p.k8 + (int64_t)kr_ * p.k8_stride + col_), (lds_void*)(sK_ + 1024 * i_), 16, 0, 0); \
        }                                                                                                       \
        { char* sK_ = sV_; AT_DMA_S(p.v, p.v_stride, AT_K8, 0, j) AT_DMA_S(p.v, p.v_stride, AT_K8, 1, j) AT_DMA_S(p.v, p.v_stride, AT_K8, 2, j) AT_DMA_S(p.v, p.v_stride, AT_K8, 3, j) } \
      }                                                                                                         \
    } else if (wave < 6) { /* scale plane (wave 4) / constant plane (wave 5): 64 floats, one dword per lane */  \
      const float* sp_ = p.k_scale + (int64_t)(wave - 4) * p.H * p.ks_stride + (int64_t)head * p.ks_stride + (int64_t)(j) * AT_KB + lane; \
      __builtin_amdgcn_global_load_lds((glb_void*)sp_, (lds_void*)(st_ + AT_SC8 + (wave - 4) * 256), 4, 0, 0);  \
    }                                                                                                           \
  } while (0)
#define AT_DMA(j, stage)                                                                                        \
  do {                                                                                                          \
    if (QK8) { AT_DMA8(j, stage); break; }                                                                      \
    if (dma_wave) {                                                                                             \
      char* sK_ = smem + (stage) * AT_STAGE + (wave & 3) * 4096;                                                \
      if (((j) + 1) * AT_KB <= p.Lk) {                                                                          \
        const char* kt_ = reinterpret_cast<const char*>(p.k) + (int64_t)(j) * (AT_KB * 2) * p.k_stride;         \
        const char* vt_ = reinterpret_cast<const char*>(p.v) + (int64_t)(j) * (AT_KB * 2) * p.v_stride;         \
        AT_DMA_F(kt_, d_k0, 0, 0) AT_DMA_F(kt_, d_k1, 0, 1) AT_DMA_F(kt_, d_k2, 0, 2) AT_DMA_F(kt_, d_k3, 0, 3)  \
        AT_DMA_F(vt_, d_v0, AT_TILE, 0) AT_DMA_F(vt_, d_v1, AT_TILE, 1) AT_DMA_F(vt_, d_v2, AT_TILE, 2) AT_DMA_F(vt_, d_v3, AT_TILE, 3) \
      } else {                                                                                                  \
        AT_DMA_S(p.k, p.k_stride, 0, 0, j) AT_DMA_S(p.k, p.k_stride, 0, 1, j) AT_DMA_S(p.k, p.k_stride, 0, 2, j) AT_DMA_S(p.k, p.k_stride, 0, 3, j) \
        AT_DMA_S(p.v, p.v_stride, AT_TILE, 0, j) AT_DMA_S(p.v, p.v_stride, AT_TILE, 1, j) AT_DMA_S(p.v, p.v_stride, AT_TILE, 2, j) AT_DMA_S(p.v, p.v_stride, AT_TILE, 3, j) \
      }                                                                                                         \
    }                                                                                                           \
  } while (0)

  // ---- transposed-read lane constants for V^T: 16-lane group g, lane 4q+p inside it
  const int tg = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int v_row = 4 * (tg >> 1) + tq;                 // + 16 ks (+8 for the second read)
  const int v_ch = 2 * (tg & 1) + (tp >> 1);            // + 4 db
  const int v_half = 8 * (tp & 1);

  // byte addresses (within a stage, before the V-tile / key-slice immediates) of the 8 transposed reads of a key slice
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
  const uint32_t va0 = at_off(v_row, 0 + v_ch) + v_half, va1 = at_off(8 + v_row, 0 + v_ch) + v_half;
  const uint32_t va2 = at_off(v_row, 4 + v_ch) + v_half, va3 = at_off(8 + v_row, 4 + v_ch) + v_half;
  const uint32_t va4 = at_off(v_row, 8 + v_ch) + v_half, va5 = at_off(8 + v_row, 8 + v_ch) + v_half;
  const uint32_t va6 = at_off(v_row, 12 + v_ch) + v_half, va7 = at_off(8 + v_row, 12 + v_ch) + v_half;

  f32x16 o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  float m_run = QK8 ? -INFINITY : 0.f, l_run = 0.f;

  // LDS-DMA instructions a wave issues per tile (= what may stay in flight behind a counted wait): 8 for the DMA waves of the
  // bf16 form (waves 4-7 issue none: any count passes); QK8: 6 for waves 0-3 (2 K + 4 V pieces), 1 for waves 4-5 (scales)
#define AT_WAIT_TILE_AHEAD()                                                                \
  do {                                                                                      \
    if (!QK8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                              \
    else if (wave < 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                     \
    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                   \
  } while (0)
  if (DMA) {
    AT_DMA(jt0, 0);
    if (jt0 + 1 < jt1) {
      AT_DMA(jt0 + 1, 1);
      AT_WAIT_TILE_AHEAD();
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
  } else {
    AT_GLOAD(0);
    AT_LSTORE(0);
    if (nt > 1) AT_GLOAD(1);
    __syncthreads();
  }

#ifdef WANQ_ATTN_STAMP
  unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_) :: "memory");
#endif
  // bf16 form: -m_run (log2 domain) in all 16 registers: the initial accumulator of both S chains, rewritten only when the
  // running maximum moves (the first tile and lazy-rescale events)
  f32x16 sinit;
#pragma unroll
  for (int r = 0; r < 16; ++r) sinit[r] = 0.f;
  // The ring position (j - jt0) % 3 is made a compile-time constant by unrolling the tile loop over the three stages (two for
  // the register-staged form): every LDS address of a tile is then `per-lane constant + immediate`, which takes ~25 address
  // VALU instructions per tile out of the loop.
  // (The split-KV form keeps the ring position in a register instead: unrolled, hipcc runs it out of registers and its spill
  // reloads -- each behind an s_waitcnt vmcnt(0) -- would drain the prefetch.)
  // Static priority for the second-dispatched half of the workgroup: waves 4-7 lose every VALU / MFMA arbitration against
  // their SIMD partners (priority, then age) and are the critical path of a tile (phase stamps: S 1219 vs 786 cycles, P.V
  // 1500 vs 1100); one s_setprio for the whole kernel, no per-phase flips (+0.6 % measured, A/B in one process).
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);
  constexpr int UNR = SPLIT ? 1 : (DMA ? 3 : 2);
  int st3 = 0;
  for (int j0 = jt0; j0 < jt1; j0 += UNR) {
#pragma unroll
  for (int u = 0; u < UNR; ++u) {
    const int j = j0 + u;
    if (j >= jt1) break;
    const int cur = UNR == 1 ? st3 : u;
    const char* sK = smem + cur * STAGE;
    const char* sV = sK + VOFF;
    // every wave is past the barrier that ended tile j-1, so the stage that held it is free: tile j+2 goes there and has
    // two tile-times to land
    const int st_free = UNR == 1 ? (st3 == 0 ? 2 : st3 - 1) : (u + 2) % 3;
    // (opaque to the optimiser on purpose: otherwise it keeps the eight lane offsets zero-extended to 64 bits -- sixteen
    // registers -- live across the whole loop and spills; re-extending them costs the DMA waves eight VALU per tile)
    asm volatile("" : "+v"(d_k0), "+v"(d_k1), "+v"(d_k2), "+v"(d_k3), "+v"(d_v0), "+v"(d_v1), "+v"(d_v2), "+v"(d_v3));
    if (DMA && j + 2 < jt1) AT_DMA(j + 2, st_free);
    STAMP(0)

    // ---------------- S^T = K . Q^T  (two 32-key blocks)
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    if (QK8) {
      v16i a0, a1;
#pragma unroll
      for (int r = 0; r < 16; ++r) { a0[r] = 0x4B400000; a1[r] = 0x4B400000; }  // bits of 12582912.0f
      v4i kf[4][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        kf[s][0] = *reinterpret_cast<const v4i*>(sK + at_off8(fr, 2 * s + fh));
        kf[s][1] = *reinterpret_cast<const v4i*>(sK + at_off8(32 + fr, 2 * s + fh));
      }
      typedef float f32x4 __attribute__((ext_vector_type(4)));
      f32x4 sk0[4], sb0[4];  // key scales / constants of one 32-key block, 4 consecutive keys each
      const uint32_t sa = lds_base + cur * STAGE + AT_SC8 + 16 * fh;
#define AT_SC(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(sa), "n"(off))
      AT_SC(sk0[0], 0); AT_SC(sk0[1], 32); AT_SC(sk0[2], 64); AT_SC(sk0[3], 96);
      AT_SC(sb0[0], 256); AT_SC(sb0[1], 288); AT_SC(sb0[2], 320); AT_SC(sb0[3], 352);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        if (s + 2 < 4) {
          kf[s + 2][0] = *reinterpret_cast<const v4i*>(sK + at_off8(fr, 2 * (s + 2) + fh));
          kf[s + 2][1] = *reinterpret_cast<const v4i*>(sK + at_off8(32 + fr, 2 * (s + 2) + fh));
        }
        __builtin_amdgcn_sched_barrier(0);
        a0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[s][0], qf8[s], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(kf[s][1], qf8[s], a1, 0, 0, 0);
      }
      // t = dot * delta_k: register 4g+e holds key 8g + 4fh + e (+32 for the second block).  The scale / constant reads are
      // inline asm for the reason the V reads are (hipcc would put s_waitcnt vmcnt(0) in front of a plain LDS load while an
      // LDS-DMA is in flight and drain the prefetch); the first block's were issued ahead of the MFMAs above.
#define AT_SCWAIT() asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sk0[0]), "+v"(sk0[1]), "+v"(sk0[2]), "+v"(sk0[3]), "+v"(sb0[0]), "+v"(sb0[1]), "+v"(sb0[2]), "+v"(sb0[3]))
      AT_SCWAIT();
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) s0[4 * g + e] = fmaf(__int_as_float(a0[4 * g + e]), sk0[g][e], sb0[g][e]);
      AT_SC(sk0[0], 128); AT_SC(sk0[1], 160); AT_SC(sk0[2], 192); AT_SC(sk0[3], 224);
      AT_SC(sb0[0], 384); AT_SC(sb0[1], 416); AT_SC(sb0[2], 448); AT_SC(sb0[3], 480);
      AT_SCWAIT();
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) s1[4 * g + e] = fmaf(__int_as_float(a1[4 * g + e]), sk0[g][e], sb0[g][e]);
#undef AT_SC
#undef AT_SCWAIT
    } else if (DMA) {
      // K fragments run three d-slices ahead of the MFMAs that consume them
      bf16x8 kf[8][2];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        kf[s][0] = *reinterpret_cast<const bf16x8*>(sK + at_off(fr, 2 * s + fh));
        kf[s][1] = *reinterpret_cast<const bf16x8*>(sK + at_off(32 + fr, 2 * s + fh));
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (s + 3 < 8) {
          kf[s + 3][0] = *reinterpret_cast<const bf16x8*>(sK + at_off(fr, 2 * (s + 3) + fh));
          kf[s + 3][1] = *reinterpret_cast<const bf16x8*>(sK + at_off(32 + fr, 2 * (s + 3) + fh));
        }
        __builtin_amdgcn_sched_barrier(0);
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s][0], qf[s], s == 0 ? sinit : s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s][1], qf[s], s == 0 ? sinit : s1, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(sK + at_off(fr, 2 * s + fh));
        const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(sK + at_off(32 + fr, 2 * s + fh));
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[s], s == 0 ? sinit : s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[s], s == 0 ? sinit : s1, 0, 0, 0);
      }
    }
    STAMP(1)
    if (j == nt - 1 && (p.Lk & (AT_KB - 1))) {  // ragged last tile: keys >= Lk get -inf
      asm volatile("" ::: "memory");  // keeps this a branch: if-converted, it costs 32 selects per lane on EVERY tile
      const int kb = j * AT_KB + 4 * fh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kb + (r & 3) + 8 * (r >> 2);
        if (key >= p.Lk) s0[r] = -INFINITY;
        if (key + 32 >= p.Lk) s1[r] = -INFINITY;
      }
    }

    // ---------------- online softmax, one query per lane (its 64 scores live in lanes l and l^32)
    float mx = s0[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s0[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s1[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float ls = 0.f;
    if (QK8) {
      // scores are dot * delta_k here; c = delta_q * scale * log2(e) is per lane.  Running max, rescaled lazily: the
      // accumulators are touched only when some query's max grew by more than 2^6 (exp2 domain) since the last rescale
      // (wave-uniform vote), so P stays <= 64 instead of <= 1 -- same 8 significant bits in bf16, fp32 accumulators unaffected.
      if (__any((mx - m_run) * c > 6.0f)) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
        m_run = m_new;
        l_run *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
      }
    } else {
      // The accumulators hold s * scale * log2(e) - m_run already (Q carries the scale, the MFMA chains started from -m_run),
      // so mx is the growth of the row maximum over the reference and p = exp2(acc) with no further arithmetic.  The first
      // tile fixes the reference at its own maximum (whatever its sign); later the reference only grows, lazily: when some
      // query's maximum exceeds it by more than 2^6 (wave-uniform vote) -- P stays <= 64 instead of <= 1, the same 8
      // significant bits in bf16 -- and then O, l and this tile's scores are brought to the new reference.
      const bool first = (j == jt0);
      if (first || __any(mx > 6.0f)) {
        asm volatile("" ::: "memory");  // keep this a branch
        const float delta = first ? mx : fmaxf(mx, 0.f);
        if (!first) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          l_run *= alpha;
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
        }
        m_run = first ? delta : m_run + delta;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s0[r] -= delta;
          s1[r] -= delta;
          sinit[r] = -m_run;
        }
      }
    }
    // P of key slice ks = 2*block + t (registers 8t..8t+7 of that block's accumulator), two scores at a time so that the
    // pieces can be placed between the P.V MFMAs of the previous slice (below): p = exp2(acc) (QK8: exp2(t * c - m * c))
    const float mc = QK8 ? m_run * c : 0.f;
    bf16x8 pf[4];
#define AT_EXP2(ks, e)                                                                                      \
  {                                                                                                         \
    float x0_ = ((ks) < 2 ? s0 : s1)[8 * ((ks) & 1) + (e)], x1_ = ((ks) < 2 ? s0 : s1)[8 * ((ks) & 1) + (e) + 1]; \
    x0_ = __builtin_amdgcn_exp2f(QK8 ? fmaf(x0_, c, -mc) : x0_);                                            \
    x1_ = __builtin_amdgcn_exp2f(QK8 ? fmaf(x1_, c, -mc) : x1_);                                            \
    ls += x0_ + x1_;                                                                                        \
    asm volatile("" : "+v"(ls)); /* keep the row-sum adds here, in the MFMA's shadow (hipcc sinks the chain to the tile's end) */ \
    pf[ks][e] = (__bf16)x0_;                                                                                \
    pf[ks][(e) + 1] = (__bf16)x1_;                                                                          \
  }
#define AT_EXP8(ks) AT_EXP2(ks, 0) AT_EXP2(ks, 2) AT_EXP2(ks, 4) AT_EXP2(ks, 6)
    if (!DMA) { AT_EXP8(0) AT_EXP8(1) AT_EXP8(2) AT_EXP8(3) }

    STAMP(2)
    // ---------------- next tile: registers -> other LDS stage, then fetch the tile after it
    if (!DMA && j + 1 < nt) {
      AT_LSTORE(cur ^ 1);
      if (j + 2 < nt) AT_GLOAD(j + 2);
    }

    // ---------------- O^T += V^T . P^T
    if (DMA) {
      // The transposed reads go through inline asm here: hipcc puts s_waitcnt vmcnt(0) in front of the builtin form
      // whenever an LDS-DMA is in flight (it cannot tell the two stages apart), which would serialise the prefetch.
      // Reads of key slice ks+1 are issued before the MFMAs of slice ks; the counted lgkmcnt wait carries the eight
      // registers it publishes as operands so that the MFMAs cannot be scheduled above it.
      const uint32_t vb = lds_base + cur * STAGE;
      s16x4 ta0, ta1, ta2, ta3, ta4, ta5, ta6, ta7, tb0, tb1, tb2, tb3, tb4, tb5, tb6, tb7;
#define AT_TR(dst, areg, ks) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vb + areg), "n"(VOFF + 4096 * (ks)))
#define AT_TR8(P, ks)                                                                                      \
  AT_TR(P##0, va0, ks); AT_TR(P##1, va1, ks); AT_TR(P##2, va2, ks); AT_TR(P##3, va3, ks);                  \
  AT_TR(P##4, va4, ks); AT_TR(P##5, va5, ks); AT_TR(P##6, va6, ks); AT_TR(P##7, va7, ks)
#define AT_WAIT8(P, n)                                                                                     \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(P##0), "+v"(P##1), "+v"(P##2), "+v"(P##3), "+v"(P##4), "+v"(P##5), "+v"(P##6), "+v"(P##7))
#define AT_PV4(P, ks)                                                                                      \
  o[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_join(P##0, P##1), pf[ks], o[0], 0, 0, 0);              \
  o[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_join(P##2, P##3), pf[ks], o[1], 0, 0, 0);              \
  o[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_join(P##4, P##5), pf[ks], o[2], 0, 0, 0);              \
  o[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_join(P##6, P##7), pf[ks], o[3], 0, 0, 0)
      // The exponentials of slice ks+1 are placed, two at a time, behind each of the four MFMAs of slice ks: an MFMA holds the
      // SIMD's issue port for 8 of its 32 cycles, so 2 x v_exp (8 cycles each) + their add / convert fit in its shadow and the
      // wave's own softmax overlaps its own matrix work (before: all 32 exponentials, then all 16 MFMAs).
#define AT_PV1(P, a, b, ks, db) o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_join(P##a, P##b), pf[ks], o[db], 0, 0, 0)
#define AT_FENCE() __builtin_amdgcn_sched_barrier(0)
      AT_TR8(ta, 0);
      AT_TR8(tb, 1);
      AT_EXP8(0)
      AT_FENCE();
      AT_WAIT8(ta, 8);
      AT_PV1(ta, 0, 1, 0, 0); AT_FENCE(); AT_EXP2(1, 0) AT_FENCE();
      AT_PV1(ta, 2, 3, 0, 1); AT_FENCE(); AT_EXP2(1, 2) AT_FENCE();
      AT_PV1(ta, 4, 5, 0, 2); AT_FENCE(); AT_EXP2(1, 4) AT_FENCE();
      AT_PV1(ta, 6, 7, 0, 3); AT_FENCE(); AT_EXP2(1, 6) AT_FENCE();
      AT_TR8(ta, 2);
      AT_WAIT8(tb, 8);
      AT_PV1(tb, 0, 1, 1, 0); AT_FENCE(); AT_EXP2(2, 0) AT_FENCE();
      AT_PV1(tb, 2, 3, 1, 1); AT_FENCE(); AT_EXP2(2, 2) AT_FENCE();
      AT_PV1(tb, 4, 5, 1, 2); AT_FENCE(); AT_EXP2(2, 4) AT_FENCE();
      AT_PV1(tb, 6, 7, 1, 3); AT_FENCE(); AT_EXP2(2, 6) AT_FENCE();
      AT_TR8(tb, 3);
      AT_WAIT8(ta, 8);
      AT_PV1(ta, 0, 1, 2, 0); AT_FENCE(); AT_EXP2(3, 0) AT_FENCE();
      AT_PV1(ta, 2, 3, 2, 1); AT_FENCE(); AT_EXP2(3, 2) AT_FENCE();
      AT_PV1(ta, 4, 5, 2, 2); AT_FENCE(); AT_EXP2(3, 4) AT_FENCE();
      AT_PV1(ta, 6, 7, 2, 3); AT_FENCE(); AT_EXP2(3, 6) AT_FENCE();
      AT_WAIT8(tb, 0);
      AT_PV4(tb, 3);
      l_run += ls;
#undef AT_PV1
#undef AT_FENCE
#undef AT_TR
#undef AT_TR8
#undef AT_WAIT8
#undef AT_PV4
    } else {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int db = 0; db < 4; ++db) {
          typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
          const char* a0 = sV + at_off(16 * ks + v_row, 4 * db + v_ch) + v_half;
          const char* a1 = sV + at_off(16 * ks + 8 + v_row, 4 * db + v_ch) + v_half;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a0);
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)a1);
          o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(at_join(lo, hi), pf[ks], o[db], 0, 0, 0);
        }
      }
      l_run += ls;
    }
#undef AT_EXP8
#undef AT_EXP2
    STAMP(3)
    if (DMA) {
      // tile j+1 must have landed; the eight instructions of tile j+2 (if issued; waves 4-7 issue none) may stay in flight
      if (j + 2 < jt1) AT_WAIT_TILE_AHEAD();
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      st3 = st3 == 2 ? 0 : st3 + 1;
      STAMP(4)
      __builtin_amdgcn_s_barrier();  // bare: __syncthreads() would drain vmcnt to 0 and with it the prefetch
    } else {
      __syncthreads();
    }
    STAMP(5)
  }
  }
#ifdef WANQ_ATTN_STAMP
  if (blockIdx.x == 7 && blockIdx.y == 3 && lane == 0) {
    for (int i = 0; i < 6; ++i) g_stamp[wave * 16 + i] = acc_[i];
  }
#endif

#ifdef WANQ_CLOCK_PROBE
  if (blockIdx.x == 70 && blockIdx.y == 3 && tid == 0) { g_clk[0] = clock64() - clk_c0; g_clk[1] = wall_clock64() - clk_w0; }
#endif
  // ---- epilogue: O[q, d] = O^T[d, q] / l ; lane holds d = 32 db + (r&3) + 8 (r>>2) + 4 fh of query fr
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (SPLIT) {  // unnormalised partials; attn_combine_kernel merges the splits
    const int qr = q0 + fr;
    if (qr < p.Lq) {
      float* po = p.part_o + ((int64_t)blockIdx.z * p.Lq + qr) * (p.H * AT_D) + head * AT_D + 4 * fh;
#pragma unroll
      for (int db = 0; db < 4; ++db)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<float4*>(po + 32 * db + 8 * g) = make_float4(o[db][4 * g], o[db][4 * g + 1], o[db][4 * g + 2], o[db][4 * g + 3]);
      if (fh == 0) {
        float* pm = p.part_ml + (((int64_t)blockIdx.z * p.Lq + qr) * p.H + head) * 2;
        // the merge kernel computes exp2((m - M) * p.c): hand it m in raw-score units (QK8: fold this query's delta_q in;
        // bf16 form: m_run already carries scale * log2(e))
        pm[0] = QK8 ? m_run * (c / p.c) : m_run / p.c;
        pm[1] = l_tot;
      }
    }
    return;
  }
  const float inv = 1.0f / l_tot;
  const int qr = q0 + fr;
  if (qr < p.Lq) {
    uint16_t* op = p.o + (int64_t)qr * p.o_stride + head * AT_D + 4 * fh;
#pragma unroll
    for (int db = 0; db < 4; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 b;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = (__bf16)(o[db][4 * g + e] * inv);
        *reinterpret_cast<bf16x4*>(op + 32 * db + 8 * g) = b;
      }
  }
}

// =====================================================================================================================
// 16x16x32 form of the bf16 kernel (LDS-DMA ring, no split-KV): the same structure on v_mfma_f32_16x16x32_bf16, whose
// streams hold a ~10 % higher clock on this part (tools/probes/mfma_shape_clock.hip).  Per wave 32 queries = two blocks
// nq of 16; per 64-key tile four key blocks kb of 16.
//   S^T block (kb, nq) = K_kb . Q_nq^T : A = K rows (ds_read_b128: lane (r = lane & 15, g = lane >> 4) reads LDS row
//       16 kb + kappa(r), 16-B chunk 4 s + g of d-slice s), B = Q fragment in registers.  Accumulator element e of lane
//       (n, g) is query 16 nq + n against key 16 kb + 4 pi(g) + e, where kappa(4 t + e) = 4 pi(t) + e, pi = (0, 2, 1, 3):
//       the swap of the two middle row quads makes the transposed V reads below conflict-free.
//   O^T block (db, nq) += V^T_(db) . P^T_nq : B = two S blocks (2 ks, 2 ks + 1) of the lane converted in place (k index
//       8 g + j <-> key 32 ks + 16 (j >> 2) + 4 pi(g) + (j & 3)); A = V^T gathered by two ds_read_b64_tr_b16 whose 16-lane
//       group g reads rows 32 ks + 16 jh + 4 pi(g) + q: the two groups of a 32-lane half sit 8 rows apart (conflict-free).
//   A query's statistics live in four lanes (n, n + 16, n + 32, n + 48): the running maximum is only reduced across them
//   when the lazy rescale fires (wave vote on the lane-local maxima); row sums are reduced once, in the epilogue.
// The K tile's LDS image uses the chunk swizzle row & 15 (conflict-free for this operand's b128 lane groups; the image of
// the 32x32 kernel is 2-way for them); the V image is the one above.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// The tile is bound by the SIMD's vector ISSUE port, not by the matrix pipe: a 16x16x32 MFMA holds the port for 8 of its 16
// cycles, and the two waves of a SIMD share it -- per wave and 64-key tile 64 MFMAs (512 issue cycles) + 32 v_exp (256) + the
// softmax bookkeeping.  Two pieces of that bookkeeping are therefore moved / shortened (round 3):
//   WANQ_ATTN_LSUM_MFMA  the row sums l = sum_k P are accumulated ON THE MATRIX CORES: one MFMA per (key slice, query block)
//                        with an all-ones A operand and the same B operand (P as bf16) as the P.V MFMAs -- 4 MFMAs (32 issue
//                        cycles) replace 32 dependent v_add_f32 (128) and the s_nop hipcc puts between a v_exp and the add that
//                        consumes it (16 per tile).  Every row of the 16x16 result is the row sum, so every lane holds its
//                        query's total (no cross-lane reduction in the epilogue), and l now sums exactly the bf16-rounded P
//                        that P.V uses.
//   WANQ_ATTN_MAX3       the lazy-rescale vote needs the maximum of the lane's 32 scores: 16 v_max3_f32 written as asm (the
//                        builtin form costs 21: hipcc canonicalises MFMA outputs with v_max_f32 x, x first).
#ifndef WANQ_ATTN_LSUM_MFMA
#define WANQ_ATTN_LSUM_MFMA 1
#endif
#ifndef WANQ_ATTN_MAX3
#define WANQ_ATTN_MAX3 0
#endif
#ifndef WANQ_ATTN_KASM
#define WANQ_ATTN_KASM 4
#endif
// The wave index as a SCALAR (readfirstlane of threadIdx.x >> 6): hipcc cannot prove it uniform, and carried in a vector register it
// made every LDS-DMA destination a v_readfirstlane + M0 write (8 per tile and DMA wave) and the dma_wave test an EXEC-mask branch; as a
// scalar the issue of a tile is s_add / s_mov m0 / global_load_lds only.  Round 5: 4.806 -> 4.751 ms at 32760 x 32760 x 12 heads (1.012x),
// cross-attention 1.026x, outputs bit-identical (profiles/r05_uw_attn_uniform_wave_ab.txt).  0 = the vector form, for A/B builds.
#ifndef WANQ_ATTN_UNIFORM_WAVE
#define WANQ_ATTN_UNIFORM_WAVE 1
#endif
#ifndef WANQ_ATTN_DMA_LATE
#define WANQ_ATTN_DMA_LATE 0
#endif
// timing-only ablations (wrong results; tools/probes/README.md): skip the lazy-rescale vote after the first tile / replace v_exp
#ifndef WANQ_ATTN_NW4_KEYS_DEFAULT  // key count up to which the plain bf16 kernel runs in its 4-wave form
#define WANQ_ATTN_NW4_KEYS_DEFAULT 1024
#endif
#ifndef WANQ_ABL_NODMA  // timing ablation (wrong results): no K / V tile is fetched inside the tile loop
#define WANQ_ABL_NODMA 0
#endif
#ifndef WANQ_ABL_NOMAX
#define WANQ_ABL_NOMAX 0
#endif
#ifndef WANQ_ABL_NOEXP
#define WANQ_ABL_NOEXP 0
#endif
#if (WANQ_ABL_NODMA || WANQ_ABL_NOMAX || WANQ_ABL_NOEXP) && !defined(WANQ_ALLOW_ABLATIONS)
#error "WANQ_ABL_* build deliberately wrong kernels (timing ablations): add -DWANQ_ALLOW_ABLATIONS, never in build.py's library"
#endif

// NW = waves per workgroup: 8 (256 queries, three ring stages, tiles requested two ahead), or 4 (128 queries per workgroup,
// two ring stages = 64 KiB, so that TWO workgroups share a CU: SIMD partners then belong to different workgroups and are not
// coupled by the per-tile barrier; one's prologue / epilogue runs under the other's tiles).
template <bool SPLIT, bool QK8, int NW = 8>
__global__ __launch_bounds__(64 * NW, 2) void attn_fwd16_kernel(const AttnParams p) {
  constexpr bool LSUM = WANQ_ATTN_LSUM_MFMA != 0 && !(SPLIT && !QK8);  // (the bf16 split-KV form has no 8 registers to spare: it spills)
  constexpr int STAGE = QK8 ? AT_STAGE8 : AT_STAGE;
  constexpr int NST = NW == 8 ? 3 : 2, AHEAD = NST - 1;  // ring stages, prefetch distance in tiles
  static_assert(NW == 8 || (NW == 4 && !SPLIT && !QK8), "the 4-wave form exists for the plain bf16 kernel only");
  constexpr int VOFF = QK8 ? AT_K8 : AT_TILE;  // byte offset of the V tile inside a stage
  typedef int v4i __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = WANQ_ATTN_UNIFORM_WAVE ? __builtin_amdgcn_readfirstlane(tid >> 6) : (tid >> 6);
  const int n16 = lane & 15, g4 = lane >> 4;
  int head = blockIdx.y, qblk = blockIdx.x;
  if (!SPLIT) {  // XCD-aware (head, query block) map, as in attn_fwd_kernel
    const int nqb = gridDim.x, T = nqb * (int)gridDim.y, L = (int)blockIdx.x + nqb * (int)blockIdx.y;
    const int xq = T >> 3, xr = T & 7, xcd = L & 7;
    const int i = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (L >> 3);
    head = i / nqb;
    qblk = i - head * nqb;
  }
  const int q0 = qblk * (AT_QW * NW) + wave * AT_QW;
  const int nt = (p.Lk + AT_KB - 1) / AT_KB;
  // key tiles of this workgroup: all of them, or one contiguous share under split-KV (the host makes every share non-empty)
  const int jt0 = SPLIT ? (int)blockIdx.z * p.tiles_per_split : 0;
  const int jt1 = SPLIT ? (jt0 + p.tiles_per_split < nt ? jt0 + p.tiles_per_split : nt) : nt;
  const float c = p.c;

  // ---- Q fragments: query q0 + 16 nq + n16, d = 32 s + 8 g4 + [0, 8), pre-scaled by softmax scale * log2(e)
  // QK8: int8 codes, d = 64 s + 16 g4 + [0, 16); the query's scale delta_q rides in the exp2 coefficient c2[nq]
  bf16x8 qf[QK8 ? 1 : 2][QK8 ? 1 : 4];
  v4i qf8[QK8 ? 2 : 1][QK8 ? 2 : 1];
  float c2[2] = {c, c};
#pragma unroll
  for (int nq = 0; nq < 2; ++nq) {
    int qr = q0 + 16 * nq + n16;
    if (qr >= p.Lq) qr = p.Lq - 1;
    if (QK8) {
      const int8_t* qp8 = p.q8 + (int64_t)qr * p.q8_stride + head * AT_D + 16 * g4;
#pragma unroll
      for (int s = 0; s < 2; ++s) qf8[QK8 ? nq : 0][QK8 ? s : 0] = *reinterpret_cast<const v4i*>(qp8 + 64 * s);
      c2[nq] = c * p.q_scale[(int64_t)head * p.qs_stride + qr];  // score = dot * delta_k * delta_q * softmax scale
    } else {
      const uint16_t* qp = p.q + (int64_t)qr * p.q_stride + head * AT_D + 8 * g4;
#pragma unroll
      for (int s = 0; s < 4; ++s) qf[QK8 ? 0 : nq][QK8 ? 0 : s] = *reinterpret_cast<const bf16x8*>(qp + 32 * s);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[QK8 ? 0 : nq][QK8 ? 0 : s][e] = (__bf16)((float)qf[QK8 ? 0 : nq][QK8 ? 0 : s][e] * c);
    }
  }

  // ---- LDS-DMA: waves 0-3 issue the whole tile, 8 pieces each (rows 16 w + 4 i + d_r of the K tile and of the V tile)
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  const bool dma_wave = wave < 4;
  const int d_r = lane >> 4;
#define A16_DOFF_V(stride, i) ((uint32_t)((16 * (wave & 3) + 4 * (i) + d_r) * (int)(stride) + head * AT_D + ((((lane & 15) ^ (d_r << 2)) ^ (i)) << 3)) * 2u)
#define A16_DOFF_K(stride, i) ((uint32_t)((16 * (wave & 3) + 4 * (i) + d_r) * (int)(stride) + head * AT_D + (((lane & 15) ^ (4 * (i) + d_r)) << 3)) * 2u)
  uint32_t d_k0 = QK8 ? 0u : A16_DOFF_K(p.k_stride, 0), d_k1 = QK8 ? 0u : A16_DOFF_K(p.k_stride, 1), d_k2 = QK8 ? 0u : A16_DOFF_K(p.k_stride, 2),
           d_k3 = QK8 ? 0u : A16_DOFF_K(p.k_stride, 3);
  // QK8: the K tile is int8 (8 rows x 128 B per 1-KiB piece, 2 pieces per DMA wave: rows 16 w + 8 i + (lane >> 3), physical chunk
  // lane & 7 holds logical chunk (lane & 7) ^ ((row >> 1) & 7)); waves 4 and 5 fetch the tile's 64 key scales / constants
  const int d8_r = lane >> 3;
#define A16_D8OFF(i) ((uint32_t)((16 * (wave & 3) + 8 * (i) + d8_r) * (int)p.k8_stride + head * AT_D + ((((lane & 7) ^ (((8 * (i) + d8_r) >> 1) & 7))) << 4)))
  const uint32_t d8_k0 = QK8 ? A16_D8OFF(0) : 0u, d8_k1 = QK8 ? A16_D8OFF(1) : 0u;
#undef A16_D8OFF
  uint32_t d_v0 = A16_DOFF_V(p.v_stride, 0), d_v1 = A16_DOFF_V(p.v_stride, 1), d_v2 = A16_DOFF_V(p.v_stride, 2), d_v3 = A16_DOFF_V(p.v_stride, 3);
#undef A16_DOFF_V
#undef A16_DOFF_K
#define A16_DMA_F(base, off, tilebyte, i) \
  __builtin_amdgcn_global_load_lds((glb_void*)((base) + (off)), (lds_void*)(sK_ + (tilebyte) + 1024 * (i)), 16, 0, 0);
#define A16_DMA_S(base, stride, tilebyte, i, j, kswz)                                                           \
  {                                                                                                             \
    int kr_ = (j) * AT_KB + 16 * (wave & 3) + 4 * (i) + d_r;                                                    \
    kr_ = kr_ < p.Lk ? kr_ : p.Lk - 1;                                                                          \
    const int col_ = head * AT_D + ((kswz ? ((lane & 15) ^ (4 * (i) + d_r)) : (((lane & 15) ^ (d_r << 2)) ^ (i))) << 3); \
    __builtin_amdgcn_global_load_lds((glb_void*)((base) + (int64_t)kr_ * (stride) + col_), (lds_void*)(sK_ + (tilebyte) + 1024 * (i)), 16, 0, 0); \
  }
#define A16_DMA8(j, stage)                                                                                      \
  do {                                                                                                          \
    char* st_ = smem + (stage) * STAGE;                                                                         \
    if (dma_wave) {                                                                                             \
      char* sK8_ = st_ + (wave & 3) * 2048;                                                                     \
      char* sK_ = st_ + (wave & 3) * 4096;                                                                      \
      if (((j) + 1) * AT_KB <= p.Lk) {                                                                          \
        const char* kt_ = reinterpret_cast<const char*>(p.k8) + (int64_t)(j) * AT_KB * p.k8_stride;             \
        const char* vt_ = reinterpret_cast<const char*>(p.v) + (int64_t)(j) * (AT_KB * 2) * p.v_stride;         \
        __builtin_amdgcn_global_load_lds((glb_void*)(kt_ + d8_k0), (lds_void*)(sK8_), 16, 0, 0);                \
        __builtin_amdgcn_global_load_lds((glb_void*)(kt_ + d8_k1), (lds_void*)(sK8_ + 1024), 16, 0, 0);         \
        A16_DMA_F(vt_, d_v0, AT_K8, 0) A16_DMA_F(vt_, d_v1, AT_K8, 1) A16_DMA_F(vt_, d_v2, AT_K8, 2) A16_DMA_F(vt_, d_v3, AT_K8, 3) \
      } else {                                                                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                      \
          int kr_ = (j) * AT_KB + 16 * (wave & 3) + 8 * i_ + d8_r;                                              \
          kr_ = kr_ < p.Lk ? kr_ : p.Lk - 1;                                                                    \
          const int col_ = head * AT_D + ((((lane & 7) ^ (((8 * i_ + d8_r) >> 1) & 7))) << 4);                  \
          __builtin_amdgcn_global_load_lds((glb_void*)(p.k8 + (int64_t)kr_ * p.k8_stride + col_), (lds_void*)(sK8_ + 1024 * i_), 16, 0, 0); \
        }                                                                                                       \
        A16_DMA_S(p.v, p.v_stride, AT_K8, 0, j, false) A16_DMA_S(p.v, p.v_stride, AT_K8, 1, j, false) A16_DMA_S(p.v, p.v_stride, AT_K8, 2, j, false) A16_DMA_S(p.v, p.v_stride, AT_K8, 3, j, false) \
      }                                                                                                         \
    } else if (wave < 6) { /* scale plane (wave 4) / constant plane (wave 5): 64 floats, one dword per lane */  \
      const float* sp_ = p.k_scale + (int64_t)(wave - 4) * p.H * p.ks_stride + (int64_t)head * p.ks_stride + (int64_t)(j) * AT_KB + lane; \
      __builtin_amdgcn_global_load_lds((glb_void*)sp_, (lds_void*)(st_ + AT_SC8 + (wave - 4) * 256), 4, 0, 0);  \
    }                                                                                                           \
  } while (0)
#define A16_DMA(j, stage)                                                                                       \
  do {                                                                                                          \
    if (QK8) { A16_DMA8(j, stage); break; }                                                                     \
    if (dma_wave) {                                                                                             \
      char* sK_ = smem + (stage) * AT_STAGE + (wave & 3) * 4096;                                                \
      if (((j) + 1) * AT_KB <= p.Lk) {                                                                          \
        const char* kt_ = reinterpret_cast<const char*>(p.k) + (int64_t)(j) * (AT_KB * 2) * p.k_stride;         \
        const char* vt_ = reinterpret_cast<const char*>(p.v) + (int64_t)(j) * (AT_KB * 2) * p.v_stride;         \
        A16_DMA_F(kt_, d_k0, 0, 0) A16_DMA_F(kt_, d_k1, 0, 1) A16_DMA_F(kt_, d_k2, 0, 2) A16_DMA_F(kt_, d_k3, 0, 3) \
        A16_DMA_F(vt_, d_v0, AT_TILE, 0) A16_DMA_F(vt_, d_v1, AT_TILE, 1) A16_DMA_F(vt_, d_v2, AT_TILE, 2) A16_DMA_F(vt_, d_v3, AT_TILE, 3) \
      } else {                                                                                                  \
        A16_DMA_S(p.k, p.k_stride, 0, 0, j, true) A16_DMA_S(p.k, p.k_stride, 0, 1, j, true) A16_DMA_S(p.k, p.k_stride, 0, 2, j, true) A16_DMA_S(p.k, p.k_stride, 0, 3, j, true) \
        A16_DMA_S(p.v, p.v_stride, AT_TILE, 0, j, false) A16_DMA_S(p.v, p.v_stride, AT_TILE, 1, j, false) A16_DMA_S(p.v, p.v_stride, AT_TILE, 2, j, false) A16_DMA_S(p.v, p.v_stride, AT_TILE, 3, j, false) \
      }                                                                                                         \
    }                                                                                                           \
  } while (0)

  // ---- fragment read offsets
  const int kap = 4 * (2 * ((n16 >> 2) & 1) + ((n16 >> 3) & 1)) + (n16 & 3);  // kappa(n16)
  uint32_t koff0 = kap * 256 + (((0 + g4) ^ kap) << 4), koff1 = kap * 256 + (((4 + g4) ^ kap) << 4);
  uint32_t koff2 = kap * 256 + (((8 + g4) ^ kap) << 4), koff3 = kap * 256 + (((12 + g4) ^ kap) << 4);
  // QK8: 128-B rows, chunk swizzle (row >> 1) & 7 (at_off8): row 16 kb + kappa, chunk 4 s + g4
  const uint32_t k8off0 = kap * 128 + (((0 + g4) ^ ((kap >> 1) & 7)) << 4), k8off1 = kap * 128 + (((4 + g4) ^ ((kap >> 1) & 7)) << 4);
  const int pg = 2 * (g4 & 1) + (g4 >> 1), tq = (lane >> 2) & 3, tp = lane & 3;
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;
#define A16_VA(db) (uint32_t)(at_off(4 * pg + tq, 2 * (db) + (tp >> 1)) + 8 * (tp & 1))
  const uint32_t va0 = A16_VA(0), va1 = A16_VA(1), va2 = A16_VA(2), va3 = A16_VA(3);
  const uint32_t va4 = A16_VA(4), va5 = A16_VA(5), va6 = A16_VA(6), va7 = A16_VA(7);
#undef A16_VA

  f32x4 o[8][2];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int nq = 0; nq < 2; ++nq)
#pragma unroll
      for (int r = 0; r < 4; ++r) o[i][nq][r] = 0.f;
  float m_run[2] = {QK8 ? -INFINITY : 0.f, QK8 ? -INFINITY : 0.f}, l_run[2] = {0.f, 0.f};
  f32x4 lacc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};  // LSUM: row sums as MFMA accumulators (all four elements equal)
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  asm volatile("" : "+v"(ones));  // one register quad for the whole kernel, not rematerialised per use
  f32x4 sinit[2];
#pragma unroll
  for (int nq = 0; nq < 2; ++nq)
#pragma unroll
    for (int r = 0; r < 4; ++r) sinit[nq][r] = 0.f;

  // LDS-DMA instructions a wave issues per tile (= what may stay in flight behind a counted wait): 8 for the DMA waves of the
  // bf16 form (waves 4-7 issue none: any count passes); QK8: 6 for waves 0-3 (2 K + 4 V pieces), 1 for waves 4-5 (scales)
#define A16_WAIT_TILE_AHEAD()                                                               \
  do {                                                                                      \
    if (!QK8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                              \
    else if (wave < 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                     \
    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                   \
  } while (0)
  A16_DMA(jt0, 0);
  if (AHEAD == 2 && jt0 + 1 < jt1) {
    A16_DMA(jt0 + 1, 1);
    A16_WAIT_TILE_AHEAD();
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (__builtin_amdgcn_readfirstlane(threadIdx.x) >= 256) __builtin_amdgcn_s_setprio(1);

  for (int j0 = jt0; j0 < jt1; j0 += NST) {
#pragma unroll
  for (int u = 0; u < NST; ++u) {
    const int j = j0 + u;
    if (j >= jt1) break;
    const char* sK = smem + u * STAGE;
    asm volatile("" : "+v"(d_k0), "+v"(d_k1), "+v"(d_k2), "+v"(d_k3), "+v"(d_v0), "+v"(d_v1), "+v"(d_v2), "+v"(d_v3));
    if (!((WANQ_ATTN_DMA_LATE != 0) && (WANQ_ATTN_KASM != 0) && !QK8) && !WANQ_ABL_NODMA && j + AHEAD < jt1) A16_DMA(j + AHEAD, (u + AHEAD) % NST);

    // ---------------- S^T blocks: fragment i = 4 kb + s read four ahead of its two MFMAs
    f32x4 sacc[4][2];
    if (!QK8 && WANQ_ATTN_KASM) {
      // K fragments by asm reads with COUNTED waits: fragment i is waited for with the KA-1 younger reads still in flight
      // (hipcc merges the waits of the builtin form into three s_waitcnt lgkmcnt(0) per tile, each of which drains the
      // lookahead it was given)
      constexpr int KA = WANQ_ATTN_KASM;  // fragments in flight
      bf16x8 kf[16];
      const uint32_t kbase = lds_base + u * STAGE;
      const uint32_t ka0 = kbase + koff0, ka1 = kbase + koff1, ka2 = kbase + koff2, ka3 = kbase + koff3;
#define A16_KR(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[i]) : "v"(((i) & 3) == 0 ? ka0 : ((i) & 3) == 1 ? ka1 : ((i) & 3) == 2 ? ka2 : ka3), "n"(((i) >> 2) * 4096))
#pragma unroll
      for (int i = 0; i < KA; ++i) { A16_KR(i); }
      if (WANQ_ATTN_DMA_LATE && !WANQ_ABL_NODMA && j + AHEAD < jt1) A16_DMA(j + AHEAD, (u + AHEAD) % NST);  // the first fragments fly under the DMA issue
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i + KA < 16) { A16_KR(i + KA); }
        const int left = 15 - i < KA ? 15 - i : KA;  // reads younger than fragment i
        if (left >= 6) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(kf[i]));
        else if (left == 5) asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(kf[i]));
        else if (left == 4) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(kf[i]));
        else if (left == 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(kf[i]));
        else if (left == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(kf[i]));
        else if (left == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(kf[i]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(kf[i]));
        __builtin_amdgcn_sched_barrier(0);
        const int kb = i >> 2, s = i & 3;
        sacc[kb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i], qf[0][QK8 ? 0 : s], s == 0 ? sinit[0] : sacc[kb][0], 0, 0, 0);
        sacc[kb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i], qf[QK8 ? 0 : 1][QK8 ? 0 : s], s == 0 ? sinit[1] : sacc[kb][1], 0, 0, 0);
      }
#undef A16_KR
    } else if (!QK8) {
      bf16x8 kf[16];
#define A16_KF(i) kf[i] = *reinterpret_cast<const bf16x8*>(sK + (((i) & 3) == 0 ? koff0 : ((i) & 3) == 1 ? koff1 : ((i) & 3) == 2 ? koff2 : koff3) + ((i) >> 2) * 4096)
      A16_KF(0); A16_KF(1); A16_KF(2); A16_KF(3);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (i + 4 < 16) { A16_KF(i + 4); }
        __builtin_amdgcn_sched_barrier(0);
        const int kb = i >> 2, s = i & 3;
        sacc[kb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i], qf[0][QK8 ? 0 : s], s == 0 ? sinit[0] : sacc[kb][0], 0, 0, 0);
        sacc[kb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[i], qf[QK8 ? 0 : 1][QK8 ? 0 : s], s == 0 ? sinit[1] : sacc[kb][1], 0, 0, 0);
      }
#undef A16_KF
    } else {
      // int8: fragment i = 2 kb + s (d-slices of 64) on v_mfma_i32_16x16x64_i8; the accumulators START at the bit pattern of
      // 12582912.0f (see attn_fwd_kernel), so their bits read as a float are 12582912 + dot, and one fma per score with the
      // key's scale and its precomputed constant gives t = dot * delta_k
      v4i ia[4][2], kf8[8];
      const v4i mg = {0x4B400000, 0x4B400000, 0x4B400000, 0x4B400000};
      f32x4 sk[4], sb[4];  // key scales / constants of the lane's four keys per key block: keys 16 kb + 4 pi(g) + e
      const uint32_t sa = lds_base + u * STAGE + AT_SC8 + 16 * pg;
#define A16_SC(dst, off) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(sa), "n"(off))
      A16_SC(sk[0], 0); A16_SC(sk[1], 64); A16_SC(sk[2], 128); A16_SC(sk[3], 192);
      A16_SC(sb[0], 256); A16_SC(sb[1], 320); A16_SC(sb[2], 384); A16_SC(sb[3], 448);
#define A16_KF8(i) kf8[i] = *reinterpret_cast<const v4i*>(sK + (((i) & 1) == 0 ? k8off0 : k8off1) + ((i) >> 1) * 2048)
      A16_KF8(0); A16_KF8(1); A16_KF8(2); A16_KF8(3);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (i + 4 < 8) { A16_KF8(i + 4); }
        __builtin_amdgcn_sched_barrier(0);
        const int kb = i >> 1, s = i & 1;
        ia[kb][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf8[i], qf8[0][QK8 ? s : 0], s == 0 ? mg : ia[kb][0], 0, 0, 0);
        ia[kb][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf8[i], qf8[QK8 ? 1 : 0][QK8 ? s : 0], s == 0 ? mg : ia[kb][1], 0, 0, 0);
      }
#undef A16_KF8
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(sk[0]), "+v"(sk[1]), "+v"(sk[2]), "+v"(sk[3]), "+v"(sb[0]), "+v"(sb[1]), "+v"(sb[2]), "+v"(sb[3]));
#undef A16_SC
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int nq = 0; nq < 2; ++nq)
#pragma unroll
          for (int e = 0; e < 4; ++e) sacc[kb][nq][e] = fmaf(__int_as_float(ia[kb][nq][e]), sk[kb][e], sb[kb][e]);
    }
    if (j == nt - 1 && (p.Lk & (AT_KB - 1))) {  // ragged last tile: keys >= Lk get -inf
      asm volatile("" ::: "memory");
      const int kbase = j * AT_KB + 4 * pg;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kbase + 16 * kb + e >= p.Lk) { sacc[kb][0][e] = -INFINITY; sacc[kb][1][e] = -INFINITY; }
    }

    // ---------------- online softmax: lane-local maxima, cross-lane only when the reference moves
    // lane-local maxima of the tile's scores for the lazy-rescale vote.  MAX3: two asm statements of eight v_max3_f32 (one
    // statement per chain: hipcc pads every asm statement with an s_nop, and canonicalises the inputs of a builtin fmaxf);
    // in the bf16 form mx1 then covers BOTH query blocks (the vote needs nothing finer; the rare branch recomputes block 1's)
    float mx0, mx1;
    if (WANQ_ATTN_MAX3) {
#define A16_M8(dst, seed, b)                                                                                                    \
  asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max3_f32 %0, %0, %8, %9\n\t"    \
      "v_max3_f32 %0, %0, %10, %11\n\tv_max3_f32 %0, %0, %12, %13\n\tv_max3_f32 %0, %0, %14, %15\n\tv_max3_f32 %0, %0, %16, %17" \
      : "=&v"(dst)                                                                                                              \
      : "v"(seed), "v"(sacc[0][b][0]), "v"(sacc[0][b][1]), "v"(sacc[0][b][2]), "v"(sacc[0][b][3]), "v"(sacc[1][b][0]),          \
        "v"(sacc[1][b][1]), "v"(sacc[1][b][2]), "v"(sacc[1][b][3]), "v"(sacc[2][b][0]), "v"(sacc[2][b][1]), "v"(sacc[2][b][2]), \
        "v"(sacc[2][b][3]), "v"(sacc[3][b][0]), "v"(sacc[3][b][1]), "v"(sacc[3][b][2]), "v"(sacc[3][b][3]))
      A16_M8(mx0, sacc[0][0][0], 0);  // (the seed repeats a score of the chain)
      if (QK8) { A16_M8(mx1, sacc[0][1][0], 1); }  // int8 form: the two blocks' scores carry different units (delta_q in c2)
      else { A16_M8(mx1, mx0, 1); }
#undef A16_M8
    } else {
      mx0 = sacc[0][0][0], mx1 = sacc[0][1][0];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e) { mx0 = fmaxf(mx0, sacc[kb][0][e]); mx1 = fmaxf(mx1, sacc[kb][1][e]); }
    }
    const bool first = (j == jt0);
    if (QK8) {
      // scores are dot * delta_k here; c2 = delta_q * scale * log2(e) is per lane and query block.  Running maximum in score
      // units, rescaled lazily (first tile: m_run = -inf, so the vote fires and alpha = 0 on the zero accumulators)
      if (__any(fmaxf((mx0 - m_run[0]) * c2[0], (mx1 - m_run[1]) * c2[1]) > 6.0f)) {
        asm volatile("" ::: "memory");
        float mx[2] = {mx0, mx1};
#pragma unroll
        for (int nq = 0; nq < 2; ++nq) {
          float m = mx[nq];
          m = fmaxf(m, __shfl_xor(m, 16, 64));
          m = fmaxf(m, __shfl_xor(m, 32, 64));
          const float m_new = fmaxf(m_run[nq], m);
          const float alpha = __builtin_amdgcn_exp2f((m_run[nq] - m_new) * c2[nq]);
          m_run[nq] = m_new;
          l_run[nq] *= alpha;
#pragma unroll
          for (int r = 0; r < 4; ++r) lacc[nq][r] *= alpha;
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[i][nq][r] *= alpha;
        }
      }
    } else if (first || (!WANQ_ABL_NOMAX && __any((WANQ_ATTN_MAX3 ? mx1 : fmaxf(mx0, mx1)) > 6.0f))) {
      asm volatile("" ::: "memory");  // keep this a branch
      float mx[2] = {mx0, mx1};
      if (WANQ_ATTN_MAX3) {  // mx1 covers both query blocks: the block's own maximum, here only
        mx[1] = sacc[0][1][0];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int e = 0; e < 4; ++e) mx[1] = fmaxf(mx[1], sacc[kb][1][e]);
      }
#pragma unroll
      for (int nq = 0; nq < 2; ++nq) {
        float m = mx[nq];
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float delta = first ? m : fmaxf(m, 0.f);
        if (!first) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
          l_run[nq] *= alpha;
#pragma unroll
          for (int r = 0; r < 4; ++r) lacc[nq][r] *= alpha;
#pragma unroll
          for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[i][nq][r] *= alpha;
        }
        m_run[nq] = first ? delta : m_run[nq] + delta;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int e = 0; e < 4; ++e) sacc[kb][nq][e] -= delta;
#pragma unroll
        for (int r = 0; r < 4; ++r) sinit[nq][r] = -m_run[nq];
      }
    }
    // P of key slice ks (32 keys) for query block nq: element j = 4 (kb & 1) + e of S block kb = 2 ks + (j >> 2)
    float ls0 = 0.f, ls1 = 0.f;
    const float mc[2] = {QK8 ? m_run[0] * c2[0] : 0.f, QK8 ? m_run[1] * c2[1] : 0.f};
    bf16x8 pf[2][2];
#define A16_EXP(kb, nq, e)                                                     \
  {                                                                            \
    const float x_ = WANQ_ABL_NOEXP ? sacc[kb][nq][e] * 0.001f : __builtin_amdgcn_exp2f(QK8 ? fmaf(sacc[kb][nq][e], c2[nq], -mc[nq]) : sacc[kb][nq][e]); \
    if (!LSUM) {                                                               \
      if (nq == 0) { ls0 += x_; asm volatile("" : "+v"(ls0)); }                \
      else { ls1 += x_; asm volatile("" : "+v"(ls1)); }                        \
    }                                                                          \
    pf[(kb) >> 1][nq][4 * ((kb) & 1) + (e)] = (__bf16)x_;                      \
  }
#define A16_EXP4(kb, nq) A16_EXP(kb, nq, 0) A16_EXP(kb, nq, 1) A16_EXP(kb, nq, 2) A16_EXP(kb, nq, 3)

    // ---------------- O^T += V^T . P^T
    const uint32_t vb = lds_base + u * STAGE;
    s16x4 ta0, ta1, ta2, ta3, ta4, ta5, ta6, ta7, tb0, tb1, tb2, tb3, tb4, tb5, tb6, tb7;
#define A16_TR(dst, areg, ks, jh) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vb + areg), "n"(VOFF + 8192 * (ks) + 4096 * (jh)))
#define A16_TR8(P, ks, a0_, a1_, a2_, a3_)                                                                      \
  A16_TR(P##0, a0_, ks, 0); A16_TR(P##1, a0_, ks, 1); A16_TR(P##2, a1_, ks, 0); A16_TR(P##3, a1_, ks, 1);     \
  A16_TR(P##4, a2_, ks, 0); A16_TR(P##5, a2_, ks, 1); A16_TR(P##6, a3_, ks, 0); A16_TR(P##7, a3_, ks, 1)
#define A16_WAIT8(P, n)                                                                                    \
  asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(P##0), "+v"(P##1), "+v"(P##2), "+v"(P##3), "+v"(P##4), "+v"(P##5), "+v"(P##6), "+v"(P##7))
#define A16_PV(P, a, b, ks, db, nq) o[db][nq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(at_join(P##a, P##b), pf[ks][nq], o[db][nq], 0, 0, 0)
#define A16_F() __builtin_amdgcn_sched_barrier(0)
#define A16_LS(ks, nq) if (LSUM) lacc[nq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pf[ks][nq], lacc[nq], 0, 0, 0)
    A16_TR8(ta, 0, va0, va1, va2, va3);
    A16_TR8(tb, 0, va4, va5, va6, va7);
    A16_EXP4(0, 0) A16_EXP4(1, 0) A16_EXP4(0, 1) A16_EXP4(1, 1)
    A16_F();
    A16_WAIT8(ta, 8);
    // slice 0, d blocks 0-3: one exponential of slice 1 behind every MFMA
    A16_PV(ta, 0, 1, 0, 0, 0); A16_F(); A16_EXP(2, 0, 0) A16_F(); A16_PV(ta, 0, 1, 0, 0, 1); A16_F(); A16_EXP(2, 0, 1) A16_F();
    A16_PV(ta, 2, 3, 0, 1, 0); A16_F(); A16_EXP(2, 0, 2) A16_F(); A16_PV(ta, 2, 3, 0, 1, 1); A16_F(); A16_EXP(2, 0, 3) A16_F();
    A16_PV(ta, 4, 5, 0, 2, 0); A16_F(); A16_EXP(3, 0, 0) A16_F(); A16_PV(ta, 4, 5, 0, 2, 1); A16_F(); A16_EXP(3, 0, 1) A16_F();
    A16_PV(ta, 6, 7, 0, 3, 0); A16_F(); A16_EXP(3, 0, 2) A16_F(); A16_PV(ta, 6, 7, 0, 3, 1); A16_F(); A16_EXP(3, 0, 3) A16_F();
    A16_LS(0, 0); A16_LS(0, 1);
    A16_TR8(ta, 1, va0, va1, va2, va3);
    A16_WAIT8(tb, 8);
    A16_PV(tb, 0, 1, 0, 4, 0); A16_F(); A16_EXP(2, 1, 0) A16_F(); A16_PV(tb, 0, 1, 0, 4, 1); A16_F(); A16_EXP(2, 1, 1) A16_F();
    A16_PV(tb, 2, 3, 0, 5, 0); A16_F(); A16_EXP(2, 1, 2) A16_F(); A16_PV(tb, 2, 3, 0, 5, 1); A16_F(); A16_EXP(2, 1, 3) A16_F();
    A16_PV(tb, 4, 5, 0, 6, 0); A16_F(); A16_EXP(3, 1, 0) A16_F(); A16_PV(tb, 4, 5, 0, 6, 1); A16_F(); A16_EXP(3, 1, 1) A16_F();
    A16_PV(tb, 6, 7, 0, 7, 0); A16_F(); A16_EXP(3, 1, 2) A16_F(); A16_PV(tb, 6, 7, 0, 7, 1); A16_F(); A16_EXP(3, 1, 3) A16_F();
    A16_TR8(tb, 1, va4, va5, va6, va7);
    A16_WAIT8(ta, 8);
    A16_PV(ta, 0, 1, 1, 0, 0); A16_PV(ta, 0, 1, 1, 0, 1); A16_PV(ta, 2, 3, 1, 1, 0); A16_PV(ta, 2, 3, 1, 1, 1);
    A16_PV(ta, 4, 5, 1, 2, 0); A16_PV(ta, 4, 5, 1, 2, 1); A16_PV(ta, 6, 7, 1, 3, 0); A16_PV(ta, 6, 7, 1, 3, 1);
    A16_LS(1, 0); A16_LS(1, 1);
    A16_WAIT8(tb, 0);
    A16_PV(tb, 0, 1, 1, 4, 0); A16_PV(tb, 0, 1, 1, 4, 1); A16_PV(tb, 2, 3, 1, 5, 0); A16_PV(tb, 2, 3, 1, 5, 1);
    A16_PV(tb, 4, 5, 1, 6, 0); A16_PV(tb, 4, 5, 1, 6, 1); A16_PV(tb, 6, 7, 1, 7, 0); A16_PV(tb, 6, 7, 1, 7, 1);
    if (!LSUM) {
      l_run[0] += ls0;
      l_run[1] += ls1;
    }
#undef A16_LS
#undef A16_EXP
#undef A16_EXP4
#undef A16_TR
#undef A16_TR8
#undef A16_WAIT8
#undef A16_PV
#undef A16_F
    // tile j+1 must have landed; the eight instructions of tile j+2 (if issued; waves 4-7 issue none) may stay in flight
    if (AHEAD == 2 && j + 2 < jt1) A16_WAIT_TILE_AHEAD();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (two stages: the tile requested at the top of this one is the next)
    __builtin_amdgcn_s_barrier();
  }
  }
#undef A16_DMA
#undef A16_DMA8
#undef A16_WAIT_TILE_AHEAD
#undef A16_DMA_F
#undef A16_DMA_S

  // ---- epilogue: O[q, d] = O^T[d, q] / l ; lane holds d = 16 db + 4 g4 + e of query 16 nq + n16
#pragma unroll
  for (int nq = 0; nq < 2; ++nq) {
    float l;
    if (LSUM) {
      l = lacc[nq][0];  // row (any) x column n16 of the ones . P^T product: the whole row sum of query 16 nq + n16
    } else {
      l = l_run[nq];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
    if (SPLIT) {  // unnormalised partials; attn_combine_kernel merges the splits
      const int qs = q0 + 16 * nq + n16;
      if (qs < p.Lq) {
        float* po = p.part_o + ((int64_t)blockIdx.z * p.Lq + qs) * (p.H * AT_D) + head * AT_D + 4 * g4;
#pragma unroll
        for (int db = 0; db < 8; ++db)
          *reinterpret_cast<float4*>(po + 16 * db) = make_float4(o[db][nq][0], o[db][nq][1], o[db][nq][2], o[db][nq][3]);
        if (g4 == 0) {
          float* pm = p.part_ml + (((int64_t)blockIdx.z * p.Lq + qs) * p.H + head) * 2;
          // the merge kernel computes exp2((m - M) * p.c): hand it m in raw-score units (QK8: fold this query's delta_q in;
          // bf16 form: m_run already carries scale * log2(e))
          pm[0] = QK8 ? m_run[nq] * (c2[nq] / p.c) : m_run[nq] / p.c;
          pm[1] = l;
        }
      }
      continue;
    }
    const float inv = 1.0f / l;
    const int qr = q0 + 16 * nq + n16;
    if (qr < p.Lq) {
      uint16_t* op = p.o + (int64_t)qr * p.o_stride + head * AT_D + 4 * g4;
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
        bf16x4 b;
#pragma unroll
        for (int e = 0; e < 4; ++e) b[e] = (__bf16)(o[db][nq][e] * inv);
        *reinterpret_cast<bf16x4*>(op + 16 * db) = b;
      }
    }
  }
}

// Split-KV merge: out[q, h, :] = sum_z w_z O_z / sum_z w_z l_z,  w_z = 2^{(m_z - max_z m_z) c}.  One thread per 4 channels.
__global__ __launch_bounds__(256) void attn_combine_kernel(const AttnParams p, int splits) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int per_q = p.H * (AT_D / 4);
  const int64_t q = t / per_q;
  if (q >= p.Lq) return;
  const int r = (int)(t - q * per_q), h = r / (AT_D / 4), d4 = r % (AT_D / 4);
  float M = -INFINITY;
  for (int z = 0; z < splits; ++z) M = fmaxf(M, p.part_ml[(((int64_t)z * p.Lq + q) * p.H + h) * 2]);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float l = 0.f;
  for (int z = 0; z < splits; ++z) {
    const float* ml = p.part_ml + (((int64_t)z * p.Lq + q) * p.H + h) * 2;
    const float w = __builtin_amdgcn_exp2f((ml[0] - M) * p.c);
    const float4 v = *reinterpret_cast<const float4*>(p.part_o + ((int64_t)z * p.Lq + q) * (p.H * AT_D) + h * AT_D + 4 * d4);
    acc.x += w * v.x; acc.y += w * v.y; acc.z += w * v.z; acc.w += w * v.w;
    l += w * ml[1];
  }
  const float inv = 1.0f / l;
  typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
  bf16x4 b;
  b[0] = (__bf16)(acc.x * inv); b[1] = (__bf16)(acc.y * inv); b[2] = (__bf16)(acc.z * inv); b[3] = (__bf16)(acc.w * inv);
  *reinterpret_cast<bf16x4*>(p.o + q * p.o_stride + h * AT_D + 4 * d4) = b;
}

}  // namespace wanq

using namespace wanq;

extern "C" int64_t wanq_attention_split_workspace(int64_t Lq, int heads, int head_dim, int splits);

struct Qk8Args {  // int8 Q.K^T operands (NULL q8 = the bf16 form)
  const int8_t* q8;
  const int8_t* k8;
  const float* q_scale;
  const float* k_scale;
  int64_t q8_stride, k8_stride, qs_stride, ks_stride;
};

static volatile int64_t g_nw4_keys = -1;  // wanq_attention_select_form

template <bool SPLIT, bool QK8>
static void launch_attn(const AttnParams& p, dim3 grid, hipStream_t st) {
  constexpr int lds = 3 * (QK8 ? AT_STAGE8 : AT_STAGE);
  static const bool attr_set = [] {  // once per instantiation, thread-safe
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<true, SPLIT, QK8>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    return true;
  }();
  (void)attr_set;
  hipLaunchKernelGGL((attn_fwd_kernel<true, SPLIT, QK8>), grid, dim3(512), lds, st, p);
}

// The bf16 kernel exists in two MFMA shapes: 16x16x32 (default: the part holds a higher clock under it and the softmax's
// exponentials fit one per MFMA; +5 % at cfg-B in one process, +3.4 % inside the step) and 32x32x16 (WANQ_ATTN_M16=0).
#ifndef WANQ_ATTN_M16_DEFAULT
#define WANQ_ATTN_M16_DEFAULT 1
#endif
static bool use_m16() {
  static const bool m16 = [] {
    const char* e = getenv("WANQ_ATTN_M16");
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd16_kernel<false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * AT_STAGE);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd16_kernel<false, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * AT_STAGE);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd16_kernel<true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * AT_STAGE);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd16_kernel<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * AT_STAGE8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd16_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * AT_STAGE8);
    return e ? e[0] == '1' : (WANQ_ATTN_M16_DEFAULT != 0);
  }();
  return m16;
}

static int attention_impl(const void* q, const void* k, const void* v, void* o, int dtype, int64_t Lq, int64_t Lk, int heads,
                          int head_dim, int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride, float scale,
                          int splits, void* workspace, int64_t workspace_bytes, void* stream, const Qk8Args* q8 = nullptr) {
  const char* what = q8 ? "wanq_attention_qk8_fwd" : "wanq_attention_fwd";
  WANQ_REQUIRE((q8 || (q && k)) && v && o, WANQ_E_ARG, "%s: NULL pointer", what);
  WANQ_REQUIRE(dtype == WANQ_BF16, WANQ_E_ARG, "%s: only bf16 is implemented (dtype code %d)", what, dtype);
  WANQ_REQUIRE(head_dim == AT_D, WANQ_E_SHAPE, "%s: head_dim=%d, only 128 is implemented", what, head_dim);
  WANQ_REQUIRE(heads >= 1 && heads <= 65535, WANQ_E_SHAPE, "%s: heads=%d out of range", what, heads);
  WANQ_REQUIRE(Lq >= 0 && Lk >= 1 && Lq < (1ll << 30) && Lk < (1ll << 30), WANQ_E_SHAPE, "%s: bad lengths", what);
  const int64_t need = (int64_t)heads * head_dim;
  WANQ_REQUIRE(v_stride >= need && o_stride >= need && (q8 || (q_stride >= need && k_stride >= need)), WANQ_E_SHAPE,
               "%s: token stride smaller than heads*head_dim", what);
  WANQ_REQUIRE((v_stride | o_stride) % 8 == 0 && (q8 || (q_stride | k_stride) % 8 == 0), WANQ_E_SHAPE,
               "%s: strides must be multiples of 8 elements", what);
  WANQ_REQUIRE(v_stride < (1ll << 24) && (q8 || k_stride < (1ll << 24)), WANQ_E_SHAPE,
               "%s: k / v token stride must be below 2^24 elements (32-bit lane offsets inside a 64-key tile)", what);
  if (q8) {
    WANQ_REQUIRE(q8->q8 && q8->k8 && q8->q_scale && q8->k_scale, WANQ_E_ARG, "%s: NULL pointer", what);
    WANQ_REQUIRE(q8->q8_stride >= need && q8->k8_stride >= need && q8->q8_stride % 16 == 0 && q8->k8_stride % 16 == 0 &&
                     q8->k8_stride < (1ll << 24),
                 WANQ_E_SHAPE, "%s: int8 token strides must be multiples of 16 bytes, >= heads*128 and below 2^24", what);
    WANQ_REQUIRE(q8->qs_stride >= Lq && q8->ks_stride >= ((Lk + AT_KB - 1) / AT_KB) * AT_KB, WANQ_E_SHAPE,
                 "%s: scale planes are [heads][stride] with q stride >= Lq and k stride >= Lk rounded up to 64", what);
  }
  if (Lq == 0) return WANQ_OK;
  AttnParams p{(const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, (uint16_t*)o, q_stride, k_stride, v_stride, o_stride,
               (int)Lq, (int)Lk, heads, scale * 1.4426950408889634f};
  if (q8) {
    p.q8 = q8->q8; p.k8 = q8->k8; p.q_scale = q8->q_scale; p.k_scale = q8->k_scale;
    p.q8_stride = q8->q8_stride; p.k8_stride = q8->k8_stride; p.qs_stride = q8->qs_stride; p.ks_stride = q8->ks_stride;
  }
  // Default: LDS-DMA staging into a three-stage ring (96 KiB).  WANQ_ATTN_V1=1 selects the register-staged two-stage form
  // (64 KiB) kept for A/B timing.
  static const bool v1 = [] { const char* e = getenv("WANQ_ATTN_V1"); return e && e[0] == '1'; }();
  dim3 grid((unsigned)((Lq + AT_QB - 1) / AT_QB), (unsigned)heads);
  const int nt = (int)((Lk + AT_KB - 1) / AT_KB);
  if (splits > nt) splits = nt;
  if (splits > 1) {
    p.tiles_per_split = (nt + splits - 1) / splits;
    splits = (nt + p.tiles_per_split - 1) / p.tiles_per_split;  // no empty share
  }
  hipStream_t st = (hipStream_t)stream;
  if (splits <= 1) {
    if (q8) {
      if (use_m16()) hipLaunchKernelGGL((attn_fwd16_kernel<false, true>), grid, dim3(512), 3 * AT_STAGE8, st, p);
      else launch_attn<false, true>(p, grid, st);
    } else if (v1) {
      static const bool attr_v1 = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * AT_STAGE);
        return true;
      }();
      (void)attr_v1;
      hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, dim3(512), 2 * AT_STAGE, st, p);
    } else if (use_m16()) {
      // 4-wave workgroups of 128 queries, two per CU (see attn_fwd16_kernel), up to WANQ_ATTN_NW4_KEYS keys (0 = never)
      static const int64_t nw4_env = [] { const char* e = getenv("WANQ_ATTN_NW4_KEYS"); return e ? atoll(e) : (int64_t)WANQ_ATTN_NW4_KEYS_DEFAULT; }();
      const int64_t nw4_sel = g_nw4_keys;  // wanq_attention_select_form: -1 = the start-up value
      if (Lk <= (nw4_sel >= 0 ? nw4_sel : nw4_env)) {
        const dim3 grid4((unsigned)((Lq + 4 * AT_QW - 1) / (4 * AT_QW)), (unsigned)heads);
        hipLaunchKernelGGL((attn_fwd16_kernel<false, false, 4>), grid4, dim3(256), 2 * AT_STAGE, st, p);
      } else {
        hipLaunchKernelGGL((attn_fwd16_kernel<false, false>), grid, dim3(512), 3 * AT_STAGE, st, p);
      }
    } else {
      launch_attn<false, false>(p, grid, st);
    }
#ifdef WANQ_CLOCK_PROBE
    {
      (void)hipStreamSynchronize((hipStream_t)stream);
      unsigned long long h[2];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof(h));
      printf("[clock] attention Lq=%lld Lk=%lld: %llu cycles in %.1f us -> %.0f MHz\n", (long long)Lq, (long long)Lk, h[0], h[1] / 100.0, h[0] / (h[1] / 100.0));
    }
#endif
#ifdef WANQ_ATTN_STAMP
    {
      (void)hipStreamSynchronize((hipStream_t)stream);
      unsigned long long h[8 * 16];
      (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof(h));
      const char* names[6] = {"dma issue", "S = K.Q^T", "softmax", "PV", "vmcnt wait", "barrier"};
      printf("[stamp] Lq=%lld Lk=%lld tiles=%d: s_memtime ticks per tile, waves 0 / 3 / 4 / 7\n", (long long)Lq, (long long)Lk, nt);
      for (int i = 0; i < 6; ++i)
        printf("[stamp] %-10s %8.1f %8.1f %8.1f %8.1f\n", names[i], (double)h[0 * 16 + i] / nt, (double)h[3 * 16 + i] / nt, (double)h[4 * 16 + i] / nt, (double)h[7 * 16 + i] / nt);
    }
#endif
    return check_launch(what);
  }
  const int64_t need_ws = wanq_attention_split_workspace(Lq, heads, head_dim, splits);
  WANQ_REQUIRE(workspace && workspace_bytes >= need_ws, WANQ_E_ARG,
               "%s: split-KV workspace of %lld bytes needed, %lld given", what, (long long)need_ws, (long long)workspace_bytes);
  WANQ_REQUIRE(((uintptr_t)workspace & 15) == 0, WANQ_E_ARG, "%s: workspace must be 16-byte aligned", what);
  p.part_o = static_cast<float*>(workspace);
  p.part_ml = p.part_o + (int64_t)splits * Lq * heads * AT_D;
  grid.z = (unsigned)splits;
  if (q8 && use_m16()) hipLaunchKernelGGL((attn_fwd16_kernel<true, true>), grid, dim3(512), 3 * AT_STAGE8, st, p);
  else if (q8) launch_attn<true, true>(p, grid, st);
  else if (use_m16()) hipLaunchKernelGGL((attn_fwd16_kernel<true, false>), grid, dim3(512), 3 * AT_STAGE, st, p);
  else launch_attn<true, false>(p, grid, st);
  const int64_t threads = Lq * heads * (AT_D / 4);
  hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, p, splits);
  return check_launch(what);
}

extern "C" int64_t wanq_attention_split_workspace(int64_t Lq, int heads, int head_dim, int splits) {
  if (splits <= 1) return 0;
  return (int64_t)splits * Lq * heads * (head_dim + 2) * (int64_t)sizeof(float);
}

extern "C" int64_t wanq_attention_select_form(int64_t nw4_keys) {
  const int64_t prev = g_nw4_keys;
  g_nw4_keys = nw4_keys < -1 ? -1 : nw4_keys;
  return prev;
}

extern "C" int wanq_attention_fwd(const void* q, const void* k, const void* v, void* o, int dtype, int64_t Lq,
                                  int64_t Lk, int heads, int head_dim, int64_t q_stride, int64_t k_stride,
                                  int64_t v_stride, int64_t o_stride, float scale, void* stream) {
  return attention_impl(q, k, v, o, dtype, Lq, Lk, heads, head_dim, q_stride, k_stride, v_stride, o_stride, scale, 1, nullptr, 0, stream);
}

extern "C" int wanq_attention_fwd_split(const void* q, const void* k, const void* v, void* o, int dtype, int64_t Lq,
                                        int64_t Lk, int heads, int head_dim, int64_t q_stride, int64_t k_stride,
                                        int64_t v_stride, int64_t o_stride, float scale, int splits, void* workspace,
                                        int64_t workspace_bytes, void* stream) {
  WANQ_REQUIRE(splits >= 1 && splits <= 64, WANQ_E_ARG, "wanq_attention_fwd_split: splits=%d must be 1..64", splits);
  return attention_impl(q, k, v, o, dtype, Lq, Lk, heads, head_dim, q_stride, k_stride, v_stride, o_stride, scale, splits, workspace,
                        workspace_bytes, stream);
}

extern "C" int wanq_attention_qk8_fwd(const int8_t* q8, const float* q_scale, int64_t qs_stride, const int8_t* k8,
                                      const float* k_scale, int64_t ks_stride, const void* v, void* o, int dtype, int64_t Lq,
                                      int64_t Lk, int heads, int head_dim, int64_t q8_stride, int64_t k8_stride,
                                      int64_t v_stride, int64_t o_stride, float scale, int splits, void* workspace,
                                      int64_t workspace_bytes, void* stream) {
  WANQ_REQUIRE(splits >= 1 && splits <= 64, WANQ_E_ARG, "wanq_attention_qk8_fwd: splits=%d must be 1..64", splits);
  const Qk8Args a{q8, k8, q_scale, k_scale, q8_stride, k8_stride, qs_stride, ks_stride};
  return attention_impl(nullptr, nullptr, v, o, dtype, Lq, Lk, heads, head_dim, 0, 0, v_stride, o_stride, scale, splits, workspace,
                        workspace_bytes, stream, &a);
}
