// Flash-attention forward for gfx950 (bf16 MFMA 32x32x16), head_dim 128, non-causal, key-length masking.
//
//   O[q, h, :] = softmax_k( Q[q,h,:] . K[k,h,:] / sqrt(d) ) V[k,h,:]
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
// tiles ahead of the math, published by a counted s_waitcnt vmcnt(4) + one bare s_barrier per tile (DMA = true, the
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
};

constexpr int AT_D = 128, AT_QW = 32, AT_NW = 8, AT_QB = AT_QW * AT_NW, AT_KB = 64;
constexpr int AT_TILE = AT_KB * AT_D * 2;  // 16 KiB per K or V tile
constexpr int AT_STAGE = 2 * AT_TILE;

__device__ __forceinline__ int at_off(int row, int ch) {
  return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

__device__ __forceinline__ bf16x8 at_join(s16x4 lo, s16x4 hi) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, vv);
}

#ifdef WANQ_ATTN_STAMP
__device__ unsigned long long g_stamp[8 * 16];
#define STAMP(i) { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); acc_[i] += t_ - last_; last_ = t_; }
#else
#define STAMP(i)
#endif

template <bool DMA, bool SPLIT = false>
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(const AttnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 31, fh = lane >> 5;
  const int head = blockIdx.y;
  const int q0 = blockIdx.x * AT_QB + wave * AT_QW;
  const int nt = (p.Lk + AT_KB - 1) / AT_KB;
  const float c = p.c;
  // key tiles of this workgroup: all of them, or one contiguous share under split-KV (the host makes every share non-empty)
  const int jt0 = SPLIT ? (int)blockIdx.z * p.tiles_per_split : 0;
  const int jt1 = SPLIT ? (jt0 + p.tiles_per_split < nt ? jt0 + p.tiles_per_split : nt) : nt;

  // ---- Q fragments: query (q0+fr), d = 16 s + 8 fh + [0,8)
  bf16x8 qf[8];
  {
    int qr = q0 + fr;
    if (qr >= p.Lq) qr = p.Lq - 1;
    const uint16_t* qp = p.q + (int64_t)qr * p.q_stride + head * AT_D + 8 * fh;
#pragma unroll
    for (int s = 0; s < 8; ++s) qf[s] = *reinterpret_cast<const bf16x8*>(qp + 16 * s);
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
  const uint32_t d_k0 = AT_DOFF(p.k_stride, 0), d_k1 = AT_DOFF(p.k_stride, 1), d_k2 = AT_DOFF(p.k_stride, 2), d_k3 = AT_DOFF(p.k_stride, 3);
  const uint32_t d_v0 = AT_DOFF(p.v_stride, 0), d_v1 = AT_DOFF(p.v_stride, 1), d_v2 = AT_DOFF(p.v_stride, 2), d_v3 = AT_DOFF(p.v_stride, 3);
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
#define AT_DMA(j, stage)                                                                                        \
  do {                                                                                                          \
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
  float m_run = -INFINITY, l_run = 0.f;

  if (DMA) {
    AT_DMA(jt0, 0);
    if (jt0 + 1 < jt1) {
      AT_DMA(jt0 + 1, 1);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
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
  int st3 = 0;  // DMA: ring of three stages, (j - jt0) % 3
  for (int j = jt0; j < jt1; ++j) {
    const int cur = DMA ? st3 : (j & 1);
    const char* sK = smem + cur * AT_STAGE;
    const char* sV = sK + AT_TILE;
    // every wave is past the barrier that ended tile j-1, so the stage that held it is free: tile j+2 goes there and has
    // two tile-times to land
    const int st_free = st3 == 0 ? 2 : st3 - 1;
    if (DMA && j + 2 < jt1) AT_DMA(j + 2, st_free);
    STAMP(0)

    // ---------------- S^T = K . Q^T  (two 32-key blocks)
    f32x16 s0, s1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
    if (DMA) {
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
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s][0], qf[s], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[s][1], qf[s], s1, 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(sK + at_off(fr, 2 * s + fh));
        const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(sK + at_off(32 + fr, 2 * s + fh));
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[s], s0, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[s], s1, 0, 0, 0);
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
    // Running max, rescaled lazily: the accumulators are touched only when some query's max grew by more than 2^6
    // (exp2 domain) since the last rescale (wave-uniform vote), so P stays <= 64 instead of <= 1 -- same 8
    // significant bits in bf16, fp32 accumulators unaffected -- and the 64-multiply rescale of O^T almost never runs
    // after the first tiles (+3.4 % measured).
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
    const float mc = m_run * c;
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s0[r] = __builtin_amdgcn_exp2f(fmaf(s0[r], c, -mc));
      s1[r] = __builtin_amdgcn_exp2f(fmaf(s1[r], c, -mc));
      ls += s0[r] + s1[r];
    }
    l_run += ls;
    bf16x8 pf[4];  // key slice ks = 2*block + t: registers 8t..8t+7 of that block's accumulator
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pf[0][e] = (__bf16)s0[e];
      pf[1][e] = (__bf16)s0[8 + e];
      pf[2][e] = (__bf16)s1[e];
      pf[3][e] = (__bf16)s1[8 + e];
    }

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
      const uint32_t vb = lds_base + cur * AT_STAGE;
      s16x4 ta0, ta1, ta2, ta3, ta4, ta5, ta6, ta7, tb0, tb1, tb2, tb3, tb4, tb5, tb6, tb7;
#define AT_TR(dst, areg, ks) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(vb + areg), "n"(AT_TILE + 4096 * (ks)))
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
      AT_TR8(ta, 0);
      AT_TR8(tb, 1);
      AT_WAIT8(ta, 8);
      AT_PV4(ta, 0);
      AT_TR8(ta, 2);
      AT_WAIT8(tb, 8);
      AT_PV4(tb, 1);
      AT_TR8(tb, 3);
      AT_WAIT8(ta, 8);
      AT_PV4(ta, 2);
      AT_WAIT8(tb, 0);
      AT_PV4(tb, 3);
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
    }
    STAMP(3)
    if (DMA) {
      // tile j+1 must have landed; the eight instructions of tile j+2 (if issued; waves 4-7 issue none) may stay in flight
      if (j + 2 < jt1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      st3 = st3 == 2 ? 0 : st3 + 1;
      STAMP(4)
      __builtin_amdgcn_s_barrier();  // bare: __syncthreads() would drain vmcnt to 0 and with it the prefetch
    } else {
      __syncthreads();
    }
    STAMP(5)
  }
#ifdef WANQ_ATTN_STAMP
  if (blockIdx.x == 7 && blockIdx.y == 3 && lane == 0) {
    for (int i = 0; i < 6; ++i) g_stamp[wave * 16 + i] = acc_[i];
  }
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
        pm[0] = m_run;
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

static int attention_impl(const void* q, const void* k, const void* v, void* o, int dtype, int64_t Lq, int64_t Lk, int heads,
                          int head_dim, int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride, float scale,
                          int splits, void* workspace, int64_t workspace_bytes, void* stream) {
  WANQ_REQUIRE(q && k && v && o, WANQ_E_ARG, "wanq_attention_fwd: NULL pointer");
  WANQ_REQUIRE(dtype == WANQ_BF16, WANQ_E_ARG, "wanq_attention_fwd: only bf16 is implemented (dtype code %d)", dtype);
  WANQ_REQUIRE(head_dim == AT_D, WANQ_E_SHAPE, "wanq_attention_fwd: head_dim=%d, only 128 is implemented", head_dim);
  WANQ_REQUIRE(heads >= 1 && heads <= 65535, WANQ_E_SHAPE, "wanq_attention_fwd: heads=%d out of range", heads);
  WANQ_REQUIRE(Lq >= 0 && Lk >= 1 && Lq < (1ll << 30) && Lk < (1ll << 30), WANQ_E_SHAPE, "wanq_attention_fwd: bad lengths");
  const int64_t need = (int64_t)heads * head_dim;
  WANQ_REQUIRE(q_stride >= need && k_stride >= need && v_stride >= need && o_stride >= need, WANQ_E_SHAPE,
               "wanq_attention_fwd: token stride smaller than heads*head_dim");
  WANQ_REQUIRE((q_stride | k_stride | v_stride | o_stride) % 8 == 0, WANQ_E_SHAPE, "wanq_attention_fwd: strides must be multiples of 8 elements");
  WANQ_REQUIRE(k_stride < (1ll << 24) && v_stride < (1ll << 24), WANQ_E_SHAPE,
               "wanq_attention_fwd: k / v token stride must be below 2^24 elements (32-bit lane offsets inside a 64-key tile)");
  if (Lq == 0) return WANQ_OK;
  AttnParams p{(const uint16_t*)q, (const uint16_t*)k, (const uint16_t*)v, (uint16_t*)o, q_stride, k_stride, v_stride, o_stride,
               (int)Lq, (int)Lk, heads, scale * 1.4426950408889634f};
  // Default: LDS-DMA staging into a three-stage ring (96 KiB).  WANQ_ATTN_V1=1 selects the register-staged two-stage form
  // (64 KiB) kept for A/B timing.
  static const bool v1 = [] { const char* e = getenv("WANQ_ATTN_V1"); return e && e[0] == '1'; }();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * AT_STAGE);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * AT_STAGE);
    attr_set = true;
  }
  dim3 grid((unsigned)((Lq + AT_QB - 1) / AT_QB), (unsigned)heads);
  const int nt = (int)((Lk + AT_KB - 1) / AT_KB);
  if (splits > nt) splits = nt;
  if (splits > 1) {
    p.tiles_per_split = (nt + splits - 1) / splits;
    splits = (nt + p.tiles_per_split - 1) / p.tiles_per_split;  // no empty share
  }
  if (splits <= 1) {
    if (v1) hipLaunchKernelGGL(attn_fwd_kernel<false>, grid, dim3(512), 2 * AT_STAGE, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(attn_fwd_kernel<true>, grid, dim3(512), 3 * AT_STAGE, (hipStream_t)stream, p);
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
    return check_launch("wanq_attention_fwd");
  }
  const int64_t need_ws = wanq_attention_split_workspace(Lq, heads, head_dim, splits);
  WANQ_REQUIRE(workspace && workspace_bytes >= need_ws, WANQ_E_ARG,
               "wanq_attention_fwd_split: workspace of %lld bytes needed, %lld given", (long long)need_ws, (long long)workspace_bytes);
  WANQ_REQUIRE(((uintptr_t)workspace & 15) == 0, WANQ_E_ARG, "wanq_attention_fwd_split: workspace must be 16-byte aligned");
  p.part_o = static_cast<float*>(workspace);
  p.part_ml = p.part_o + (int64_t)splits * Lq * heads * AT_D;
  static bool attr_split = false;
  if (!attr_split) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_kernel<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 3 * AT_STAGE);
    attr_split = true;
  }
  grid.z = (unsigned)splits;
  hipLaunchKernelGGL((attn_fwd_kernel<true, true>), grid, dim3(512), 3 * AT_STAGE, (hipStream_t)stream, p);
  const int64_t threads = Lq * heads * (AT_D / 4);
  hipLaunchKernelGGL(attn_combine_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, splits);
  return check_launch("wanq_attention_fwd_split");
}

extern "C" int64_t wanq_attention_split_workspace(int64_t Lq, int heads, int head_dim, int splits) {
  if (splits <= 1) return 0;
  return (int64_t)splits * Lq * heads * (head_dim + 2) * (int64_t)sizeof(float);
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
