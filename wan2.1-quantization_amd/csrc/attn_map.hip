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
//           then a small kernel turns c_k into (delta_k, 1 / delta_k)
//   pass 2  O = sum_k P~[q,k] V[k]                           (P~ rounded to bf16 for the P.V MFMA, fp32 accumulation)
//
// Round 3: the passes share the structure of the flash-attention kernel (attention.hip, 16x16 form) instead of reading their
// operands straight from global memory: one workgroup = 8 waves x 32 queries of one head, 64-key tiles, K (and in pass 2 V)
// tiles by LDS-DMA into a ring of three stages two tiles ahead, one counted s_waitcnt vmcnt + one s_barrier per tile, the same
// swizzled LDS images and fragment reads (K rows as the MFMA A operand, Q fragments resident in registers, V^T gathered by
// ds_read_b64_tr_b16).  Pass 1 reduces a key's maximum over the workgroup's 256 queries in registers (DPP) and LDS and issues
// ONE 64-lane atomic instruction per tile and workgroup (round 2: one single-lane atomic per key and 32-query wave -- 4e8 of
// them per cfg-B launch, the bulk of its 49 ms).
// QK8 = the reference's full recipe (quant_opensora.py:431-476 applies the q / k / v quantisers AND the map quantiser): q and k
// arrive as per-(token, head) int8 codes + fp32 scales (wanq_rmsnorm_rope_q8) and S = K8 . Q8^T runs on v_mfma_i32_16x16x64_i8 in
// all three passes; v is fake-quantised by the caller before the call (wanq_fake_quant_cols).
#include "wanq_common.h"

namespace wanq {

typedef __bf16 mbf16x8 __attribute__((ext_vector_type(8)));
typedef float mf32x4 __attribute__((ext_vector_type(4)));
typedef int mi32x4 __attribute__((ext_vector_type(4)));
typedef short ms16x4 __attribute__((ext_vector_type(4)));

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
  float* delta;  // [2][H][Lk]: delta_k, 1 / delta_k
  // QK8: per-(token, head) int8 codes [tokens, H*128] and fp32 scale planes [H][stride] (delta; the second plane is unused here)
  const int8_t* q8;
  const int8_t* k8;
  int64_t q8_stride, k8_stride;  // bytes between consecutive tokens
  const float* q_scale;
  const float* k_scale;
  int64_t qs_stride, ks_stride;
};

constexpr int AM_KB = 64;                  // keys per tile
constexpr int AM_TILE = AM_KB * 128 * 2;   // one bf16 K or V tile: 16 KiB
// ring stage: K | V in pass 2; K alone in passes 0 and 1 (48 KiB of ring: two workgroups per CU, four waves per SIMD -- those
// passes are exponentials and lane reductions around 32 MFMAs per tile and want the latency hiding).  The int8 K tile takes
// the first 8 KiB of its slot.
constexpr int am_stage(int pass) { return pass == 2 ? 2 * AM_TILE : AM_TILE; }
constexpr int am_cm(int pass) { return 3 * am_stage(pass); }  // pass 1: per-wave column maxima, [2][8][64] floats, behind the ring

__device__ __forceinline__ int am_off(int row, int ch) {  // V image (attention.hip at_off)
  return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}
__device__ __forceinline__ mbf16x8 am_join(ms16x4 lo, ms16x4 hi) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 vv = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(mbf16x8, vv);
}

template <int PASS, bool QK8>
__global__ __launch_bounds__(512, PASS == 2 ? 2 : 4) void attn_map_kernel(const AttnMapParams p) {
  constexpr int AM_STAGE = am_stage(PASS), AM_CM = am_cm(PASS);
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n16 = lane & 15, g4 = lane >> 4;
  const int head = blockIdx.y;
  const int q0 = blockIdx.x * 256 + wave * 32;
  const int nt = (p.Lk + AM_KB - 1) / AM_KB;
  constexpr int PIECES = (QK8 ? 1 : 2) + (PASS == 2 ? 2 : 0);  // LDS-DMA instructions per wave and tile

  // ---- Q fragments (bf16: pre-scaled by softmax scale * log2(e); int8: the query's scale rides in c2)
  int qr[2];
  bool q_ok[2];
  mbf16x8 qf[QK8 ? 1 : 2][QK8 ? 1 : 4];
  mi32x4 qf8[QK8 ? 2 : 1][QK8 ? 2 : 1];
  float c2[2] = {p.c, p.c};
#pragma unroll
  for (int nq = 0; nq < 2; ++nq) {
    qr[nq] = q0 + 16 * nq + n16;
    q_ok[nq] = qr[nq] < p.Lq;
    qr[nq] = q_ok[nq] ? qr[nq] : p.Lq - 1;
    if (QK8) {
      const int8_t* qp8 = p.q8 + (int64_t)qr[nq] * p.q8_stride + head * 128 + 16 * g4;
#pragma unroll
      for (int s = 0; s < 2; ++s) qf8[QK8 ? nq : 0][QK8 ? s : 0] = *reinterpret_cast<const mi32x4*>(qp8 + 64 * s);
      c2[nq] = p.c * p.q_scale[(int64_t)head * p.qs_stride + qr[nq]];
    } else {
      const uint16_t* qp = p.q + (int64_t)qr[nq] * p.q_stride + head * 128 + 8 * g4;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        mbf16x8 t = *reinterpret_cast<const mbf16x8*>(qp + 32 * s);
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = (__bf16)((float)t[e] * p.c);
        qf[QK8 ? 0 : nq][QK8 ? 0 : s] = t;
      }
    }
  }
  const int64_t stat[2] = {(int64_t)head * p.Lq + qr[0], (int64_t)head * p.Lq + qr[1]};

  // ---- LDS-DMA of tile j into ring stage st: every wave issues PIECES 1-KiB pieces.  bf16 K / V tile: 16 pieces, piece (gk, i)
  // = rows 16 gk + 4 i + (lane >> 4), the lane's 16-B chunk; int8 K tile: 8 pieces of 8 rows x 128 B.  The DMA writes
  // lane-linearly, so the image's chunk swizzle is applied on the source address (K: chunk ^ (row & 15); V: am_off; K8:
  // chunk ^ ((row >> 1) & 7)); rows at or beyond Lk repeat the last key (their scores are masked).
  const int d_r = lane >> 4, gk = wave & 3, ih = wave >> 2;
  auto dma_tile = [&](int j, int st) {
    char* sb = smem + st * AM_STAGE;
    if (QK8) {
      int kr = j * AM_KB + 16 * gk + 8 * ih + (lane >> 3);
      kr = kr < p.Lk ? kr : p.Lk - 1;
      const int col = head * 128 + (((lane & 7) ^ (((8 * ih + (lane >> 3)) >> 1) & 7)) << 4);
      __builtin_amdgcn_global_load_lds((glb_void*)(p.k8 + (int64_t)kr * p.k8_stride + col), (lds_void*)(sb + gk * 2048 + ih * 1024), 16, 0, 0);
    } else {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * ih + ii;
        int kr = j * AM_KB + 16 * gk + 4 * i + d_r;
        kr = kr < p.Lk ? kr : p.Lk - 1;
        const int col = head * 128 + (((lane & 15) ^ (4 * i + d_r)) << 3);
        __builtin_amdgcn_global_load_lds((glb_void*)(p.k + (int64_t)kr * p.k_stride + col), (lds_void*)(sb + gk * 4096 + i * 1024), 16, 0, 0);
      }
    }
    if (PASS == 2) {
#pragma unroll
      for (int ii = 0; ii < 2; ++ii) {
        const int i = 2 * ih + ii;
        int kr = j * AM_KB + 16 * gk + 4 * i + d_r;
        kr = kr < p.Lk ? kr : p.Lk - 1;
        const int col = head * 128 + ((((lane & 15) ^ (d_r << 2)) ^ i) << 3);
        __builtin_amdgcn_global_load_lds((glb_void*)(p.v + (int64_t)kr * p.v_stride + col), (lds_void*)(sb + AM_TILE + gk * 4096 + i * 1024), 16, 0, 0);
      }
    }
  };
  auto wait_tile_ahead = [&](bool more) {  // the tile issued LAST may stay in flight; everything older has landed
    if (!more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (PIECES == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (PIECES == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (PIECES == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  };

  // ---- fragment read offsets (attention.hip, attn_fwd16_kernel)
  const int kap = 4 * (2 * ((n16 >> 2) & 1) + ((n16 >> 3) & 1)) + (n16 & 3);  // kappa(n16)
  const int pg = 2 * (g4 & 1) + (g4 >> 1), tq = (lane >> 2) & 3, tp = lane & 3;
  const uint32_t lds_base = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)smem;

  // S^T blocks of tile j (stage st): sacc[kb][nq][e] = score of query 16 nq + n16 against key 64 j + 16 kb + 4 pg + e, in the
  // log2 domain (bf16 form) or as dot * delta_k (int8 form: times c2[nq] gives the log2-domain score); keys >= Lk: -inf
  // (int8 form) the four key-scale vectors of tile j, loaded IN FRONT OF the tile-(j+2) LDS-DMA of the same iteration: vmcnt retires
  // in order, so a wait for loads issued behind that DMA would wait for the DMA too and the two-tiles-ahead ring would drain in
  // every tile (pass 2 orders its delta loads the same way)
  auto load_sk = [&](int j, mf32x4 (&skv)[4]) {
    if (QK8) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)  // (the planes are padded to whole tiles: the host checks the stride)
        skv[kb] = *reinterpret_cast<const mf32x4*>(p.k_scale + (int64_t)head * p.ks_stride + j * AM_KB + 16 * kb + 4 * pg);
    }
  };
  auto s_tile = [&](int j, int st, mf32x4 (&sacc)[4][2], const mf32x4 (&skv)[4]) {
    const char* sK = smem + st * AM_STAGE;
    if (!QK8) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        sacc[kb][0] = mf32x4{0.f, 0.f, 0.f, 0.f};
        sacc[kb][1] = mf32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const mbf16x8 kf = *reinterpret_cast<const mbf16x8*>(sK + kb * 4096 + kap * 256 + (((4 * s + g4) ^ kap) << 4));
          sacc[kb][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][QK8 ? 0 : s], sacc[kb][0], 0, 0, 0);
          sacc[kb][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[QK8 ? 0 : 1][QK8 ? 0 : s], sacc[kb][1], 0, 0, 0);
        }
      }
    } else {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        mi32x4 ia0 = {0, 0, 0, 0}, ia1 = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const mi32x4 kf8 = *reinterpret_cast<const mi32x4*>(sK + kb * 2048 + kap * 128 + (((4 * s + g4) ^ ((kap >> 1) & 7)) << 4));
          ia0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf8, qf8[0][QK8 ? s : 0], ia0, 0, 0, 0);
          ia1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(kf8, qf8[QK8 ? 1 : 0][QK8 ? s : 0], ia1, 0, 0, 0);
        }
        const mf32x4 sk = skv[kb];  // the lane's four keys of this block: their scales
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          sacc[kb][0][e] = (float)ia0[e] * sk[e];
          sacc[kb][1][e] = (float)ia1[e] * sk[e];
        }
      }
    }
    if (j == nt - 1 && (p.Lk & (AM_KB - 1))) {
      const int kbase = j * AM_KB + 4 * pg;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (kbase + 16 * kb + e >= p.Lk) { sacc[kb][0][e] = -INFINITY; sacc[kb][1][e] = -INFINITY; }
    }
  };

  // ---- prologue: tiles 0 and 1 in flight, tile 0 published
  dma_tile(0, 0);
  if (nt > 1) dma_tile(1, 1);
  wait_tile_ahead(nt > 1);
  __builtin_amdgcn_s_barrier();

  if (PASS == 0) {
    // the column maxima pass 1 accumulates by atomic max start from zero: cleared here (one launch fewer than a memset)
    if (blockIdx.x == 0)
      for (int i = tid; i < p.Lk; i += 512) p.cmax[(int64_t)head * p.Lk + i] = 0.f;
    // lane-local online statistics over the keys this lane sees (a query's four lanes n, n+16, n+32, n+48 see different keys);
    // merged across the four lanes at the end
    float m[2] = {-INFINITY, -INFINITY}, l[2] = {0.f, 0.f};
    for (int j = 0; j < nt; ++j) {
      const int st = j % 3;
      mf32x4 skv[4];
      load_sk(j, skv);
      if (j + 2 < nt) dma_tile(j + 2, (j + 2) % 3);
      mf32x4 sacc[4][2];
      s_tile(j, st, sacc, skv);
#pragma unroll
      for (int nq = 0; nq < 2; ++nq) {
        float mx = sacc[0][nq][0];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int e = 0; e < 4; ++e) mx = fmaxf(mx, sacc[kb][nq][e]);
        const float cq = QK8 ? c2[nq] : 1.0f;
        const float mn = fmaxf(m[nq], mx * cq);  // finite from the first tile on for every lane (each lane's first key < Lk or Lk < 16)
        float acc = l[nq] * __builtin_amdgcn_exp2f(m[nq] - mn);
        if (mn == -INFINITY) acc = 0.f;  // (a lane whose keys are all masked so far)
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc += __builtin_amdgcn_exp2f(QK8 ? fmaf(sacc[kb][nq][e], cq, -mn) : sacc[kb][nq][e] - mn);
        if (mn != -INFINITY) { l[nq] = acc; m[nq] = mn; }
      }
      wait_tile_ahead(j + 2 < nt);
      __builtin_amdgcn_s_barrier();
    }
#pragma unroll
    for (int nq = 0; nq < 2; ++nq) {
      float mt = m[nq];
      mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      float lt = m[nq] == -INFINITY ? 0.f : l[nq] * __builtin_amdgcn_exp2f(m[nq] - mt);
      lt += __shfl_xor(lt, 16, 64);
      lt += __shfl_xor(lt, 32, 64);
      if (q_ok[nq] && g4 == 0) {
        p.m[stat[nq]] = mt;
        p.l[stat[nq]] = lt;
      }
    }
    return;
  }

  // m in the units the scores are compared in (int8 form: divided out of c2 below), 1 / l
  float mq[2] = {p.m[stat[0]], p.m[stat[1]]};
  const float inv_l[2] = {1.0f / p.l[stat[0]], 1.0f / p.l[stat[1]]};

  if (PASS == 1) {
    // ---- column maxima of the normalised map.  Per tile: the maximum over this wave's 32 queries of each of its 64 keys (two
    // query blocks in registers, 16 lanes by DPP) -> LDS [tile parity][wave][key]; behind the tile's barrier wave 0 reduces
    // the PREVIOUS tile's eight rows and issues one 64-lane atomic max (non-negative floats order like their bit patterns).
    // Clamped duplicates of a ragged last query block repeat a real query: harmless.
    float* cm = reinterpret_cast<float*>(smem + AM_CM);
    unsigned int* gcm = reinterpret_cast<unsigned int*>(p.cmax) + (int64_t)head * p.Lk;
    auto flush = [&](int jprev) {  // wave 0: tile jprev's maxima (written before the barrier every wave has passed)
      if (wave == 0) {
        const float* row = cm + (jprev & 1) * 512 + lane;
        float mxv = row[0];
#pragma unroll
        for (int w = 1; w < 8; ++w) mxv = fmaxf(mxv, row[64 * w]);
        const int key = jprev * AM_KB + lane;
        if (key < p.Lk) atomicMax(gcm + key, __float_as_uint(mxv));
      }
    };
    for (int j = 0; j < nt; ++j) {
      const int st = j % 3;
      mf32x4 skv[4];
      load_sk(j, skv);
      if (j + 2 < nt) dma_tile(j + 2, (j + 2) % 3);
      if (j > 0) flush(j - 1);
      mf32x4 sacc[4][2];
      s_tile(j, st, sacc, skv);
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        mf32x4 cv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float p0 = __builtin_amdgcn_exp2f(QK8 ? fmaf(sacc[kb][0][e], c2[0], -mq[0]) : sacc[kb][0][e] - mq[0]) * inv_l[0];
          const float p1 = __builtin_amdgcn_exp2f(QK8 ? fmaf(sacc[kb][1][e], c2[1], -mq[1]) : sacc[kb][1][e] - mq[1]) * inv_l[1];
          float pv = fmaxf(p0, p1);  // masked keys: 0
          pv = fmaxf(pv, lane_xor_dpp<1>(pv)); pv = fmaxf(pv, lane_xor_dpp<2>(pv));
          pv = fmaxf(pv, lane_xor_dpp<4>(pv)); pv = fmaxf(pv, lane_xor_dpp<8>(pv));
          cv[e] = pv;
        }
        if (n16 == 0) *reinterpret_cast<mf32x4*>(cm + (j & 1) * 512 + wave * 64 + 16 * kb + 4 * pg) = cv;
      }
      wait_tile_ahead(j + 2 < nt);
      // the column maxima above are handed to wave 0 through LDS across this barrier: make the write (and wave 0's reads of the
      // slot it is about to be reused for) complete, not merely issued -- hipcc puts no lgkmcnt wait in front of a raw s_barrier
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    flush(nt - 1);
    return;
  }

  // ---- PASS 2: O^T += V^T . P~^T.  P~ of key slice ks (32 keys) for query block nq: element 4 (kb & 1) + e of S block
  // kb = 2 ks + (idx >> 2) (the k index 8 g + idx of the P.V MFMA <-> key 32 ks + 16 (idx >> 2) + 4 pi(g) + (idx & 3)); the A
  // operand V^T is gathered by two ds_read_b64_tr_b16 per d block.
  mf32x4 o[8][2];
#pragma unroll
  for (int db = 0; db < 8; ++db) {
    o[db][0] = mf32x4{0.f, 0.f, 0.f, 0.f};
    o[db][1] = mf32x4{0.f, 0.f, 0.f, 0.f};
  }
  uint32_t va[8];
#pragma unroll
  for (int db = 0; db < 8; ++db) va[db] = lds_base + AM_TILE + (uint32_t)(am_off(4 * pg + tq, 2 * db + (tp >> 1)) + 8 * (tp & 1));
  const float* dl = p.delta + (int64_t)head * p.Lk;
  const float* dinv = p.delta + ((int64_t)p.H + head) * p.Lk;
  for (int j = 0; j < nt; ++j) {
    const int st = j % 3;
    // this tile's quantisation steps first (older than the DMA below, so the counted wait at the tile's end stays exact)
    mf32x4 dk[4], di[4];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int k0 = j * AM_KB + 16 * kb + 4 * pg;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int kc = k0 + e < p.Lk ? k0 + e : p.Lk - 1;
        dk[kb][e] = dl[kc];
        di[kb][e] = dinv[kc];
      }
    }
    mf32x4 skv[4];
    load_sk(j, skv);
    if (j + 2 < nt) dma_tile(j + 2, (j + 2) % 3);
    mf32x4 sacc[4][2];
    s_tile(j, st, sacc, skv);
    mbf16x8 pf[2][2];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int nq = 0; nq < 2; ++nq)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pr = __builtin_amdgcn_exp2f(QK8 ? fmaf(sacc[kb][nq][e], c2[nq], -mq[nq]) : sacc[kb][nq][e] - mq[nq]) * inv_l[nq];
          pf[kb >> 1][nq][4 * (kb & 1) + e] = (__bf16)(__builtin_rintf(pr * di[kb][e]) * dk[kb][e]);
        }
    const uint32_t vst = (uint32_t)(st * AM_STAGE);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      ms16x4 t0[8], t1[8];
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        if (ks == 0) {
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:0" : "=v"(t0[db]) : "v"(va[db] + vst));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:4096" : "=v"(t1[db]) : "v"(va[db] + vst));
        } else {
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:8192" : "=v"(t0[db]) : "v"(va[db] + vst));
          asm volatile("ds_read_b64_tr_b16 %0, %1 offset:12288" : "=v"(t1[db]) : "v"(va[db] + vst));
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(t0[0]), "+v"(t0[1]), "+v"(t0[2]), "+v"(t0[3]), "+v"(t0[4]), "+v"(t0[5]), "+v"(t0[6]), "+v"(t0[7]), "+v"(t1[0]),
                     "+v"(t1[1]), "+v"(t1[2]), "+v"(t1[3]), "+v"(t1[4]), "+v"(t1[5]), "+v"(t1[6]), "+v"(t1[7]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int db = 0; db < 8; ++db) {
        const mbf16x8 vf = am_join(t0[db], t1[db]);
        o[db][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[ks][0], o[db][0], 0, 0, 0);
        o[db][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[ks][1], o[db][1], 0, 0, 0);
      }
    }
    wait_tile_ahead(j + 2 < nt);
    __builtin_amdgcn_s_barrier();
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

// column maxima -> (delta_k, 1 / delta_k): delta = max / levels, floored at eps
__global__ __launch_bounds__(256) void attn_map_delta_kernel(const float* cmax, float* delta, int64_t n, float levels, float eps) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float d = cmax[i] / levels;
  d = d < eps ? eps : d;
  delta[i] = d;
  delta[n + i] = 1.0f / d;
}

template <bool QK8>
static int launch_attn_map(const AttnMapParams& p, hipStream_t st, const char* what) {
  const size_t lds0 = am_cm(0), lds1 = am_cm(1) + 2 * 8 * 64 * sizeof(float), lds2 = am_cm(2);
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute((const void*)attn_map_kernel<0, QK8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds0) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn_map_kernel<1, QK8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1) != hipSuccess ||
        hipFuncSetAttribute((const void*)attn_map_kernel<2, QK8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2) != hipSuccess) {
      set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed", what);
      return WANQ_E_LAUNCH;
    }
    attr_done = true;
  }
  const int64_t nk = (int64_t)p.Lk * p.H;
  const dim3 grid((unsigned)((p.Lq + 255) / 256), (unsigned)p.H);  // 8 waves x 32 queries
  hipLaunchKernelGGL((attn_map_kernel<0, QK8>), grid, dim3(512), lds0, st, p);  // also clears cmax
  hipLaunchKernelGGL((attn_map_kernel<1, QK8>), grid, dim3(512), lds1, st, p);
  hipLaunchKernelGGL(attn_map_delta_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, p.cmax, p.delta, nk, p.levels, p.eps);
  hipLaunchKernelGGL((attn_map_kernel<2, QK8>), grid, dim3(512), lds2, st, p);
  return check_launch(what);
}

}  // namespace wanq

using namespace wanq;

extern "C" int64_t wanq_attention_map_workspace(int64_t Lq, int64_t Lk, int heads) {
  return (2 * Lq + 3 * Lk) * (int64_t)heads * (int64_t)sizeof(float);
}

static int attn_map_common(const char* what, AttnMapParams& p, const void* v, void* o, int dtype, int64_t Lq, int64_t Lk, int heads, int head_dim,
                           int64_t v_stride, int64_t o_stride, float scale, int n_bits, int sym, void* workspace, int64_t workspace_bytes) {
  WANQ_REQUIRE(v && o, WANQ_E_ARG, "%s: NULL pointer", what);
  WANQ_REQUIRE(dtype == WANQ_BF16, WANQ_E_ARG, "%s: only bf16 is implemented (dtype code %d)", what, dtype);
  WANQ_REQUIRE(head_dim == 128, WANQ_E_SHAPE, "%s: head_dim=%d, only 128 is implemented", what, head_dim);
  WANQ_REQUIRE(heads >= 1 && heads <= 65535, WANQ_E_SHAPE, "%s: heads=%d out of range", what, heads);
  WANQ_REQUIRE(Lq >= 0 && Lk >= 1 && Lq < (1ll << 30) && Lk < (1ll << 30), WANQ_E_SHAPE, "%s: bad lengths", what);
  WANQ_REQUIRE(n_bits >= 2 && n_bits <= 8, WANQ_E_ARG, "%s: n_bits=%d must be 2..8", what, n_bits);
  const int64_t need = (int64_t)heads * head_dim;
  WANQ_REQUIRE(v_stride >= need && o_stride >= need, WANQ_E_SHAPE, "%s: token stride smaller than heads*head_dim", what);
  WANQ_REQUIRE(v_stride % 8 == 0 && o_stride % 4 == 0, WANQ_E_SHAPE, "%s: v stride must be a multiple of 8, o of 4 elements", what);
  const int64_t need_ws = wanq_attention_map_workspace(Lq, Lk, heads);
  WANQ_REQUIRE(workspace && workspace_bytes >= need_ws, WANQ_E_ARG, "%s: workspace of %lld bytes needed, %lld given", what, (long long)need_ws,
               (long long)workspace_bytes);
  WANQ_REQUIRE(((uintptr_t)workspace & 15) == 0, WANQ_E_ARG, "%s: workspace must be 16-byte aligned", what);
  p.v = (const uint16_t*)v; p.o = (uint16_t*)o;
  p.v_stride = v_stride; p.o_stride = o_stride;
  p.Lq = (int)Lq; p.Lk = (int)Lk; p.H = heads;
  p.c = scale * 1.4426950408889634f;
  p.levels = sym ? (float)((1 << (n_bits - 1)) - 1) : (float)((1 << n_bits) - 1);
  p.eps = sym ? 1e-6f : 1e-8f;
  p.m = static_cast<float*>(workspace);
  p.l = p.m + Lq * heads;
  p.cmax = p.l + Lq * heads;
  p.delta = p.cmax + Lk * heads;
  return WANQ_OK;
}

extern "C" int wanq_attention_map_quant_fwd(const void* q, const void* k, const void* v, void* o, int dtype, int64_t Lq, int64_t Lk, int heads,
                                            int head_dim, int64_t q_stride, int64_t k_stride, int64_t v_stride, int64_t o_stride, float scale,
                                            int n_bits, int sym, void* workspace, int64_t workspace_bytes, void* stream) {
  const char* what = "wanq_attention_map_quant_fwd";
  WANQ_REQUIRE(q && k, WANQ_E_ARG, "%s: NULL pointer", what);
  if (Lq == 0) return WANQ_OK;
  AttnMapParams p{};
  if (int e = attn_map_common(what, p, v, o, dtype, Lq, Lk, heads, head_dim, v_stride, o_stride, scale, n_bits, sym, workspace, workspace_bytes)) return e;
  const int64_t need = (int64_t)heads * head_dim;
  WANQ_REQUIRE(q_stride >= need && k_stride >= need && (q_stride | k_stride) % 8 == 0, WANQ_E_SHAPE,
               "%s: q / k strides must be >= heads*head_dim and multiples of 8 elements", what);
  p.q = (const uint16_t*)q; p.k = (const uint16_t*)k;
  p.q_stride = q_stride; p.k_stride = k_stride;
  return launch_attn_map<false>(p, (hipStream_t)stream, what);
}

// The reference's full quantised-attention recipe in one call (quant_opensora.py:431-476): q / k as per-(token, head) int8 codes
// + fp32 scale planes [heads][stride] (wanq_rmsnorm_rope_q8), v already fake-quantised by the caller, the map quantised per key.
extern "C" int wanq_attention_map_quant_qk8_fwd(const int8_t* q8, const float* q_scale, int64_t qs_stride, const int8_t* k8, const float* k_scale,
                                                int64_t ks_stride, const void* v, void* o, int dtype, int64_t Lq, int64_t Lk, int heads,
                                                int head_dim, int64_t q8_stride, int64_t k8_stride, int64_t v_stride, int64_t o_stride,
                                                float scale, int n_bits, int sym, void* workspace, int64_t workspace_bytes, void* stream) {
  const char* what = "wanq_attention_map_quant_qk8_fwd";
  WANQ_REQUIRE(q8 && k8 && q_scale && k_scale, WANQ_E_ARG, "%s: NULL pointer", what);
  if (Lq == 0) return WANQ_OK;
  AttnMapParams p{};
  if (int e = attn_map_common(what, p, v, o, dtype, Lq, Lk, heads, head_dim, v_stride, o_stride, scale, n_bits, sym, workspace, workspace_bytes)) return e;
  const int64_t need = (int64_t)heads * head_dim;
  WANQ_REQUIRE(q8_stride >= need && k8_stride >= need && (q8_stride | k8_stride) % 16 == 0, WANQ_E_SHAPE,
               "%s: code strides must be >= heads*head_dim and multiples of 16 bytes", what);
  WANQ_REQUIRE(qs_stride >= Lq && ks_stride >= ((Lk + AM_KB - 1) / AM_KB) * AM_KB && ks_stride % 4 == 0 && ((uintptr_t)k_scale & 15) == 0, WANQ_E_SHAPE,
               "%s: the key scale planes must be 16-byte aligned with a stride that is a multiple of 4 and covers whole 64-key tiles", what);
  p.q8 = q8; p.k8 = k8; p.q8_stride = q8_stride; p.k8_stride = k8_stride;
  p.q_scale = q_scale; p.k_scale = k_scale; p.qs_stride = qs_stride; p.ks_stride = ks_stride;
  return launch_attn_map<true>(p, (hipStream_t)stream, what);
}
