// Launch parameters shared by the int8 GEMM kernels (gemm_w8a8.hip: v1 / persistent v2; gemm_w8a8_pp.hip: ping-pong v3).
#pragma once
#include "wanq_common.h"

namespace wanq {

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
  int group_m;  // persistent kernels: m-tiles per L2 panel
  int coh_n, coh_ticks;  // ping-pong kernel: start-time stagger (number of cohorts, 10-ns ticks per cohort step)
};

// gemm_w8a8_pp.hip -- the ping-pong persistent kernel (W8 operands, M >= 512, K % 128 == 0, K >= 256; fp16 / bf16 / fp32 / int32
// output; gate + residual with an fp32 output only).  `eligible` says whether a problem may take it; `launch` fills mt / nt itself.
bool gemm_pp_eligible(const GemmParams& p, int out_dtype, bool w4);
int launch_gemm_pp(const GemmParams& p, int out_dtype, hipStream_t st);

}  // namespace wanq
