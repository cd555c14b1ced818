// Step-level elementwise work of the denoising loop as ONE kernel: classifier-free-guidance combine + scheduler update
// (ViDiT-Q/examples/Wan2.1/wan/text2video.py:260-269; wan/utils/fm_solvers_unipc.py:303-307,354-630 -- every line of the
// UniPC / DPM++ / Euler update is a linear combination of a handful of latent-sized tensors with scalar coefficients).
//   out[o][e] = sum_i coef[o][i] * in[i][e]        o < n_out <= 4,  i < n_in <= 8,  fp32 tensors of `numel` elements
// coef is HOST memory (fp32 [n_out][n_in], row-major) and travels BY VALUE in the kernel arguments: nothing the host does after
// the call can change what the launch computes.  (An earlier form kept it in device memory refreshed by an async copy from one
// pinned buffer per step; with the CPU a step ahead of the GPU the next step's coefficients overwrote the buffer before the copy
// ran.)  HBM-bound: (n_in + n_out) * 4 B per element.
#include "wanq_common.h"

namespace wanq {

struct LinParams {
  const float* in[8];
  float* out[4];
  float coef[32];
  int n_in, n_out;
  int64_t n4;  // numel / 4
};

__global__ __launch_bounds__(256) void lincomb_kernel(const LinParams p) {
  const float* c = p.coef;  // kernel arguments: scalar loads
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p.n4; i += (int64_t)gridDim.x * 256) {
    float4 acc[4];
#pragma unroll
    for (int o = 0; o < 4; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k < p.n_in) {
        const float4 v = reinterpret_cast<const float4*>(p.in[k])[i];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          if (o < p.n_out) {
            const float w = c[o * p.n_in + k];
            acc[o].x = fmaf(w, v.x, acc[o].x); acc[o].y = fmaf(w, v.y, acc[o].y);
            acc[o].z = fmaf(w, v.z, acc[o].z); acc[o].w = fmaf(w, v.w, acc[o].w);
          }
        }
      }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
      if (o < p.n_out) reinterpret_cast<float4*>(p.out[o])[i] = acc[o];
  }
}

}  // namespace wanq

using namespace wanq;

extern "C" int wanq_lincomb(int n_out, int n_in, const float* coef, const float* const* in, float* const* out, int64_t numel,
                            void* stream) {
  WANQ_REQUIRE(n_out >= 1 && n_out <= 4 && n_in >= 1 && n_in <= 8, WANQ_E_ARG, "wanq_lincomb: n_out=%d (1..4), n_in=%d (1..8)", n_out, n_in);
  WANQ_REQUIRE(coef && in && out, WANQ_E_ARG, "wanq_lincomb: NULL pointer");
  WANQ_REQUIRE(numel >= 0 && numel % 4 == 0, WANQ_E_SHAPE, "wanq_lincomb: numel=%lld must be a multiple of 4", (long long)numel);
  if (numel == 0) return WANQ_OK;
  LinParams p{};
  for (int i = 0; i < n_in; ++i) {
    WANQ_REQUIRE(in[i] && ((uintptr_t)in[i] & 15) == 0, WANQ_E_ARG, "wanq_lincomb: input %d is NULL or not 16-byte aligned", i);
    p.in[i] = in[i];
  }
  for (int o = 0; o < n_out; ++o) {
    WANQ_REQUIRE(out[o] && ((uintptr_t)out[o] & 15) == 0, WANQ_E_ARG, "wanq_lincomb: output %d is NULL or not 16-byte aligned", o);
    p.out[o] = out[o];
  }
  for (int k = 0; k < n_out * n_in; ++k) p.coef[k] = coef[k];
  p.n_in = n_in; p.n_out = n_out; p.n4 = numel / 4;
  const int64_t blocks = (p.n4 + 255) / 256;
  hipLaunchKernelGGL(lincomb_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream, p);
  return check_launch("wanq_lincomb");
}
