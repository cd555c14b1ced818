// What does a bare v_mfma_f32_32x32x16_bf16 stream sustain on this part (clock under load included)?
//   MODE 0: 4 independent accumulators, operands in registers, no other work
//   MODE 1: the attention dependency shape per "tile": 16 MFMAs on 2 accumulators (S), accumulators -> bf16 (cvt),
//           16 MFMAs on 4 accumulators with the converted values as B operand (PV)
//   MODE 2: v_mfma_f32_16x16x32_bf16, 16 independent accumulators, same FLOPs per iteration as MODE 0
// WAVES = waves per workgroup (4 = one per SIMD, 8 = two per SIMD); one workgroup per CU x 256 CUs x REPS rounds.
// Build: hipcc -O3 --offload-arch=gfx950 -DMODE=0 -DWAVES=8 mfma_bf16_peak.hip -o mfma_bf16_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#ifndef MODE
#define MODE 0
#endif
#ifndef WAVES
#define WAVES 8
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void peak_kernel(const float* in, float* out, int iters, unsigned long long* clk) {
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  const int lane = threadIdx.x & 63;
  bf16x8 a[8], b[4];
  for (int i = 0; i < 8; ++i)
    for (int e = 0; e < 8; ++e) a[i][e] = (__bf16)in[(lane + 8 * i + e) & 255];
  for (int i = 0; i < 4; ++i)
    for (int e = 0; e < 8; ++e) b[i][e] = (__bf16)in[(lane + 3 * i + e) & 255];
  f32x16 o[4];
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
#if MODE == 2
    // same FLOPs per iteration with v_mfma_f32_16x16x32_bf16: 64 MFMAs on 16 accumulators of 4 registers
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4* o4 = reinterpret_cast<f32x4*>(o);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 16; ++i) o4[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(s + i) & 7], b[i & 3], o4[i], 0, 0, 0);
#elif MODE == 0
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], b[i], o[i], 0, 0, 0);
#else
    f32x16 s0, s1;
    for (int r = 0; r < 16; ++r) { s0[r] = 0.f; s1[r] = 0.f; }
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], a[(s + 1) & 7], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + 2) & 7], a[(s + 1) & 7], s1, 0, 0, 0);
    }
    bf16x8 pf[4];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      pf[0][e] = (__bf16)s0[e];
      pf[1][e] = (__bf16)s0[8 + e];
      pf[2][e] = (__bf16)s1[e];
      pf[3][e] = (__bf16)s1[8 + e];
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks + i], pf[ks], o[i], 0, 0, 0);
#endif
  }
  float acc = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int r = 0; r < 16; ++r) acc += o[i][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

int main() {
  const int blocks = 256 * 4, iters = 4096;
  float *in, *out;
  hipMalloc(&in, 256 * 4);
  hipMalloc(&out, blocks * WAVES * 64 * 4);
  unsigned long long* clk;
  hipMalloc(&clk, 16);
  float h[256];
  for (int i = 0; i < 256; ++i) h[i] = (float)((i * 37) % 17 - 8) * 0.01f;
  hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(peak_kernel, dim3(blocks), dim3(WAVES * 64), 0, 0, in, out, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = 2.0 * 32 * 32 * 16 * 32.0 * iters * (double)blocks * WAVES;
    unsigned long long hc[2];
    hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    // workgroup 0: MFMA issue cycles it should have taken at 32 cycles per MFMA per SIMD vs its wall time (100 MHz counter)
    const double wg_us = hc[1] / 100.0, mfma_cycles = 32.0 * 32 * iters * (WAVES / 4);
    printf("MODE %d WAVES %d: %.3f ms  %.1f TFLOP/s (%.1f %% of 2.5 PF) | wg0: clock64 %llu, wall %.1f us -> >= %.0f MHz if the matrix pipe never idled\n",
           MODE, WAVES, ms, flops / ms * 1e-9, flops / ms * 1e-9 / 2500 * 100, hc[0], wg_us, mfma_cycles / wg_us);
  }
  return 0;
}
