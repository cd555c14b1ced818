// Does the MFMA SHAPE change what the part sustains under load (guide: 'DVFS give-back' item 7)?
// Bare register-only streams on pseudo-random operands, one binary, four kernels run back to back:
//   bf16 32x32x16 | bf16 16x16x32 | i8 32x32x32 | i8 16x16x64      (same FLOP / op per iteration and wave inside a dtype)
// WAVES per workgroup = 4 (one per SIMD) or 8 (two per SIMD); 1024 workgroups.
// Build: hipcc -O3 --offload-arch=gfx950 -DWAVES=8 mfma_shape_clock.hip -o mfma_shape_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#ifndef WAVES
#define WAVES 8
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void shape_kernel(const uint32_t* in, float* out, int iters, unsigned long long* clk) {
  const int lane = threadIdx.x & 63;
  uint4 ra[8], rb[4];
  for (int i = 0; i < 8; ++i) ra[i] = reinterpret_cast<const uint4*>(in)[(lane * 8 + i + 7 * blockIdx.x) & 1023];
  for (int i = 0; i < 4; ++i) rb[i] = reinterpret_cast<const uint4*>(in)[(lane * 4 + i + 512 + 3 * blockIdx.x) & 1023];
  const unsigned long long c0 = clock64(), w0 = wall_clock64();
  float acc = 0.f;
  if (MODE == 0) {
    f32x16 o[4];
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) o[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          o[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<bf16x8*>(&ra[s]), *reinterpret_cast<bf16x8*>(&rb[i]), o[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) acc += o[i][r];
  } else if (MODE == 1) {
    f32x4 o[16];
    for (int i = 0; i < 16; ++i)
      for (int r = 0; r < 4; ++r) o[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          o[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<bf16x8*>(&ra[(s + i) & 7]), *reinterpret_cast<bf16x8*>(&rb[i & 3]), o[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i)
      for (int r = 0; r < 4; ++r) acc += o[i][r];
  } else if (MODE == 2) {
    v16i o[4];
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) o[i][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          o[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(*reinterpret_cast<v4i*>(&ra[s]), *reinterpret_cast<v4i*>(&rb[i]), o[i], 0, 0, 0);
    }
    for (int i = 0; i < 4; ++i)
      for (int r = 0; r < 16; ++r) acc += (float)o[i][r];
  } else {
    v4i o[16];
    for (int i = 0; i < 16; ++i)
      for (int r = 0; r < 4; ++r) o[i][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 16; ++i)
          o[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(*reinterpret_cast<v4i*>(&ra[(s + i) & 7]), *reinterpret_cast<v4i*>(&rb[i & 3]), o[i], 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i)
      for (int r = 0; r < 4; ++r) acc += (float)o[i][r];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
  if (blockIdx.x == 300 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

static uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }

template <int MODE>
static void run(const char* name, const uint32_t* in, float* out, unsigned long long* clk, double work_per_mfma, int mfma_per_iter, int cyc) {
  const int blocks = 256 * 4, iters = 4096;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(shape_kernel<MODE>, dim3(blocks), dim3(WAVES * 64), 0, 0, in, out, iters, clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double work = work_per_mfma * mfma_per_iter * iters * (double)blocks * WAVES;
    unsigned long long hc[2];
    hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost);
    const double mhz = (double)hc[0] / ((double)hc[1] / 100.0);
    printf("%-14s WAVES %d: %7.3f ms  %7.1f T/s | wg300: %llu cycles for %d MFMAs per wave (%.1f per MFMA and SIMD, floor %d), clock %.0f MHz\n", name,
           WAVES, ms, work / ms * 1e-9, hc[0], mfma_per_iter * iters, (double)hc[0] / (mfma_per_iter * iters) / (WAVES / 4), cyc, mhz);
  }
}

int main() {
  uint32_t* in;
  float* out;
  unsigned long long* clk;
  hipMalloc(&in, 1024 * 16);
  hipMalloc(&out, 1024 * WAVES * 64 * 4);
  hipMalloc(&clk, 16);
  uint32_t* h = (uint32_t*)malloc(1024 * 16);
  uint32_t s = 12345;
  // bf16 pairs with exponents near 1.0 and random mantissas / signs (also fine as int8 bytes: every byte random)
  for (int i = 0; i < 4096; ++i) {
    const uint32_t r = lcg(s) >> 8, r2 = lcg(s) >> 8;
    const uint32_t lo = ((r & 0x8000u) | 0x3f00u | (r & 0xffu)) ^ ((r >> 8 & 3u) << 7);
    const uint32_t hi = ((r2 & 0x8000u) | 0x3f00u | (r2 & 0xffu)) ^ ((r2 >> 8 & 3u) << 7);
    h[i] = lo | (hi << 16);
  }
  hipMemcpy(in, h, 1024 * 16, hipMemcpyHostToDevice);
  for (int round = 0; round < 2; ++round) {
    run<0>("bf16 32x32x16", in, out, clk, 2.0 * 32 * 32 * 16, 32, 32);
    run<1>("bf16 16x16x32", in, out, clk, 2.0 * 16 * 16 * 32, 64, 16);
    run<2>("i8 32x32x32", in, out, clk, 2.0 * 32 * 32 * 32, 32, 32);
    run<3>("i8 16x16x64", in, out, clk, 2.0 * 16 * 16 * 64, 64, 16);
  }
  return 0;
}
