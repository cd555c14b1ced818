// Probe: the int8 GEMM main loop with SIXTEEN waves per workgroup (4 per SIMD, 128 VGPRs each): 256x256x128 tile, wave tile
// 64 x 64 (2 W fragments + 2 X fragments per k-step, 4 MFMAs), 4 LDS-DMA instructions per wave per K-tile.  Question: do four
// waves per SIMD hide the LDS-DMA issue cost that two waves per SIMD do not (mfma_lds_probe.hip: 2234 -> 3857 cycles per K-tile)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int MODE>  // 0 reads || mfma, 1 mfma only, 4 reads || mfma + 4 DMA per wave per K-tile at the top, 6 DMA spread 1 per k-step
__global__ __launch_bounds__(1024) void probe(int* out, long long* cyc, int nit, const char* gbuf, int gspan) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 131072 / 4; i += blockDim.x) ((int*)smem)[i] = i * 2654435761u;
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5, wm = wave & 3, wn = wave >> 2;
  const int fsw = (fr >> 1) & 7;
  int ck[4];
  for (int ks = 0; ks < 4; ++ks) ck[ks] = ((2 * ks + fh) ^ fsw) << 4;
  const int rowx = (wm * 64 + fr) * 128, roww = 32768 + (wn * 64 + fr) * 128;
  v16i acc[2][2];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
  v4i wf0[2], xf0[2], wf1[2], xf1[2];
#define FR(w, x, st, ks) { _Pragma("unroll") for (int i = 0; i < 2; ++i) w[i] = *(const v4i*)((st) + roww + ck[ks] + i * 4096); _Pragma("unroll") for (int j = 0; j < 2; ++j) x[j] = *(const v4i*)((st) + rowx + ck[ks] + j * 4096); }
#define MM(w, x) { _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[i], x[j], acc[i][j], 0, 0, 0); }
  FR(wf0, xf0, smem, 0)
  __syncthreads();
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  const char* gsrc = gbuf + tid * 16;
  char* dst = smem + 131072 + wave * 1024;
#define DMA1(q) __builtin_amdgcn_global_load_lds((glb_void*)(gsrc + (((size_t)blockIdx.x * 65536 + (size_t)(it * 4 + (q)) * 16384) % gspan)), (lds_void*)dst, 16, 0, 0)
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < nit; ++it) {
    const char* st = smem + (it & 1) * 65536;
    if (MODE == 4) { DMA1(0); DMA1(1); DMA1(2); DMA1(3); asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
    if (MODE == 6) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    if (MODE == 1) { MM(wf0, xf0) MM(wf0, xf0) MM(wf0, xf0) MM(wf0, xf0) }
    else {
      FR(wf1, xf1, st, 1) if (MODE == 6) DMA1(0); __builtin_amdgcn_sched_barrier(0); MM(wf0, xf0)
      FR(wf0, xf0, st, 2) if (MODE == 6) DMA1(1); __builtin_amdgcn_sched_barrier(0); MM(wf1, xf1)
      FR(wf1, xf1, st, 3) if (MODE == 6) DMA1(2); __builtin_amdgcn_sched_barrier(0); MM(wf0, xf0)
      FR(wf0, xf0, smem + ((it + 1) & 1) * 65536, 0) if (MODE == 6) DMA1(3); __builtin_amdgcn_sched_barrier(0); MM(wf1, xf1)
    }
    if (BARRIER) __syncthreads();
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * blockDim.x + tid] = s + wf0[0][0] + xf0[0][0];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  int* out; long long* cyc;
  hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
  const int nit = 200;
  char* gbuf; const int gspan = 1 << 20; hipMalloc(&gbuf, (size_t)gspan + (1 << 20)); hipMemset(gbuf, 1, (size_t)gspan + (1 << 20));
#define RUN(MODE, name) { hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 16384); \
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(1024), 131072 + 16384, 0, out, cyc, nit, gbuf, gspan); hipDeviceSynchronize(); \
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(1024), 131072 + 16384, 0, out, cyc, nit, gbuf, gspan); hipDeviceSynchronize(); \
    hipError_t e = hipGetLastError(); \
    std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost); double a = 0; for (auto v : h) a += v; \
    printf("%-52s %8.0f cycles per K-tile (ideal MFMA: 2048)  %s\n", name, a / 256 / nit, hipGetErrorString(e)); }
  RUN(0, "16 waves: reads(s+1) || mfma(s)")
  RUN(1, "16 waves: mfma only")
  RUN(4, "16 waves: reads || mfma + 4 LDS-DMA/wave at the top")
  RUN(6, "16 waves: reads || mfma + LDS-DMA spread 1 per k-step")
  return 0;
}
