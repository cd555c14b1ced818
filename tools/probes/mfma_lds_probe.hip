// Probe: does a wave's ds_read_b128 stream overlap with its (and its SIMD partner's) int8 MFMAs?
// Mimics the GEMM main loop on one CU-resident workgroup: LDS preloaded once, NIT "K-tiles" of 4 k-steps,
// each k-step = 6 ds_read_b128 (next fragments) + 8 v_mfma_i32_32x32x32_i8.  Reports cycles per K-tile.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int MODE>  // 0 both, 1 mfma only, 2 reads only, 3 both but reads right before use (no double buffer)
__global__ __launch_bounds__(512, 2) void probe(int* out, long long* cyc, int nit, const char* gbuf, int gspan) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 131072 / 4; i += blockDim.x) ((int*)smem)[i] = i * 2654435761u;
  __syncthreads();
  const int fr = lane & 31, fh = lane >> 5, wm = wave & 1, wn = wave >> 1;
  const int fsw = (fr >> 1) & 7;
  int ck[4];
  for (int ks = 0; ks < 4; ++ks) ck[ks] = ((2 * ks + fh) ^ fsw) << 4;
  const int rowx = (wm * 128 + fr) * 128, roww = 32768 + (wn * 64 + fr) * 128;
  v16i acc[2][4];
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;
  v4i wf0[2], xf0[4], wf1[2], xf1[4];
  v4i wga[8], wgb[8];
  for (int q = 0; q < 8; ++q) { wga[q] = v4i{q, lane, 1, 2}; wgb[q] = v4i{lane, q, 3, 4}; }
#define FR(w, x, st, ks) { _Pragma("unroll") for (int i = 0; i < 2; ++i) w[i] = *(const v4i*)((st) + roww + ck[ks] + i * 4096); _Pragma("unroll") for (int j = 0; j < 4; ++j) x[j] = *(const v4i*)((st) + rowx + ck[ks] + j * 4096); }
#define MM(w, x) { if (MODE != 2) { _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[i], x[j], acc[i][j], 0, 0, 0); } else { _Pragma("unroll") for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(w[i])); _Pragma("unroll") for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(x[j])); } }
  FR(wf0, xf0, smem, 0)
  __syncthreads();
  typedef __attribute__((address_space(3))) void lds_void;
  typedef __attribute__((address_space(1))) const void glb_void;
  const char* gsrc = gbuf + tid * 16;  // every address below stays inside [0, gspan + 8 KiB) (allocation: gspan + 1 MiB)
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < nit; ++it) {
    const char* st; st = smem + (it & 1) * 65536;
    if (MODE == 4 || MODE == 5) {  // 8 LDS-DMA instructions per wave per K-tile into the OTHER stage (here: a scratch region), one tile ahead
      char* dst = smem + 131072 + wave * 1024;
      _Pragma("unroll") for (int q = 0; q < 8; ++q)
        __builtin_amdgcn_global_load_lds((glb_void*)(gsrc + (((size_t)blockIdx.x * 65536 + (size_t)(it * 8 + q) * 8192) % gspan)), (lds_void*)(dst + (q & 1) * 8192), 16, 0, 0);
      if (MODE == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    if (MODE == 1 || MODE == 5) { if (MODE == 5) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); MM(wf0, xf0) MM(wf0, xf0) MM(wf0, xf0) MM(wf0, xf0) }
    else if (MODE == 3) {
      FR(wf0, xf0, st, 0) MM(wf0, xf0) FR(wf0, xf0, st, 1) MM(wf0, xf0) FR(wf0, xf0, st, 2) MM(wf0, xf0) FR(wf0, xf0, st, 3) MM(wf0, xf0)
    } else if (MODE == 6 || MODE == 8) {
      char* dst = smem + 131072 + wave * 1024;
#define DMA2(q0) { _Pragma("unroll") for (int q = q0; q < q0 + 2; ++q) __builtin_amdgcn_global_load_lds((glb_void*)(gsrc + (((size_t)blockIdx.x * 65536 + (size_t)(it * 8 + q) * 8192) % gspan)), (lds_void*)(dst + (q & 1) * 8192), 16, 0, 0); }
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      FR(wf1, xf1, st, 1) DMA2(0) __builtin_amdgcn_sched_barrier(0); if (MODE == 8) __builtin_amdgcn_s_setprio(1); MM(wf0, xf0) if (MODE == 8) __builtin_amdgcn_s_setprio(0);
      FR(wf0, xf0, st, 2) DMA2(2) __builtin_amdgcn_sched_barrier(0); if (MODE == 8) __builtin_amdgcn_s_setprio(1); MM(wf1, xf1) if (MODE == 8) __builtin_amdgcn_s_setprio(0);
      FR(wf1, xf1, st, 3) DMA2(4) __builtin_amdgcn_sched_barrier(0); if (MODE == 8) __builtin_amdgcn_s_setprio(1); MM(wf0, xf0) if (MODE == 8) __builtin_amdgcn_s_setprio(0);
      FR(wf0, xf0, smem + ((it + 1) & 1) * 65536, 0) DMA2(6) __builtin_amdgcn_sched_barrier(0); if (MODE == 8) __builtin_amdgcn_s_setprio(1); MM(wf1, xf1) if (MODE == 8) __builtin_amdgcn_s_setprio(0);
    } else if (MODE == 9 || MODE == 10) {
      char* dst = smem + 131072 + wave * 1024;
#define DMA8() { _Pragma("unroll") for (int q = 0; q < 8; ++q) __builtin_amdgcn_global_load_lds((glb_void*)(gsrc + (((size_t)blockIdx.x * 65536 + (size_t)(it * 8 + q) * 8192) % gspan)), (lds_void*)(dst + (q & 1) * 8192), 16, 0, 0); }
      const bool early = (MODE == 9) ? (wave < 4) : ((wave & 1) == 0);
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      if (early) DMA8()
      FR(wf1, xf1, st, 1) __builtin_amdgcn_sched_barrier(0); MM(wf0, xf0)
      FR(wf0, xf0, st, 2) __builtin_amdgcn_sched_barrier(0); MM(wf1, xf1)
      if (!early) DMA8()
      FR(wf1, xf1, st, 3) __builtin_amdgcn_sched_barrier(0); MM(wf0, xf0)
      FR(wf0, xf0, smem + ((it + 1) & 1) * 65536, 0) __builtin_amdgcn_sched_barrier(0); MM(wf1, xf1)
    } else if (MODE == 11 || MODE == 12) {
      // register-staged: 8 x global_load_dwordx4 issued at the top, written to LDS (ds_write_b128) after the MFMAs;
      // MODE 12: half by LDS-DMA (4), half register-staged (4)
      char* dst = smem + 131072 + wave * 1024 + lane * 16;
      char* dstd = smem + 131072 + wave * 1024;
      uint4 rg[8];
      const int nreg = (MODE == 11) ? 8 : 4;
      _Pragma("unroll") for (int q = 0; q < 8; ++q) {
        const char* src = gsrc + (((size_t)blockIdx.x * 65536 + (size_t)(it * 8 + q) * 8192) % gspan);
        if (q < nreg) rg[q] = *(const uint4*)src;
        else __builtin_amdgcn_global_load_lds((glb_void*)src, (lds_void*)(dstd + (q & 1) * 8192), 16, 0, 0);
      }
      FR(wf1, xf1, st, 1) __builtin_amdgcn_sched_barrier(0); MM(wf0, xf0)
      FR(wf0, xf0, st, 2) __builtin_amdgcn_sched_barrier(0); MM(wf1, xf1)
      FR(wf1, xf1, st, 3) __builtin_amdgcn_sched_barrier(0); MM(wf0, xf0)
      FR(wf0, xf0, smem + ((it + 1) & 1) * 65536, 0) __builtin_amdgcn_sched_barrier(0); MM(wf1, xf1)
      _Pragma("unroll") for (int q = 0; q < 8; ++q) if (q < nreg) *(uint4*)(dst + (q & 1) * 8192) = rg[q];
    } else if (MODE == 13 || MODE == 14 || MODE == 15) {
      // MODE 13: the W fragments of the NEXT K-tile straight from global memory into registers (8 global_load_dwordx4 per wave, the
      // two waves of a weight-row group read the same addresses), X by 4 LDS-DMA per wave; no W reads from LDS.
      // MODE 14: 4 LDS-DMA per wave only (X), W fragments from LDS as in mode 0 (= half the ingest, nothing else changed).
      // MODE 15: the 8 register loads only (no LDS-DMA), W fragments from them.
      char* dst = smem + 131072 + wave * 1024;
      if (MODE != 15) {
        _Pragma("unroll") for (int q = 0; q < 4; ++q)
          __builtin_amdgcn_global_load_lds((glb_void*)(gsrc + (((size_t)blockIdx.x * 65536 + (size_t)(it * 8 + q) * 8192) % gspan)), (lds_void*)(dst + (q & 1) * 8192), 16, 0, 0);
      }
      if (MODE == 14) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        FR(wf1, xf1, st, 1) __builtin_amdgcn_sched_barrier(0); MM(wf0, xf0)
        FR(wf0, xf0, st, 2) __builtin_amdgcn_sched_barrier(0); MM(wf1, xf1)
        FR(wf1, xf1, st, 3) __builtin_amdgcn_sched_barrier(0); MM(wf0, xf0)
        FR(wf0, xf0, smem + ((it + 1) & 1) * 65536, 0) __builtin_amdgcn_sched_barrier(0); MM(wf1, xf1)
      } else {
        const char* wsrc; wsrc = gbuf + (((size_t)blockIdx.x * 65536 + (size_t)it * 32768) % gspan) + wn * 8192 + lane * 16;
#define FRX(x, st, ks) { _Pragma("unroll") for (int j = 0; j < 4; ++j) x[j] = *(const v4i*)((st) + rowx + ck[ks] + j * 4096); }
#define WLD(dstw) { _Pragma("unroll") for (int q = 0; q < 8; ++q) dstw[q] = *(const v4i*)(wsrc + q * 1024); }
#define MMW(w, k, x) { _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(w[2 * (k) + i], x[j], acc[i][j], 0, 0, 0); }
        // two K-tiles per trip, so that the two register sets alternate without a runtime select
        WLD(wgb)
        FRX(xf1, st, 1) __builtin_amdgcn_sched_barrier(0); MMW(wga, 0, xf0)
        FRX(xf0, st, 2) __builtin_amdgcn_sched_barrier(0); MMW(wga, 1, xf1)
        FRX(xf1, st, 3) __builtin_amdgcn_sched_barrier(0); MMW(wga, 2, xf0)
        FRX(xf0, smem + ((it + 1) & 1) * 65536, 0) __builtin_amdgcn_sched_barrier(0); MMW(wga, 3, xf1)
        if (BARRIER) __syncthreads();
        ++it;
        st = smem + (it & 1) * 65536;
        wsrc = gbuf + (((size_t)blockIdx.x * 65536 + (size_t)it * 32768) % gspan) + wn * 8192 + lane * 16;
        if (MODE != 15) {
          _Pragma("unroll") for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_global_load_lds((glb_void*)(gsrc + (((size_t)blockIdx.x * 65536 + (size_t)(it * 8 + q) * 8192) % gspan)), (lds_void*)(dst + (q & 1) * 8192), 16, 0, 0);
        }
        WLD(wga)
        FRX(xf1, st, 1) __builtin_amdgcn_sched_barrier(0); MMW(wgb, 0, xf0)
        FRX(xf0, st, 2) __builtin_amdgcn_sched_barrier(0); MMW(wgb, 1, xf1)
        FRX(xf1, st, 3) __builtin_amdgcn_sched_barrier(0); MMW(wgb, 2, xf0)
        FRX(xf0, smem + ((it + 1) & 1) * 65536, 0) __builtin_amdgcn_sched_barrier(0); MMW(wgb, 3, xf1)
      }
    } else if (MODE == 7) {
      if (wave < 4) { char* dst = smem + 131072 + wave * 1024; _Pragma("unroll") for (int q = 0; q < 16; ++q) __builtin_amdgcn_global_load_lds((glb_void*)(gsrc + (((size_t)blockIdx.x * 65536 + (size_t)(it * 16 + q) * 4096) % gspan)), (lds_void*)(dst + (q & 3) * 4096), 16, 0, 0); asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
      FR(wf1, xf1, st, 1) __builtin_amdgcn_sched_barrier(0); MM(wf0, xf0)
      FR(wf0, xf0, st, 2) __builtin_amdgcn_sched_barrier(0); MM(wf1, xf1)
      FR(wf1, xf1, st, 3) __builtin_amdgcn_sched_barrier(0); MM(wf0, xf0)
      FR(wf0, xf0, smem + ((it + 1) & 1) * 65536, 0) __builtin_amdgcn_sched_barrier(0); MM(wf1, xf1)
    } else {
      FR(wf1, xf1, st, 1) __builtin_amdgcn_sched_barrier(0); MM(wf0, xf0)
      FR(wf0, xf0, st, 2) __builtin_amdgcn_sched_barrier(0); MM(wf1, xf1)
      FR(wf1, xf1, st, 3) __builtin_amdgcn_sched_barrier(0); MM(wf0, xf0)
      FR(wf0, xf0, smem + ((it + 1) & 1) * 65536, 0) __builtin_amdgcn_sched_barrier(0); MM(wf1, xf1)
    }
    if (BARRIER) __syncthreads();
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  out[blockIdx.x * blockDim.x + tid] = s + wf0[0][0] + xf0[0][0] + wga[0][0] + wgb[7][3];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  int* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  const int nit = 200;
  char* gbuf; const int gspan = 1 << 20; hipMalloc(&gbuf, (size_t)gspan + (1 << 20)); hipMemset(gbuf, 1, (size_t)gspan + (1 << 20));
#define RUN(MODE, threads, name) { hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072 + 16384); \
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 131072 + 16384, 0, out, cyc, nit, gbuf, gspan); hipDeviceSynchronize(); \
    hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(threads), 131072 + 16384, 0, out, cyc, nit, gbuf, gspan); hipDeviceSynchronize(); \
    std::vector<long long> h(256); hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost); double a = 0; for (auto v : h) a += v; \
    printf("%-44s threads=%d  %8.0f cycles per K-tile (ideal MFMA: %d)\n", name, threads, a / 256 / nit, threads == 512 ? 2048 : 1024); }
  RUN(0, 512, "reads(s+1) || mfma(s), 2 waves/SIMD")
  RUN(1, 512, "mfma only, 2 waves/SIMD")
  RUN(2, 512, "reads only, 2 waves/SIMD")
  RUN(3, 512, "reads right before use, 2 waves/SIMD")
  RUN(4, 512, "reads || mfma + 8 LDS-DMA/wave/tile (64 MB span)")
  RUN(5, 512, "mfma only + 8 LDS-DMA/wave/tile")
  RUN(6, 512, "DMA spread 2 per k-step")
  RUN(8, 512, "DMA spread 2 per k-step + setprio(1) on MFMA")
  RUN(7, 512, "DMA by waves 0..3 only (16 each)")
  RUN(11, 512, "register-staged: 8 global_load_dwordx4 + ds_write_b128")
  RUN(12, 512, "half LDS-DMA (4) + half register-staged (4)")
  RUN(14, 512, "X only by LDS-DMA (4/wave), W as if resident")
  RUN(13, 512, "W: global -> registers (8 loads/wave), X: 4 LDS-DMA/wave")
  RUN(15, 512, "W: global -> registers only (no LDS-DMA)")
  RUN(9, 512, "DMA: waves 0-3 at k-step 0, waves 4-7 at k-step 2")
  RUN(10, 512, "DMA: even waves at k-step 0, odd waves at k-step 2")
  RUN(0, 256, "reads(s+1) || mfma(s), 1 wave/SIMD")
  RUN(1, 256, "mfma only, 1 wave/SIMD")
  RUN(2, 256, "reads only, 1 wave/SIMD")
  return 0;
}
