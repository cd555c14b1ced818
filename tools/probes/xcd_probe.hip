// Diagnostic: which XCD does block b land on?  (speed-only information; nothing in the product depends on it)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__global__ void probe(int* xcc, unsigned long long* t0) {
  if (threadIdx.x == 0) {
    int id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x] = id & 0xf;
    t0[blockIdx.x] = __builtin_amdgcn_s_memrealtime();
  }
  // keep the block alive a little so that the whole grid is co-resident
  unsigned long long s = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - s < 20000) {}
}
int main() {
  for (int threads : {256, 512}) {
    const int n = 768;
    int* d; unsigned long long* t;
    hipMalloc(&d, n * sizeof(int)); hipMalloc(&t, n * 8);
    hipLaunchKernelGGL(probe, dim3(n), dim3(threads), 65536, 0, d, t);
    hipDeviceSynchronize();
    std::vector<int> h(n); std::vector<unsigned long long> ht(n);
    hipMemcpy(h.data(), d, n * sizeof(int), hipMemcpyDeviceToHost);
    hipMemcpy(ht.data(), t, n * 8, hipMemcpyDeviceToHost);
    printf("threads=%d first 64 blocks xcc:", threads);
    for (int i = 0; i < 64; ++i) printf(" %d", h[i]);
    int mism = 0; for (int i = 0; i < n; ++i) mism += (h[i] != h[i % 8]);
    printf("\n  blocks whose xcc != xcc[b%%8]: %d of %d\n", mism, n);
    unsigned long long mn = ~0ull; for (auto v : ht) mn = v < mn ? v : mn;
    printf("  start time (x10ns) of blocks 0,8,..,: ");
    for (int i = 0; i < n; i += 64) printf(" b%d:%llu", i, ht[i] - mn);
    printf("\n");
  }
  hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
  printf("CUs=%d l2=%d clock=%d\n", pr.multiProcessorCount, pr.l2CacheSize, pr.clockRate);
  return 0;
}
