// Micro-victim for profiles/r04_z_corun_corruption.txt: does a register-returning load's data always sit in its VGPRs when
// s_waitcnt says the load is complete?  One wave per 2-KB table row, each lane two 16-byte loads A (floats 0..3) and B (4..7) of its
// 32 bytes, exactly like the RoPE stage of rmsnorm_rope_kernel.  The four registers of A are copied by v_mov RIGHT BEHIND the wait
// the variant uses, and the copies are stored; the host compares them with the table.
//   MODE 0: A, B issued, s_waitcnt vmcnt(1) (what hipcc emits), copy A      MODE 1: A alone, s_waitcnt vmcnt(0), copy A
//   MODE 2: A, B issued, s_waitcnt vmcnt(0), copy A                          MODE 3: as 0, but the copy is taken twice, 8 v_nop apart
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int MODE>
__global__ __launch_bounds__(256) void victim_kernel(const float* table, float* early, float* second, int rows) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = table + ((size_t)row * 64 + lane) * 8;
  typedef float v4f __attribute__((ext_vector_type(4)));
  v4f a, b = {0.f, 0.f, 0.f, 0.f};
  float e0, e1, e2, e3, s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  // sentinel in the destination registers first, so that "not landed yet" reads as the sentinel
  a = (v4f){-7.f, -7.f, -7.f, -7.f};
  asm volatile("" : "+v"(a));
  if (MODE == 1) {
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "+v"(a) : "v"(p) : "memory");
  } else if (MODE == 2) {
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\ts_waitcnt vmcnt(0)" : "+v"(a), "+v"(b) : "v"(p) : "memory");
  } else {
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16\n\ts_waitcnt vmcnt(1)" : "+v"(a), "+v"(b) : "v"(p) : "memory");
  }
  float ax = a.x, ay = a.y, az = a.z, aw = a.w;
  asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
               : "=&v"(e0), "=&v"(e1), "=&v"(e2), "=&v"(e3) : "v"(ax), "v"(ay), "v"(az), "v"(aw));
  if (MODE == 3) {
    asm volatile("s_nop 7\n\tv_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7"
                 : "=&v"(s0), "=&v"(s1), "=&v"(s2), "=&v"(s3) : "v"(ax), "v"(ay), "v"(az), "v"(aw));
  }
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(b));
  float* o = early + ((size_t)row * 64 + lane) * 8;
  *reinterpret_cast<float4*>(o) = make_float4(e0, e1, e2, e3);
  *reinterpret_cast<float4*>(o + 4) = make_float4(b.x, b.y, b.z, b.w);
  if (MODE == 3) *reinterpret_cast<float4*>(second + ((size_t)row * 64 + lane) * 4) = make_float4(s0, s1, s2, s3);
}

// MODE 4: the instruction sequence of rmsnorm_rope_kernel's RoPE stage itself, registers hard-coded as hipcc allocated them: the two
// loads, the overwrite of the address register, s_waitcnt vmcnt(1), the two packed multiplies with their op_sel forms (multipliers
// 1.0, so the products are the table's values: v20 = A.y, v21 = A.x, v22 = A.w, v23 = A.z)
template <int VAR>
__global__ __launch_bounds__(256) void victim_pk_kernel(const float* table, float* early, int rows) {
  const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* p = table + ((size_t)row * 64 + lane) * 8;
  float o0, o1, o2, o3, b0, b1, b2, b3;
#define PK_HEAD                                                                                              \
  "v_mov_b32 v0, 1.0\n\tv_mov_b32 v1, 1.0\n\tv_mov_b32 v3, 1.0\n\tv_mov_b32 v2, 1.0\n\t"                      \
  "v_mov_b32 v10, 0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, 0\n\tv_mov_b32 v13, 0\n\t"                          \
  "v_mov_b32 v8, %8\n\tv_mov_b32 v9, %9\n\t"                                                                   \
  "s_nop 4\n\t"                                                                                                 \
  "global_load_dwordx4 v[10:13], v[8:9], off\n\t"                                                               \
  "global_load_dwordx4 v[26:29], v[8:9], off offset:16\n\t"
#define PK_TAIL                                                                                              \
  "s_waitcnt vmcnt(0)\n\t"                                                                                      \
  "v_mov_b32 %0, v20\n\tv_mov_b32 %1, v21\n\tv_mov_b32 %2, v22\n\tv_mov_b32 %3, v23\n\t"                      \
  "v_mov_b32 %4, v26\n\tv_mov_b32 %5, v27\n\tv_mov_b32 %6, v28\n\tv_mov_b32 %7, v29"
#define PK_MUL0 "v_pk_mul_f32 v[20:21], v[0:1], v[10:11] op_sel:[1,1] op_sel_hi:[1,0]\n\t"
#define PK_MUL1 "v_pk_mul_f32 v[22:23], v[8:9], v[12:13] op_sel:[0,1] op_sel_hi:[0,0]\n\t"
#define PK_MUL1_V2 "v_pk_mul_f32 v[22:23], v[2:3], v[12:13] op_sel:[0,1] op_sel_hi:[0,0]\n\t"
#define PK_IO                                                                                                \
  : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)                   \
  : "v"((uint32_t)(uintptr_t)p), "v"((uint32_t)((uintptr_t)p >> 32))                                        \
  : "memory", "v0", "v1", "v2", "v3", "v8", "v9", "v10", "v11", "v12", "v13", "v20", "v21", "v22", "v23", "v26", "v27", "v28", "v29"
  if (VAR == 0) asm volatile(PK_HEAD "v_mov_b32 v8, v3\n\ts_waitcnt vmcnt(1)\n\t" PK_MUL0 PK_MUL1 PK_TAIL PK_IO);
  else if (VAR == 1) asm volatile(PK_HEAD "v_mov_b32 v8, v3\n\ts_waitcnt vmcnt(1)\n\ts_nop 3\n\t" PK_MUL0 PK_MUL1 PK_TAIL PK_IO);
  else if (VAR == 2) asm volatile(PK_HEAD "s_waitcnt vmcnt(1)\n\t" PK_MUL0 PK_MUL1_V2 PK_TAIL PK_IO);
  else if (VAR == 3) asm volatile(PK_HEAD "v_mov_b32 v8, v3\n\ts_waitcnt vmcnt(1)\n\t" PK_MUL1 PK_MUL0 PK_TAIL PK_IO);
  else if (VAR == 4)  // destination registers pre-filled with 5.0 instead of 0: a stale read shows as 5.0
    asm volatile("v_mov_b32 v0, 1.0\n\tv_mov_b32 v1, 1.0\n\tv_mov_b32 v3, 1.0\n\tv_mov_b32 v2, 1.0\n\t"
                 "v_mov_b32 v10, 0x40a00000\n\tv_mov_b32 v11, 0x40a00000\n\tv_mov_b32 v12, 0x40a00000\n\tv_mov_b32 v13, 0x40a00000\n\t"
                 "v_mov_b32 v8, %8\n\tv_mov_b32 v9, %9\n\ts_nop 4\n\t"
                 "global_load_dwordx4 v[10:13], v[8:9], off\n\tglobal_load_dwordx4 v[26:29], v[8:9], off offset:16\n\t"
                 "v_mov_b32 v8, v3\n\ts_waitcnt vmcnt(1)\n\t" PK_MUL0 PK_MUL1 PK_TAIL PK_IO);
  else if (VAR == 6)  // the RESULT registers pre-filled with 9.0: a lost write shows as 9.0
    asm volatile("v_mov_b32 v22, 0x41100000\n\tv_mov_b32 v23, 0x41100000\n\tv_mov_b32 v20, 0x41100000\n\tv_mov_b32 v21, 0x41100000\n\t"
                 PK_HEAD "v_mov_b32 v8, v3\n\ts_waitcnt vmcnt(1)\n\t" PK_MUL0 PK_MUL1 PK_TAIL PK_IO);
  else if (VAR == 7)  // two plain v_mul_f32 in place of the second packed multiply
    asm volatile("v_mov_b32 v22, 0x41100000\n\tv_mov_b32 v23, 0x41100000\n\t"
                 PK_HEAD "v_mov_b32 v8, v3\n\ts_waitcnt vmcnt(1)\n\t" PK_MUL0 "v_mul_f32 v22, v8, v13\n\tv_mul_f32 v23, v8, v12\n\t" PK_TAIL PK_IO);
  else if (VAR == 8)  // the packed multiply WITHOUT op_sel: v22 = v8 * v12, v23 = v9 * v13 (v9 = 1.0 too)
    asm volatile(PK_HEAD "v_mov_b32 v8, v3\n\tv_mov_b32 v9, v3\n\ts_waitcnt vmcnt(1)\n\t" PK_MUL0
                 "v_pk_mul_f32 v[22:23], v[8:9], v[12:13]\n\tv_mov_b32 v9, v22\n\tv_mov_b32 v22, v23\n\tv_mov_b32 v23, v9\n\t" PK_TAIL PK_IO);
  else if (VAR == 9)  // v_pk_fma_f32 without op_sel on the same pair: v[22:23] = v[8:9] * v[12:13] + 0
    asm volatile(PK_HEAD "v_mov_b32 v8, v3\n\tv_mov_b32 v9, v3\n\tv_mov_b32 v14, 0\n\tv_mov_b32 v15, 0\n\ts_waitcnt vmcnt(1)\n\t" PK_MUL0
                 "v_pk_fma_f32 v[22:23], v[8:9], v[12:13], v[14:15]\n\tv_mov_b32 v9, v22\n\tv_mov_b32 v22, v23\n\tv_mov_b32 v23, v9\n\t" PK_TAIL
                 : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(b0), "=&v"(b1), "=&v"(b2), "=&v"(b3)
                 : "v"((uint32_t)(uintptr_t)p), "v"((uint32_t)((uintptr_t)p >> 32))
                 : "memory", "v0", "v1", "v2", "v3", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v20", "v21", "v22", "v23", "v26", "v27", "v28", "v29");
  else  // a plain v_mov of v13 right BEHIND the packed multiply that reads it: returned in place of the B load's first register
    asm volatile(PK_HEAD "v_mov_b32 v8, v3\n\ts_waitcnt vmcnt(1)\n\t" PK_MUL0 PK_MUL1 "v_mov_b32 v21, v13\n\t" PK_TAIL PK_IO);

  float* o = early + ((size_t)row * 64 + lane) * 8;
  *reinterpret_cast<float4*>(o) = make_float4(o1, o0, o3, o2);  // back in table order A.x A.y A.z A.w
  *reinterpret_cast<float4*>(o + 4) = make_float4(b0, b1, b2, b3);
}

extern "C" int late_beat_victim(int mode, const float* table, float* early, float* second, int rows, void* stream) {
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL(victim_kernel<0>, grid, block, 0, st, table, early, second, rows); break;
    case 1: hipLaunchKernelGGL(victim_kernel<1>, grid, block, 0, st, table, early, second, rows); break;
    case 2: hipLaunchKernelGGL(victim_kernel<2>, grid, block, 0, st, table, early, second, rows); break;
    case 4: hipLaunchKernelGGL(victim_pk_kernel<0>, grid, block, 0, st, table, early, rows); break;
    case 5: hipLaunchKernelGGL(victim_pk_kernel<1>, grid, block, 0, st, table, early, rows); break;
    case 6: hipLaunchKernelGGL(victim_pk_kernel<2>, grid, block, 0, st, table, early, rows); break;
    case 7: hipLaunchKernelGGL(victim_pk_kernel<3>, grid, block, 0, st, table, early, rows); break;
    case 8: hipLaunchKernelGGL(victim_pk_kernel<4>, grid, block, 0, st, table, early, rows); break;
    case 9: hipLaunchKernelGGL(victim_pk_kernel<5>, grid, block, 0, st, table, early, rows); break;
    case 10: hipLaunchKernelGGL(victim_pk_kernel<6>, grid, block, 0, st, table, early, rows); break;
    case 11: hipLaunchKernelGGL(victim_pk_kernel<7>, grid, block, 0, st, table, early, rows); break;
    case 12: hipLaunchKernelGGL(victim_pk_kernel<8>, grid, block, 0, st, table, early, rows); break;
    case 13: hipLaunchKernelGGL(victim_pk_kernel<9>, grid, block, 0, st, table, early, rows); break;
    default: hipLaunchKernelGGL(victim_kernel<3>, grid, block, 0, st, table, early, second, rows); break;
  }
  return (int)hipGetLastError();
}
