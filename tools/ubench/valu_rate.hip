// Micro-benchmark: VALU issue rates on gfx950 -- scalar f32 vs packed f32 ops, at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, long long* cyc, int iters) {
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
  const float c = 1.0001f;
  const f2 cc = {c, c};
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) {        // 8 independent v_add_f32
        asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                     "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
      } else if (MODE == 1) { // 4 independent v_pk_add_f32 (same flops as MODE 0)
        asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
      } else if (MODE == 2) { // 8 independent v_fma_f32
        asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n"
                     "v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
      } else if (MODE == 3) { // 4 independent v_pk_fma_f32
        asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(cc));
      } else if (MODE == 4) { // 8 v_mov_b32
        asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                     "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
      } else {                // 8 v_cndmask_b32 (VOP3 with SGPR pair condition)
        asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                     "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) :: "vcc");
      }
    }
  }
  long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

int main() {
  float* d_o; long long* d_c; hipMalloc(&d_o, 1024 * 4 * 64); hipMalloc(&d_c, 64 * 8 * 64);
  const char* names[] = {"8 x v_add_f32", "4 x v_pk_add_f32", "8 x v_fma_f32", "4 x v_pk_fma_f32", "8 x v_mov_b32", "8 x v_cndmask_b32"};
  const int iters = 2000;
  for (int waves = 4; waves <= 16; waves *= 2) {      // waves per block = per CU (1 block): 4 -> 1/SIMD, 8 -> 2/SIMD, 16 -> 4/SIMD
    for (int m = 0; m < 6; ++m) {
      void (*fn)(float*, long long*, int) = m == 0 ? k<0> : m == 1 ? k<1> : m == 2 ? k<2> : m == 3 ? k<3> : m == 4 ? k<4> : k<5>;
      hipLaunchKernelGGL(fn, dim3(1), dim3(64 * waves), 0, 0, d_o, d_c, iters);
      hipLaunchKernelGGL(fn, dim3(1), dim3(64 * waves), 0, 0, d_o, d_c, iters);
      hipDeviceSynchronize();
      long long c[16]; hipMemcpy(c, d_c, waves * 8, hipMemcpyDeviceToHost);
      long long mx = 0; for (int w = 0; w < waves; ++w) mx = c[w] > mx ? c[w] : mx;
      const double groups = (double)iters * 16;
      printf("waves/SIMD=%d  %-20s : %.2f cycles per group per wave -> %.2f SIMD-cycles per instruction\n", waves / 4, names[m],
             mx / groups, mx / groups / ((m == 1 || m == 3) ? 4 : 8) / (waves / 4));
    }
  }
  return 0;
}
