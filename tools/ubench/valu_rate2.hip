// Micro-benchmark 2: SIMD cycles per wave64 instruction for the ops of the reassignment epilogue (gfx950),
// 2 waves per SIMD, 8 independent chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int MODE>
__global__ void k(float* out, long long* cyc, int iters, float cin) {
  float a[8]; int b[8];
  for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i + cin; b[i] = threadIdx.x * 3 + i; }
  const float c = 1.0001f + cin;
  unsigned long long m = 0x5555555555555555ull + (unsigned long long)cin;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (MODE == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c), "v"(a[(j + 3) & 7]));
        if (MODE == 2) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[j]) : "v"(c), "s"(m));
        if (MODE == 3) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(a[j]), "v"(c));
        if (MODE == 4) asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(b[j]) : "v"(a[j]));
        if (MODE == 5) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[j]));
        if (MODE == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[j]));
        if (MODE == 7) asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(b[j]) : "v"(b[(j + 1) & 7]));
        if (MODE == 8) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(b[j]) : "v"(b[(j + 1) & 7]));
        if (MODE == 9) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[j]) : "v"(c));
        if (MODE == 10) asm volatile("v_add_f32 %0, |%0|, |%1|" : "+v"(a[j]) : "v"(c));
        if (MODE == 11) asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(a[j]) : "v"(b[j]));
      }
    }
  }
  long long t1 = clock64();
  if ((threadIdx.x & 63) == 0) cyc[threadIdx.x >> 6] = t1 - t0;
  float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + b[i];
  out[threadIdx.x] = s + (float)m;
}
int main() {
  float* d_o; long long* d_c; hipMalloc(&d_o, 4096); hipMalloc(&d_c, 64 * 8);
  const char* names[] = {"v_mul_f32", "v_fma_f32 (3 regs)", "v_cndmask_b32_e64 (sgpr mask)", "v_cmp_lt_f32_e64 -> sgpr", "v_cvt_i32_f32",
                         "v_rndne_f32", "v_rcp_f32", "v_mad_u32_u24", "v_mul_lo_u32", "v_min_f32", "v_add_f32 |a|,|b|", "v_cvt_f32_i32"};
  const int iters = 1000, waves = 8;
  for (int m = 0; m < 12; ++m) {
    void (*fn)(float*, long long*, int, float) = m == 0 ? k<0> : m == 1 ? k<1> : m == 2 ? k<2> : m == 3 ? k<3> : m == 4 ? k<4> : m == 5 ? k<5>
        : m == 6 ? k<6> : m == 7 ? k<7> : m == 8 ? k<8> : m == 9 ? k<9> : m == 10 ? k<10> : k<11>;
    hipLaunchKernelGGL(fn, dim3(1), dim3(64 * waves), 0, 0, d_o, d_c, iters, 0.f);
    hipLaunchKernelGGL(fn, dim3(1), dim3(64 * waves), 0, 0, d_o, d_c, iters, 0.f);
    hipDeviceSynchronize();
    long long c[8]; hipMemcpy(c, d_c, waves * 8, hipMemcpyDeviceToHost);
    long long mx = 0; for (int w = 0; w < waves; ++w) mx = c[w] > mx ? c[w] : mx;
    printf("%-32s %.2f SIMD-cycles per instruction (2 waves/SIMD)\n", names[m], (double)mx / ((double)iters * 16 * 8) / 2);
  }
  return 0;
}
