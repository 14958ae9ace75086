// Micro-benchmark: one ds_add_u64 on an interleaved (re,im) cell versus two ds_add_u32 on separate planes, for the
// fixed-point Tx tile scatter (gfx950).  Index pattern = the kernel's: row k*17 + frame, with k near the lane's own bin.
// Build: hipcc -O3 --offload-arch=gfx950 lds_atomic64.hip -o lds_atomic64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define ITERS 256
template <int MODE>
__global__ void k(const int* idx, unsigned* out, long long* cyc) {
  __shared__ unsigned long long lds[513 * 17 + 16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 513 * 17; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  int my[8];
  for (int j = 0; j < 8; ++j) my[j] = idx[j * 64 + lane] * 17 + wave;   // cell index (row k, frame = wave)
  unsigned* planes = (unsigned*)lds;                                    // MODE 0: re plane [0, 8721), im plane after it
  __syncthreads();
  long long t0 = clock64();
#pragma unroll 1
  for (int it = 0; it < ITERS / 8; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int a = my[j];
      if (MODE == 0) {
        atomicAdd(&planes[a], (unsigned)lane);
        atomicAdd(&planes[513 * 17 + a], (unsigned)(lane + 1));
      } else {
        atomicAdd(&lds[a], ((unsigned long long)(lane + 1) << 32) | (unsigned)lane);
      }
    }
  }
  long long t1 = clock64();
  __syncthreads();
  if (lane == 0) cyc[wave] = t1 - t0;
  out[threadIdx.x] = (unsigned)lds[threadIdx.x];
}

int main() {
  std::vector<int> h(8 * 64);
  int* d_idx; unsigned* d_out; long long* d_cyc;
  hipMalloc(&d_idx, h.size() * 4); hipMalloc(&d_out, 1024 * 4); hipMalloc(&d_cyc, 16 * 8);
  const char* pat[] = {"own bin (k = lane + 64 j)", "own bin +- 2 (reassigned)", "runs of 8 equal bins", "random bins"};
  for (int p = 0; p < 4; ++p) {
    for (size_t i = 0; i < h.size(); ++i) {
      int lane = i & 63, j = (i >> 6);
      int kk = lane + 64 * j;
      if (p == 1) kk += (rand() % 5) - 2;
      if (p == 2) kk = (lane / 8) * 8 + 64 * j;
      if (p == 3) kk = rand() % 513;
      h[i] = kk < 0 ? 0 : (kk > 512 ? 512 : kk);
    }
    hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int m = 0; m < 2; ++m) {
      long long c[16];
      void (*fn)(const int*, unsigned*, long long*) = m == 0 ? k<0> : k<1>;
      hipLaunchKernelGGL(fn, dim3(1), dim3(1024), 0, 0, d_idx, d_out, d_cyc);
      hipLaunchKernelGGL(fn, dim3(1), dim3(1024), 0, 0, d_idx, d_out, d_cyc);
      hipDeviceSynchronize();
      hipMemcpy(c, d_cyc, 16 * 8, hipMemcpyDeviceToHost);
      double s = 0;
      for (int w = 0; w < 16; ++w) s += c[w];
      printf("%-28s %-22s %7.1f cycles per (re,im) scatter of a wave, 16 waves/CU\n", pat[p],
             m == 0 ? "2 x ds_add_u32 (planes)" : "1 x ds_add_u64 (cells)", s / 16 / ITERS);
    }
  }
  return 0;
}
