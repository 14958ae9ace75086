// Micro-benchmark: cost of LDS operations relevant to the reassignment scatter (gfx950).
// Build: hipcc -O3 --offload-arch=gfx950 lds_atomics.hip -o lds_atomics ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define ITERS 256
template <int MODE>
__global__ void k(const int* idx, float* out, long long* cyc, int active) {
  __shared__ float lds[64 * 1024 / 4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  int my[8];
  for (int j = 0; j < 8; ++j) my[j] = idx[(wave * 8 + j) * 64 + lane] & 16383;
  float v = 1.0f + lane;
  unsigned* ul = (unsigned*)lds;
  __syncthreads();
  long long t0 = clock64();
  if (lane < active) {
#pragma unroll 1
    for (int it = 0; it < ITERS / 8; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int a = my[j];
        if (MODE == 0) atomicAdd(&lds[a], v);                               // ds_add_f32
        else if (MODE == 1) atomicAdd(&ul[a], (unsigned)lane);              // ds_add_u32
        else if (MODE == 2) lds[a] = v;                                     // ds_write_b32
        else if (MODE == 3) v += lds[a];                                    // ds_read_b32
        else if (MODE == 4) { float o = lds[a]; lds[a] = o + v; }           // plain RMW (racy)
        else if (MODE == 5) v += __shfl(v, (lane * 7 + j) & 63);            // ds_bpermute
        else if (MODE == 6) {                                               // owner election + RMW
          bool pending = true;
          while (__ballot(pending)) {
            if (pending) ul[8192 + (a & 8191)] = lane;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (pending && ul[8192 + (a & 8191)] == (unsigned)lane) {
              float o = lds[a & 8191];
              lds[a & 8191] = o + v;
              pending = false;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
          }
        }
      }
    }
  }
  long long t1 = clock64();
  __syncthreads();
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
  out[blockIdx.x * blockDim.x + threadIdx.x] = v + lds[threadIdx.x];
}

int main() {
  const int NW = 8;
  std::vector<int> h(NW * 8 * 64);
  int* d_idx; float* d_out; long long* d_cyc;
  hipMalloc(&d_idx, h.size() * 4); hipMalloc(&d_out, 256 * 512 * 4); hipMalloc(&d_cyc, 256 * 8 * 8);
  const char* pat[] = {"distinct banks (stride 1)", "random", "all same address", "runs of 8 equal", "pitch17 rows (k*17+f)"};
  const char* mode[] = {"ds_add_f32", "ds_add_u32", "ds_write_b32", "ds_read_b32", "plain RMW", "ds_bpermute", "owner-elect RMW"};
  for (int p = 0; p < 5; ++p) {
    for (size_t i = 0; i < h.size(); ++i) {
      int lane = i & 63, j = (i >> 6);
      if (p == 0) h[i] = lane + 64 * j;
      else if (p == 1) h[i] = (rand() % 8192);
      else if (p == 2) h[i] = 5 + j;
      else if (p == 3) h[i] = (lane / 8) * 33 + 64 * j;
      else h[i] = ((rand() % 513) * 17 + (j & 15));
    }
    hipMemcpy(d_idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int waves = 1; waves <= 8; waves *= 8) {
      for (int active = 64; active >= 16; active -= 48) {
        printf("pattern: %-28s waves/CU=%d active=%2d :", pat[p], waves, active);
        for (int m = 0; m < 7; ++m) {
          long long c[64];
          void (*fn)(const int*, float*, long long*, int) =
              m == 0 ? k<0> : m == 1 ? k<1> : m == 2 ? k<2> : m == 3 ? k<3> : m == 4 ? k<4> : m == 5 ? k<5> : k<6>;
          hipLaunchKernelGGL(fn, dim3(1), dim3(64 * waves), 0, 0, d_idx, d_out, d_cyc, active);
          hipLaunchKernelGGL(fn, dim3(1), dim3(64 * waves), 0, 0, d_idx, d_out, d_cyc, active);
          hipDeviceSynchronize();
          hipMemcpy(c, d_cyc, waves * 8, hipMemcpyDeviceToHost);
          long long mx = 0;
          for (int w = 0; w < waves; ++w) mx = c[w] > mx ? c[w] : mx;
          printf(" %s=%.1f", mode[m], (double)mx / ITERS);
        }
        printf("\n");
      }
    }
  }
  return 0;
}
