// Micro-benchmark: cost of the fused kernel's sample-load pattern on gfx950.
//   mode 0: per frame 16 x global_load_dword (lane t reads x[pos + t + 64q]), frames hop 256 apart (75 % overlap)
//   mode 1: per frame 4 x global_load_dwordx4 (lane t reads 16 contiguous samples)
//   mode 2: per frame 4 x dword (only the 256 NEW samples of the hop)  -- what an LDS-staged design would fetch
// 8 waves per CU, 256 blocks, persistent over tiles like the real kernel.  Prints ns per frame per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// mode 3: the read-out store pattern alone: per tile 513 rows x 16 frames of float2, 8-byte stores, 4 rows per wave instr
// mode 4: mode 0 loads + mode 3 stores (the kernel's whole HBM-side traffic, no arithmetic)
// mode 5: as 4 but 16-byte stores (two frames per lane)
// mode 6: as 3 but tiles of 8 frames (64-byte row segments), 512 blocks = two per CU (the 8-wave variant's pattern)
template <int MODE>
__global__ __launch_bounds__(512) void kst(const float* __restrict__ x, float2* __restrict__ out, long long n_signal,
                                            int n_frames, long long total_tiles, int tiles_per_sig) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (long long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const long long sig = tile / tiles_per_sig;
    const int frame0 = (int)(tile % tiles_per_sig) * 16;
    const float* xs = x + sig * n_signal;
    if (MODE >= 4) {
      for (int it = 0; it < 2; ++it) {
        const int frame = frame0 + it * 8 + wave;
        long long pos = (long long)frame * 256;
        if (pos + 1024 > n_signal) pos = n_signal - 1024;
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = xs[pos + lane + 64 * q];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += v[q];
      }
    }
    float2* og = out + sig * 513LL * n_frames + frame0;
    if (MODE == 5) {
      for (int i = threadIdx.x; i < 513 * 8; i += 512) {
        const int k = i / 8, f = (i % 8) * 2;
        float4 val = {acc, (float)i, acc, (float)i};
        *reinterpret_cast<float4*>(&og[(long long)k * n_frames + f]) = val;
      }
    } else {
      for (int i = threadIdx.x; i < 513 * 16; i += 512) {
        const int k = i / 16, f = i % 16;
        og[(long long)k * n_frames + f] = float2{acc, (float)i};
      }
    }
  }
  if (acc == 123.456f) out[0].x = acc;
}

__global__ __launch_bounds__(512) void kst8(float2* __restrict__ out, int n_frames, long long total_tiles, int tiles_per_sig) {
  for (long long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const long long sig = tile / tiles_per_sig;
    const int frame0 = (int)(tile % tiles_per_sig) * 8;
    float2* og = out + sig * 513LL * n_frames + frame0;
    for (int i = threadIdx.x; i < 513 * 8; i += 512) {
      const int k = i / 8, f = i % 8;
      og[(long long)k * n_frames + f] = float2{(float)tile, (float)i};
    }
  }
}

template <int MODE>
__global__ __launch_bounds__(512) void k(const float* __restrict__ x, float* out, long long n_signal, int frames_per_sig,
                                          long long total_tiles, int tiles_per_sig) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.f;
  for (long long tile = blockIdx.x; tile < total_tiles; tile += gridDim.x) {
    const long long sig = tile / tiles_per_sig;
    const int frame0 = (int)(tile % tiles_per_sig) * 16;
    const float* xs = x + sig * n_signal;
    for (int it = 0; it < 2; ++it) {
      const int frame = frame0 + it * 8 + wave;
      long long pos = (long long)frame * 256;
      if (pos + 1024 > n_signal) pos = n_signal - 1024;
      if (MODE == 0) {
        float v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = xs[pos + lane + 64 * q];
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += v[q];
      } else if (MODE == 1) {
        float4 v[4];
        const float4* p4 = reinterpret_cast<const float4*>(xs + pos);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = p4[lane + 64 * q];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += v[q].x + v[q].y + v[q].z + v[q].w;
      } else {
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = xs[pos + 768 + lane + 64 * q];
#pragma unroll
        for (int q = 0; q < 4; ++q) acc += v[q];
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

int main() {
  const long long N = 1 << 20;
  const int B = 64;
  float* d_x; float* d_o;
  hipMalloc(&d_x, B * N * 4); hipMalloc(&d_o, 256 * 512 * 4);
  hipMemset(d_x, 0, B * N * 4);
  const int tps = 256; const long long tiles = (long long)B * tps;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(512), 0, 0, d_x, d_o, N, 4096, tiles, tps);
      if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(512), 0, 0, d_x, d_o, N, 4096, tiles, tps);
      if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(512), 0, 0, d_x, d_o, N, 4096, tiles, tps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double frames = (double)B * 4096;
    printf("mode %d: %.3f ms  -> %.1f ns per frame per CU, %.1f GB/s requested\n", mode, best,
           best * 1e6 / (frames / 256), frames * (mode == 2 ? 1024.0 : 4096.0) / (best * 1e-3) / 1e9);
  }
  float2* d_out; hipMalloc(&d_out, (size_t)B * 513 * 4096 * 8);
  for (int mode = 3; mode < 7; ++mode) {
    float best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEventRecord(e0);
      if (mode == 3) hipLaunchKernelGGL(kst<3>, dim3(256), dim3(512), 0, 0, d_x, d_out, N, 4096, tiles, tps);
      if (mode == 4) hipLaunchKernelGGL(kst<4>, dim3(256), dim3(512), 0, 0, d_x, d_out, N, 4096, tiles, tps);
      if (mode == 5) hipLaunchKernelGGL(kst<5>, dim3(256), dim3(512), 0, 0, d_x, d_out, N, 4096, tiles, tps);
      if (mode == 6) hipLaunchKernelGGL(kst8, dim3(512), dim3(512), 0, 0, d_out, 4096, 2 * tiles, 2 * tps);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    const double frames = (double)B * 4096;
    printf("mode %d: %.3f ms  -> %.1f ns per frame per CU, %.1f GB/s written\n", mode, best,
           best * 1e6 / (frames / 256), frames * 513 * 8.0 / (best * 1e-3) / 1e9);
  }
  return 0;
}
