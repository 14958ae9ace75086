// fft_wave1024.h -- a 1024-point complex fp32 FFT held by ONE wavefront (gfx950): lane t owns elements t + 64 q,
// q < 16, on entry and X[t + 64 q] (natural order) on exit.  The register core the 16-wave STFT kernel introduced
// (stft_fused.hip::stft_tx1024_kernel), factored out for the CWT row / column kernels (cwt_reg.hip):
//   pass 0  radix 16 in registers
//   exch 1  through an LDS row of 64 x 9 elements by register halves (16 ds_write_b64 + 16 ds_read_b64 per lane)
//   pass 1  twiddle W_256^(k m) from the compact [m][k] table, radix 16
//   exch 2  4 x 4 transpose between the four 16-lane rows and two register-index bits: v_permlane32_swap +
//           v_permlane16_swap, no LDS
//   pass 2  twiddle W_1024^((t + 64 b) m) from the compact [m][j] table, four radix-4 butterflies
// Forward sign (e^{-2 pi i nk/1024}); an inverse transform runs it on conjugated data.
#pragma once
#include "fft_core.h"

namespace ssq {

constexpr int kWave1024ExchElems = 64 * 9;     // LDS elements (8 B each) of one wave's exchange row
constexpr int kWave1024TwElems = 256 + 768;    // tw1 [16][16] + tw2 [3][256]

// fill the compact twiddle tables from the W_1024 table (tw[j] = e^{-2 pi i j/1024}); every thread of the block calls
// it, the caller synchronises
__device__ __forceinline__ void wave1024_tables(cpx<float>* tw1, cpx<float>* tw2, const cpx<float>* __restrict__ tw,
                                                int tid, int threads) {
  for (int i = tid; i < 256; i += threads) tw1[i] = tw[((i & 15) * (i >> 4) * 4) & 1023];
  for (int i = tid; i < 768; i += threads) tw2[i] = tw[((i & 255) * ((i >> 8) + 1)) & 1023];
}

__device__ __forceinline__ void wave1024_rows_transpose4(float& r0, float& r1, float& r2, float& r3) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(r0), __float_as_uint(r2), false, false);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1), __float_as_uint(r3), false, false);
  auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
  auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
  r0 = __uint_as_float(c[0]);
  r1 = __uint_as_float(c[1]);
  r2 = __uint_as_float(d[0]);
  r3 = __uint_as_float(d[1]);
}

// first half: pass 0 + exchange 1 (the caller may issue global prefetches between the halves: the exchange holds two
// register sets of the transform, the rest only one)
__device__ __forceinline__ void wave1024_front(cpx<float> (&v)[16], cpx<float>* exch, int t) {
  const cpx<float> unused[3][16] = {};
  fft_compute<float, 10, 0, false, false>(v, unused, nullptr, t);
  cpx<float> nv[16];
  const int rbase = 9 * (t >> 4) + (t & 7);
#pragma unroll
  for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
    for (int u = 0; u < 8; ++u) exch[9 * t + u] = v[8 * ph + u];
    frame_sync<false>();
    if (((t >> 3) & 1) == ph) {
#pragma unroll
      for (int q = 0; q < 16; ++q) nv[q] = exch[rbase + 36 * q];
    }
    frame_sync<false>();
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) v[q] = nv[q];
}

__device__ __forceinline__ void wave1024_back(cpx<float> (&v)[16], const cpx<float>* tw1, const cpx<float>* tw2, int t) {
  const cpx<float> unused[3][16] = {};
  fft_compute<float, 10, 1, false, false, true>(v, unused, tw1, t);
#pragma unroll
  for (int uh = 0; uh < 4; ++uh) {
    wave1024_rows_transpose4(v[4 * uh + 0].x, v[4 * uh + 1].x, v[4 * uh + 2].x, v[4 * uh + 3].x);
    wave1024_rows_transpose4(v[4 * uh + 0].y, v[4 * uh + 1].y, v[4 * uh + 2].y, v[4 * uh + 3].y);
  }
#define SSQ_SWAP_(i, j)          \
  {                              \
    const cpx<float> t_ = v[i];  \
    v[i] = v[j];                 \
    v[j] = t_;                   \
  }
  SSQ_SWAP_(1, 4) SSQ_SWAP_(2, 8) SSQ_SWAP_(3, 12) SSQ_SWAP_(6, 9) SSQ_SWAP_(7, 13) SSQ_SWAP_(11, 14)
#undef SSQ_SWAP_
  fft_compute<float, 10, 2, false, false, true>(v, unused, tw2 - 256, t);
}

}  // namespace ssq
