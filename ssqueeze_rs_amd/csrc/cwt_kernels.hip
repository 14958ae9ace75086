// cwt_kernels.hip -- CWT-family kernels for gfx950 (MI355X).
//
// Frequency-domain CWT as the reference does it (rust/src/spectral/cwt.rs:85-326,
// ssq_cwt.rs:329-435): FFT of the padded signal once, then per scale a wavelet multiply and an
// inverse FFT of the full padded length P (a power of two, utils/array.rs:9-11).  On the GPU the
// length-P transforms are two-step ("four-step") FFTs, P = P1*P2 with P1,P2 <= 4096:
//   step A: C adjacent columns of the [P1][P2] view per block, length-P1 FFTs in LDS, times W_P^(c*k1)
//   step B: C adjacent rows per block, length-P2 FFTs in LDS, output transposed in LDS so that global
//           stores are C-element contiguous segments (natural order, unpadded: cwt.rs:108-129)
// The wavelet multiply (cwt.rs:238-240, :275-277) is fused into step A's load and the 1/P and
// sqrt(scale) normalisation (cwt.rs:251-262) into step B's store; xh (16.8 MB at P = 2^21) is
// re-read by every scale and stays resident in the 256 MB Infinity Cache.
// Inverse transforms run the forward FFT core on conjugated data (ifft(x) = conj(fft(conj(x)))).
#include "cwt_kernels.h"
#include "fft_core.h"
#include "stft_kernels.h"   // load_padded

namespace ssq {

constexpr int kTileThreads = 512;   // 8 waves per block (one block per CU: the tile takes most of the LDS)

constexpr int pow2_floor(int v) {
  int r = 1;
  while (2 * r <= v) r *= 2;
  return r;
}

template <typename T, int LOGM>
struct TileCfg {
  static constexpr int M = 1 << LOGM;
  static constexpr int L = M / 16;                                    // lanes per transform
  static constexpr int TPR = (L >= kTileThreads) ? 1 : kTileThreads / L;   // transforms per round
  static constexpr int ROWP = M + M / 16 + 1;                         // odd-ish pitch: bank spread
  static constexpr int ROW_BYTES = ROWP * (int)sizeof(cpx<T>);
  static constexpr int CCAP = (sizeof(T) == 4) ? 16 : 8;              // >= 128-B global segments
  static constexpr int CFIT = pow2_floor(160 * 1024 / ROW_BYTES);
  static constexpr int CWANT = (TPR > CCAP) ? TPR : CCAP;
  static constexpr int C = (CWANT < CFIT) ? CWANT : CFIT;             // transforms per tile
  static constexpr int LDS_BYTES = C * ROW_BYTES;
  static constexpr bool MULTIWAVE = (L > 64);
  static_assert(LOGM >= 4 && LOGM <= 12, "tile FFT length");
  static_assert(C >= TPR && C % TPR == 0, "whole rounds");
};

template <typename T>
__device__ __forceinline__ cpx<T> conj_if(cpx<T> v, bool inv) {
  if (inv) v.y = -v.y;
  return v;
}

// element n of the spectrum fed to an inverse transform: xh[n] * psih_s[n] (* i*xi_n/dt)
template <typename T>
__device__ __forceinline__ cpx<T> load_spectrum(const CwtDev<T>& p, int tr, long long n) {
  const long long half = p.P >> 1;
  if (n > half) return {(T)0, (T)0};                    // analytic wavelets: w < 0 -> 0 (cwt.rs:512,:536)
  const int s = p.scale0 + tr / p.n_kinds;
  const int kind = tr % p.n_kinds;
  const T psi = p.psih[(long long)s * (half + 1) + n];
  const cpx<T> xv = p.xh[n];
  cpx<T> v = {xv.x * psi, xv.y * psi};                  // cwt.rs:238-240
  if (kind == 1) {                                      // * Complex(0, xi/dt)  cwt.rs:205-208
    const T xi = (T)n * p.xi_step;
    v = {-v.y * xi, v.x * xi};
  }
  return v;
}

template <typename T>
__device__ __forceinline__ void store_time(const CwtDev<T>& p, int tr, long long n, cpx<T> v) {
  const int s = p.scale0 + tr / p.n_kinds;
  const int kind = tr % p.n_kinds;
  const T sc = p.out_scale[s];
  v = {v.x * sc, v.y * sc};
  cpx<T>* dst = kind ? p.dWx : p.Wx;
  if (p.rpadded) {
    dst[(long long)s * p.P + n] = v;
  } else if (n >= p.n1 && n < p.n1 + p.n_signal) {      // cwt.rs:115
    dst[(long long)s * p.n_signal + (n - p.n1)] = v;
  }
}

template <typename T, int LOGM>
__global__ __launch_bounds__(kTileThreads) void cwt_tile_kernel(CwtDev<T> p, int mode) {
  using K = TileCfg<T, LOGM>;
  constexpr int M = K::M, L = K::L, C = K::C, ROWP = K::ROWP;
  constexpr bool TW_REGS = (sizeof(T) == 4);
  __shared__ __attribute__((aligned(16))) unsigned char smem[K::LDS_BYTES];
  cpx<T>* rows = reinterpret_cast<cpx<T>*>(smem);

  const int tid = threadIdx.x;
  const int tr = blockIdx.y;
  const long long tile = blockIdx.x;
  const bool inv = mode >= CWT_INV_A;
  const long long P2 = 1LL << p.log_p2;
  const long long P1 = 1LL << p.log_p1;
  const bool stepA = (mode == CWT_FWD_A || mode == CWT_INV_A);
  const bool stepB = (mode == CWT_FWD_B || mode == CWT_INV_B);

  // ---------------- load (conjugated for inverse transforms) ----------------
  if (stepA) {
    const long long c0 = tile * C;
    for (int e = tid; e < C * M; e += kTileThreads) {
      const int c = e % C, r = e / C;
      const long long n = (long long)r * P2 + c0 + c;
      cpx<T> v;
      if (mode == CWT_FWD_A) {
        v = {load_padded(p.x, n - p.n1, p.n_signal, p.padtype), (T)0};
      } else {
        v = conj_if(load_spectrum(p, tr, n), true);
      }
      rows[c * ROWP + exch_phys(r)] = v;
    }
  } else if (stepB) {
    const long long r0 = tile * C;
    const cpx<T>* __restrict__ src = p.ybuf + (long long)tr * p.P + r0 * P2;
    for (int e = tid; e < C * M; e += kTileThreads) {
      const int m = e % M, c = e / M;
      rows[c * ROWP + exch_phys(m)] = src[(long long)c * P2 + m];   // step A already left it conjugated
    }
  } else {
    for (int e = tid; e < C * M; e += kTileThreads) {
      const int m = e % M, c = e / M;
      const long long trc = tile * C + c;
      cpx<T> v = {(T)0, (T)0};
      if (trc < p.n_transforms) {
        if (mode == CWT_FWD_S) v = {load_padded(p.x, (long long)m - p.n1, p.n_signal, p.padtype), (T)0};
        else v = conj_if(load_spectrum(p, (int)trc, m), true);
      }
      rows[c * ROWP + exch_phys(m)] = v;
    }
  }
  __syncthreads();

  // ---------------- length-M forward FFTs in LDS ----------------
  {
    const int slot = tid / L;
    const int t = tid % L;
    cpx<T> twr[3][16];
    if constexpr (TW_REGS) {
#pragma unroll
      for (int P = 1; P < num_passes(LOGM); ++P) {
        const int R = pass_radix(LOGM, P), NS = pass_ns(LOGM, P), NB = 16 / R;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
#pragma unroll
          for (int m = 1; m < 16; ++m) {
            if (b < NB && m < R) {
              const int k = (t + L * b) & (NS - 1);
              twr[P - 1][b + m * NB] = p.tw_m[k * m * (M / (NS * R))];
            }
          }
        }
      }
    }
#pragma unroll 1
    for (int round = 0; round < C / K::TPR; ++round) {
      cpx<T>* row = rows + (round * K::TPR + slot) * ROWP;
      cpx<T> v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = row[exch_phys(t + L * q)];
      frame_sync<K::MULTIWAVE>();
      fft_pass<T, LOGM, 0, false, TW_REGS, K::MULTIWAVE>(v, row, twr, p.tw_m, t);
#pragma unroll
      for (int q = 0; q < 16; ++q) row[exch_phys(t + L * q)] = v[q];
    }
  }
  __syncthreads();

  // ---------------- store ----------------
  if (stepA) {
    const long long c0 = tile * C;
    cpx<T>* __restrict__ dst = p.ybuf + (long long)tr * p.P;
    for (int e = tid; e < C * M; e += kTileThreads) {
      const int c = e % C, k1 = e / C;
      const long long col = c0 + c;
      const long long r = col * k1;                          // < P1*P2 = P: no reduction needed
      const cpx<T> w = cmul(p.tw_hi[r >> 12], p.tw_lo[r & 4095]);   // W_P^r (forward sign)
      // data is conj(true value) for inverse transforms; conj(y * conj(W)) = conj(y) * W
      dst[(long long)k1 * P2 + col] = cmul(rows[c * ROWP + exch_phys(k1)], w);
    }
  } else if (stepB) {
    const long long r0 = tile * C;
    for (int e = tid; e < C * M; e += kTileThreads) {
      const int c = e % C, k2 = e / C;
      const long long n = r0 + c + P1 * k2;
      const cpx<T> v = conj_if(rows[c * ROWP + exch_phys(k2)], inv);
      if (mode == CWT_FWD_B) p.xh[n] = v;
      else store_time(p, tr, n, v);
    }
  } else {
    for (int e = tid; e < C * M; e += kTileThreads) {
      const int m = e % M, c = e / M;
      const long long trc = tile * C + c;
      if (trc >= p.n_transforms) continue;
      const cpx<T> v = conj_if(rows[c * ROWP + exch_phys(m)], inv);
      if (mode == CWT_FWD_S) p.xh[m] = v;
      else store_time(p, (int)trc, m, v);
    }
  }
}

template <typename T, int LOGM>
static hipError_t launch_tile_one(int mode, const CwtDev<T>& p, hipStream_t stream) {
  using K = TileCfg<T, LOGM>;
  dim3 grid;
  if (mode == CWT_FWD_A || mode == CWT_INV_A) {
    grid = dim3((unsigned)(((1LL << p.log_p2) + K::C - 1) / K::C), (unsigned)p.n_transforms, 1);
  } else if (mode == CWT_FWD_B || mode == CWT_INV_B) {
    grid = dim3((unsigned)(((1LL << p.log_p1) + K::C - 1) / K::C), (unsigned)p.n_transforms, 1);
  } else {
    grid = dim3((unsigned)((p.n_transforms + K::C - 1) / K::C), 1, 1);
  }
  hipLaunchKernelGGL((cwt_tile_kernel<T, LOGM>), grid, dim3(kTileThreads), 0, stream, p, mode);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_cwt_tile(int mode, const CwtDev<T>& p, hipStream_t stream) {
  int logm;
  if (mode == CWT_FWD_A || mode == CWT_INV_A) logm = p.log_p1;
  else if (mode == CWT_FWD_B || mode == CWT_INV_B) logm = p.log_p2;
  else logm = p.log_p1;
  switch (logm) {
    case 4: return launch_tile_one<T, 4>(mode, p, stream);
    case 5: return launch_tile_one<T, 5>(mode, p, stream);
    case 6: return launch_tile_one<T, 6>(mode, p, stream);
    case 7: return launch_tile_one<T, 7>(mode, p, stream);
    case 8: return launch_tile_one<T, 8>(mode, p, stream);
    case 9: return launch_tile_one<T, 9>(mode, p, stream);
    case 10: return launch_tile_one<T, 10>(mode, p, stream);
    case 11: return launch_tile_one<T, 11>(mode, p, stream);
    case 12: return launch_tile_one<T, 12>(mode, p, stream);
  }
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------- wavelet table ----
// cwt.rs:492-547; evaluated in fp64 for both dtypes, rounded once to T.
template <typename T>
__global__ void wavelet_table_kernel(T* __restrict__ psih, const double* __restrict__ scales, int na,
                                     long long P, int wavelet) {
  const long long half = P >> 1;
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (k > half || s >= na) return;
  const double h = 1.0 * (2.0 * 3.14159265358979323846) / (double)P;   // base.rs:20
  const double xi = (double)k * h;
  const double w = scales[s] * xi;
  double v = 0.0;
  if (wavelet == 1) {                                   // "morlet"  cwt.rs:497-520
    if (w >= 0.0) {
      const double mu = 6.0;
      const double norm = pow(3.14159265358979323846, -0.25) * 1.41421356237309504880;
      const double k_exp = exp(-0.5 * mu * mu);
      const double wm = w - mu;
      v = norm * (exp(-0.5 * (wm * wm)) - k_exp * exp(-0.5 * (w * w)));
    }
  } else {                                              // "gmw" | _  cwt.rs:522-542
    if (w > 0.0) v = 2.0 * exp(60.0 * log(w) - pow(w, 3.0));
  }
  psih[(long long)s * (half + 1) + k] = (T)v;
}

template <typename T>
hipError_t launch_wavelet_table(T* psih, const double* d_scales, int na, long long P, int wavelet,
                                hipStream_t stream) {
  const long long half = P >> 1;
  dim3 grid((unsigned)((half + 1 + 255) / 256), (unsigned)na, 1);
  hipLaunchKernelGGL(wavelet_table_kernel<T>, grid, dim3(256), 0, stream, psih, d_scales, na, P, wavelet);
  return hipGetLastError();
}

// ------------------------------------------------------ tiny-P direct sums ----
template <typename T>
__global__ void cwt_naive_fwd_kernel(CwtDev<T> p) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= p.P) return;
  double sr = 0, si = 0;
  for (long long n = 0; n < p.P; ++n) {
    const double xv = (double)load_padded(p.x, n - p.n1, p.n_signal, p.padtype);
    double s, c;
    sincospi(-2.0 * (double)((n * k) % p.P) / (double)p.P, &s, &c);
    sr += xv * c;
    si += xv * s;
  }
  p.xh[k] = {(T)sr, (T)si};
}

template <typename T>
__global__ void cwt_naive_inv_kernel(CwtDev<T> p) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int tr = blockIdx.y;
  if (n >= p.P) return;
  double sr = 0, si = 0;
  for (long long k = 0; k < p.P; ++k) {
    const cpx<T> v = load_spectrum(p, tr, k);
    double s, c;
    sincospi(2.0 * (double)((n * k) % p.P) / (double)p.P, &s, &c);
    sr += (double)v.x * c - (double)v.y * s;
    si += (double)v.x * s + (double)v.y * c;
  }
  store_time(p, tr, n, cpx<T>{(T)sr, (T)si});
}

template <typename T>
hipError_t launch_cwt_naive_fwd(const CwtDev<T>& p, hipStream_t stream) {
  hipLaunchKernelGGL(cwt_naive_fwd_kernel<T>, dim3((unsigned)((p.P + 63) / 64)), dim3(64), 0, stream, p);
  return hipGetLastError();
}
template <typename T>
hipError_t launch_cwt_naive_inv(const CwtDev<T>& p, int n_transforms, hipStream_t stream) {
  hipLaunchKernelGGL(cwt_naive_inv_kernel<T>, dim3((unsigned)((p.P + 63) / 64), (unsigned)n_transforms),
                     dim3(64), 0, stream, p);
  return hipGetLastError();
}

// ------------------------------------------------ phase transform + reassignment ----
// ssq_cwt.rs:15-47 (phase_cwt) and :116-222 (ssqueeze).  One thread owns one time column and
// walks the scales in ascending order -- the reference's accumulation order, no atomics.
template <typename T>
__global__ void cwt_reassign_kernel(CwtSsqDev<T> p) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p.N) return;
  const T two_pi = (T)(2.0 * 3.14159265358979323846);
  for (int i = 0; i < p.na; ++i) {
    const long long o = (long long)i * p.N + j;
    const cpx<T> Wv = p.Wx[o], dW = p.dWx[o];
    T w;
    bool small;
    if constexpr (sizeof(T) == 4) {
      // pre-scaled ratio: the reference's GMW is un-normalised (peak ~4e17), so |Wx|^2 and
      // b*c - a*d overflow fp32 long before the ratio does
      const T mx = fmaxf(fabsf(Wv.x), fabsf(Wv.y));
      const T sc = (T)1 / mx;
      const T cs = Wv.x * sc, ds = Wv.y * sc;
      const T den = Wv.x * cs + Wv.y * ds;               // |Wx|^2 / mx
      small = !(mx * sqrtf(cs * cs + ds * ds) >= p.gamma);
      w = fabsf((dW.y * cs - dW.x * ds) / (den * two_pi));
    } else {
      const T den = Wv.x * Wv.x + Wv.y * Wv.y;
      small = hypot(Wv.x, Wv.y) < p.gamma;               // Complex::norm()  ssq_cwt.rs:29
      w = fabs((dW.y * Wv.x - dW.x * Wv.y) / (den * two_pi));
    }
    if (small) w = (T)INFINITY;
    int kk = -1;
    if (!(isinf(w) || w != w)) {                         // ssq_cwt.rs:167
      T v;
      if (p.is_log) v = (log2(w) - p.bin_min) / p.bin_step;    // ssq_cwt.rs:175-176
      else v = (w - p.bin_min) / p.bin_step;                   // ssq_cwt.rs:187
      const T r = round(v);                              // half away from zero
      int bin;
      if (r != r) bin = 0;                               // NaN as isize == 0
      else if (r < (T)0 || r >= (T)p.na) bin = -1;       // out of range: dropped (:177,:188)
      else bin = (int)r;
      if (bin >= 0) kk = p.flipud ? (p.na - 1 - bin) : bin;
    }
    if (p.wk) p.wk[o] = {w, (T)kk};
    if (kk >= 0) {
      const long long d = (long long)kk * p.N + j;
      cpx<T> acc = p.Tx[d];
      if (p.squeezing == 1) {
        acc.x += p.leb_val;
      } else {
        acc.x += Wv.x;
        acc.y += Wv.y;
      }
      p.Tx[d] = acc;
    }
  }
}

template <typename T>
hipError_t launch_cwt_reassign(const CwtSsqDev<T>& p, hipStream_t stream) {
  hipLaunchKernelGGL(cwt_reassign_kernel<T>, dim3((unsigned)((p.N + 63) / 64)), dim3(64), 0, stream, p);
  return hipGetLastError();
}

#define SSQ_INST(T)                                                                                   \
  template hipError_t launch_cwt_tile<T>(int, const CwtDev<T>&, hipStream_t);                         \
  template hipError_t launch_wavelet_table<T>(T*, const double*, int, long long, int, hipStream_t);   \
  template hipError_t launch_cwt_naive_fwd<T>(const CwtDev<T>&, hipStream_t);                         \
  template hipError_t launch_cwt_naive_inv<T>(const CwtDev<T>&, int, hipStream_t);                    \
  template hipError_t launch_cwt_reassign<T>(const CwtSsqDev<T>&, hipStream_t);
SSQ_INST(float)
SSQ_INST(double)
#undef SSQ_INST

}  // namespace ssq
