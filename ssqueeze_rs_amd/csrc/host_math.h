// host_math.h -- fp64 host-side helpers of libssq_hip: the once-per-call quantities the
// reference also computes once per call (window sizing, spectral diff-window, scale and
// frequency vectors).  Pure host C++, no device code.
#pragma once
#include <cmath>
#include <complex>
#include <cstdint>
#include <vector>

namespace ssq {
namespace host {

using cd = std::complex<double>;

inline bool is_pow2(int64_t n) { return n > 0 && (n & (n - 1)) == 0; }

// 2.0_f64.powf(p) (cwt.rs:485, ssq_cwt.rs:72): call libm pow through a volatile pointer so the
// compiler cannot rewrite pow(2, p) as exp2(p) (the two differ in the last ulp)
inline double pow2f(double p) {
  static double (*volatile fn)(double, double) = static_cast<double (*)(double, double)>(std::pow);
  return fn(2.0, p);
}

// in-place radix-2 FFT, n a power of two; sign = -1 forward, +1 inverse (unnormalised)
inline void fft_pow2(std::vector<cd>& a, int sign) {
  const size_t n = a.size();
  for (size_t i = 1, j = 0; i < n; ++i) {
    size_t bit = n >> 1;
    for (; j & bit; bit >>= 1) j ^= bit;
    j ^= bit;
    if (i < j) std::swap(a[i], a[j]);
  }
  std::vector<cd> w(n / 2 > 0 ? n / 2 : 1);
  for (size_t i = 0; i < n / 2; ++i) {
    const long double ang = (long double)sign * 2.0L * 3.14159265358979323846264338327950288L * (long double)i / (long double)n;
    w[i] = cd((double)cosl(ang), (double)sinl(ang));
  }
  for (size_t len = 2; len <= n; len <<= 1) {
    const size_t step = n / len;
    for (size_t i = 0; i < n; i += len) {
      for (size_t k = 0; k < len / 2; ++k) {
        const cd u = a[i + k], v = a[i + k + len / 2] * w[k * step];
        a[i + k] = u + v;
        a[i + k + len / 2] = u - v;
      }
    }
  }
}

// any-length unnormalised DFT: power of two -> radix-2; small n -> direct sum with an exact-index
// long-double twiddle table; otherwise Bluestein on the radix-2 transform.
inline void fft_any(std::vector<cd>& a, int sign) {
  const int64_t n = (int64_t)a.size();
  if (n <= 1) return;
  if (is_pow2(n)) {
    fft_pow2(a, sign);
    return;
  }
  const long double PI = 3.14159265358979323846264338327950288L;
  if (n <= 8192) {
    std::vector<cd> tw(n);
    for (int64_t i = 0; i < n; ++i) {
      const long double ang = (long double)sign * 2.0L * PI * (long double)i / (long double)n;
      tw[i] = cd((double)cosl(ang), (double)sinl(ang));
    }
    std::vector<cd> out(n);
    for (int64_t k = 0; k < n; ++k) {
      long double sr = 0, si = 0;
      int64_t idx = 0;
      for (int64_t j = 0; j < n; ++j) {
        sr += (long double)a[j].real() * tw[idx].real() - (long double)a[j].imag() * tw[idx].imag();
        si += (long double)a[j].real() * tw[idx].imag() + (long double)a[j].imag() * tw[idx].real();
        idx += k;
        if (idx >= n) idx -= n;
      }
      out[k] = cd((double)sr, (double)si);
    }
    a.swap(out);
    return;
  }
  int64_t m = 1;
  while (m < 2 * n - 1) m <<= 1;
  std::vector<cd> chirp(n), A(m, cd(0, 0)), B(m, cd(0, 0));
  for (int64_t i = 0; i < n; ++i) {
    const int64_t i2 = (i * i) % (2 * n);
    const long double ang = (long double)sign * PI * (long double)i2 / (long double)n;
    chirp[i] = cd((double)cosl(ang), (double)sinl(ang));
  }
  for (int64_t i = 0; i < n; ++i) A[i] = a[i] * chirp[i];
  B[0] = std::conj(chirp[0]);
  for (int64_t i = 1; i < n; ++i) B[i] = B[m - i] = std::conj(chirp[i]);
  fft_pow2(A, -1);
  fft_pow2(B, -1);
  for (int64_t i = 0; i < m; ++i) A[i] *= B[i];
  fft_pow2(A, +1);
  for (int64_t i = 0; i < n; ++i) a[i] = A[i] * (1.0 / (double)m) * chirp[i];
}

// ssq_stft.rs:104-119
inline std::vector<double> size_window(const double* w, int64_t L, int64_t n_fft) {
  std::vector<double> out((size_t)n_fft, 0.0);
  if (L < n_fft) {
    const int64_t pl = (n_fft - L) / 2;
    for (int64_t i = 0; i < L; ++i) out[i + pl] = w[i];
  } else if (L > n_fft) {
    const int64_t s = (L - n_fft) / 2;
    for (int64_t i = 0; i < n_fft; ++i) out[i] = w[s + i];
  } else {
    for (int64_t i = 0; i < n_fft; ++i) out[i] = w[i];
  }
  return out;
}

// ssq_stft.rs:131-179; zero_nyquist: the upstream variant (old/ssqueezepy/_stft.py:293-299) drops the Nyquist term of
// even lengths
inline std::vector<double> diff_window(const double* win, int64_t n, bool zero_nyquist = false) {
  std::vector<double> freqs((size_t)n);
  for (int64_t i = 0; i < n / 2 + 1 && i < n; ++i) freqs[i] = (double)i;
  for (int64_t i = n / 2 + 1; i < n; ++i) freqs[i] = (double)i - (double)n;
  for (int64_t i = 0; i < n; ++i) freqs[i] *= 2.0 * M_PI / (double)n;
  if (zero_nyquist && n % 2 == 0 && n > 0) freqs[n / 2] = 0.0;
  std::vector<cd> W((size_t)n);
  for (int64_t i = 0; i < n; ++i) W[i] = cd(win[i], 0.0);
  fft_any(W, -1);
  for (int64_t i = 0; i < n; ++i) W[i] = cd(-W[i].imag() * freqs[i], W[i].real() * freqs[i]);
  fft_any(W, +1);
  std::vector<double> out((size_t)n);
  const double scale = 1.0 / (double)n;
  for (int64_t i = 0; i < n; ++i) out[i] = W[i].real() * scale;
  return out;
}

// numpy.linspace(a, b, n) as upstream uses it (old/ssqueezepy/_ssq_stft.py:248-257): arange(n)*step + a, last = b
inline std::vector<double> np_linspace(double a, double b, int64_t n) {
  std::vector<double> y((size_t)(n > 0 ? n : 0));
  if (n == 1) y[0] = a;
  if (n < 2) return y;
  const double step = (b - a) / (double)(n - 1);
  for (int64_t i = 0; i < n; ++i) y[i] = (double)i * step + a;
  y[n - 1] = b;
  return y;
}

// old/ssqueezepy/utils/common.py:32-51: padded length 2^(1 + round(log2 n)) (numpy rounds half to even), left pad the
// larger half
inline void p2up(int64_t n, int64_t* up, int64_t* n1, int64_t* n2) {
  const double r = std::nearbyint(std::log2((double)n));
  *up = (int64_t)std::llround(std::pow(2.0, 1.0 + r));
  *n2 = (*up - n) / 2;
  *n1 = *up - n - *n2;
}

// utils/array.rs:9-11
inline int64_t next_power_of_2(int64_t n) {
  if (n <= 0) return 1;
  const double l = std::ceil(std::log2((double)n));
  return (int64_t)1 << (int64_t)l;
}

// cwt.rs:461-489 / ssq_cwt.rs:300-326 ; cwt_simd.rs:474-545 when simd_variant
inline std::vector<double> log_scales(int64_t N, int64_t nv, bool simd_variant) {
  const double log_min = std::log2(2.0);
  const double log_max = std::log2((double)N * 0.5);
  const double num_octaves = log_max - log_min;
  const double c = std::ceil(num_octaves * (double)nv);
  const int64_t num = (std::isfinite(c) && c > 0) ? (int64_t)c : 0;
  const double sf = num > 1 ? (log_max - log_min) / (double)(num - 1) : 0.0;
  std::vector<double> s((size_t)num);
  for (int64_t i = 0; i < num; ++i) {
    const double p = log_min + (double)i * sf;
    s[i] = (simd_variant && num >= 16) ? std::exp(p * M_LN2) : pow2f(p);
  }
  return s;
}

// ssq_cwt.rs:50-113
inline std::vector<double> cwt_ssq_freqs(int64_t n, double fmin, double fmax, bool linear) {
  std::vector<double> f((size_t)n);
  if (linear) {
    const double step = n > 1 ? (fmax - fmin) / (double)(n - 1) : 0.0;
    for (int64_t i = 0; i < n; ++i) f[i] = fmin + (double)i * step;
  } else {
    const double lmin = std::log2(fmin), lmax = std::log2(fmax);
    const double sf = n > 1 ? (lmax - lmin) / (double)(n - 1) : 0.0;
    for (int64_t i = 0; i < n; ++i) f[i] = pow2f(lmin + (double)i * sf);
  }
  return f;
}

}  // namespace host
}  // namespace ssq
