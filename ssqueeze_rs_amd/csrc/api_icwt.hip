// api_icwt.hip -- inverse CWT and the wavelet helper functions of the reference:
//   icwt                      rust/src/spectral/cwt.rs:550-718      (GPU: row-sum reduction / FFT filter bank)
//   morlet, morlet_freq/time  rust/src/wavelets/morlet.rs:59-145    (host, fp64: a few thousand points)
//   gmw, gmw_freq/time,
//   gmw_center_frequency      rust/src/wavelets/gmw.rs:236-357
// All are `#[pyfunction]`s the reference implements and advertises (src/ssqueeze/_rs.pyi:61-132) but does not
// register in lib.rs:25-32.
#include <cmath>
#include <complex>
#include <string>
#include <vector>

#include "../../include/ssq_hip.h"
#include "fft_generic.h"
#include "host_math.h"

using namespace ssq;

namespace {

// ---------------------------------------------------------------------------------------- icwt kernels ----
// one-integral (cwt.rs:588-627): x[j] = (sum_i Re Wx[i][j] * nf[i]) * final_norm + x_mean, scales ascending per
// column (the reference's order, so fp64 is bit-comparable); one thread per time sample, coalesced along j.
template <typename T>
__global__ void icwt_one_int_kernel(const cpx<T>* __restrict__ Wx, long long n_times, int na, const double* __restrict__ nf,
                                    double final_norm, double x_mean, long long x_len, double* __restrict__ x) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= x_len) return;
  double acc = 0.0;
  for (int i = 0; i < na; ++i) acc += (double)Wx[(long long)i * n_times + j].x * nf[i];
  x[j] = acc * final_norm + x_mean;
}

// two-integral: rows -> complex work rows of length x_len
template <typename T>
__global__ void icwt_gather_kernel(const cpx<T>* __restrict__ Wx, long long n_times, long long x_len, int na,
                                   cpx<double>* __restrict__ rows) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= x_len * na) return;
  const long long i = idx / x_len, j = idx - i * x_len;
  const cpx<T> v = Wx[i * n_times + j];
  rows[idx] = {(double)v.x, (double)v.y};
}

// wavelet of the hot path (cwt.rs:492-547) at xi_k = xifn(1, n)[k] (base.rs:18-33)
__device__ __forceinline__ double psih_at(long long k, long long n, double scale, int wavelet) {
  const double h = 1.0 * (2.0 * 3.14159265358979323846) / (double)n;
  const double xi = (k <= n / 2) ? (double)k * h : (double)(k - n) * h;
  const double w = scale * xi;
  if (wavelet == SSQ_WAVELET_MORLET) {
    if (!(w >= 0.0)) return 0.0;
    const double mu = 6.0;
    const double norm = pow(3.14159265358979323846, -0.25) * 1.41421356237309504880;
    const double wm = w - mu;
    return norm * (exp(-0.5 * (wm * wm)) - exp(-0.5 * mu * mu) * exp(-0.5 * (w * w)));
  }
  if (!(w > 0.0)) return 0.0;
  return 2.0 * exp(60.0 * log(w) - pow(w, 3.0));
}

// acc[k] = sum_i FFT(Wx_i)[k] * conj(psih_i[k]) * sn_i   (cwt.rs:662-687; the sum over scales commutes with the
// linear inverse transform, so ONE inverse FFT serves all scales)
__global__ void icwt_filter_sum_kernel(const cpx<double>* __restrict__ rows, long long n, int na,
                                       const double* __restrict__ scales, const double* __restrict__ sn, int wavelet,
                                       cpx<double>* __restrict__ acc) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  double ar = 0.0, ai = 0.0;
  for (int i = 0; i < na; ++i) {
    const double g = psih_at(k, n, scales[i], wavelet) * sn[i];
    const cpx<double> v = rows[(long long)i * n + k];
    ar += v.x * g;
    ai += v.y * g;
  }
  acc[k] = {ar, ai};
}

__global__ void icwt_finish_kernel(const cpx<double>* __restrict__ acc, long long n, double inv_n, double final_norm,
                                   double x_mean, double* __restrict__ x) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  x[j] = (acc[j].x * inv_n) * final_norm + x_mean;
}

struct DevBuf {
  void* p = nullptr;
  ~DevBuf() { hipFree(p); }
  int alloc(long long bytes) {
    SSQ_HIP(hipMalloc(&p, (size_t)(bytes > 0 ? bytes : 16)));
    return 0;
  }
};

// ------------------------------------------------------------------------------ wavelet helpers (host) ----
using cd = std::complex<double>;

std::vector<double> xifn_host(double scale, int64_t n) {          // wavelets/base.rs:18-33
  std::vector<double> xi((size_t)n, 0.0);
  const double h = scale * (2.0 * M_PI) / (double)n;
  for (int64_t i = 0; i < n / 2 + 1 && i < n; ++i) xi[i] = (double)i * h;
  for (int64_t i = n / 2 + 1; i < n; ++i) xi[i] = (double)(i - n) * h;
  return xi;
}

void morlet_psih(const double* w, int64_t n, double mu, double* out) {      // wavelets/morlet.rs:22-41
  const double cs = std::pow(1.0 + std::exp(-(mu * mu)) - 2.0 * std::exp(-3.0 / 4.0 * (mu * mu)), -0.5);
  const double ks = std::exp(-0.5 * (mu * mu));
  const double factor = std::sqrt(2.0) * cs * std::pow(M_PI, 0.25);
  for (int64_t i = 0; i < n; ++i) {
    const double t1 = std::exp(-0.5 * ((w[i] - mu) * (w[i] - mu)));
    const double t2 = ks * std::exp(-0.5 * (w[i] * w[i]));
    out[2 * i] = factor * (t1 - t2);
    out[2 * i + 1] = 0.0;
  }
}

double gamma_fn(double x) {                                                    // wavelets/gmw.rs:172-201
  if (x < 0.5) return M_PI / (std::sin(M_PI * x) * gamma_fn(1.0 - x));
  static const double p[8] = {676.5203681218851,   -1259.1392167224028,  771.32342877765313,   -176.61502916214059,
                              12.507343278686905,  -0.13857109526572012, 9.9843695780195716e-6, 1.5056327351493116e-7};
  x -= 1.0;
  double y = 0.99999999999980993;
  for (int i = 0; i < 8; ++i) y += p[i] / (x + (double)i + 1.0);
  const double t = x + 8.0 - 0.5;
  return std::sqrt(2.0 * M_PI) * std::pow(t, x + 0.5) * std::exp(-t) * y;
}
double factorial_fn(int n) {                                                   // gmw.rs:204-209
  double r = 1.0;
  for (int i = 1; i <= n; ++i) r *= (double)i;
  return r;
}
double binomial_fn(int n, int k) {                                             // gmw.rs:212-232
  if (k < 0 || k > n) return 0.0;
  if (k == 0 || k == n) return 1.0;
  if (n <= 20) return factorial_fn(n) / (factorial_fn(k) * factorial_fn(n - k));
  double c = 0.0;
  for (int i = 1; i <= k; ++i) c += std::log((double)(n - k + i)) - std::log((double)i);
  return std::exp(c);
}

void gmw_psih(const double* w, int64_t n, double gamma, double beta, bool bandpass, int order, double* out) {   // gmw.rs:72-159
  const double wc = std::pow(beta / gamma, 1.0 / gamma);
  const double r = (2.0 * beta + 1.0) / gamma;
  double nc = 0.0, coeff = 0.0;
  const int k = order;
  const double c = r - 1.0;
  const int ci = (int)c;
  if (order == 0) {
    nc = bandpass ? 2.0 / std::exp(beta * std::log(wc) - std::pow(wc, gamma))
                  : std::sqrt(2.0 * M_PI * gamma * std::pow(2.0, r) / gamma_fn(r));
  } else {
    coeff = bandpass ? 2.0 * std::sqrt(gamma_fn(r) * gamma_fn((double)k + 1.0) / gamma_fn((double)k + r))
                     : std::sqrt(2.0 * M_PI * gamma * std::pow(2.0, r) * gamma_fn((double)k + 1.0) / gamma_fn((double)k + r));
  }
  for (int64_t i = 0; i < n; ++i) {
    double v = 0.0;
    const double wi = w[i];
    if (wi > 0.0) {
      if (order == 0) {
        v = bandpass ? nc * std::exp(beta * std::log(wi) - std::pow(wi, gamma))
                     : nc * std::pow(wi, beta) * std::exp(-std::pow(wi, gamma));
      } else {
        const double xx = 2.0 * std::pow(wi, gamma);
        double lag = 0.0;
        for (int m = 0; m <= k; ++m) {
          const double b = binomial_fn(k + ci + 1, ci + m + 1) * binomial_fn(k, m);
          lag += b * ((m % 2 ? -1.0 : 1.0) * std::pow(xx, (double)m) / factorial_fn(m));
        }
        v = bandpass ? coeff * lag * std::exp(-beta * std::log(wc) + std::pow(wc, gamma) + beta * std::log(wi) - std::pow(wi, gamma))
                     : coeff * lag * std::pow(wi, beta) * std::exp(-std::pow(wi, gamma));
      }
    }
    out[2 * i] = v;
    out[2 * i + 1] = 0.0;
  }
}

// morlet.rs:114-141 == gmw.rs:306-333
void time_from_freq(double* psih, int64_t n) {
  std::vector<cd> a((size_t)n);
  for (int64_t i = 0; i < n; ++i) a[i] = cd(psih[2 * i], psih[2 * i + 1]) * ((i % 2) ? -1.0 : 1.0);
  if (n % 2 == 0 && n > 0) a[n / 2] /= 2.0;
  host::fft_any(a, +1);
  const double s = 1.0 / (double)n;
  for (int64_t i = 0; i < n; ++i) {
    psih[2 * i] = a[i].real() * s;
    psih[2 * i + 1] = a[i].imag() * s;
  }
}

bool is_bandpass(const char* norm) {
  std::string s = norm ? norm : "bandpass";
  for (auto& ch : s) ch = (char)std::tolower((unsigned char)ch);
  return s == "bandpass";
}

}  // namespace

extern "C" {

int ssq_icwt_host(int dtype, const void* Wx, int64_t na, int64_t n_times, int wavelet, const double* scales,
                  int64_t n_scales_given, int one_int, int64_t x_len, double x_mean, int l1_norm, double* x_out) {
  if (!Wx || !x_out) SSQ_FAIL("NULL pointer");
  if (dtype != SSQ_F32 && dtype != SSQ_F64) SSQ_FAIL("dtype must be SSQ_F32 or SSQ_F64");
  if (!scales) SSQ_FAIL("Scales must be provided");                                    // cwt.rs:571-575
  if (x_len < 0) x_len = n_times;
  if (x_len > n_times || n_scales_given < na) SSQ_FAIL("index out of bounds: Wx[[i, j]] / scales[i] (cwt.rs:602,:614)");
  if (na > 0x7fffffff) SSQ_FAIL("too many scales");
  if (x_len == 0) return 0;
  const double adm = (wavelet == SSQ_WAVELET_MORLET) ? 0.776 : 1.0;                   // cwt.rs:578-582
  const double dj = (na > 1 && scales[1] > scales[0]) ? std::log(scales[1] / scales[0]) : 0.1;   // :593-597
  const double final_norm = (2.0 / adm) * dj;
  const long long esz = dtype == SSQ_F32 ? 8 : 16;
  std::vector<double> fac((size_t)(na > 0 ? na : 1));
  for (int64_t i = 0; i < na; ++i) {
    if (one_int) fac[i] = l1_norm ? 1.0 : 1.0 / std::sqrt(scales[i]);                 // :605-609
    else fac[i] = l1_norm ? 1.0 / scales[i] : 1.0 / (std::sqrt(scales[i]) * std::sqrt(scales[i]));   // :679-683
  }
  DevBuf dW, dfac, dx, dsc;
  if (int rc = dW.alloc(na * n_times * esz)) return rc;
  if (int rc = dfac.alloc(na * 8)) return rc;
  if (int rc = dsc.alloc(na * 8)) return rc;
  if (int rc = dx.alloc(x_len * 8)) return rc;
  if (na > 0) {
    SSQ_HIP(hipMemcpy(dW.p, Wx, (size_t)(na * n_times * esz), hipMemcpyHostToDevice));
    SSQ_HIP(hipMemcpy(dfac.p, fac.data(), (size_t)(na * 8), hipMemcpyHostToDevice));
    SSQ_HIP(hipMemcpy(dsc.p, scales, (size_t)(na * 8), hipMemcpyHostToDevice));
  }
  const dim3 blk(256), grd((unsigned)((x_len + 255) / 256));
  if (one_int) {
    if (dtype == SSQ_F32)
      hipLaunchKernelGGL(icwt_one_int_kernel<float>, grd, blk, 0, nullptr, (const cpx<float>*)dW.p, (long long)n_times, (int)na,
                         (const double*)dfac.p, final_norm, x_mean, (long long)x_len, (double*)dx.p);
    else
      hipLaunchKernelGGL(icwt_one_int_kernel<double>, grd, blk, 0, nullptr, (const cpx<double>*)dW.p, (long long)n_times,
                         (int)na, (const double*)dfac.p, final_norm, x_mean, (long long)x_len, (double*)dx.p);
    SSQ_HIP(hipGetLastError());
  } else {
    // FFT of every row (any length: rustfft plans any x_len), filter bank sum, one inverse FFT
    DevBuf rows, work, acc;
    const long long nrow = na > 0 ? na : 1;
    long long we = fft_work_elems(x_len, nrow);
    const long long we1 = fft_work_elems(x_len, 1);
    if (we1 > we) we = we1;
    if (int rc = rows.alloc(nrow * x_len * 16)) return rc;
    if (int rc = work.alloc(we * 16)) return rc;
    if (int rc = acc.alloc(x_len * 16)) return rc;
    if (na > 0) {
      const dim3 g2((unsigned)((x_len * na + 255) / 256));
      if (dtype == SSQ_F32)
        hipLaunchKernelGGL(icwt_gather_kernel<float>, g2, blk, 0, nullptr, (const cpx<float>*)dW.p, (long long)n_times,
                           (long long)x_len, (int)na, (cpx<double>*)rows.p);
      else
        hipLaunchKernelGGL(icwt_gather_kernel<double>, g2, blk, 0, nullptr, (const cpx<double>*)dW.p, (long long)n_times,
                           (long long)x_len, (int)na, (cpx<double>*)rows.p);
      SSQ_HIP(fft_any_batched<double>((cpx<double>*)rows.p, (cpx<double>*)work.p, x_len, na, -1, nullptr));
    }
    hipLaunchKernelGGL(icwt_filter_sum_kernel, grd, blk, 0, nullptr, (const cpx<double>*)rows.p, (long long)x_len, (int)na,
                       (const double*)dsc.p, (const double*)dfac.p, wavelet, (cpx<double>*)acc.p);
    SSQ_HIP(fft_any_batched<double>((cpx<double>*)acc.p, (cpx<double>*)work.p, x_len, 1, +1, nullptr));
    hipLaunchKernelGGL(icwt_finish_kernel, grd, blk, 0, nullptr, (const cpx<double>*)acc.p, (long long)x_len,
                       1.0 / (double)x_len, final_norm, x_mean, (double*)dx.p);
    SSQ_HIP(hipGetLastError());
  }
  SSQ_HIP(hipDeviceSynchronize());
  SSQ_HIP(hipMemcpy(x_out, dx.p, (size_t)(x_len * 8), hipMemcpyDeviceToHost));
  return 0;
}

/* ---- wavelet helper functions: host fp64, interleaved complex out[2*n] ---- */
int ssq_morlet(const double* w, int64_t n, double mu, double* out) {
  if (n > 0 && (!w || !out)) SSQ_FAIL("NULL pointer");
  morlet_psih(w, n, mu, out);
  return 0;
}
int ssq_morlet_freq(int64_t n, double scale, double mu, double* out) {
  if (n > 0 && !out) SSQ_FAIL("NULL pointer");
  const std::vector<double> xi = xifn_host(scale, n);
  morlet_psih(xi.data(), n, mu, out);
  return 0;
}
int ssq_morlet_time(int64_t n, double scale, double mu, double* out) {
  if (int rc = ssq_morlet_freq(n, scale, mu, out)) return rc;
  time_from_freq(out, n);
  return 0;
}
int ssq_gmw(const double* w, int64_t n, double gamma, double beta, const char* norm, int order, double* out) {
  if (gamma <= 0.0) SSQ_FAIL("gamma must be positive");                               // gmw.rs:246-254
  if (beta < 0.0) SSQ_FAIL("beta must be non-negative");
  if (order < 0) SSQ_FAIL("order must be non-negative");
  if (n > 0 && (!w || !out)) SSQ_FAIL("NULL pointer");
  gmw_psih(w, n, gamma, beta, is_bandpass(norm), order, out);
  return 0;
}
int ssq_gmw_freq(int64_t n, double scale, double gamma, double beta, const char* norm, int order, double* out) {
  if (n > 0 && !out) SSQ_FAIL("NULL pointer");
  const std::vector<double> xi = xifn_host(scale, n);
  gmw_psih(xi.data(), n, gamma, beta, is_bandpass(norm), order, out);                 // no validation (gmw.rs:265-289)
  return 0;
}
int ssq_gmw_time(int64_t n, double scale, double gamma, double beta, const char* norm, int order, double* out) {
  if (int rc = ssq_gmw_freq(n, scale, gamma, beta, norm, order, out)) return rc;
  time_from_freq(out, n);
  return 0;
}
int ssq_gmw_center_frequency(double gamma, double beta, const char* kind, double* out) {
  if (!out) SSQ_FAIL("NULL pointer");
  const std::string k = kind ? kind : "peak";
  if (k == "peak") {
    *out = std::pow(beta / gamma, 1.0 / gamma);                                       // gmw.rs:347-350
  } else if (k == "energy") {
    *out = (1.0 / std::pow(2.0, 1.0 / gamma)) * (gamma_fn((2.0 * beta + 2.0) / gamma) / gamma_fn((2.0 * beta + 1.0) / gamma));
  } else {
    SSQ_FAIL("Unknown center frequency kind: " + k);
  }
  return 0;
}

}  // extern "C"
