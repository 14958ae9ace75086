// api_upstream.hip -- the inverses and the host-side constants of the upstream-parity mode.
// Follows the vendored upstream /root/reference/old/ssqueezepy (the Python library the Rust crate was derived from):
//   istft        _stft.py:196-254, utils/stft_utils.py:141-191
//   issq_stft    _ssq_stft.py:139-198            issq_cwt   _ssq_cwt.py:313-378      (full inverses)
//   adm_ssq/cwt  utils/cwt_utils.py:28-63, :583-627   center_frequency('peak')  wavelets.py:691-716   p2up  common.py:32-51
// The forward transforms of the mode live beside the Rust-variant ones (api_stft.hip, api_cwt.hip: *_v entry points).
#include <cmath>
#include <complex>
#include <cstring>
#include <vector>

#include "../../include/ssq_hip.h"
#include "fft_generic.h"
#include "host_math.h"
#include "ssq_common.h"

using namespace ssq;

namespace {

// Sx [n_freqs][n_frames] -> Hermitian rows Z [n_frames][n] as numpy.irfft reads them: bins above n/2 are the
// conjugates, the imaginary parts of DC and (even n) Nyquist are ignored
template <typename T>
__global__ void istft_expand_kernel(const cpx<T>* __restrict__ Sx, int n, int n_frames, cpx<T>* __restrict__ Z) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= n_frames) return;
  for (int k = blockIdx.y; k < n; k += gridDim.y) {          // (grid.y is capped at 65535; n_fft may be larger)
    const int kk = k <= n / 2 ? k : n - k;
    cpx<T> v = Sx[(long long)kk * n_frames + f];
    if (k > n / 2) v.y = -v.y;
    if (k == 0 || (n % 2 == 0 && k == n / 2)) v.y = (T)0;
    Z[(long long)f * n + k] = v;
  }
}

// overlap-add of the windowed frames, window-norm division and unpadding in one gather per output sample
// (utils/stft_utils.py:178-191; _stft.py:238-252)
template <typename T>
__global__ void istft_ola_kernel(const cpx<T>* __restrict__ Y, int n, int n_frames, int hop, long long N, int modulated,
                                 const double* __restrict__ wpow, const double* __restrict__ wnorm, T* __restrict__ x) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const long long p = i + n / 2;                           // position in the padded signal (x[n_fft//2 : ...])
  long long f_lo = p - n + 1 <= 0 ? 0 : (p - n + 1 + hop - 1) / hop;
  long long f_hi = p / hop;
  if (f_hi > n_frames - 1) f_hi = n_frames - 1;
  const int sh = n - n / 2;                                // fftshift: xbuf[m] = y[(m + n - n//2) mod n]
  const double inv_n = 1.0 / (double)n;
  double acc = 0.0, wn = 0.0;
  for (long long f = f_lo; f <= f_hi; ++f) {
    const int m = (int)(p - f * hop);
    int src = modulated ? m + sh : m;
    if (src >= n) src -= n;
    acc += (double)Y[f * n + src].x * inv_n * wpow[m];
    wn += wnorm[m];
  }
  const double tiny = sizeof(T) == 4 ? 1.1754943508222875e-38 : 2.2250738585072014e-308;
  x[i] = (T)(wn > tiny ? acc / wn : acc);
}

template <typename T>
__global__ void issq_colsum_kernel(const cpx<T>* __restrict__ Tx, long long rows, long long cols, double scale,
                                   const double* __restrict__ row_scale, T* __restrict__ x) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= cols) return;
  double acc = 0.0;
  if (row_scale) {
    for (long long r = 0; r < rows; ++r) acc += (double)Tx[r * cols + j].x * row_scale[r];
  } else {
    for (long long r = 0; r < rows; ++r) acc += (double)Tx[r * cols + j].x;
  }
  x[j] = (T)(acc * scale);
}

template <typename T>
int istft_typed(const void* Sx, int64_t n_frames, const std::vector<double>& wpow, const std::vector<double>& wnorm,
                int64_t n, int64_t hop, int64_t N, int modulated, void* x_out) {
  const int64_t nf = n / 2 + 1;
  cpx<T>*d_S = nullptr, *d_Z = nullptr, *d_work = nullptr;
  double *d_wp = nullptr, *d_wn = nullptr;
  T* d_x = nullptr;
  int rc = 0;
  auto fail = [&](hipError_t e, const char* what) {
    if (e != hipSuccess && rc == 0) {
      set_error(std::string(what) + ": " + hipGetErrorString(e));
      rc = 2;
    }
    return e != hipSuccess;
  };
  do {
    if (fail(hipMalloc((void**)&d_S, sizeof(cpx<T>) * nf * n_frames), "hipMalloc")) break;
    if (fail(hipMalloc((void**)&d_Z, sizeof(cpx<T>) * n * n_frames), "hipMalloc")) break;
    const long long we = fft_work_elems(n, n_frames);
    if (fail(hipMalloc((void**)&d_work, sizeof(cpx<T>) * (we > 0 ? we : 1)), "hipMalloc")) break;
    if (fail(hipMalloc((void**)&d_wp, sizeof(double) * n), "hipMalloc")) break;
    if (fail(hipMalloc((void**)&d_wn, sizeof(double) * n), "hipMalloc")) break;
    if (fail(hipMalloc((void**)&d_x, sizeof(T) * N), "hipMalloc")) break;
    if (fail(hipMemcpy(d_S, Sx, sizeof(cpx<T>) * nf * n_frames, hipMemcpyHostToDevice), "hipMemcpy")) break;
    if (fail(hipMemcpy(d_wp, wpow.data(), sizeof(double) * n, hipMemcpyHostToDevice), "hipMemcpy")) break;
    if (fail(hipMemcpy(d_wn, wnorm.data(), sizeof(double) * n, hipMemcpyHostToDevice), "hipMemcpy")) break;
    hipLaunchKernelGGL(istft_expand_kernel<T>, dim3((unsigned)((n_frames + 255) / 256), (unsigned)(n < 65535 ? n : 65535)), dim3(256), 0,
                       nullptr,
                       d_S, (int)n, (int)n_frames, d_Z);
    if (fail(fft_any_batched<T>(d_Z, d_work, n, n_frames, +1, nullptr), "fft_any_batched")) break;
    hipLaunchKernelGGL(istft_ola_kernel<T>, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, nullptr, d_Z, (int)n,
                       (int)n_frames, (int)hop, (long long)N, modulated, d_wp, d_wn, d_x);
    if (fail(hipGetLastError(), "istft kernels")) break;
    if (fail(hipMemcpy(x_out, d_x, sizeof(T) * N, hipMemcpyDeviceToHost), "hipMemcpy")) break;
  } while (false);
  hipFree(d_S);
  hipFree(d_Z);
  hipFree(d_work);
  hipFree(d_wp);
  hipFree(d_wn);
  hipFree(d_x);
  return rc;
}

template <typename T>
int issq_typed(const void* Tx, int64_t rows, int64_t cols, double scale, const double* row_scale, void* x_out) {
  cpx<T>* d_T = nullptr;
  T* d_x = nullptr;
  double* d_r = nullptr;
  SSQ_HIP(hipMalloc((void**)&d_T, sizeof(cpx<T>) * rows * cols));
  hipError_t e = hipMalloc((void**)&d_x, sizeof(T) * cols);
  if (e == hipSuccess && row_scale) e = hipMalloc((void**)&d_r, sizeof(double) * rows);
  if (e == hipSuccess && row_scale) e = hipMemcpy(d_r, row_scale, sizeof(double) * rows, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d_T, Tx, sizeof(cpx<T>) * rows * cols, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(issq_colsum_kernel<T>, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, nullptr, d_T,
                       (long long)rows, (long long)cols, scale, d_r, d_x);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpy(x_out, d_x, sizeof(T) * cols, hipMemcpyDeviceToHost);
  hipFree(d_T);
  hipFree(d_x);
  hipFree(d_r);
  SSQ_HIP(e);
  return 0;
}

// ---- upstream wavelets in fp64 (host) ----
double gmw_l1(double w, double gamma, double beta) {             // _gmw.py:204-210
  if (!(w >= 0.0)) return 0.0;
  const double wc = std::exp((1.0 / gamma) * (std::log(beta) - std::log(gamma)));   // _gmw.py:611-657 morsefreq
  const double wcl = std::log(wc);
  if (w == 0.0) return 0.0;                                        // exp(beta * log 0) = exp(-inf)
  return 2.0 * std::exp(-beta * wcl + std::pow(wc, gamma) + beta * std::log(w) - std::pow(w, gamma));
}
double morlet_up(double w, double mu) {                          // wavelets.py:497-523
  const double cs = std::pow(1.0 + std::exp(-mu * mu) - 2.0 * std::exp(-0.75 * mu * mu), -0.5);
  const double ks = std::exp(-0.5 * mu * mu);
  return std::sqrt(2.0) * cs * std::pow(M_PI, 0.25) * (std::exp(-0.5 * (w - mu) * (w - mu)) - ks * std::exp(-0.5 * w * w));
}
double psih_up(int wavelet, double p0, double p1, double w) {
  return wavelet == SSQ_WAVELET_MORLET ? morlet_up(w, p0) : gmw_l1(w, p0, p1);
}

double trapz(const std::vector<double>& y, const std::vector<double>& t, size_t n) {
  double s = 0.0;
  for (size_t i = 1; i < n; ++i) s += 0.5 * (y[i] + y[i - 1]) * (t[i] - t[i - 1]);
  return s;
}

// utils/cwt_utils.py:583-627 for a real, non-negative integrand
double integrate_analytic(int wavelet, double p0, double p1, bool squared) {
  auto fn = [&](double w) {
    const double v = psih_up(wavelet, p0, p1, w);
    return (squared ? v * v : v) / w;
  };
  std::vector<double> t0(1000), a0(1000);
  for (int i = 0; i < 1000; ++i) {                                 // np.logspace(-15, -1, 1000)
    t0[i] = std::pow(10.0, -15.0 + 14.0 * (double)i / 999.0);
    a0[i] = fn(t0[i]);
  }
  const double int_nz = trapz(a0, t0, 1000);
  const int ms[4] = {1, 1, 4, 8};
  const double lims[4] = {1, 20, 80, 160};
  std::vector<double> t, arr;
  size_t keep = 0;
  for (int c = 0; c < 4; ++c) {
    const size_t n = (size_t)10000 * ms[c];
    t.assign(n, 0.0);
    arr.assign(n, 0.0);
    const double step = (0.1 - lims[c]) / (double)n;               // np.linspace(mxlim, .1, n, endpoint=False)[::-1]
    for (size_t i = 0; i < n; ++i) t[n - 1 - i] = lims[c] + (double)i * step;
    size_t mi = 0;
    double sum_abs = 0.0;
    for (size_t i = 0; i < n; ++i) {
      arr[i] = fn(t[i]);
      sum_abs += std::fabs(arr[i]);
      if (arr[i] > arr[mi]) mi = i;
    }
    size_t idx = n - 1 - mi;                                       // algos.py:616-622 on |arr[mi:]|, th = 1e-15
    for (size_t i = mi; i < n; ++i)
      if (std::fabs(arr[i]) < 1e-15) {
        idx = i - mi;
        break;
      }
    keep = idx + mi;
    if ((n - keep > (size_t)1000 * ms[c]) && sum_abs > 1e-5) break;
  }
  return trapz(arr, t, keep) + int_nz;
}

}  // namespace

extern "C" {

int ssq_istft_host(int dtype, const void* Sx, int64_t n_frames, const double* window, int64_t n_fft, int64_t hop,
                   int64_t n_signal, int modulated, int win_exp, void* x_out) {
  if (!Sx || !window || !x_out) SSQ_FAIL("NULL argument");
  if (dtype != SSQ_F32 && dtype != SSQ_F64) SSQ_FAIL("dtype must be SSQ_F32 or SSQ_F64");
  if (n_fft < 1 || hop < 1 || n_frames < 1 || n_signal < 1 || win_exp < 0) SSQ_FAIL("bad istft shape");
  if (n_fft > (1 << 24)) SSQ_FAIL("n_fft too large");
  if ((n_signal - 1) / hop + 1 != n_frames) SSQ_FAIL("Sx has the wrong number of frames for (N, hop_len)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) SSQ_FAIL("no HIP device visible (there is no CPU fallback)");
  std::vector<double> wpow((size_t)n_fft), wnorm((size_t)n_fft);
  for (int64_t i = 0; i < n_fft; ++i) {
    wpow[i] = win_exp == 0 ? 1.0 : std::pow(window[i], (double)win_exp);       // utils/stft_utils.py:159-162
    wnorm[i] = std::pow(window[i], (double)(win_exp + 1));                      // :186
  }
  return dtype == SSQ_F32 ? istft_typed<float>(Sx, n_frames, wpow, wnorm, n_fft, hop, n_signal, modulated, x_out)
                          : istft_typed<double>(Sx, n_frames, wpow, wnorm, n_fft, hop, n_signal, modulated, x_out);
}

int ssq_issq_host(int dtype, const void* Tx, int64_t rows, int64_t cols, double scale, const double* row_scale,
                  void* x_out) {
  if (!Tx || !x_out) SSQ_FAIL("NULL argument");
  if (dtype != SSQ_F32 && dtype != SSQ_F64) SSQ_FAIL("dtype must be SSQ_F32 or SSQ_F64");
  if (rows < 1 || cols < 1) SSQ_FAIL("empty Tx");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) SSQ_FAIL("no HIP device visible (there is no CPU fallback)");
  return dtype == SSQ_F32 ? issq_typed<float>(Tx, rows, cols, scale, row_scale, x_out)
                          : issq_typed<double>(Tx, rows, cols, scale, row_scale, x_out);
}

int ssq_upstream_adm(int wavelet, double p0, double p1, int which_cwt, double* out) {
  if (!out) SSQ_FAIL("out is NULL");
  if (wavelet != SSQ_WAVELET_GMW && wavelet != SSQ_WAVELET_MORLET) SSQ_FAIL("unknown wavelet");
  *out = integrate_analytic(wavelet, p0, p1, which_cwt != 0);
  return 0;
}

int ssq_upstream_center_frequency(int wavelet, double p0, double p1, double scale, int64_t n, double* wc) {
  if (!wc || n < 2) SSQ_FAIL("bad arguments");
  if (wavelet != SSQ_WAVELET_GMW && wavelet != SSQ_WAVELET_MORLET) SSQ_FAIL("unknown wavelet");
  const double h = (2.0 * M_PI) / (double)n;
  auto xi = [&](int64_t i) { return i <= n / 2 ? (double)i * h : (double)(i - n) * h; };   // wavelets.py:473-483
  // grid order of wavelets.py:950-962 (aifftshift), first maximum wins like np.argmax
  double best = -1.0, best_w = 0.0;
  auto visit = [&](int64_t i) {
    const double w = xi(i);
    const double v = psih_up(wavelet, p0, p1, scale * w);
    if (v * v > best) {
      best = v * v;
      best_w = w;
    }
  };
  if (n % 2 == 0) {
    for (int64_t i = n / 2 + 1; i < n; ++i) visit(i);
    for (int64_t i = 0; i <= n / 2; ++i) visit(i);
  } else {
    for (int64_t i = n / 2; i < n; ++i) visit(i);           // np.fft.ifftshift for odd n: starts at n//2
    for (int64_t i = 0; i < n / 2; ++i) visit(i);
  }
  *wc = best_w;
  return 0;
}

int ssq_upstream_p2up(int64_t n_signal, int64_t* n_up, int64_t* n1, int64_t* n2) {
  if (n_signal < 1 || !n_up || !n1 || !n2) SSQ_FAIL("bad arguments");
  host::p2up(n_signal, n_up, n1, n2);
  return 0;
}

}  // extern "C"
