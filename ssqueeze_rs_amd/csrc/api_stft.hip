// api_stft.hip -- STFT-family plans and host entry points of the C-ABI (include/ssq_hip.h).
// Replaces the PyO3 functions `stft` (rust/src/spectral/stft.rs:12-95) and `ssq_stft`
// (rust/src/spectral/ssq_stft.rs:72-313).
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ssq_hip.h"
#include <algorithm>
#include <mutex>

#include "fft_generic.h"
#include "host_cache.h"
#include "host_math.h"
#include "stft_kernels.h"

using namespace ssq;

struct ssq_stft_plan {
  int dtype = SSQ_F32;
  long long n_signal = 0;
  int n_fft = 0, hop = 0, n_freqs = 0, n_frames = 0, pad_left = 0;
  int padtype = 0, squeezing = 0;
  int variant = 0, rot = 0;        // upstream-parity mode (SSQ_VARIANT_*), frame rotation of the modulated STFT
  double fs = 1.0, gamma = 0.0;
  bool fused = false;
  bool fft_path = false;           // unfused, but through the batched any-length device FFT instead of direct sums
  int fft_len = 0;                 // length of the fused kernel's transforms: n_fft, or m >= 2*n_fft - 1 in Bluestein mode
  bool blue = false;
  int mr_np = 0;                   // mixed-radix mode (n_fft = 2^a 3^b 5^c 7^d 11^e 13^f): passes, radices 4 bits each (R - 1)
  unsigned mr_radix = 0;
  void* d_blue_b = nullptr;        // Bluestein: spectrum of the chirp filter / m
  void* d_blue_post = nullptr;     // Bluestein: output chirp
  int tile_frames = 0;
  int cu_count = 256;
  // host-side fp64 quantities (reference expressions)
  std::vector<double> ssq_freqs;   // ssq_stft.rs:42-54
  double sfs_step = 0, dw = 0;
  double alpha = 1.0;              // power of two: scales the derivative channel of the packed FFT
  // device tables (typed by dtype)
  void* d_tw = nullptr;
  void* d_win2 = nullptr;
  void* d_ssq_freqs = nullptr;
  // generic-path tables (double)
  double *d_g = nullptr, *d_gd = nullptr, *d_twre = nullptr, *d_twim = nullptr;
};

namespace {

// n = 2^a 3^b 5^c 7^d 11^e 13^f -> the pass list of fft_mixed.h.  Fewest passes first (every pass is one LDS round trip
// of the frame): 2s and 3s pair with 5s and 3s into the twiddle-free radices 10, 15, 12, 6, the other 2s group into
// 16 / 8 / 4 / 2; odd radices go first (the first pass writes with stride R: an odd stride spreads over the LDS banks).
// False when n has another prime factor or needs more than 8 passes.
bool mixed_radix_plan(int n, int& np, unsigned& packed) {
  int e2 = 0, e3 = 0, e5 = 0;
  std::vector<int> odd, even;
  while (n % 2 == 0) n /= 2, ++e2;
  while (n % 3 == 0) n /= 3, ++e3;
  while (n % 5 == 0) n /= 5, ++e5;
  for (int pr : {13, 11, 7})
    while (n % pr == 0) {
      n /= pr;
      odd.push_back(pr);
    }
  if (n != 1) {
    np = 0;
    return false;
  }
  while (e5 > 0 && e2 > 0) even.push_back(10), --e5, --e2;
  while (e5 > 0 && e3 > 0) odd.push_back(15), --e5, --e3;
  while (e3 > 0 && e2 >= 2) even.push_back(12), --e3, e2 -= 2;
  while (e3 > 0 && e2 >= 1) even.push_back(6), --e3, --e2;
  while (e5 > 0) odd.push_back(5), --e5;
  while (e3 > 0) odd.push_back(3), --e3;
  while (e2 >= 4 && e2 != 5) even.push_back(16), e2 -= 4;
  while (e2 >= 3) even.push_back(8), e2 -= 3;
  if (e2 == 2) even.push_back(4);
  if (e2 == 1) even.push_back(2);
  std::vector<int> r(odd);
  r.insert(r.end(), even.begin(), even.end());
  if (r.size() > 8) {
    np = 0;
    return false;
  }
  np = (int)r.size();
  packed = 0;
  for (size_t i = 0; i < r.size(); ++i) packed |= (unsigned)(r[i] - 1) << (4 * i);
  return true;
}

template <typename T>
int upload_tables(ssq_stft_plan* pl, const std::vector<double>& g, const std::vector<double>& gdfs) {
  const int n = pl->n_fft;
  const long double PI = 3.14159265358979323846264338327950288L;
  const int m = pl->fft_len > 0 ? pl->fft_len : n;             // fused kernel's transform length
  // (fp64 plans of the power-of-two kernel: behind the W_m table the passes' compact tables [mm][k] = exp(-2 pi i k mm / (NS R)),
  //  read coalesced by the 8-wave n_fft = 1024 kernel -- FusedCfg::SPLIT -- instead of gathered from W_m)
  long long tw_extra = 0;
  int logm = 0;
  while ((1 << logm) < m) ++logm;
  if (sizeof(T) == 8 && (1 << logm) == m && !pl->blue && pl->mr_np == 0)
    for (int P = 1; P < num_passes(logm); ++P) tw_extra += (long long)pass_radix(logm, P) * pass_ns(logm, P);
  std::vector<cpx<T>> tw((size_t)(m + tw_extra)), win2((size_t)m, cpx<T>{(T)0, (T)0});
  std::vector<double> twre((size_t)n), twim((size_t)n);
  for (int i = 0; i < n; ++i) {
    const long double ang = 2.0L * PI * (long double)i / (long double)n;
    twre[i] = (double)cosl(ang);
    twim[i] = (double)(-sinl(ang));
  }
  const int tw_n = pl->mr_np > 0 ? n : m;                      // mixed-radix mode: the W_n table itself
  for (int i = 0; i < m; ++i) {
    const long double ang = 2.0L * PI * (long double)(i % tw_n) / (long double)tw_n;
    tw[i] = {(T)cosl(ang), (T)(-sinl(ang))};
  }
  if (tw_extra > 0) {
    long long off = m;
    for (int P = 1; P < num_passes(logm); ++P) {
      const int R = pass_radix(logm, P), NS = pass_ns(logm, P);
      for (int mm = 0; mm < R; ++mm)
        for (int k = 0; k < NS; ++k) {
          const long double ang = 2.0L * PI * (long double)((long long)k * mm) / (long double)((long long)NS * R);
          tw[(size_t)(off + (long long)mm * NS + k)] = {(T)cosl(ang), (T)(-sinl(ang))};
        }
      off += (long long)R * NS;
    }
  }
  std::vector<host::cd> chirp((size_t)n);                      // exp(-i*pi*j^2/n), j^2 reduced mod 2n
  if (pl->blue) {
    for (long long j = 0; j < n; ++j) {
      const long double a = -PI * (long double)((j * j) % (2LL * n)) / (long double)n;
      chirp[j] = host::cd((double)cosl(a), (double)sinl(a));
    }
  }
  for (int i = 0; i < n; ++i) {
    host::cd w(0.5 * g[i], 0.5 * gdfs[i] * pl->alpha);         // halved: the unpack then needs no scaling
    if (pl->blue) w *= chirp[i];                               // x*w*chirp: still one real-by-complex multiply per sample
    win2[i] = {(T)w.real(), (T)w.imag()};
  }
  if (pl->blue) {
    std::vector<host::cd> B((size_t)m, host::cd(0, 0));
    B[0] = std::conj(chirp[0]);
    for (int j = 1; j < n; ++j) B[j] = B[m - j] = std::conj(chirp[j]);
    host::fft_pow2(B, -1);
    std::vector<cpx<T>> bh((size_t)m), post((size_t)n);
    for (int i = 0; i < m; ++i) bh[i] = {(T)(B[i].real() / (double)m), (T)(B[i].imag() / (double)m)};
    for (int i = 0; i < n; ++i) post[i] = {(T)chirp[i].real(), (T)chirp[i].imag()};
    SSQ_HIP(hipMalloc(&pl->d_blue_b, sizeof(cpx<T>) * m));
    SSQ_HIP(hipMalloc(&pl->d_blue_post, sizeof(cpx<T>) * n));
    SSQ_HIP(hipMemcpy(pl->d_blue_b, bh.data(), sizeof(cpx<T>) * m, hipMemcpyHostToDevice));
    SSQ_HIP(hipMemcpy(pl->d_blue_post, post.data(), sizeof(cpx<T>) * n, hipMemcpyHostToDevice));
  }
  std::vector<T> fr((size_t)pl->n_freqs);
  for (int i = 0; i < pl->n_freqs; ++i) fr[i] = (T)pl->ssq_freqs[i];
  SSQ_HIP(hipMalloc(&pl->d_tw, sizeof(cpx<T>) * tw.size()));
  SSQ_HIP(hipMalloc(&pl->d_win2, sizeof(cpx<T>) * m));
  SSQ_HIP(hipMalloc(&pl->d_ssq_freqs, sizeof(T) * pl->n_freqs));
  SSQ_HIP(hipMemcpy(pl->d_tw, tw.data(), sizeof(cpx<T>) * tw.size(), hipMemcpyHostToDevice));
  SSQ_HIP(hipMemcpy(pl->d_win2, win2.data(), sizeof(cpx<T>) * m, hipMemcpyHostToDevice));
  SSQ_HIP(hipMemcpy(pl->d_ssq_freqs, fr.data(), sizeof(T) * pl->n_freqs, hipMemcpyHostToDevice));
  SSQ_HIP(hipMalloc((void**)&pl->d_g, sizeof(double) * n));
  SSQ_HIP(hipMalloc((void**)&pl->d_gd, sizeof(double) * n));
  SSQ_HIP(hipMalloc((void**)&pl->d_twre, sizeof(double) * n));
  SSQ_HIP(hipMalloc((void**)&pl->d_twim, sizeof(double) * n));
  SSQ_HIP(hipMemcpy(pl->d_g, g.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  SSQ_HIP(hipMemcpy(pl->d_gd, gdfs.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  SSQ_HIP(hipMemcpy(pl->d_twre, twre.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  SSQ_HIP(hipMemcpy(pl->d_twim, twim.data(), sizeof(double) * n, hipMemcpyHostToDevice));
  return 0;
}

struct SigLayout {
  long long group = 1, group_stride = 0, sig_stride = 0;     // group_stride = 0: plain batch (n_signal apart)
};

template <typename T>
StftDev<T> make_dev(const ssq_stft_plan* pl, int out_kind, const void* d_x, void* d_out, long long batch,
                    const SigLayout& lay) {
  StftDev<T> p;
  p.x = (const T*)d_x;
  p.group = (int)lay.group;
  p.group_stride = lay.group_stride > 0 ? lay.group_stride : pl->n_signal;
  p.sig_stride = lay.sig_stride;
  p.out = (cpx<T>*)d_out;
  p.tw = (const cpx<T>*)pl->d_tw;
  p.win2 = (const cpx<T>*)pl->d_win2;
  p.ssq_freqs = (const T*)pl->d_ssq_freqs;
  p.n_signal = pl->n_signal;
  p.n_frames = pl->n_frames;
  p.n_freqs = pl->n_freqs;
  p.hop = pl->hop;
  p.pad_left = pl->pad_left;
  p.padtype = pl->padtype;
  const int F = pl->tile_frames > 0 ? pl->tile_frames : 1;
  p.tiles_per_signal = (pl->n_frames + F - 1) / F;
  p.ta0 = 0;
  p.ta_n = p.tiles_per_signal;
  p.tb0 = 0;
  p.total_tiles = (long long)p.tiles_per_signal * batch;
  p.out_kind = out_kind;
  p.squeezing = pl->squeezing;
  p.sfs_step = (T)pl->sfs_step;
  p.dw = (T)pl->dw;
  p.inv_dw = (T)(1.0 / pl->dw);
  p.gamma2 = (T)(pl->gamma * pl->gamma);
  p.gamma = (T)pl->gamma;
  p.variant = pl->variant;
  p.rot = pl->rot;
  p.leb_val = (T)((1.0 / (double)pl->n_freqs) * pl->dw);
  p.f_last = (T)pl->ssq_freqs[pl->n_freqs - 1];
  p.inv_alpha = (T)(1.0 / pl->alpha);
  p.two_pi_eff = (T)(6.283185307179586 * (pl->fused ? pl->alpha : 1.0));
  p.leb_unit = (T)(1.0 / (double)pl->n_freqs);
  p.n_eff = pl->fused ? ((pl->blue || pl->mr_np > 0) ? pl->n_fft : pl->fft_len) : pl->n_fft;
  p.blue_b = (const cpx<T>*)pl->d_blue_b;
  p.blue_post = (const cpx<T>*)pl->d_blue_post;
  p.mr_np = pl->mr_np;
  p.mr_radix = pl->mr_radix;
  {
    // keep  <=>  den >= g2 (fp32 values).  With BIG = 1/ulp(g2): (den - g2)*BIG is 0 at equality and <= -1 for every
    // representable den < g2, so clamp(den*BIG + (1 - g2*BIG)) is exactly the 0/1 mask (g2*BIG is an integer < 2^24).
    const float g2 = (float)(pl->gamma * pl->gamma);
    float big = 1.0f, bias = 1.0f;
    if (g2 > 0.0f && std::isfinite(g2)) {
      int e = 0;
      (void)std::frexp(g2, &e);                    // g2 = m * 2^e, m in [0.5, 1): ulp(g2) = 2^(e-24)
      int be = 24 - e;
      if (be > 126) be = 126;
      big = std::ldexp(1.0f, be);
      bias = 1.0f - g2 * big;
    }
    p.keep_big = big;
    p.keep_bias = bias;
  }
  {
    const char* ab = tune_env("SSQ_ABLATE");            // variant builds only
    p.ablate = ab ? std::atoi(ab) : 0;
    const char* st = tune_env("SSQ_STAMPS_PTR");   // diagnostic builds: device buffer address
    p.stamps = st ? (unsigned long long*)std::strtoull(st, nullptr, 0) : nullptr;
  }
  return p;
}

template <typename T>
int exec_typed(ssq_stft_plan* pl, int out_kind, const void* d_x, long long batch, void* d_out,
               void* d_ws, long long ws_bytes, hipStream_t stream, const SigLayout& lay) {
  StftDev<T> p = make_dev<T>(pl, out_kind, d_x, d_out, batch, lay);
  if (pl->fused) {
    SSQ_HIP(launch_stft_fused<T>(p, pl->fft_len, pl->cu_count, batch, stream));
    return 0;
  }
  const long long bins = batch * (long long)pl->n_freqs * pl->n_frames;
  const long long need = ssq_stft_plan_workspace_bytes(pl, batch, out_kind);
  if (need > 0 && (!d_ws || ws_bytes < need)) SSQ_FAIL("workspace too small for the unfused STFT path");
  GenericTabs tabs{pl->d_g, pl->d_gd, pl->d_twre, pl->d_twim};
  cpx<T>* ws = (cpx<T>*)d_ws;
  if (pl->fft_path) {
    // workspace: [Sx bins][dSx bins] (as the direct-sum path, when the output is not Sx itself), then Z and the FFT work
    const bool need_d = out_kind != SSQ_OUT_SX;
    cpx<T>* Sx = out_kind == SSQ_OUT_SX ? (cpx<T>*)d_out : ws;
    cpx<T>* dSx = need_d ? (out_kind == SSQ_OUT_DSX ? (cpx<T>*)d_out : ws + bins) : nullptr;
    cpx<T>* Z = ws + (out_kind == SSQ_OUT_SX ? 0 : (out_kind == SSQ_OUT_DSX ? bins : 2 * bins));
    cpx<T>* work = Z + (long long)pl->n_frames * pl->n_fft;
    const long long per = (long long)pl->n_freqs * pl->n_frames;
    for (long long b = 0; b < batch; ++b)
      SSQ_HIP(launch_fft_frames<T>(p, b, pl->n_fft, tabs, pl->alpha, Z, work, Sx + b * per, dSx ? dSx + b * per : nullptr,
                                   stream));
    if (out_kind == SSQ_OUT_TX || out_kind == SSQ_OUT_WK) {
      SSQ_HIP(hipMemsetAsync(d_out, 0, (size_t)bins * sizeof(cpx<T>), stream));
      SSQ_HIP(launch_reassign_cols<T>(p, Sx, dSx, batch, stream));
    }
    return 0;
  }
  if (out_kind == SSQ_OUT_SX) {
    SSQ_HIP(launch_dft_frames<T>(p, batch, pl->n_fft, tabs, (cpx<T>*)d_out, nullptr, stream));
  } else if (out_kind == SSQ_OUT_DSX) {
    SSQ_HIP(launch_dft_frames<T>(p, batch, pl->n_fft, tabs, ws, (cpx<T>*)d_out, stream));
  } else {
    cpx<T>* Sx = ws;
    cpx<T>* dSx = ws + bins;
    SSQ_HIP(launch_dft_frames<T>(p, batch, pl->n_fft, tabs, Sx, dSx, stream));
    SSQ_HIP(hipMemsetAsync(d_out, 0, (size_t)bins * sizeof(cpx<T>), stream));
    SSQ_HIP(launch_reassign_cols<T>(p, Sx, dSx, batch, stream));
  }
  return 0;
}

}  // namespace

extern "C" {

int ssq_stft_plan_create(ssq_stft_plan** plan, int dtype, int64_t n_signal, const double* window,
                         int64_t n_fft, int64_t hop, double fs, int padtype, int squeezing,
                         double gamma, int force_generic) {
  return ssq_stft_plan_create_v(plan, dtype, n_signal, window, n_fft, hop, fs, padtype, squeezing, gamma, force_generic,
                                SSQ_VARIANT_RUST);
}

int ssq_stft_plan_create_v(ssq_stft_plan** plan, int dtype, int64_t n_signal, const double* window,
                           int64_t n_fft, int64_t hop, double fs, int padtype, int squeezing,
                           double gamma, int force_generic, int variant) {
  if (!plan) SSQ_FAIL("plan is NULL");
  *plan = nullptr;
  if (dtype != SSQ_F32 && dtype != SSQ_F64) SSQ_FAIL("dtype must be SSQ_F32 or SSQ_F64");
  if (!window) SSQ_FAIL("window is NULL");
  if (n_fft > (1 << 24)) SSQ_FAIL("n_fft too large");
  int64_t nf = 0, nfr = 0;
  if (int rc = ssq_stft_shape(n_signal, n_fft, hop, &nf, &nfr)) return rc;
  if (nfr > 0x7fffffff) SSQ_FAIL("too many frames");
  ssq_stft_plan* pl = new ssq_stft_plan();
  pl->dtype = dtype;
  pl->n_signal = n_signal;
  pl->n_fft = (int)n_fft;
  pl->hop = (int)hop;
  pl->n_freqs = (int)nf;
  pl->n_frames = (int)nfr;
  pl->pad_left = (int)((n_fft - 1) / 2);                     // stft_utils.rs:21-22
  pl->padtype = padtype;
  pl->squeezing = squeezing;
  pl->fs = fs;
  pl->gamma = gamma < 0 ? 10.0 * 2.2204460492503131e-16 : gamma;   // ssq_stft.rs:258-261
  const bool ups = (variant & SSQ_VARIANT_UPSTREAM) != 0;
  pl->variant = variant;
  if (ups) {
    // old/ssqueezepy: padlength N + n_fft - 1 with the LARGER half on the left (utils/common.py:111-116), frames
    // rotated by n_fft/2 when modulated (utils/stft_utils.py:70-83), gamma = 10 eps of the dtype (_ssq_stft.py:103-104)
    pl->pad_left = (int)(n_fft / 2);
    pl->rot = (variant & SSQ_VARIANT_MODULATED) ? (int)(n_fft / 2) : 0;
    if (gamma < 0) pl->gamma = 10.0 * (dtype == SSQ_F64 ? 2.2204460492503131e-16 : 1.1920928955078125e-07);
    force_generic = force_generic ? 1 : 2;                   // unfused kernels only (2: through the device FFT)
  }
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
    pl->cu_count = prop.multiProcessorCount;
  // reference expressions, fp64
  pl->ssq_freqs.resize((size_t)nf);
  for (int64_t i = 0; i < nf; ++i)
    pl->ssq_freqs[i] = ((double)i * 0.5 * fs) / ((double)nf - 1.0);          // ssq_stft.rs:50
  pl->dw = nf > 1 ? pl->ssq_freqs[1] - pl->ssq_freqs[0] : 0.0;               // ssq_stft.rs:273
  pl->sfs_step = nf > 1 ? (0.5 * fs - 0.0) / (double)(nf - 1) : 0.0;         // ssq_stft.rs:255 (linspace)
  if (ups) {                                                                 // Sfs = ssq_freqs = np.linspace(0, fs/2, n)
    pl->ssq_freqs = host::np_linspace(0.0, 0.5 * fs, nf);                    // _ssq_stft.py:248-257, :117-118
    pl->dw = nf > 1 ? pl->ssq_freqs[1] - pl->ssq_freqs[0] : 0.0;             // ssqueezing.py:129-130
  }
  std::vector<double> g(window, window + n_fft);
  std::vector<double> gd = host::diff_window(g.data(), n_fft, ups);          // ssq_stft.rs:131-179 | _stft.py:293-299
  // ssq_stft.rs:208.  Upstream scales the diff-window by fs only inside its `modulated` branch (_stft.py:132-135):
  // reproduced as written
  if (!ups || (variant & SSQ_VARIANT_MODULATED))
    for (auto& v : gd) v *= fs;
  {
    // The fused kernel packs z = x*g + i*x*g'*fs*alpha into one complex FFT.  alpha (a power of
    // two, so exact) makes the two channels the same magnitude: otherwise the small channel
    // inherits the rounding error of the large one (fp32: |g'| ~ |g|*pi/n_fft, times fs).
    double mg = 0, md = 0;
    for (int64_t i = 0; i < n_fft; ++i) {
      mg = std::fmax(mg, std::fabs(g[i]));
      md = std::fmax(md, std::fabs(gd[i]));
    }
    if (mg > 0 && md > 0 && std::isfinite(mg / md)) {
      int e = 0;
      std::frexp(mg / md, &e);
      pl->alpha = std::ldexp(1.0, e - 1);
    }
  }
  const bool f32 = dtype == SSQ_F32;
  pl->fft_len = (int)n_fft;
  pl->fused = !force_generic && (f32 ? fused_supported<float>((int)n_fft) : fused_supported<double>((int)n_fft));
  if (!pl->fused && !force_generic && !host::is_pow2(n_fft) && n_fft >= 24 && n_fft <= 4096 &&
      mixed_radix_plan((int)n_fft, pl->mr_np, pl->mr_radix)) {
    // rustfft plans any length (stft.rs:43-44): lengths with prime factors <= 13 run mixed-radix passes inside the fused
    // kernel of the next power of two (fft_mixed.h) ...
    int m = 64;
    while (m < (int)n_fft) m <<= 1;
    pl->fft_len = m;
    pl->fused = true;
  } else if (!pl->fused && !force_generic && !host::is_pow2(n_fft) && n_fft >= 24 && n_fft <= 2048) {
    // ... and the others Bluestein's chirp-z, through two of its power-of-two transforms of length
    // m >= 2*n_fft - 1 (m <= 4096)
    pl->mr_np = 0;
    int m = 64;
    while (m < 2 * (int)n_fft - 1) m <<= 1;
    pl->fft_len = m;
    pl->blue = true;
    pl->fused = true;
  }
  pl->tile_frames = pl->fused ? (f32 ? fused_tile_frames<float>(pl->fft_len) : fused_tile_frames<double>(pl->fft_len)) : 1;
  // everything else that is long enough to matter: the batched any-length device FFT (force_generic keeps the direct
  // sums as the independent second implementation of the parity tests)
  pl->fft_path = !pl->fused && (force_generic == 0 || force_generic == 2) && n_fft >= 24;
  int rc = f32 ? upload_tables<float>(pl, g, gd) : upload_tables<double>(pl, g, gd);
  if (rc) {
    ssq_stft_plan_destroy(pl);
    return rc;
  }
  *plan = pl;
  return 0;
}

int ssq_stft_plan_destroy(ssq_stft_plan* pl) {
  if (!pl) return 0;
  hipFree(pl->d_tw);
  hipFree(pl->d_win2);
  hipFree(pl->d_ssq_freqs);
  hipFree(pl->d_blue_b);
  hipFree(pl->d_blue_post);
  hipFree(pl->d_g);
  hipFree(pl->d_gd);
  hipFree(pl->d_twre);
  hipFree(pl->d_twim);
  delete pl;
  return 0;
}

int ssq_stft_plan_is_fused(const ssq_stft_plan* pl) { return pl && pl->fused ? 1 : 0; }

int64_t ssq_stft_plan_workspace_bytes(const ssq_stft_plan* pl, int64_t batch, int out_kind) {
  if (!pl || pl->fused) return 0;
  const int64_t elem = (pl->dtype == SSQ_F32 ? 8 : 16);
  const int64_t bins = batch * (int64_t)pl->n_freqs * pl->n_frames;
  int64_t extra = 0;
  if (pl->fft_path) extra = ((int64_t)pl->n_frames * pl->n_fft + fft_work_elems(pl->n_fft, pl->n_frames)) * elem;
  if (out_kind == SSQ_OUT_SX) return extra;
  if (out_kind == SSQ_OUT_DSX) return bins * elem + extra;
  return 2 * bins * elem + extra;
}

static int exec_any(ssq_stft_plan* pl, int out_kind, const void* d_x, int64_t batch, void* d_out,
                    void* d_workspace, int64_t workspace_bytes, void* stream, const SigLayout& lay) {
  if (!pl) SSQ_FAIL("plan is NULL");
  if (batch <= 0) return 0;
  if (batch > 0x7fffffff) SSQ_FAIL("batch too large");
  if (!d_x || !d_out) SSQ_FAIL("device pointer is NULL");
  if (out_kind < SSQ_OUT_TX || out_kind > SSQ_OUT_WK) SSQ_FAIL("bad out_kind");
  if ((out_kind == SSQ_OUT_TX || out_kind == SSQ_OUT_WK) && pl->n_freqs < 2)
    SSQ_FAIL("index out of bounds: ssq_freqs[1] with fewer than 2 bins (ssq_stft.rs:273)");
  if (pl->dtype == SSQ_F32)
    return exec_typed<float>(pl, out_kind, d_x, batch, d_out, d_workspace, workspace_bytes, (hipStream_t)stream, lay);
  return exec_typed<double>(pl, out_kind, d_x, batch, d_out, d_workspace, workspace_bytes, (hipStream_t)stream, lay);
}

int ssq_stft_plan_exec(ssq_stft_plan* pl, int out_kind, const void* d_x, int64_t batch, void* d_out,
                       void* d_workspace, int64_t workspace_bytes, void* stream) {
  return exec_any(pl, out_kind, d_x, batch, d_out, d_workspace, workspace_bytes, stream, SigLayout{});
}

int ssq_stft_plan_exec_strided(ssq_stft_plan* pl, int out_kind, const void* d_x, int64_t n_groups, int64_t group,
                               int64_t group_stride, int64_t sig_stride, void* d_out, void* d_workspace,
                               int64_t workspace_bytes, void* stream) {
  if (n_groups <= 0 || group <= 0) return 0;
  if (group_stride <= 0 || sig_stride < 0) SSQ_FAIL("bad strides");
  SigLayout lay;
  lay.group = group;
  lay.group_stride = group_stride;
  lay.sig_stride = sig_stride;
  return exec_any(pl, out_kind, d_x, n_groups * group, d_out, d_workspace, workspace_bytes, stream, lay);
}

}  // extern "C"

// ---- host-pointer entry points: cached plans, cached device buffers, a two-stream pipeline ----------------------
namespace {

struct PlanKey {
  int dtype, padtype, squeezing, variant;
  int64_t n_signal, n_fft, hop;
  double fs, gamma;
  std::vector<double> window;
  bool operator==(const PlanKey& o) const {
    return dtype == o.dtype && padtype == o.padtype && squeezing == o.squeezing && variant == o.variant &&
           n_signal == o.n_signal &&
           n_fft == o.n_fft && hop == o.hop && fs == o.fs && gamma == o.gamma && window == o.window;
  }
};
struct CachedPlan {
  PlanKey key;
  ssq_stft_plan* pl;
  int dev;
};
std::vector<CachedPlan> g_plans;               // most recently used first; guarded by hostpath::mutex()
constexpr size_t kMaxPlans = 8;

int cached_plan(const PlanKey& key, ssq_stft_plan** out) {
  int dev = 0;
  SSQ_HIP(hipGetDevice(&dev));
  for (size_t i = 0; i < g_plans.size(); ++i) {
    if (g_plans[i].dev == dev && g_plans[i].key == key) {
      CachedPlan c = g_plans[i];
      g_plans.erase(g_plans.begin() + (long)i);
      g_plans.insert(g_plans.begin(), c);
      *out = c.pl;
      return 0;
    }
  }
  ssq_stft_plan* pl = nullptr;
  if (int rc = ssq_stft_plan_create_v(&pl, key.dtype, key.n_signal, key.window.data(), key.n_fft, key.hop, key.fs,
                                      key.padtype, key.squeezing, key.gamma, 0, key.variant))
    return rc;
  g_plans.insert(g_plans.begin(), CachedPlan{key, pl, dev});
  while (g_plans.size() > kMaxPlans) {
    ssq_stft_plan_destroy(g_plans.back().pl);
    g_plans.pop_back();
  }
  *out = pl;
  return 0;
}

}  // namespace

namespace ssq {
namespace hostpath {
void clear_stft_plans() {
  for (auto& c : g_plans) ssq_stft_plan_destroy(c.pl);
  g_plans.clear();
}
}  // namespace hostpath
}  // namespace ssq

// Shared body of the two host entry points.  The batch is cut into groups of signals; group g runs on stream g % 2:
// H2D of its samples, the kernels of every requested output, D2H of each result -- so one group's copies overlap the
// other's kernels and, with results in pinned memory (ssq_pinned_alloc: what `_rs.*` hands to NumPy), the host
// thread only ever waits for the upload staging of pageable inputs.
static int run_host(int dtype, const void* x, int64_t batch, int64_t n_signal, const double* window,
                    int64_t n_fft, int64_t hop, double fs, int padtype, int squeezing, double gamma,
                    int n_out, const int* kinds, void* const* outs, int variant = SSQ_VARIANT_RUST) {
  if (!x) SSQ_FAIL("x is NULL");
  if (batch <= 0) SSQ_FAIL("batch must be positive");
  if (!window) SSQ_FAIL("window is NULL");
  if (n_fft <= 0) SSQ_FAIL("n_fft must be positive");
  std::lock_guard<std::mutex> lk(hostpath::mutex());
  PlanKey key{dtype, padtype, squeezing, variant, n_signal, n_fft, hop, fs, gamma,
              std::vector<double>(window, window + n_fft)};
  ssq_stft_plan* pl = nullptr;
  if (int rc = cached_plan(key, &pl)) return rc;
  const int64_t esz = dtype == SSQ_F32 ? 4 : 8;
  const int64_t bins1 = (int64_t)pl->n_freqs * pl->n_frames;            // per signal
  const int64_t in1 = n_signal * esz, out1 = bins1 * 2 * esz;
  // group size: enough work per launch for small signals, a few MB per copy for big ones
  int64_t grp = (8LL << 20) / (out1 > 0 ? out1 : 1);
  if (grp < 1) grp = 1;
  if (grp > batch) grp = batch;
  int64_t ws1 = 0;
  for (int i = 0; i < n_out; ++i)
    if (outs[i]) ws1 = std::max<int64_t>(ws1, ssq_stft_plan_workspace_bytes(pl, grp, kinds[i]));
  void *d_x = nullptr, *d_out = nullptr, *d_ws[2] = {nullptr, nullptr};
  if (int rc = hostpath::scratch(hostpath::SLOT_X, batch * in1, &d_x)) return rc;
  if (int rc = hostpath::scratch(hostpath::SLOT_OUT, 2 * grp * out1, &d_out)) return rc;     // one slot per stream
  if (ws1 > 0) {
    if (int rc = hostpath::scratch(hostpath::SLOT_WS0, ws1, &d_ws[0])) return rc;
    if (int rc = hostpath::scratch(hostpath::SLOT_WS1, ws1, &d_ws[1])) return rc;
  }
  hipStream_t st[2] = {hostpath::stream(0), hostpath::stream(1)};
  if (!st[0] || !st[1]) SSQ_FAIL("hipStreamCreate failed");
  int rc = 0;
  int64_t g = 0;
  for (int64_t b0 = 0; b0 < batch && rc == 0; b0 += grp, ++g) {
    const int64_t nb = std::min<int64_t>(grp, batch - b0);
    const int s = (int)(g & 1);
    char* dxg = (char*)d_x + b0 * in1;
    char* dog = (char*)d_out + (int64_t)s * grp * out1;
    SSQ_HIP(hipMemcpyAsync(dxg, (const char*)x + b0 * in1, (size_t)(nb * in1), hipMemcpyHostToDevice, st[s]));
    for (int i = 0; i < n_out && rc == 0; ++i) {
      if (!outs[i]) continue;
      rc = ssq_stft_plan_exec(pl, kinds[i], dxg, nb, dog, d_ws[s], ws1, st[s]);
      if (rc) break;
      SSQ_HIP(hipMemcpyAsync((char*)outs[i] + b0 * out1, dog, (size_t)(nb * out1), hipMemcpyDeviceToHost, st[s]));
    }
  }
  const hipError_t e0 = hipStreamSynchronize(st[0]), e1 = hipStreamSynchronize(st[1]);
  if (rc) return rc;
  SSQ_HIP(e0);
  SSQ_HIP(e1);
  return 0;
}

extern "C" {

int ssq_stft_host(int dtype, const void* x, int64_t batch, int64_t n_signal, const double* window,
                  int64_t n_fft, int64_t hop, int padtype, void* Sx, double* freqs) {
  if (!Sx) SSQ_FAIL("Sx is NULL");
  const int kinds[1] = {SSQ_OUT_SX};
  void* outs[1] = {Sx};
  if (int rc = run_host(dtype, x, batch, n_signal, window, n_fft, hop, 1.0, padtype, 0, -1.0, 1, kinds, outs))
    return rc;
  if (freqs) {                                               // stft.rs:40 linspace(0, 0.5, n_freqs)
    const int64_t nf = n_fft / 2 + 1;
    const double step = nf > 1 ? 0.5 / (double)(nf - 1) : 0.0;
    for (int64_t i = 0; i < nf; ++i) freqs[i] = 0.0 + step * (double)i;
  }
  return 0;
}

int ssq_ssq_stft_host(int dtype, const void* x, int64_t batch, int64_t n_signal, const double* window,
                      int64_t n_fft, int64_t hop, double fs, int padtype, int squeezing, double gamma,
                      void* Tx, double* ssq_freqs, void* dbg_Sx, void* dbg_dSx, void* dbg_wk) {
  if (!Tx) SSQ_FAIL("Tx is NULL");
  if (n_fft / 2 + 1 < 2) SSQ_FAIL("index out of bounds: ssq_freqs[1] with fewer than 2 bins (ssq_stft.rs:273)");
  const int kinds[4] = {SSQ_OUT_TX, SSQ_OUT_SX, SSQ_OUT_DSX, SSQ_OUT_WK};
  void* outs[4] = {Tx, dbg_Sx, dbg_dSx, dbg_wk};
  if (int rc = run_host(dtype, x, batch, n_signal, window, n_fft, hop, fs, padtype, squeezing, gamma, 4, kinds, outs))
    return rc;
  if (ssq_freqs) {
    const int64_t nf = n_fft / 2 + 1;
    for (int64_t i = 0; i < nf; ++i) ssq_freqs[i] = ((double)i * 0.5 * fs) / ((double)nf - 1.0);
  }
  return 0;
}

// ---- upstream-parity mode: the same pipeline on the unfused kernels with the variant's numerics ----
int ssq_stft_host_v(int dtype, const void* x, int64_t batch, int64_t n_signal, const double* window, int64_t n_fft,
                    int64_t hop, double fs, int padtype, int variant, void* Sx, void* dSx) {
  if (!Sx) SSQ_FAIL("Sx is NULL");
  const int kinds[2] = {SSQ_OUT_SX, SSQ_OUT_DSX};
  void* outs[2] = {Sx, dSx};
  return run_host(dtype, x, batch, n_signal, window, n_fft, hop, fs, padtype, 0, -1.0, 2, kinds, outs, variant);
}

int ssq_ssq_stft_host_v(int dtype, const void* x, int64_t batch, int64_t n_signal, const double* window, int64_t n_fft,
                        int64_t hop, double fs, int padtype, int squeezing, double gamma, int variant, void* Tx,
                        double* ssq_freqs, void* Sx, void* dSx, void* wk) {
  if (!Tx) SSQ_FAIL("Tx is NULL");
  if (n_fft / 2 + 1 < 2) SSQ_FAIL("ssq_stft needs at least 2 frequency bins");
  const int kinds[4] = {SSQ_OUT_TX, SSQ_OUT_SX, SSQ_OUT_DSX, SSQ_OUT_WK};
  void* outs[4] = {Tx, Sx, dSx, wk};
  if (int rc = run_host(dtype, x, batch, n_signal, window, n_fft, hop, fs, padtype, squeezing, gamma, 4, kinds, outs,
                        variant))
    return rc;
  if (ssq_freqs) {
    const int64_t nf = n_fft / 2 + 1;
    if (variant & SSQ_VARIANT_UPSTREAM) {                      // Sfs, reversed with flipud (ssqueezing.py:199-205)
      const std::vector<double> f = host::np_linspace(0.0, 0.5 * fs, nf);
      for (int64_t i = 0; i < nf; ++i) ssq_freqs[i] = (variant & SSQ_VARIANT_FLIPUD) ? f[nf - 1 - i] : f[i];
    } else {
      for (int64_t i = 0; i < nf; ++i) ssq_freqs[i] = ((double)i * 0.5 * fs) / ((double)nf - 1.0);
    }
  }
  return 0;
}

}  // extern "C"
