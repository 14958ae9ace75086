/*
 * ssq_ref.c -- plain-C restatement of the reference's `ssq_stft` CPU path, with the SAME
 * structure as the Rust code so that timing it is a fair stand-in for "the reference
 * Rust+Rayon path" (which cannot be built here: no Rust toolchain).
 *
 * TEST INFRASTRUCTURE ONLY: used by tests/ (checked against the NumPy oracle) and by the
 * `cpu_baseline` leg of bench.py.  Nothing under ssqueeze_rs_amd/ links or loads this.
 * PARITY UNPINNED (see oracle/ssq_oracle.py): no golden vectors exist in the reference.
 *
 * Structure followed, rust/src/spectral/ssq_stft.rs:
 *   :124-128 pad (stft_utils.rs:19-65)      serial
 *   :131-179 spectral diff-window           serial, 2 FFTs of n_fft
 *   :191-244 per-frame TWO complex FFTs     parallel over frames (Rayon -> OpenMP)
 *   :247-252 gather into Sx,dSx             serial
 *   :264     phase_stft (:11-39)            serial
 *   :276-301 reassignment                   serial; mode 0 = the reference's linear scan over all
 *                                           n_freqs bins per element (O(n_frames*n_freqs^2));
 *                                           mode 1 = "optimised CPU": arithmetic bin + exact
 *                                           neighbour check, columns in parallel
 * FFT: rustfft (not in tree) is restated as an iterative radix-2 (power-of-two lengths) or a
 * direct O(n^2) DFT (other lengths); both unnormalised like rustfft.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct { double re, im; } cd;

static int is_pow2(long n) { return n > 0 && (n & (n - 1)) == 0; }

/* twiddles w[i] = exp(sign*2*pi*i*k/n), k < n/2 (pow2) or k < n (general) */
static void make_tw(cd* w, long n, int sign) {
  for (long i = 0; i < n; ++i) {
    double a = sign * 2.0 * M_PI * (double)i / (double)n;
    w[i].re = cos(a);
    w[i].im = sin(a);
  }
}

static void fft_inplace(cd* a, long n, const cd* w /* n entries */, cd* scratch) {
  if (is_pow2(n)) {
    for (long i = 1, j = 0; i < n; ++i) {
      long bit = n >> 1;
      for (; j & bit; bit >>= 1) j ^= bit;
      j ^= bit;
      if (i < j) { cd t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (long len = 2; len <= n; len <<= 1) {
      long step = n / len;
      for (long i = 0; i < n; i += len) {
        for (long k = 0; k < len / 2; ++k) {
          cd u = a[i + k], x = a[i + k + len / 2], t = w[k * step], v;
          v.re = x.re * t.re - x.im * t.im;
          v.im = x.re * t.im + x.im * t.re;
          a[i + k].re = u.re + v.re; a[i + k].im = u.im + v.im;
          a[i + k + len / 2].re = u.re - v.re; a[i + k + len / 2].im = u.im - v.im;
        }
      }
    }
  } else {
    for (long k = 0; k < n; ++k) {
      double sr = 0, si = 0;
      long idx = 0;
      for (long j = 0; j < n; ++j) {
        sr += a[j].re * w[idx].re - a[j].im * w[idx].im;
        si += a[j].re * w[idx].im + a[j].im * w[idx].re;
        idx += k; if (idx >= n) idx -= n;
      }
      scratch[k].re = sr; scratch[k].im = si;
    }
    memcpy(a, scratch, (size_t)n * sizeof(cd));
  }
}

/* stft_utils.rs:19-65 */
static double* pad_signal(const double* x, long n, long n_fft, int padtype, long* out_len) {
  long pad = n_fft - 1, pl = pad / 2, pr = pad - pl;
  double* p = (double*)calloc((size_t)(n + pad), sizeof(double));
  memcpy(p + pl, x, (size_t)n * sizeof(double));
  if (padtype == 0) {
    for (long i = 0; i < pl; ++i) { long m = pl - i; if (m < n) p[i] = x[m]; }
    for (long i = 0; i < pr; ++i) { long m = n - 2 - i; if (m >= 0 && m < n) p[n + pl + i] = x[m]; }
  }
  *out_len = n + pad;
  return p;
}

/*
 * Tx: interleaved complex [n_freqs][n_frames]; ssq_freqs: [n_freqs]; k_out (nullable): int [n_freqs][n_frames]
 * (-1 where skipped).  Returns 0 on success.
 */
int ssq_ref_ssq_stft(const double* x, long n, const double* window /* sized to n_fft */, long n_fft, long hop,
                     double fs, int padtype, int squeezing, double gamma, int mode, int nthreads,
                     double* Tx, double* ssq_freqs, int* k_out) {
  if (n <= 0 || n_fft < 2 || hop <= 0) return 1;
  if (gamma < 0) gamma = 10.0 * 2.2204460492503131e-16;          /* :258-261 */
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  long plen = 0;
  double* padded = pad_signal(x, n, n_fft, padtype, &plen);      /* :124-128 */
  long n_frames = (plen - n_fft) / hop + 1;                      /* :183 */
  long n_freqs = n_fft / 2 + 1;                                  /* :184 */
  cd* wf = (cd*)malloc((size_t)n_fft * sizeof(cd));
  cd* wi = (cd*)malloc((size_t)n_fft * sizeof(cd));
  make_tw(wf, n_fft, -1);
  make_tw(wi, n_fft, +1);

  /* :131-179 diff window */
  double* dwin = (double*)malloc((size_t)n_fft * sizeof(double));
  {
    cd* W = (cd*)malloc((size_t)n_fft * sizeof(cd));
    cd* scr = (cd*)malloc((size_t)n_fft * sizeof(cd));
    for (long i = 0; i < n_fft; ++i) { W[i].re = window[i]; W[i].im = 0.0; }
    fft_inplace(W, n_fft, wf, scr);
    for (long i = 0; i < n_fft; ++i) {
      double f = (i < n_fft / 2 + 1) ? (double)i : (double)i - (double)n_fft;
      f *= 2.0 * M_PI / (double)n_fft;
      double re = W[i].re, im = W[i].im;
      W[i].re = -im * f; W[i].im = re * f;
    }
    fft_inplace(W, n_fft, wi, scr);
    double scale = 1.0 / (double)n_fft;
    for (long i = 0; i < n_fft; ++i) dwin[i] = W[i].re * scale;
    free(W); free(scr);
  }

  cd* Sx = (cd*)malloc((size_t)(n_freqs * n_frames) * sizeof(cd));
  cd* dSx = (cd*)malloc((size_t)(n_freqs * n_frames) * sizeof(cd));
  /* :191-244 frames in parallel, two FFTs each; results kept per frame then gathered (:247-252) */
  cd* fr_s = (cd*)malloc((size_t)(n_frames * n_freqs) * sizeof(cd));
  cd* fr_d = (cd*)malloc((size_t)(n_frames * n_freqs) * sizeof(cd));
#pragma omp parallel
  {
    cd* a = (cd*)malloc((size_t)n_fft * sizeof(cd));
    cd* b = (cd*)malloc((size_t)n_fft * sizeof(cd));
    cd* scr = (cd*)malloc((size_t)n_fft * sizeof(cd));
#pragma omp for schedule(static)
    for (long f = 0; f < n_frames; ++f) {
      const double* seg = padded + f * hop;
      for (long i = 0; i < n_fft; ++i) {
        a[i].re = seg[i] * window[i]; a[i].im = 0.0;             /* :202 */
        b[i].re = seg[i] * dwin[i] * fs; b[i].im = 0.0;          /* :208 */
      }
      fft_inplace(a, n_fft, wf, scr);                            /* :226 */
      fft_inplace(b, n_fft, wf, scr);                            /* :227 */
      memcpy(fr_s + f * n_freqs, a, (size_t)n_freqs * sizeof(cd));
      memcpy(fr_d + f * n_freqs, b, (size_t)n_freqs * sizeof(cd));
    }
    free(a); free(b); free(scr);
  }
  for (long f = 0; f < n_frames; ++f)                            /* :247-252 serial strided gather */
    for (long i = 0; i < n_freqs; ++i) {
      Sx[i * n_frames + f] = fr_s[f * n_freqs + i];
      dSx[i * n_frames + f] = fr_d[f * n_freqs + i];
    }
  free(fr_s); free(fr_d);

  /* :255 Sfs = linspace(0, fs/2, n_freqs); :264 phase_stft (serial, row-major) */
  double sfs_step = (0.5 * fs - 0.0) / (double)(n_freqs - 1);
  double* w = (double*)malloc((size_t)(n_freqs * n_frames) * sizeof(double));
  for (long i = 0; i < n_freqs; ++i) {
    double sfs = 0.0 + sfs_step * (double)i;
    for (long j = 0; j < n_frames; ++j) {
      cd s = Sx[i * n_frames + j], d = dSx[i * n_frames + j];
      if (hypot(s.re, s.im) < gamma) {
        w[i * n_frames + j] = INFINITY;
      } else {
        double pd = (d.im * s.re - d.re * s.im) / ((s.re * s.re + s.im * s.im) * 6.283185307179586);
        w[i * n_frames + j] = fabs(sfs - pd);
      }
    }
  }
  for (long i = 0; i < n_freqs; ++i) ssq_freqs[i] = ((double)i * 0.5 * fs) / ((double)n_freqs - 1.0);  /* :50 */
  double dw = ssq_freqs[1] - ssq_freqs[0];                       /* :273 */
  memset(Tx, 0, (size_t)(2 * n_freqs * n_frames) * sizeof(double));
  double leb = 1.0 / (double)n_freqs;

  if (mode == 0) {
    /* :276-301 the reference's serial loops, linear scan for the nearest bin */
    for (long j = 0; j < n_frames; ++j) {
      for (long i = 0; i < n_freqs; ++i) {
        double wv = w[i * n_frames + j];
        long k = -1;
        if (!isinf(wv)) {
          k = 0;
          double min_dist = INFINITY;
          for (long idx = 0; idx < n_freqs; ++idx) {
            double dist = fabs(wv - ssq_freqs[idx]);
            if (dist < min_dist) { min_dist = dist; k = idx; }
          }
          double wr = squeezing == 1 ? leb : Sx[i * n_frames + j].re;
          double wim = squeezing == 1 ? 0.0 : Sx[i * n_frames + j].im;
          Tx[2 * (k * n_frames + j)] += wr * dw;
          Tx[2 * (k * n_frames + j) + 1] += wim * dw;
        }
        if (k_out) k_out[i * n_frames + j] = (int)k;
      }
    }
  } else {
    /* optimised CPU: arithmetic candidate + exact first-min over the neighbours, columns in parallel */
#pragma omp parallel for schedule(static)
    for (long j = 0; j < n_frames; ++j) {
      for (long i = 0; i < n_freqs; ++i) {
        double wv = w[i * n_frames + j];
        long k = -1;
        if (!isinf(wv)) {
          if (wv != wv) {
            k = 0;
          } else if (wv > ssq_freqs[n_freqs - 1]) {
            double target = fabs(wv - ssq_freqs[n_freqs - 1]);
            k = n_freqs - 1;
            while (k > 0 && fabs(wv - ssq_freqs[k - 1]) == target) --k;
          } else {
            double t = wv / dw;
            long c = (long)llrint(t);
            long k0 = c - 2 < 0 ? 0 : c - 2, k1 = c + 2 > n_freqs - 1 ? n_freqs - 1 : c + 2;
            double best = INFINITY;
            k = 0;
            for (long idx = k0; idx <= k1; ++idx) {
              double dist = fabs(wv - ssq_freqs[idx]);
              if (dist < best) { best = dist; k = idx; }
            }
          }
          double wr = squeezing == 1 ? leb : Sx[i * n_frames + j].re;
          double wim = squeezing == 1 ? 0.0 : Sx[i * n_frames + j].im;
          Tx[2 * (k * n_frames + j)] += wr * dw;
          Tx[2 * (k * n_frames + j) + 1] += wim * dw;
        }
        if (k_out) k_out[i * n_frames + j] = (int)k;
      }
    }
  }
  free(w); free(Sx); free(dSx); free(dwin); free(wf); free(wi); free(padded);
  return 0;
}

int ssq_ref_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------------------------------
 * ssq_cwt, the same way: the structure of rust/src/spectral/ssq_cwt.rs (bench.py's cpu_baseline of the C4 leg)
 *   :339-350  forward FFT of the padded signal (utils/array.rs:9-11, :52-98)        serial
 *   :365-423  per scale: wavelet (cwt.rs:492-547), TWO inverse FFTs of length P       parallel over scales
 *   :434-435  unpad                                                                   (inside the loop here)
 *   :15-47    phase_cwt                                                               parallel over rows
 *   :116-222  ssqueeze: bin formula, drop out-of-range, flipud, Tx += Wx              parallel over columns
 * wavelet: 1 = morlet (mu = 6), else the un-normalised GMW.  Tx: interleaved complex [na][n]; ssq_freqs: [na].
 * Memory: Wx, dWx, Tx of na * n complex doubles each (C4: 3 x 4.3 GB).
 */
static long next_pow2_ceil(long n) {
  long p = 1;
  while (p < n) p <<= 1;
  return p;
}

int ssq_ref_ssq_cwt(const double* x, long n, const double* scales, long na, int wavelet, double dt, int flipud,
                    double gamma, int nthreads, double* Tx, double* ssq_freqs) {
  if (n < 1 || na < 2) return 1;
#ifdef _OPENMP
  if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
  const long P = next_pow2_ceil(n + n / 2);                       /* utils/array.rs:9-11 with cwt.rs:87 */
  const long n1 = (P - n) / 2;
  cd* xh = (cd*)calloc((size_t)P, sizeof(cd));
  cd* twf = (cd*)malloc((size_t)P * sizeof(cd));
  cd* twi = (cd*)malloc((size_t)P * sizeof(cd));
  cd* Wx = (cd*)malloc((size_t)na * n * sizeof(cd));
  cd* dWx = (cd*)malloc((size_t)na * n * sizeof(cd));
  if (!xh || !twf || !twi || !Wx || !dWx) return 2;
  for (long i = 0; i < n; ++i) xh[n1 + i].re = x[i];
  for (long i = 0; i < n1; ++i) { long m = n1 - i; if (m < n) xh[i].re = x[m]; }                   /* reflect */
  for (long i = 0; i < P - n1 - n; ++i) { long m = n - 2 - i; if (m >= 0) xh[n1 + n + i].re = x[m]; }
  make_tw(twf, P, -1);
  make_tw(twi, P, +1);
  fft_inplace(xh, P, twf, NULL);
  const double h = 2.0 * M_PI / (double)P, norm = 1.0 / (double)P;
  if (gamma < 0) gamma = 10.0 * 2.220446049250313e-16;
#pragma omp parallel
  {
    cd* a = (cd*)malloc((size_t)P * sizeof(cd));
    cd* b = (cd*)malloc((size_t)P * sizeof(cd));
#pragma omp for schedule(dynamic, 1)
    for (long s = 0; s < na; ++s) {
      for (long k = 0; k < P; ++k) {
        const double xi = (k <= P / 2) ? (double)k * h : (double)(k - P) * h;     /* base.rs:18-33 */
        const double w = scales[s] * xi;
        double v = 0.0;
        if (wavelet == 1) {
          if (w >= 0.0) v = pow(M_PI, -0.25) * M_SQRT2 * (exp(-0.5 * (w - 6.0) * (w - 6.0)) - exp(-18.0) * exp(-0.5 * w * w));
        } else if (w > 0.0) {
          v = 2.0 * exp(60.0 * log(w) - pow(w, 3.0));
        }
        a[k].re = xh[k].re * v;
        a[k].im = xh[k].im * v;
        b[k].re = -a[k].im * (xi / dt);                                            /* (i xi / dt) * a */
        b[k].im = a[k].re * (xi / dt);
      }
      fft_inplace(a, P, twi, NULL);
      fft_inplace(b, P, twi, NULL);
      for (long j = 0; j < n; ++j) {
        Wx[s * n + j].re = a[n1 + j].re * norm;
        Wx[s * n + j].im = a[n1 + j].im * norm;
        dWx[s * n + j].re = b[n1 + j].re * norm;
        dWx[s * n + j].im = b[n1 + j].im * norm;
      }
    }
    free(a);
    free(b);
  }
  /* ssq_cwt.rs:450-469 (maprange "peak", log) */
  const double fmin = 1.0 / scales[na - 1], fmax = 1.0 / scales[0];
  const double lmin = log2(fmin), lstep = (log2(fmax) - lmin) / (double)(na - 1);
  for (long i = 0; i < na; ++i) ssq_freqs[i] = pow(2.0, lmin + (double)i * lstep);
  const int is_log = ssq_freqs[1] / ssq_freqs[0] > 1.1;                            /* :135-139 */
  const double bmin = is_log ? log2(ssq_freqs[0]) : ssq_freqs[0];
  const double bstep = is_log ? (log2(ssq_freqs[na - 1]) - bmin) / (double)(na - 1)
                              : (ssq_freqs[na - 1] - ssq_freqs[0]) / (double)(na - 1);
  memset(Tx, 0, (size_t)na * n * 2 * sizeof(double));
#pragma omp parallel for schedule(static)
  for (long j = 0; j < n; ++j) {                                                   /* :160-213, one column per task */
    for (long i = 0; i < na; ++i) {
      const cd W = Wx[i * n + j], D = dWx[i * n + j];
      if (hypot(W.re, W.im) < gamma) continue;                                     /* :23-29 */
      const double w = fabs((D.im * W.re - D.re * W.im) / ((W.re * W.re + W.im * W.im) * 6.283185307179586));
      if (isinf(w) || isnan(w)) continue;
      const double v = ((is_log ? log2(w) : w) - bmin) / bstep;
      const double r = round(v);                                                   /* half away from zero */
      if (!(r >= 0.0) || r >= (double)na) continue;                                /* dropped (:177, :188) */
      const long bin = (long)r;
      const long k = flipud ? na - 1 - bin : bin;
      Tx[2 * (k * n + j)] += W.re;
      Tx[2 * (k * n + j) + 1] += W.im;
    }
  }
  free(xh); free(twf); free(twi); free(Wx); free(dWx);
  return 0;
}
