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
