// stft_generic.hip -- unfused STFT-family kernels for ANY n_fft (the reference takes any
// length rustfft can plan, i.e. any n: stft.rs:43-44, ssq_stft.rs:198-199).  Used when the
// fused LDS-tile kernel does not apply (n_fft not a power of two, < 64 or > 4096) and as an
// independent second GPU implementation in the parity tests.
//
//   dft_frames_kernel    : Sx/dSx[k, frame] by direct O(n_fft) sums per output bin with an
//                          exact-index twiddle table (double accumulation for both dtypes)
//                          (stft.rs:47-85, ssq_stft.rs:191-252)
//   reassign_cols_kernel : one thread per time column, rows ascending -- the reference's own
//                          accumulation order, no atomics (ssq_stft.rs:276-301)
#include "fft_generic.h"
#include "stft_kernels.h"

namespace ssq {

// ---- FFT path for long / odd n_fft: pack -> batched FFT of any length -> unpack ----
template <typename T>
__global__ void frames_pack_kernel(StftDev<T> p, long long sig, int n_fft, GenericTabs tabs, double alpha,
                                   cpx<T>* __restrict__ Z) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;     // sample inside the frame
  if (j >= n_fft) return;
  const T* xs = sig_base(p, sig);
  int m = j - p.rot;                                       // modulated frames (upstream variant): rotated transform input
  if (m < 0) m += n_fft;
  const double g = tabs.g[j], gd = tabs.gd[j] * alpha;
  for (int f = blockIdx.y; f < p.n_frames; f += gridDim.y) {   // (grid.y is capped at 65535: hop 1 on a long signal has more frames)
    const double xv = (double)load_padded(xs, (long long)f * p.hop - p.pad_left + j, p.n_signal, p.padtype);
    Z[(long long)f * n_fft + m] = {(T)(xv * g), (T)(xv * gd)};
  }
}

// Z[frame][k] -> Sx[k][frame] = (Z[k] + conj Z[n-k])/2, dSx[k][frame] = (Z[k] - conj Z[n-k])/(2i) / alpha, through a
// 32x32 LDS tile so that both the reads (along k) and the writes (along frame) are contiguous
template <typename T>
__global__ void frames_unpack_kernel(const cpx<T>* __restrict__ Z, int n_fft, int n_frames, int n_freqs, T inv_alpha,
                                     cpx<T>* __restrict__ Sx, cpx<T>* __restrict__ dSx) {
  __shared__ cpx<T> ts[32][33], td[32][33];
  const int k0 = blockIdx.x * 32, f0 = blockIdx.y * 32;
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int f = f0 + r, k = k0 + threadIdx.x;
    cpx<T> s = {(T)0, (T)0}, d = {(T)0, (T)0};
    if (f < n_frames && k < n_freqs) {
      const cpx<T> zk = Z[(long long)f * n_fft + k];
      const cpx<T> zn = Z[(long long)f * n_fft + (k == 0 ? 0 : n_fft - k)];
      s = {(zk.x + zn.x) * (T)0.5, (zk.y - zn.y) * (T)0.5};
      d = {(zk.y + zn.y) * (T)0.5 * inv_alpha, (zn.x - zk.x) * (T)0.5 * inv_alpha};
    }
    ts[r][threadIdx.x] = s;
    td[r][threadIdx.x] = d;
  }
  __syncthreads();
  for (int r = threadIdx.y; r < 32; r += blockDim.y) {
    const int k = k0 + r, f = f0 + threadIdx.x;
    if (k < n_freqs && f < n_frames) {
      Sx[(long long)k * n_frames + f] = ts[threadIdx.x][r];
      if (dSx) dSx[(long long)k * n_frames + f] = td[threadIdx.x][r];
    }
  }
}

template <typename T>
hipError_t launch_fft_frames(const StftDev<T>& p, long long sig, int n_fft, const GenericTabs& tabs, double alpha,
                             cpx<T>* Z, cpx<T>* work, cpx<T>* Sx, cpx<T>* dSx, hipStream_t stream) {
  if (p.n_frames > 65535 * 32) return hipErrorInvalidValue;
  hipLaunchKernelGGL(frames_pack_kernel<T>, dim3((n_fft + 255) / 256, p.n_frames < 65535 ? p.n_frames : 65535), dim3(256), 0, stream, p,
                     sig, n_fft, tabs,
                     dSx ? alpha : 0.0, Z);
  hipError_t e = fft_any_batched<T>(Z, work, n_fft, p.n_frames, -1, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(frames_unpack_kernel<T>, dim3((p.n_freqs + 31) / 32, (p.n_frames + 31) / 32), dim3(32, 8), 0, stream, Z,
                     n_fft, p.n_frames, p.n_freqs, (T)(1.0 / alpha), Sx, dSx);
  return hipGetLastError();
}

template <typename T>
__global__ void dft_frames_kernel(StftDev<T> p, int n_fft, GenericTabs tabs, cpx<T>* __restrict__ Sx,
                                  cpx<T>* __restrict__ dSx) {
  const long long n_signal = p.n_signal;
  const int hop = p.hop, pad_left = p.pad_left, padtype = p.padtype, n_frames = p.n_frames, n_freqs = p.n_freqs;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;   // frame
  const int k = blockIdx.y * blockDim.y + threadIdx.y;   // bin
  const long long b = blockIdx.z;
  if (j >= n_frames || k >= n_freqs) return;
  const T* xs = sig_base(p, b);
  const long long pos0 = (long long)j * hop - pad_left;
  double sr = 0, si = 0, dr = 0, di = 0;
  // (m*k) mod n_fft for the transform input m = (n - rot) mod n_fft of sample n (rot = 0 outside the upstream variant)
  int idx = (int)(((long long)((n_fft - p.rot) % n_fft) * k) % n_fft);
  for (int n = 0; n < n_fft; ++n) {
    const double xv = (double)load_padded(xs, pos0 + n, n_signal, padtype);
    const double c = tabs.tw_re[idx], s = tabs.tw_im[idx];
    const double u = xv * tabs.g[n];
    sr += u * c;
    si += u * s;
    if (dSx) {
      const double v = xv * tabs.gd[n];
      dr += v * c;
      di += v * s;
    }
    idx += k;
    if (idx >= n_fft) idx -= n_fft;
  }
  const long long o = (b * n_freqs + k) * (long long)n_frames + j;
  Sx[o] = {(T)sr, (T)si};
  if (dSx) dSx[o] = {(T)dr, (T)di};
}

template <typename T>
__global__ void reassign_cols_kernel(StftDev<T> p, const cpx<T>* __restrict__ Sx,
                                     const cpx<T>* __restrict__ dSx) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  const long long b = blockIdx.y;
  if (j >= p.n_frames) return;
  const long long base = b * (long long)p.n_freqs * p.n_frames + j;
  for (int i = 0; i < p.n_freqs; ++i) {
    const long long o = base + (long long)i * p.n_frames;
    const cpx<T> S = Sx[o], dS = dSx[o];
    T w;
    int kk;
    const bool keep = (p.variant & 1) ? phase_bin_upstream<T>(p, i, S, dS, w, kk) : phase_bin<T>(p, i, S, dS, w, kk);
    if (p.out_kind == 3) {
      p.out[o] = {w, keep ? (T)kk : (T)-1};
    } else if (keep) {
      const long long d = base + (long long)kk * p.n_frames;
      cpx<T> acc = p.out[d];
      if (p.squeezing == 1) {
        acc.x += p.leb_val;
      } else {
        acc.x += S.x * p.dw;
        acc.y += S.y * p.dw;
      }
      p.out[d] = acc;
    }
  }
}

template <typename T>
hipError_t launch_dft_frames(const StftDev<T>& p, long long batch, int n_fft, const GenericTabs& tabs,
                             cpx<T>* Sx, cpx<T>* dSx, hipStream_t stream) {
  dim3 block(64, 4, 1);
  dim3 grid((p.n_frames + 63) / 64, (p.n_freqs + 3) / 4, (unsigned)batch);
  hipLaunchKernelGGL(dft_frames_kernel<T>, grid, block, 0, stream, p, n_fft, tabs, Sx, dSx);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_reassign_cols(const StftDev<T>& p, const cpx<T>* Sx, const cpx<T>* dSx,
                                long long batch, hipStream_t stream) {
  dim3 block(64, 1, 1);
  dim3 grid((p.n_frames + 63) / 64, (unsigned)batch, 1);
  hipLaunchKernelGGL(reassign_cols_kernel<T>, grid, block, 0, stream, p, Sx, dSx);
  return hipGetLastError();
}

template hipError_t launch_dft_frames<float>(const StftDev<float>&, long long, int, const GenericTabs&, cpx<float>*,
                                             cpx<float>*, hipStream_t);
template hipError_t launch_dft_frames<double>(const StftDev<double>&, long long, int, const GenericTabs&, cpx<double>*,
                                              cpx<double>*, hipStream_t);
template hipError_t launch_fft_frames<float>(const StftDev<float>&, long long, int, const GenericTabs&, double, cpx<float>*,
                                             cpx<float>*, cpx<float>*, cpx<float>*, hipStream_t);
template hipError_t launch_fft_frames<double>(const StftDev<double>&, long long, int, const GenericTabs&, double,
                                              cpx<double>*, cpx<double>*, cpx<double>*, cpx<double>*, hipStream_t);
template hipError_t launch_reassign_cols<float>(const StftDev<float>&, const cpx<float>*, const cpx<float>*,
                                                long long, hipStream_t);
template hipError_t launch_reassign_cols<double>(const StftDev<double>&, const cpx<double>*,
                                                 const cpx<double>*, long long, hipStream_t);

}  // namespace ssq
