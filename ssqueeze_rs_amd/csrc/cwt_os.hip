// cwt_os.hip -- ssq_cwt of the short-wavelet scales by time tiles (overlap-save), fp32.
//
// The reference computes every scale as a length-P circular convolution in the frequency domain (cwt.rs:228-310) and
// reassigns afterwards (ssq_cwt.rs:116-222); on the GPU that costs a step buffer and the Wx / dWx workspaces out to
// memory and back (csrc/cwt_reg.hip).  For a scale whose wavelet is short in TIME -- support below
// 2 * kOsHalo samples to fp32 accuracy, and psih(a*pi) negligible so that the Nyquist cut leaves no slow tail -- the same
// convolution restricted to a tile of kOsL = 4096 output samples needs only the kOsF = 8192 input samples around it:
//     Wx[n0 + i] = ifft_F( fft_F(x_pad[n0 - halo ...]) * psih(a * 2 pi k / F) )[halo + i]
// (the F-grid samples of psih are the spectrum of the same time-domain wavelet; what differs from the reference is the
// wavelet's tail beyond the halo, chosen below 2^-24 of its peak).  One block owns one time tile and walks ALL eligible
// scales in ascending order: the 8192-point transforms live in LDS + registers (8 x 1024 on the per-wave core, Wx on
// waves 0-7 and dWx on waves 8-15), the phase transform and the bin (cwt_bin.h) run on the tile in LDS, and every thread
// keeps the run state of its 4 time columns, so Wx and dWx of these scales never reach memory and Tx receives one
// read-modify-write per run (the block owns its columns: no atomics).
// Two geometries (template R = rows of 1024 points): R = 8 as above (16 waves, one block per CU), and R = 4 -- 4096-point
// transforms, 2048 output samples, a 1024-sample halo, 8 waves, TWO blocks per CU whose phases overlap -- for the scales
// whose wavelet fits the shorter halo.
#include <type_traits>
#include "cwt_bin.h"
#include "cwt_kernels.h"
#include "fft_wave1024.h"
#include "stft_kernels.h"   // load_padded

namespace ssq {

#ifndef SSQ_OS_WAVES_PER_SIMD
#define SSQ_OS_WAVES_PER_SIMD 4     // 4: 128 VGPRs (two 8-wave blocks per CU); 2: 256 VGPRs, no spills, one 8-wave block per CU
#endif

namespace {

template <int R>
struct OsCfg {
  static_assert(R == 4 || R == 8, "rows");
  static constexpr int F = 1024 * R;          // transform length
  static constexpr int L = F / 2;             // output samples per tile
  static constexpr int HALO = F / 4;
  static constexpr int THREADS = 128 * R;     // 2 R waves: R rows of Wx, R rows of dWx
  static constexpr int PT = 1024 + 32 / R;    // row pitch: the epilogue's (row = n % R, n / R) reads hit distinct banks
  static constexpr int LOGR = R == 8 ? 3 : 2;
};

// forward-sign unit root e^{-2 pi i r / (1024 R)} from W_1024 (global, cache resident) and the R low steps
template <int R>
__device__ __forceinline__ cpx<float> os_wF(const cpx<float>* __restrict__ tw1024, int r) {
  // W_8192^l, l = 0..7 (R = 4 uses the even entries: W_4096^l = W_8192^(2 l))
  constexpr float lo[8][2] = {{1.0f, 0.0f},
                              {0.99999970586288221916f, -0.00076699031874270453f},
                              {0.99999882345170187925f, -0.00153398018628476561f},
                              {0.99999735276697821091f, -0.00230096915142580450f},
                              {0.99999529380957617151f, -0.00306795676296597627f},
                              {0.99999264658070719110f, -0.00383494256970622610f},
                              {0.99998941108192840321f, -0.00460192612044857020f},
                              {0.99998558731514319867f, -0.00536890696399634140f}};
  constexpr int LOGR = R == 8 ? 3 : 2;
  const cpx<float> h = tw1024[(r >> LOGR) & 1023];
  const int l = (r & (R - 1)) * (8 / R);
  return cmul(h, cpx<float>{lo[l][0], lo[l][1]});
}

template <int R>
__device__ __forceinline__ void os_dft(cpx<float> (&v)[R]) {
  if constexpr (R == 8) dft8<false>(v);
  else dft4<false>(v[0], v[1], v[2], v[3]);
}

}  // namespace

// One block = one time tile of OsCfg<R>::L output samples; see the header.
// LOGM > 0 (long wavelets that are band-limited below 1 / (4 M) cycles per sample, M = 2^LOGM): the tile works on the
// M-fold DECIMATED time grid -- F decimated samples span F M original ones, so the halo is M times longer -- and a block
// computes ONE output phase r of its tile: with S = F M,
//   Wx[n0 + M m + r] = (1/S) sum_{k < F/2} (X_seg[k] psih(a 2 pi k / S) e^{2 pi i k r / S}) e^{2 pi i k m / F},
// where X_seg[k], k < F/2, are the low bins of the S-point spectrum of the segment (cwt_os_dec_fwd_kernel: one F-point
// transform per input phase, summed here).  The M phases of a tile run on the same XCD, so their 8-byte read-modify-
// writes of neighbouring Tx columns meet in one L2.
// LOGM < 0 (FULL circle, decimation 2^p.log_dec chosen at run time with S = F M = the padded length P, for 2 N <= P):
// the "segment" is the whole padded signal, so the circular convolution is the reference's own, exact for every scale
// whose spectrum ends below F / 2 bins -- the band-limited scales of mode Z -- and X_seg is simply the spectrum xh the
// forward transform of the call has left in natural order; ONE tile, P / F phases = blocks.
// CPLX (plain tiles only): the input is the ANALYTIC signal xa = ifft_P(X 1[k <= P/2]) of the padded signal (complex, one
// extra inverse transform per call) and the wavelet's spectrum is continued smoothly beyond the Nyquist frequency
// (psih(a w) for w up to 2 pi, all F bins live).  The reference zeroes the negative frequencies of X psih; with xa as
// the input that cut is already in the signal, the filter stays smooth -- hence short in time -- and the finest scales,
// whose psih is NOT negligible at Nyquist, become tile-able as well.
template <int R, int LOGM, bool CPLX = false>
__global__ __launch_bounds__(OsCfg<R>::THREADS, SSQ_OS_WAVES_PER_SIMD) void cwt_os_kernel(CwtOsDev p) {
  using K = OsCfg<R>;
  constexpr int F = K::F, L = K::L, HALO = K::HALO, THREADS = K::THREADS, PT = K::PT, LOGR = K::LOGR;
  constexpr int H2 = CPLX ? R : R / 2;                         // live rows of the spectrum
  constexpr int NS = CPLX ? F : F / 2;                         // spectrum bins kept per tile / wavelet row
  static_assert(!CPLX || LOGM == 0, "analytic input: plain tiles");
  constexpr bool FULL = LOGM < 0;
  const int logm = FULL ? p.log_dec : LOGM;                    // (a compile-time constant unless FULL)
  const int M = 1 << logm;
  __shared__ __attribute__((aligned(16))) cpx<float> zb[2][R * PT];         // kind 0 | kind 1: [row j][column]
  __shared__ __attribute__((aligned(16))) cpx<float> tws[kWave1024TwElems];
  const int tid = threadIdx.x;
  const int t = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  cpx<float>* tw1 = tws;
  cpx<float>* tw2 = tws + 256;
  wave1024_tables(tw1, tw2, p.tw1024, tid, THREADS);
  int tile = blockIdx.x, phase = 0;
  if constexpr (FULL) {
    const int bid = blockIdx.x;                                // gridDim.x = M, a multiple of 8: XCD x takes M / 8
    tile = 0;                                                  // neighbouring phases = neighbouring Tx columns
    phase = (bid & 7) * (M >> 3) + (bid >> 3);
  } else if constexpr (LOGM > 0) {
    const int bid = blockIdx.x;
    if (((gridDim.x >> LOGM) & 7) == 0) {                      // blocks x, x + 8, ... share XCD x: a tile's phases together
      const int j = bid >> 3;
      tile = (bid & 7) + 8 * (j >> LOGM);
      phase = j & (M - 1);
    } else {
      tile = bid >> LOGM;
      phase = bid & (M - 1);
    }
  }
  // first output sample of the tile in unpadded time; FULL: the window [P/4, 3P/4) of the padded signal starts at
  // unpadded time P/4 - n1 <= 0 (2 N <= P), columns outside [0, N) are dropped
  const long long n0 = FULL ? p.full_n0 : (long long)tile * ((long long)L << logm);
  cpx<float>* __restrict__ xs = p.xs + (long long)tile * NS * M;        // (unused when FULL)
  __syncthreads();

  // W_F^(c j), j < R, for the columns c = tid + THREADS * u this thread transforms (forward sign): 1, w, w^2 ...
  constexpr int CPT = 1024 / THREADS;                          // columns per thread in the length-R transforms: 1 or 2
#ifndef SSQ_OS_WJ_KEEP
#define SSQ_OS_WJ_KEEP 0     // 1: the R powers live in registers across all scales (2 R CPT of the 128 the kernel may use)
#endif
  cpx<float> wj[CPT][R];
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    wj[u][0] = {1.0f, 0.0f};
    wj[u][1] = os_wF<R>(p.tw1024, tid + THREADS * u);
#pragma unroll
    for (int j = 2; j < R; ++j) wj[u][j] = cmul(wj[u][j - 1], wj[u][1]);
  }

  // ---- forward transform of the tile's F input samples: n = 1024 r + c, k = j + R m ----
  if constexpr (LOGM == 0) {
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const int c = tid + THREADS * u;
    cpx<float> v[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if constexpr (CPLX) v[r] = p.xa[p.xa_off + n0 - HALO + 1024 * r + c];
      else v[r] = {load_padded(p.x, n0 - HALO + 1024 * r + c, p.n_signal, p.padtype), 0.0f};
    }
    os_dft<R>(v);
#pragma unroll
    for (int j = 0; j < R; ++j) zb[0][j * PT + c] = cmul(v[j], wj[u][j]);
  }
  __syncthreads();
  if (wv < R) {
    cpx<float>* row = zb[0] + wv * PT;
    cpx<float> v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = row[t + 64 * q];
    frame_sync<false>();
    wave1024_front(v, row, t);
    wave1024_back(v, tw1, tw2, t);
    // X[k = wv + R m], m = t + 64 q; only k < NS is used: m < NS / R
#pragma unroll
    for (int q = 0; q < (CPLX ? 16 : 8); ++q) xs[R * (t + 64 * q) + wv] = v[q];
  }
  __threadfence_block();
  __syncthreads();                                             // xs of this tile is visible to the whole block (same CU)
  }

  // ---- all eligible scales, ascending ----
  const int kind_w = wv >= R ? 1 : 0;                          // waves 0 .. R-1: Wx rows, R .. 2R-1: dWx rows
  cpx<float>* myrow = zb[kind_w] + (wv & (R - 1)) * PT;
  CwtSsqDev<float> q = p.q;
  cpx<float>* __restrict__ Tx = q.Tx;
  // (no-return float atomics instead of the read-modify-write of a finished run: 0.75 ms SLOWER on C4,
  // profiles/r02_ab_cwt_os.txt -- 1.3e8 L2 atomics cost more than the loads they replace)
  // this thread's spectrum values stay in registers for all scales; the wavelet row of the NEXT scale is requested
  // behind the transform of the current one
  cpx<float> xk[CPT][H2];
  float hk[CPT][H2];
#pragma unroll
  for (int u = 0; u < CPT; ++u)
#pragma unroll
    for (int r = 0; r < H2; ++r) {
      const int k = 1024 * r + tid + THREADS * u;
      if constexpr (LOGM == 0) {
        xk[u][r] = xs[k];
      } else {
        cpx<float> sum = {0.0f, 0.0f};
        if constexpr (FULL) {
          sum = p.xh[k];                                         // the padded signal's own spectrum
        } else {
          for (int rp = 0; rp < M; ++rp) {                       // the S-point spectrum's bin k from the M input phases
            const cpx<float> v = xs[(long long)rp * NS + k];
            sum.x += v.x;
            sum.y += v.y;
          }
        }
        double sn, cs;                                         // e^{+2 pi i k phase / S}, once per block
        sincospi(2.0 * (double)k * (double)phase / (double)((long long)F << logm), &sn, &cs);
        xk[u][r] = cmul(sum, cpx<float>{(float)cs, (float)sn});
      }
      hk[u][r] = p.H[k];
    }
  int k_cur[4] = {-1, -1, -1, -1};                              // (declared behind the spectrum set-up: its double-precision
  cpx<float> acc[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};   //  sincospi is the kernel's register peak)
  auto flush = [&](int i, long long col) {
    cpx<float>* d = Tx + (long long)k_cur[i] * q.N + col;
    const cpx<float> tv = *d;
    *d = {tv.x + acc[i].x, tv.y + acc[i].y};
  };
#pragma unroll 1
  for (int s = p.s_begin; s < p.s_end; ++s) {
    // phase 1: Y[k] = X_b[k] H_s[k] (* i xi_k / dt), k = 1024 r + c, r < R / 2; on conjugated data (ifft = conj fft
    //          conj); length-R transform over r (R / 2 live inputs), twiddle W_F^(c j), rows j of both kinds
    const cpx<float>* xs_it = xs;
    if constexpr (CPLX) {
      int z;
      asm volatile("s_mov_b32 %0, 0" : "=s"(z));
      xs_it += z;
    }
#if !SSQ_OS_WJ_KEEP
    {
      // the twiddle powers are rebuilt per scale (one cached table read and R - 2 products per column) instead of
      // living in 2 R CPT registers through the wave transform and the epilogue, where the kernel spills: the opaque
      // zero offset keeps the (loop-invariant) rebuild inside the loop.  C4 3.44 -> 3.25 ms (profiles/r03_ab_os_wj.txt)
      int z;
      asm volatile("s_mov_b32 %0, 0" : "=s"(z));
      const cpx<float>* twp = p.tw1024 + z;
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        wj[u][0] = {1.0f, 0.0f};
        wj[u][1] = os_wF<R>(twp, tid + THREADS * u);
#pragma unroll
        for (int j = 2; j < R; ++j) wj[u][j] = cmul(wj[u][j - 1], wj[u][1]);
      }
    }
#endif
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const int c = tid + THREADS * u;
      cpx<float> a[R], b[R];
#pragma unroll
      for (int r = 0; r < H2; ++r) {
        const int k = 1024 * r + c;
        // (analytic-input tiles keep all R rows of the spectrum: re-read per scale from the tile's workspace -- cache
        //  resident -- instead of holding 2 R CPT more registers than the kernel has)
        const cpx<float> x = CPLX ? xs_it[k] : xk[u][r];
        const float h = hk[u][r];
        const cpx<float> y = {x.x * h, -x.y * h};                // conj(X H)
        a[r] = y;
        const float xi = (float)k * p.xi_step;                   // conj(Y * i xi) = conj(Y) * (-i) * xi
        b[r] = {y.y * xi, -y.x * xi};
      }
#pragma unroll
      for (int r = H2; r < R; ++r) {
        a[r] = {0.0f, 0.0f};
        b[r] = {0.0f, 0.0f};
      }
      os_dft<R>(a);
      os_dft<R>(b);
#pragma unroll
      for (int j = 0; j < R; ++j) {
        zb[0][j * PT + c] = cmul(a[j], wj[u][j]);
        zb[1][j * PT + c] = cmul(b[j], wj[u][j]);
      }
    }
    __syncthreads();
    // phase 2: the 1024-point transform of this wave's row of its kind; x[R m + j] back into the row
    {
      cpx<float> v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = myrow[t + 64 * i];
      frame_sync<false>();
      wave1024_front(v, myrow, t);
      if (s + 1 < p.s_end) {
        const float* __restrict__ Hn = p.H + (long long)(s + 1 - p.s_begin) * NS;
#pragma unroll
        for (int u = 0; u < CPT; ++u)
#pragma unroll
          for (int r = 0; r < H2; ++r) hk[u][r] = Hn[1024 * r + tid + THREADS * u];
      }
      wave1024_back(v, tw1, tw2, t);
      const float sc = p.out_mul ? p.inv_F * p.out_mul[s] : p.inv_F;   // (cwt with the L2 norm: times sqrt(scale))
#pragma unroll
      for (int i = 0; i < 16; ++i) myrow[t + 64 * i] = {v[i].x * sc, -v[i].y * sc};
    }
    __syncthreads();
    // phase 3: the tile's L valid samples: phase transform, bin, run merge; thread -> columns tid + THREADS i.
    // The runs that end at this scale are collected first and read-modify-written together (one memory round trip
    // for the four columns instead of one per column)
    cpx<float>* fl_ptr[4];
    cpx<float> fl_val[4];
    // The strided columns of the decimated / full-circle tiles are loop invariant, and hoisted out of the scale loop they
    // and the addresses built on them (Tx, the test hooks) hold ~30 registers -- which pushed the run state into scratch.
    // An opaque zero per scale keeps the (two-instruction) column arithmetic here.
    int ph = phase;
    if constexpr (LOGM != 0) {
      int z;
      asm volatile("s_mov_b32 %0, 0" : "=s"(z));
      ph += z;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int nl = HALO + tid + THREADS * i;                   // position inside the F-sample frame
      // (decimated / full-circle tiles: 32-bit column arithmetic -- tiled plans have N < 2^24 -- the kernel runs at its
      //  128-register cap and four 64-bit strided columns pushed the run state into scratch)
      using col_t = std::conditional_t<LOGM == 0, long long, int>;
      const col_t col = (col_t)n0 + ((col_t)(tid + THREADS * i) << logm) + (col_t)ph;
      const cpx<float> Wv = zb[0][(nl & (R - 1)) * PT + (nl >> LOGR)];
      const cpx<float> dW = zb[1][(nl & (R - 1)) * PT + (nl >> LOGR)];
      fl_ptr[i] = nullptr;
      fl_val[i] = {0.0f, 0.0f};
      if (col >= 0 && col < q.N) {
        if (p.dbg_Wx) p.dbg_Wx[(long long)s * q.N + col] = Wv;
        if (p.dbg_dWx) p.dbg_dWx[(long long)s * q.N + col] = dW;
        if (p.store_only) continue;                              // `cwt`: Wx / dWx are the result, no bins
        float w;
        const int kk = reassign_bin(q, Wv, dW, w);
        if (q.wk) q.wk[(long long)s * q.N + col] = {w, (float)kk};
        if (kk != k_cur[i]) {
          if (k_cur[i] >= 0) {
            fl_ptr[i] = Tx + (long long)k_cur[i] * q.N + col;
            fl_val[i] = acc[i];
          }
          k_cur[i] = kk;
          acc[i] = {0.0f, 0.0f};
        }
        if (kk >= 0) {
          if (q.squeezing == 1) {
            acc[i].x += q.leb_val;
          } else {
            acc[i].x += Wv.x;
            acc[i].y += Wv.y;
          }
        }
      }
    }
    {
      cpx<float> tv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) tv[i] = fl_ptr[i] ? *fl_ptr[i] : cpx<float>{0.0f, 0.0f};
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (fl_ptr[i]) *fl_ptr[i] = {tv[i].x + fl_val[i].x, tv[i].y + fl_val[i].y};
    }
    __syncthreads();       // (an LDS-only barrier here and behind phases 1 / 2 -- no vector-memory drain -- measured +-0: 3.00 ms)
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long long col = n0 + ((long long)(tid + THREADS * i) << logm) + phase;
    if (col >= 0 && col < q.N && k_cur[i] >= 0) flush(i, col);
  }
}

// Decimated tiles, forward part: block (tile, input phase r') transforms the F samples x_seg[M m + r'] and writes
// e^{-2 pi i k r' / S} * fft_F(...)[k], k < F/2, to xs[tile][r'][k]; the sum over r' is the S-point spectrum's bin k.
template <int R, int LOGM>
__global__ __launch_bounds__(OsCfg<R>::THREADS, 4) void cwt_os_dec_fwd_kernel(CwtOsDev p) {
  using K = OsCfg<R>;
  constexpr int F = K::F, L = K::L, HALO = K::HALO, THREADS = K::THREADS, PT = K::PT;
  constexpr int M = 1 << LOGM;
  static_assert(THREADS == 1024, "one column per thread");
  __shared__ __attribute__((aligned(16))) cpx<float> zb[R * PT];
  __shared__ __attribute__((aligned(16))) cpx<float> tws[kWave1024TwElems];
  const int tid = threadIdx.x;
  const int t = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  cpx<float>* tw1 = tws;
  cpx<float>* tw2 = tws + 256;
  wave1024_tables(tw1, tw2, p.tw1024, tid, THREADS);
  const int tile = blockIdx.x >> LOGM, rp = blockIdx.x & (M - 1);
  const long long seg0 = (long long)tile * ((long long)L << LOGM) - ((long long)HALO << LOGM);   // segment origin
  __syncthreads();
  {
    cpx<float> wj[R];
    wj[0] = {1.0f, 0.0f};
    wj[1] = os_wF<R>(p.tw1024, tid);
#pragma unroll
    for (int j = 2; j < R; ++j) wj[j] = cmul(wj[j - 1], wj[1]);
    cpx<float> v[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
      v[r] = {load_padded(p.x, seg0 + ((long long)(1024 * r + tid) << LOGM) + rp, p.n_signal, p.padtype), 0.0f};
    os_dft<R>(v);
#pragma unroll
    for (int j = 0; j < R; ++j) zb[j * PT + tid] = cmul(v[j], wj[j]);
  }
  __syncthreads();
  if (wv < R) {
    cpx<float>* row = zb + wv * PT;
    cpx<float> v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = row[t + 64 * q];
    frame_sync<false>();
    wave1024_front(v, row, t);
    wave1024_back(v, tw1, tw2, t);
    cpx<float>* __restrict__ dst = p.xs + ((long long)tile * M + rp) * (F / 2);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int k = R * (t + 64 * q) + wv;                       // < F / 2
      double sn, cs;
      sincospi(-2.0 * (double)k * (double)rp / (double)((long long)F << LOGM), &sn, &cs);
      dst[k] = cmul(v[q], cpx<float>{(float)cs, (float)sn});
    }
  }
}

// H[s - s_begin][k] = psih(scale_s * 2 pi k / F), k < F / 2 (fp64, rounded once), the formulas of wavelet_table_kernel
__global__ void cwt_os_table_kernel(float* __restrict__ H, const double* __restrict__ scales, int s_begin, int n_scales,
                                    int wavelet, int F, int log_dec, int entries) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int sl = blockIdx.y;
  if (sl >= n_scales || k >= entries) return;
  const double xi = (double)k * (2.0 * 3.14159265358979323846 / ((double)F * (double)(1 << log_dec)));   // grid of F M points
  const double w = scales[s_begin + sl] * xi;
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
  H[(long long)sl * entries + k] = (float)v;
}

hipError_t launch_cwt_os_table(float* H, const double* d_scales, int s_begin, int n_scales, int wavelet, int rows,
                               int log_dec, bool all_bins, hipStream_t stream) {
  if (n_scales <= 0) return hipSuccess;
  const int F = 1024 * rows;
  const int entries = all_bins ? F : F / 2;                    // all_bins: continued beyond Nyquist (analytic-input tiles)
  hipLaunchKernelGGL(cwt_os_table_kernel, dim3((unsigned)((entries + 255) / 256), (unsigned)n_scales), dim3(256), 0, stream, H,
                     d_scales, s_begin, n_scales, wavelet, F, log_dec, entries);
  return hipGetLastError();
}

hipError_t launch_cwt_os_analytic(const CwtOsDev& p, hipStream_t stream) {
  if (p.s_end <= p.s_begin) return hipSuccess;
  if (!p.xa) return hipErrorInvalidValue;
  const long long tiles = (p.q.N + OsCfg<4>::L - 1) / OsCfg<4>::L;
  hipLaunchKernelGGL((cwt_os_kernel<4, 0, true>), dim3((unsigned)tiles), dim3(OsCfg<4>::THREADS), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_cwt_os_full(const CwtOsDev& p, hipStream_t stream) {
  if (p.s_end <= p.s_begin) return hipSuccess;
  if (p.log_dec < 3 || !p.xh) return hipErrorInvalidValue;
  hipLaunchKernelGGL((cwt_os_kernel<4, -1>), dim3(1u << p.log_dec), dim3(OsCfg<4>::THREADS), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_cwt_os(const CwtOsDev& p, int rows, int log_dec, hipStream_t stream) {
  if (p.s_end <= p.s_begin) return hipSuccess;
  if (log_dec == kOsLogDec && rows == 8) {
    const long long span = (long long)OsCfg<8>::L << kOsLogDec;
    const long long blocks = ((p.q.N + span - 1) / span) << kOsLogDec;
    hipLaunchKernelGGL((cwt_os_dec_fwd_kernel<8, kOsLogDec>), dim3((unsigned)blocks), dim3(OsCfg<8>::THREADS), 0, stream, p);
    hipLaunchKernelGGL((cwt_os_kernel<8, kOsLogDec>), dim3((unsigned)blocks), dim3(OsCfg<8>::THREADS), 0, stream, p);
  } else if (log_dec != 0) {
    return hipErrorInvalidValue;
  } else if (rows == 8) {
    const long long tiles = (p.q.N + OsCfg<8>::L - 1) / OsCfg<8>::L;
    hipLaunchKernelGGL((cwt_os_kernel<8, 0>), dim3((unsigned)tiles), dim3(OsCfg<8>::THREADS), 0, stream, p);
  } else {
    const long long tiles = (p.q.N + OsCfg<4>::L - 1) / OsCfg<4>::L;
    hipLaunchKernelGGL((cwt_os_kernel<4, 0>), dim3((unsigned)tiles), dim3(OsCfg<4>::THREADS), 0, stream, p);
  }
  return hipGetLastError();
}

}  // namespace ssq
