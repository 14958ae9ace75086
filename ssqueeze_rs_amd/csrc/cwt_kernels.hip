// cwt_kernels.hip -- CWT-family kernels for gfx950 (MI355X).
//
// Frequency-domain CWT as the reference does it (rust/src/spectral/cwt.rs:85-326,
// ssq_cwt.rs:329-435): FFT of the padded signal once, then per scale a wavelet multiply and an
// inverse FFT of the full padded length P (a power of two, utils/array.rs:9-11).  On the GPU the
// length-P transforms are two-step ("four-step") FFTs, P = P1*P2 with P1,P2 <= 4096:
//   step A: C adjacent columns of the [P1][P2] view per block, length-P1 FFTs in LDS, times W_P^(c*k1)
//   step B: C adjacent rows per block, length-P2 FFTs in LDS, output transposed in LDS so that global
//           stores are C-element contiguous segments (natural order, unpadded: cwt.rs:108-129)
// Band-limited scales skip step A altogether (mode Z): when psih_s[k] == 0 for all k >= Q (Q <= 2048 a power
// of two, D = P/Q), x[D m + d] = sum_{k<Q} (X[k] e^{+2 pi i k d/P}) e^{+2 pi i k m/Q} is one length-Q iFFT per
// residue d; a block takes C adjacent residues, so it stores the same C-element segments as step B and the
// only HBM traffic of the scale is its output.
// The wavelet multiply (cwt.rs:238-240, :275-277) is fused into step A's load and the 1/P and
// sqrt(scale) normalisation (cwt.rs:251-262) into step B's store; xh (16.8 MB at P = 2^21) is
// re-read by every scale and stays resident in the 256 MB Infinity Cache.
// Inverse transforms run the forward FFT core on conjugated data (ifft(x) = conj(fft(conj(x)))).
#include "cwt_kernels.h"
#include "cwt_bin.h"
#include "fft_core.h"
#include "fft_generic.h"
#include "stft_kernels.h"   // load_padded

namespace ssq {

// Measured dead ends on C4 (interleaved A/B on one device, profiles/README.md): batches of 16 instead of 8 loads,
// XCD-contiguous tile numbering, 16-row -> 8-row tiles with two blocks per CU, 1024-thread blocks, an LDS-resident
// Tx tile for the reassignment, and a software-pipelined step B (next tile's ybuf loads in registers, pass twiddles
// in LDS, stores behind the prefetch commit), inverse step A at 2048 points as two 1024-point half transforms (radix-2
// split with one live input: 5 % slower): none beat this form by more than 5 %.  The tile FFT is bound by the
// LDS round trips of 8 waves (ablation: load 18, FFT 26, store 11, rest 11 us of a 62 us step-B launch).
// Threads per tile block: 8 waves (one block per CU: the tile takes most of the LDS).  16 waves with twiddles read
// from the table instead of registers measured 10 % slower on C4.
#ifndef SSQ_CWT_F64_HALF
#define SSQ_CWT_F64_HALF 1   // fp64 inverse step B on 2048 points: 4-wave blocks with 2-row tiles on HALF the LDS, so two blocks share a
                             // CU and their load / FFT / store phases overlap (C4 fp64 16.85 -> 15.85 ms; for 4096 points the tile
                             // would be ONE row = 16-byte segments, and step A loses its f1 table: C5 +10 %, profiles/r03_ab_c64half.txt)
#endif
template <typename T, int LOGM, int MODE>
constexpr bool tile_half() {
  return SSQ_CWT_F64_HALF && sizeof(T) == 8 && LOGM == 11 && MODE == 4 /* CWT_INV_B */;
}
template <typename T>
constexpr int tile_threads() {
  return 512;
}

constexpr int pow2_floor(int v) {
  int r = 1;
  while (2 * r <= v) r *= 2;
  return r;
}

// NARROW (inverse step B in fp32): 8 transforms per tile instead of 16, so that two blocks share a CU and one's load /
// store phases overlap the other's FFT (C4: -2 % against 16-row tiles; for step A and mode Z the 16-row tile wins).
template <typename T, int LOGM, bool NARROW = false, bool HALF_ = false>
struct TileCfg {
  static constexpr int M = 1 << LOGM;
  static constexpr int L = M / 16;                                    // lanes per transform
  static constexpr bool HALF = HALF_;
  static constexpr int THREADS = HALF ? 256 : 512;
  static constexpr int LDS_BUDGET = (HALF ? 80 : 160) * 1024;
  static constexpr int TPR = (L >= THREADS) ? 1 : THREADS / L;   // transforms per round
  static constexpr int ROWP = M + M / 16 + 1;                         // odd-ish pitch: bank spread
  static constexpr int ROW_BYTES = ROWP * (int)sizeof(cpx<T>);
#ifndef SSQ_CWT_CCAP32
#define SSQ_CWT_CCAP32 16
#endif
  static constexpr int CCAP = (sizeof(T) == 4) ? (NARROW ? 8 : SSQ_CWT_CCAP32) : 8;   // >= 128-B (64-B) global segments
  static constexpr int CFIT = pow2_floor(LDS_BUDGET / ROW_BYTES);
  static constexpr int CWANT = (TPR > CCAP) ? TPR : CCAP;
  static constexpr int C = (CWANT < CFIT) ? CWANT : CFIT;             // transforms per tile
  static constexpr int LDS_BYTES = C * ROW_BYTES;
  static constexpr bool MULTIWAVE = (L > 64);
  // W_P^(t0 k), k < M, of the tile's first column / residue t0, kept beside the tile when it fits: the W_P twiddle
  // of element (c, k) is then f1[k] * tw_f2[c, k] -- one LDS read and one COALESCED table load instead of two
  // 64-address gathers from the split W_P table (which made step A's store phase TA-bound)
  static constexpr bool F1 = (LDS_BYTES + M * (int)sizeof(cpx<T>) <= LDS_BUDGET);
  static constexpr int LDS_TOTAL = LDS_BYTES + (F1 ? M * (int)sizeof(cpx<T>) : 0);
  static_assert(LOGM >= 4 && LOGM <= 12, "tile FFT length");
  static_assert(C >= TPR && C % TPR == 0, "whole rounds");
};

template <typename T>
__device__ __forceinline__ cpx<T> conj_if(cpx<T> v, bool inv) {
  if (inv) v.y = -v.y;
  return v;
}

// element n of the spectrum fed to an inverse transform: xh[n] * psih_s[n] (* i*xi_n/dt).
// The loads are unconditional (clamped index) so that a batch of them can be in flight together.
template <typename T>
__device__ __forceinline__ cpx<T> load_spectrum(const CwtDev<T>& p, int tr, long long n) {
  const long long half = p.P >> 1;
  const long long nn = n > half ? half : n;
  const int s = p.scale0 + tr / p.n_kinds;
  const int kind = tr % p.n_kinds;
  const long long bs = p.band[s];                       // psih_s is exactly zero from here on: not stored
  T psi = p.psih[p.psi_off[s] + (nn < bs ? nn : bs - 1)];
  const cpx<T> xv = p.xh[nn];
  // analytic wavelets: w < 0 -> 0 (cwt.rs:512,:536).  As a FACTOR, not a select: with `psi = cond ? 0 : psi` the compiler
  // sinks the table load behind the condition -- a branch per element, and every element of a batch then waits for its
  // own loads (the step-A load phase ran its 16 elements per thread as 16 serial memory round trips)
  psi *= (n > half || nn >= bs) ? (T)0 : (T)1;
  cpx<T> v = {xv.x * psi, xv.y * psi};                  // cwt.rs:238-240
  if (kind == 1) {                                      // * Complex(0, xi/dt)  cwt.rs:205-208
    const T xi = (T)n * p.xi_step;
    v = {-v.y * xi, v.x * xi};
  }
  return v;
}

// where transform tr of this launch stores its time samples, and the factor it applies (cwt.rs:251-262)
template <typename T>
struct TimeDst {
  cpx<T>* row;
  T sc;
};
template <typename T>
__device__ __forceinline__ TimeDst<T> time_dst(const CwtDev<T>& p, int tr) {
  const int s = p.scale0 + tr / p.n_kinds;
  cpx<T>* dst = (tr % p.n_kinds) ? p.dWx : p.Wx;
  return {dst + (long long)s * p.cols, p.out_scale[s]};
}
template <typename T>
__device__ __forceinline__ void store_time(const CwtDev<T>& p, const TimeDst<T>& d, long long n, cpx<T> v) {
  v = {v.x * d.sc, v.y * d.sc};
  if (p.rpadded) {
    d.row[n] = v;
  } else if (n >= p.n1 && n < p.n1 + p.n_signal) {      // cwt.rs:115
    d.row[n - p.n1] = v;
  }
}

// W_P^r from the split table (forward sign), r < P
template <typename T>
__device__ __forceinline__ cpx<T> twiddle_P(const CwtDev<T>& p, long long r) {
  return cmul(p.tw_hi[r >> 12], p.tw_lo[r & 4095]);
}

template <typename T, int MODE>
constexpr bool tile_narrow() {
  return sizeof(T) == 4 && MODE == CWT_INV_B;
}

// Two blocks per CU (<= 128 VGPRs, <= 80 KB of LDS each) where the tile is small enough: inverse step B with narrow
// tiles and the single-pass scales up to Q = 512 (fp32); their load / FFT / store phases then overlap across blocks.
template <typename T, int LOGM, int MODE>
constexpr int tile_blocks_per_cu() {
  using K = TileCfg<T, LOGM, tile_narrow<T, MODE>(), tile_half<T, LOGM, MODE>()>;
  if (tile_half<T, LOGM, MODE>()) return 2;
  return (sizeof(T) == 4 && (MODE == CWT_INV_B || MODE == CWT_INV_Z) && K::LDS_TOTAL <= 80 * 1024 && K::C == K::TPR) ? 2 : 1;
}

// One tile = C transforms of length M in LDS.  MODE is a compile-time CwtMode: every phase is straight-line code
// over batches of U elements per thread, so U global loads (or stores) are in flight per thread instead of one.
template <typename T, int LOGM, int MODE>
__global__ __launch_bounds__((TileCfg<T, LOGM, tile_narrow<T, MODE>(), tile_half<T, LOGM, MODE>()>::THREADS), (tile_blocks_per_cu<T, LOGM, MODE>()))
void cwt_tile_kernel(CwtDev<T> p) {
  using K = TileCfg<T, LOGM, tile_narrow<T, MODE>(), tile_half<T, LOGM, MODE>()>;
  constexpr int M = K::M, L = K::L, C = K::C, ROWP = K::ROWP;
  constexpr int kTileThreads = K::THREADS;
  // twiddles in registers pay only when a thread runs several rounds with them
  constexpr bool TW_REGS = (sizeof(T) == 4) && (C / K::TPR > 1);
  constexpr bool inv = MODE >= CWT_INV_A;
  constexpr bool stepA = (MODE == CWT_FWD_A || MODE == CWT_INV_A);
  constexpr bool stepB = (MODE == CWT_FWD_B || MODE == CWT_INV_B);
  constexpr bool stepZ = (MODE == CWT_INV_Z);
  static_assert((C * M) % kTileThreads == 0, "whole sweeps");
  constexpr int PER = C * M / kTileThreads;             // elements per thread and phase
#ifndef SSQ_CWT_U
#define SSQ_CWT_U 8
#endif
#ifndef SSQ_CWT_U64
#define SSQ_CWT_U64 8
#endif
  constexpr int UCAP = sizeof(T) == 8 ? SSQ_CWT_U64 : SSQ_CWT_U;
  constexpr int U = PER < UCAP ? PER : UCAP;            // batch
  static_assert(PER % U == 0, "whole batches");
  constexpr bool USE_F1 = K::F1 && (MODE == CWT_FWD_A || MODE == CWT_INV_A || MODE == CWT_INV_Z);
  __shared__ __attribute__((aligned(16))) unsigned char smem[USE_F1 ? K::LDS_TOTAL : K::LDS_BYTES];
  cpx<T>* rows = reinterpret_cast<cpx<T>*>(smem);
  cpx<T>* f1 = reinterpret_cast<cpx<T>*>(smem + K::LDS_BYTES);

  const int tid = threadIdx.x;
  const int ty = blockIdx.y;                 // launch-local transform: indexes the step buffer
  const int tr = p.tr0 + ty;                 // logical transform: scale and kind
  // blocks go round-robin over the 8 XCDs: give each XCD a contiguous range of tiles, so that the neighbouring tiles
  // whose SUB-LINE segments share 128-byte lines meet in one L2 and leave it as whole lines.  fp64 only: its tiles hold
  // 2 - 8 columns = 32 - 128-byte segments (C5: 106 -> 92 ms, with the shorter step A 86 ms); fp32 tiles are whole lines
  // already and measured no gain (round 1)
  const long long tile = (sizeof(T) == 8 && gridDim.x % 8 == 0)
                             ? (long long)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8
                             : (long long)blockIdx.x;
  const long long P2 = 1LL << p.log_p2;
  const long long P1 = 1LL << p.log_p1;
  const long long t0 = tile * C;                        // first column (A) / row (B) / residue (Z) / transform (S)

  if constexpr (USE_F1) {
    for (int k = tid; k < M; k += kTileThreads) f1[k] = twiddle_P(p, t0 * k);   // t0 k < P
    if constexpr (stepZ) __syncthreads();                 // Z uses it in the load phase, A only in the store phase
  }

  // ---------------- load (conjugated for inverse transforms) ----------------
  // element e of the tile -> its LDS slot; A walks columns fastest (C-element global segments), the others walk
  // the transform index fastest
  auto slot_of = [&](int e) {
    if constexpr (stepA) return (e % C) * ROWP + exch_phys(e / C);
    else return (e / M) * ROWP + exch_phys(e % M);
  };
  auto fetch = [&](int e) -> cpx<T> {
    if constexpr (MODE == CWT_FWD_A) {
      const long long n = (long long)(e / C) * P2 + t0 + (e % C);
      return {load_padded_flat(p.x, n - p.n1, p.n_signal, p.padtype, true), (T)0};   // (no branch: the batch stays in flight)
    } else if constexpr (MODE == CWT_INV_A) {
      const long long n = (long long)(e / C) * P2 + t0 + (e % C);
      return conj_if(load_spectrum(p, tr, n), true);
    } else if constexpr (stepB) {
      // step A already left it conjugated
      return p.ybuf[(long long)ty * p.P + (t0 + e / M) * P2 + (e % M)];
    } else if constexpr (stepZ) {
      const int k = e % M;
      long long d = t0 + e / M;
      const bool live = d < P1;                          // D < C: partial tile
      if (!live) d = 0;
      // W_P^(k d) = conj(e^{+2 pi i k d/P}), k d < Q D = P
      cpx<T> w;
      if constexpr (USE_F1) w = cmul(f1[k], p.tw_f2[(e / M) * M + k]);
      else w = twiddle_P(p, (long long)k * d);
      cpx<T> v = cmul(conj_if(load_spectrum(p, tr, k), true), w);
      if (!live) v = {(T)0, (T)0};
      return v;
    } else {
      const int m = e % M;
      const long long trc = t0 + e / M;
      if (trc >= p.n_transforms) return {(T)0, (T)0};
      if constexpr (MODE == CWT_FWD_S) return {load_padded_flat(p.x, (long long)m - p.n1, p.n_signal, p.padtype, true), (T)0};
      else return conj_if(load_spectrum(p, (int)trc, m), true);
    }
  };
  // step A of an inverse transform: rows at and above ceil(band / P2) of the [P1][P2] view hold only zeros
  long long live_elems = (long long)C * M;
  if constexpr (MODE == CWT_INV_A) {
    const long long band = p.band[p.scale0 + tr / p.n_kinds];
    live_elems = ((band + P2 - 1) / P2) * C;              // e = r*C + c < live_elems  <=>  r < live rows
  }
#pragma unroll 1
  for (int i0 = 0; i0 < PER; i0 += U) {
    cpx<T> buf[U];
#ifndef SSQ_CWT_ABL
#define SSQ_CWT_ABL 0
#endif
    if (!(SSQ_CWT_ABL & 1) && (long long)i0 * kTileThreads < live_elems) {   // uniform: whole batches of dead rows issue no loads
#pragma unroll
      for (int u = 0; u < U; ++u) buf[u] = fetch(tid + (i0 + u) * kTileThreads);
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u) buf[u] = {(T)0, (T)0};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) rows[slot_of(tid + (i0 + u) * kTileThreads)] = buf[u];
  }
  __syncthreads();

  // ---------------- length-M forward FFTs in LDS ----------------
  {
    const int slot = (L >= 64) ? __builtin_amdgcn_readfirstlane(tid / L) : tid / L;   // (a wave-uniform row index stays scalar)
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
    for (int round = 0; round < ((SSQ_CWT_ABL & 2) ? 0 : C / K::TPR); ++round) {
      cpx<T>* row = rows + (round * K::TPR + slot) * ROWP;
      cpx<T> v[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = row[exch_phys(t + L * q)];
      frame_sync<K::MULTIWAVE>();
      if (sizeof(T) == 8 && p.tw_compact) fft_pass_compact<T, LOGM, 0, K::MULTIWAVE>(v, row, p.tw_m + M, t);
      else fft_pass<T, LOGM, 0, false, TW_REGS, K::MULTIWAVE>(v, row, twr, p.tw_m, t);
#pragma unroll
      for (int q = 0; q < 16; ++q) row[exch_phys(t + L * q)] = v[q];
    }
  }
  __syncthreads();

  // ---------------- store ----------------
  // A: element (column c fastest, k1): ybuf[k1][col] = y * W_P^(col k1).  The data is conj(true value) for inverse
  //    transforms and conj(y conj(W)) = conj(y) W, so the forward twiddle serves both directions.
  // B / Z: element (row / residue c fastest, k2): n = t0 + c + P1 k2, natural order (Z: n = d + D m)
  TimeDst<T> td = {nullptr, (T)0};
  if constexpr (MODE == CWT_INV_B || MODE == CWT_INV_Z) td = time_dst(p, tr);
#pragma unroll 1
  for (int i0 = 0; i0 < PER; i0 += U) {
    cpx<T> buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + (i0 + u) * kTileThreads;
      if constexpr (stepA) {
        const int c = e % C, k1 = e / C;
        const long long col = t0 + c;
        cpx<T> w;                                            // W_P^(col k1), col k1 < P1 P2 = P
        if constexpr (USE_F1) w = cmul(f1[k1], p.tw_f2[k1 * C + c]);
        else w = twiddle_P(p, col * k1);
        buf[u] = cmul(rows[c * ROWP + exch_phys(k1)], w);
      } else if constexpr (stepB || stepZ) {
        buf[u] = conj_if(rows[(e % C) * ROWP + exch_phys(e / C)], inv);
      } else {
        buf[u] = conj_if(rows[(e / M) * ROWP + exch_phys(e % M)], inv);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + (i0 + u) * kTileThreads;
      if ((SSQ_CWT_ABL & 4) && buf[u].x != (T)12345.678) continue;   // ablation: compute, do not write
      if constexpr (stepA) {
        const int c = e % C, k1 = e / C;
        p.ybuf[(long long)ty * p.P + (long long)k1 * P2 + t0 + c] = buf[u];
      } else if constexpr (stepB || stepZ) {
        const int c = e % C, k2 = e / C;
        if (stepZ && t0 + c >= P1) continue;
        const long long n = t0 + c + P1 * k2;
        if constexpr (MODE == CWT_FWD_B) p.xh[n] = buf[u];
        else store_time(p, td, n, buf[u]);
      } else {
        const int m = e % M;
        const long long trc = t0 + e / M;
        if (trc >= p.n_transforms) continue;
        if constexpr (MODE == CWT_FWD_S) p.xh[m] = buf[u];
        else store_time(p, time_dst(p, (int)trc), m, buf[u]);
      }
    }
  }
}

template <typename T, int LOGM, int MODE>
static hipError_t launch_tile_mode(const CwtDev<T>& p, hipStream_t stream) {
  using K = TileCfg<T, LOGM, tile_narrow<T, MODE>(), tile_half<T, LOGM, MODE>()>;
  dim3 grid;
  if (MODE == CWT_FWD_A || MODE == CWT_INV_A) {
    grid = dim3((unsigned)(((1LL << p.log_p2) + K::C - 1) / K::C), (unsigned)p.n_transforms, 1);
  } else if (MODE == CWT_FWD_B || MODE == CWT_INV_B || MODE == CWT_INV_Z) {
    grid = dim3((unsigned)(((1LL << p.log_p1) + K::C - 1) / K::C), (unsigned)p.n_transforms, 1);
  } else {
    grid = dim3((unsigned)((p.n_transforms + K::C - 1) / K::C), 1, 1);
  }
  hipLaunchKernelGGL((cwt_tile_kernel<T, LOGM, MODE>), grid, dim3(K::THREADS), 0, stream, p);
  return hipGetLastError();
}

template <typename T, int LOGM>
static hipError_t launch_tile_one(int mode, const CwtDev<T>& p, hipStream_t stream) {
  switch (mode) {
    case CWT_FWD_A: return launch_tile_mode<T, LOGM, CWT_FWD_A>(p, stream);
    case CWT_FWD_B: return launch_tile_mode<T, LOGM, CWT_FWD_B>(p, stream);
    case CWT_FWD_S: return launch_tile_mode<T, LOGM, CWT_FWD_S>(p, stream);
    case CWT_INV_A: return launch_tile_mode<T, LOGM, CWT_INV_A>(p, stream);
    case CWT_INV_B: return launch_tile_mode<T, LOGM, CWT_INV_B>(p, stream);
    case CWT_INV_S: return launch_tile_mode<T, LOGM, CWT_INV_S>(p, stream);
    case CWT_INV_Z: return launch_tile_mode<T, LOGM, CWT_INV_Z>(p, stream);
  }
  return hipErrorInvalidValue;
}

template <typename T>
hipError_t launch_cwt_tile(int mode, const CwtDev<T>& p, hipStream_t stream) {
  int logm;
  if (mode == CWT_FWD_A || mode == CWT_INV_A) logm = p.log_p1;
  else if (mode == CWT_FWD_B || mode == CWT_INV_B || mode == CWT_INV_Z) logm = p.log_p2;
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
__global__ void wavelet_table_kernel(T* __restrict__ psih, const long long* __restrict__ off, const int* __restrict__ band,
                                     const double* __restrict__ scales, int na, long long P, int wavelet, double p0,
                                     double p1) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int s = blockIdx.y;
  if (s >= na || k >= band[s]) return;
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
  } else if (wavelet == 2) {                            // upstream GMW, L1 norm: old/ssqueezepy/_gmw.py:204-210
    if (w > 0.0) {
      const double wc = exp((1.0 / p0) * (log(p1) - log(p0)));       // morsefreq, _gmw.py:611-657  (p0 = gamma, p1 = beta)
      v = 2.0 * exp(-p1 * log(wc) + pow(wc, p0) + p1 * log(w) - pow(w, p0));
    }
  } else if (wavelet == 3) {                            // upstream Morlet: old/ssqueezepy/wavelets.py:497-523 (p0 = mu)
    const double cs = pow(1.0 + exp(-p0 * p0) - 2.0 * exp(-0.75 * p0 * p0), -0.5);
    const double ks = exp(-0.5 * p0 * p0);
    v = 1.41421356237309504880 * cs * pow(3.14159265358979323846, 0.25) *
        (exp(-0.5 * (w - p0) * (w - p0)) - ks * exp(-0.5 * w * w));
  } else {                                              // "gmw" | _  cwt.rs:522-542
    if (w > 0.0) v = 2.0 * exp(60.0 * log(w) - pow(w, 3.0));
  }
  if (wavelet >= 2 && 2 * k == P) v *= 0.5;             // upstream halves the Nyquist bin (wavelets.py:87-95)
  psih[off[s] + k] = (T)v;
}

template <typename T>
hipError_t launch_wavelet_table(T* psih, const long long* d_off, const int* d_band, int max_band, const double* d_scales,
                                int na, long long P, int wavelet, hipStream_t stream, double p0, double p1) {
  dim3 grid((unsigned)((max_band + 255) / 256), (unsigned)na, 1);
  hipLaunchKernelGGL(wavelet_table_kernel<T>, grid, dim3(256), 0, stream, psih, d_off, d_band, d_scales, na, P, wavelet, p0,
                     p1);
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
  store_time(p, time_dst(p, tr), n, cpx<T>{(T)sr, (T)si});
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

// ------------------------------------------------------ P > 2^24: generic FFT path ----
template <typename T>
__global__ void cwt_big_pad_kernel(CwtDev<T> p) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (n < p.P) p.xh[n] = {load_padded(p.x, n - p.n1, p.n_signal, p.padtype), (T)0};
}
template <typename T>
__global__ void cwt_big_spectrum_kernel(CwtDev<T> p) {
  const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int tr = blockIdx.y;
  if (k < p.P) p.ybuf[(long long)tr * p.P + k] = load_spectrum(p, tr, k);
}
template <typename T>
__global__ void cwt_big_store_kernel(CwtDev<T> p) {
  const long long n = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int tr = blockIdx.y;
  if (n < p.P) store_time(p, time_dst(p, tr), n, p.ybuf[(long long)tr * p.P + n]);
}
template <typename T>
hipError_t launch_cwt_big_fwd(const CwtDev<T>& p, cpx<T>* work, hipStream_t stream) {
  hipLaunchKernelGGL(cwt_big_pad_kernel<T>, dim3((unsigned)((p.P + 255) / 256)), dim3(256), 0, stream, p);
  return fft_any_batched<T>(p.xh, work, p.P, 1, -1, stream);
}
template <typename T>
hipError_t launch_cwt_big_inv(const CwtDev<T>& p, cpx<T>* work, hipStream_t stream) {
  const dim3 grid((unsigned)((p.P + 255) / 256), (unsigned)p.n_transforms);
  hipLaunchKernelGGL(cwt_big_spectrum_kernel<T>, grid, dim3(256), 0, stream, p);
  const hipError_t e = fft_any_batched<T>(p.ybuf, work, p.P, p.n_transforms, +1, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(cwt_big_store_kernel<T>, grid, dim3(256), 0, stream, p);
  return hipGetLastError();
}

#ifndef SSQ_CWT_WIDE_STORE
#define SSQ_CWT_WIDE_STORE 0
#endif
// ---------------------------------------------------------------- fused ssq step B / mode Z ----
// See cwt_kernels.h.  The LDS holds the same 2*Cf rows as the unfused tile (Cf rows of the Wx transform, then Cf rows
// of the dWx transform of the SAME scale); loads, FFT rounds and the transposed read-back are those of
// cwt_tile_kernel, the store phase pairs element (c, k2) of the two halves.
template <typename T, int LOGM, int MODE>
__global__ __launch_bounds__((tile_threads<T>())) void cwt_tile_ssq_kernel(CwtDev<T> p, CwtSsqDev<T> q) {
  using K = TileCfg<T, LOGM, false>;
  constexpr int M = K::M, L = K::L, C2 = K::C, CF = K::C / 2, ROWP = K::ROWP;
  constexpr int kTileThreads = K::THREADS;
  static_assert(MODE == CWT_INV_B || MODE == CWT_INV_Z, "fused modes");
  static_assert(K::C >= 2 && K::C % 2 == 0, "two halves");
  constexpr bool TW_REGS = (sizeof(T) == 4) && (C2 / K::TPR > 1);
  constexpr bool stepZ = (MODE == CWT_INV_Z);
  constexpr int PER = C2 * M / kTileThreads;
  constexpr int U = PER < SSQ_CWT_U ? PER : SSQ_CWT_U;
  static_assert(PER % U == 0, "whole batches");
  constexpr bool USE_F1 = K::F1 && stepZ;
  __shared__ __attribute__((aligned(16))) unsigned char smem[USE_F1 ? K::LDS_TOTAL : K::LDS_BYTES];
  cpx<T>* rows = reinterpret_cast<cpx<T>*>(smem);
  cpx<T>* f1 = reinterpret_cast<cpx<T>*>(smem + K::LDS_BYTES);

  const int tid = threadIdx.x;
  const int s_local = blockIdx.y;                        // transforms 2*s_local (Wx) and 2*s_local + 1 (dWx)
  const long long tile = blockIdx.x;
  const long long P2 = 1LL << p.log_p2;
  const long long P1 = 1LL << p.log_p1;
  const long long t0 = tile * CF;                        // first row (B) / residue (Z)

  if constexpr (USE_F1) {
    for (int k = tid; k < M; k += kTileThreads) f1[k] = twiddle_P(p, t0 * k);
    __syncthreads();
  }
  auto fetch = [&](int e) -> cpx<T> {
    const int r = e / M, m = e % M;
    const int kind = r / CF, c = r % CF;
    const int tr = 2 * s_local + kind;
    if constexpr (!stepZ) {
      if (t0 + c >= P1) return {(T)0, (T)0};
      return p.ybuf[(long long)tr * p.P + (t0 + c) * P2 + m];
    } else {
      long long d = t0 + c;
      const bool live = d < P1;
      if (!live) d = 0;
      cpx<T> w;
      if constexpr (USE_F1) w = cmul(f1[m], p.tw_f2[c * M + m]);
      else w = twiddle_P(p, (long long)m * d);
      cpx<T> v = cmul(conj_if(load_spectrum(p, tr, m), true), w);
      if (!live) v = {(T)0, (T)0};
      return v;
    }
  };
#pragma unroll 1
  for (int i0 = 0; i0 < PER; i0 += U) {
    cpx<T> buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) buf[u] = fetch(tid + (i0 + u) * kTileThreads);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int e = tid + (i0 + u) * kTileThreads;
      rows[(e / M) * ROWP + exch_phys(e % M)] = buf[u];
    }
  }
  __syncthreads();
  {
    const int slot = (L >= 64) ? __builtin_amdgcn_readfirstlane(tid / L) : tid / L;   // (a wave-uniform row index stays scalar)
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
    for (int round = 0; round < C2 / K::TPR; ++round) {
      cpx<T>* row = rows + (round * K::TPR + slot) * ROWP;
      cpx<T> v[16];
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) v[qq] = row[exch_phys(t + L * qq)];
      frame_sync<K::MULTIWAVE>();
      fft_pass<T, LOGM, 0, false, TW_REGS, K::MULTIWAVE>(v, row, twr, p.tw_m, t);
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) row[exch_phys(t + L * qq)] = v[qq];
    }
  }
  __syncthreads();
  // store: thread -> one output index k2 and ALL CF rows c of both halves: time n = t0 + c + P1*k2 (B) / d + D*m (Z),
  // so a thread owns CF contiguous samples and writes them as 16-byte pieces (Wx: CF*8 bytes, K: CF*2 bytes)
  const int s = p.scale0 + s_local;
  const T sc = p.out_scale[s];
  constexpr int SWEEPS = (M + kTileThreads - 1) / kTileThreads;
  constexpr bool WIDE = SSQ_CWT_WIDE_STORE && (CF <= 8);   // per-thread contiguous pieces: measured 3 % slower on C4
  if constexpr (!WIDE) {
    // short transforms (many rows per tile): element (c fastest, k2) per thread, as cwt_tile_kernel stores
    constexpr int PERS = CF * M / kTileThreads;
    static_assert((CF * M) % kTileThreads == 0, "whole store sweeps");
#pragma unroll 4
    for (int i = 0; i < PERS; ++i) {
      const int e = tid + i * kTileThreads;
      const int c = e % CF, k2 = e / CF;
      if (t0 + c >= P1) continue;
      const long long n = t0 + c + P1 * k2;
      if (n < p.n1 || n >= p.n1 + p.n_signal) continue;                  // unpad (ssq_cwt.rs:434-435)
      cpx<T> a = conj_if(rows[c * ROWP + exch_phys(k2)], true);
      cpx<T> b = conj_if(rows[(CF + c) * ROWP + exch_phys(k2)], true);
      a = {a.x * sc, a.y * sc};
      b = {b.x * sc, b.y * sc};
      T w;
      const int kk = reassign_bin(q, a, b, w);
      const long long o = (long long)s * p.n_signal + (n - p.n1);
      p.Wx[o] = a;
      p.K[o] = (short)kk;
      if (p.dWx) p.dWx[o] = b;
      if (q.wk) q.wk[o] = {w, (T)kk};
    }
    return;
  }
  constexpr int CW = WIDE ? CF : 1;                       // (array extents of the wide path only)
  const bool whole = (t0 + CF <= P1);
#pragma unroll 1
  for (int i = 0; i < SWEEPS; ++i) {
    const int k2 = tid + i * kTileThreads;
    if (k2 >= M) break;
    const long long nb = t0 + P1 * (long long)k2;                       // time of row c = 0
    if (nb + CW <= p.n1 || nb >= p.n1 + p.n_signal) continue;            // wholly inside the padding
    cpx<T> Wv[CW];
    short kv[CW];
    T wv[CW];
    cpx<T> dv[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) {
      cpx<T> a = conj_if(rows[c * ROWP + exch_phys(k2)], true);
      cpx<T> b = conj_if(rows[(CF + c) * ROWP + exch_phys(k2)], true);
      a = {a.x * sc, a.y * sc};                                          // 1/P (ssq_cwt.rs:405-418)
      b = {b.x * sc, b.y * sc};
      Wv[c] = a;
      dv[c] = b;
      kv[c] = (short)reassign_bin(q, a, b, wv[c]);
    }
    const long long o = (long long)s * p.n_signal + (nb - p.n1);
    const bool full = whole && nb >= p.n1 && nb + CW <= p.n1 + p.n_signal;
    if (full && !p.dWx && !q.wk && sizeof(T) == 4 && CW % 2 == 0 && ((o & 1) == 0)) {
      // 16-byte stores: two complex floats at a time; CF shorts in 16-byte (CF = 8) or 8-byte pieces
      float4* dst = reinterpret_cast<float4*>(p.Wx + o);
#pragma unroll
      for (int c = 0; c < CW; c += 2) dst[c / 2] = make_float4((float)Wv[c].x, (float)Wv[c].y, (float)Wv[c + 1].x, (float)Wv[c + 1].y);
      if constexpr (CW == 8) {
        if ((o & 7) == 0) {
          int4 pk;
          pk.x = (unsigned short)kv[0] | ((unsigned)(unsigned short)kv[1] << 16);
          pk.y = (unsigned short)kv[2] | ((unsigned)(unsigned short)kv[3] << 16);
          pk.z = (unsigned short)kv[4] | ((unsigned)(unsigned short)kv[5] << 16);
          pk.w = (unsigned short)kv[6] | ((unsigned)(unsigned short)kv[7] << 16);
          *reinterpret_cast<int4*>(p.K + o) = pk;
        } else {
#pragma unroll
          for (int c = 0; c < CW; c += 2)
            *reinterpret_cast<unsigned*>(p.K + o + c) = (unsigned short)kv[c] | ((unsigned)(unsigned short)kv[c + 1] << 16);
        }
      } else {
#pragma unroll
        for (int c = 0; c < CW; c += 2)
          *reinterpret_cast<unsigned*>(p.K + o + c) = (unsigned short)kv[c] | ((unsigned)(unsigned short)kv[c + 1] << 16);
      }
    } else {
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        const long long n = nb + c;
        if (t0 + c >= P1 || n < p.n1 || n >= p.n1 + p.n_signal) continue;    // unpad (ssq_cwt.rs:434-435)
        p.Wx[o + c] = Wv[c];
        p.K[o + c] = kv[c];
        if (p.dWx) p.dWx[o + c] = dv[c];
        if (q.wk) q.wk[o + c] = {wv[c], (T)kv[c]};
      }
    }
  }
}

template <typename T, int LOGM, int MODE>
static hipError_t launch_tile_ssq_mode(const CwtDev<T>& p, const CwtSsqDev<T>& q, hipStream_t stream) {
  using K = TileCfg<T, LOGM, false>;
  constexpr int CF = K::C / 2;
  const dim3 grid((unsigned)(((1LL << p.log_p1) + CF - 1) / CF), (unsigned)(p.n_transforms / 2), 1);
  hipLaunchKernelGGL((cwt_tile_ssq_kernel<T, LOGM, MODE>), grid, dim3(K::THREADS), 0, stream, p, q);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_cwt_tile_ssq(int mode, const CwtDev<T>& p, const CwtSsqDev<T>& q, hipStream_t stream) {
#define SSQ_CASE(LM)                                                                         \
  case LM:                                                                                   \
    return mode == CWT_INV_Z ? launch_tile_ssq_mode<T, LM, CWT_INV_Z>(p, q, stream)          \
                             : launch_tile_ssq_mode<T, LM, CWT_INV_B>(p, q, stream);
  switch (p.log_p2) {
    SSQ_CASE(4) SSQ_CASE(5) SSQ_CASE(6) SSQ_CASE(7) SSQ_CASE(8) SSQ_CASE(9) SSQ_CASE(10) SSQ_CASE(11) SSQ_CASE(12)
  }
#undef SSQ_CASE
  return hipErrorInvalidValue;
}

// Tx from (Wx, K): see cwt_kernels.h
template <typename T>
__global__ void cwt_reassign_k_kernel(CwtSsqDev<T> p, const short* __restrict__ K) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p.N) return;
  const cpx<T>* __restrict__ Wxp = p.Wx + j;
  const short* __restrict__ Kp = K + j;
  constexpr int UN = 8;
  int k_cur = -1;
  cpx<T> acc = {(T)0, (T)0};
  auto flush = [&]() {
    if (k_cur >= 0) {
      const long long d = (long long)k_cur * p.N + j;
      cpx<T> t = p.Tx[d];
      t.x += acc.x;
      t.y += acc.y;
      p.Tx[d] = t;
    }
  };
  for (int i0 = 0; i0 < p.na; i0 += UN) {
    cpx<T> Wb[UN];
    int kb[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int ii = (i0 + u < p.na) ? i0 + u : p.na - 1;
      Wb[u] = Wxp[(long long)ii * p.N];
      kb[u] = Kp[(long long)ii * p.N];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      if (i0 + u >= p.na) break;
      const int kk = kb[u];
      if (kk != k_cur) {
        flush();
        k_cur = kk;
        acc = {(T)0, (T)0};
      }
      if (kk >= 0) {
        if (p.squeezing == 1) {
          acc.x += p.leb_val;
        } else {
          acc.x += Wb[u].x;
          acc.y += Wb[u].y;
        }
      }
    }
  }
  flush();
}

template <typename T>
hipError_t launch_cwt_reassign_k(const CwtSsqDev<T>& p, const short* K, hipStream_t stream) {
  hipLaunchKernelGGL(cwt_reassign_k_kernel<T>, dim3((unsigned)((p.N + 63) / 64)), dim3(64), 0, stream, p, K);
  return hipGetLastError();
}

// One thread owns one time column and walks the scales in ascending order (no atomics, deterministic),
// read-modify-writing a zero-filled Tx; runs of scales that land in the same row are summed in registers first (the
// reference adds them to the row one by one: same sum up to the order of two roundings).  (An LDS-resident Tx tile [na][64 columns] with the rows split
// over 4 waves was measured 2.5x slower on C4: 128 KB of LDS leaves 4 waves per CU, too few loads in flight.)
// Scales [p.s_begin, p.s_end) only: the host may reassign group by group right behind the transforms that produced the
// group, while its Wx / dWx are still in the Infinity Cache (api_cwt.hip); rows are then read-modify-written once per
// group instead of once per call.  UN = scales whose Wx / dWx loads are in flight together.
template <typename T, int UN>
__global__ void cwt_reassign_kernel(CwtSsqDev<T> p) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= p.N) return;
  const cpx<T>* __restrict__ Wxp = p.Wx + j;
  const cpx<T>* __restrict__ dWxp = p.dWx + j;
  int k_cur = -1;
  cpx<T> acc = {(T)0, (T)0};
  for (int i0 = p.s_begin; i0 < p.s_end; i0 += UN) {
    cpx<T> Wb[UN], dWb[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int ii = (i0 + u < p.s_end) ? i0 + u : p.s_end - 1;
      Wb[u] = Wxp[(long long)ii * p.N];
      dWb[u] = dWxp[(long long)ii * p.N];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = i0 + u;
      if (i >= p.s_end) break;
      const long long o = (long long)i * p.N + j;
      const cpx<T> Wv = Wb[u];
      T w;
      const int kk = p.variant ? reassign_bin_upstream(p, Wv, dWb[u], w) : reassign_bin(p, Wv, dWb[u], w);
      if (p.wk) p.wk[o] = {w, (T)kk};
      // consecutive scales that land in the same row are summed in registers and written once (near a ridge many do)
      if (kk != k_cur) {
        if (k_cur >= 0) {
          const long long d = (long long)k_cur * p.N + j;
          cpx<T> t = p.Tx[d];
          t.x += acc.x;
          t.y += acc.y;
          p.Tx[d] = t;
        }
        k_cur = kk;
        acc = {(T)0, (T)0};
      }
      if (kk >= 0) {
        if (p.variant) {                     // upstream: out[k, j] += Wx[i, j] * const  (algos.py:910)
          if (p.squeezing == 1) {
            acc.x += p.leb_val * p.tx_const;
          } else {
            acc.x += Wv.x * p.tx_const;
            acc.y += Wv.y * p.tx_const;
          }
        } else if (p.squeezing == 1) {
          acc.x += p.leb_val;
        } else {
          acc.x += Wv.x;
          acc.y += Wv.y;
        }
      }
    }
  }
  if (k_cur >= 0) {
    const long long d = (long long)k_cur * p.N + j;
    cpx<T> t = p.Tx[d];
    t.x += acc.x;
    t.y += acc.y;
    p.Tx[d] = t;
  }
}

// The whole call in one launch WITHOUT a cleared Tx: every lane keeps a bitmap of the rows of its column it has written
// (in LDS, [word][lane]); the first run that lands in a row stores, a later one read-modify-writes, and at the end the
// rows never touched are stored as zeros -- every Tx cell is written once and (revisits aside) never read: 2.15 GB of
// traffic at C4 instead of clear 2.15 + read 1.8 + write 0.97 GB (profiles/r02_cwt_traffic_reg.json).
template <typename T, bool ZERO_FILL>
__global__ void cwt_reassign_sweep_kernel(CwtSsqDev<T> p) {
  extern __shared__ unsigned sweep_bits[];                // [ceil(na / 32)][64]
  constexpr int UN = 8;
  const int lane = threadIdx.x;
  const long long j0 = (long long)blockIdx.x * 64 + lane;
  const bool live = j0 < p.N;
  const long long j = live ? j0 : p.N - 1;                // dead lanes of the last block recompute the last column, store nothing
  const int words = (p.na + 31) >> 5;
  for (int w = 0; w < words; ++w) sweep_bits[w * 64 + lane] = 0u;
  const cpx<T>* __restrict__ Wxp = p.Wx + j;
  const cpx<T>* __restrict__ dWxp = p.dWx + j;
  cpx<T>* __restrict__ Txp = p.Tx + j;
  auto flush = [&](int k, cpx<T> acc) {
    unsigned* wp = sweep_bits + (k >> 5) * 64 + lane;
    const unsigned m = 1u << (k & 31), old = *wp;
    *wp = old | m;
    cpx<T>* d = Txp + (long long)k * p.N;
    if (old & m) {                                         // a second run in this row (the bins are not monotonic in the scale)
      const cpx<T> t = *d;
      acc.x += t.x;
      acc.y += t.y;
    }
    if (live) *d = acc;
  };
  int k_cur = -1;
  cpx<T> acc = {(T)0, (T)0};
  for (int i0 = p.s_begin; i0 < p.s_end; i0 += UN) {     // (the scales [s_begin, s_end); the bitmap spans all na rows)
    cpx<T> Wb[UN], dWb[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int ii = (i0 + u < p.s_end) ? i0 + u : p.s_end - 1;
      Wb[u] = Wxp[(long long)ii * p.N];
      dWb[u] = dWxp[(long long)ii * p.N];
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = i0 + u;
      if (i >= p.s_end) break;
      const cpx<T> Wv = Wb[u];
      T w;
      const int kk = reassign_bin(p, Wv, dWb[u], w);
      if (p.wk && live) p.wk[(long long)i * p.N + j] = {w, (T)kk};
      if (kk != k_cur) {
        if (k_cur >= 0) flush(k_cur, acc);
        k_cur = kk;
        acc = {(T)0, (T)0};
      }
      if (kk >= 0) {
        if (p.squeezing == 1) {
          acc.x += p.leb_val;
        } else {
          acc.x += Wv.x;
          acc.y += Wv.y;
        }
      }
    }
  }
  if (k_cur >= 0) flush(k_cur, acc);
  if (!live || !ZERO_FILL) return;                         // !ZERO_FILL: Tx was cleared beside the transforms
  for (int w = 0; w < words; ++w) {
    const unsigned done = sweep_bits[w * 64 + lane];
    const int rows = (p.na - 32 * w < 32) ? p.na - 32 * w : 32;
#pragma unroll 8
    for (int b = 0; b < rows; ++b)
      if (!((done >> b) & 1u)) Txp[(long long)(32 * w + b) * p.N] = {(T)0, (T)0};
  }
}

template <typename T>
bool cwt_reassign_can_sweep(int na) {
  return na <= 8192;                                       // 64 lanes x na / 8 bytes of LDS per block
}
template <typename T>
hipError_t launch_cwt_reassign_sweep(const CwtSsqDev<T>& p, hipStream_t stream, bool zero_fill) {
  if (p.s_end <= p.s_begin && !zero_fill) return hipSuccess;
  const size_t lds = (size_t)((p.na + 31) / 32) * 64 * sizeof(unsigned);
  const dim3 grid((unsigned)((p.N + 63) / 64));
  if (zero_fill) hipLaunchKernelGGL((cwt_reassign_sweep_kernel<T, true>), grid, dim3(64), lds, stream, p);
  else hipLaunchKernelGGL((cwt_reassign_sweep_kernel<T, false>), grid, dim3(64), lds, stream, p);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_cwt_reassign(const CwtSsqDev<T>& p, hipStream_t stream, bool clear) {
  if (clear) {
    const hipError_t e = hipMemsetAsync(p.Tx, 0, (size_t)p.na * (size_t)p.N * sizeof(cpx<T>), stream);
    if (e != hipSuccess) return e;
  }
  if (p.s_end <= p.s_begin) return hipSuccess;
  const dim3 grid((unsigned)((p.N + 63) / 64));
  if (p.s_end - p.s_begin <= 4) hipLaunchKernelGGL((cwt_reassign_kernel<T, 4>), grid, dim3(64), 0, stream, p);
  else hipLaunchKernelGGL((cwt_reassign_kernel<T, 8>), grid, dim3(64), 0, stream, p);
  return hipGetLastError();
}

template <typename T>
int cwt_tile_rows(int logm) {
  switch (logm) {
    case 4: return TileCfg<T, 4>::C;
    case 5: return TileCfg<T, 5>::C;
    case 6: return TileCfg<T, 6>::C;
    case 7: return TileCfg<T, 7>::C;
    case 8: return TileCfg<T, 8>::C;
    case 9: return TileCfg<T, 9>::C;
    case 10: return TileCfg<T, 10>::C;
    case 11: return TileCfg<T, 11>::C;
    case 12: return TileCfg<T, 12>::C;
  }
  return 0;
}

#define SSQ_INST(T)                                                                                   \
  template int cwt_tile_rows<T>(int);                                                                 \
  template hipError_t launch_cwt_tile<T>(int, const CwtDev<T>&, hipStream_t);                         \
  template hipError_t launch_wavelet_table<T>(T*, const long long*, const int*, int, const double*, int, long long, int, \
                                              hipStream_t, double, double);                          \
  template hipError_t launch_cwt_naive_fwd<T>(const CwtDev<T>&, hipStream_t);                         \
  template hipError_t launch_cwt_big_fwd<T>(const CwtDev<T>&, cpx<T>*, hipStream_t);                  \
  template hipError_t launch_cwt_big_inv<T>(const CwtDev<T>&, cpx<T>*, hipStream_t);                  \
  template hipError_t launch_cwt_naive_inv<T>(const CwtDev<T>&, int, hipStream_t);                    \
  template hipError_t launch_cwt_reassign<T>(const CwtSsqDev<T>&, hipStream_t, bool);                 \
  template hipError_t launch_cwt_reassign_sweep<T>(const CwtSsqDev<T>&, hipStream_t, bool);            \
  template bool cwt_reassign_can_sweep<T>(int);                                                        \
  template hipError_t launch_cwt_tile_ssq<T>(int, const CwtDev<T>&, const CwtSsqDev<T>&, hipStream_t); \
  template hipError_t launch_cwt_reassign_k<T>(const CwtSsqDev<T>&, const short*, hipStream_t);
SSQ_INST(float)
SSQ_INST(double)
#undef SSQ_INST

}  // namespace ssq
