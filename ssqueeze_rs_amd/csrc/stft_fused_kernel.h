// stft_fused_kernel.h -- the fused STFT / synchrosqueezed-STFT kernel template for gfx950 (MI355X): shared by
// stft_fused.hip (power-of-two n_fft, + the 16-wave n_fft = 1024 kernel) and stft_anylen.hip (any-length modes).
//
// One launch does, per tile of F consecutive frames of one signal:
//   reflect/zero padding by index mirroring   (stft_utils.rs:19-65)
//   window and diff-window multiply, packed as z = x*g + i*x*g'*fs
//                                              (stft_utils.rs:7-16, ssq_stft.rs:202-211)
//   ONE complex n_fft-point FFT per frame (the reference does two, ssq_stft.rs:226-227),
//   unpacked as Sx = (Z[k]+conj Z[N-k])/2, dSx = (Z[k]-conj Z[N-k])/(2i)
//   phase transform + nearest-bin index        (ssq_stft.rs:11-39, :280-289)
//   scatter-accumulate Tx[k, frame] += Sx*dw   (ssq_stft.rs:292-298) into an LDS tile
//   coalesced row-segment stores of the tile   (replaces the strided gather at :247-252)
// so HBM traffic is the algorithmic minimum: x once in, Tx (or Sx) once out.
//
// Work decomposition (CDNA4): a frame's N = 2^LOGN points are held by L = N/16 lanes,
// 16 complex values per lane (lane t owns elements t + L*q).  The FFT is a Stockham
// autosort with in-register radix-16 butterflies and one LDS exchange between passes;
// for N <= 1024 a frame lives inside one wavefront, so the exchanges need no block
// barrier.  The N-k partner for the real-pair unpack comes by ds_bpermute, not LDS.
// The next frame's samples are prefetched into registers before the current frame's FFT.
//
// The Tx tile is two planes [n_freqs][F+1] (odd pitch => the per-frame scatter spreads over all
// 32 banks) that accumulate FIXED POINT: LDS float atomics cost ~3 cycles per LANE on gfx950
// (187 cycles per wave instruction, measured: tools/ubench/lds_atomics.hip) while integer LDS
// atomics run at ~4 cycles per wave instruction.  Each column (frame) gets its own power-of-two
// scale 2^(FRAC-e) with 2^e > the column's L1 mass, so no partial sum can overflow, the
// quantisation step (2^-30 resp. 2^-50 of the column's L1 mass) sits below the FFT's own
// rounding error, and -- integer adds being associative -- the result is bitwise reproducible.
//
// The window tables arrive pre-multiplied by 1/2 (so the unpack needs no scaling) and the
// derivative channel by a power of two alpha (balances the two packed channels); alpha is
// folded into the 2*pi of the phase transform (StftDev::two_pi_eff).
#pragma once
#include <cstdlib>
#include <type_traits>
#include "fft_core.h"
#include "fft_pk1024.h"
#include "fft_mixed.h"
#include "stft_kernels.h"

namespace ssq {

#ifndef SSQ_F64_MAGIC
#define SSQ_F64_MAGIC 1     // fp64 fixed point by the 1.5 * 2^52 rounding trick (50 fraction bits) instead of f64 -> i64 conversions (62)
#endif

#ifndef SSQ_F64_W8
#define SSQ_F64_W8 1        // fp64 n_fft = 1024 with EIGHT waves per block (two per SIMD): exchange rows of T (re, then im), one-deep
                            // prefetch, weights parked in the row, window table in LDS: 1.95 -> 1.61 ms at batch 64 (r03_ab_f64_w8.txt)
#endif

// ANY: the any-length modes of the kernel (stft_anylen.hip) keep the plain configuration (their transforms use the
// exchange row as a row of complex values)
template <typename T, int LOGN, bool ANY = false>
struct FusedCfg {
  static constexpr int N = 1 << LOGN;
  static constexpr int L = N / 16;                         // lanes per frame
  static constexpr bool SPLIT = SSQ_F64_W8 && sizeof(T) == 8 && LOGN == 10 && !ANY;   // exchange the components one after the other
  static constexpr int W = (sizeof(T) == 4 || SPLIT) ? 8 : 4;   // waves per block
  static constexpr int FPW = (L >= 64) ? 1 : 64 / L;       // frames per wave
  static constexpr int WPF = (L <= 64) ? 1 : L / 64;       // waves per frame
  static constexpr int FIF = W * FPW / WPF;                // frames in flight per block
  static constexpr int NF = N / 2 + 1;
  static constexpr int EXCH_ELEMS = N + N / 16;            // +1 element per 16: bank spread
  static constexpr int EXCH_BYTES = FIF * EXCH_ELEMS * (int)(SPLIT ? sizeof(T) : sizeof(cpx<T>));
  // window table in LDS (SPLIT: -14 %; the W_1024 table there instead, random-index reads, measured 5 % slower than
  // leaving it to L1: profiles/r03_ab_f64_w8.txt)
  static constexpr bool WIN_LDS = ((sizeof(T) == 4) && (N <= 1024)) || SPLIT;
  static constexpr int WIN_BYTES = WIN_LDS ? N * (int)sizeof(cpx<T>) : 0;
  static constexpr int TWL_BYTES = (WIN_LDS && !SPLIT) ? N * (int)sizeof(cpx<T>) : 0;   // W_N table in LDS (paired-frame / mixed-radix kernels)
  static constexpr int LDS_MAX = 160 * 1024;
  static constexpr int FMAX = (LDS_MAX - EXCH_BYTES - WIN_BYTES - TWL_BYTES - 1024) / (2 * NF * (int)sizeof(T)) - 1;
  static constexpr int FT = (sizeof(T) == 4) ? 16 : 8;     // target: >=128-B row segments
  static constexpr int FCAP = (FT < FMAX) ? FT : FMAX;
  static constexpr int F = (FIF >= FT) ? FIF : (FCAP / FIF) * FIF;
  static constexpr int PITCH = F + 1;
  static constexpr int PLANE = NF * PITCH;                 // elements per plane
  static constexpr int TILE_BYTES = (((2 * PLANE + F) * (int)sizeof(T) + 15) / 16) * 16;   // + col_scale[F]
  using IT = std::conditional_t<sizeof(T) == 4, int, long long>;
  using UT = std::conditional_t<sizeof(T) == 4, unsigned int, unsigned long long>;
  static constexpr int FRAC = (sizeof(T) == 4) ? 30 : (SSQ_F64_MAGIC ? 50 : 62);      // fixed-point fraction bits (fp64: below 2^51, to_fixed)
  static constexpr int EMIN = (sizeof(T) == 4) ? -90 : -960;   // keeps 2^(FRAC-e) finite
  static constexpr int LDS_BYTES = TILE_BYTES + EXCH_BYTES + WIN_BYTES + TWL_BYTES;
  static constexpr int NP = num_passes(LOGN);
  static constexpr bool TW_REGS = (sizeof(T) == 4);
  static constexpr int ITERS = F / FIF;                    // frame groups per tile
  static_assert(F >= FIF && F % FIF == 0, "tile must hold whole in-flight groups");
  static_assert(LDS_BYTES <= LDS_MAX, "LDS budget");
};

__device__ __forceinline__ int cvt_round_i32(float x);

// bit casts between T and its integer twin (debug outputs travel through the integer tile)
template <typename T>
__device__ __forceinline__ std::conditional_t<sizeof(T) == 4, int, long long> as_int(T v) {
  if constexpr (sizeof(T) == 4) return __float_as_int(v);
  else return __double_as_longlong(v);
}
template <typename T>
__device__ __forceinline__ T from_int(std::conditional_t<sizeof(T) == 4, int, long long> v) {
  if constexpr (sizeof(T) == 4) return __int_as_float(v);
  else return __longlong_as_double(v);
}
template <typename T>
__device__ __forceinline__ std::conditional_t<sizeof(T) == 4, int, long long> to_fixed(T v) {
  if constexpr (sizeof(T) == 4) return cvt_round_i32(v);      // floor(v + 1/2): one instruction
  else if constexpr (!SSQ_F64_MAGIC) return __double2ll_rn(v);
  else {
    // |v| <= 2^50 (FRAC): adding 1.5 * 2^52 leaves round-to-nearest-even(v) in the low mantissa bits -- one fp64 add and
    // one 64-bit subtract instead of the ~18-instruction f64 -> i64 conversion (there is no native one)
    constexpr double kMagic = 6755399441055744.0;
    return __double_as_longlong(v + kMagic) - __double_as_longlong(kMagic);
  }
}
// the way back for |i| < 2^51: exact
template <typename T>
__device__ __forceinline__ T fixed_to_real(std::conditional_t<sizeof(T) == 4, int, long long> i) {
  if constexpr (sizeof(T) == 4 || !SSQ_F64_MAGIC) return (T)i;
  else {
    constexpr double kMagic = 6755399441055744.0;
    return __longlong_as_double(i + __double_as_longlong(kMagic)) - kMagic;
  }
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Sum of `v` over the L lanes that hold one frame; every lane of the frame gets the SAME value
// (each step adds a lane and its exchange partner, and fp addition commutes).
template <typename T, int L, bool MULTIWAVE>
__device__ __forceinline__ T frame_allreduce(T v, int lane, T* scratch, int t) {
  constexpr int LW = (L < 64) ? L : 64;
  if constexpr (sizeof(T) == 4) {
    if (LW >= 2) v += dpp_mov<0xB1>(v);       // quad_perm [1,0,3,2]
    if (LW >= 4) v += dpp_mov<0x4E>(v);       // quad_perm [2,3,0,1]
    if (LW >= 8) v += dpp_mov<0x141>(v);      // row_half_mirror
    if (LW >= 16) v += dpp_mov<0x140>(v);     // row_mirror
    if (LW >= 32) {
      const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0));
      const float r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
      const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32));
      const float r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
      if (LW == 32) v = (lane < 32) ? (r0 + r1) : (r2 + r3);
      else v = (r0 + r1) + (r2 + r3);
    }
  } else {
#pragma unroll
    for (int m = LW / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  }
  if constexpr (MULTIWAVE) {
    // a frame spans L/64 waves: combine the wave totals through the frame's (idle) exchange row
    __syncthreads();
    if (lane == 0) scratch[t >> 6] = v;
    __syncthreads();
    T s = (T)0;
#pragma unroll
    for (int w = 0; w < L / 64; ++w) s += scratch[w];
    __syncthreads();
    v = s;
  }
  return v;
}

// single-instruction helpers (inline asm: no builtin exists for these forms)
__device__ __forceinline__ float fma_clamp01(float a, float b, float c) {   // clamp(a*b + c) to [0,1]; NaN -> 0
  float r;
  asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ int cvt_floor_i32(float x) {                     // floor(x); NaN -> 0; saturates
  int r;
  asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}
__device__ __forceinline__ int cvt_round_i32(float x) {                     // floor(x + 0.5); saturates
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
}

// Power-of-two fixed-point scale of a column: 2^e > tot, scale = dw * 2^(FRAC-e), inv = 2^(e-FRAC).
template <typename T, int FRAC, int EMIN>
__device__ __forceinline__ void column_scale(T tot, T dw, T& scale, T& inv_scale) {
  if constexpr (sizeof(T) == 4) {
    const int ex = (__float_as_int(tot) >> 23) & 0xff;          // tot >= 0
    int e = ex - 126;                                            // frexp exponent: tot < 2^e
    e = e < EMIN ? EMIN : e;
    scale = dw * __int_as_float((127 + FRAC - e) << 23);
    inv_scale = __int_as_float((127 + e - FRAC) << 23);
    if (ex == 255) {                                             // NaN/Inf: the column comes out NaN
      scale = 0.0f;
      inv_scale = __int_as_float(0x7fc00000);
    }
  } else {
    int e = 0;
    (void)frexp(tot, &e);
    if (e < EMIN) e = EMIN;
    scale = ldexp(dw, FRAC - e);
    inv_scale = ldexp((T)1, e - FRAC);
    if (!(tot < (T)INFINITY)) {
      scale = (T)0;
      inv_scale = tot - tot;
    }
  }
}

// one unit of work of a lane: frame `fl` of tile (sig, ft) -> where its samples are
template <typename T>
struct FrameItem {
  const T* xs;          // signal base
  long long pos0;       // original-signal index of this lane's element q = 0
  int fl;               // frame index inside the tile
  int valid;            // frame < n_frames  (int, not bool: sub-dword struct members end up in an
                        // LDS-promoted alloca with unaligned 16-bit accesses = 64-cycle replays)
};

// a tile of F frames of one signal
struct TileItem {
  long long sig;
  int ft;               // tile index inside the signal
  int frame0;           // first frame
};

// j = index of the tile among those this launch covers for one signal (StftDev::ta0/ta_n/tb0)
template <typename T, int LOGN>
__device__ __forceinline__ TileItem make_tile(const StftDev<T>& p, long long sig, int j) {
  using C = FusedCfg<T, LOGN>;
  TileItem w;
  w.sig = sig;
  w.ft = j;
  const int ft = (j < p.ta_n) ? p.ta0 + j : p.tb0 + (j - p.ta_n);
  w.frame0 = ft * C::F;
  return w;
}

template <typename T, int LOGN, bool EDGE, bool ANY = false>
__device__ __forceinline__ FrameItem<T> make_frame(const StftDev<T>& p, const TileItem& tl, int it, int slot, int t) {
  using C = FusedCfg<T, LOGN, ANY>;
  FrameItem<T> w;
  w.fl = it * C::FIF + slot;
  const int frame = tl.frame0 + w.fl;
  w.valid = EDGE ? ((frame < p.n_frames) ? 1 : 0) : 1;
  w.xs = sig_base(p, tl.sig);
  w.pos0 = (long long)frame * p.hop - p.pad_left + t;
  return w;
}

template <typename T, int LOGN, bool EDGE, int MODE = 0>
__device__ __forceinline__ void load_samples(const StftDev<T>& p, const TileItem& tl, const FrameItem<T>& w,
                                             T (&xv)[16], int t = 0) {
  constexpr int L = FusedCfg<T, LOGN>::L;
  constexpr bool BLUE = MODE == 1;
  if constexpr (MODE == 2) {
    // mixed-radix mode: the n_eff <= 2^LOGN samples of the frame, the rest of the row's slots stay unused
    // (one wave-uniform branch around two straight runs of loads: a branch per sample makes every load wait for itself)
    const long long first = w.pos0 - t;
    const bool inside = w.valid && first >= 0 && first + p.n_eff <= p.n_signal;
    if (__all(inside)) {
#pragma unroll
      for (int q = 0; q < 16; ++q) xv[q] = (t + L * q < p.n_eff) ? w.xs[w.pos0 + L * q] : (T)0;
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q)
        xv[q] = load_padded_flat(w.xs, w.pos0 + L * q, p.n_signal, p.padtype, w.valid && t + L * q < p.n_eff);
    }
  } else if constexpr (BLUE) {
    // only the n_eff samples of the frame are read (the table's zero padding must not meet a NaN beyond it); they all
    // sit in q < 8 (n_eff <= (m + 1)/2), and a frame that lies inside the signal needs no mirroring logic
    const long long first = w.pos0 - t;
    const bool inside = w.valid && first >= 0 && first + p.n_eff <= p.n_signal;
    if (__all(inside)) {
#pragma unroll
      for (int q = 0; q < 8; ++q) xv[q] = (t + L * q < p.n_eff) ? w.xs[w.pos0 + L * q] : (T)0;
    } else {
#pragma unroll
      for (int q = 0; q < 8; ++q)
        xv[q] = load_padded_flat(w.xs, w.pos0 + L * q, p.n_signal, p.padtype, w.valid && t + L * q < p.n_eff);
    }
#pragma unroll
    for (int q = 8; q < 16; ++q) xv[q] = (T)0;
  } else if constexpr (!EDGE) {
#pragma unroll
    for (int q = 0; q < 16; ++q) xv[q] = w.xs[w.pos0 + L * q];
  } else {
    const long long first = w.pos0 - t;
    if (__all(w.valid && first >= 0 && first + FusedCfg<T, LOGN>::N <= p.n_signal)) {
#pragma unroll
      for (int q = 0; q < 16; ++q) xv[q] = w.xs[w.pos0 + L * q];
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) xv[q] = load_padded_flat(w.xs, w.pos0 + L * q, p.n_signal, p.padtype, w.valid != 0);
    }
  }
}

#ifdef SSQ_STAMPS
// In-kernel phase stamps (diagnostic build only; never quote its run time, read its SHARES).
__device__ __forceinline__ unsigned long long ssq_stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define SSQ_STAMP(i)                         \
  do {                                       \
    const unsigned long long t_ = ssq_stamp(); \
    st_acc[i] += t_ - st_prev;               \
    st_prev = t_;                            \
  } while (0)
#elif defined(SSQ_MARK)
// static section markers for instruction counting in the .s (tools/count_sections.py)
#define SSQ_STAMP(i)                                   \
  do {                                                 \
    __builtin_amdgcn_sched_barrier(0);                 \
    asm volatile("; SSQ_SECTION " #i ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                 \
  } while (0)
#else
#define SSQ_STAMP(i) do { } while (0)
#endif

#ifdef SSQ_ABLATE_HOOKS
#define SSQ_ABL(mask) (p.ablate & (mask))       // timing experiments (tools/ablate.sh); results are wrong
#else
#define SSQ_ABL(mask) false
#endif

// TXONLY = true : out_kind == SSQ_OUT_TX (the hot path: branch-free epilogue; at n_fft = 1024 fp32
//                 a wave runs its two frames of a tile staggered, so one frame's LDS round trips
//                 hide behind the other frame's arithmetic)
// TXONLY = false: SSQ_OUT_SX / DSX / WK  (stft and the test hooks)
// EDGE = false: tiles whose frames all lie inside the signal (direct loads, every frame valid);
// EDGE = true : the few tiles per signal that touch a boundary (padding by index mirroring).
// WKDBG (test hook, SSQ_OUT_WK): the TXONLY epilogue stores ITS OWN (w, k) of every bin instead of scattering, so the
// tests observe the bins of the very arithmetic that serves SSQ_OUT_TX (k = -1 where the bin is skipped).
// BLUE: Bluestein mode (any n_fft = p.n_eff with 2*n_eff - 1 <= N): chirp folded into the window table, FFT, multiply by
// the chirp filter's spectrum, second FFT (the inverse, on conjugated data), output chirp; bins and their partners are
// then Z[k], Z[(n - k) mod n], k < n_freqs = n_eff/2 + 1 (fetched through the exchange row: no lane symmetry here).
// MODE: 0 = power-of-two n_fft; 1 = BLUE; 2 = MIXR, any n_fft = p.n_eff <= 2^LOGN whose prime factors are <= 13: the
// frame is transformed in its exchange row by the mixed-radix passes of fft_mixed.h (p.mr_radix), same epilogue as BLUE.
template <typename T, int LOGN, bool TXONLY, bool EDGE, bool LEB, bool WKDBG = false, int MODE = 0>
__global__ __launch_bounds__((FusedCfg<T, LOGN, MODE != 0>::W * 64)) void stft_fused_kernel(StftDev<T> p) {
  using C = FusedCfg<T, LOGN, MODE != 0>;
  constexpr int N = C::N, L = C::L, NF = C::NF, F = C::F, PITCH = C::PITCH;
  constexpr bool MULTIWAVE = (C::WPF > 1);
  constexpr bool BLUE = MODE == 1, MIXR = MODE == 2, ANYLEN = MODE != 0;
  static_assert(!ANYLEN || EDGE, "the any-length modes run the edge-capable loader (masked samples)");
  // Staggering a wave's two frames (fft_pass_pair) measured SLOWER here (4.77 vs 3.56 ms): the second
  // frame's registers push the kernel into scratch.  Kept behind this switch for the next round.
  constexpr bool PAIR = false && TXONLY && !MULTIWAVE && (sizeof(T) == 4) && (C::ITERS % 2 == 0);
  constexpr int NFW = PAIR ? 2 : 1;              // frames a wave works on together
  constexpr int NG = C::ITERS / NFW;             // such groups per tile
  __shared__ __attribute__((aligned(16))) unsigned char smem[C::LDS_BYTES];
  using IT = typename C::IT;                     // integer twin of T: the tile accumulates fixed point
  using UT = typename C::UT;
  IT* tile_re = reinterpret_cast<IT*>(smem);
  IT* tile_im = tile_re + C::PLANE;
  T* col_scale = reinterpret_cast<T*>(tile_im + C::PLANE);     // [F] 2^(e-FRAC) per column
  cpx<T>* exch_all = reinterpret_cast<cpx<T>*>(smem + C::TILE_BYTES);
  cpx<T>* win_lds = reinterpret_cast<cpx<T>*>(smem + C::TILE_BYTES + C::EXCH_BYTES);
  cpx<T>* tw_lds = reinterpret_cast<cpx<T>*>(smem + C::TILE_BYTES + C::EXCH_BYTES + C::WIN_BYTES);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // wave-uniform: keeps its derived values in scalar registers
  int slot, t;
  if constexpr (L <= 64) {
    slot = wave * C::FPW + lane / L;
    t = lane % L;
  } else {
    slot = wave / C::WPF;
    t = (wave % C::WPF) * 64 + lane;
  }
  cpx<T>* exch = exch_all + (C::SPLIT ? 0 : slot * C::EXCH_ELEMS);
  T* exch_s = reinterpret_cast<T*>(exch_all) + slot * C::EXCH_ELEMS;       // SPLIT: a row of T per frame

  // ---- per-lane constants, live across all tiles this block processes ----
  constexpr bool TW_REGS = C::TW_REGS && !PAIR && !MIXR;   // paired frames: registers go to the second frame, twiddles to LDS
  cpx<T> twr[3][16];
  if constexpr (TW_REGS) {
#pragma unroll
    for (int P = 1; P < C::NP; ++P) {
      const int R = pass_radix(LOGN, P), NS = pass_ns(LOGN, P), NB = 16 / R;
#pragma unroll
      for (int b = 0; b < 16; ++b) {
#pragma unroll
        for (int m = 1; m < 16; ++m) {
          if (b < NB && m < R) {
            const int k = (t + L * b) & (NS - 1);
            twr[P - 1][b + m * NB] = p.tw[k * m * (N / (NS * R))];
          }
        }
      }
    }
  }
  if constexpr (C::WIN_LDS) {
    for (int i = tid; i < N; i += C::W * 64) win_lds[i] = p.win2[i];
  }
  constexpr bool TW_LDS = PAIR || (MIXR && C::WIN_LDS && !C::SPLIT);      // W_n table in LDS (the area exists with WIN_LDS)
  if constexpr (TW_LDS) {
    for (int i = tid; i < N; i += C::W * 64) tw_lds[i] = p.tw[i];
  }
  const cpx<T>* tw_src = TW_LDS ? tw_lds : p.tw;
  // zero the tile once; afterwards the read-out pass re-zeroes what it reads
  for (int i = tid; i < 2 * C::PLANE; i += C::W * 64) tile_re[i] = 0;
  __syncthreads();

  if ((long long)blockIdx.x >= p.total_tiles) return;
  // Work items of this lane: (tile, frame group).  Samples are prefetched TWO items ahead, so that
  // the loads of the next tile never queue behind this tile's read-out burst of stores.
  struct Item {
    TileItem tl;
    int ig;
    int ok;
  };
  const long long n_sig = p.total_tiles / p.tiles_per_signal;
  const int grid_n = (int)gridDim.x;          // (read once: inside the loop it is a scalar load + wait per iteration)
  auto advance = [&](const Item& c) {
    Item n = c;
    n.ig = c.ig + 1;
    if (n.ig == NG) {
      n.ig = 0;
      long long ns = c.tl.sig;
      int nft = c.tl.ft + grid_n;
      while (nft >= p.tiles_per_signal) {
        nft -= p.tiles_per_signal;
        ++ns;
      }
      n.tl = make_tile<T, LOGN>(p, ns, nft);
    }
    n.ok = (c.ok && (n.tl.sig < n_sig)) ? 1 : 0;
    return n;
  };
  Item i0;
  i0.tl = make_tile<T, LOGN>(p, (long long)(blockIdx.x / (unsigned)p.tiles_per_signal),
                             (int)(blockIdx.x % (unsigned)p.tiles_per_signal));
  i0.ig = 0;
  i0.ok = 1;
  Item i1 = advance(i0);
  static_assert(NFW == 1, "frame pairing is parked (see PAIR)");
  constexpr bool DEEP = !C::SPLIT;     // SPLIT: one item ahead (registers; two waves per SIMD hide the rest)
  FrameItem<T> cur[NFW];
  T xn[NFW][16];      // samples of the current item
  T xb[DEEP ? 16 : 1];           // samples of the next item
  cur[0] = make_frame<T, LOGN, EDGE, MODE != 0>(p, i0.tl, i0.ig, slot, t);
  load_samples<T, LOGN, EDGE, MODE>(p, i0.tl, cur[0], xn[0], t);
  FrameItem<T> fr1 = cur[0];
  if (i1.ok) {
    fr1 = make_frame<T, LOGN, EDGE, MODE != 0>(p, i1.tl, i1.ig, slot, t);
    if constexpr (DEEP) load_samples<T, LOGN, EDGE, MODE>(p, i1.tl, fr1, xb, t);
  }
#ifdef SSQ_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_prev = ssq_stamp();
#endif

#pragma unroll 1
  while (true) {
    const TileItem tl = i0.tl;
    const int ig = i0.ig;
    // SPLIT: the window and twiddle tables are read from memory (L1) in EVERY iteration -- the loads are loop
    // invariant, and hoisted out of the loop they would hold (and spill) ~180 registers; an opaque zero offset per
    // iteration keeps them where they are used
    const cpx<T>* win_it = p.win2;
    const cpx<T>* tw_it = tw_src;
    if constexpr (C::SPLIT) {
      int z;
      asm volatile("s_mov_b32 %0, 0" : "=s"(z));
      win_it += z;
      tw_it += z;
    }
    // ---- window multiply ----
    cpx<T> v[NFW][16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const cpx<T> wq = C::WIN_LDS ? win_lds[t + L * q] : win_it[t + L * q];
      v[0][q] = {xn[0][q] * wq.x, xn[0][q] * wq.y};
    }
    SSQ_STAMP(0);
    const bool has_next = i1.ok;
    // rotate the prefetch ring: the next item's samples (loaded one iteration ago) move to xn and the
    // loads of the item after next go out now, a full iteration before they are needed
    if constexpr (DEEP) {
#pragma unroll
      for (int q = 0; q < 16; ++q) xn[0][q] = xb[q];
    } else {
      // (issued behind the transform instead -- so that its twiddle loads do not queue behind these -- measured 3 %
      //  slower: the epilogue alone does not cover the latency before the tile barrier drains the loads)
      if (i1.ok) load_samples<T, LOGN, EDGE, MODE>(p, i1.tl, fr1, xn[0], t);
    }
    const Item i2 = advance(i1);
    FrameItem<T> fr2 = fr1;
    if (i2.ok) {
      fr2 = make_frame<T, LOGN, EDGE, MODE != 0>(p, i2.tl, i2.ig, slot, t);
      if constexpr (DEEP) {
        if (!SSQ_ABL(1)) load_samples<T, LOGN, EDGE, MODE>(p, i2.tl, fr2, xb, t);
      }
    }

    SSQ_STAMP(1);
    if constexpr (MIXR) {
      // the windowed frame goes to its row, is transformed there, and the bins come back below (Z[k] and partner)
#pragma unroll
      for (int q = 0; q < 16; ++q)
        if (t + L * q < p.n_eff) exch[exch_phys(t + L * q)] = v[0][q];
      frame_sync<MULTIWAVE>();
      fft_mixed_row<T, L, MULTIWAVE>(exch, p.n_eff, p.mr_np, p.mr_radix, tw_src, t);
    } else if (!SSQ_ABL(2)) {
      if constexpr (PAIR) fft_pass_pair<T, LOGN, 0, false, false>(v[0], v[1], exch, twr, tw_src, t);
      else if constexpr (C::SPLIT) fft_pass_split<T, LOGN, 0>(v[0], exch_s, twr, tw_it + N, t);   // compact tables behind W_N
      else fft_pass<T, LOGN, 0, false, TW_REGS, MULTIWAVE>(v[0], exch, twr, tw_src, t);
    }
    // lane t now holds Z[t + L*q], q = 0..15 (natural order residue class t mod L)
    if constexpr (BLUE) {
      // Y * B^ (B^ carries the 1/m), conjugate, forward FFT again = conj of the inverse transform, output chirp
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const cpx<T> y = cmul(v[0][q], p.blue_b[t + L * q]);
        v[0][q] = {y.x, -y.y};
      }
      if constexpr (MULTIWAVE) __syncthreads();          // the exchange row is reused by the second transform
      fft_pass<T, LOGN, 0, false, TW_REGS, MULTIWAVE>(v[0], exch, twr, tw_src, t);
#pragma unroll
      for (int q = 0; q < 8; ++q) {                        // outputs k < n_eff <= m/2 only: q < 8
        const int k = t + L * q;
        const cpx<T> c = (k < p.n_eff) ? p.blue_post[k] : cpx<T>{(T)0, (T)0};
        v[0][q] = cmul(cpx<T>{v[0][q].x, -v[0][q].y}, c);
      }
    }

    SSQ_STAMP(2);
    // ---- partner Z[N-k] for the bins this lane owns: k = t + L*q, q < 8 (+ k = N/2 on t == 0)
    cpx<T> zp[NFW][9];
#pragma unroll
    for (int f = 0; f < NFW; ++f) {
      if constexpr (MIXR) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {                                        // k <= n_eff/2 < 8L: q < 8 holds every bin
          const int k = t + L * q;
          const int kc = k < p.n_eff ? k : 0;
          const int kp = (k == 0 || k >= p.n_eff) ? 0 : p.n_eff - k;       // (n - k) mod n
          v[f][q] = exch[exch_phys(kc)];
          zp[f][q] = exch[exch_phys(kp)];
        }
        v[f][8] = {(T)0, (T)0};
        zp[f][8] = v[f][8];
        frame_sync<MULTIWAVE>();
      } else if constexpr (BLUE) {
        frame_sync<MULTIWAVE>();
#pragma unroll
        for (int q = 0; q < 8; ++q) exch[exch_phys(t + L * q)] = v[f][q];   // every k < n_eff (and every partner) has q < 8
        frame_sync<MULTIWAVE>();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = t + L * q;
          const int kp = (k == 0 || k >= p.n_eff) ? 0 : p.n_eff - k;       // (n - k) mod n
          zp[f][q] = exch[exch_phys(kp)];
        }
        zp[f][8] = v[f][8];
        frame_sync<MULTIWAVE>();
      } else if constexpr (C::SPLIT && TXONLY) {
        // (fetched bin by bin inside the epilogue: registers)
      } else if constexpr (!MULTIWAVE) {
        const int src = (lane - t) + ((L - t) & (L - 1));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          cpx<T> r;
          r.x = __shfl(v[f][15 - q].x, src);
          r.y = __shfl(v[f][15 - q].y, src);
          if (t == 0) r = (q == 0) ? v[f][0] : v[f][16 - q];
          zp[f][q] = r;
        }
        zp[f][8] = v[f][8];                      // k = N/2 pairs with itself (t == 0 only)
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) exch[exch_phys(t + L * q)] = v[f][q];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = t + L * q;
          zp[f][q] = exch[exch_phys((N - k) & (N - 1))];
        }
        zp[f][8] = v[f][8];
        __syncthreads();
      }
    }

    SSQ_STAMP(3);
#pragma unroll
    for (int f = 0; f < NFW; ++f) {
      const int fl = cur[f].fl;
      if constexpr (TXONLY) {
        // ---- unpack, phase transform, bin index: branch-free; skipped bins contribute 0 ----
        cpx<T> cv[9];
        int dstb[9];                                 // byte offset of the destination inside a plane
        T wdbg[9];
        int kdbg[9];
        T l1 = (T)0;
        const int fl4 = fl * (int)sizeof(T);
        if constexpr (sizeof(T) == 4) {
          // fp32 hot path, tuned by the measured op costs (tools/ubench/valu_rate2.hip: compares, selects,
          // min/max, conversions are half rate): masks by clamped fma instead of compare+select, bin
          // index by one floor-convert, no select on the destination (a masked-out bin adds 0 anywhere)
          const float lane_on = EDGE ? (cur[f].valid ? 1.0f : 0.0f) : 1.0f;
          const float sfs0 = (float)t * p.sfs_step, sfs_q = (float)L * p.sfs_step;
          const int neg_last = -(p.n_freqs - 1);
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            const cpx<T> zk = v[f][q], zn = zp[f][q];
            const cpx<T> S = {zk.x + zn.x, zk.y - zn.y};
            const cpx<T> dS = {zk.y + zn.y, zn.x - zk.x};       // alpha * dSx
            const float den = S.x * S.x + S.y * S.y;
            const float num = dS.y * S.x - dS.x * S.y;
            const float pd = num * __builtin_amdgcn_rcpf(den * p.two_pi_eff);
            const float w = fabsf((sfs0 + (float)q * sfs_q) - pd);            // ssq_stft.rs:33
            // keep = (|Sx|^2 >= gamma^2) and (w finite)   (ssq_stft.rs:23, :278) as a 0/1 float
            float m = fma_clamp01(den, p.keep_big, p.keep_bias) * fma_clamp01(w, 0.0f, 1.0f);
            if (EDGE) m *= lane_on;
            if (q == 8) m *= (t == 0) ? 1.0f : 0.0f;                           // bin N/2 lives on lane 0 only
            if constexpr (ANYLEN) m *= (q < 8 && t + L * q < p.n_freqs) ? 1.0f : 0.0f;   // bins of the n_eff-point transform
            if (SSQ_ABL(8)) m = lane_on;
            const cpx<T> c = LEB ? cpx<T>{p.leb_unit * m, 0.0f} : cpx<T>{S.x * m, S.y * m};   // weight (:292-296)
            cv[q] = c;
            // kk = ceil(w/dw - 1/2) = -floor(1/2 - w/dw), clamped to the last bin (ssq_stft.rs:280-289)
            int kneg = cvt_floor_i32(__builtin_fmaf(-w, p.inv_dw, 0.5f));
            kneg = kneg < neg_last ? neg_last : kneg;
            dstb[q] = __mul24(kneg, -(PITCH * (int)sizeof(T))) + fl4;
            l1 += fabsf(c.x) + fabsf(c.y);
            if constexpr (WKDBG) {
              wdbg[q] = w;
              kdbg[q] = (m != 0.0f) ? -kneg : -1;
            }
          }
        } else {
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            const int k = t + L * q;
            cpx<T> zn;
            if constexpr (C::SPLIT) {
              // partner Z[N-k] right here, and the weight parked in the frame's (idle) exchange row until the column's
              // scale is known: the lane reads back its own slots, so no ordering point is needed
              const int src = (lane - t) + ((L - t) & (L - 1));
              if (q < 8) {
                zn.x = __shfl(v[f][15 - q].x, src);
                zn.y = __shfl(v[f][15 - q].y, src);
                if (t == 0) zn = (q == 0) ? v[f][0] : v[f][16 - q];
              } else {
                zn = v[f][8];
              }
            } else {
              zn = zp[f][q];
            }
            const cpx<T> zk = v[f][q];
            const cpx<T> S = {zk.x + zn.x, zk.y - zn.y};
            const cpx<T> dS = {zk.y + zn.y, zn.x - zk.x};       // alpha * dSx
            int kk = k;
            T w;
            bool keep = phase_bin<T>(p, k, S, dS, w, kk);
            keep = keep && cur[f].valid && (q < 8 || t == 0);
            if constexpr (ANYLEN) keep = keep && q < 8 && k < p.n_freqs;
            cpx<T> c = LEB ? cpx<T>{p.leb_unit, (T)0} : S;   // weight  (ssq_stft.rs:292-296)
            c.x = keep ? c.x : (T)0;
            c.y = keep ? c.y : (T)0;
            if constexpr (C::SPLIT && !WKDBG) {
              if (q < 8) {
                exch_s[(2 * q) * 64 + t] = c.x;
                exch_s[(2 * q + 1) * 64 + t] = c.y;
              } else {
                cv[8] = c;
              }
            } else
            cv[q] = c;
            dstb[q] = (keep ? kk : 0) * (PITCH * (int)sizeof(T)) + fl4;
            l1 += fabs(c.x) + fabs(c.y);
            if constexpr (WKDBG) {
              wdbg[q] = w;
              kdbg[q] = keep ? kk : -1;
            }
          }
        }
        SSQ_STAMP(4);
        // fixed-point scatter: every partial sum of this column is bounded by its L1 mass dw*sum|c|;
        // pick 2^e above it and accumulate round(c * dw * 2^(FRAC-e)) with integer LDS atomics
        const T tot = frame_allreduce<T, L, MULTIWAVE>(l1, lane, reinterpret_cast<T*>(exch), t) * p.dw;
        T scale, inv_scale;
        column_scale<T, C::FRAC, C::EMIN>(tot, p.dw, scale, inv_scale);
        if (t == 0 && cur[f].valid) col_scale[fl] = inv_scale;
        SSQ_STAMP(5);
        char* pre = reinterpret_cast<char*>(tile_re);
        char* pim = reinterpret_cast<char*>(tile_im);
        if constexpr (WKDBG) {
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            if ((q < 8 || t == 0) && cur[f].valid && (!ANYLEN || (q < 8 && t + L * q < p.n_freqs))) {
              const int o = (t + L * q) * PITCH + fl;
              tile_re[o] = as_int<T>(wdbg[q]);
              tile_im[o] = as_int<T>((T)kdbg[q]);
            }
          }
        } else if constexpr (LEB) {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if constexpr (C::SPLIT) cv[q].x = exch_s[(2 * q) * 64 + t];
            atomicAdd(reinterpret_cast<UT*>(pre + dstb[q]), (UT)to_fixed<T>(cv[q].x * scale));
          }
          if (t == 0) atomicAdd(reinterpret_cast<UT*>(pre + dstb[8]), (UT)to_fixed<T>(cv[8].x * scale));
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if constexpr (C::SPLIT) cv[q] = {exch_s[(2 * q) * 64 + t], exch_s[(2 * q + 1) * 64 + t]};
            atomicAdd(reinterpret_cast<UT*>(pre + dstb[q]), (UT)to_fixed<T>(cv[q].x * scale));
            atomicAdd(reinterpret_cast<UT*>(pim + dstb[q]), (UT)to_fixed<T>(cv[q].y * scale));
          }
          if (t == 0) {
            atomicAdd(reinterpret_cast<UT*>(pre + dstb[8]), (UT)to_fixed<T>(cv[8].x * scale));
            atomicAdd(reinterpret_cast<UT*>(pim + dstb[8]), (UT)to_fixed<T>(cv[8].y * scale));
          }
        }
      } else {
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          if (q == 8 && t != 0) break;
          if (!cur[f].valid) continue;
          const int k = t + L * q;
          if (ANYLEN && (q == 8 || k >= p.n_freqs)) continue;
          const cpx<T> zk = v[f][q], zn = zp[f][q];
          const cpx<T> S = {zk.x + zn.x, zk.y - zn.y};
          const cpx<T> dS = {zk.y + zn.y, zn.x - zk.x};
          const int o = k * PITCH + fl;
          if (p.out_kind == 1) {                     // SSQ_OUT_SX
            tile_re[o] = as_int<T>(S.x);
            tile_im[o] = as_int<T>(S.y);
          } else if (p.out_kind == 2) {              // SSQ_OUT_DSX
            tile_re[o] = as_int<T>(dS.x * p.inv_alpha);
            tile_im[o] = as_int<T>(dS.y * p.inv_alpha);
          } else {                                   // SSQ_OUT_WK
            T w;
            int kk;
            const bool keep = phase_bin<T>(p, k, S, dS, w, kk);
            tile_re[o] = as_int<T>(w);
            tile_im[o] = as_int<T>(keep ? (T)kk : (T)-1);
          }
        }
      }
    }

    SSQ_STAMP(6);
    if (ig == NG - 1 && !SSQ_ABL(64)) {
      __syncthreads();
      SSQ_STAMP(7);
      // ---- tile read-out: row segments of F frames, re-zeroing as we go ----
      // thread -> (fixed frame f, rows k0, k0 + RSTEP, ...): LDS offsets and the global row stride are
      // loop constants, so an element costs 2 reads + 2 zero-writes + convert + one 8-byte store
      {
        constexpr int NT = C::W * 64;
        constexpr int RSTEP = NT / F;                  // rows covered per sweep
        static_assert(NT % F == 0, "threads per block must be a multiple of F");
        const int f = tid % F;
        const int k0 = tid / F;
        cpx<T>* __restrict__ og =
            p.out + tl.sig * (long long)p.n_freqs * p.n_frames + tl.frame0 + f + (long long)k0 * p.n_frames;
        const long long gstep = (long long)RSTEP * p.n_frames;
        const bool fvalid = (tl.frame0 + f < p.n_frames) && !SSQ_ABL(32);
        const T sc = (TXONLY && !WKDBG) ? col_scale[f] : (T)1;
        IT* tr = tile_re + k0 * PITCH + f;
        IT* ti = tile_im + k0 * PITCH + f;
        constexpr int NFULL = NF / RSTEP;              // sweeps in which every thread has a row
        auto sweep = [&](int j, bool store) {
          const IT ire = tr[j * RSTEP * PITCH], iim = ti[j * RSTEP * PITCH];
          tr[j * RSTEP * PITCH] = 0;
          ti[j * RSTEP * PITCH] = 0;
          cpx<T> val;
          if constexpr (TXONLY && !WKDBG) val = {fixed_to_real<T>(ire) * sc, fixed_to_real<T>(iim) * sc};
          else val = {from_int<T>(ire), from_int<T>(iim)};
          if (store && (!ANYLEN || k0 + j * RSTEP < p.n_freqs)) og[j * gstep] = val;
        };
        if (fvalid) {
#pragma unroll 8
          for (int j = 0; j < NFULL; ++j) sweep(j, true);
          if (k0 + NFULL * RSTEP < NF) sweep(NFULL, true);
        } else {
#pragma unroll 8
          for (int j = 0; j < NFULL; ++j) sweep(j, false);
          if (k0 + NFULL * RSTEP < NF) sweep(NFULL, false);
        }
      }
      SSQ_STAMP(8);
      __syncthreads();
      SSQ_STAMP(9);
    }
#ifdef SSQ_STAMPS
    if (!has_next) {
      if (p.stamps && lane == 0)
        for (int i = 0; i < 12; ++i) p.stamps[((long long)blockIdx.x * C::W + wave) * 12 + i] = st_acc[i];
    }
#endif
    if (!has_next) break;
    i0 = i1;
    i1 = i2;
    cur[0] = fr1;
    fr1 = fr2;
  }
}

}  // namespace ssq
