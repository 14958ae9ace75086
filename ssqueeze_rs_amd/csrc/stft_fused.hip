// stft_fused.hip -- fused STFT / synchrosqueezed-STFT kernel for gfx950 (MI355X).
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
// quantisation step (2^-30 resp. 2^-62 of the column's L1 mass) sits below the FFT's own
// rounding error, and -- integer adds being associative -- the result is bitwise reproducible.
//
// The window tables arrive pre-multiplied by 1/2 (so the unpack needs no scaling) and the
// derivative channel by a power of two alpha (balances the two packed channels); alpha is
// folded into the 2*pi of the phase transform (StftDev::two_pi_eff).
#include <cstdlib>
#include <type_traits>
#include "fft_core.h"
#include "fft_pk1024.h"
#include "stft_kernels.h"

namespace ssq {

template <typename T, int LOGN>
struct FusedCfg {
  static constexpr int N = 1 << LOGN;
  static constexpr int L = N / 16;                         // lanes per frame
  static constexpr int W = (sizeof(T) == 4) ? 8 : 4;       // waves per block
  static constexpr int FPW = (L >= 64) ? 1 : 64 / L;       // frames per wave
  static constexpr int WPF = (L <= 64) ? 1 : L / 64;       // waves per frame
  static constexpr int FIF = W * FPW / WPF;                // frames in flight per block
  static constexpr int NF = N / 2 + 1;
  static constexpr int EXCH_ELEMS = N + N / 16;            // +1 element per 16: bank spread
  static constexpr int EXCH_BYTES = FIF * EXCH_ELEMS * (int)sizeof(cpx<T>);
  static constexpr bool WIN_LDS = (sizeof(T) == 4) && (N <= 1024);   // window table in LDS
  static constexpr int WIN_BYTES = WIN_LDS ? N * (int)sizeof(cpx<T>) : 0;
  static constexpr int TWL_BYTES = WIN_LDS ? N * (int)sizeof(cpx<T>) : 0;   // W_N table in LDS (paired-frame kernels)
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
  static constexpr int FRAC = (sizeof(T) == 4) ? 30 : 62;      // fixed-point fraction bits
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
  else return __double2ll_rn(v);
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

template <typename T, int LOGN, bool EDGE>
__device__ __forceinline__ FrameItem<T> make_frame(const StftDev<T>& p, const TileItem& tl, int it, int slot, int t) {
  using C = FusedCfg<T, LOGN>;
  FrameItem<T> w;
  w.fl = it * C::FIF + slot;
  const int frame = tl.frame0 + w.fl;
  w.valid = EDGE ? ((frame < p.n_frames) ? 1 : 0) : 1;
  w.xs = sig_base(p, tl.sig);
  w.pos0 = (long long)frame * p.hop - p.pad_left + t;
  return w;
}

template <typename T, int LOGN, bool EDGE, bool BLUE = false>
__device__ __forceinline__ void load_samples(const StftDev<T>& p, const TileItem& tl, const FrameItem<T>& w,
                                             T (&xv)[16], int t = 0) {
  constexpr int L = FusedCfg<T, LOGN>::L;
  if constexpr (BLUE) {
    // only the n_eff samples of the frame are read (the table's zero padding must not meet a NaN beyond it); they all
    // sit in q < 8 (n_eff <= (m + 1)/2), and a frame that lies inside the signal needs no mirroring logic
    const long long first = w.pos0 - t;
    const bool inside = first >= 0 && first + p.n_eff <= p.n_signal;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const bool live = w.valid && t + L * q < p.n_eff;
      if (inside) xv[q] = live ? w.xs[w.pos0 + L * q] : (T)0;
      else xv[q] = live ? load_padded(w.xs, w.pos0 + L * q, p.n_signal, p.padtype) : (T)0;
    }
#pragma unroll
    for (int q = 8; q < 16; ++q) xv[q] = (T)0;
  } else if constexpr (!EDGE) {
#pragma unroll
    for (int q = 0; q < 16; ++q) xv[q] = w.xs[w.pos0 + L * q];
  } else {
#pragma unroll
    for (int q = 0; q < 16; ++q)
      xv[q] = w.valid ? load_padded(w.xs, w.pos0 + L * q, p.n_signal, p.padtype) : (T)0;
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
template <typename T, int LOGN, bool TXONLY, bool EDGE, bool LEB, bool WKDBG = false, bool BLUE = false>
__global__ __launch_bounds__((FusedCfg<T, LOGN>::W * 64)) void stft_fused_kernel(StftDev<T> p) {
  using C = FusedCfg<T, LOGN>;
  constexpr int N = C::N, L = C::L, NF = C::NF, F = C::F, PITCH = C::PITCH;
  constexpr bool MULTIWAVE = (C::WPF > 1);
  static_assert(!BLUE || EDGE, "Bluestein mode runs the edge-capable loader (masked samples)");
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
  const int wave = tid >> 6;
  int slot, t;
  if constexpr (L <= 64) {
    slot = wave * C::FPW + lane / L;
    t = lane % L;
  } else {
    slot = wave / C::WPF;
    t = (wave % C::WPF) * 64 + lane;
  }
  cpx<T>* exch = exch_all + slot * C::EXCH_ELEMS;

  // ---- per-lane constants, live across all tiles this block processes ----
  constexpr bool TW_REGS = C::TW_REGS && !PAIR;   // paired frames: registers go to the second frame, twiddles to LDS
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
  if constexpr (PAIR) {
    for (int i = tid; i < N; i += C::W * 64) tw_lds[i] = p.tw[i];
  }
  const cpx<T>* tw_src = PAIR ? tw_lds : p.tw;
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
  auto advance = [&](const Item& c) {
    Item n = c;
    n.ig = c.ig + 1;
    if (n.ig == NG) {
      n.ig = 0;
      long long ns = c.tl.sig;
      int nft = c.tl.ft + (int)gridDim.x;
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
  FrameItem<T> cur[NFW];
  T xn[NFW][16];      // samples of the current item
  T xb[16];           // samples of the next item
  cur[0] = make_frame<T, LOGN, EDGE>(p, i0.tl, i0.ig, slot, t);
  load_samples<T, LOGN, EDGE, BLUE>(p, i0.tl, cur[0], xn[0], t);
  FrameItem<T> fr1 = cur[0];
  if (i1.ok) {
    fr1 = make_frame<T, LOGN, EDGE>(p, i1.tl, i1.ig, slot, t);
    load_samples<T, LOGN, EDGE, BLUE>(p, i1.tl, fr1, xb, t);
  }
#ifdef SSQ_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_prev = ssq_stamp();
#endif

#pragma unroll 1
  while (true) {
    const TileItem tl = i0.tl;
    const int ig = i0.ig;
    // ---- window multiply ----
    cpx<T> v[NFW][16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const cpx<T> wq = C::WIN_LDS ? win_lds[t + L * q] : p.win2[t + L * q];
      v[0][q] = {xn[0][q] * wq.x, xn[0][q] * wq.y};
    }
    SSQ_STAMP(0);
    const bool has_next = i1.ok;
    // rotate the prefetch ring: the next item's samples (loaded one iteration ago) move to xn and the
    // loads of the item after next go out now, a full iteration before they are needed
#pragma unroll
    for (int q = 0; q < 16; ++q) xn[0][q] = xb[q];
    const Item i2 = advance(i1);
    FrameItem<T> fr2 = fr1;
    if (i2.ok) {
      fr2 = make_frame<T, LOGN, EDGE>(p, i2.tl, i2.ig, slot, t);
      if (!SSQ_ABL(1)) load_samples<T, LOGN, EDGE, BLUE>(p, i2.tl, fr2, xb, t);
    }

    SSQ_STAMP(1);
    if (!SSQ_ABL(2)) {
      if constexpr (PAIR) fft_pass_pair<T, LOGN, 0, false, false>(v[0], v[1], exch, twr, tw_src, t);
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
      if constexpr (BLUE) {
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
            if constexpr (BLUE) m *= (q < 8 && t + L * q < p.n_freqs) ? 1.0f : 0.0f;   // bins of the n_eff-point transform
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
            const cpx<T> zk = v[f][q], zn = zp[f][q];
            const cpx<T> S = {zk.x + zn.x, zk.y - zn.y};
            const cpx<T> dS = {zk.y + zn.y, zn.x - zk.x};       // alpha * dSx
            int kk = k;
            T w;
            bool keep = phase_bin<T>(p, k, S, dS, w, kk);
            keep = keep && cur[f].valid && (q < 8 || t == 0);
            if constexpr (BLUE) keep = keep && q < 8 && k < p.n_freqs;
            cpx<T> c = LEB ? cpx<T>{p.leb_unit, (T)0} : S;   // weight  (ssq_stft.rs:292-296)
            c.x = keep ? c.x : (T)0;
            c.y = keep ? c.y : (T)0;
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
            if ((q < 8 || t == 0) && cur[f].valid && (!BLUE || (q < 8 && t + L * q < p.n_freqs))) {
              const int o = (t + L * q) * PITCH + fl;
              tile_re[o] = as_int<T>(wdbg[q]);
              tile_im[o] = as_int<T>((T)kdbg[q]);
            }
          }
        } else if constexpr (LEB) {
#pragma unroll
          for (int q = 0; q < 8; ++q) atomicAdd(reinterpret_cast<UT*>(pre + dstb[q]), (UT)to_fixed<T>(cv[q].x * scale));
          if (t == 0) atomicAdd(reinterpret_cast<UT*>(pre + dstb[8]), (UT)to_fixed<T>(cv[8].x * scale));
        } else {
#pragma unroll
          for (int q = 0; q < 8; ++q) {
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
          if (BLUE && (q == 8 || k >= p.n_freqs)) continue;
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
          if constexpr (TXONLY && !WKDBG) val = {(T)ire * sc, (T)iim * sc};
          else val = {from_int<T>(ire), from_int<T>(iim)};
          if (store && (!BLUE || k0 + j * RSTEP < p.n_freqs)) og[j * gstep] = val;
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

// ---------------------------------------------------------------------------------------------
// High-occupancy variant for fp32, n_fft = 1024, SSQ_OUT_TX: 16 waves per CU (4 per SIMD) instead of 8.
// The 8-wave kernel is bound by exposed latency (two waves per SIMD cannot cover the LDS round trips);
// to fit 16 waves the per-wave LDS row shrinks to HALF a frame and the registers to <= 128:
//   * exchange 1 goes through the half-size row in two phases (lanes 0-31 write, all read their first
//     8 values; lanes 32-63 write, all read the other 8) -- the DS unit runs a wave's ops in order;
//   * exchange 2 is a 4x4 transpose between the four 16-lane rows and the low two bits of the register
//     index: v_permlane32_swap + v_permlane16_swap, no LDS;
//   * twiddles come from an LDS copy of the W_1024 table, samples are prefetched one tile ahead.
// One tile = one frame per wave, so the tile barrier comes once per frame per wave.
// ---------------------------------------------------------------------------------------------
#ifndef SSQ_TX_CELL64
#define SSQ_TX_CELL64 1     // Tx tile of the 16-wave kernel as 64-bit (re, im) cells: one ds_add_u64 per bin
#endif
#ifndef SSQ_TX_EXPAD
#define SSQ_TX_EXPAD 1     // exchange-row padding per 16 elements (2 = conflict-free 16-element writes: measured neutral)
#endif
#ifndef SSQ_HIOCC_DEFAULT
#define SSQ_HIOCC_DEFAULT 1
#endif
#ifndef SSQ_XHALF
#define SSQ_XHALF 1                // 1: exchange 1 of the 16-wave kernel by register halves (full-width LDS stores)
#endif
#ifndef SSQ_PRIO
#define SSQ_PRIO 0                 // s_setprio experiments: bit 0 = raise around the scatter, bit 1 = raise in the read-out
#endif
#ifndef SSQ_LATE_PREFETCH
#define SSQ_LATE_PREFETCH 1        // 1: issue the next tile's sample loads after exchange 1 (keeps the kernel out of scratch)
#endif
#ifndef SSQ_DPP_FUSE
#define SSQ_DPP_FUSE 0             // 1: lane-pair merge sums as v_add_u32_dpp (inline asm) instead of v_mov_dpp + v_add
#endif
#ifndef SSQ_T0_ROTATE
#define SSQ_T0_ROTATE 1            // 1: lane 0's self-partner bins by a masked register rotation instead of selects
                                   // (re-fetching them by 16 one-lane ds_bpermute instead: +17 %, profiles/r02_ab_libs4.txt)
#endif
#ifndef SSQ_TX_BIAS
#define SSQ_TX_BIAS 1              // 1: the 64-bit Tx cells of the 16-wave kernel carry RE + 2^31 (no borrow to undo at the read-out)
#endif
#ifndef SSQ_NT_STORE
#define SSQ_NT_STORE 1             // 1: nontemporal Tx stores in the paired read-out (-0.7 % on the bench shape, profiles/r02_ab_nt.txt)
#endif
#ifndef SSQ_RO_PAIR
#define SSQ_RO_PAIR 1              // 1: read-out of the interior 16-wave kernel with 16-byte stores (two frames per thread)
#endif
#ifndef SSQ_PK
#define SSQ_PK 0                   // 1: the 16-wave kernel's FFT on packed fp32 (fft_pk1024.h): 25 % fewer vector instructions,
                                   // SAME time (-1.5 %; VALU-only floor 1.52 vs 1.46 ms): two waves already share the SIMD's 32
                                   // lanes for scalar fp32 add/mul/fma, a packed op takes the slot of two (profiles/r03_ab_pk.txt)
#endif
#ifndef SSQ_TX_MERGE
#define SSQ_TX_MERGE 0      // (measured: a net loss since the read-out/exchange rework, profiles/r02_ab_libs2.txt) merge the contributions of lane pairs with equal destinations before the LDS atomic
#endif

// WAVES = 16: one block per CU.  WAVES = 8: two independent blocks per CU (tile of 8 frames, exchange 1 through a
// quarter-size row in four phases), whose barriers are not coupled, so that one block's LDS-bound phases can
// overlap the other's VALU-bound ones.
template <int WAVES>
struct Hi1024 {
  static constexpr int N = 1024, L = 64, NF = 513, W = WAVES, F = WAVES, PITCH = F + 1, THREADS = WAVES * 64;
  static constexpr int PLANE = NF * PITCH;
  static constexpr int TILE_BYTES = (((2 * PLANE + F) * 4 + 15) / 16) * 16;
  static constexpr int NPH = 32 / WAVES;                    // exchange-1 phases: 2 (half rows) or 4 (quarter rows)
  static constexpr int EXH_PAD = SSQ_TX_EXPAD;              // pad elements per 16: 2 makes a lane's 16-element write (stride 36 dwords) conflict-free
  // the register-half exchange (WAVES = 16) lays a half row out as 64 writers x (8 values + 1 pad)
  static constexpr int EXH_ELEMS = (WAVES == 16) ? 64 * 9 : N / NPH + (N / NPH / 16) * EXH_PAD;
  static constexpr int EXH_BYTES = W * EXH_ELEMS * 8;
  static constexpr int TAB_BYTES = N * 8;                   // window table; twiddle tables [16][16] + [3][256] (+pad)
  static constexpr int LDS_BYTES = TILE_BYTES + EXH_BYTES + 2 * TAB_BYTES;
  static constexpr int FRAC = 30, EMIN = -90;
  static_assert(LDS_BYTES * (16 / WAVES) <= 160 * 1024, "LDS budget");
};

// 4x4 transpose of R[0..3] across the four 16-lane rows of the wave (one dword per lane per register)
__device__ __forceinline__ void rows_transpose4(float& r0, float& r1, float& r2, float& r3) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(r0), __float_as_uint(r2), false, false);
  auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(r1), __float_as_uint(r3), false, false);
  auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
  auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
  r0 = __uint_as_float(c[0]);
  r1 = __uint_as_float(c[1]);
  r2 = __uint_as_float(d[0]);
  r3 = __uint_as_float(d[1]);
}

// (Measured and removed in round 2: replacing the two tile barriers by arrival counters in LDS, with the read-out of
// tile i-1 placed inside tile i's FFT or at the loop top, ran 10-14 % SLOWER -- profiles/r02_ab_freerun.txt.)
template <bool EDGE, bool LEB, int WAVES, bool WKDBG = false>
__global__ __launch_bounds__(WAVES * 64, 16 / WAVES) void stft_tx1024_kernel(StftDev<float> p) {
  using H = Hi1024<WAVES>;
  constexpr int THREADS = H::THREADS;
  using T = float;
  constexpr int N = H::N, L = H::L, NF = H::NF, F = H::F, PITCH = H::PITCH;
  __shared__ __attribute__((aligned(16))) unsigned char smem[H::LDS_BYTES];
  int* tile_re = reinterpret_cast<int*>(smem);
  int* tile_im = tile_re + H::PLANE;
  float* col_scale = reinterpret_cast<float*>(tile_im + H::PLANE);
  cpx<T>* exch_all = reinterpret_cast<cpx<T>*>(smem + H::TILE_BYTES);
  cpx<T>* win_lds = reinterpret_cast<cpx<T>*>(smem + H::TILE_BYTES + H::EXH_BYTES);
  cpx<T>* tw1 = win_lds + N;        // pass 1: [m = 0..15][k = 0..15]   W_256^(k m)
  cpx<T>* tw2 = tw1 + 256;          // pass 2: [m = 0..2][j = 0..255]   W_1024^(j (m+1))  (row m+1 of the compact layout)

  const int tid = threadIdx.x;
  const int t = tid & 63;          // lane = position inside the frame
  const int fl = tid >> 6;         // wave = frame inside the tile
  cpx<T>* exch = exch_all + fl * H::EXH_ELEMS;
  auto xphys = [](int i) { return i + H::EXH_PAD * (i >> 4); };

  for (int i = tid; i < N; i += THREADS) win_lds[i] = p.win2[i];
  if (tid < 256) tw1[tid] = p.tw[((tid & 15) * (tid >> 4) * 4) & (N - 1)];
  for (int i = tid; i < 768; i += THREADS) tw2[i] = p.tw[((i & 255) * ((i >> 8) + 1)) & (N - 1)];
  // 64-bit cells start at RE = 2^31 (SSQ_TX_BIAS): RE + 2^31 stays in [0, 2^32), so no borrow ever reaches the high word
  // and the read-out takes IM = high word, RE = low word ^ 2^31 -- one instruction less per cell than undoing a borrow
  constexpr long long CELL0 = (SSQ_TX_CELL64 && SSQ_TX_BIAS && !WKDBG) ? 0x80000000LL : 0LL;
  if constexpr (SSQ_TX_CELL64) {
    for (int i = tid; i < H::PLANE; i += THREADS) reinterpret_cast<long long*>(tile_re)[i] = CELL0;
  } else {
    for (int i = tid; i < 2 * H::PLANE; i += THREADS) tile_re[i] = 0;
  }
  __syncthreads();
  // 8-wave variant: tiles 2i and 2i+1 hold the two 64-byte halves of the same output lines; blocks b and b + 8 run
  // on the same XCD (round-robin dispatch), so give THEM the adjacent tiles and let the halves meet in one L2
  unsigned bid = blockIdx.x;
  if (WAVES == 8 && gridDim.x % 16 == 0) bid = (bid / 16) * 16 + (bid % 8) * 2 + ((bid / 8) % 2);
  if ((long long)bid >= p.total_tiles) return;

  const long long n_sig = p.total_tiles / p.tiles_per_signal;
  long long sig = (long long)(bid / (unsigned)p.tiles_per_signal);
  int jt = (int)(bid % (unsigned)p.tiles_per_signal);
  auto tile_frame0 = [&](int j) { return ((j < p.ta_n) ? p.ta0 + j : p.tb0 + (j - p.ta_n)) * F; };
  auto load_frame = [&](long long sg, int frame0, T (&xv)[16]) {
    const int frame = frame0 + fl;
    const T* xs = sig_base(p, sg);
    const long long pos0 = (long long)frame * p.hop - p.pad_left + t;
    if constexpr (!EDGE) {
#pragma unroll
      for (int q = 0; q < 16; ++q) xv[q] = xs[pos0 + L * q];
    } else {
      const bool valid = frame < p.n_frames;
#pragma unroll
      for (int q = 0; q < 16; ++q) xv[q] = valid ? load_padded(xs, pos0 + L * q, p.n_signal, p.padtype) : 0.0f;
    }
  };
  T xn[16];
  load_frame(sig, tile_frame0(jt), xn);
  const cpx<T> twr_unused[3][16] = {};

  // ---- tile read-out: thread -> (frame f, rows k0 + 64 j); re-zeroes what it reads ----
  auto read_out = [&](long long rsig, int rframe0) {
    if (SSQ_ABL(8)) return;
#if SSQ_RO_PAIR && SSQ_TX_CELL64
    if constexpr (!EDGE && !WKDBG && WAVES == 16) {
      // thread -> (frame pair fp, rows k0 + 128 j): two adjacent cells per thread, ONE 16-byte store per row
      // (half as many store instructions; T21 of the programming guide).  Needs even n_frames for the alignment.
      if ((p.n_frames & 1) == 0) {
        constexpr int RS2 = THREADS / (F / 2);              // 128 rows per sweep
        const int fp = tid % (F / 2);
        const int k0 = tid / (F / 2);
        float4* __restrict__ og4 = reinterpret_cast<float4*>(p.out + rsig * (long long)NF * p.n_frames + rframe0 + 2 * fp +
                                                             (long long)k0 * p.n_frames);
        const long long gstep4 = (long long)RS2 * p.n_frames / 2;     // in float4 units
        // (round 3: a wave-uniform base + 32-bit per-thread offset does not make the compiler take the SGPR-base store form --
        //  loop strength reduction rebuilds the 64-bit vector address chain either way)
        const T sc0 = col_scale[2 * fp], sc1 = col_scale[2 * fp + 1];
        long long* tc = reinterpret_cast<long long*>(tile_re) + k0 * PITCH + 2 * fp;
        auto sweep2 = [&](int j) {
          const long long c0 = tc[j * RS2 * PITCH], c1 = tc[j * RS2 * PITCH + 1];
          tc[j * RS2 * PITCH] = CELL0;
          tc[j * RS2 * PITCH + 1] = CELL0;
#if SSQ_TX_BIAS
          const int r0 = (int)c0 ^ (int)0x80000000, r1 = (int)c1 ^ (int)0x80000000;
          const int i0 = (int)(c0 >> 32), i1 = (int)(c1 >> 32);
#else
          const int r0 = (int)c0, r1 = (int)c1;
          const int i0 = (int)(c0 >> 32) - (r0 >> 31), i1 = (int)(c1 >> 32) - (r1 >> 31);
#endif
          if (!SSQ_ABL(4)) {
            const float4 val = make_float4((T)r0 * sc0, (T)i0 * sc0, (T)r1 * sc1, (T)i1 * sc1);
#if SSQ_NT_STORE
            typedef float vf4 __attribute__((ext_vector_type(4)));
            const vf4 nv = {val.x, val.y, val.z, val.w};
            __builtin_nontemporal_store(nv, reinterpret_cast<vf4*>(&og4[j * gstep4]));   // Tx is written once, never read back here
#else
            og4[j * gstep4] = val;
#endif
          }
        };
        constexpr int NFULL2 = NF / RS2;                      // 4 full sweeps
#pragma unroll
        for (int j = 0; j < NFULL2; ++j) sweep2(j);
        if (k0 + NFULL2 * RS2 < NF) sweep2(NFULL2);
        return;
      }
    }
#endif
    constexpr int RSTEP = THREADS / F;                    // 64 rows per sweep
    const int f = tid % F;
    const int k0 = tid / F;
    cpx<T>* __restrict__ og = p.out + rsig * (long long)NF * p.n_frames + rframe0 + f + (long long)k0 * p.n_frames;
    const long long gstep = (long long)RSTEP * p.n_frames;
    const bool fvalid = (EDGE ? (rframe0 + f < p.n_frames) : true) && !SSQ_ABL(4);
    const T sc = col_scale[f];
    constexpr int NFULL = NF / RSTEP;                     // 8 full sweeps
#if SSQ_TX_CELL64
    long long* tc = reinterpret_cast<long long*>(tile_re) + k0 * PITCH + f;
    auto sweep = [&](int j) {
      const long long c = tc[j * RSTEP * PITCH];
      tc[j * RSTEP * PITCH] = CELL0;
      if constexpr (WKDBG) {
        if (fvalid) og[j * gstep] = cpx<T>{__int_as_float((int)c), __int_as_float((int)(c >> 32))};
        return;
      }
#if SSQ_TX_BIAS
      const int ire = (int)c ^ (int)0x80000000;
      const int iim = (int)(c >> 32);
#else
      const int ire = (int)c;
      const int iim = (int)(c >> 32) - (ire >> 31);
#endif
      if (fvalid) og[j * gstep] = cpx<T>{(T)ire * sc, (T)iim * sc};
    };
#else
    int* tr = tile_re + k0 * PITCH + f;
    int* ti = tile_im + k0 * PITCH + f;
    auto sweep = [&](int j) {
      const int ire = tr[j * RSTEP * PITCH], iim = ti[j * RSTEP * PITCH];
      tr[j * RSTEP * PITCH] = 0;
      ti[j * RSTEP * PITCH] = 0;
      if (fvalid) og[j * gstep] = cpx<T>{(T)ire * sc, (T)iim * sc};
    };
#endif
#pragma unroll
    for (int j = 0; j < NFULL; ++j) sweep(j);
    if (k0 + NFULL * RSTEP < NF) sweep(NFULL);
  };

#ifdef SSQ_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_prev = ssq_stamp();
#endif
  // (ablation bit 0x100, timing only / racy: ONE barrier per tile and an ASYMMETRIC wave schedule -- the waves with
  // (fl & ((ablate >> 12) & 15)) != 0 read the previous tile out BEFORE their frame, the others AFTER their scatter, so
  // that every SIMD has LDS/store-bound and VALU-bound waves at the same time; prices a double-buffered tile)
  const bool asym = SSQ_ABL(0x100);
  const bool asym_early = asym && ((fl & ((p.ablate >> 12) & 15)) != 0);
  bool have_prev = false;
  long long psig = 0;
  int pframe0 = 0;
#pragma unroll 1
  while (true) {
    const int frame0 = tile_frame0(jt);
    const bool valid = EDGE ? (frame0 + fl < p.n_frames) : true;
    if (asym_early && have_prev) read_out(psig, pframe0);
    // next tile of this block; prefetch its samples behind this frame's FFT
    long long nsig = sig;
    int njt = jt + (int)gridDim.x;
    while (njt >= p.tiles_per_signal) {
      njt -= p.tiles_per_signal;
      ++nsig;
    }
    const bool has_next = nsig < n_sig;
    cpx<T> v[16];
#if SSQ_PK
    constexpr bool kPk = (WAVES == 16);
#else
    constexpr bool kPk = false;
#endif
    if constexpr (kPk) {
      // ---- the whole transform on packed fp32 (fft_pk1024.h); same passes, exchanges and tables as below ----
      using pk::v2f;
      v2f pv[16];
      const v2f* win2v = reinterpret_cast<const v2f*>(win_lds);
#pragma unroll
      for (int q = 0; q < 16; ++q) pv[q] = win2v[t + L * q] * xn[q];
      SSQ_STAMP(0);
      SSQ_STAMP(1);
      pk::dft16(pv);
      SSQ_STAMP(10);
      {
        v2f* ex2 = reinterpret_cast<v2f*>(exch);
        v2f nv[16];
        const int rbase = 9 * (t >> 4) + (t & 7);
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
          for (int u = 0; u < 8; ++u) ex2[9 * t + u] = pv[8 * ph + u];
          frame_sync<false>();
          if (((t >> 3) & 1) == ph) {
#pragma unroll
            for (int q = 0; q < 16; ++q) nv[q] = ex2[rbase + 36 * q];
          }
          frame_sync<false>();
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) pv[q] = nv[q];
      }
      SSQ_STAMP(11);
      if (has_next && !SSQ_ABL(32)) load_frame(nsig, tile_frame0(njt), xn);
      pk::pass1(pv, reinterpret_cast<const v2f*>(tw1), t);
      SSQ_STAMP(12);
#pragma unroll
      for (int uh = 0; uh < 4; ++uh) {
        float x0 = pv[4 * uh].x, x1 = pv[4 * uh + 1].x, x2 = pv[4 * uh + 2].x, x3 = pv[4 * uh + 3].x;
        float y0 = pv[4 * uh].y, y1 = pv[4 * uh + 1].y, y2 = pv[4 * uh + 2].y, y3 = pv[4 * uh + 3].y;
        rows_transpose4(x0, x1, x2, x3);
        rows_transpose4(y0, y1, y2, y3);
        pv[4 * uh] = v2f{x0, y0};
        pv[4 * uh + 1] = v2f{x1, y1};
        pv[4 * uh + 2] = v2f{x2, y2};
        pv[4 * uh + 3] = v2f{x3, y3};
      }
      {
#define SSQ_SWAP(i, j)     \
  {                        \
    const v2f t_ = pv[i];  \
    pv[i] = pv[j];         \
    pv[j] = t_;            \
  }
        SSQ_SWAP(1, 4) SSQ_SWAP(2, 8) SSQ_SWAP(3, 12) SSQ_SWAP(6, 9) SSQ_SWAP(7, 13) SSQ_SWAP(11, 14)
#undef SSQ_SWAP
      }
      SSQ_STAMP(13);
      pk::pass2(pv, reinterpret_cast<const v2f*>(tw2), t);
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = cpx<T>{pv[q].x, pv[q].y};
    } else {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const cpx<T> wq = win_lds[t + L * q];
      v[q] = {xn[q] * wq.x, xn[q] * wq.y};
    }
    SSQ_STAMP(0);
#if !SSQ_LATE_PREFETCH
    if (has_next && !SSQ_ABL(32)) load_frame(nsig, tile_frame0(njt), xn);
#endif
    SSQ_STAMP(1);

    // ---- pass 0: radix 16 over elements t + 64q ----
    fft_compute<T, 10, 0, false, false>(v, twr_unused, tw1, t);
    // ---- exchange 1 through the 1/NPH-size row: in phase ph the lanes [64 ph / NPH, 64 (ph+1) / NPH) write their
    //      16 values (elements 16 t + u) and every lane reads back its elements t + 64 q, q in [16 ph / NPH, ...) ----
    if (WAVES == 16 && SSQ_XHALF) {
      // exchange 1 by REGISTER halves: in phase ph EVERY lane writes its values u = 8 ph .. 8 ph + 7 (elements
      // 16 t + u) -- full-width stores, half as many store instructions as the lane-half scheme -- and the lanes whose
      // element residue (t & 15) lies in that half read all 16 of their elements t + 64 q = 16 ((t >> 4) + 4 q) + (t & 15).
      // Row layout: writer lane t' at 9 t' + (u & 7): pitch 9 elements = 18 dwords keeps both the 16-lane store groups
      // and the 32-lane load groups on distinct banks.
      cpx<T> nv[16];
      const int rbase = 9 * (t >> 4) + (t & 7);
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
#pragma unroll
        for (int u = 0; u < 8; ++u) exch[9 * t + u] = v[8 * ph + u];
        frame_sync<false>();
        if (((t >> 3) & 1) == ph) {
#pragma unroll
          for (int q = 0; q < 16; ++q) nv[q] = exch[rbase + 36 * q];
        }
        frame_sync<false>();
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = nv[q];
    } else if (!SSQ_ABL(1)) {
      constexpr int NPH = H::NPH, LPP = 64 / NPH, QPP = 16 / NPH;
      cpx<T> nv[16];
#pragma unroll
      for (int ph = 0; ph < NPH; ++ph) {
        if (t >= ph * LPP && t < (ph + 1) * LPP) {
#pragma unroll
          for (int u = 0; u < 16; ++u) exch[xphys(16 * (t - ph * LPP) + u)] = v[u];
        }
        frame_sync<false>();
#pragma unroll
        for (int q = 0; q < QPP; ++q) nv[ph * QPP + q] = exch[xphys(t + L * q)];
        frame_sync<false>();
      }
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = nv[q];
    }
#if SSQ_LATE_PREFETCH
    // the next tile's samples: issued only now, after exchange 1 -- during the exchange both the old and the new
    // register set of the frame are live, and 16 more registers in flight there push the kernel into scratch; the rest
    // of this tile (pass 1, pass 2, epilogue, read-out: > 10k cycles) still covers the HBM latency many times over
    if (has_next && !SSQ_ABL(32)) load_frame(nsig, tile_frame0(njt), xn);
#endif
    // ---- pass 1: twiddle W_256^(k m), radix 16 ----
    fft_compute<T, 10, 1, false, false, true>(v, twr_unused, tw1, t);
#ifdef SSQ_SENS
    {   // resource sensitivity: p.ablate = extra VALU instructions | extra LDS reads << 16 per frame (results unused)
      const int nv = p.ablate & 0xffff, nl = (p.ablate >> 16) & 0xffff;
      float d0 = v[0].x, d1 = v[1].x, d2 = v[2].x, d3 = v[3].x, d4 = v[4].x, d5 = v[5].x, d6 = v[6].x, d7 = v[7].x;
      for (int i = 0; i < nv; i += 8)
        asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                     "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7));
      for (int i = 0; i < nl; ++i) {
        const float r = reinterpret_cast<volatile float*>(exch)[2 * xphys(t + L * (i & 7))];
        d0 += r;
      }
      if (d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 == 12345.678f) v[0].x = 0.0f;
    }
#endif
    // ---- exchange 2: producer (row m, k), reg u = 4 uh + ul  ->  consumer (row ul, k), reg 4 m + uh ----
    {
#pragma unroll
      for (int uh = 0; uh < 4; ++uh) {
        rows_transpose4(v[4 * uh + 0].x, v[4 * uh + 1].x, v[4 * uh + 2].x, v[4 * uh + 3].x);
        rows_transpose4(v[4 * uh + 0].y, v[4 * uh + 1].y, v[4 * uh + 2].y, v[4 * uh + 3].y);
      }
      // slot 4*uh + a now holds consumer register q = 4*a + uh: transpose the register indices
#define SSQ_SWAP(i, j)       \
  {                          \
    const cpx<T> t_ = v[i];  \
    v[i] = v[j];             \
    v[j] = t_;               \
  }
      SSQ_SWAP(1, 4) SSQ_SWAP(2, 8) SSQ_SWAP(3, 12) SSQ_SWAP(6, 9) SSQ_SWAP(7, 13) SSQ_SWAP(11, 14)
#undef SSQ_SWAP
    }
    // ---- pass 2: twiddle W_1024^((t + 64 b) m), four radix-4 butterflies ----
    fft_compute<T, 10, 2, false, false, true>(v, twr_unused, tw2 - 256, t);   // compact index m*256 + j, m = 1..3
    }
    // lane t now holds Z[t + 64 q]

    SSQ_STAMP(2);
    // ---- partner Z[N-k] for the bins this lane owns ----
    cpx<T> zp[9];
    {
      const int src = (L - t) & (L - 1);
#if SSQ_T0_ROTATE
      // lane 0 pairs with ITSELF one register up (N - 64 q = 64 (16 - q)): rotate its upper registers once (16 moves
      // under a one-lane mask) instead of 16 selects; nobody else reads lane 0's upper half (src == 0 only for t == 0)
      zp[8] = v[8];
      if (t == 0) {
#pragma unroll
        for (int j = 8; j < 15; ++j) v[j] = v[j + 1];
        v[15] = v[0];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        zp[q].x = __shfl(v[15 - q].x, src);
        zp[q].y = __shfl(v[15 - q].y, src);
      }
      v[8] = zp[8];
#else
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        cpx<T> r;
        if (SSQ_ABL(16)) {
          r = v[15 - q];
        } else {
          r.x = __shfl(v[15 - q].x, src);
          r.y = __shfl(v[15 - q].y, src);
        }
        if (t == 0) r = (q == 0) ? v[0] : v[16 - q];
        zp[q] = r;
      }
      zp[8] = v[8];
#endif
    }

    SSQ_STAMP(3);
    // ---- unpack, phase transform, bin index, fixed-point scatter (same arithmetic as stft_fused_kernel) ----
    {
      cpx<T> cv[9];
      int dstb[9];
      T wdbg[9];
      int kdbg[9];
      T l1 = 0.0f;
      constexpr int CELL = SSQ_TX_CELL64 ? 8 : 4;            // bytes per tile cell (interleaved re,im) or plane element
      const float lane_on = valid ? 1.0f : 0.0f;
      const float sfs0 = (float)t * p.sfs_step, sfs_q = (float)L * p.sfs_step;
      const int neg_last = -(p.n_freqs - 1);
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const cpx<T> zk = v[q], zn = zp[q];
        const cpx<T> S = {zk.x + zn.x, zk.y - zn.y};
        const cpx<T> dS = {zk.y + zn.y, zn.x - zk.x};
        const float den = S.x * S.x + S.y * S.y;
        const float num = dS.y * S.x - dS.x * S.y;
        const float pd = num * __builtin_amdgcn_rcpf(den * p.two_pi_eff);
        // d = Sfs - pd; w = |d| (ssq_stft.rs:33) is only ever used through modifiers: the finiteness mask ignores the sign and
        // the bin fma takes -|d| (round 3: nine v_and per frame less -- those do not pair with another wave's instruction)
        const float d = (sfs0 + (float)q * sfs_q) - pd;
        const float w = fabsf(d);
        float m = fma_clamp01(den, p.keep_big, p.keep_bias) * fma_clamp01(d, 0.0f, 1.0f);
        if (EDGE) m *= lane_on;
        if (q == 8) m *= (t == 0) ? 1.0f : 0.0f;
        const cpx<T> c = LEB ? cpx<T>{p.leb_unit * m, 0.0f} : cpx<T>{S.x * m, S.y * m};
        cv[q] = c;
        int kneg = cvt_floor_i32(__builtin_fmaf(-w, p.inv_dw, 0.5f));
        kneg = kneg < neg_last ? neg_last : kneg;
        dstb[q] = __mul24(kneg, -(PITCH * CELL)) + fl * CELL;
        l1 += fabsf(c.x) + fabsf(c.y);
        if constexpr (WKDBG) {
          wdbg[q] = w;
          kdbg[q] = (m != 0.0f) ? -kneg : -1;
        }
      }
      SSQ_STAMP(4);
      const T tot = frame_allreduce<T, L, false>(l1, t, nullptr, t) * p.dw;
      T scale, inv_scale;
      column_scale<T, H::FRAC, H::EMIN>(tot, p.dw, scale, inv_scale);
      if (t == 0 && valid) col_scale[fl] = inv_scale;
      SSQ_STAMP(5);
#if SSQ_PRIO & 1
      __builtin_amdgcn_s_setprio(1);               // the short LDS-bound tail of a frame goes first
#endif
      // fixed-point contributions; scatter one 64-bit add per bin into the (re, im) cell: the cell holds the signed
      // integer IM * 2^32 + RE (|RE| < 2^31), so a borrow of a negative RE into the high word is undone exactly at
      // the read-out (IM = high - (RE >> 31)) whatever the order of the adds
      char* ptile = reinterpret_cast<char*>(tile_re);
      const bool odd_lane = (t & 1) != 0;
      if constexpr (WKDBG) {
        static_assert(!WKDBG || SSQ_TX_CELL64, "the (w, k) hook uses the 64-bit cells");
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          if ((q < 8 || t == 0) && valid) {
            const unsigned lo = (unsigned)__float_as_int(wdbg[q]), hi = (unsigned)__float_as_int((float)kdbg[q]);
            reinterpret_cast<unsigned long long*>(ptile)[(t + L * q) * PITCH + fl] = ((unsigned long long)hi << 32) | lo;
          }
        }
      } else {
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        int ia = cvt_round_i32(cv[q].x * scale);
        int ib = LEB ? 0 : cvt_round_i32(cv[q].y * scale);
        bool skip = (q == 8) && (t != 0);
#if SSQ_TX_MERGE
        if (q < 8) {
          // neighbouring bins are often reassigned to the same row: lanes (2i, 2i+1) with equal destinations merge
          // their (integer, hence order-exact) contributions into one add -- same-address LDS atomics serialise
          const int ksw = __builtin_amdgcn_update_dpp(0, dstb[q], 0xB1, 0xF, 0xF, true);     // quad_perm [1,0,3,2]
#if SSQ_DPP_FUSE
          int sa, sb = 0;                          // own + neighbour in ONE instruction (the DPP operand rides on the add)
          asm("v_add_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(sa) : "v"(ia));
          if (!LEB) asm("v_add_u32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(sb) : "v"(ib));
#else
          const int sa = ia + __builtin_amdgcn_update_dpp(0, ia, 0xB1, 0xF, 0xF, true);
          const int sb = LEB ? 0 : ib + __builtin_amdgcn_update_dpp(0, ib, 0xB1, 0xF, 0xF, true);
#endif
          const bool same = (ksw == dstb[q]);
          ia = same ? sa : ia;
          ib = same ? sb : ib;
          skip = same && odd_lane;
        }
#endif
        if (SSQ_ABL(2)) {
          asm volatile("" ::"v"(ia), "v"(ib), "v"(dstb[q]));
          skip = true;
        }
        if (!skip) {
#if SSQ_TX_CELL64
          if (LEB) {
            atomicAdd(reinterpret_cast<unsigned*>(ptile + dstb[q]), (unsigned)ia);             // RE >= 0: no borrow
          } else {
            const unsigned hi = (unsigned)(ib + (ia >> 31));
            atomicAdd(reinterpret_cast<unsigned long long*>(ptile + dstb[q]),
                      ((unsigned long long)hi << 32) | (unsigned)ia);
          }
#else
          atomicAdd(reinterpret_cast<unsigned*>(ptile + dstb[q]), (unsigned)ia);
          if (!LEB) atomicAdd(reinterpret_cast<unsigned*>(ptile + H::PLANE * 4 + dstb[q]), (unsigned)ib);
#endif
        }
      }
      }
    }
#if SSQ_PRIO & 1
    __builtin_amdgcn_s_setprio(0);
#endif
#ifdef SSQ_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // charge the atomics' drain to their own phase
#endif
    SSQ_STAMP(6);
    if (asym) {
      if (!asym_early && have_prev) read_out(psig, pframe0);
      __syncthreads();
      have_prev = true;
      psig = sig;
      pframe0 = frame0;
    } else {
    __syncthreads();
    SSQ_STAMP(7);
#if SSQ_PRIO & 2
    __builtin_amdgcn_s_setprio(2);
#endif
    read_out(sig, frame0);
#if SSQ_PRIO & 2
    __builtin_amdgcn_s_setprio(0);
#endif
    SSQ_STAMP(8);
    if (!SSQ_ABL(64)) __syncthreads();      // (ablation bit 64: what ONE barrier per tile would buy -- racy, results wrong)
    }
    SSQ_STAMP(9);
    if (!has_next) break;
    sig = nsig;
    jt = njt;
  }
  if (asym && have_prev) read_out(psig, pframe0);
#ifdef SSQ_STAMPS
  if (p.stamps && t == 0)
    for (int i = 0; i < 16; ++i) p.stamps[((long long)blockIdx.x * H::W + fl) * 16 + i] = st_acc[i];
#endif
}

// ------------------------------------------------------------------ launch ----
template <typename T>
bool fused_supported(int n_fft) {
  return n_fft >= 64 && n_fft <= 4096 && (n_fft & (n_fft - 1)) == 0;
}

template <typename T, int LOGN>
static hipError_t launch_one(const StftDev<T>& p0, int cu_count, long long batch, hipStream_t stream) {
  using C = FusedCfg<T, LOGN>;
  int per_cu = (160 * 1024) / C::LDS_BYTES;
  if (per_cu < 1) per_cu = 1;
  if (per_cu * C::W > 32) per_cu = 32 / C::W;
  // which kernel: SSQ_HIOCC = 0 generic 8-wave template; 1 = stft_tx1024_kernel with 16 waves, one block per CU;
  // 2 = the same kernel with 8 waves and two independent blocks per CU (fp32, n_fft = 1024, Tx output only)
  int hiocc = 0;
  if constexpr (sizeof(T) == 4 && LOGN == 10) {
    static const int mode = []() {
      const char* e = tune_env("SSQ_HIOCC");              // variant builds only (both alternatives measured slower)
      return e ? std::atoi(e) : SSQ_HIOCC_DEFAULT;
    }();
    if ((p0.out_kind == 0 || p0.out_kind == 3) && p0.n_eff == C::N) hiocc = mode;   // SSQ_OUT_WK: the (w, k) hook of the kernel that serves Tx
  }
  const int TF = hiocc == 2 ? 8 : (hiocc == 1 ? 16 : C::F);   // frames per tile of the kernel that will run
  // interior tiles [lo, hi): every frame of the tile reads only inside the signal
  const long long span = (long long)TF * p0.hop;
  const int tps_all = (p0.n_frames + TF - 1) / TF;
  long long lo = (p0.pad_left + span - 1) / span;
  long long hi_num = p0.n_signal - C::N - (long long)(TF - 1) * p0.hop + p0.pad_left;
  long long hi = hi_num >= 0 ? hi_num / span + 1 : 0;
  const long long full = p0.n_frames / TF;
  if (hi > full) hi = full;
  if (lo > tps_all) lo = tps_all;
  if (hi < lo) hi = lo;
  if (p0.n_eff != C::N) {
    // Bluestein mode: one launch of the edge-capable loader over all tiles (the transform, not the loader, bounds it)
    StftDev<T> p = p0;
    p.ta0 = 0;
    p.ta_n = tps_all;
    p.tb0 = 0;
    p.tiles_per_signal = tps_all;
    p.total_tiles = (long long)tps_all * batch;
    if (p.total_tiles <= 0) return hipSuccess;
    long long blocks = (long long)cu_count * per_cu;
    if (blocks > p.total_tiles) blocks = p.total_tiles;
    const dim3 g((unsigned)blocks), b(C::W * 64);
    if (p.out_kind == 0 && p.squeezing == 1) hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, true, true, true, false, true>), g, b, 0, stream, p);
    else if (p.out_kind == 0) hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, true, true, false, false, true>), g, b, 0, stream, p);
    else if (p.out_kind == 3) hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, true, true, false, true, true>), g, b, 0, stream, p);
    else hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, false, true, false, false, true>), g, b, 0, stream, p);
    return hipGetLastError();
  }
  // small jobs (a few waves of blocks, e.g. one to four 2^20-sample signals): ONE launch of the edge-capable kernel
  // over all tiles beats two launches -- the second launch costs more than the validity logic of the first
  const long long blocks_one_wave = (long long)cu_count * (hiocc == 2 ? 2 : (hiocc == 1 ? 1 : per_cu));
  bool single_launch = (long long)tps_all * batch <= 4 * blocks_one_wave;   // measured break-even: a few waves
  if (const char* e = std::getenv("SSQ_SINGLE_LAUNCH")) single_launch = std::atoi(e) != 0;   // tests: force either path
  for (int edge = single_launch ? 1 : 0; edge < 2; ++edge) {
    StftDev<T> p = p0;
    if (single_launch) {
      p.ta0 = 0;
      p.ta_n = tps_all;
      p.tb0 = 0;
      p.tiles_per_signal = tps_all;
    } else if (!edge) {
      p.ta0 = (int)lo;
      p.ta_n = (int)(hi - lo);
      p.tb0 = 0;
      p.tiles_per_signal = p.ta_n;
    } else {
      p.ta0 = 0;
      p.ta_n = (int)lo;
      p.tb0 = (int)hi;
      p.tiles_per_signal = (int)lo + (tps_all - (int)hi);
    }
    p.total_tiles = (long long)p.tiles_per_signal * batch;
    if (p.total_tiles <= 0) continue;
    long long blocks = (long long)cu_count * per_cu;
    if (blocks > p.total_tiles) blocks = p.total_tiles;
    const dim3 g((unsigned)blocks), b(C::W * 64);
    if constexpr (sizeof(T) == 4 && LOGN == 10) {
      if (hiocc) {
        const int per = hiocc == 2 ? 2 : 1;           // blocks per CU
        long long nb = (long long)cu_count * per;
        if (nb > p.total_tiles) nb = p.total_tiles;
        const dim3 gh((unsigned)nb), bh(hiocc == 2 ? 512 : 1024);
#define SSQ_LAUNCH_HI(E, LB)                                                                            \
  do {                                                                                                  \
    if (hiocc == 2) hipLaunchKernelGGL((stft_tx1024_kernel<E, LB, 8>), gh, bh, 0, stream, p);           \
    else hipLaunchKernelGGL((stft_tx1024_kernel<E, LB, 16>), gh, bh, 0, stream, p);                     \
  } while (0)
        if (p.out_kind == 3) {
          if (hiocc == 2) {
            if (edge) hipLaunchKernelGGL((stft_tx1024_kernel<true, false, 8, true>), gh, bh, 0, stream, p);
            else hipLaunchKernelGGL((stft_tx1024_kernel<false, false, 8, true>), gh, bh, 0, stream, p);
          } else {
            if (edge) hipLaunchKernelGGL((stft_tx1024_kernel<true, false, 16, true>), gh, bh, 0, stream, p);
            else hipLaunchKernelGGL((stft_tx1024_kernel<false, false, 16, true>), gh, bh, 0, stream, p);
          }
        } else if (p.squeezing == 1) {
          if (edge) SSQ_LAUNCH_HI(true, true);
          else SSQ_LAUNCH_HI(false, true);
        } else {
          if (edge) SSQ_LAUNCH_HI(true, false);
          else SSQ_LAUNCH_HI(false, false);
        }
#undef SSQ_LAUNCH_HI
        const hipError_t eh = hipGetLastError();
        if (eh != hipSuccess) return eh;
        continue;
      }
    }
    if (p.out_kind == 0 && p.squeezing == 1) {
      if (edge) hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, true, true, true>), g, b, 0, stream, p);
      else hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, true, false, true>), g, b, 0, stream, p);
    } else if (p.out_kind == 0) {
      if (edge) hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, true, true, false>), g, b, 0, stream, p);
      else hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, true, false, false>), g, b, 0, stream, p);
    } else if (p.out_kind == 3) {
      // (w, k) test hook: the Tx epilogue's own bins
      if (edge) hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, true, true, false, true>), g, b, 0, stream, p);
      else hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, true, false, false, true>), g, b, 0, stream, p);
    } else {
      if (edge) hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, false, true, false>), g, b, 0, stream, p);
      else hipLaunchKernelGGL((stft_fused_kernel<T, LOGN, false, false, false>), g, b, 0, stream, p);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

template <typename T>
int fused_tile_frames(int n_fft) {
  switch (n_fft) {
    case 64: return FusedCfg<T, 6>::F;
    case 128: return FusedCfg<T, 7>::F;
    case 256: return FusedCfg<T, 8>::F;
    case 512: return FusedCfg<T, 9>::F;
    case 1024: return FusedCfg<T, 10>::F;
    case 2048: return FusedCfg<T, 11>::F;
    case 4096: return FusedCfg<T, 12>::F;
  }
  return 0;
}

template <typename T>
hipError_t launch_stft_fused(const StftDev<T>& p, int n_fft, int cu_count, long long batch, hipStream_t stream) {
  switch (n_fft) {
    case 64: return launch_one<T, 6>(p, cu_count, batch, stream);
    case 128: return launch_one<T, 7>(p, cu_count, batch, stream);
    case 256: return launch_one<T, 8>(p, cu_count, batch, stream);
    case 512: return launch_one<T, 9>(p, cu_count, batch, stream);
    case 1024: return launch_one<T, 10>(p, cu_count, batch, stream);
    case 2048: return launch_one<T, 11>(p, cu_count, batch, stream);
    case 4096: return launch_one<T, 12>(p, cu_count, batch, stream);
  }
  return hipErrorInvalidValue;
}

template bool fused_supported<float>(int);
template bool fused_supported<double>(int);
template int fused_tile_frames<float>(int);
template int fused_tile_frames<double>(int);
template hipError_t launch_stft_fused<float>(const StftDev<float>&, int, int, long long, hipStream_t);
template hipError_t launch_stft_fused<double>(const StftDev<double>&, int, int, long long, hipStream_t);

}  // namespace ssq
