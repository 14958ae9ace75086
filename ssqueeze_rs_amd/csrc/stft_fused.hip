// stft_fused.hip -- launches of the fused STFT / synchrosqueezed-STFT kernel (stft_fused_kernel.h) for power-of-two
// n_fft, and the 16-wave kernel for fp32 n_fft = 1024 (the headline path).
#include "stft_fused_kernel.h"

namespace ssq {

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
#ifndef SSQ_FL_SCALAR
#define SSQ_FL_SCALAR 1
#endif
#if SSQ_FL_SCALAR
  // wave-uniform by construction: say so, and everything derived from it (exchange row, column-scale slot, frame index)
  // lives in scalar registers instead of (spilled) vector ones
  const int fl = __builtin_amdgcn_readfirstlane(tid >> 6);         // wave = frame inside the tile
#else
  const int fl = tid >> 6;         // wave = frame inside the tile
#endif
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
      // one wave-uniform branch around two straight runs of loads (a branch per sample makes every load wait for itself)
      const bool valid = frame < p.n_frames;
      const long long first = pos0 - t;
      if (__all(valid && first >= 0 && first + N <= p.n_signal)) {
#pragma unroll
        for (int q = 0; q < 16; ++q) xv[q] = xs[pos0 + L * q];
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) xv[q] = load_padded_flat(xs, pos0 + L * q, p.n_signal, p.padtype, valid);
      }
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
  const int grid_n = (int)gridDim.x;          // (read once: inside the loop it is a scalar load + wait per tile)
#pragma unroll 1
  while (true) {
    const int frame0 = tile_frame0(jt);
    const bool valid = EDGE ? (frame0 + fl < p.n_frames) : true;
    if (asym_early && have_prev) read_out(psig, pframe0);
    // next tile of this block; prefetch its samples behind this frame's FFT
    long long nsig = sig;
    int njt = jt + grid_n;
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
  if (p.n_eff != n_fft) return launch_stft_anylen<T>(p, n_fft, cu_count, batch, stream);   // stft_anylen.hip
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
