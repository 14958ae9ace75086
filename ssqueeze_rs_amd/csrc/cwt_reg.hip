// cwt_reg.hip -- inverse length-P transforms of the CWT on the per-wave register FFT core (fp32, P = 2^20 or 2^21).
//
// cwt_kernels.hip runs a length-P inverse FFT (cwt.rs:228-310) as two tile passes in which 8 waves share a tile in
// LDS and every FFT pass is an LDS round trip behind a block barrier.  Here each 1024-point transform lives in ONE
// wavefront's registers (fft_wave1024.h) and LDS is only the transposition buffer:
//   * analytic wavelets: psih_s[k] == 0 for k > P/2 (cwt.rs:512, :536), so with D = P / 2^20 (1 or 2) the time samples
//     split by residue, x[D m + d] = sum_{k < 2^20} (Y[k] e^{+2 pi i k d/P}) e^{+2 pi i k m/2^20}: D transforms of
//     length 2^20 = 1024 x 1024 (the k = P/2 term of D = 2, where it is not zero, is added as (-1)^n Y[P/2]);
//   * the spectrum is kept TRANSPOSED, xc[d][b][a] = conj(X[k] W_P^(-k d)) with k = 1024 a + b, and so is the wavelet
//     table: a row b is contiguous, so step R1 -- for every row b the transform over a, times W_{2^20}^(b n_a) --
//     is a pure streaming kernel: one wave per row, coalesced loads straight into the register layout, coalesced
//     stores, NO block barrier;
//   * step R2 -- for every column n_a the transform over b -- takes 16 adjacent output columns per block (16 waves),
//     transposes them through an LDS tile [column][row] on the way in and on the way out (128-byte global segments);
//     a wave's own tile column doubles as its exchange row.
// Everything is computed on conjugated data with the forward core (ifft(x) = conj(fft(conj(x)))).
#include "cwt_kernels.h"
#include "fft_wave1024.h"

namespace ssq {

// ablation bits of a diagnostic build (python -m ssqueeze_rs_amd.build --variant X -DSSQ_REG_ABL=n; results wrong by
// construction): R1 1 no loads, 2 no stores, 4 no FFT; R2 8 no loads, 16 no stores, 32 no FFT.  profiles/r02_abl_cwt_reg.txt
#ifndef SSQ_REG_ABL
#define SSQ_REG_ABL 0
#endif
#define REG_ABL(bit) ((SSQ_REG_ABL & (bit)) != 0)
#ifndef SSQ_R2_XCDPAIR
#define SSQ_R2_XCDPAIR 1
#endif
#ifndef SSQ_R2_EARLY
#define SSQ_R2_EARLY 0      // 1: issue the next tile's 16 loads at the top of the tile (32 more registers across the FFT): +1.7 % on C4
#endif

namespace {

constexpr int kRegThreads = 1024;       // 16 waves, one block per CU
constexpr int kRegWaves = 16;

}  // namespace

// xc[d][b][a] = conj(X[k] * e^{+2 pi i k d/P}),  k = 1024 a + b < 2^20
__global__ void cwt_reg_prep_kernel(CwtRegDev p) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)p.D << 20) return;
  const int d = (int)(idx >> 20);
  const int k = (int)(idx & ((1 << 20) - 1));          // coalesced read of the spectrum
  const int a = k >> 10, b = k & 1023;
  cpx<float> x = p.xh[k];
  if (d) {
    const long long r = (long long)k * d;               // < P
    const cpx<float> w = cmul(p.tw_hi[r >> 12], p.tw_lo[r & 4095]);   // W_P^r = e^{-2 pi i r/P}
    x = cmul(x, cpx<float>{w.x, -w.y});
  }
  p.xc[((long long)(d * 1024 + b) << 10) + a] = {x.x, -x.y};
}

// psiT[offT[s] + b*A[s] + a] = psih_s[1024 a + b] (0 beyond the band); only the scales with A[s] > 0
__global__ void cwt_reg_table_kernel(float* __restrict__ psiT, const long long* __restrict__ offT, const int* __restrict__ A,
                                     const float* __restrict__ psih, const long long* __restrict__ off,
                                     const int* __restrict__ band, int na) {
  const int s = blockIdx.y;
  if (s >= na) return;
  const int As = A[s];
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)As * 1024) return;
  const int a = (int)(idx >> 10), b = (int)(idx & 1023);   // coalesced read of the natural table
  const long long k = idx;                                  // = 1024 a + b
  const float v = (k < band[s]) ? psih[off[s] + k] : 0.0f;
  psiT[offT[s] + (long long)b * As + a] = v;
}

// ---------------------------------------------------------------------------------------------------- step R1 ----
// item = (transform tr, residue d, row b): one wave each.  A wave takes the row groups (scale, b) g, g + G, ... and
// runs the residues and kinds of a group back to back: they share the wavelet row and (per residue) the spectrum row,
// so three of four loads hit the cache.
__global__ __launch_bounds__(kRegThreads, 1) void cwt_reg_r1_kernel(CwtRegDev p) {
  __shared__ __attribute__((aligned(16))) cpx<float> exch_all[kRegWaves * kWave1024ExchElems];
  __shared__ __attribute__((aligned(16))) cpx<float> tws[kWave1024TwElems];
  __shared__ __attribute__((aligned(16))) cpx<float> thi[1024];   // W_{2^20}^(1024 j) = W_1024^j
  __shared__ __attribute__((aligned(16))) cpx<float> tlo[1024];   // W_{2^20}^i
  const int tid = threadIdx.x;
  const int t = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: keeps the item bookkeeping on the scalar unit
  cpx<float>* tw1 = tws;
  cpx<float>* tw2 = tws + 256;
  wave1024_tables(tw1, tw2, p.tw1024, tid, kRegThreads);
  thi[tid] = p.tw1024[tid];
  tlo[tid] = p.tw20[tid];
  __syncthreads();
  cpx<float>* exch = exch_all + wv * kWave1024ExchElems;

  const int logD = p.D >> 1;                                     // D = 1 or 2
  const int kshift = p.n_kinds - 1;                              // n_kinds = 1 or 2
  const int lsub = logD + kshift;                                // sub-items (d, kind) of a row group
  const int groups = (p.n_transforms >> kshift) << 10;           // (scale, b)
  const int G = (int)gridDim.x * kRegWaves;
  const int g0 = (int)blockIdx.x * kRegWaves + wv;
  if (g0 >= groups) return;
  // counter c of this wave -> item: row group g0 + (c >> lsub) G, sub-item c & (2^lsub - 1) = (d << kshift) | kind
  const int items = (((groups - 1 - g0) / G) + 1) << lsub;       // this wave's item count
  int it = 0;

  cpx<float> xr[16];
  float pr[16];
  // stage 0: the first 8 spectrum values (issued behind exchange 1, when one register set of the transform is live);
  // stage 1: the other 8 and the wavelet row (issued behind the last pass, in front of the stores)
  auto fetch = [&](int item, int stage) {
    const int rg = g0 + (item >> lsub) * G;
    const int b = rg & 1023;
    const int d = (item >> kshift) & (p.D - 1);
    const int s = p.scale0 + (rg >> 10);
    const int A = p.psiT_A[s];
    const cpx<float>* __restrict__ xrow = p.xc + ((long long)(d * 1024 + b) << 10);
    const float* __restrict__ prow = p.psiT + p.psiT_off[s] + (long long)b * A;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (64 * q < A && !REG_ABL(1)) {                   // wave-uniform: dead spans issue no loads
        const int a = t + 64 * q;
        if ((q < 8) == (stage == 0)) xr[q] = xrow[a];
        if (stage == 1) pr[q] = prow[a < A ? a : A - 1];
      }
    }
  };
  fetch(it, 0);
  fetch(it, 1);
#pragma unroll 1
  while (true) {
    const int rg = g0 + (it >> lsub) * G;
    const int b = rg & 1023;
    const int d = (it >> kshift) & (p.D - 1);
    const int kind = it & kshift;
    const int tr = ((rg >> 10) << kshift) | kind;
    const int s = p.scale0 + (rg >> 10);
    const int A = p.psiT_A[s];
    cpx<float> v[16];
    // conj(X psih (i xi)^kind) = xc * psih  |  xc * (-i) * (psih * xi)      (cwt.rs:238-240, :205-208)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int a = t + 64 * q;
      float ps = (a < A) ? pr[q] : 0.0f;
      cpx<float> x = xr[q];
      if (64 * q >= A) x = {0.0f, 0.0f};
      if (kind == 1) {
        const float xi = (float)(1024 * a + b) * p.xi_step;
        ps *= xi;
        x = {x.y, -x.x};
      }
      v[q] = {x.x * ps, x.y * ps};
    }
    if (!REG_ABL(4)) wave1024_front(v, exch, t);
    const int nxt = it + 1;
    const bool has_next = nxt < items;
    if (has_next) fetch(nxt, 0);
    if (!REG_ABL(4)) wave1024_back(v, tw1, tw2, t);
    if (has_next) fetch(nxt, 1);
    // times W_{2^20}^(b n_a), n_a = t + 64 q:  W^(b t) per lane, W^(64 b q) wave-uniform (lane q computes it)
    cpx<float> base, sq;
    {
      const int r = b * t;                                // < 2^16
      base = cmul(thi[r >> 10], tlo[r & 1023]);
      const int r2 = (64 * b * (t & 15)) & ((1 << 20) - 1);
      sq = cmul(thi[r2 >> 10], tlo[r2 & 1023]);
    }
    cpx<float>* __restrict__ yrow = p.ybuf + ((long long)((tr << logD) + d) << 20) + (b << 10) + t;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      cpx<float> sw;
      sw.x = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sq.x), q));
      sw.y = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(sq.y), q));
      const cpx<float> o = cmul(v[q], cmul(base, sw));
      if (!REG_ABL(2) || o.x == 12345.678f) yrow[64 * q] = o;
    }
    if (!has_next) break;
    it = nxt;
  }
}

// ---------------------------------------------------------------------------------------------------- step R2 ----
// tile = 16 adjacent output columns c = D (n_a - n_a0) + d of one transform; time sample n = D n_a0 + c + 1024 D n_b
__global__ __launch_bounds__(kRegThreads, 1) void cwt_reg_r2_kernel(CwtRegDev p) {
  constexpr int PT = 1024 + 2;          // column pitch: the transposing accesses (16 columns x 2 rows per 32 lanes) hit 64 banks
  __shared__ __attribute__((aligned(16))) cpx<float> tile[16 * PT];
  __shared__ __attribute__((aligned(16))) cpx<float> tws[kWave1024TwElems];
  const int tid = threadIdx.x;
  const int t = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  cpx<float>* tw1 = tws;
  cpx<float>* tw2 = tws + 256;
  wave1024_tables(tw1, tw2, p.tw1024, tid, kRegThreads);

  const int logD = p.D >> 1;                              // D = 1 or 2
  const int kshift = p.n_kinds - 1;                       // n_kinds = 1 or 2
  const int tshift = 6 + logD;                            // 64 D tiles per transform
  const int tiles = p.n_transforms << tshift;
  // D = 2: a tile reads 64-byte halves of ybuf's 128-byte lines and tile j ^ 1 reads the other halves.  Blocks go
  // round-robin over the 8 XCDs, so blocks b and b + 8 share an L2: give THEM the two tiles of a pair (the counters
  // showed every line fetched twice without this, profiles/r02_cwt_traffic_reg.json)
  int bid = blockIdx.x;
#if SSQ_R2_XCDPAIR
  if (gridDim.x % 16 == 0) bid = 2 * (bid & 7) + 16 * (bid >> 4) + ((bid >> 3) & 1);
#endif
  int tl = bid;
  if (tl >= tiles) return;
  const int c_ld = tid & 15, r_ld = tid >> 4;            // transposing accesses: column fastest
  const int d_ld = c_ld & (p.D - 1), na_ld = c_ld >> logD;
  const int ca = 16 >> logD;                              // n_a per tile

  cpx<float> pf[16];
  auto fetch = [&](int tile_id, int stage) {               // stage 0 behind exchange 1, stage 1 behind the last pass
    const int tr = tile_id >> tshift;
    const int j = tile_id & ((1 << tshift) - 1);
    const cpx<float>* __restrict__ src =
        p.ybuf + ((long long)(tr << logD) << 20) + ((d_ld << 20) + (r_ld << 10) + (j * ca + na_ld));
    if REG_ABL(8) return;
#pragma unroll
    for (int i = 0; i < 8; ++i) pf[8 * stage + i] = src[(long long)(8 * stage + i) << 16];      // rows r_ld + 64 i
  };
  fetch(tl, 0);
  fetch(tl, 1);
  cpx<float>* col = tile + wv * PT;
#pragma unroll 1
  while (true) {
    const int tr = tl >> tshift;
    const int j = tl & ((1 << tshift) - 1);
    const int s = p.scale0 + (tr >> kshift);
    const int kind = tr & kshift;
#pragma unroll
    for (int i = 0; i < 16; ++i) tile[c_ld * PT + r_ld + 64 * i] = pf[i];
    const int nxt = tl + (int)gridDim.x;
    const bool has_next = nxt < tiles;
#if SSQ_R2_EARLY
    if (has_next) {
      fetch(nxt, 0);
      fetch(nxt, 1);
    }
#endif
    __syncthreads();
    cpx<float> v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = col[t + 64 * q];
    frame_sync<false>();
    if (!REG_ABL(32)) wave1024_front(v, col, t);
#if !SSQ_R2_EARLY
    if (has_next) fetch(nxt, 0);
#endif
    if (!REG_ABL(32)) wave1024_back(v, tw1, tw2, t);
#if !SSQ_R2_EARLY
    if (has_next) fetch(nxt, 1);
#endif
    // D = 2: the k = P/2 term, (-1)^n Y[P/2] with n = d (mod 2), d = this wave's column (mod 2)
    cpx<float> nyq = {0.0f, 0.0f};
    const float sc = p.out_scale[s];
    if (p.D == 2) {
      const long long half = p.P >> 1;
      if ((long long)p.band[s] > half) {
        const cpx<float> xv = p.xh[half];
        float ps = p.psih[p.psi_off[s] + half];
        cpx<float> y = {xv.x * ps, xv.y * ps};
        if (kind == 1) {
          const float xi = (float)half * p.xi_step;
          y = {-y.y * xi, y.x * xi};
        }
        nyq = (wv & 1) ? cpx<float>{-y.x, -y.y} : y;
      }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) col[t + 64 * q] = {(v[q].x + nyq.x) * sc, (nyq.y - v[q].y) * sc};
    __syncthreads();
    {
      cpx<float>* __restrict__ row = ((kind == 1) ? p.dWx : p.Wx) + (long long)s * p.cols;   // wave-uniform base
      const int n0 = j * 16 + c_ld + ((1024 * r_ld) << logD);
      const int lo = p.rpadded ? 0 : (int)p.n1;
      const int cnt = p.rpadded ? (int)p.P : (int)p.n_signal;                                 // cwt.rs:115
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const cpx<float> val = tile[c_ld * PT + r_ld + 64 * i];
        const int n = n0 + ((65536 * i) << logD) - lo;
        if (REG_ABL(16) && val.x != 12345.678f) continue;
        if ((unsigned)n < (unsigned)cnt) row[n] = val;
      }
    }
    if (!has_next) break;
    __syncthreads();
    tl = nxt;
  }
}

// (Measured and removed: R2 keeping a tile's outputs in registers and issuing their global stores only after the next tile
// is in LDS, so that they drain behind its transform: +3 % on C4, profiles/r02_ab_cwt_latestore.txt.)
// (Measured and removed: the band-limited scales with psih_s[k] == 0 for k >= 1024 as ONE zero-padded wave transform per
// residue on this core, sharing step R2's store phase -- 0.36 ms SLOWER on C4 than the tile kernel's mode Z
// (profiles/r02_ab_cwt_regz.txt): with one 16-wave block per CU nothing overlaps the store phase, and the short
// transforms (Q = 16 ... 512) of the tile kernel run two blocks per CU.)

hipError_t launch_cwt_reg_prep(const CwtRegDev& p, hipStream_t stream) {
  const long long n = (long long)p.D << 20;
  hipLaunchKernelGGL(cwt_reg_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p);
  return hipGetLastError();
}

hipError_t launch_cwt_reg_table(float* psiT, const long long* d_offT, const int* d_A, int max_A, const float* psih,
                                const long long* d_off, const int* d_band, int na, hipStream_t stream) {
  if (max_A <= 0 || na <= 0) return hipSuccess;
  hipLaunchKernelGGL(cwt_reg_table_kernel, dim3((unsigned)(((long long)max_A * 1024 + 255) / 256), (unsigned)na), dim3(256), 0,
                     stream, psiT, d_offT, d_A, psih, d_off, d_band, na);
  return hipGetLastError();
}

hipError_t launch_cwt_reg_inv(const CwtRegDev& p, int n_cus, hipStream_t stream) {
  if (p.n_transforms <= 0) return hipSuccess;
  const long long groups = (long long)(p.n_transforms / p.n_kinds) * 1024;       // (scale, row) groups
  long long g1 = (groups + kRegWaves - 1) / kRegWaves;
  if (g1 > n_cus) g1 = n_cus;
  hipLaunchKernelGGL(cwt_reg_r1_kernel, dim3((unsigned)g1), dim3(kRegThreads), 0, stream, p);
  long long g2 = (long long)p.n_transforms * 64 * p.D;
  if (g2 > n_cus) g2 = n_cus;
  hipLaunchKernelGGL(cwt_reg_r2_kernel, dim3((unsigned)g2), dim3(kRegThreads), 0, stream, p);
  return hipGetLastError();
}

}  // namespace ssq
