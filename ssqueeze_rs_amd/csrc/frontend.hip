// frontend.hip -- device side of the chunked-overlap multi-channel front end.
//
// The reference's real callers do not hand `_rs.*` whole recordings: `process_stft_ssq` / `process_ssq_cwt`
// (tests/stft_ssq_test.py:163-283, tests/ssq_cwt_test.py:66-195) cut a (samples, channels) array into chunks of
// 1 000 000 samples (:302), extend every chunk by `depth` samples on both sides (`map_overlap`, neighbours' samples
// inside the array, boundary="reflect" at its two ends, :275-281), loop over the channels in Python calling `_rs.*`
// once per channel per extended chunk (:230-248) and stack the results as (freq, frames, channels) (:265-267).
//
// Here the channel array is uploaded ONCE into [channels][depth + samples + depth]; the two array-end halos are
// filled on the device (ssq_chunk_halo_fill), every extended chunk is then a contiguous window of that buffer and
// all (channel, chunk) windows go through the plan as one strided batch (ssq_stft_plan_exec_strided), and the
// (freq, frames, channels) stacking is one device pass (ssq_chunks_relayout).  No padded or per-chunk copies.
#include "../../include/ssq_hip.h"
#include "ssq_common.h"

using namespace ssq;

namespace {

// dask.array.overlap.reflect: the halo mirrors the array INCLUDING its edge sample (numpy "symmetric"):
// position -m (m = 1..depth) holds x[m-1]; position S-1+m holds x[S-m].   boundary 1: constant 0.
template <typename T>
__global__ void halo_fill_kernel(T* __restrict__ xext, long long channels, long long samples, long long depth,
                                 int boundary) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // 0 .. 2*depth-1
  const long long ch = blockIdx.y;
  if (i >= 2 * depth || ch >= channels) return;
  T* row = xext + ch * (samples + 2 * depth);
  const T* x = row + depth;
  if (i < depth) {
    const long long m = depth - i;                 // position -m
    row[i] = boundary == 0 ? x[m - 1] : (T)0;
  } else {
    const long long m = i - depth + 1;             // position S-1+m
    row[depth + samples - 1 + m] = boundary == 0 ? x[samples - m] : (T)0;
  }
}

// out[(r * out_cols + out_col_base + j * ncols + c) * out_channels + ch_base + ch] = in[((ch * chunks + j) * rows + r) * cols_in + col0 + c]
template <typename CT>
__global__ void relayout_kernel(const CT* __restrict__ in, long long channels, long long chunks, long long rows,
                                long long cols_in, long long col0, long long ncols, CT* __restrict__ out,
                                long long out_cols, long long out_col_base, long long out_channels, long long ch_base) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // over ncols * channels
  const long long r = blockIdx.y;
  const long long j = blockIdx.z;
  if (i >= ncols * channels) return;
  const long long c = i / channels, ch = i - c * channels;
  out[(r * out_cols + out_col_base + j * ncols + c) * out_channels + ch_base + ch] =
      in[((ch * chunks + j) * rows + r) * cols_in + col0 + c];
}

struct c8 {
  float x, y;
};
struct c16 {
  double x, y;
};

}  // namespace

extern "C" {

int ssq_chunk_halo_fill(int dtype, void* d_xext, int64_t channels, int64_t samples, int64_t depth, int boundary,
                        void* stream) {
  if (!d_xext) SSQ_FAIL("device pointer is NULL");
  if (channels <= 0 || samples <= 0 || depth < 0) SSQ_FAIL("bad shape");
  if (depth > samples) SSQ_FAIL("overlap depth larger than the array (dask.map_overlap rejects this too)");
  if (depth == 0) return 0;
  if (channels > 65535) SSQ_FAIL("too many channels");
  dim3 grid((unsigned)((2 * depth + 255) / 256), (unsigned)channels, 1);
  if (dtype == SSQ_F32)
    hipLaunchKernelGGL(halo_fill_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (float*)d_xext, channels,
                       samples, depth, boundary);
  else if (dtype == SSQ_F64)
    hipLaunchKernelGGL(halo_fill_kernel<double>, grid, dim3(256), 0, (hipStream_t)stream, (double*)d_xext, channels,
                       samples, depth, boundary);
  else
    SSQ_FAIL("dtype must be SSQ_F32 or SSQ_F64");
  SSQ_HIP(hipGetLastError());
  return 0;
}

int ssq_chunks_relayout(int dtype, const void* d_in, int64_t channels, int64_t chunks, int64_t rows, int64_t cols_in,
                        int64_t col0, int64_t ncols, void* d_out, int64_t out_cols, int64_t out_col_base,
                        int64_t out_channels, int64_t ch_base, void* stream) {
  if (!d_in || !d_out) SSQ_FAIL("device pointer is NULL");
  if (channels <= 0 || chunks <= 0 || rows <= 0 || ncols <= 0) return 0;
  if (col0 < 0 || col0 + ncols > cols_in || ch_base < 0 || ch_base + channels > out_channels ||
      out_col_base < 0 || out_col_base + chunks * ncols > out_cols)
    SSQ_FAIL("relayout window out of range");
  if (rows > 65535 || chunks > 65535) SSQ_FAIL("too many rows or chunks for one relayout launch");
  dim3 grid((unsigned)((ncols * channels + 255) / 256), (unsigned)rows, (unsigned)chunks);
  if (dtype == SSQ_F32)
    hipLaunchKernelGGL(relayout_kernel<c8>, grid, dim3(256), 0, (hipStream_t)stream, (const c8*)d_in, channels, chunks,
                       rows, cols_in, col0, ncols, (c8*)d_out, out_cols, out_col_base, out_channels, ch_base);
  else if (dtype == SSQ_F64)
    hipLaunchKernelGGL(relayout_kernel<c16>, grid, dim3(256), 0, (hipStream_t)stream, (const c16*)d_in, channels,
                       chunks, rows, cols_in, col0, ncols, (c16*)d_out, out_cols, out_col_base, out_channels, ch_base);
  else
    SSQ_FAIL("dtype must be SSQ_F32 or SSQ_F64");
  SSQ_HIP(hipGetLastError());
  return 0;
}

}  // extern "C"
