/*
 * ssq_hip.h -- C-ABI of libssq_hip.so, the MI355X (gfx950) synchrosqueezing engine.
 *
 * This header is the drop-in boundary for the hot path of jesusdpa1/ssqueeze_rs:
 * each entry point replaces one PyO3 `#[pyfunction]` of the reference's `_rs`
 * extension module (registered at rust/src/lib.rs:22-35).  Plain pointers and
 * sizes only; no Python, no torch types.  All functions return 0 on success and
 * a non-zero status on failure; `ssq_last_error()` then holds a message
 * (thread-local).  The library never frees or retains caller memory.
 *
 * dtype: SSQ_F32 (float in, interleaved complex-float out) or SSQ_F64
 * (double in, interleaved complex-double out).  SSQ_F64 with batch == 1 is the
 * reference's own configuration (PyReadonlyArray1<f64> in, complex128 out).
 * Batches are C-contiguous `[batch][N]` in and `[batch][rows][cols]` out.
 *
 * Two families:
 *   *_host  : host pointers in/out (H2D, kernels, D2H inside) -- what the
 *             Python/NumPy mirror `ssqueeze_rs_amd._rs` binds.
 *   plans   : device pointers + a caller stream, no allocation and no host
 *             synchronisation in the exec call (hipGraph-capturable) -- what
 *             batch pipelines and bench.py bind.
 */
#ifndef SSQ_HIP_H
#define SSQ_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { SSQ_F32 = 0, SSQ_F64 = 1 };
enum { SSQ_PAD_REFLECT = 0, SSQ_PAD_ZERO = 1 };          /* stft_utils.rs:19-65, utils/array.rs:52-98 */
enum { SSQ_SQUEEZE_SUM = 0, SSQ_SQUEEZE_LEBESGUE = 1 };  /* ssq_stft.rs:292-296, ssq_cwt.rs:199-206 */
enum { SSQ_WAVELET_GMW = 0, SSQ_WAVELET_MORLET = 1 };    /* cwt.rs:496-543 */
enum { SSQ_FREQS_LOG = 0, SSQ_FREQS_LINEAR = 1 };        /* ssq_cwt.rs:56-112 */
enum { SSQ_MAPRANGE_PEAK = 0, SSQ_MAPRANGE_MAXIMAL = 1 };/* ssq_cwt.rs:450-461 */
/* numerics variant of a plan / host call (bit flags).  0 = the Rust reference (rust/src/spectral/ *.rs).
 * SSQ_VARIANT_UPSTREAM = the vendored upstream ssqueezepy the Rust crate was derived from (SURVEY 8(f)-4,
 * /root/reference/old/ssqueezepy): pad split, Nyquist-zeroed diff-window, np.linspace frequencies, clamped
 * round-half-even bins, |.| > gamma, normalised wavelets with a halved Nyquist bin, p2up padding, ln2/nv constant.
 * MODULATED (STFT family, _stft.py:127-147) and FLIPUD (ssqueezing.py: k -> n-1-k) qualify UPSTREAM. */
enum { SSQ_VARIANT_RUST = 0, SSQ_VARIANT_UPSTREAM = 1, SSQ_VARIANT_MODULATED = 2, SSQ_VARIANT_FLIPUD = 4 };
/* what a STFT-family plan writes to its output */
enum {
  SSQ_OUT_TX  = 0,   /* synchrosqueezed STFT            (ssq_stft.rs:270-301) */
  SSQ_OUT_SX  = 1,   /* STFT                            (stft.rs:47-85)       */
  SSQ_OUT_DSX = 2,   /* derivative STFT                 (ssq_stft.rs:205-211,227) -- test hook */
  SSQ_OUT_WK  = 3    /* (w, k) per bin as (re, im)      (ssq_stft.rs:11-39,280-289) -- test hook */
};

/* ---- library / device ----------------------------------------------------- */
const char* ssq_last_error(void);
const char* ssq_hello_from_bin(void);                    /* lib.rs:16-19 */
/* 1 in -DSSQ_TUNING variant builds (tuning / ablation environment switches compiled in), 0 in the product library */
int ssq_build_has_tuning(void);
int ssq_device_count(int* count);
int ssq_set_device(int device);
int ssq_device_info(int* cu_count, int64_t* hbm_bytes, char* name, int name_len);

/* ---- shape helpers (so callers can allocate outputs) ---------------------- */
/* stft.rs:32-34 / ssq_stft.rs:182-184 */
int ssq_stft_shape(int64_t n_signal, int64_t n_fft, int64_t hop, int64_t* n_freqs, int64_t* n_frames);
/* utils/array.rs:9-11 with cwt.rs:87,98: P = next_power_of_2(N + N/2), n1 = (P-N)/2 */
int ssq_cwt_pad_len(int64_t n_signal, int64_t* pad_len, int64_t* n1);
/* cwt.rs:461-489 / ssq_cwt.rs:300-326; simd_variant selects cwt_simd.rs:474-545.
 * Call with scales == NULL to obtain *na only. */
int ssq_log_scales(int64_t n_signal, int64_t nv, int simd_variant, int64_t* na, double* scales);
/* ssq_stft.rs:104-119: centre zero-pad / centre-crop `window[win_n]` to n_fft */
int ssq_size_window(const double* window, int64_t win_n, int64_t n_fft, double* out);
/* ssq_stft.rs:131-179: spectral derivative of the window (Nyquist term kept) */
int ssq_diff_window(const double* window, int64_t n_fft, double* out);
/* ssq_cwt.rs:450-469: the `ssq_freqs` vector ssq_cwt returns */
int ssq_cwt_ssq_freqs(const double* scales, int64_t na, int64_t n_signal, double dt,
                      int maprange, int freq_dist, double* ssq_freqs);

/* ---- host-pointer entry points (replace the PyO3 functions) ---------------- */
/* _rs.stft            rust/src/spectral/stft.rs:12-95
 * window has n_fft entries.  Sx: [batch][n_freqs][n_frames]; freqs: [n_freqs] (cycles/sample). */
int ssq_stft_host(int dtype, const void* x, int64_t batch, int64_t n_signal,
                  const double* window, int64_t n_fft, int64_t hop, int padtype,
                  void* Sx, double* freqs);

/* _rs.ssq_stft        rust/src/spectral/ssq_stft.rs:72-313
 * `window` is already sized to n_fft (ssq_size_window).  gamma < 0 selects the default
 * 10*EPS64 (ssq_stft.rs:258-261).  Tx: [batch][n_freqs][n_frames]; ssq_freqs: [n_freqs].
 * dbg_* may be NULL; when given they receive Sx, dSx (complex) and (w,k) pairs. */
int ssq_ssq_stft_host(int dtype, const void* x, int64_t batch, int64_t n_signal,
                      const double* window, int64_t n_fft, int64_t hop, double fs,
                      int padtype, int squeezing, double gamma,
                      void* Tx, double* ssq_freqs,
                      void* dbg_Sx, void* dbg_dSx, void* dbg_wk);

/* _rs.cwt / _rs.cwt_simd   rust/src/spectral/cwt.rs:46-144, cwt_simd.rs:52-150
 * Wx, dWx: [batch][na][cols], cols = rpadded ? P : N.  dWx may be NULL (derivative=False). */
int ssq_cwt_host(int dtype, const void* x, int64_t batch, int64_t n_signal,
                 int wavelet, const double* scales, int64_t na, double dt,
                 int l1_norm, int padtype, int rpadded,
                 void* Wx, void* dWx);

/* _rs.ssq_cwt         rust/src/spectral/ssq_cwt.rs:244-493
 * Tx: [batch][na][N]; ssq_freqs: [na] (not flipped).  dbg_* may be NULL:
 * Wx, dWx (unpadded, complex) and (w, k-or--1) pairs. */
int ssq_ssq_cwt_host(int dtype, const void* x, int64_t batch, int64_t n_signal,
                     int wavelet, const double* scales, int64_t na, double dt,
                     int freq_dist, int maprange, int padtype, int squeezing,
                     int flipud, double gamma,
                     void* Tx, double* ssq_freqs,
                     void* dbg_Wx, void* dbg_dWx, void* dbg_wk);

/* _rs.icwt            rust/src/spectral/cwt.rs:550-718   (implemented there, advertised by _rs.pyi:61-73, not registered)
 * Wx: [na][n_times] interleaved complex of `dtype`; scales: [n_scales_given >= na] (NULL -> "Scales must be
 * provided"); x_len < 0 selects n_times; x_out: [x_len] float64.  one_int: scaled row sum of Re Wx (:588-627);
 * otherwise the FFT filter bank of :629-714 for ANY x_len. */
int ssq_icwt_host(int dtype, const void* Wx, int64_t na, int64_t n_times, int wavelet, const double* scales,
                  int64_t n_scales_given, int one_int, int64_t x_len, double x_mean, int l1_norm, double* x_out);

/* wavelet helper functions (host side, fp64): rust/src/wavelets/morlet.rs:59-145, gmw.rs:236-357, _rs.pyi:89-132.
 * out: [n] interleaved complex128.  norm: "bandpass" (any case) or anything else = the L2 branch. */
int ssq_morlet(const double* w, int64_t n, double mu, double* out);
int ssq_morlet_freq(int64_t n, double scale, double mu, double* out);
int ssq_morlet_time(int64_t n, double scale, double mu, double* out);
int ssq_gmw(const double* w, int64_t n, double gamma, double beta, const char* norm, int order, double* out);
int ssq_gmw_freq(int64_t n, double scale, double gamma, double beta, const char* norm, int order, double* out);
int ssq_gmw_time(int64_t n, double scale, double gamma, double beta, const char* norm, int order, double* out);
int ssq_gmw_center_frequency(double gamma, double beta, const char* kind, double* out);

/* ---- upstream-parity mode and the inverses (/root/reference/old/ssqueezepy) ---------------------
 * ssqueezepy.stft      old/ssqueezepy/_stft.py:13-193  (window already n_fft long: get_window, :257-309, is host logic
 *                      of the Python mirror).  Sx, dSx: [batch][n_fft/2+1][(N-1)/hop+1]; dSx may be NULL. */
int ssq_stft_host_v(int dtype, const void* x, int64_t batch, int64_t n_signal, const double* window, int64_t n_fft,
                    int64_t hop, double fs, int padtype, int variant, void* Sx, void* dSx);
/* ssqueezepy.ssq_stft  old/ssqueezepy/_ssq_stft.py:12-137 + algos.py:957-968.  gamma < 0: 10 eps of the dtype.
 * Sx, dSx, wk may be NULL; ssq_freqs: [n_freqs] (reversed with SSQ_VARIANT_FLIPUD). */
int ssq_ssq_stft_host_v(int dtype, const void* x, int64_t batch, int64_t n_signal, const double* window, int64_t n_fft,
                        int64_t hop, double fs, int padtype, int squeezing, double gamma, int variant, void* Tx,
                        double* ssq_freqs, void* Sx, void* dSx, void* wk);
/* ssqueezepy.istft     old/ssqueezepy/_stft.py:196-254 (+ utils/stft_utils.py:141-191): per-frame inverse real FFT,
 * fftshift when modulated, windowed overlap-add, division by the window norm, unpadding.
 * Sx: [n_fft/2+1][n_frames] complex of `dtype`; window: [n_fft]; x_out: [N] real of `dtype`. */
int ssq_istft_host(int dtype, const void* Sx, int64_t n_frames, const double* window, int64_t n_fft, int64_t hop,
                   int64_t n_signal, int modulated, int win_exp, void* x_out);
/* ssqueezepy.issq_stft / issq_cwt, full inverse (_ssq_stft.py:139-198, _ssq_cwt.py:313-378):
 * x_out[j] = scale * sum_rows row_scale[row] * Re Tx[row][j]   (scale = 2 / window[n_fft/2]  resp.  2 / adm_ssq;
 * row_scale NULL = 1; the one-integral icwt of _cwt.py:477-492 is the same sum with 1/sqrt(a) rows for the L2 norm).
 * Tx: [rows][cols] complex of `dtype`; x_out: [cols] real of `dtype`. */
int ssq_issq_host(int dtype, const void* Tx, int64_t rows, int64_t cols, double scale, const double* row_scale,
                  void* x_out);
/* upstream wavelets: SSQ_WAVELET_GMW with (p0, p1) = (gamma, beta), L1 / bandpass norm (_gmw.py:187-210);
 * SSQ_WAVELET_MORLET with p0 = mu (wavelets.py:497-523).
 * adm_ssq = int_0^inf conj(psih(w)) / w dw, adm_cwt = int |psih|^2 / w (utils/cwt_utils.py:28-63, trapezoid on the
 * grids of integrate_analytic, :583-627); the peak centre frequency on the padded grid (wavelets.py:713-716). */
int ssq_upstream_adm(int wavelet, double p0, double p1, int which_cwt, double* out);
int ssq_upstream_center_frequency(int wavelet, double p0, double p1, double scale, int64_t n_padded, double* wc);
/* utils/common.py:32-51: padded length 2^(1 + round(log2 n)), left pad n1 >= right pad n2 */
int ssq_upstream_p2up(int64_t n_signal, int64_t* n_up, int64_t* n1, int64_t* n2);
/* ssqueezepy.cwt       old/ssqueezepy/_cwt.py:12-318 with explicit scales.  Wx, dWx: [batch][na][cols],
 * cols = rpadded ? n_up : N; dWx may be NULL. */
int ssq_cwt_host_v(int dtype, const void* x, int64_t batch, int64_t n_signal, int wavelet, double p0, double p1,
                   const double* scales, int64_t na, double dt, int l1_norm, int padtype, int rpadded, int variant,
                   void* Wx, void* dWx);
/* ssqueezepy.ssq_cwt   old/ssqueezepy/_ssq_cwt.py:12-311 + ssqueezing.py:122-146 + algos.py:899-910 (exponential
 * scales, difftype 'trig').  ssq_freqs_asc: [na] the ascending frequencies the bins refer to (the caller reverses
 * them like ssqueezing.py:199-205); nv: voices per octave of `scales` (the constant ln2/nv).
 * Wx, dWx, wk may be NULL. */
int ssq_ssq_cwt_host_v(int dtype, const void* x, int64_t batch, int64_t n_signal, int wavelet, double p0, double p1,
                       const double* scales, int64_t na, double dt, int nv, const double* ssq_freqs_asc, int freq_dist,
                       int padtype, int squeezing, double gamma, int variant, void* Tx, void* Wx, void* dWx, void* wk);

/* ---- plans: device-resident batch pipelines -------------------------------- */
typedef struct ssq_stft_plan ssq_stft_plan;
/* One plan = one (dtype, N, n_fft, hop, window, fs, padtype, squeezing, gamma) configuration.
 * force_generic != 0 selects the unfused any-n_fft kernels (test hook). */
int ssq_stft_plan_create(ssq_stft_plan** plan, int dtype, int64_t n_signal,
                         const double* window, int64_t n_fft, int64_t hop, double fs,
                         int padtype, int squeezing, double gamma, int force_generic);
/* the same with a numerics variant (SSQ_VARIANT_*); upstream plans run on the unfused kernels */
int ssq_stft_plan_create_v(ssq_stft_plan** plan, int dtype, int64_t n_signal,
                           const double* window, int64_t n_fft, int64_t hop, double fs,
                           int padtype, int squeezing, double gamma, int force_generic, int variant);
int ssq_stft_plan_destroy(ssq_stft_plan* plan);
/* 1 if the fused LDS-tile kernel serves this plan, 0 if the generic kernels do */
int ssq_stft_plan_is_fused(const ssq_stft_plan* plan);
/* bytes of device scratch exec needs for `batch` signals with output kind `out_kind` */
int64_t ssq_stft_plan_workspace_bytes(const ssq_stft_plan* plan, int64_t batch, int out_kind);
/* d_x: [batch][N]; d_out: [batch][n_freqs][n_frames] complex; async on `stream` (hipStream_t). */
int ssq_stft_plan_exec(ssq_stft_plan* plan, int out_kind, const void* d_x, int64_t batch,
                       void* d_out, void* d_workspace, int64_t workspace_bytes, void* stream);

/* The same over a STRIDED batch of n_groups * group signals: signal s starts at
 * d_x + (s / group) * group_stride + (s % group) * sig_stride (elements); windows may overlap.
 * This is how the chunked-overlap front end (tests/stft_ssq_test.py:216-281: map_overlap with depth = n_fft) runs
 * all extended chunks of all channels as one batch: group = chunks per channel, sig_stride = chunk,
 * group_stride = channel pitch.  d_out: [n_groups * group][n_freqs][n_frames]. */
int ssq_stft_plan_exec_strided(ssq_stft_plan* plan, int out_kind, const void* d_x, int64_t n_groups,
                               int64_t group, int64_t group_stride, int64_t sig_stride, void* d_out,
                               void* d_workspace, int64_t workspace_bytes, void* stream);

/* ---- chunked-overlap multi-channel front end (device side) ------------------
 * Replaces the Dask harness around `_rs.*`: tests/stft_ssq_test.py:163-283, tests/ssq_cwt_test.py:66-195. */
/* d_xext: [channels][depth + samples + depth]; the middle already holds the channel.  Fills the two array-end
 * halos like dask.map_overlap(boundary="reflect") (mirror INCLUDING the edge sample; boundary = 1: zeros). */
int ssq_chunk_halo_fill(int dtype, void* d_xext, int64_t channels, int64_t samples, int64_t depth,
                        int boundary, void* stream);
/* (channels, chunks, rows, cols) -> (rows, all chunks' columns, channels), the reference's np.transpose(stacked,
 * (1, 2, 0)) plus the concatenation of the chunk outputs (stft_ssq_test.py:265-267), complex elements of `dtype`:
 * out[(r*out_cols + out_col_base + j*ncols + c)*out_channels + ch_base + ch] = in[((ch*chunks + j)*rows + r)*cols_in + col0 + c] */
int ssq_chunks_relayout(int dtype, const void* d_in, int64_t channels, int64_t chunks, int64_t rows,
                        int64_t cols_in, int64_t col0, int64_t ncols, void* d_out, int64_t out_cols,
                        int64_t out_col_base, int64_t out_channels, int64_t ch_base, void* stream);

typedef struct ssq_cwt_plan ssq_cwt_plan;
int ssq_cwt_plan_create(ssq_cwt_plan** plan, int dtype, int64_t n_signal, int wavelet,
                        const double* scales, int64_t na, double dt, int padtype);
/* the same with a numerics variant (SSQ_VARIANT_*) and the upstream wavelet's parameters (p0, p1) = (gamma, beta) of
 * the GMW or (mu, -) of the Morlet wavelet; upstream plans pad by p2up and run on the generic transforms */
int ssq_cwt_plan_create_v(ssq_cwt_plan** plan, int dtype, int64_t n_signal, int wavelet, double p0, double p1,
                          const double* scales, int64_t na, double dt, int padtype, int variant);
int ssq_cwt_plan_destroy(ssq_cwt_plan* plan);
int64_t ssq_cwt_plan_workspace_bytes(const ssq_cwt_plan* plan, int64_t batch);
/* cwt: d_Wx/d_dWx [batch][na][cols]; d_dWx may be NULL */
int ssq_cwt_plan_exec_cwt(ssq_cwt_plan* plan, const void* d_x, int64_t batch, int l1_norm,
                          int rpadded, void* d_Wx, void* d_dWx,
                          void* d_workspace, int64_t workspace_bytes, void* stream);
/* ssq_cwt: d_Tx [batch][na][N]; d_dbg_* may be NULL */
int ssq_cwt_plan_exec_ssq(ssq_cwt_plan* plan, const void* d_x, int64_t batch,
                          int freq_dist, int maprange, int squeezing, int flipud, double gamma,
                          void* d_Tx, void* d_dbg_Wx, void* d_dbg_dWx, void* d_dbg_wk,
                          void* d_workspace, int64_t workspace_bytes, void* stream);

/* ---- multi-GPU: the optional final gather of the batch-sharded results over xGMI ------------------------------
 * Signals are independent (the reference's batch is a Python loop over channels, tests/stft_ssq_test.py:230), so the
 * data path has no collective; a consumer that wants every rank to hold all shards calls ssq_gather_shards after its
 * plan exec.  RCCL (librccl.so) is dlopen'd on first use -- no link-time dependency, a clear error when absent.
 * One process per GPU: rank 0 makes the 128-byte id (ssq_rccl_unique_id) and hands it to the others out of band
 * (MPI, a file, torch.distributed, ...); every rank then calls ssq_rccl_comm_init on its own device.  `comm` may also
 * be a ncclComm_t the caller created itself. */
int ssq_rccl_available(void);                                 /* 1 if librccl.so loads */
int ssq_rccl_unique_id(void* id128);                          /* ncclGetUniqueId */
int ssq_rccl_comm_init(void** comm, int n_ranks, const void* id128, int rank);   /* ncclCommInitRank (collective) */
int ssq_rccl_comm_info(void* comm, int* n_ranks, int* rank);  /* ncclCommCount / ncclCommUserRank */
int ssq_rccl_comm_destroy(void* comm);
/* ncclAllGather of `bytes_per_rank` bytes: d_recv[rank][bytes_per_rank], rank order = batch order; async on `stream` */
int ssq_gather_shards(void* comm, const void* d_send, void* d_recv, int64_t bytes_per_rank, void* stream);

/* ---- host-path caches ---------------------------------------------------------
 * The *_host entry points keep plans (keyed by their configuration), device scratch and two streams between
 * calls, and pipeline H2D / kernels / D2H over the signals of a batch.  Results that live in blocks of the
 * library's pinned pool arrive by DMA without a host-side copy; `ssqueeze_rs_amd._rs` allocates its NumPy results
 * there (the reference hands NumPy freshly allocated arrays too: ssq_stft.rs:307-312). */
int ssq_pinned_alloc(void** ptr, int64_t bytes);         /* pooled hipHostMalloc */
int ssq_pinned_free(void* ptr);                          /* back to the pool */
int ssq_host_cache_limit(int64_t idle_pinned_bytes);     /* idle pinned memory the pool may keep (default 8 GiB) */
int ssq_host_cache_stats(int64_t* live_pinned_bytes, int64_t* idle_pinned_bytes);
int ssq_host_cache_clear(void);                          /* drop cached plans, device scratch, idle pinned blocks */

/* ---- device memory / stream / event plumbing for FFI callers --------------- */
int ssq_dev_malloc(void** ptr, int64_t bytes);
int ssq_dev_free(void* ptr);
int ssq_dev_memset(void* ptr, int value, int64_t bytes, void* stream);
int ssq_memcpy_h2d(void* dst, const void* src, int64_t bytes, void* stream);
int ssq_memcpy_d2h(void* dst, const void* src, int64_t bytes, void* stream);
int ssq_memcpy_d2d(void* dst, const void* src, int64_t bytes, void* stream);
int ssq_stream_create(void** stream);
int ssq_stream_destroy(void* stream);
int ssq_stream_sync(void* stream);     /* NULL = default stream */
int ssq_device_sync(void);
int ssq_event_create(void** event);
int ssq_event_destroy(void* event);
int ssq_event_record(void* event, void* stream);
int ssq_event_sync(void* event);
int ssq_event_elapsed_ms(void* start, void* stop, float* ms);
/* HIP graphs: everything the plan execs enqueue on `stream` between begin and end (incl. the ssq_cwt exec's side-stream
 * fork / join) becomes ONE replayable launch.  Run the sequence once before capturing (lazy one-time setup in the plans);
 * replay with the same device pointers (new contents).  `stream` must be an explicit stream (ssq_stream_create). */
int ssq_graph_capture_begin(void* stream);
int ssq_graph_capture_end(void* stream, void** graph_exec);
int ssq_graph_launch(void* graph_exec, void* stream);
int ssq_graph_destroy(void* graph_exec);

#ifdef __cplusplus
}
#endif
#endif /* SSQ_HIP_H */
