/*
 * ppq_hip.h -- C ABI of libppq_hip.so: MI355X (gfx950) kernels for PPQ's quantization-simulation
 * hot path.  This is the drop-in boundary: every entry point replaces one function of the
 * reference's pybind module `PPQ_Cuda_Impls` (ppq/csrc/export.cc:8-34), reached in the reference
 * through `ppq.core.ffi.CUDA.*` (ppq/core/ffi.py:56-350).  Plain pointers and sizes only; no torch
 * types.  The Python binding that presents the pybind names lives in ppq_amd/ffi.py; the stub a
 * PPQ maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - all tensor pointers are DEVICE pointers to contiguous fp32 (histograms: int32) unless a
 *     parameter is documented as host memory;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call only
 *     enqueues work on that stream and returns -- no call synchronises the device;
 *   - outputs are never aliased with inputs by the library (the reference allocates a fresh
 *     tensor, linear.cu:111); in-place use (out == x) is nevertheless safe for the elementwise ops;
 *   - return value: PPQHIP_OK (0) or a negative ppqhip_status; ppqhip_last_error() returns a
 *     thread-local message.  PPQHIP_ERR_INVALID_VALUE corresponds to the reference's
 *     InvalidValueException (common.cuh:41-48: empty tensor, numel > 0x7fffffff, bad histogram
 *     shape); dtype errors (ValueTypeException, common.cuh:32-39) cannot occur behind a typed C
 *     ABI and are raised by the Python binding instead;
 *   - `rounding` uses the values of ppq.core.RoundingPolicy (quant.py:123-142) /
 *     common.cuh:17-24: 0 HALF_EVEN, 1 HALF_UP, 2 HALF_DOWN, 3 HALF_TOWARDS_ZERO,
 *     4 HALF_FAR_FORM_ZERO, 5 TO_NEAR_INT, 6 UP, 7 DOWN;
 *   - per-channel ops address a contiguous tensor as [outer, num_channel, elem_per_channel]:
 *     channel(i) = (i / elem_per_channel) % num_channel (linear.cu:146, floating.cu:94).
 */
#ifndef PPQ_HIP_H_
#define PPQ_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PPQHIP_OK = 0,
    PPQHIP_ERR_INVALID_VALUE = -1, /* InvalidValueException, common.cuh:41-48 */
    PPQHIP_ERR_UNSUPPORTED = -2,   /* parameter combination the kernels do not implement */
    PPQHIP_ERR_HIP = -3            /* a HIP runtime call failed (launch error, bad pointer ...) */
} ppqhip_status;

#define PPQHIP_ABI_VERSION 4   /* 2: quantile hints (round 3); 3: *_multi LSQ / min-max entry points, quantile sequence (round 4);
                                * 4: ppqhip_minmax_c_multi carries its job table in the kernel arguments (no device table / upload); split per-tensor LSQ
 *    backward (ppqhip_fq_linear_t_bwd_main / ppqhip_lsq_finish_multi) (round 5) */

/* library / device introspection ------------------------------------------------------------- */
const char* ppqhip_last_error(void);
int ppqhip_version(void);                 /* ABI version == PPQHIP_ABI_VERSION; bumped on incompatible change */
int ppqhip_device_arch(char* buf, int n); /* gcnArchName of the current device, e.g. "gfx950:..." */

/* linear (integer) fake quant ---------------------------------------------------------------- */
/* replaces QuantizeTensor_LT, ppq/csrc/cuda/linear.cu:88-130 (CUDA.LinearQuantize_T ffi.py:78-90).
 * out[i] = (clip(round(x[i] / s) + round(o), qmin, qmax) - round(o)) * s ; scale/offset: 1 elem. */
int ppqhip_fq_linear_t(const float* x, const float* scale, const float* offset, float* out,
                       int64_t n, int clip_min, int clip_max, int rounding, void* stream);

/* replaces QuantizeTensor_LC, linear.cu:188-233 (CUDA.LinearQuantize_C ffi.py:92-103). */
int ppqhip_fq_linear_c(const float* x, const float* scale, const float* offset, float* out,
                       int64_t n, int64_t num_channel, int64_t elem_per_channel,
                       int clip_min, int clip_max, int rounding, void* stream);

/* quantise WITHOUT dequantising: PPQLinearQuant_toInt, ppq/quantization/qfunction/linear.py:218-238 (torch ops in the
 * reference; used when an exporter writes integer weights).  q = clamp(ppq_tensor_round(x / s) + o, qmin, qmax)
 * evaluated in float32 with the RAW offset and the float32 rounding formulas of ppq/utils/round.py:9-49, then
 * truncated to out_dtype: 0 = int8, 1 = uint8, 2 = int32 (`out` has n elements of that type).
 * ROUND_TO_NEAR_INT has no tensor form in the reference (round.py:47-49): invalid value here. */
int ppqhip_to_int_t(const float* x, const float* scale, const float* offset, void* out, int64_t n,
                    int clip_min, int clip_max, int rounding, int out_dtype, void* stream);
int ppqhip_to_int_c(const float* x, const float* scale, const float* offset, void* out, int64_t n,
                    int64_t num_channel, int64_t elem_per_channel, int clip_min, int clip_max,
                    int rounding, int out_dtype, void* stream);

/* many tensors, one launch (MI355X-native addition): every job is fake-quantised exactly as
 * ppqhip_fq_linear_c would (a per-tensor job has num_channel = 1, elem_per_channel = n).  Meant for the
 * weights of a graph, which the executor re-quantises on every forward: `jobs` is a HOST array;
 * `device_table` is caller-owned device memory of ppqhip_fq_linear_multi_table_bytes(num_jobs) bytes
 * that holds the converted job table -- pass upload = 1 on the first call and whenever a pointer or
 * shape in `jobs` changed, 0 otherwise (no host-to-device traffic on the steady path). */
typedef struct ppqhip_fq_job {
    const float* x;
    const float* scale;
    const float* offset;
    float* out;
    int64_t n, num_channel, elem_per_channel;
    int32_t clip_min, clip_max;
} ppqhip_fq_job;
int64_t ppqhip_fq_linear_multi_table_bytes(int num_jobs);
int ppqhip_fq_linear_multi(const ppqhip_fq_job* jobs, int num_jobs, int rounding, void* device_table,
                           int upload, void* stream);

/* replaces QuantizeTensor_LT_B, linear.cu:284-324 (CUDA.LinearQuantize_T_B ffi.py:105-118).
 * grad_s (1 elem) is OVERWRITTEN with sum(...) * rsqrt(n * (clip_max - clip_min)). */
int ppqhip_fq_linear_t_bwd(const float* x, const float* scale, const float* offset,
                           const float* grad_y, float* grad_x, float* grad_s, int64_t n,
                           int clip_min, int clip_max, int rounding, void* stream);

/* replaces QuantizeTensor_LC_B, linear.cu:383-433 (CUDA.LinearQuantize_C_B ffi.py:120-134).
 * grad_s (num_channel elems) is OVERWRITTEN; factor rsqrt(n * clip_max) (linear.cu:402). */
int ppqhip_fq_linear_c_bwd(const float* x, const float* scale, const float* offset,
                           const float* grad_y, float* grad_x, float* grad_s, int64_t n,
                           int64_t num_channel, int64_t elem_per_channel,
                           int clip_min, int clip_max, int rounding, void* stream);

/* The same backward in two halves, for callers that back-propagate through SEVERAL per-tensor configs in one sweep (the
 * activation delegators of a block-wise LSQ step): `_main` computes grad_x and leaves ppqhip_fq_linear_t_bwd_partials(n)
 * per-workgroup partial sums of the scale gradient in the caller-owned `partial`; ONE ppqhip_lsq_finish_multi launch at the end
 * of the sweep turns the partials of all jobs into their grad_s (OVERWRITTEN) -- each exactly as ppqhip_fq_linear_t_bwd's own
 * finish would (same lanes, same order, double accumulation: bit-identical).  Job table in the kernel arguments. */
int64_t ppqhip_fq_linear_t_bwd_partials(int64_t n);
int ppqhip_fq_linear_t_bwd_main(const float* x, const float* scale, const float* offset, const float* grad_y, float* grad_x,
                                float* partial, int64_t n, int clip_min, int clip_max, int rounding, void* stream);
typedef struct ppqhip_lsq_finish_job {
    const float* partial;     /* written by ppqhip_fq_linear_t_bwd_main for a tensor of n elements */
    float* grad_s;            /* 1 element */
    int64_t n;
    int32_t clip_min, clip_max;
} ppqhip_lsq_finish_job;
int ppqhip_lsq_finish_multi(const ppqhip_lsq_finish_job* jobs, int num_jobs, void* stream);

/* LSQ backward of MANY per-channel tensors in one launch -- what a block-wise LSQ step needs for all the weights of its
 * block (LearnedStepSizePass.finetune, optim/training.py:728-826, calls CuLSQ_LC.backward -> QuantizeTensor_LC_B once per
 * weight per step, algorithm/training.py:63-90).  Per job exactly ppqhip_fq_linear_c_bwd: grad_x and grad_s OVERWRITTEN.
 * One workgroup owns one channel: no atomics, no memset, fixed summation order (grad_x bit-identical to the per-tensor entry
 * point; grad_s too when outer == 1 and 256 <= elem_per_channel <= 4096, else equal to float-summation tolerance).
 * The job table travels in the kernel arguments (32 jobs per launch): nothing is uploaded, graph-capturable. */
typedef struct ppqhip_lsq_job {
    const float* x;
    const float* scale;
    const float* offset;
    const float* grad_y;
    float* grad_x;
    float* grad_s;
    int64_t n, num_channel, elem_per_channel;
    int32_t clip_min, clip_max;
} ppqhip_lsq_job;
int ppqhip_fq_linear_c_bwd_multi(const ppqhip_lsq_job* jobs, int num_jobs, int rounding, void* stream);

/* low-precision float (FP8 E4M3 / E5M2 / generic E,M) fake quant ------------------------------ */
/* replaces QuantizeTensor_FT, floating.cu:57-75 (CUDA.FloatingQuantize_T ffi.py:272-288);
 * scalar algorithm QuantizeScalarFloating common.cuh:154-226. */
int ppqhip_fq_float_t(const float* x, const float* scale, const float* offset, float* out,
                      int64_t n, int exponent, int mantissa, float clip_min, float clip_max,
                      int rounding, void* stream);

/* replaces QuantizeTensor_FC, floating.cu:102-131 (CUDA.FloatingQuantize_C ffi.py:290-306). */
int ppqhip_fq_float_c(const float* x, const float* scale, const float* offset, float* out,
                      int64_t n, int64_t num_channel, int64_t elem_per_channel,
                      int exponent, int mantissa, float clip_min, float clip_max,
                      int rounding, void* stream);

/* the scale search of the FP8 `floating` observer (DirectMSEObserver, observer/floating.py:88-143), batched
 * (MI355X-native addition): for every row of every job -- a job is `rows` x `row_len` contiguous floats: one row per
 * per-tensor collection, one row per channel of a weight -- and every candidate scale,
 *     out[row][c] = sum over the row of (fake_quant(x; scale = candidates[c], offset = 0) - x)^2      (double)
 * with rows numbered consecutively over the jobs.  ONE launch per 128 jobs; the caller takes the arg-min per row.
 * `jobs`, `candidates` are HOST arrays; `device_table`: ppqhip_float_scale_search_table_bytes(num_jobs) bytes of
 * device scratch; `out`: device, total_rows * num_candidates doubles. */
typedef struct ppqhip_float_search_job {
    const float* x;
    int64_t rows, row_len;
    int32_t exponent, mantissa;
    float clip_min, clip_max;
} ppqhip_float_search_job;
int64_t ppqhip_float_scale_search_table_bytes(int num_jobs);
int ppqhip_float_scale_search(const ppqhip_float_search_job* jobs, int num_jobs, const float* candidates,
                              int num_candidates, int rounding, void* device_table, double* out, void* stream);

/* many tensors, one launch (MI355X-native addition; the FP8 twin of ppqhip_fq_linear_multi): every job is
 * fake-quantised exactly as ppqhip_fq_float_c would (a per-tensor job has num_channel = 1, elem_per_channel = n).
 * Meant for the per-channel FP8 weights the TRT_FP8 policy re-quantises on every forward.  `jobs` is a HOST
 * array; `device_table` is caller-owned device memory of ppqhip_fq_float_multi_table_bytes(num_jobs) bytes; pass
 * upload = 1 on the first call and whenever a pointer, shape or format in `jobs` changed, 0 otherwise. */
typedef struct ppqhip_fq_float_job {
    const float* x;
    const float* scale;
    const float* offset;
    float* out;
    int64_t n, num_channel, elem_per_channel;
    int32_t exponent, mantissa;
    float clip_min, clip_max;
} ppqhip_fq_float_job;
int64_t ppqhip_fq_float_multi_table_bytes(int num_jobs);
int ppqhip_fq_float_multi(const ppqhip_fq_float_job* jobs, int num_jobs, int rounding, void* device_table,
                          int upload, void* stream);

/* replace QuantizeTensor_FT_B / _FC_B, floating.cu:186-221 / :286-331 (ffi.py:308-344; no
 * Python caller in the reference).  grad_s is OVERWRITTEN.  For the per-tensor form pass
 * num_channel = 1, elem_per_channel = n. */
int ppqhip_fq_float_c_bwd(const float* x, const float* scale, const float* offset,
                          const float* grad_y, float* grad_x, float* grad_s, int64_t n,
                          int64_t num_channel, int64_t elem_per_channel,
                          int exponent, int mantissa, float clip_min, float clip_max,
                          int rounding, void* stream);

/* histograms --------------------------------------------------------------------------------- */
/* replaces Histogram_T, sort.cu:91-111 (CUDA.Histogram_T ffi.py:136-145).
 * b = floor(|x| / hist_scale); b > bins-1 is dropped (clip_outliers) or clamped; hist[b] += 1.
 * ACCUMULATES into the caller's int32 hist[num_bins].
 * `workspace`: device scratch of ppqhip_hist_workspace_bytes(n, num_bins) bytes for the two-stage
 * (atomic-free) flush, or NULL to let the library use its own per-stream arena (which allocates on
 * first use and therefore cannot be used while the stream is being captured into a hipGraph). */
int64_t ppqhip_hist_workspace_bytes(int64_t n, int64_t num_bins);
int ppqhip_hist_sym_t(const float* x, int64_t n, float hist_scale, int clip_outliers,
                      int32_t* hist, int64_t num_bins, void* workspace, void* stream);

/* replaces Histogram_Asymmetric_T, sort.cu:141-165 (CUDA.Histogram_Asymmetric_T ffi.py:147-157).
 * hist_scale = (max - min) / bins; b = floor((x - min) / hist_scale). */
int ppqhip_hist_asym_t(const float* x, int64_t n, float min_value, float max_value,
                       int clip_outliers, int32_t* hist, int64_t num_bins, void* workspace,
                       void* stream);

/* replaces Histogram_C, sort.cu:187-218 (CUDA.Histogram_C ffi.py:159-169); hist is
 * [num_channel, num_bins]. */
int ppqhip_hist_sym_c(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                      float hist_scale, int clip_outliers, int32_t* hist, int64_t num_bins,
                      void* stream);

/* MI355X-native extension: the same per-channel histogram with ONE hist_scale PER
 * CHANNEL (device float[num_channel]) -- channel c is binned exactly as ppqhip_hist_sym_t would bin the
 * slice of channel c with hist_scales[c].  Feeds the per-channel KL search the reference refuses
 * (range.py:288-289). */
int ppqhip_hist_sym_c_scales(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                             const float* hist_scales, int clip_outliers, int32_t* hist,
                             int64_t num_bins, void* stream);
/* ... and the asymmetric rule per channel: channel c is binned exactly as ppqhip_hist_asym_t would bin the slice
 * of channel c with (mins[c], maxs[c]) (device float[num_channel] each).  Feeds the per-channel MSE search
 * (TorchMSEObserver raises on PER_CHANNEL, observer/range.py:496-497). */
int ppqhip_hist_asym_c_ranges(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                              const float* mins, const float* maxs, int clip_outliers, int32_t* hist,
                              int64_t num_bins, void* stream);

/* order statistics ---------------------------------------------------------------------------- */
/* replaces Quantile_T, sort.cu:42-59 (CUDA.Quantile ffi.py:171-176): dest[0] = sorted[rn(n*q)],
 * dest[1] = sorted[rn(n*(1-q))], indices clamped to [0, n-1].  Never a sort, never a copy of the data: one
 * streaming pass FILTERS the keys beyond two thresholds into short lists and the order statistics are selected
 * inside them; sides the filter cannot settle go through an exact 3-pass radix select.  The result is exact
 * in every case (quantile.hip).
 * `hint`: NULL, or 8 uint32 words of caller-owned device memory, zero-initialised, that belong to ONE
 * stream of similar tensors (an observer: the same activation, batch after batch).  The library keeps the
 * thresholds that worked in it, and the next call with the same n and q skips the sampling launch that
 * otherwise estimates them (words: [0] hi valid, [1] T_hi key, [2] lo valid, [3] T_lo key, [4] n, [5] k_hi,
 * [6] k_lo, [7] calls settled from the hint; a valid word is 1 in its low byte -- bits 8-9 hold how long a list the side's next
 * threshold is aimed at, raised when a batch used the list up, see quantile.hip: qh_target).  A stale or foreign hint costs time, never correctness -- but the words belong
 * to the call until `stream` has passed it: its launches read and write them, so nothing else may touch them meanwhile.
 * One tensor WITH a hint (16-B aligned, n >= 2^18, at most 8192 wanted keys per side) takes two launches: a filter that
 * leaves every workgroup's keys in its own record, and a select that settles both sides from the records or -- no usable
 * hint yet, a list that came up short -- runs the exact radix passes itself and leaves thresholds for the next call.
 * The FIRST call this process makes on a hint address takes the general sequence (it samples its thresholds: 35 us on 6.4 MB
 * where the exact passes take 73) and leaves the hint the two launches then start from.
 * `workspace` is device scratch of ppqhip_quantile_workspace_bytes(n) bytes (>= 8.7 MB: the records and slots of that path). */
int64_t ppqhip_quantile_workspace_bytes(int64_t n);
int ppqhip_quantile_t(const float* x, int64_t n, float q, float* dest, uint32_t* hint, void* workspace,
                      void* stream);

/* many tensors, ONE launch sequence (7 launches, any number of jobs: the job table is device
 * resident): dest_k[0..1] = (q, 1-q) order statistics of job k exactly as ppqhip_quantile_t.  `jobs` is a
 * HOST array; workspace holds ppqhip_quantile_multi_workspace_bytes(num_jobs, total_elems) bytes,
 * total_elems = the sum of the jobs' n (the filter lists are sized n / 128 per job and side, 16384 keys at
 * least). */
typedef struct ppqhip_quantile_job {
    const float* x;   /* device, n floats */
    float* dest;      /* device, 2 floats */
    uint32_t* hint;   /* device, 8 words, or NULL (see ppqhip_quantile_t) */
    int64_t n;
} ppqhip_quantile_job;
int64_t ppqhip_quantile_multi_workspace_bytes(int num_jobs, int64_t total_elems);
int ppqhip_quantile_t_multi(const ppqhip_quantile_job* jobs, int num_jobs, float q, void* workspace,
                            void* stream);
/* developer aid (tools/quantile_diag.py): where a finished call left its per-job state inside `workspace` --
 * out[0] bytes before job 0's state, [1] words per job, [2] side records, [3] filter record, [4] tickets,
 * [5] byte offset of the job table, [6] / [7] word offsets of the list / tie counters in the filter record. */
void ppqhip_quantile_debug_layout(int64_t* out);
/* the same for the two-launch path of one hinted tensor (quantile.hip "ONE hinted tensor"), in uint32 words: out[0] words of the
 * whole layout, [1] first record, [2] first head, [3] first slot, [4] filter workgroups at most, [5] keys per slot, [6] first word
 * the filter does NOT zero (the header, flags and exact histograms lie below it), [7] keys per head. */
void ppqhip_quantile_hot_layout(int64_t* out);

/* replaces Isotone_T, sort.cu:61-73: dest = [max, 2nd max, min, 2nd min] (with multiplicity).
 * `workspace`: device scratch of ppqhip_quantile_workspace_bytes(n) bytes (shared sizing). */
int ppqhip_isotone_t(const float* x, int64_t n, float* dest, void* workspace, void* stream);

/* range reductions (what TorchMinMaxObserver.observe computes with torch.min/torch.max,
 * ppq/quantization/observer/range.py:86-98; no native twin in the reference) ---------------- */
/* minmax[0] = min(minmax[0], min x), minmax[1] = max(minmax[1], max x): ACCUMULATES, so the
 * caller seeds minmax with {+inf, -inf}.  NaNs are ignored.  `workspace`: device scratch of
 * ppqhip_minmax_workspace_bytes(n) bytes, or NULL (library arena; see ppqhip_hist_sym_t). */
int64_t ppqhip_minmax_workspace_bytes(int64_t n);
int ppqhip_minmax_t(const float* x, int64_t n, float* minmax, void* workspace, void* stream);

/* per channel: mins[c], maxs[c] ACCUMULATE (seed with +inf / -inf). */
int ppqhip_minmax_c(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                    float* mins, float* maxs, void* stream);

/* per-channel min / max of MANY tensors in one launch: all the weights ParameterQuantizePass observes
 * (optim/parameters.py:156-215 -> TorchMinMaxObserver.observe per parameter, observer/range.py:99-107).  Per job exactly
 * ppqhip_minmax_c (bit-identical: min / max are order independent).  `fresh` != 0: mins / maxs are OVERWRITTEN instead of
 * accumulated (the caller need not seed them with +-inf); allowed only when n == num_channel * elem_per_channel (channel
 * axis outermost) and elem_per_channel <= 8192, i.e. one wave owns a channel -- refused otherwise.
 * The job table travels in the kernel arguments (<= 64 jobs per launch, more are chunked): no host-to-device copy, no host
 * synchronisation, legal inside a HIP-graph capture -- the tensors of a calibration forward are fresh every call. */
typedef struct ppqhip_minmax_c_job {
    const float* x;
    float* mins;
    float* maxs;
    int64_t n, num_channel, elem_per_channel;
    int32_t fresh, reserved;
} ppqhip_minmax_c_job;
int ppqhip_minmax_c_multi(const ppqhip_minmax_c_job* jobs, int num_jobs, void* stream);

/* per-channel sums in double, deterministic order: sums[c] += sum of channel c.  The DC term of
 * BiasCorrectionPass.collect_bias (ppq/quantization/optim/training.py:438-448: torch.mean over
 * every dim but the channel one); the caller divides by n / num_channel. */
int ppqhip_channel_sum(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                       double* sums, void* stream);

/* persistent accumulators for repeated observation (MI355X-native; no twin in the reference) -----
 * An observer that sees many batches keeps one accumulator ROW per workgroup resident in HBM:
 *   rows  : int32 [ppqhip_hist_rows()][num_bins], zero-initialised by the caller
 *   slots : float [ppqhip_minmax_slots()][2],     seeded with {+inf, -inf}
 * Each launch adds into its own row / slot with plain stream-ordered read-modify-writes (no atomics,
 * no per-launch reduction kernel); *_finish folds them into hist[num_bins] (+=) / minmax[2]
 * (running min / max) once, when the statistic is needed.  Bin / range rules as above. */
int64_t ppqhip_hist_rows(void);
int ppqhip_hist_sym_t_rows(const float* x, int64_t n, float hist_scale, int clip_outliers,
                           int32_t* rows, int64_t num_bins, void* stream);
int ppqhip_hist_asym_t_rows(const float* x, int64_t n, float min_value, float max_value,
                            int clip_outliers, int32_t* rows, int64_t num_bins, void* stream);
/* many tensors, one launch: job k folds its tensor into ITS OWN slots exactly as ppqhip_minmax_t_slots.
 * `jobs` is a HOST array, copied into the kernel arguments. */
typedef struct ppqhip_minmax_job {
    const float* x;   /* device, n floats */
    float* slots;     /* device, float [ppqhip_minmax_slots()][2] */
    int64_t n;
} ppqhip_minmax_job;
int ppqhip_minmax_t_slots_multi(const ppqhip_minmax_job* jobs, int num_jobs, void* stream);

/* many tensors, one launch: every job bins its tensor into ITS OWN rows buffer exactly as
 * ppqhip_hist_sym_t_rows (asymmetric = 0: p0 = hist_scale) or ppqhip_hist_asym_t_rows
 * (asymmetric = 1: p0 = min_value, p1 = max_value) would.  `jobs` is a HOST array; it is copied into
 * the kernel arguments, so it may be reused as soon as the call returns. */
typedef struct ppqhip_hist_job {
    const float* x;   /* device, n floats */
    int32_t* rows;    /* device, int32 [ppqhip_hist_rows()][num_bins] */
    int64_t n;
    float p0, p1;
} ppqhip_hist_job;
int ppqhip_hist_t_rows_multi(const ppqhip_hist_job* jobs, int num_jobs, int asymmetric,
                             int clip_outliers, int64_t num_bins, void* stream);
int ppqhip_hist_rows_finish(const int32_t* rows, int64_t num_bins, int32_t* hist, void* stream);
/* test aid (tests/test_gpu_kernels.py: all 2^32 float patterns): the histogram kernels compute floor(a / hist_scale) from a
 * reciprocal multiply with a proven exactness test and a true-division fallback (ppq_amd/csrc/hist.hip: Binner::bins4 / bin1), where
 * the reference divides (sort.cu:84-86 / :133-135).  This runs BOTH forms of that device code on every element of x (n % 4 == 0,
 * 16-B aligned) and counts raw bin indices that differ: out[0] packed path, out[1] scalar-tail path, out[2] one offending bit
 * pattern.  `out`: device, 3 x uint64, zeroed by the caller.  asymmetric = 0: a = |x|; 1: a = x - min_value. */
int ppqhip_check_bin_rule(const float* x, int64_t n, float min_value, float hist_scale, int asymmetric, uint64_t* out, void* stream);
int64_t ppqhip_minmax_slots(void);
int ppqhip_minmax_t_slots(const float* x, int64_t n, float* slots, void* stream);
int ppqhip_minmax_slots_finish(const float* slots, float* minmax, void* stream);

/* clipping searches -------------------------------------------------------------------------- */
/* replaces compute_mse_loss, ppq/csrc/cpu/hist_mse.cc:3-28 (CUDA.compute_mse_loss ffi.py:263-270).
 * HOST function on a HOST histogram, float accumulation exactly as the reference. */
float ppqhip_mse_loss_host(const int64_t* hist, int64_t num_bins, int start, int step, int end);

/* Device-side restatement of TorchMSEObserver.hist_to_scale_offset's candidate sweep
 * (observer/range.py:456-520) for `num_hist` histograms at once.  For histogram h:
 *   hist + h*num_bins : int32[num_bins];  hist_scale[h], min_value[h] : float64 (device)
 * Candidate k's loss is accumulated sequentially in float exactly like hist_mse.cc; the first
 * minimum wins.  Writes best[h*4 + {0,1,2,3}] = {start, end, step, candidate index} (int32).
 * `symmetrical` selects the start==0 sweep (range.py:499-509).  `workspace`: device scratch of
 * ppqhip_mse_search_workspace_bytes(num_hist) bytes. */
int64_t ppqhip_mse_search_workspace_bytes(int64_t num_hist);
int ppqhip_mse_search(const int32_t* hist, int64_t num_hist, int64_t num_bins,
                      const double* hist_scale, const double* min_value,
                      int quant_min, int quant_max, int symmetrical, int32_t* best,
                      void* workspace, void* stream);

/* Device-side KL candidate losses of TorchHistObserver.hist_to_scale_offset
 * (observer/range.py:190-282, torch_KL_divergence measure/statistic.py:3-12) for `num_hist`
 * histograms at once.  losses is float64 [num_hist, num_candidates] with
 * num_candidates = ppqhip_kl_num_candidates(num_bins, num_of_bits); candidate j has
 * bin_range = (j + 1) * 2^(num_of_bits-1).  The caller picks the arg-min (the reference uses a
 * stable sort, range.py:271). */
int64_t ppqhip_kl_num_candidates(int64_t num_bins, int num_of_bits);
int ppqhip_kl_losses(const int32_t* hist, int64_t num_hist, int64_t num_bins, int num_of_bits,
                     double* losses, void* stream);

/* training helpers (exported by the reference, no caller in ppq/) ------------------------------ */
/* replace TensorClip_T / TensorClip_C, train.cu:80-113 / :35-78 (kernel + host wrapper each). */
int ppqhip_tensor_clip_t(const float* value, const float* reference, const float* limit,
                         float* out, int64_t n, void* stream);
int ppqhip_tensor_clip_c(const float* value, const float* reference, const float* limit,
                         float* out, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                         void* stream);
/* replace RoundingLoss_LT / _LC (train.cu:115-168, :220-280): out[0] is OVERWRITTEN.
 * num_channel = 0 selects the per-tensor form. */
int ppqhip_rounding_loss(const float* x, const float* scale, const float* offset, float* out,
                         int64_t n, int64_t num_channel, int64_t elem_per_channel,
                         int clip_min, int clip_max, int rounding, void* stream);
/* replace RoundingLoss_LT_B / _LC_B (train.cu:170-218, :282-338); dy is one device float. */
int ppqhip_rounding_loss_bwd(const float* x, const float* dy, const float* scale,
                             const float* offset, float* dx, int64_t n, int64_t num_channel,
                             int64_t elem_per_channel, int clip_min, int clip_max, int rounding,
                             void* stream);

/* profiling aid used by bench.py: when enabled, every kernel launch made through this library
 * on this thread is bracketed by hipEvents on its own stream; ppqhip_prof_collect() synchronises
 * those events and returns, per kernel id, launches / total ms / total algorithmic bytes. */
#define PPQHIP_PROF_MAX_KERNELS 32
typedef struct {
    char name[48];
    int64_t launches;
    double total_ms;
    double total_bytes;
} ppqhip_prof_entry;
int ppqhip_prof_enable(int on);
int ppqhip_prof_collect(ppqhip_prof_entry* entries, int max_entries); /* returns #entries */
/* average elapsed microseconds of `pairs` EMPTY event pairs on `stream` (the bracketing overhead
 * included in every total_ms above); negative on failure. */
double ppqhip_prof_event_overhead_us(void* stream, int pairs);

#ifdef __cplusplus
}
#endif
#endif /* PPQ_HIP_H_ */
