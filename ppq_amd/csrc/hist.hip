// hist.hip -- calibration histograms for gfx950 (replaces the global-atomic kernels of
// ppq/csrc/cuda/sort.cu:75-218).
//
// Bin rule == reference, bit for bit: b = floor(|x| / hist_scale) [sym] or
// floor((x - min) / hist_scale) [asym] with the IEEE quotient and a saturating float->int
// conversion (NaN -> bin 0); b > bins-1 (asym: or b < 0) is dropped (clip_outliers) or clamped.
//
// Design (PMC: VALU busy ~63 %, half of the wave cycles waiting on memory, LDS bank conflicts
// irrelevant -- the kernel sits between VALU issue and memory latency, ~17 VALU instructions per
// element, every wave64 VALU instruction occupying its SIMD for ~4 cycles):
//   * one streaming pass, 16-B BRANCH-FREE loads (clamped index; a conditional load costs a branch
//     and an immediate vmcnt(0)), two register tiles that ping-pong: the next tile's kHistU loads
//     per lane are in flight while the current one is binned (and while the LDS histogram is zeroed);
//   * the quotient comes from a reciprocal multiply with a provable exactness test and a rare
//     true-division fallback (quotient_is_safe), ONE divergent region per float4;
//   * every wavefront group owns a private LDS copy of the histogram (ds_add_u32); nothing is
//     predicated: clipped / out-of-range / hot values are redirected to a per-lane "trash" slot
//     behind the histogram, so the ds_add is unconditional and conflict free;
//   * heavily repeated values (ReLU zeros, saturated or already-quantised activations) would
//     serialise the LDS atomic unit (a k-way same-address ds_add costs ~k cycles): each wave keeps
//     one wave-uniform "hot bin" whose hits are counted in a VGPR instead (WaveAcc);
//   * no global atomics on the hot path: every workgroup stores its merged histogram to a scratch
//     row with plain coalesced stores and hist_reduce_kernel adds the column sums into the caller's
//     histogram (hundreds of workgroups x thousands of bins of same-line device atomics serialise
//     at ~12 ns each and would cost more than the streaming pass); observers that see many batches
//     keep the rows resident instead (accumulate mode) and fold them once;
//   * hist_t_multi_kernel bins MANY tensors in one launch (job table in the kernel arguments): what a
//     calibration forward needs, where the tensors are 0.1 .. 100 MB and launch latency dominates.
#include <cstdlib>

#include "common.hpp"

namespace ppqhip {

constexpr int kMaxLdsBins = 16384;     // 64 KiB of int32 per copy at most
constexpr int kLdsBudgetInts = 8192;   // target: copies * bins <= 8192 ints (32 KiB) per workgroup
#ifndef PPQHIP_HIST_UBIG
#define PPQHIP_HIST_UBIG 2   // sweep on MI355X (tools/variants.sh): 2 > 3 > 4 > 1 > 8 with the ping-pong loop
#endif
constexpr int kHistUBig = PPQHIP_HIST_UBIG;           // float4 loads in flight per lane (large tensors)
constexpr int kHistUSmall = 1;         // small tensors: less code to fetch, more workgroups
constexpr int kHistMaxBlock = 1024;    // histogram workgroups: 256 .. 1024 threads (runtime)
constexpr int kTrash = 64;             // per-lane trash slots behind every histogram copy
constexpr int kHotMin = 12;            // lanes that must share the candidate bin to make it hot

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v ? atoi(v) : dflt;
}
static int hist_blocks_per_cu() { static int v = env_int("PPQHIP_HIST_BLOCKS_PER_CU", 1); return v; }
static int hist_block() { static int v = env_int("PPQHIP_HIST_BLOCK", 1024); return v; }
static int hist_hot() { static int v = env_int("PPQHIP_HIST_HOT", 1); return v; }
static int hist_copies() { static int v = env_int("PPQHIP_HIST_COPIES", 0); return v; }

struct BinRule {
    float a;      // sym: unused; asym: min
    float hs;     // hist_scale
    float rcp;    // RN(1 / hs), see quotient_is_safe
    int bins;
    int clip;     // clip_outliers
    int asym;
};
static BinRule make_rule(float a, float hs, int bins, int clip, int asym) {
    BinRule r; r.a = a; r.hs = hs; r.rcp = 1.0f / hs; r.bins = bins; r.clip = clip; r.asym = asym;
    return r;
}

// floor(RN(a / hs)) without the 11-instruction IEEE division on the common path.
//   t = RN(a * RN(1/hs)) differs from q = RN(a / hs) by less than |t| * 0.8 * 2^-22 (two roundings of
//   2^-24 relative for t, one for q), so floor(t) != floor(q) requires an integer within that distance
//   of t -- necessarily rint(t).  When |t - rint(t)| >= |t| * 2^-22 (an exact float test: the
//   difference is exact and the bound is a power-of-two scaling) the floors are provably equal;
//   otherwise (a few lanes in 10^4, plus every inf / NaN / huge outlier / degenerate hs) the lane
//   takes the true division.  The result is therefore ALWAYS the reference's floor(a / hs).
__device__ __forceinline__ bool quotient_is_safe(float t) {
    return __builtin_fabsf(t - __builtin_rintf(t)) >= __builtin_fabsf(t) * 0x1p-22f;
}

// Per-wavefront accumulator.  h points at this wave's histogram copy: bins counters followed by
// kTrash per-lane trash slots.  ASYM / CLIP specialise the bin rule, HOT enables the hot-bin register.
template <bool ASYM, bool CLIP, bool HOT>
struct WaveAcc {
    int* h;
    float a0, hs, rcp;
    int last;        // bins - 1
    int trash;       // bins + lane
    int hot_bin;     // wave-uniform, -1 = none
    int hot_cnt;     // wave-uniform: hits are counted with ballot + s_bcnt1 (no VALU)

    __device__ __forceinline__ void init(int* copy, const BinRule& r) {
        h = copy; a0 = r.a; hs = r.hs; rcp = r.rcp; last = r.bins - 1;
        trash = r.bins + (int)(threadIdx.x & 63);
        hot_bin = -1; hot_cnt = 0;
    }

    __device__ __forceinline__ float arg(float v) const { return ASYM ? (v - a0) : __builtin_fabsf(v); }

    // quotient -> effective slot (bin, or the lane's trash slot when the reference skips the value)
    __device__ __forceinline__ int slot_of(float t) const {
        const int b = f2i_sat(__builtin_floorf(t));
        if (CLIP) return ((unsigned)b > (unsigned)last) ? trash : b;      // b < 0 wraps above `last`
        if (ASYM) return b < 0 ? 0 : (b > last ? last : b);
        return b > last ? last : b;
    }

    // (also tried: v_cvt_flr_i32_f32 for floor+convert -- 1 % and it mis-bins special values --, +inf padding
    //  instead of the `in` select, a select-free path for full tiles: all within noise on MI355X.)
    __device__ __forceinline__ void commit(int slot) {
        if (HOT) {
            const bool hit = slot == hot_bin;
            hot_cnt += __popcll(__ballot(hit));     // wave-uniform count: s_bcnt1 + s_add, no VALU
            slot = hit ? trash : slot;
        }
        atomicAdd(&h[slot], 1);
    }

    __device__ __forceinline__ void add4(const float4& v, bool in) {
        const float ax = arg(v.x), ay = arg(v.y), az = arg(v.z), aw = arg(v.w);
        float tx = ax * rcp, ty = ay * rcp, tz = az * rcp, tw = aw * rcp;
        const bool sx = quotient_is_safe(tx), sy = quotient_is_safe(ty), sz = quotient_is_safe(tz),
                   sw = quotient_is_safe(tw);
        if (!(sx && sy && sz && sw)) {          // rare: some lane sits next to a bin boundary
            if (!sx) tx = ax / hs;
            if (!sy) ty = ay / hs;
            if (!sz) tz = az / hs;
            if (!sw) tw = aw / hs;
        }
        int bx = slot_of(tx), by = slot_of(ty), bz = slot_of(tz), bw = slot_of(tw);
        if (!in) { bx = trash; by = trash; bz = trash; bw = trash; }   // CLIP: padded with +inf
        commit(bx); commit(by); commit(bz); commit(bw);
    }

    __device__ __forceinline__ int slot1(float v) const {
        const float a = arg(v);
        float t = a * rcp;
        if (!quotient_is_safe(t)) t = a / hs;
        return slot_of(t);
    }

    __device__ __forceinline__ void add1(float v, bool in) {
        int b = slot1(v);
        if (!in) b = trash;
        commit(b);
    }

    __device__ __forceinline__ void flush_hot() {
        if (!HOT) return;
        int c = hot_cnt;
        if ((threadIdx.x & 63) == 0 && c != 0 && hot_bin >= 0) atomicAdd(&h[hot_bin], c);
        hot_cnt = 0;
    }

    // Re-elect the hot bin from one sample value per lane; all lanes of the wave call this together.
    // The bin of the first in-range lane becomes hot when at least kHotMin lanes share it.
    __device__ __forceinline__ void elect(float v, bool in) {
        if (!HOT) return;
        const int b = slot1(v);
        const bool valid = in && b <= last;
        const unsigned long long act = __ballot(valid);
        if (act == 0ull) return;
        const int leader = __ffsll((long long)act) - 1;
        const int cand = __builtin_amdgcn_readlane(b, leader);
        if (cand == hot_bin) return;
        const int cnt = __popcll(__ballot(valid && b == cand));
        if (cnt >= kHotMin) { flush_hot(); hot_bin = cand; }
    }
};

__device__ __forceinline__ void lds_hist_zero(int* lds, int total) {
    for (int i = threadIdx.x; i < total; i += blockDim.x) lds[i] = 0;
    __syncthreads();
}

// partial == nullptr: merge the copies and flush the non-zero bins with global atomics (few
// workgroups).  Otherwise store this workgroup's merged histogram to partial[blockIdx.x][bins].
// accumulate == true: partial is a persistent [workgroups][bins] accumulator owned by the caller
// (one row per workgroup, so a plain read-modify-write is race free and stream ordered): nothing is
// reduced per launch, ppqhip_hist_rows_finish sums the rows once, when the histogram is needed.
__device__ __forceinline__ void lds_hist_flush(const int* lds, int bins, int copies, int* __restrict__ hist,
                                               int* __restrict__ partial = nullptr, bool accumulate = false,
                                               uint32_t row = blockIdx.x) {
    __syncthreads();
    const int pitch = bins + kTrash;
    int* dst = partial ? partial + (size_t)row * bins : nullptr;
    for (int b = threadIdx.x; b < bins; b += blockDim.x) {
        int s = 0;
        for (int c = 0; c < copies; c++) s += lds[c * pitch + b];
        if (dst) { if (accumulate) { if (s) dst[b] += s; } else dst[b] = s; }
        else if (s) atomicAdd(&hist[b], s);
    }
}

// column sums of partial[count][bins] into hist.  grid = (ceil(bins / 256), slices): each thread
// sums its bin over one slice of the partial histograms (coalesced 1-KiB rows, 8 loads in flight)
// and issues at most one atomic -- `slices` atomics per bin in total, no long serial chain.
constexpr int kReduceSlices = 16;
__global__ __launch_bounds__(kBlock) void hist_reduce_kernel(const int* __restrict__ partial, int count, int bins,
                                                             int* __restrict__ hist) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= bins) return;
    const int per = (count + gridDim.y - 1) / gridDim.y;
    const int lo = blockIdx.y * per;
    const int hi = min(lo + per, count);
    int acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int i = lo;
    for (; i + 8 <= hi; i += 8) {
#pragma unroll
        for (int k = 0; k < 8; k++) acc[k] += partial[(size_t)(i + k) * bins + b];
    }
    for (; i < hi; i++) acc[0] += partial[(size_t)i * bins + b];
    const int s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    if (s) atomicAdd(&hist[b], s);
}

// Shared streaming loop.  FQ = true additionally writes out = fake_quant(x) (fused calibration step).
template <bool ASYM, bool CLIP, bool HOT, bool FQ, int R, bool NT, int kHistU>
__device__ __forceinline__ void hist_stream(const float* __restrict__ x, uint32_t n, int vec_ok, const BinRule& rule,
                                            int copies, int* lds, float* __restrict__ out, float s, int o, int qmin,
                                            int qmax, int rounding, uint32_t bidx, uint32_t nblk) {
    // bidx / nblk: this workgroup's index among the workgroups that share the tensor (== blockIdx.x /
    // gridDim.x for the single-tensor kernels; a sub-range of the grid for hist_t_multi_kernel)
    const uint32_t stride = nblk * blockDim.x;
    const uint32_t nvec = vec_ok ? (n >> 2) : 0u;
    const float4* xv = reinterpret_cast<const float4*>(x);
    float4* ov = reinterpret_cast<float4*>(out);
    // Every workgroup owns one contiguous chunk of `trips` tiles (blockDim * kHistU float4 each): better
    // DRAM-page / TLB locality than a grid-strided interleave.  The trip count is uniform over the
    // whole grid, so the ballots of WaveAcc::elect always see whole wavefronts.
    const uint32_t bd = blockDim.x;
    const uint32_t tile = bd * kHistU;
    const uint32_t trips = ((nvec + tile - 1) / tile + nblk - 1) / nblk;
    const uint32_t hi = min((bidx + 1) * trips * tile, nvec);
    uint32_t v = bidx * trips * tile + threadIdx.x;
    const uint32_t last_v = nvec ? nvec - 1 : 0u;
    float4 bufa[kHistU], bufb[kHistU];
    auto fetch = [&](float4 (&buf)[kHistU], uint32_t at) {
        // branch-free: out-of-range lanes re-read the tensor's last float4 (their values are ignored
        // through `in`), so the loads stay straight-line code and remain in flight while the
        // previous tile is binned -- a conditional load costs a branch and an immediate vmcnt(0).
#pragma unroll
        for (int k = 0; k < kHistU; k++) buf[k] = load4<NT>(&xv[min(at + k * bd, last_v)]);
    };
    if (trips) fetch(bufa, v);
    const int pitch = rule.bins + kTrash;
    lds_hist_zero(lds, copies * pitch);          // first trip's loads are in flight meanwhile
    WaveAcc<ASYM, CLIP, HOT> acc;
    acc.init(lds + ((threadIdx.x >> 6) % copies) * pitch, rule);
    auto consume = [&](const float4 (&buf)[kHistU], uint32_t at) {
        acc.elect(buf[0].x, at < hi);
#pragma unroll
        for (int k = 0; k < kHistU; k++) {
            const bool in = at + k * bd < hi;
            if (FQ && in) {
                float4 r;
                r.x = fq_linear_scalar<R>(buf[k].x, s, o, qmin, qmax, rounding);
                r.y = fq_linear_scalar<R>(buf[k].y, s, o, qmin, qmax, rounding);
                r.z = fq_linear_scalar<R>(buf[k].z, s, o, qmin, qmax, rounding);
                r.w = fq_linear_scalar<R>(buf[k].w, s, o, qmin, qmax, rounding);
                ov[at + k * bd] = r;
            }
            acc.add4(buf[k], in);
        }
    };
    // two trips per iteration, ping-ponging between the register tiles (no tile-sized copy); the
    // next tile's loads are issued before the current tile is binned.
    for (uint32_t t = 0; t < trips; t += 2, v += 2 * tile) {
        fetch(bufb, v + tile);
        consume(bufa, v);
        if (t + 1 < trips) {
            fetch(bufa, v + 2 * tile);
            consume(bufb, v + tile);
        }
    }
    {   // scalar remainder (the whole tensor when it is not 16-B aligned)
        const uint32_t done = nvec << 2;
        const uint32_t rem = n - done;
        const uint32_t rtrips = (rem + stride - 1) / stride;
        uint32_t i = bidx * blockDim.x + threadIdx.x;
        for (uint32_t t = 0; t < rtrips; t++, i += stride) {
            const bool in = i < rem;
            const float a = in ? x[done + i] : 0.f;
            if (FQ && in) out[done + i] = fq_linear_scalar<R>(a, s, o, qmin, qmax, rounding);
            if ((t & 15u) == 0) acc.elect(a, in);
            acc.add1(a, in);
        }
    }
    acc.flush_hot();
}

template <bool ASYM, bool CLIP, bool HOT, bool NT, int U>
__global__ __launch_bounds__(kHistMaxBlock) void hist_t_lds_kernel(const float* __restrict__ x, uint32_t n, int vec_ok,
                                                                   BinRule rule, int copies, int* __restrict__ hist,
                                                                   int* __restrict__ partial, int accumulate) {
    extern __shared__ int lds[];
    hist_stream<ASYM, CLIP, HOT, false, 0, NT, U>(x, n, vec_ok, rule, copies, lds, nullptr, 0.f, 0, 0, 0, 0, blockIdx.x,
                                                  gridDim.x);
    lds_hist_flush(lds, rule.bins, copies, hist, partial, accumulate != 0);
}

// ---- many tensors, one launch --------------------------------------------------------------------
// A calibration forward observes ~70 activation tensors of 3..100 MB; launched one by one every
// histogram pays ~5 us of launch / fill / drain latency on top of its streaming time.  The observers
// therefore queue their tensors and ONE launch bins them all: the job table travels by value in the
// kernel arguments, job j owns workgroups [first_block[j], first_block[j+1]) and each of them streams
// a contiguous chunk of its tensor into LDS and adds it to its own row of that job's persistent
// rows buffer (same accumulate-mode contract as ppqhip_hist_*_t_rows).
constexpr int kMultiMax = 96;                 // jobs per launch (3.1 KB of kernel arguments; the limit is 4 KB)
#ifndef PPQHIP_MULTI_CHUNK
#define PPQHIP_MULTI_CHUNK (128u << 10)
#endif
#ifndef PPQHIP_MULTI_U
#define PPQHIP_MULTI_U kHistUBig     // MI355X sweep (tools/multi_bench.py): U=2 5.0 TB/s, U=1 4.8; chunk 512 KB > 256 KB, 1 MB, 2 MB
#endif
constexpr uint32_t kMultiChunk = PPQHIP_MULTI_CHUNK;  // elements per workgroup (512 KB): rows RMW is 3 % of the read
struct HistJob {                              // 32 B
    const float* x;
    int* rows;
    uint32_t n;
    float a, hs;
    uint32_t first_block;
};
struct HistJobs {
    HistJob job[kMultiMax];
    uint32_t count;
    int bins, clip, copies;
};

template <bool ASYM, bool CLIP, bool HOT, int U>
__global__ __launch_bounds__(kHistMaxBlock) void hist_t_multi_kernel(const HistJobs jobs) {
    extern __shared__ int lds[];
    uint32_t lo = 0, hi = jobs.count;          // largest lo with first_block[lo] <= blockIdx.x (uniform)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobs.job[mid].first_block <= blockIdx.x) lo = mid; else hi = mid;
    }
    const HistJob& j = jobs.job[lo];
    const uint32_t end = lo + 1 < jobs.count ? jobs.job[lo + 1].first_block : gridDim.x;
    const uint32_t bidx = blockIdx.x - j.first_block, nblk = end - j.first_block;
    BinRule rule;
    rule.a = j.a; rule.hs = j.hs; rule.rcp = 1.0f / j.hs; rule.bins = jobs.bins; rule.clip = jobs.clip; rule.asym = ASYM;
    const int vec_ok = (reinterpret_cast<uintptr_t>(j.x) & 15u) == 0;
    hist_stream<ASYM, CLIP, HOT, false, 0, false, U>(j.x, j.n, vec_ok, rule, jobs.copies, lds, nullptr, 0.f, 0, 0,
                                                     0, 0, bidx, nblk);
    lds_hist_flush(lds, jobs.bins, jobs.copies, nullptr, j.rows, true, bidx);
}

// fused: out = fake_quant(x) (== fq_linear_t) and hist += histogram(x) (== hist_sym_t), one read
template <int R, bool CLIP, bool HOT>
__global__ __launch_bounds__(kHistMaxBlock) void fq_linear_t_hist_kernel(
    const float* __restrict__ x, const float* __restrict__ scale, const float* __restrict__ offset,
    float* __restrict__ out, uint32_t n, int vec_ok, int qmin, int qmax, int rounding, BinRule rule, int copies,
    int* __restrict__ hist, int* __restrict__ partial) {
    extern __shared__ int lds[];
    const float s = scale[0];
    const int o = round_offset(offset[0]);
    hist_stream<false, CLIP, HOT, true, R, false, kHistUBig>(x, n, vec_ok, rule, copies, lds, out, s, o, qmin, qmax, rounding,
                                                             blockIdx.x, gridDim.x);
    lds_hist_flush(lds, rule.bins, copies, hist, partial);
}

// histograms too large for LDS: global atomics (the reference's strategy)
__device__ __forceinline__ bool bin_of(float v, const BinRule& r, int* b_out) {
    const float a = r.asym ? (v - r.a) : __builtin_fabsf(v);
    int b = f2i_sat(__builtin_floorf(a / r.hs));
    const int last = r.bins - 1;
    bool out = b > last;
    if (r.asym) out = out || (b < 0);
    *b_out = b < 0 ? 0 : (b > last ? last : b);
    return !(out && r.clip);
}

__global__ __launch_bounds__(kBlock) void hist_t_global_kernel(const float* __restrict__ x, uint32_t n, BinRule rule,
                                                               int* __restrict__ hist) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        int b;
        if (bin_of(x[i], rule, &b)) atomicAdd(&hist[b], 1);
    }
}

// per channel, long rows: workgroup = (row, chunk); one channel per workgroup -> LDS histogram
template <bool CLIP>
__global__ __launch_bounds__(kBlock) void hist_c_row_kernel(const float* __restrict__ x, uint32_t epc, FastDiv chunks,
                                                            FastDiv num_channel, uint32_t chunk_elems, BinRule rule,
                                                            int copies, int* __restrict__ hist,
                                                            const float* __restrict__ scales) {
    extern __shared__ int lds[];
    const int pitch = rule.bins + kTrash;
    lds_hist_zero(lds, copies * pitch);
    const uint32_t row = fdiv(blockIdx.x, chunks);
    const uint32_t chunk = blockIdx.x - row * chunks.d;
    const uint32_t c = row - fdiv(row, num_channel) * num_channel.d;
    if (scales != nullptr) { rule.hs = scales[c]; rule.rcp = 1.0f / rule.hs; }     // per-channel hist_scale
    WaveAcc<false, CLIP, true> acc;
    acc.init(lds + ((threadIdx.x >> 6) % copies) * pitch, rule);
    const uint32_t lo = chunk * chunk_elems;
    const uint32_t hi = min(lo + chunk_elems, epc);
    const float* xr = x + (size_t)row * epc;
    const uint32_t trips = (hi - lo + kBlock - 1) / kBlock;
    uint32_t j = lo + threadIdx.x;
    for (uint32_t t = 0; t < trips; t++, j += kBlock) {
        const bool in = j < hi;
        const float a = in ? xr[j] : 0.f;
        if ((t & 15u) == 0) acc.elect(a, in);
        acc.add1(a, in);
    }
    acc.flush_hot();
    lds_hist_flush(lds, rule.bins, copies, hist + (size_t)c * rule.bins);
}

__global__ __launch_bounds__(kBlock) void hist_c_global_kernel(const float* __restrict__ x, uint32_t n,
                                                               FastDiv elem_per_channel, FastDiv num_channel,
                                                               BinRule rule, int* __restrict__ hist,
                                                               const float* __restrict__ scales) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t row = fdiv(i, elem_per_channel);
        const uint32_t c = row - fdiv(row, num_channel) * num_channel.d;
        BinRule r = rule;
        if (scales != nullptr) r.hs = scales[c];
        int b;
        if (bin_of(x[i], r, &b)) atomicAdd(&hist[(size_t)c * rule.bins + b], 1);
    }
}

static int validate(int64_t n, int64_t bins, const char* what) {
    if (n <= 0) { set_error("%s: tensor is empty", what); return PPQHIP_ERR_INVALID_VALUE; }
    if (n > 0x7fffffffLL) { set_error("%s: too many elements", what); return PPQHIP_ERR_INVALID_VALUE; }
    if (bins <= 0 || bins > 0x3fffffffLL) {
        set_error("%s: histogram is empty or too large", what); return PPQHIP_ERR_INVALID_VALUE;
    }
    return PPQHIP_OK;
}

static int pick_copies(int bins, int block) {
    int c = kLdsBudgetInts / bins;
    if (hist_copies() > 0) c = hist_copies();
    if (c < 1) c = 1;
    if (c > block / kWave) c = block / kWave;
    return c;
}

static size_t lds_bytes(int bins, int copies) { return sizeof(int) * (size_t)copies * (bins + kTrash); }

static int hist_grid(int64_t n, int u) {
    // one trip (u float4 per lane) per workgroup at least; bounded workgroup count: every
    // workgroup pays LDS zeroing + a flush of `bins` counters
    return stream_grid(n, (int64_t)hist_block() * 4 * u, kNumCU * hist_blocks_per_cu());
}

constexpr int kAtomicFlushMaxBlocks = 8;

// scratch for the two-stage flush, or nullptr when the launch is small enough for atomics
static int* partial_for(int grid, int bins, hipStream_t s, bool* failed, void* workspace) {
    *failed = false;
    if (grid <= kAtomicFlushMaxBlocks) return nullptr;
    if (workspace) return (int*)workspace;
    int* p = (int*)scratch(s, sizeof(int) * (size_t)grid * bins);
    if (p == nullptr) *failed = true;
    return p;
}

static void launch_reduce(const int* partial, int grid, int bins, int32_t* hist, hipStream_t s) {
    hipLaunchKernelGGL(hist_reduce_kernel, dim3((bins + kBlock - 1) / kBlock, kReduceSlices), dim3(kBlock), 0, s,
                       partial, grid, bins, hist);
}

static int launch_hist_t(const float* x, int64_t n, BinRule rule, int32_t* hist, void* workspace, hipStream_t s,
                         int32_t* rows = nullptr) {
    if (rule.bins > kMaxLdsBins) {
        hipLaunchKernelGGL(hist_t_global_kernel, dim3(stream_grid(n, kBlock * 4)), dim3(kBlock), 0, s, x, (uint32_t)n,
                           rule, hist);
        return PPQHIP_OK;
    }
    const int block = hist_block();
    const int copies = pick_copies(rule.bins, block);
    const size_t lds = lds_bytes(rule.bins, copies);
    const int vec_ok = aligned16(x) ? 1 : 0;
    static const int small_elems = env_int("PPQHIP_HIST_SMALL_ELEMS", 48 << 20);   // sweep: U=1 wins up to ~100 MB
    const bool small = n < small_elems;
    const int grid = hist_grid(n, small ? kHistUSmall : kHistUBig);
    bool failed = false;
    const int accumulate = rows != nullptr;
    int* partial = rows ? rows : partial_for(grid, rule.bins, s, &failed, workspace);
    if (failed) return PPQHIP_ERR_HIP;
#define PPQ_LAUNCH_HIST(A, C, H)                                                                                  \
    do {                                                                                                          \
        if (small) hipLaunchKernelGGL((hist_t_lds_kernel<A, C, H, false, kHistUSmall>), dim3(grid), dim3(block), \
                                      lds, s, x, (uint32_t)n, vec_ok, rule, copies, hist, partial, accumulate);  \
        else if (nt) hipLaunchKernelGGL((hist_t_lds_kernel<A, C, H, true, kHistUBig>), dim3(grid), dim3(block),  \
                                        lds, s, x, (uint32_t)n, vec_ok, rule, copies, hist, partial, accumulate);\
        else hipLaunchKernelGGL((hist_t_lds_kernel<A, C, H, false, kHistUBig>), dim3(grid), dim3(block), lds, s, \
                                x, (uint32_t)n, vec_ok, rule, copies, hist, partial, accumulate);                \
    } while (0)
    static const int nt_env = env_int("PPQHIP_HIST_NT", -1);
    const bool nt = nt_env >= 0 ? nt_env != 0 : n >= (48ll << 20);    // streaming loads beyond cache residency
    const int sel = (rule.asym ? 4 : 0) | (rule.clip ? 2 : 0) | (hist_hot() ? 1 : 0);
    switch (sel) {
        case 0: PPQ_LAUNCH_HIST(false, false, false); break;
        case 1: PPQ_LAUNCH_HIST(false, false, true); break;
        case 2: PPQ_LAUNCH_HIST(false, true, false); break;
        case 3: PPQ_LAUNCH_HIST(false, true, true); break;
        case 4: PPQ_LAUNCH_HIST(true, false, false); break;
        case 5: PPQ_LAUNCH_HIST(true, false, true); break;
        case 6: PPQ_LAUNCH_HIST(true, true, false); break;
        default: PPQ_LAUNCH_HIST(true, true, true); break;
    }
#undef PPQ_LAUNCH_HIST
    if (partial && !accumulate) launch_reduce(partial, grid, rule.bins, hist, s);
    return PPQHIP_OK;
}

static int launch_hist_multi(const ppqhip_hist_job* jobs, int count, int bins, int clip, int asym, hipStream_t s) {
    const int block = hist_block();
    const int copies = pick_copies(bins, block);
    const size_t lds = lds_bytes(bins, copies);
    const uint32_t max_rows = (uint32_t)(kNumCU * hist_blocks_per_cu());
    for (int base = 0; base < count; base += kMultiMax) {
        HistJobs args;
        args.count = (uint32_t)((count - base) < kMultiMax ? (count - base) : kMultiMax);
        args.bins = bins; args.clip = clip; args.copies = copies;
        uint32_t blocks = 0;
        for (uint32_t k = 0; k < args.count; k++) {
            const ppqhip_hist_job& src = jobs[base + k];
            HistJob& d = args.job[k];
            d.x = src.x; d.rows = src.rows; d.n = (uint32_t)src.n;
            if (asym) { d.a = src.p0; d.hs = (src.p1 - src.p0) / (float)bins; }     // sort.cu:123
            else { d.a = 0.f; d.hs = src.p0; }
            d.first_block = blocks;
            uint32_t nb = (uint32_t)((src.n + kMultiChunk - 1) / kMultiChunk);
            if (nb > max_rows) nb = max_rows;
            if (nb < 1) nb = 1;
            blocks += nb;
        }
#define PPQ_LAUNCH_MULTI(A, C, H)                                                                              \
        hipLaunchKernelGGL((hist_t_multi_kernel<A, C, H, PPQHIP_MULTI_U>), dim3(blocks), dim3(block), lds, s, args)
        const int sel = (asym ? 4 : 0) | (clip ? 2 : 0) | (hist_hot() ? 1 : 0);
        switch (sel) {
            case 0: PPQ_LAUNCH_MULTI(false, false, false); break;
            case 1: PPQ_LAUNCH_MULTI(false, false, true); break;
            case 2: PPQ_LAUNCH_MULTI(false, true, false); break;
            case 3: PPQ_LAUNCH_MULTI(false, true, true); break;
            case 4: PPQ_LAUNCH_MULTI(true, false, false); break;
            case 5: PPQ_LAUNCH_MULTI(true, false, true); break;
            case 6: PPQ_LAUNCH_MULTI(true, true, false); break;
            default: PPQ_LAUNCH_MULTI(true, true, true); break;
        }
#undef PPQ_LAUNCH_MULTI
    }
    return PPQHIP_OK;
}

}  // namespace ppqhip

using namespace ppqhip;

extern "C" {

int64_t ppqhip_hist_workspace_bytes(int64_t n, int64_t num_bins) {
    (void)n;
    if (num_bins <= 0 || num_bins > kMaxLdsBins) return 0;
    return (int64_t)sizeof(int) * kNumCU * hist_blocks_per_cu() * num_bins;
}

int ppqhip_hist_sym_t(const float* x, int64_t n, float hist_scale, int clip_outliers, int32_t* hist,
                      int64_t num_bins, void* workspace, void* stream) {
    if (int st = validate(n, num_bins, "hist_sym_t")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_HIST_SYM_T, 4.0 * (double)n, s);
    BinRule rule = make_rule(0.f, hist_scale, (int)num_bins, clip_outliers ? 1 : 0, 0);
    if (int st = launch_hist_t(x, n, rule, hist, workspace, s)) return st;
    return finish_launch("hist_sym_t");
}

int ppqhip_hist_asym_t(const float* x, int64_t n, float min_value, float max_value, int clip_outliers,
                       int32_t* hist, int64_t num_bins, void* workspace, void* stream) {
    if (int st = validate(n, num_bins, "hist_asym_t")) return st;
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_HIST_ASYM_T, 4.0 * (double)n, s);
    // float hist_scale = (max - min) / num_of_bins: sort.cu:123 (float / int64 -> float)
    const float hs = (max_value - min_value) / (float)num_bins;
    BinRule rule = make_rule(min_value, hs, (int)num_bins, clip_outliers ? 1 : 0, 1);
    if (int st = launch_hist_t(x, n, rule, hist, workspace, s)) return st;
    return finish_launch("hist_asym_t");
}

/* ---- persistent-row variants (MI355X-native: the observer keeps [rows][bins] resident) ---- */
int64_t ppqhip_hist_rows(void) { return (int64_t)kNumCU * hist_blocks_per_cu(); }

int ppqhip_hist_sym_t_rows(const float* x, int64_t n, float hist_scale, int clip_outliers, int32_t* rows,
                           int64_t num_bins, void* stream) {
    if (int st = validate(n, num_bins, "hist_sym_t_rows")) return st;
    if (num_bins > kMaxLdsBins) { set_error("hist_sym_t_rows: at most %d bins", kMaxLdsBins); return PPQHIP_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_HIST_SYM_T, 4.0 * (double)n, s);
    BinRule rule = make_rule(0.f, hist_scale, (int)num_bins, clip_outliers ? 1 : 0, 0);
    if (int st = launch_hist_t(x, n, rule, nullptr, nullptr, s, rows)) return st;
    return finish_launch("hist_sym_t_rows");
}

int ppqhip_hist_asym_t_rows(const float* x, int64_t n, float min_value, float max_value, int clip_outliers,
                            int32_t* rows, int64_t num_bins, void* stream) {
    if (int st = validate(n, num_bins, "hist_asym_t_rows")) return st;
    if (num_bins > kMaxLdsBins) { set_error("hist_asym_t_rows: at most %d bins", kMaxLdsBins); return PPQHIP_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_HIST_ASYM_T, 4.0 * (double)n, s);
    const float hs = (max_value - min_value) / (float)num_bins;
    BinRule rule = make_rule(min_value, hs, (int)num_bins, clip_outliers ? 1 : 0, 1);
    if (int st = launch_hist_t(x, n, rule, nullptr, nullptr, s, rows)) return st;
    return finish_launch("hist_asym_t_rows");
}

int ppqhip_hist_t_rows_multi(const ppqhip_hist_job* jobs, int num_jobs, int asymmetric, int clip_outliers,
                             int64_t num_bins, void* stream) {
    if (num_jobs <= 0) return PPQHIP_OK;
    if (jobs == nullptr) { set_error("hist_t_rows_multi: jobs is null"); return PPQHIP_ERR_INVALID_VALUE; }
    if (num_bins <= 0 || num_bins > kMaxLdsBins) {
        set_error("hist_t_rows_multi: 1..%d bins", kMaxLdsBins); return PPQHIP_ERR_UNSUPPORTED;
    }
    double bytes = 0.0;
    for (int k = 0; k < num_jobs; k++) {
        if (int st = validate(jobs[k].n, num_bins, "hist_t_rows_multi")) return st;
        if (jobs[k].x == nullptr || jobs[k].rows == nullptr) {
            set_error("hist_t_rows_multi: job %d has a null pointer", k); return PPQHIP_ERR_INVALID_VALUE;
        }
        bytes += 4.0 * (double)jobs[k].n;
    }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(asymmetric ? K_HIST_ASYM_T : K_HIST_SYM_T, bytes, s);
    if (int st = launch_hist_multi(jobs, num_jobs, (int)num_bins, clip_outliers ? 1 : 0, asymmetric ? 1 : 0, s)) return st;
    return finish_launch("hist_t_rows_multi");
}

int ppqhip_hist_rows_finish(const int32_t* rows, int64_t num_bins, int32_t* hist, void* stream) {
    if (num_bins <= 0 || num_bins > kMaxLdsBins) { set_error("hist_rows_finish: bad bins"); return PPQHIP_ERR_INVALID_VALUE; }
    hipStream_t s = (hipStream_t)stream;
    launch_reduce((const int*)rows, (int)ppqhip_hist_rows(), (int)num_bins, hist, s);
    return finish_launch("hist_rows_finish");
}

static int hist_sym_c_impl(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                           float hist_scale, const float* scales, int clip_outliers, int32_t* hist, int64_t num_bins,
                           void* stream) {
    if (int st = validate(n, num_bins, "hist_sym_c")) return st;
    if (num_channel <= 0 || elem_per_channel <= 0 || n % (num_channel * elem_per_channel) != 0) {
        set_error("hist_sym_c: Kernel Failure, Histogram shape is invalid."); return PPQHIP_ERR_INVALID_VALUE;
    }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_HIST_SYM_C, 4.0 * (double)n, s);
    BinRule rule = make_rule(0.f, hist_scale, (int)num_bins, clip_outliers ? 1 : 0, 0);
    const FastDiv nc = make_fastdiv((uint32_t)num_channel);
    if (elem_per_channel >= 1024 && num_bins <= kMaxLdsBins) {
        const int copies = pick_copies(rule.bins, kBlock);
        const uint32_t chunk_elems = 16384;
        const uint32_t chunks = (uint32_t)((elem_per_channel + chunk_elems - 1) / chunk_elems);
        const int64_t rows = n / elem_per_channel;
        if (rule.clip)
            hipLaunchKernelGGL((hist_c_row_kernel<true>), dim3((uint32_t)(rows * chunks)), dim3(kBlock),
                               lds_bytes(rule.bins, copies), s, x, (uint32_t)elem_per_channel, make_fastdiv(chunks), nc,
                               chunk_elems, rule, copies, hist, scales);
        else
            hipLaunchKernelGGL((hist_c_row_kernel<false>), dim3((uint32_t)(rows * chunks)), dim3(kBlock),
                               lds_bytes(rule.bins, copies), s, x, (uint32_t)elem_per_channel, make_fastdiv(chunks), nc,
                               chunk_elems, rule, copies, hist, scales);
    } else {
        hipLaunchKernelGGL(hist_c_global_kernel, dim3(stream_grid(n, kBlock * 4)), dim3(kBlock), 0, s, x, (uint32_t)n,
                           make_fastdiv((uint32_t)elem_per_channel), nc, rule, hist, scales);
    }
    return finish_launch("hist_sym_c");
}

int ppqhip_hist_sym_c(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                      float hist_scale, int clip_outliers, int32_t* hist, int64_t num_bins, void* stream) {
    return hist_sym_c_impl(x, n, num_channel, elem_per_channel, hist_scale, nullptr, clip_outliers, hist, num_bins, stream);
}

int ppqhip_hist_sym_c_scales(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                             const float* hist_scales, int clip_outliers, int32_t* hist, int64_t num_bins,
                             void* stream) {
    if (hist_scales == nullptr) { set_error("hist_sym_c_scales: hist_scales is null"); return PPQHIP_ERR_INVALID_VALUE; }
    return hist_sym_c_impl(x, n, num_channel, elem_per_channel, 1.0f, hist_scales, clip_outliers, hist, num_bins, stream);
}

int ppqhip_fq_linear_t_hist_sym(const float* x, const float* scale, const float* offset, float* out, int64_t n,
                                int clip_min, int clip_max, int rounding, float hist_scale, int clip_outliers,
                                int32_t* hist, int64_t num_bins, void* workspace, void* stream) {
    if (int st = validate(n, num_bins, "fq_linear_t_hist_sym")) return st;
    if (num_bins > kMaxLdsBins) {
        set_error("fq_linear_t_hist_sym: at most %d bins", kMaxLdsBins); return PPQHIP_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_FQ_HIST_FUSED, 8.0 * (double)n, s);
    BinRule rule = make_rule(0.f, hist_scale, (int)num_bins, clip_outliers ? 1 : 0, 0);
    const int block = hist_block();
    const int copies = pick_copies(rule.bins, block);
    const size_t lds = lds_bytes(rule.bins, copies);
    const int vec_ok = (aligned16(x) && aligned16(out)) ? 1 : 0;
    const int grid = hist_grid(n, kHistUBig);
    bool failed;
    int* partial = partial_for(grid, rule.bins, s, &failed, workspace);
    if (failed) return PPQHIP_ERR_HIP;
#define PPQ_LAUNCH_FUSED(R, C, H)                                                                                   \
    hipLaunchKernelGGL((fq_linear_t_hist_kernel<R, C, H>), dim3(grid), dim3(block), lds, s, x, scale, offset, out,  \
                       (uint32_t)n, vec_ok, clip_min, clip_max, rounding, rule, copies, hist, partial)
    const int sel = (rounding == ROUND_HALF_EVEN ? 4 : 0) | (rule.clip ? 2 : 0) | (hist_hot() ? 1 : 0);
    switch (sel) {
        case 0: PPQ_LAUNCH_FUSED(-1, false, false); break;
        case 1: PPQ_LAUNCH_FUSED(-1, false, true); break;
        case 2: PPQ_LAUNCH_FUSED(-1, true, false); break;
        case 3: PPQ_LAUNCH_FUSED(-1, true, true); break;
        case 4: PPQ_LAUNCH_FUSED(ROUND_HALF_EVEN, false, false); break;
        case 5: PPQ_LAUNCH_FUSED(ROUND_HALF_EVEN, false, true); break;
        case 6: PPQ_LAUNCH_FUSED(ROUND_HALF_EVEN, true, false); break;
        default: PPQ_LAUNCH_FUSED(ROUND_HALF_EVEN, true, true); break;
    }
#undef PPQ_LAUNCH_FUSED
    if (partial) launch_reduce(partial, grid, rule.bins, hist, s);
    return finish_launch("fq_linear_t_hist_sym");
}

}  // extern "C"
