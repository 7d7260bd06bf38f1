// hist.hip -- calibration histograms for gfx950 (replaces the global-atomic kernels of
// ppq/csrc/cuda/sort.cu:75-218).
//
// Bin rule == reference, bit for bit: b = floor(|x| / hist_scale) [sym] or
// floor((x - min) / hist_scale) [asym] with the IEEE quotient and a saturating float->int
// conversion (NaN -> bin 0); b > bins-1 (asym: or b < 0) is dropped (clip_outliers) or clamped.
//
// Design (round 2; round 1's kernel was VALU-issue bound at ~17 VALU instructions per element):
//   * ONE persistent kernel serves every entry point: the work is the concatenated list of TILES
//     (kHistBlock * kHistU float4) of all jobs of the launch, split evenly over a chip-sized grid
//     (kHistWgPerCu co-resident workgroups per CU, so one workgroup's start-up / flush hides behind
//     its neighbour's streaming); a workgroup walks its contiguous tile range, keeps counting in LDS
//     and flushes ONCE per job it touches -- the per-launch row traffic drops from one flush per
//     512 KB chunk to (grid + jobs) flushes;
//   * full tiles only in the main loop: no bounds logic, 16-B loads at a wave-uniform base +
//     constant lane offset, two register tiles that ping-pong (the next tile is in flight while the
//     current one is binned); the (at most one) ragged tail tile of a job takes a masked scalar path;
//   * ~8 VALU per element: packed-f32 (v_pk_mul/add) quotient + exactness test on float2 pairs,
//     floor+convert in ONE instruction (|t| truncation for the symmetric rule, v_cvt_flr for the
//     asymmetric one), predicated ds_add instead of trash-slot selects;
//   * the quotient comes from a reciprocal multiply with a provable exactness test and a rare
//     true-division fallback (quot4), ONE divergent region per float4;
//   * heavily repeated values (ReLU zeros, saturated or already-quantised activations) would
//     serialise the LDS atomic unit (a k-way same-address ds_add costs ~k cycles): each wave keeps
//     one wave-uniform "hot bin" whose hits are counted on the scalar unit (ballot + s_bcnt1);
//   * no global atomics on the hot path: a workgroup adds its LDS histogram into ITS OWN row of the
//     caller's persistent [rows][bins] accumulator (plain stream-ordered read-modify-write), folded
//     once at render; the one-shot entry points store rows to scratch + one reduce launch, or use
//     device atomics when only a handful of workgroups run (small tensors: one launch).
#include <cstdlib>

#include "common.hpp"

namespace ppqhip {

constexpr int kMaxLdsBins = 16384;     // 64 KiB of int32 per copy at most
#ifndef PPQHIP_HIST_LDS_INTS
#define PPQHIP_HIST_LDS_INTS 2048      // copies * bins <= this many ints: ONE copy at >= 2048 bins (more copies cost
                                       // zeroing / flush time and buy nothing: LDS conflicts arise within a wave only)
#endif
#ifndef PPQHIP_HIST_BLOCK
#define PPQHIP_HIST_BLOCK 512
#endif
#ifndef PPQHIP_HIST_WGPC
#define PPQHIP_HIST_WGPC 2
#endif
#ifndef PPQHIP_HIST_U
#define PPQHIP_HIST_U 2               // MI355X sweep (tools/hist_variants.py, profiles/r02_hist_variants.txt): in situ
#endif                                // 512x2/CU: U=2 327 us, U=1 333, U=3 330, U=4 345; 1024x1 339; 256x4 332; 768x1 328
#ifndef PPQHIP_HIST_MIN_TILES
#define PPQHIP_HIST_MIN_TILES 2        // a workgroup is worth launching for at least this many tiles
#endif
#ifndef PPQHIP_HIST_ATOMIC_MAX_WG
#define PPQHIP_HIST_ATOMIC_MAX_WG 128  // one-shot entry points: flush with device atomics up to this grid (B = [1,512,56,56]:
                                       // ONE launch of 7.4 us instead of 8.3 us + a 4.0 us reduce launch; rocprofv3 durations)
#endif
#ifndef PPQHIP_HIST_NT_ELEMS
#define PPQHIP_HIST_NT_ELEMS (48ll << 20)   // streaming (nontemporal) loads beyond cache residency
#endif
constexpr int kLdsBudgetInts = PPQHIP_HIST_LDS_INTS;
constexpr int kHistBlock = PPQHIP_HIST_BLOCK;          // threads per histogram workgroup (multiple of 64)
constexpr int kHistWgPerCu = PPQHIP_HIST_WGPC;         // co-resident workgroups per CU
constexpr int kHistU = PPQHIP_HIST_U;                  // float4 loads in flight per lane and register tile
constexpr int kHistRows = kNumCU * kHistWgPerCu;       // grid limit == rows of a persistent accumulator
constexpr uint32_t kTileVec = (uint32_t)kHistBlock * kHistU;   // float4 per tile
constexpr uint32_t kTileElems = kTileVec * 4;
static_assert(kHistBlock % 64 == 0 && kHistBlock >= 64 && kHistBlock <= 1024, "histogram workgroup: whole waves");

typedef float v2f __attribute__((ext_vector_type(2)));

struct BinRule {
    float a;      // sym: unused; asym: min
    float hs;     // hist_scale
    float rcp;    // RN(1 / hs), see quot4
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
//   (The test is sign symmetric, so the symmetric rule may run it on x * rcp and use |t| afterwards.)
__device__ __forceinline__ bool quotient_is_safe(float t) {
    return __builtin_fabsf(t - __builtin_rintf(t)) >= __builtin_fabsf(t) * 0x1p-22f;
}

// truncation of |t|: == floor for the symmetric rule's non-negative quotient (saturating, NaN -> 0)
__device__ __forceinline__ int f2i_abs(float t) {
    int r;
    asm("v_cvt_i32_f32_e64 %0, |%1|" : "=v"(r) : "v"(t));
    return r;
}
// floor + convert in one instruction; only ever sees finite |t| < 2^23 (everything else took the
// true-division path, which converts with floorf + v_cvt_i32_f32 like round 1)
__device__ __forceinline__ int f2i_floor(float t) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(t));
    return r;
}

// Per-wavefront accumulator over one LDS histogram copy.  ASYM / CLIP specialise the bin rule, HOT
// enables the hot-bin register.
template <bool ASYM, bool CLIP, bool HOT>
struct Binner : WaveBinCounter<ASYM, CLIP, HOT> {
    using Base = WaveBinCounter<ASYM, CLIP, HOT>;
    using Base::h; using Base::last; using Base::hot_bin; using Base::hot_cnt;
    float a0, hs, rcp;

    __device__ __forceinline__ void set_rule(float a, float scale) { a0 = a; hs = scale; rcp = 1.0f / scale; }

    // slots of four values.  Returns bins with the reference's conversion semantics; `exact` lanes
    // (special values) already went through floorf + saturating convert.
    __device__ __forceinline__ void bins4(const float4& v, int (&b)[4]) const {
        v2f x01 = {v.x, v.y}, x23 = {v.z, v.w};
        if (ASYM) { x01 = x01 - a0; x23 = x23 - a0; }
        const v2f t01 = x01 * rcp, t23 = x23 * rcp;
        const v2f r01 = {__builtin_rintf(t01.x), __builtin_rintf(t01.y)};
        const v2f r23 = {__builtin_rintf(t23.x), __builtin_rintf(t23.y)};
        const v2f d01 = t01 - r01, d23 = t23 - r23;
        const v2f e01 = t01 * 0x1p-22f, e23 = t23 * 0x1p-22f;
        // u_k: lane k is NOT provably safe (also true for NaN: the comparison is unordered)
        const bool u0 = !(__builtin_fabsf(d01.x) >= __builtin_fabsf(e01.x));
        const bool u1 = !(__builtin_fabsf(d01.y) >= __builtin_fabsf(e01.y));
        const bool u2 = !(__builtin_fabsf(d23.x) >= __builtin_fabsf(e23.x));
        const bool u3 = !(__builtin_fabsf(d23.y) >= __builtin_fabsf(e23.y));
        b[0] = ASYM ? f2i_floor(t01.x) : f2i_abs(t01.x);
        b[1] = ASYM ? f2i_floor(t01.y) : f2i_abs(t01.y);
        b[2] = ASYM ? f2i_floor(t23.x) : f2i_abs(t23.x);
        b[3] = ASYM ? f2i_floor(t23.y) : f2i_abs(t23.y);
        if (u0 | u1 | u2 | u3) {                // rare: some lane sits next to a bin boundary
            if (u0) b[0] = exact_bin(ASYM ? x01.x : __builtin_fabsf(v.x));
            if (u1) b[1] = exact_bin(ASYM ? x01.y : __builtin_fabsf(v.y));
            if (u2) b[2] = exact_bin(ASYM ? x23.x : __builtin_fabsf(v.z));
            if (u3) b[3] = exact_bin(ASYM ? x23.y : __builtin_fabsf(v.w));
        }
    }
    __device__ __forceinline__ int exact_bin(float a) const { return f2i_sat(__builtin_floorf(a / hs)); }

    __device__ __forceinline__ int bin1(float v) const {
        const float a = ASYM ? (v - a0) : __builtin_fabsf(v);
        const float t = a * rcp;
        if (quotient_is_safe(t)) return ASYM ? f2i_floor(t) : f2i_abs(t);
        return exact_bin(a);
    }

};

// column sums of partial[count][bins] into hist: hist[b] += sum over rows.  One workgroup of 1024 lanes owns kReduceBins
// consecutive bins (a 128-B segment of every row): lane (slice, bin) adds its bin over every kReduceSlices-th row with 16 loads in
// flight, the slices meet in LDS and ONE lane per bin adds the total with a plain read-modify-write -- a bin has exactly one
// owner, so there are no atomics and the integer sum has no order anyway.  (Round 2's form -- 128 workgroups, 16 same-address
// device atomics per bin -- took 4.7 us for the 4 MB of rows of a 512-workgroup launch: 0.85 TB/s.)
constexpr int kReduceBins = 32, kReduceSlices = 32, kReduceBlock = kReduceBins * kReduceSlices;
__global__ __launch_bounds__(kReduceBlock) void hist_reduce_kernel(const int* __restrict__ partial, int count, int bins,
                                                                   int* __restrict__ hist) {
    __shared__ int lds[kReduceSlices][kReduceBins];
    const int lane_bin = threadIdx.x % kReduceBins, slice = threadIdx.x / kReduceBins;
    const int b = blockIdx.x * kReduceBins + lane_bin;
    int total = 0;
    if (b < bins) {
        int i = slice;
        for (; i + 15 * kReduceSlices < count; i += 16 * kReduceSlices) {       // 512 rows = one trip: all 16 loads of a lane in flight
            int v[16];
#pragma unroll
            for (int k = 0; k < 16; k++) v[k] = partial[(size_t)(i + k * kReduceSlices) * bins + b];
#pragma unroll
            for (int k = 0; k < 16; k++) total += v[k];
        }
        for (; i < count; i += kReduceSlices) total += partial[(size_t)i * bins + b];
    }
    lds[slice][lane_bin] = total;
    __syncthreads();
    if (slice == 0 && b < bins) {
        int t = 0;
#pragma unroll
        for (int k = 0; k < kReduceSlices; k++) t += lds[k][lane_bin];
        if (t) hist[b] += t;
    }
}

// ---- the persistent kernel ---------------------------------------------------------------------
constexpr int kMultiMax = 96;          // jobs per launch (3.1 KB of kernel arguments; the limit is 4 KB)
enum FlushMode : int {
    FLUSH_ROWS_ADD = 0,     // job.rows[blockIdx.x][bins] += (persistent accumulator, race free: one row per workgroup)
    FLUSH_ROWS_STORE = 1,   // job.rows[blockIdx.x][bins]  = (scratch for hist_reduce_kernel)
    FLUSH_ATOMIC = 2        // atomicAdd into job.rows[bins] (== the caller's histogram; few workgroups)
};
struct HistJob {                              // 32 B
    const float* x;
    int* rows;
    uint32_t n;
    float a, hs;
    uint32_t first_tile;                      // prefix sum of tiles over the jobs of the launch
};
struct HistJobs {
    HistJob job[kMultiMax];
    uint32_t count;
    uint32_t total_tiles;
    int bins, copies, mode;
};
__host__ __device__ inline uint32_t job_tiles(uint32_t n, bool vec_ok) {
    if (!vec_ok) return (n + kTileElems - 1) / kTileElems;          // every tile through the scalar path
    const uint32_t full = (n >> 2) / kTileVec;
    return full + (n > full * kTileElems ? 1u : 0u);                 // + one ragged tail tile
}

// merge this workgroup's LDS copies into its destination and leave the copies zeroed
__device__ __forceinline__ void lds_hist_flush(int* lds, int bins, int copies, int* __restrict__ dst, int mode) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the inline-assembly ds_adds are invisible to the compiler's counters
    __syncthreads();
    for (int b = threadIdx.x; b < bins; b += kHistBlock) {
        int s = 0;
        for (int c = 0; c < copies; c++) { s += lds[c * bins + b]; lds[c * bins + b] = 0; }
        if (mode == FLUSH_ROWS_STORE) dst[b] = s;
        else if (s) {
            if (mode == FLUSH_ROWS_ADD) dst[b] += s;
            else atomicAdd(&dst[b], s);
        }
    }
    __syncthreads();
}

template <bool ASYM, bool CLIP, bool HOT, bool NT>
__global__ __launch_bounds__(kHistBlock, (kHistBlock * kHistWgPerCu + 255) / 256)
void hist_persistent_kernel(const HistJobs jobs) {
    extern __shared__ int lds[];
    const uint32_t G = gridDim.x, g = blockIdx.x;
    uint32_t t, t_end;
    even_split(jobs.total_tiles, G, g, t, t_end);
    const int bins = jobs.bins, copies = jobs.copies;
    // The LDS copies are zeroed AFTER the first tile's loads have been issued (their latency covers the stores and the
    // barrier; on a 6 MB tensor the kernel is a single latency chain, nothing else hides them).  The barrier is the bare
    // s_barrier behind an LDS-only wait: __syncthreads() would also drain vmcnt, i.e. wait for the loads just issued.
    bool zeroed = false;
    auto zero_lds = [&]() {
        if (zeroed) return;
        for (int i = threadIdx.x; i < copies * bins; i += kHistBlock) lds[i] = 0;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        zeroed = true;
    };
    if (t >= t_end) {                       // more workgroups than tiles: nothing to bin
        if (jobs.mode == FLUSH_ROWS_STORE) for (int b = threadIdx.x; b < bins; b += kHistBlock) jobs.job[0].rows[(size_t)g * bins + b] = 0;
        return;
    }
    uint32_t lo = 0, hi = jobs.count;       // largest lo with first_tile[lo] <= t (wave uniform)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (jobs.job[mid].first_tile <= t) lo = mid; else hi = mid;
    }
    Binner<ASYM, CLIP, HOT> acc;
    acc.init(lds + ((threadIdx.x >> 6) % copies) * bins, bins);

    for (uint32_t j = lo; t < t_end; j++) {
        const HistJob& job = jobs.job[j];
        const uint32_t j_end = (j + 1 < jobs.count) ? jobs.job[j + 1].first_tile : jobs.total_tiles;
        uint32_t k = t - job.first_tile;                                   // tiles [k, k1) of this job
        const uint32_t k1 = min(t_end, j_end) - job.first_tile;
        t = min(t_end, j_end);
        acc.set_rule(job.a, job.hs);
        const float* __restrict__ x = job.x;
        const uint32_t n = job.n;
        const bool vec_ok = (reinterpret_cast<uintptr_t>(x) & 15u) == 0;
        const uint32_t full = vec_ok ? (n >> 2) / kTileVec : 0u;
        const uint32_t kf = min(k1, full);                                 // full tiles [k, kf)
        if (k < kf) {
            const float4* xv = reinterpret_cast<const float4*>(x) + threadIdx.x;
            float4 bufa[kHistU], bufb[kHistU];
            auto fetch = [&](float4 (&buf)[kHistU], uint32_t tile) {
                const float4* p = xv + (size_t)tile * kTileVec;
#pragma unroll
                for (int u = 0; u < kHistU; u++) buf[u] = load4<NT>(p + u * kHistBlock);
            };
            auto consume = [&](const float4 (&buf)[kHistU], uint32_t tile) {
#pragma unroll
                for (int u = 0; u < kHistU; u++) {
                    int b[4];
                    acc.bins4(buf[u], b);
                    if (u == 0) acc.elect(b[0], true);
                    if (CLIP && HOT) acc.commit4_exec(b);
                    else {
                        acc.template commit<true>(b[0], true); acc.template commit<true>(b[1], true);
                        acc.template commit<true>(b[2], true); acc.template commit<true>(b[3], true);
                    }
                }
            };
            // ping-pong between two register tiles; the prefetch index is clamped (wave uniform), so the
            // loads stay unconditional straight-line code: the last tile of the range is fetched twice.
            fetch(bufa, k);
            zero_lds();
            for (;;) {
                fetch(bufb, min(k + 1, kf - 1));
                consume(bufa, k);
                if (++k >= kf) break;
                fetch(bufa, min(k + 1, kf - 1));
                consume(bufb, k);
                if (++k >= kf) break;
            }
        }
        zero_lds();
        for (; k < k1; k++) {             // ragged tail tile (or a tensor that is not 16-B aligned): masked 4-B loads
            const uint32_t e0 = k * kTileElems + threadIdx.x;
#pragma unroll 4
            for (int r = 0; r < 4 * kHistU; r++) {
                const uint32_t i = e0 + r * kHistBlock;
                const bool in = i < n;
                const float a = in ? x[i] : 0.f;
                const int b = acc.bin1(a);
                if ((r & 3) == 0) acc.elect(b, in);
                acc.template commit<false>(b, in);
            }
        }
        acc.flush_hot();
        acc.hot_bin = -1;
        int* dst = jobs.mode == FLUSH_ATOMIC ? job.rows : job.rows + (size_t)g * bins;
        lds_hist_flush(lds, bins, copies, dst, jobs.mode);
    }
}

// ---- the single-tensor kernel for LATENCY-BOUND sizes (one launch on a few MB: B = [1,512,56,56] is 6.4 MB) ------------------
// On such a tensor a launch is one dependent chain -- dispatch, kernel-argument fetch, first-byte latency, binning, flush -- and the
// persistent kernel above spends ~2 us of it on its own generality: its 3 KB job table is fetched in two DEPENDENT scalar-load
// rounds (launch header, then the job the workgroup landed in), its tiles ping-pong two at a time, the LDS copies are zeroed before
// the first load is issued.  This kernel takes ONE tensor with its few arguments by value (one scalar-load round), issues up to
// 8 16-B loads per lane before anything else, zeroes the LDS copy while they fly and bins them in arrival order with the same
// Binner.  Counts are the same integers whatever the traversal.  Flush: device atomics either into the caller's histogram (few
// workgroups: every workgroup touches every 64-B line of the histogram once and the memory-side atomic unit serialises the
// operations on a line, ~15 ns each -- tools/floor_table.py, `floor_atomic`) or into the workgroup's OWN row of a persistent
// accumulator (no contention at all, so the grid can cover every CU; atomics instead of a load-add-store keep the row's
// read latency out of the chain).
// PING (mid-size tensors, tens of MB: a share of many rows): two register tiles of kSmallK rows ping-pong, as in the persistent kernel, but
// still one tensor with its arguments by value, own-row / shared atomics at the end and no job table -- the persistent kernel's
// generality costs ~2-3 us per launch there (Bx8: 13.7 us against a 10.8 us read; profiles/r06_frac_vs_size_first.txt).
template <bool ASYM, bool CLIP, int kSmallK, bool PING = false, bool NT = false>       // kSmallK: loads in flight per lane and tile (2 | 4 | 8)
__global__ __launch_bounds__(kHistBlock) void hist_small_kernel(const float* __restrict__ x, uint32_t n, float a, float hs, int bins,
                                                                 int copies, int* __restrict__ dst, uint32_t row_mod) {
    extern __shared__ int lds[];
    const uint32_t G = gridDim.x, g = blockIdx.x;
    const uint32_t nvec = n >> 2;
    const uint32_t full_rows = nvec / kHistBlock;                     // rows of kHistBlock float4 in which every lane has one
    uint32_t r0, r1;
    even_split(full_rows, G, g, r0, r1);
    const float4* xv = reinterpret_cast<const float4*>(x) + threadIdx.x;
    Binner<ASYM, CLIP, true> acc;
    acc.init(lds + ((threadIdx.x >> 6) % copies) * bins, bins);
    acc.set_rule(a, hs);
    bool zeroed = false;
    auto zero_lds = [&]() {
        if (zeroed) return;
        for (int i = threadIdx.x; i < copies * bins; i += kHistBlock) lds[i] = 0;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // LDS only: the loads stay in flight
        zeroed = true;
    };
    // straight-line loads with a clamped row index (the last row of the share is fetched again instead of branching): a
    // load under `if (k < cnt)` lands in its own basic block and the compiler waits for ALL earlier loads in front of it
    auto fetch = [&](float4 (&buf)[kSmallK], uint32_t r) {
#pragma unroll
        for (int k = 0; k < kSmallK; k++) buf[k] = load4<NT>(xv + (size_t)min(r + (uint32_t)k, r1 - 1) * kHistBlock);
    };
    auto consume = [&](const float4 (&buf)[kSmallK], uint32_t cnt) {      // cnt: workgroup uniform
#pragma unroll
        for (int k = 0; k < kSmallK; k++) {
            if ((uint32_t)k < cnt) {
                int b[4];
                acc.bins4(buf[k], b);
                if ((k & 1) == 0) acc.elect(b[0], true);
                if (CLIP) acc.commit4_exec(b);
                else {
                    acc.template commit<true>(b[0], true); acc.template commit<true>(b[1], true);
                    acc.template commit<true>(b[2], true); acc.template commit<true>(b[3], true);
                }
            }
        }
    };
    if (PING) {
        if (r0 < r1) {
            float4 bufa[kSmallK], bufb[kSmallK];
            uint32_t r = r0;
            fetch(bufa, r);
            zero_lds();
            for (;;) {
                fetch(bufb, r + kSmallK);
                consume(bufa, min((uint32_t)kSmallK, r1 - r));
                r += kSmallK;
                if (r >= r1) break;
                fetch(bufa, r + kSmallK);
                consume(bufb, min((uint32_t)kSmallK, r1 - r));
                r += kSmallK;
                if (r >= r1) break;
            }
        }
    } else {
        for (uint32_t r = r0; r < r1; r += kSmallK) {
            float4 buf[kSmallK];
            fetch(buf, r);
            zero_lds();
            consume(buf, min((uint32_t)kSmallK, r1 - r));
        }
    }
    zero_lds();
    if (g == G - 1) {                                                  // the ragged rest: < kHistBlock float4 + n % 4 elements, masked 4-B loads
        const uint32_t e0 = full_rows * kHistBlock * 4;
        const uint32_t trips = (n - e0 + kHistBlock - 1) / kHistBlock;
        for (uint32_t t = 0; t < trips; t++) {
            const uint32_t i = e0 + t * kHistBlock + threadIdx.x;
            const bool in = i < n;
            const float v = in ? x[i] : 0.f;
            const int b = acc.bin1(v);
            if ((t & 3) == 0) acc.elect(b, in);
            acc.template commit<false>(b, in);
        }
    }
    acc.flush_hot();
    int* row = dst + (size_t)(row_mod ? g % row_mod : 0u) * bins;     // row_mod: 0 one shared histogram, G own rows, R partial rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __syncthreads();
    for (int b = threadIdx.x; b < bins; b += kHistBlock) {
        int sum = 0;
        for (int c = 0; c < copies; c++) sum += lds[c * bins + b];
        if (sum) atomicAdd(&row[b], sum);
    }
}

// ---- exactness self-check of the reciprocal bin rule (test aid: tests/test_gpu_kernels.py sweeps all 2^32 float patterns) ----
// Every float4 goes through Binner::bins4 (the packed reciprocal path with its fallback) AND through Binner::exact_bin (the
// reference's floor(a / hs), sort.cu:84-86) element by element; every element also through bin1 (the scalar-tail form).  RAW bin
// indices are compared (before clipping / clamping: a stronger statement than equal histograms).  out[0] / out[1]: elements whose
// bins4 / bin1 index differs from exact_bin; out[2]: bit pattern of one offending element.
template <bool ASYM>
__global__ __launch_bounds__(kBlock) void check_bin_rule_kernel(const float4* __restrict__ x, uint32_t nvec, float a, float hs,
                                                                unsigned long long* __restrict__ out) {
    Binner<ASYM, true, false> acc;
    acc.init(nullptr, 1);
    acc.set_rule(a, hs);
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < nvec; i += stride) {
        const float4 v = x[i];
        int b[4];
        acc.bins4(v, b);
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int want = acc.exact_bin(ASYM ? (e[k] - a) : __builtin_fabsf(e[k]));
            if (b[k] != want) { atomicAdd(&out[0], 1ull); out[2] = __float_as_uint(e[k]); }
            if (acc.bin1(e[k]) != want) { atomicAdd(&out[1], 1ull); out[2] = __float_as_uint(e[k]); }
        }
    }
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

// per channel (scales: one hist_scale per channel; mins / maxs: one asymmetric range per channel, hist_scale =
// (max - min) / bins as in sort.cu:123): ONE workgroup per (channel, split).  Round 2's kernel took one workgroup per ROW: it
// zeroed, filled and flushed an LDS histogram for every 12.5 KB row of a [32, 512, 56, 56] activation -- 16384 workgroups,
// 13 M device atomics, 4-B loads: 98.9 us = 0.26 of the roofline.  Here a workgroup owns its channel: it streams the channel's
// rows (stride C * epc) as one virtual float4 range with the persistent kernel's machinery (two ping-pong register tiles,
// packed-f32 binning, EXEC-mask LDS commits, hot bin) and adds its histogram to the channel's row of `hist` ONCE -- with
// plain read-modify-writes when it is the channel's only workgroup (splits == 1: no atomics at all).
#ifndef PPQHIP_HIST_C_WGPC
#define PPQHIP_HIST_C_WGPC 1             // workgroups per CU a launch should have before channels stop being split
#endif
template <bool ASYM, bool CLIP, bool NT>
__global__ __launch_bounds__(kHistBlock, (kHistBlock * kHistWgPerCu + 255) / 256)
void hist_c_channel_kernel(const float* __restrict__ x, FastDiv vec_per_row, uint32_t C, uint32_t outer, uint32_t splits,
                           BinRule rule, int copies, int* __restrict__ hist, const float* __restrict__ scales,
                           const float* __restrict__ mins, const float* __restrict__ maxs, int flush_mode) {
    extern __shared__ int lds[];
    const int bins = rule.bins;
    bool zeroed = false;                  // zeroed behind the first tile's loads (see hist_persistent_kernel)
    auto zero_lds = [&]() {
        if (zeroed) return;
        for (int i = threadIdx.x; i < copies * bins; i += kHistBlock) lds[i] = 0;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        zeroed = true;
    };
    const uint32_t c = blockIdx.x % C, sp = blockIdx.x / C;
    const uint32_t vpr = vec_per_row.d;
    // the channel's rows as ONE virtual float4 range [0, outer * vpr); split `sp` owns [v0, v0 + V) of it
    const uint32_t all = outer * vpr;
    uint32_t v0, v_end;
    even_split(all, splits, sp, v0, v_end);
    const uint32_t V = v_end - v0;
    Binner<ASYM, CLIP, true> acc;
    acc.init(lds + ((threadIdx.x >> 6) % copies) * bins, bins);
    if (ASYM) acc.set_rule(mins[c], (maxs[c] - mins[c]) / (float)bins);      // per-channel range (hist_scale as in sort.cu:123)
    else acc.set_rule(0.f, scales != nullptr ? scales[c] : rule.hs);        // per-channel hist_scale
    const float4* xc = reinterpret_cast<const float4*>(x) + (size_t)c * vpr;
    const size_t row_stride = (size_t)C * vpr;
    auto at = [&](uint32_t v) -> const float4* {                           // index inside this split -> address (v < V)
        const uint32_t r = fdiv(v0 + v, vec_per_row);
        return xc + (size_t)r * row_stride + (v0 + v - r * vpr);
    };
    const uint32_t tiles = V / kTileVec;                                   // full tiles: every lane has a float4
    if (tiles > 0) {
        float4 bufa[kHistU], bufb[kHistU];
        auto fetch = [&](float4 (&buf)[kHistU], uint32_t tile) {
#pragma unroll
            for (int u = 0; u < kHistU; u++) buf[u] = load4<NT>(at(tile * kTileVec + u * kHistBlock + threadIdx.x));
        };
        auto consume = [&](const float4 (&buf)[kHistU]) {
#pragma unroll
            for (int u = 0; u < kHistU; u++) {
                int b[4];
                acc.bins4(buf[u], b);
                if (u == 0) acc.elect(b[0], true);
                if (CLIP) acc.commit4_exec(b);
                else {
                    acc.template commit<true>(b[0], true); acc.template commit<true>(b[1], true);
                    acc.template commit<true>(b[2], true); acc.template commit<true>(b[3], true);
                }
            }
        };
        uint32_t k = 0;
        fetch(bufa, k);
        zero_lds();
        for (;;) {
            fetch(bufb, min(k + 1, tiles - 1));
            consume(bufa);
            if (++k >= tiles) break;
            fetch(bufa, min(k + 1, tiles - 1));
            consume(bufb);
            if (++k >= tiles) break;
        }
    }
    {   // the ragged rest (< one tile; ALL of a [1, C, 56, 56] channel: 784 float4 against a tile of 1024): every load issued
        // up front at a clamped index, then whole waves through the EXEC-mask commits and only the straddling wave masked
        const uint32_t v0r = tiles * kTileVec + threadIdx.x;
        float4 rest[kHistU];
        if (V > tiles * kTileVec) {
#pragma unroll
            for (int t = 0; t < kHistU; t++) rest[t] = load4<NT>(at(min(v0r + t * kHistBlock, V - 1)));
        }
        zero_lds();
        if (V > tiles * kTileVec) {
#pragma unroll
            for (int t = 0; t < kHistU; t++) {
                const uint32_t v = v0r + t * kHistBlock;
                const bool in = v < V;
                if (__builtin_amdgcn_ballot_w64(in) == 0ull) continue;            // wave uniform: nothing of this wave left
                int b[4];
                acc.bins4(rest[t], b);
                acc.elect(b[0], in);
                if (CLIP && __builtin_amdgcn_ballot_w64(!in) == 0ull) acc.commit4_exec(b);   // the whole wave holds values
                else {
                    acc.template commit<false>(b[0], in); acc.template commit<false>(b[1], in);
                    acc.template commit<false>(b[2], in); acc.template commit<false>(b[3], in);
                }
            }
        }
    }
    acc.flush_hot();
    lds_hist_flush(lds, bins, copies, hist + (size_t)c * bins, flush_mode);
}

// the same ownership for rows that are not float4-addressable (elem_per_channel % 4 != 0: the 7 x 7 planes of a CNN's last
// stage; unaligned views): 4-B loads over the channel's virtual ELEMENT range -- still one LDS histogram and one flush per
// channel split instead of one device atomic per element (hist_c_global_kernel)
template <bool ASYM, bool CLIP>
__global__ __launch_bounds__(kHistBlock) void hist_c_channel_scalar_kernel(
    const float* __restrict__ x, FastDiv elem_per_row, uint32_t C, uint32_t outer, uint32_t splits, BinRule rule, int copies,
    int* __restrict__ hist, const float* __restrict__ scales, const float* __restrict__ mins, const float* __restrict__ maxs) {
    extern __shared__ int lds[];
    const int bins = rule.bins;
    for (int i = threadIdx.x; i < copies * bins; i += kHistBlock) lds[i] = 0;
    __syncthreads();
    const uint32_t c = blockIdx.x % C, sp = blockIdx.x / C;
    const uint32_t epc = elem_per_row.d;
    const uint32_t all = outer * epc;
    uint32_t v0, v_end;
    even_split(all, splits, sp, v0, v_end);
    const uint32_t V = v_end - v0;
    Binner<ASYM, CLIP, true> acc;
    acc.init(lds + ((threadIdx.x >> 6) % copies) * bins, bins);
    if (ASYM) acc.set_rule(mins[c], (maxs[c] - mins[c]) / (float)bins);
    else acc.set_rule(0.f, scales != nullptr ? scales[c] : rule.hs);
    const float* xc = x + (size_t)c * epc;
    const size_t row_stride = (size_t)C * epc;
    const uint32_t trips = (V + kHistBlock - 1) / kHistBlock;              // block-uniform: the commits use wave ballots
    uint32_t v = threadIdx.x;
    for (uint32_t t = 0; t < trips; t++, v += kHistBlock) {
        const bool in = v < V;
        float a = 0.f;
        if (in) { const uint32_t r = fdiv(v0 + v, elem_per_row); a = xc[(size_t)r * row_stride + (v0 + v - r * epc)]; }
        const int b = acc.bin1(a);
        if ((t & 15u) == 0) acc.elect(b, in);
        acc.template commit<false>(b, in);
    }
    acc.flush_hot();
    lds_hist_flush(lds, bins, copies, hist + (size_t)c * bins, splits == 1 ? FLUSH_ROWS_ADD : FLUSH_ATOMIC);
}

__global__ __launch_bounds__(kBlock) void hist_c_global_kernel(const float* __restrict__ x, uint32_t n,
                                                               FastDiv elem_per_channel, FastDiv num_channel,
                                                               BinRule rule, int* __restrict__ hist,
                                                               const float* __restrict__ scales,
                                                               const float* __restrict__ mins,
                                                               const float* __restrict__ maxs) {
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += stride) {
        const uint32_t row = fdiv(i, elem_per_channel);
        const uint32_t c = row - fdiv(row, num_channel) * num_channel.d;
        BinRule r = rule;
        if (scales != nullptr) r.hs = scales[c];
        if (rule.asym) { r.a = mins[c]; r.hs = (maxs[c] - mins[c]) / (float)rule.bins; }
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
    if (c < 1) c = 1;
    if (c > block / kWave) c = block / kWave;
    return c;
}

static size_t lds_bytes(int bins, int copies) { return sizeof(int) * (size_t)copies * bins; }

static void launch_reduce(const int* partial, int grid, int bins, int32_t* hist, hipStream_t s) {
    hipLaunchKernelGGL(hist_reduce_kernel, dim3((bins + kReduceBins - 1) / kReduceBins), dim3(kReduceBlock), 0, s,
                       partial, grid, bins, hist);
}

static void launch_persistent(const HistJobs& args, int grid, int asym, int clip, bool nt, hipStream_t s) {
    const size_t lds = lds_bytes(args.bins, args.copies);
#define PPQ_LAUNCH_HIST(A, C)                                                                                         \
    do {                                                                                                              \
        if (nt) hipLaunchKernelGGL((hist_persistent_kernel<A, C, true, true>), dim3(grid), dim3(kHistBlock), lds, s, \
                                   args);                                                                             \
        else hipLaunchKernelGGL((hist_persistent_kernel<A, C, true, false>), dim3(grid), dim3(kHistBlock), lds, s,   \
                                args);                                                                                \
    } while (0)
    switch ((asym ? 2 : 0) | (clip ? 1 : 0)) {
        case 0: PPQ_LAUNCH_HIST(false, false); break;
        case 1: PPQ_LAUNCH_HIST(false, true); break;
        case 2: PPQ_LAUNCH_HIST(true, false); break;
        default: PPQ_LAUNCH_HIST(true, true); break;
    }
#undef PPQ_LAUNCH_HIST
}

// grid for `tiles` tiles: every workgroup gets at least PPQHIP_HIST_MIN_TILES of them (twice that once the grid would
// no longer fit the single-launch atomic flush), at most kHistRows workgroups
static int persistent_grid(uint32_t tiles) {
    uint32_t g = tiles / PPQHIP_HIST_MIN_TILES;
    if (g > PPQHIP_HIST_ATOMIC_MAX_WG) g = tiles / (2 * PPQHIP_HIST_MIN_TILES);
#ifdef PPQHIP_DEV_KNOBS                       // measurement builds only (tools/floor_table.py sweeps the grid of the one-shot launch)
    if (const char* e = getenv("PPQHIP_DEV_HIST_WG")) { const int v = atoi(e); if (v > 0) g = (uint32_t)v; }
#endif
    if (g < 1) g = 1;
    const uint32_t cap = (uint32_t)(num_cu() * kHistWgPerCu);      // <= kHistRows: a partitioned device runs a smaller grid
    if (g > cap) g = cap;
    return (int)g;
}

static bool dev_force_atomic() {
#ifdef PPQHIP_DEV_KNOBS
    if (const char* e = getenv("PPQHIP_DEV_HIST_ATOMIC")) return atoi(e) != 0;
#endif
    return false;
}

// one-shot histogram of one tensor, accumulated into hist[bins] (rows == nullptr) or into the caller's
// persistent rows[kHistRows][bins]
#ifndef PPQHIP_HIST_SMALL_ELEMS
#define PPQHIP_HIST_SMALL_ELEMS (4ll << 20)     // single tensors up to 16 MB take hist_small_kernel
#endif
#ifndef PPQHIP_HIST_SMALL_WG_ATOMIC
#define PPQHIP_HIST_SMALL_WG_ATOMIC 112         // one-shot (shared histogram): workgroups at most
#endif
static int dev_knob(const char* name, int fallback) {
#ifdef PPQHIP_DEV_KNOBS
    if (const char* e = getenv(name)) return atoi(e);
#endif
    (void)name;
    return fallback;
}

#ifndef PPQHIP_HIST_STREAM_K
#define PPQHIP_HIST_STREAM_K 2                  // rows per register tile of the ping-pong form
#endif
#ifndef PPQHIP_HIST_STREAM_ELEMS
#define PPQHIP_HIST_STREAM_ELEMS (0x7fffffffll) // single tensors up to here take the ping-pong form of hist_small_kernel when they have own rows
#endif
// mid-size and large single tensors with own rows: hist_small_kernel<.., PING>
// partial rows of the one-shot form: the grid adds with device atomics into row g % R of the zero-kept arena (R rows instead of
// one: 512 workgroups on ONE row serialise on its 128 lines -- floor_atomic 6.5 us against 3.4 us for 8 rows, profiles/r05_floor_table_head.txt),
// this kernel adds the R rows into the caller's histogram and hands the arena back zeroed.  One thread per bin.
__global__ __launch_bounds__(kBlock) void hist_partial_reduce_kernel(int* __restrict__ partial, int R, int bins, int* __restrict__ hist) {
    const int b = blockIdx.x * kBlock + threadIdx.x;
    if (b >= bins) return;
    int v[32];
    const int hv = hist[b];                                                // (in flight with the rows: one round trip)
#pragma unroll
    for (int r = 0; r < 32; r++) v[r] = r < R ? partial[(size_t)r * bins + b] : 0;
    int t = 0;
#pragma unroll
    for (int r = 0; r < 32; r++) { t += v[r]; if (r < R && v[r] != 0) partial[(size_t)r * bins + b] = 0; }
    if (t) hist[b] = hv + t;
}
#ifndef PPQHIP_HIST_ONESHOT_ROWS
#define PPQHIP_HIST_ONESHOT_ROWS 8
#endif

static bool launch_hist_stream(const float* x, int64_t n, BinRule rule, hipStream_t s, int32_t* rows, int32_t* hist = nullptr) {
    const int kk = dev_knob("PPQHIP_DEV_HIST_STREAM", PPQHIP_HIST_STREAM_K);
    int R = 0;
    bool direct = false;
    if (rows == nullptr) {                                                // one-shot: partial rows in the zero-kept arena
        R = dev_knob("PPQHIP_DEV_HIST_ONESHOT", PPQHIP_HIST_ONESHOT_ROWS);
        if (R == 0 || hist == nullptr) return false;
        if (R > 32) R = 32;
        // (measured and not kept: every workgroup adding into the caller's histogram -- 512 adds per line are +3.7 us behind a 34 us
        // stream, a wash against the 4.7 us a dependent reduce launch costs whatever it reads, and +6.7 us on Bx4 / Bx8;
        // profiles/r06_hist_variants.txt.  PPQHIP_DEV_HIST_ONESHOT=-1 in a developer build still selects it.)
        if (R < 0) { direct = true; rows = hist; }
        else {
            rows = (int32_t*)zeroed_arena(s, sizeof(int) * (size_t)R * rule.bins);
            if (rows == nullptr) return false;                            // (first use inside a graph capture: the reduce-launch form)
        }
    }
    if (kk <= 0 || n > PPQHIP_HIST_STREAM_ELEMS || !aligned16(x)) return false;
    const uint32_t full_rows = (uint32_t)((n >> 2) / kHistBlock);
    uint32_t g = full_rows / (2u * (uint32_t)kk);                         // at least two tiles per workgroup
    const uint32_t cap = (uint32_t)(num_cu() * kHistWgPerCu);
    if (g > cap) g = cap;
    g = (uint32_t)dev_knob("PPQHIP_DEV_HIST_WG", (int)g);
    if (g > (uint32_t)kHistRows) g = kHistRows;
    if (g < 1) g = 1;
    const int copies = pick_copies(rule.bins, kHistBlock);
    const bool nt = n >= PPQHIP_HIST_NT_ELEMS;
    const uint32_t row_mod = direct ? 0u : (R ? (uint32_t)R : (uint32_t)kHistRows);
#define PPQ_LAUNCH_HIST_STREAM_K(A, C, K)                                                                             \
    do {                                                                                                              \
        if (nt) hipLaunchKernelGGL((hist_small_kernel<A, C, K, true, true>), dim3(g), dim3(kHistBlock), lds_bytes(rule.bins, copies), s, x, \
                                   (uint32_t)n, rule.a, rule.hs, rule.bins, copies, rows, row_mod);                  \
        else hipLaunchKernelGGL((hist_small_kernel<A, C, K, true, false>), dim3(g), dim3(kHistBlock), lds_bytes(rule.bins, copies), s, x, \
                                (uint32_t)n, rule.a, rule.hs, rule.bins, copies, rows, row_mod);                     \
    } while (0)
#define PPQ_LAUNCH_HIST_STREAM(A, C)                                                                                  \
    do {                                                                                                              \
        if (kk >= 4) PPQ_LAUNCH_HIST_STREAM_K(A, C, 4);                                                               \
        else PPQ_LAUNCH_HIST_STREAM_K(A, C, 2);                                                                       \
    } while (0)
    switch ((rule.asym ? 2 : 0) | (rule.clip ? 1 : 0)) {
        case 0: PPQ_LAUNCH_HIST_STREAM(false, false); break;
        case 1: PPQ_LAUNCH_HIST_STREAM(false, true); break;
        case 2: PPQ_LAUNCH_HIST_STREAM(true, false); break;
        default: PPQ_LAUNCH_HIST_STREAM(true, true); break;
    }
#undef PPQ_LAUNCH_HIST_STREAM
#undef PPQ_LAUNCH_HIST_STREAM_K
    if (R > 0) hipLaunchKernelGGL(hist_partial_reduce_kernel, dim3((rule.bins + kBlock - 1) / kBlock), dim3(kBlock), 0, s, (int*)rows, R, rule.bins, (int*)hist);
    return true;
}

static bool launch_hist_small(const float* x, int64_t n, BinRule rule, int32_t* hist, hipStream_t s, int32_t* rows) {
    if (!dev_knob("PPQHIP_DEV_HIST_SMALL", 1) || n > dev_knob("PPQHIP_DEV_HIST_SMALL_ELEMS", (int)PPQHIP_HIST_SMALL_ELEMS) || !aligned16(x)) return false;
    const uint32_t full_rows = (uint32_t)((n >> 2) / kHistBlock);
    uint32_t g;
    if (rows) g = (full_rows + 1) / 2;                                   // own rows: no contention, two loads per lane
    else g = full_rows / 4;                                              // shared histogram: every workgroup touches every line
    const uint32_t cap = rows ? (uint32_t)(num_cu() * kHistWgPerCu) : (uint32_t)PPQHIP_HIST_SMALL_WG_ATOMIC;
    if (g > cap) g = cap;
    g = (uint32_t)dev_knob("PPQHIP_DEV_HIST_WG", (int)g);
    if (g > (uint32_t)kHistRows) g = kHistRows;
    if (g < 1) g = 1;
    const int copies = pick_copies(rule.bins, kHistBlock);
    int* dst = rows ? rows : hist;
    const uint32_t own = rows ? (uint32_t)kHistRows : 0u;             // (g < kHistRows: g % kHistRows == g)
    const uint32_t share = (full_rows + g - 1) / g;                         // rows of the largest share: all of them in flight when <= 8
#define PPQ_LAUNCH_HIST_SMALL_K(A, C, K)                                                                              \
    hipLaunchKernelGGL((hist_small_kernel<A, C, K>), dim3(g), dim3(kHistBlock), lds_bytes(rule.bins, copies), s, x,  \
                       (uint32_t)n, rule.a, rule.hs, rule.bins, copies, dst, own)
#define PPQ_LAUNCH_HIST_SMALL(A, C)                                                                                   \
    do {                                                                                                              \
        if (share <= 2) PPQ_LAUNCH_HIST_SMALL_K(A, C, 2);                                                             \
        else if (share <= 4) PPQ_LAUNCH_HIST_SMALL_K(A, C, 4);                                                        \
        else PPQ_LAUNCH_HIST_SMALL_K(A, C, 8);                                                                        \
    } while (0)
    switch ((rule.asym ? 2 : 0) | (rule.clip ? 1 : 0)) {
        case 0: PPQ_LAUNCH_HIST_SMALL(false, false); break;
        case 1: PPQ_LAUNCH_HIST_SMALL(false, true); break;
        case 2: PPQ_LAUNCH_HIST_SMALL(true, false); break;
        default: PPQ_LAUNCH_HIST_SMALL(true, true); break;
    }
#undef PPQ_LAUNCH_HIST_SMALL
#undef PPQ_LAUNCH_HIST_SMALL_K
    return true;
}

static int launch_hist_one(const float* x, int64_t n, BinRule rule, int32_t* hist, void* workspace, hipStream_t s,
                           int32_t* rows) {
    if (launch_hist_small(x, n, rule, hist, s, rows)) return PPQHIP_OK;
    if (launch_hist_stream(x, n, rule, s, rows, hist)) return PPQHIP_OK;
    HistJobs args;
    args.count = 1; args.bins = rule.bins; args.copies = pick_copies(rule.bins, kHistBlock);
    HistJob& d = args.job[0];
    d.x = x; d.n = (uint32_t)n; d.a = rule.a; d.hs = rule.hs; d.first_tile = 0;
    args.total_tiles = job_tiles((uint32_t)n, aligned16(x));
    const int grid = persistent_grid(args.total_tiles);
    int* partial = nullptr;
    if (rows) { args.mode = FLUSH_ROWS_ADD; d.rows = rows; }
    else if (grid <= PPQHIP_HIST_ATOMIC_MAX_WG || dev_force_atomic()) { args.mode = FLUSH_ATOMIC; d.rows = hist; }
    else {
        partial = workspace ? (int*)workspace : (int*)scratch(s, sizeof(int) * (size_t)grid * rule.bins);
        if (partial == nullptr) return PPQHIP_ERR_HIP;
        args.mode = FLUSH_ROWS_STORE; d.rows = partial;
    }
    launch_persistent(args, grid, rule.asym, rule.clip, n >= PPQHIP_HIST_NT_ELEMS, s);
    if (partial) launch_reduce(partial, grid, rule.bins, hist, s);
    return PPQHIP_OK;
}

static int launch_hist_t(const float* x, int64_t n, BinRule rule, int32_t* hist, void* workspace, hipStream_t s,
                         int32_t* rows = nullptr) {
    if (rule.bins > kMaxLdsBins) {
        hipLaunchKernelGGL(hist_t_global_kernel, dim3(stream_grid(n, kBlock * 4)), dim3(kBlock), 0, s, x, (uint32_t)n,
                           rule, hist);
        return PPQHIP_OK;
    }
    return launch_hist_one(x, n, rule, hist, workspace, s, rows);
}

static int launch_hist_multi(const ppqhip_hist_job* jobs, int count, int bins, int clip, int asym, hipStream_t s) {
    const int copies = pick_copies(bins, kHistBlock);
    for (int base = 0; base < count; base += kMultiMax) {
        HistJobs args;
        args.count = (uint32_t)((count - base) < kMultiMax ? (count - base) : kMultiMax);
        args.bins = bins; args.copies = copies; args.mode = FLUSH_ROWS_ADD;
        uint32_t tiles = 0;
        int64_t elems = 0;
        for (uint32_t k = 0; k < args.count; k++) {
            const ppqhip_hist_job& src = jobs[base + k];
            HistJob& d = args.job[k];
            d.x = src.x; d.rows = src.rows; d.n = (uint32_t)src.n;
            if (asym) { d.a = src.p0; d.hs = (src.p1 - src.p0) / (float)bins; }     // sort.cu:123
            else { d.a = 0.f; d.hs = src.p0; }
            d.first_tile = tiles;
            tiles += job_tiles(d.n, aligned16(src.x));
            elems += src.n;
        }
        args.total_tiles = tiles;
        launch_persistent(args, persistent_grid(tiles), asym, clip, elems >= PPQHIP_HIST_NT_ELEMS, s);
    }
    return PPQHIP_OK;
}

}  // namespace ppqhip

using namespace ppqhip;

extern "C" {

int64_t ppqhip_hist_workspace_bytes(int64_t n, int64_t num_bins) {
    (void)n;
    if (num_bins <= 0 || num_bins > kMaxLdsBins) return 0;
    return (int64_t)sizeof(int) * kHistRows * num_bins;
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
int64_t ppqhip_hist_rows(void) { return (int64_t)kHistRows; }

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

int ppqhip_check_bin_rule(const float* x, int64_t n, float min_value, float hist_scale, int asymmetric, uint64_t* out, void* stream) {
    if (n <= 0 || n > 0x7fffffffLL || (n & 3) || !aligned16(x) || out == nullptr) {
        set_error("check_bin_rule: n must be a positive multiple of 4 below 2^31, x 16-B aligned, out non-null"); return PPQHIP_ERR_INVALID_VALUE;
    }
    hipStream_t s = (hipStream_t)stream;
    const uint32_t nvec = (uint32_t)(n >> 2);
    if (asymmetric) hipLaunchKernelGGL((check_bin_rule_kernel<true>), dim3(stream_grid(nvec, kBlock)), dim3(kBlock), 0, s, (const float4*)x, nvec,
                                       min_value, hist_scale, (unsigned long long*)out);
    else hipLaunchKernelGGL((check_bin_rule_kernel<false>), dim3(stream_grid(nvec, kBlock)), dim3(kBlock), 0, s, (const float4*)x, nvec,
                            0.f, hist_scale, (unsigned long long*)out);
    return finish_launch("check_bin_rule");
}

int ppqhip_hist_rows_finish(const int32_t* rows, int64_t num_bins, int32_t* hist, void* stream) {
    if (num_bins <= 0 || num_bins > kMaxLdsBins) { set_error("hist_rows_finish: bad bins"); return PPQHIP_ERR_INVALID_VALUE; }
    hipStream_t s = (hipStream_t)stream;
    launch_reduce((const int*)rows, (int)ppqhip_hist_rows(), (int)num_bins, hist, s);
    return finish_launch("hist_rows_finish");
}

static int hist_c_impl(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                       float hist_scale, const float* scales, const float* mins, const float* maxs, int clip_outliers,
                       int32_t* hist, int64_t num_bins, void* stream, const char* what) {
    if (int st = validate(n, num_bins, what)) return st;
    if (num_channel <= 0 || elem_per_channel <= 0 || n % (num_channel * elem_per_channel) != 0) {
        set_error("%s: Kernel Failure, Histogram shape is invalid.", what); return PPQHIP_ERR_INVALID_VALUE;
    }
    hipStream_t s = (hipStream_t)stream;
    const int asym = mins != nullptr;
    LaunchScope scope(K_HIST_SYM_C, 4.0 * (double)n, s);
    BinRule rule = make_rule(0.f, hist_scale, (int)num_bins, clip_outliers ? 1 : 0, asym);
    const FastDiv nc = make_fastdiv((uint32_t)num_channel);
    if (elem_per_channel % 4 == 0 && elem_per_channel >= 64 && aligned16(x) && num_bins <= kMaxLdsBins) {
        const uint32_t C = (uint32_t)num_channel, outer = (uint32_t)(n / (num_channel * elem_per_channel));
        // one workgroup per channel; channels are split over row ranges only while that is what fills the chip
        const uint32_t want_wgs = (uint32_t)num_cu() * PPQHIP_HIST_C_WGPC;
        uint32_t splits = C >= want_wgs ? 1u : (want_wgs + C - 1) / C;
        const uint64_t tiles_per_channel = ((uint64_t)outer * (uint64_t)(elem_per_channel / 4)) / kTileVec;      // a split should have >= 1 full tile
        if (splits > tiles_per_channel) splits = tiles_per_channel > 0 ? (uint32_t)tiles_per_channel : 1u;
        const int copies = pick_copies(rule.bins, kHistBlock);
        // a channel's only workgroup adds with plain read-modify-writes when the launch is HBM bound; on a latency-bound
        // tensor the (uncontended) atomic form keeps the read of the counter row out of the dependent chain
        const int flush_mode = (splits == 1 && n > PPQHIP_HIST_SMALL_ELEMS) ? FLUSH_ROWS_ADD : FLUSH_ATOMIC;
        const bool nt = n >= PPQHIP_HIST_NT_ELEMS;      // streaming loads beyond cache residency, as in the per-tensor kernel
#define PPQ_LAUNCH_HIST_CC(A, CL)                                                                                     \
        do {                                                                                                          \
            if (nt) hipLaunchKernelGGL((hist_c_channel_kernel<A, CL, true>), dim3(C * splits), dim3(kHistBlock), lds_bytes(rule.bins, copies), s, x, \
                           make_fastdiv((uint32_t)(elem_per_channel / 4)), C, outer, splits, rule, copies, hist, scales, mins, maxs, \
                           flush_mode);                                                                               \
            else hipLaunchKernelGGL((hist_c_channel_kernel<A, CL, false>), dim3(C * splits), dim3(kHistBlock), lds_bytes(rule.bins, copies), s, x, \
                           make_fastdiv((uint32_t)(elem_per_channel / 4)), C, outer, splits, rule, copies, hist, scales, mins, maxs, \
                           flush_mode);                                                                               \
        } while (0)
        switch ((asym ? 2 : 0) | (rule.clip ? 1 : 0)) {
            case 0: PPQ_LAUNCH_HIST_CC(false, false); break;
            case 1: PPQ_LAUNCH_HIST_CC(false, true); break;
            case 2: PPQ_LAUNCH_HIST_CC(true, false); break;
            default: PPQ_LAUNCH_HIST_CC(true, true); break;
        }
#undef PPQ_LAUNCH_HIST_CC
    } else if (n / num_channel >= 256 && num_bins <= kMaxLdsBins) {        // >= 256 elements per channel, not float4-addressable
        const uint32_t C = (uint32_t)num_channel, outer = (uint32_t)(n / (num_channel * elem_per_channel));
        const uint32_t want_wgs = (uint32_t)num_cu() * PPQHIP_HIST_C_WGPC;
        uint32_t splits = C >= want_wgs ? 1u : (want_wgs + C - 1) / C;
        const uint64_t trips_per_channel = ((uint64_t)outer * (uint64_t)elem_per_channel) / (kHistBlock * 8);   // >= 8 trips per split
        if (splits > trips_per_channel) splits = trips_per_channel > 0 ? (uint32_t)trips_per_channel : 1u;
        const int copies = pick_copies(rule.bins, kHistBlock);
#define PPQ_LAUNCH_HIST_CS(A, CL)                                                                                     \
        hipLaunchKernelGGL((hist_c_channel_scalar_kernel<A, CL>), dim3(C * splits), dim3(kHistBlock), lds_bytes(rule.bins, copies), s, \
                           x, make_fastdiv((uint32_t)elem_per_channel), C, outer, splits, rule, copies, hist, scales, mins, maxs)
        switch ((asym ? 2 : 0) | (rule.clip ? 1 : 0)) {
            case 0: PPQ_LAUNCH_HIST_CS(false, false); break;
            case 1: PPQ_LAUNCH_HIST_CS(false, true); break;
            case 2: PPQ_LAUNCH_HIST_CS(true, false); break;
            default: PPQ_LAUNCH_HIST_CS(true, true); break;
        }
#undef PPQ_LAUNCH_HIST_CS
    } else {
        hipLaunchKernelGGL(hist_c_global_kernel, dim3(stream_grid(n, kBlock * 4)), dim3(kBlock), 0, s, x, (uint32_t)n,
                           make_fastdiv((uint32_t)elem_per_channel), nc, rule, hist, scales, mins, maxs);
    }
    return finish_launch(what);
}

int ppqhip_hist_sym_c(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                      float hist_scale, int clip_outliers, int32_t* hist, int64_t num_bins, void* stream) {
    return hist_c_impl(x, n, num_channel, elem_per_channel, hist_scale, nullptr, nullptr, nullptr, clip_outliers, hist,
                       num_bins, stream, "hist_sym_c");
}

int ppqhip_hist_sym_c_scales(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                             const float* hist_scales, int clip_outliers, int32_t* hist, int64_t num_bins,
                             void* stream) {
    if (hist_scales == nullptr) { set_error("hist_sym_c_scales: hist_scales is null"); return PPQHIP_ERR_INVALID_VALUE; }
    return hist_c_impl(x, n, num_channel, elem_per_channel, 1.0f, hist_scales, nullptr, nullptr, clip_outliers, hist,
                       num_bins, stream, "hist_sym_c_scales");
}

int ppqhip_hist_asym_c_ranges(const float* x, int64_t n, int64_t num_channel, int64_t elem_per_channel,
                              const float* mins, const float* maxs, int clip_outliers, int32_t* hist,
                              int64_t num_bins, void* stream) {
    if (mins == nullptr || maxs == nullptr) { set_error("hist_asym_c_ranges: mins / maxs is null"); return PPQHIP_ERR_INVALID_VALUE; }
    return hist_c_impl(x, n, num_channel, elem_per_channel, 1.0f, nullptr, mins, maxs, clip_outliers, hist, num_bins,
                       stream, "hist_asym_c_ranges");
}

}  // extern "C"
