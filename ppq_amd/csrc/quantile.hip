// quantile.hip -- the two order statistics of Quantile_T (ppq/csrc/cuda/sort.cu:42-59) for gfx950, many tensors per
// launch sequence, no sort and no copy of the data.
//
// The reference clones the tensor and thrust::sorts it to read sorted[rn(n * q)] and sorted[rn(n * (1 - q))].  Calibration
// asks for extreme ranks (q = 0.9999: the answer is one of the n / 10000 largest / smallest elements), so this file
// FILTERS instead of sorting:
//
//   init      (4 wg / job)    the job table goes to device memory (no 64-job kernel-argument limit: one launch sequence per
//                             forward), the per-job counters are zeroed, the jobs whose thresholds must be estimated are counted
//   sample    (cold jobs)     histogram of the top 12 key bits over ~2 % of the tensor (jittered 64-B granules)
//   filter    (all data)      ONE streaming pass: every key above T_hi / below T_lo is staged in LDS and appended to the
//                             job's key list; a lower bound of the keys EQUAL to a threshold is counted (ties: the zeros after
//                             a ReLU, the sixes after a ReLU6).  The thresholds come from the job's HINT (the thresholds that
//                             worked for the previous batch of the same observer -- calibration sees the same distribution
//                             batch after batch) or, cold, from the sample histogram, computed by every workgroup at the
//                             head of the job's tiles (a separate 1-workgroup launch cost 8 us + a boundary).
//   select A  (1 wg / side)   the list holds the `count` most extreme keys exactly, so the wanted order statistic is the
//                             (k - (n - count))-th smallest listed key (radix select on the key range, 1024 lanes, the list
//                             kept in LDS), or the threshold itself when it lies within the counted ties.  Keeps or drops
//                             the hint for the next batch.
//   F1 F2 F3  (open sides)    exact radix select over the whole tensor (12 + 12 + 8 key bits) for the sides the filter
//                             could not settle (unlucky sample, list overflow, tie on an odd value, unaligned or tiny
//                             tensors): each is an all-data pass whose LAST workgroup (ticket per job) runs the single-
//                             workgroup step that used to be its own launch (and F2 / F3 leave a hint that works: a
//                             threshold with a known, sufficient number of keys beyond it, or ON a heavily tied answer).
//                             With nothing open each returns on one load.
//
// The result is exact in every case; the hint only decides how much is read.  Hot path: 7 launches, 4 of them return on one load;
// ONE small tensor with an extreme q and no hint: 5 launches -- F1 F2 F3 are one launch of one workgroup there
// (quantile_f123_single_kernel).  ONE tensor WITH a hint (what the reference's percentile observer calls per tensor and batch
// through install_into_ppq()): two launches, see "ONE hinted tensor" below -- 26.4 -> 10.3 us on [1,512,56,56], 0.36 -> 0.55 of
// 8 TB/s on 32 x that (rocprofv3 medians, profiles/r06_*).
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <unordered_set>
#include "common.hpp"

namespace ppqhip {

// order-preserving key: ascending uint32 order == ascending float order
__device__ __forceinline__ uint32_t f2key(float f) {
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}
// (explicit unsigned min / max: `max` resolves to the int overload in the host pass of this translation unit)
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
// Wave64 inclusive scan / reductions on the DPP path (row_shr 1,2,4,8 inside each row of 16 lanes, then row_bcast 15 / 31 across
// the rows): six VALU instructions.  The __shfl_up / __shfl_xor forms compile to ds_bpermute_b32 -- a ~120-cycle LDS-crossbar round
// trip each, and a scan is six of them in a dependent chain.
template <typename Op>
__device__ __forceinline__ uint32_t wave_scan_dpp(uint32_t v, const uint32_t identity, Op op) {
#define PPQ_DPP(ctrl, rows) (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, ctrl, rows, 0xf, false)
    v = op(v, PPQ_DPP(0x111, 0xf));        // row_shr:1
    v = op(v, PPQ_DPP(0x112, 0xf));        // row_shr:2
    v = op(v, PPQ_DPP(0x114, 0xf));        // row_shr:4
    v = op(v, PPQ_DPP(0x118, 0xf));        // row_shr:8
    v = op(v, PPQ_DPP(0x142, 0xa));        // row_bcast:15 into rows 1 and 3
    v = op(v, PPQ_DPP(0x143, 0xc));        // row_bcast:31 into rows 2 and 3
#undef PPQ_DPP
    return v;
}
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t v) { return wave_scan_dpp(v, 0u, [](uint32_t a, uint32_t b) { return a + b; }); }
__device__ __forceinline__ uint32_t wave_all_min(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_dpp(v, 0xFFFFFFFFu, [](uint32_t a, uint32_t b) { return a < b ? a : b; }), 63);
}
__device__ __forceinline__ uint32_t wave_all_max(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_dpp(v, 0u, [](uint32_t a, uint32_t b) { return a > b ? a : b; }), 63);
}
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)wave_scan_add(v), 63); }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return wave_all_min(v); }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return wave_all_max(v); }
__device__ __forceinline__ int popc_mask(unsigned long long m) {
    return __builtin_popcount((unsigned)m) + __builtin_popcount((unsigned)(m >> 32));
}

// ---- workspace of one job (uint32 words); the first kQZeroWords are zeroed by the init launch --------------------
constexpr int kQ1 = 4096, kQ2 = 4096, kQ3 = 256;
constexpr uint32_t kQCap = 8192;            // candidate keys kept per side when the selected bucket is small (F2)
constexpr int kOffH1 = 0;                   // hist1[4096]        : key >> 20                            (F1)
constexpr int kOffH2 = kOffH1 + kQ1;        // hist2[2][4096]     : (key >> 8) & 0xFFF | prefix12 match  (F2)
constexpr int kOffH3 = kOffH2 + 2 * kQ2;    // hist3[2][256]      : key & 0xFF        | prefix24 match  (F3)
constexpr int kOffSel = kOffH3 + 2 * kQ3;   // sel[2][8], side 0 = the q order statistic, side 1 = the (1-q) one:
enum { kSTop = 0,     // 12-bit prefix of the bucket that holds the rank
       kSRank = 1,    // rank inside that bucket
       kSMode = 2,    // kModeHist (0, after the zeroing: OPEN) | kModeCompact | kModeDone
       kSCount = 3,   // COMPACT: candidates appended so far
       kSMin = 4,     // HIST: smallest / largest key seen in the bucket (all equal -> done after F2)
       kSMax = 5,
       kSP24 = 6,     // HIST, after F2: 24-bit prefix and the rank inside it (F3)
       kSR24 = 7 };
enum { kModeHist = 0, kModeCompact = 1, kModeDone = 2 };
constexpr int kOffH0 = kOffSel + 16;        // hist0[4096]: key >> 20 of the SAMPLE
constexpr int kOffR0 = kOffH0 + kQ1;        // round0[4096]: those of them that ARE their bucket's round key (0, 6.0, -1.0 ..)
constexpr int kOffSpec = kOffR0 + kQ1;      // spec[48]: thresholds + counters of the filter lists
constexpr int kQShards = 8;                 // a big job's lists are 8 segments with their own counters (see q_job_shards)
enum { kPEnabled = 0,  // 1: the filter ran with the thresholds below
       kPTHi = 1,      // keys > T_hi are appended to the hi list (0xFFFFFFFF: none)
       kPTLo = 2,      // keys < T_lo are appended to the lo list (0: none);  T_lo <= T_hi
       kPHot = 7,                         // 1: the thresholds came from the hint (statistics only)
       kPOvfHi = 9, kPOvfLo = 10,         // some workgroup met more matching keys than it can stage: list incomplete
       kPCnt = 16,                        // cnt[2][8]: keys appended per side and segment (exact unless an overflow flag is set)
       kPTie = 32 };                      // tie[2][8]: LOWER BOUNDS of the number of keys == T_hi / == T_lo (one element in eight is looked at)
constexpr int kOffTick = kOffSpec + 48;     // tick[8]: tiles finished per F pass (the last workgroup runs the pass's tail)
constexpr int kQZeroWords = kOffTick + 8;
constexpr int kOffCand = kQZeroWords;       // cand[2][kQCap]: full keys of the bucket's elements (F2, COMPACT)
constexpr int kQWords = kOffCand + 2 * (int)kQCap;
static_assert(kQZeroWords % 4 == 0 && kQWords % 4 == 0, "16-B granularity");

// the hint of a job (8 words of caller-owned device memory, zero = no knowledge): see ppq_hip.h
enum { kHValidHi = 0, kHTHi = 1, kHValidLo = 2, kHTLo = 3, kHN = 4, kHKHi = 5, kHKLo = 6, kHUses = 7 };

// words of the launch sequence's header
enum { kGCold = 0,     // jobs of the sequence without a usable hint (0: the sample launch returns at once)
       kGOpen = 1,     // sides select A left open (0: F1 / F2 return at once)
       kGOpen3 = 2 };  // sides still open after F2's tail (0: F3 returns at once)
constexpr int kQHeaderWords = 64;

// the key of the smallest-magnitude value of bucket b (3 mantissa bits): the values activations TIE on -- 0 after a ReLU, 6.0
// after a ReLU6 / clip, +-1 after a saturating function -- are of this form
__host__ __device__ inline uint32_t round_key_of_bucket(uint32_t b) { return b >= 0x800u ? (b << 20) : ((b << 20) | 0xFFFFFu); }
// capacity (keys per side) of a job's filter lists; they live behind the fixed parts of all jobs
__host__ __device__ inline uint32_t quantile_spec_cap(uint64_t n) {
    uint64_t c = n / 128;
    if (c < 16384) c = 16384;
    if (c > (1u << 20)) c = 1u << 20;
    return (uint32_t)((c + 31) & ~31ull);           // lists and their 8 segments stay 16-B aligned
}
// Every workgroup of the filter reserves its slice of a list with ONE returning device atomic -- at the same moment as
// all the others (a persistent grid finishes together), and same-address atomics serialise at ~11 ns: 1024 workgroups on
// one counter were a 10-20 us tail behind a 36 us stream.  Jobs big enough to occupy the whole grid split their lists
// into 8 segments (workgroup g appends to segment g % 8: the XCD it runs on); small jobs keep one list (few workgroups,
// and a small list cut in 8 would overflow on channel-structured data).
__host__ __device__ inline uint32_t q_job_shards(uint32_t tiles) { return tiles >= 2048u ? (uint32_t)kQShards : 1u; }
#ifndef PPQHIP_Q_SPEC_MIN_ELEMS
#define PPQHIP_Q_SPEC_MIN_ELEMS (1ll << 18)
#endif
constexpr int64_t kQSpeculateMinElems = PPQHIP_Q_SPEC_MIN_ELEMS;   // smaller sequences go straight to F1..F3

// ---- geometry: a TILE is 1024 float4 (4096 elements); the all-data passes split the concatenated tiles of all jobs
// evenly over a chip-sized grid.  A sample UNIT is what one workgroup of the old sampler read: 4 chunks x 64 granules.
constexpr uint32_t kQTileVec = 1024, kQTileElems = kQTileVec * 4;
constexpr uint32_t kQSampleChunk = 32u << 10;      // elements per sample chunk (128 KB) ..
constexpr uint32_t kQSampleChunksMax = 1024;       // .. at most this many chunks per job (then the chunks grow)
constexpr uint32_t kQSampleStride = 4;             // chunks per unit
constexpr int kQMaxJobs = 1024;                    // jobs per launch sequence (the prefix arrays live in LDS)
__host__ __device__ inline uint32_t q_job_tiles(uint32_t n, bool vec_ok) {
    if (!vec_ok) return (n + kQTileElems - 1) / kQTileElems;
    const uint32_t full = (n >> 2) / kQTileVec;
    return full + (n > full * kQTileElems ? 1u : 0u);
}
__host__ __device__ inline uint32_t q_job_chunks(uint32_t n) {
    uint32_t nb = (uint32_t)(((uint64_t)n + kQSampleChunk - 1) / kQSampleChunk);
    if (nb > kQSampleChunksMax) nb = kQSampleChunksMax;
    return nb < 1 ? 1u : nb;
}
__host__ __device__ inline uint32_t q_job_units(uint32_t n) { return (q_job_chunks(n) + kQSampleStride - 1) / kQSampleStride; }

struct QJob {                 // 64 B, device resident
    const float* x;
    float* dest;
    uint32_t* hint;           // may be null
    uint32_t* ws;             // kQWords words
    uint32_t* spec;           // [2][cap] filter lists (hi, lo)
    uint32_t n, k_hi, k_lo, cap, tiles, units;
};
struct QSeq {                 // what every kernel of a sequence receives
    const QJob* job;
    const uint32_t* first_tile;   // [count] prefix of QJob::tiles
    const uint32_t* first_unit;   // [count] prefix of QJob::units
    uint32_t* header;
    uint32_t* fixed;              // record of the sequence's job 0 (record j: fixed + j * kQWords == job[j].ws)
    uint32_t count, total_tiles, total_units, all_open;
};
// layout of the sequence prefix inside the workspace (bytes)
constexpr size_t kQPrefHeader = 0;
constexpr size_t kQPrefTile = kQPrefHeader + kQHeaderWords * 4;
constexpr size_t kQPrefUnit = kQPrefTile + (size_t)(kQMaxJobs + 4) * 4;
constexpr size_t kQPrefTable = kQPrefUnit + (size_t)(kQMaxJobs + 4) * 4;
constexpr size_t kQPrefBytes = kQPrefTable + (size_t)kQMaxJobs * sizeof(QJob);
static_assert(sizeof(QJob) == 64 && kQPrefTable % 16 == 0 && kQPrefBytes % 16 == 0, "alignment of the prefix");

// Tensor pointers come out of the device-resident job table, so the compiler cannot tell they are global memory and would
// emit FLAT loads -- which tick both vmcnt and lgkmcnt and return out of order with LDS traffic, so every wait becomes
// vmcnt(0) and the ping-pong prefetch of the streaming loops is lost.  These loads name the address space.
typedef __attribute__((address_space(1))) const v4f* gv4f_ptr;
typedef __attribute__((address_space(1))) const float* gf32_ptr;
template <bool NT>
__device__ __forceinline__ float4 gload4(const float4* p) {
    gv4f_ptr g = (gv4f_ptr)p;
    const v4f t = NT ? __builtin_nontemporal_load(g) : *g;
    return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ float gload1(const float* p) { return *(gf32_ptr)p; }

__device__ __forceinline__ bool aligned16_dev(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ bool hint_valid(const uint32_t* __restrict__ hint, uint32_t n, uint32_t k_hi, uint32_t k_lo) {
    if (hint == nullptr) return false;
    // (the valid words' low byte: the two-launch path of one tensor keeps its list-length level in bits 8-9, see qh_target)
    return (hint[kHValidHi] & 0xFFu) == 1u && (hint[kHValidLo] & 0xFFu) == 1u && hint[kHN] == n && hint[kHKHi] == k_hi && hint[kHKLo] == k_lo;
}

// ---- block-wide helpers (THREADS = blockDim.x, a multiple of 64) --------------------------------------------------
// exclusive prefix of v over the workgroup + the total; scratch: THREADS / 64 words.  All threads call this.
template <int THREADS>
__device__ __forceinline__ void block_scan_excl(uint32_t v, uint32_t* scratch, uint32_t& excl, uint32_t& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t inc = wave_scan_add(v);
    __syncthreads();                  // scratch may still be read from a previous call
    if (lane == 63) scratch[wid] = inc;
    __syncthreads();
    uint32_t woff = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; w++) {
        const uint32_t s = scratch[w];
        woff += w < wid ? s : 0u;
        tot += s;
    }
    excl = woff + inc - v;
    total = tot;
}

// find the bin of `hist[0..nbins)` that holds rank k (0-based) and the rank inside it; nbins in {256, 4096}, THREADS threads:
// thread t owns `per` consecutive bins (threads past the last bin own none).  Result: sel[0], sel[1] (LDS), valid after
// the call for every thread.  scratch: THREADS / 64 words.
template <int THREADS>
__device__ void select_bin(const uint32_t* __restrict__ hist, int nbins, uint32_t k, uint32_t* scratch, uint32_t* sel) {
    constexpr int kMaxPer = kQ1 / THREADS;                          // 16 (256 threads) or 4 (1024)
    const int per = nbins >= THREADS ? nbins / THREADS : 1;
    const int t = threadIdx.x;
    const bool owner = t * per < nbins;
    uint32_t mine[kMaxPer];
    uint32_t local = 0;
#pragma unroll
    for (int j = 0; j < kMaxPer; j++) {
        mine[j] = (owner && j < per) ? hist[t * per + j] : 0u;
        local += mine[j];
    }
    uint32_t excl, total;
    block_scan_excl<THREADS>(local, scratch, excl, total);
    const uint32_t kk = k < total ? k : (total ? total - 1 : 0u);   // k < n always; guard anyway
    if (total == 0u && t == 0) { sel[0] = 0u; sel[1] = 0u; }
    if (kk >= excl && kk < excl + local) {
        uint32_t run = excl;
        int j = 0;
#pragma unroll
        for (int jj = 0; jj < kMaxPer - 1; jj++) {
            if (jj < per - 1 && j == jj && run + mine[jj] <= kk) { run += mine[jj]; j = jj + 1; }
        }
        sel[0] = (uint32_t)(t * per + j);
        sel[1] = kk - run;
    }
    __syncthreads();
}

// ---- the walk every all-data kernel shares -------------------------------------------------------------------------
// The prefix array goes to LDS once (one coalesced load), then every lookup is an LDS binary search.
__device__ __forceinline__ void load_prefix(uint32_t* lds, const uint32_t* __restrict__ g, uint32_t count) {
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) lds[i] = g[i];
    __syncthreads();
}
__device__ __forceinline__ uint32_t find_job(const uint32_t* lds, uint32_t count, uint32_t t) {
    uint32_t lo = 0, hi = count;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (lds[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;
}

// tiles [k0, k1) of one job, 256 threads: on_tile(sample, valid) once per 16 elements of a lane (ballot-safe: the trip
// counts are wave-uniform), on_elem(value, valid) for every slot.  The rare-path kernels (F1..F3) stream with this.
template <typename FT, typename FE>
__device__ __forceinline__ void walk_job_tiles(const float* __restrict__ x, uint32_t n, uint32_t k0, uint32_t k1, FT on_tile, FE on_elem) {
    const bool vec_ok = aligned16_dev(x);
    const uint32_t full = vec_ok ? (n >> 2) / kQTileVec : 0u;
    const uint32_t kf = umin(k1, full);
    uint32_t k = k0;
    for (; k < kf; k++) {
        const float4* p = reinterpret_cast<const float4*>(x) + (size_t)k * kQTileVec + threadIdx.x;
        float4 a[4];
#pragma unroll
        for (int u = 0; u < 4; u++) a[u] = gload4<false>(p + u * kBlock);
        on_tile(a[0].x, true);
#pragma unroll
        for (int u = 0; u < 4; u++) { on_elem(a[u].x, true); on_elem(a[u].y, true); on_elem(a[u].z, true); on_elem(a[u].w, true); }
    }
    for (; k < k1; k++) {                 // ragged tail tile / unaligned tensor: masked 4-B loads
        const uint32_t e0 = k * kQTileElems + threadIdx.x;
#pragma unroll 4
        for (int r = 0; r < 16; r++) {
            const uint32_t i = e0 + r * kBlock;
            const bool in = i < n;
            const float a = in ? gload1(x + i) : 0.f;
            if ((r & 3) == 0) on_tile(a, in);
            on_elem(a, in);
        }
    }
}

// "last workgroup done": every thread's global writes of this job are drained, one lane publishes them (agent-scope
// release) and adds the workgroup's tiles to the job's ticket; the workgroup that completes the count acquires and runs
// the tail.  Only the rare F passes pay this (a release is an L2 write-back per workgroup on this 8-XCD part).
__device__ __forceinline__ bool job_ticket(uint32_t* tick, uint32_t mine, uint32_t total, uint32_t* flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t old = __hip_atomic_fetch_add(tick, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool last = old + mine == total;
        if (last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *flag = last ? 1u : 0u;
    }
    __syncthreads();
    const bool last = *flag != 0u;
    __syncthreads();
    return last;
}

// ---- init ------------------------------------------------------------------------------------------------------------
constexpr uint32_t kQInitSplit = 4;        // workgroups per job
constexpr int kQInitMax = 96;              // jobs per init launch (3.8 KB of kernel arguments)
struct QUpload {                           // 40 B
    const float* x;
    float* dest;
    uint32_t* hint;
    uint32_t n, k_hi, k_lo, pad;
};
struct QInitArgs {
    QUpload e[kQInitMax];
    uint8_t* prefix;            // sequence prefix (header, prefix arrays, table)
    uint32_t* fixed;            // per-job fixed parts of job `base` .. (kQWords words each)
    uint32_t* spec;             // filter lists of job `base` ..
    uint32_t count, base, tile_base, unit_base;
};

__global__ __launch_bounds__(kBlock) void quantile_init_kernel(const QInitArgs a) {
    __shared__ uint32_t red[4][kBlock / kWave];
    const uint32_t b = blockIdx.x / kQInitSplit, part = blockIdx.x % kQInitSplit, t = threadIdx.x;
    {   // every workgroup zeroes its quarter of the job's record; the first of the four also writes the table entry
        uint4* z = reinterpret_cast<uint4*>(a.fixed + (size_t)b * kQWords);
        constexpr uint32_t nz = (uint32_t)kQZeroWords / 4, per = (nz + kQInitSplit - 1) / kQInitSplit;
        for (uint32_t i = part * per + t; i < umin(nz, (part + 1) * per); i += kBlock) z[i] = make_uint4(0u, 0u, 0u, 0u);
        if (part != 0) return;
    }
    // prefix sums of (tiles, units, list words) over the jobs before this one; cold jobs of the whole chunk
    uint32_t tiles = 0, units = 0, words = 0, cold = 0;
    if (t < a.count) {
        const QUpload& e = a.e[t];
        if (t < b) {
            tiles = q_job_tiles(e.n, aligned16_dev(e.x));
            units = q_job_units(e.n);
            words = 2u * quantile_spec_cap(e.n);
        }
        if (b == 0) cold = hint_valid(e.hint, e.n, e.k_hi, e.k_lo) ? 0u : 1u;
    }
    tiles = wave_sum_u32(tiles); units = wave_sum_u32(units); words = wave_sum_u32(words); cold = wave_sum_u32(cold);
    if ((t & 63u) == 0) { red[0][t >> 6] = tiles; red[1][t >> 6] = units; red[2][t >> 6] = words; red[3][t >> 6] = cold; }
    __syncthreads();
    uint32_t* header = reinterpret_cast<uint32_t*>(a.prefix + kQPrefHeader);
    uint32_t* ws = a.fixed + (size_t)b * kQWords;
    if (t == 0) {
        tiles = units = words = cold = 0;
        for (int w = 0; w < kBlock / kWave; w++) { tiles += red[0][w]; units += red[1][w]; words += red[2][w]; cold += red[3][w]; }
        const QUpload& e = a.e[b];
        QJob j;
        j.x = e.x; j.dest = e.dest; j.hint = e.hint; j.ws = ws; j.spec = a.spec + words;
        j.n = e.n; j.k_hi = e.k_hi; j.k_lo = e.k_lo; j.cap = quantile_spec_cap(e.n);
        j.tiles = q_job_tiles(e.n, aligned16_dev(e.x)); j.units = q_job_units(e.n);
        reinterpret_cast<QJob*>(a.prefix + kQPrefTable)[a.base + b] = j;
        reinterpret_cast<uint32_t*>(a.prefix + kQPrefTile)[a.base + b] = a.tile_base + tiles;
        reinterpret_cast<uint32_t*>(a.prefix + kQPrefUnit)[a.base + b] = a.unit_base + units;
        if (b == 0) {
            if (a.base == 0) { header[kGCold] = cold; header[kGOpen] = 0u; header[kGOpen3] = 0u; }
            else header[kGCold] += cold;            // stream-ordered behind the previous chunk's launch
        }
    }
}

// ---- sample (cold jobs only) -----------------------------------------------------------------------------------------
// One unit = 4 chunks x 64 granules of 64 B (4 float4: one memory sector each), one granule every chunk / 64 with a hashed
// offset inside its window -- NOT the contiguous head of the chunk: activations are channel-structured ([N, C, H, W]; the
// extreme quantile lives in a few channels), a contiguous 4 KB run sees one channel's rows and on real networks the
// thresholds came out wrong often enough to send half of the data through the fall-back passes; the jitter breaks any
// period the channel stride shares with the window.
__global__ __launch_bounds__(kBlock) void quantile_sample_kernel(const QSeq s) {
    __shared__ uint32_t fu[kQMaxJobs];
    __shared__ uint32_t h[kQ1], hr[kQ1];
    if (s.header[kGCold] == 0u) return;                    // every job has its hint: nothing to estimate
    const uint32_t G = gridDim.x, g = blockIdx.x;
    uint32_t u, u_end;
    even_split(s.total_units, G, g, u, u_end);
    if (u >= u_end) return;
    load_prefix(fu, s.first_unit, s.count);
    for (int i = threadIdx.x; i < kQ1; i += kBlock) { h[i] = 0; hr[i] = 0; }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    // every lane of the wave calls this; when >= 16 lanes share the first lane's bucket they count with ONE ds_add (after a
    // ReLU half of the sample is the same key, and 256 same-address LDS atomics per value made this launch 3x longer)
    auto count = [&](float f, bool valid) {
        const uint32_t key = f2key(f), top = key >> 20;
        const bool round = key == round_key_of_bucket(top);
        const uint32_t lead = (uint32_t)__builtin_amdgcn_readfirstlane((int)top);
        const bool same = valid && top == lead;
        const unsigned long long m = __builtin_amdgcn_ballot_w64(same);
        if (__builtin_popcountll(m) < 16) {            // wave-uniform: no tie worth aggregating
            if (valid) { atomicAdd(&h[top], 1u); if (round) atomicAdd(&hr[top], 1u); }
            return;
        }
        const unsigned long long mr = __builtin_amdgcn_ballot_w64(same && round);
        if (same) {
            if (lane == __builtin_ctzll(m)) {
                atomicAdd(&h[top], (uint32_t)__builtin_popcountll(m));
                if (mr) atomicAdd(&hr[top], (uint32_t)__builtin_popcountll(mr));
            }
        } else if (valid) {
            atomicAdd(&h[top], 1u);
            if (round) atomicAdd(&hr[top], 1u);
        }
    };
    for (uint32_t j = find_job(fu, s.count, u); u < u_end; j++) {
        const QJob job = s.job[j];
        const uint32_t j_end = (j + 1 < s.count) ? fu[j + 1] : s.total_units;
        uint32_t uu = u - fu[j];
        const uint32_t uu1 = umin(u_end, j_end) - fu[j];
        u = umin(u_end, j_end);
        // unaligned tensors: no sample -> no thresholds -> F1..F3
        if (!aligned16_dev(job.x) || job.n < kQTileElems || hint_valid(job.hint, job.n, job.k_hi, job.k_lo)) continue;
        const uint32_t nblk = q_job_chunks(job.n);
        const uint32_t nvec = job.n >> 2;
        const uint32_t tiles = (nvec + kQTileVec - 1) / kQTileVec;
        const uint32_t per = (tiles + nblk - 1) / nblk;
        const uint32_t chunk_vec = per * kQTileVec, window = chunk_vec / 64u, granule = threadIdx.x >> 2, sub = threadIdx.x & 3u;
        for (; uu < uu1; uu++) {
            const uint32_t bidx = uu * kQSampleStride;
            float4 a[kQSampleStride];
            bool ok[kQSampleStride];
#pragma unroll
            for (uint32_t c = 0; c < kQSampleStride; c++) {
                const uint32_t lo = (bidx + c) * chunk_vec;
                uint32_t v = lo + threadIdx.x;                              // tiny chunks: the contiguous head
                if (window >= 8u) {
                    const uint32_t slots = window / 4u;                     // 64-B aligned positions inside the window
                    const uint32_t hh = ((granule * 2654435761u) ^ ((bidx + c) * 40503u + 0x9E3779B9u)) >> 9;
                    v = lo + granule * window + (hh % slots) * 4u + sub;
                }
                ok[c] = bidx + c < nblk && v < nvec && v < lo + chunk_vec;
                a[c] = gload4<false>(reinterpret_cast<const float4*>(job.x) + (ok[c] ? v : 0u));
            }
#pragma unroll
            for (uint32_t c = 0; c < kQSampleStride; c++) {
                count(a[c].x, ok[c]); count(a[c].y, ok[c]); count(a[c].z, ok[c]); count(a[c].w, ok[c]);
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < kQ1; i += kBlock) {
            if (h[i]) { atomicAdd(&job.ws[kOffH0 + i], h[i]); h[i] = 0; }
            if (hr[i]) { atomicAdd(&job.ws[kOffR0 + i], hr[i]); hr[i] = 0; }
        }
        __syncthreads();
    }
}

// ---- thresholds from the sample histogram (cold jobs; every workgroup of the filter that touches the job) ----------
// hi side: bucket b = the LARGEST with (sample count of top >= b) >= need; the threshold lies INSIDE it -- at its round key
// when at least half of the bucket's sample is that one value (ties), else interpolated so that about 1.6x the still missing
// count lies above it (the density falls towards the extreme, a linear share would come up short).  lo side mirrored.
// Deterministic in its inputs: every workgroup arrives at the same thresholds.  out[0] = enabled, out[1] = T_hi, out[2] = T_lo.
#ifndef PPQHIP_Q_NEED
#define PPQHIP_Q_NEED 2.0
#endif
#ifndef PPQHIP_Q_TAKE
#define PPQHIP_Q_TAKE 2.5
#endif
template <int THREADS>
__device__ __noinline__ void thresholds_from_sample(const uint32_t* __restrict__ ws, uint32_t n, uint32_t k_hi, uint32_t k_lo, uint32_t cap,
                                       uint32_t* scratch, uint32_t* stop, uint32_t* out) {
    constexpr int per = kQ1 / THREADS;
    const int t = threadIdx.x;
    uint32_t mine[per];
    uint32_t local = 0;
#pragma unroll
    for (int j = 0; j < per; j++) { mine[j] = ws[kOffH0 + t * per + j]; local += mine[j]; }
    uint32_t excl, m;
    block_scan_excl<THREADS>(local, scratch, excl, m);     // m = sample size
    if (t == 0) { stop[0] = 0u; stop[1] = kQ1; out[0] = 0u; out[1] = 0xFFFFFFFFu; out[2] = 0u; }
    __syncthreads();
    if (m == 0) return;                                    // block-uniform
    const double frac = (double)m / (double)n;
    // 2x the expected sample count + 24 (a list a few times longer than needed costs select A nothing; a too short one costs
    // three passes over the tensor -- and the granules of a sample of channel-structured activations are correlated, so the
    // estimate is looser than its size suggests); a too short list is caught by select A (F1..F3 then), never wrong
    const double need_hi = PPQHIP_Q_NEED * frac * (double)(n - 1 - k_hi) + 24.0;
    const double need_lo = PPQHIP_Q_NEED * frac * (double)k_lo + 24.0;
    const double budget = frac * (double)(cap / 2);
    uint32_t F = excl;                                     // F(b) = sample count with top < b, here b = t * per
    uint32_t best_hi = 0, best_lo = kQ1;
    bool any_hi = false;
#pragma unroll
    for (int j = 0; j < per; j++) {
        const uint32_t b = (uint32_t)(t * per + j);
        const uint32_t Fb = F, Fb1 = F + mine[j];
        if ((double)Fb1 >= need_lo && b < best_lo) best_lo = b;
        if ((double)(m - Fb) >= need_hi) { best_hi = b; any_hi = true; }
        F = Fb1;
    }
    if (best_lo < kQ1) atomicMin(&stop[1], best_lo);
    if (any_hi) atomicMax(&stop[0], best_hi);
    __syncthreads();
    const uint32_t bh = stop[0], bl = stop[1];
    const bool have_hi = (double)m >= need_hi, have_lo = bl < kQ1;
    F = excl;
#pragma unroll
    for (int j = 0; j < per; j++) {
        const uint32_t b = (uint32_t)(t * per + j);
        const uint32_t Fb = F, Fb1 = F + mine[j];
        const uint32_t L = b << 20, H = L | 0xFFFFFu, R = round_key_of_bucket(b);
        const double cnt = (double)mine[j];
        if (have_hi && b == bh) {
            const double above = (double)(m - Fb1), missing = need_hi - above;          // missing in (0, cnt]
            const double round = (double)ws[kOffR0 + b];
            uint32_t T = 0xFFFFFFFFu;
            if (2.0 * round >= cnt) {
                if (above + (R == L ? cnt - round : 0.0) <= budget) T = R;
            } else {
                const double take = fmin(cnt, PPQHIP_Q_TAKE * missing);
                if (above + take <= budget) {
                    const uint32_t w = (uint32_t)(take / cnt * 1048576.0);
                    T = w >= 0x100000u ? (L ? L - 1u : 0u) : H - w;
                }
            }
            out[1] = T;
        }
        if (have_lo && b == bl) {
            const double below = (double)Fb, missing = need_lo - below;
            const double round = (double)ws[kOffR0 + b];
            uint32_t T = 0u;
            if (2.0 * round >= cnt) {
                if (below + (R == H ? cnt - round : 0.0) <= budget) T = R;
            } else {
                const double take = fmin(cnt, PPQHIP_Q_TAKE * missing);
                if (below + take <= budget) {
                    const uint32_t w = (uint32_t)(take / cnt * 1048576.0);
                    T = w >= 0x100000u ? (H == 0xFFFFFFFFu ? H : H + 1u) : L + w;
                }
            }
            out[2] = T;
        }
        F = Fb1;
    }
    __syncthreads();
    if (t == 0) out[0] = out[2] <= out[1] ? 1u : 0u;       // thresholds cross (tiny / degenerate sample): no filter
    __syncthreads();
}

// ---- the filter: one streaming pass over every job with thresholds ------------------------------------------------
#ifndef PPQHIP_QF_BLOCK
#define PPQHIP_QF_BLOCK 512
#endif
#ifndef PPQHIP_QF_WGPC
#define PPQHIP_QF_WGPC 2
#endif
#ifndef PPQHIP_QF_NT
#define PPQHIP_QF_NT 1
#endif
#ifndef PPQHIP_QF_STAGE
#define PPQHIP_QF_STAGE 2048
#endif
constexpr int kQFBlock = PPQHIP_QF_BLOCK, kQFU = (int)kQTileVec / kQFBlock, kQFWgPerCu = PPQHIP_QF_WGPC;
constexpr uint32_t kQFLocalCap = PPQHIP_QF_STAGE;             // keys a workgroup can stage per side and job
static_assert(kQFBlock * kQFU == (int)kQTileVec && kQFU >= 1, "a tile is 1024 float4");

// A key outside [T_lo, T_hi]: stage it for the list of its side.  INLINED: behind a call the compiler no longer knows the
// state of vmcnt, and every wait of the streaming loop became vmcnt(0) -- also for the tile it had just prefetched.
__device__ __forceinline__ void qf_rare_key(uint32_t key, uint32_t t_hi, uint32_t* staged_hi, uint32_t* staged_lo,
                                         uint32_t* staged_n) {
    const int w = key > t_hi ? 0 : 1;
    const uint32_t at = atomicAdd(&staged_n[w], 1u);
    if (at < kQFLocalCap) (w ? staged_lo : staged_hi)[at] = key;
}

__global__ __launch_bounds__(kQFBlock, (kQFBlock * kQFWgPerCu + 255) / 256)
void quantile_filter_kernel(const QSeq s) {
    __shared__ uint32_t ft[kQMaxJobs];
    __shared__ uint32_t staged[2][kQFLocalCap];
    __shared__ uint32_t staged_n[2], staged_base[2];
    __shared__ uint32_t ties[2];                  // keys seen == T_hi / == T_lo (this workgroup, this job)
    __shared__ uint32_t scratch[kQFBlock / kWave], stop[2], thr[3];
    const uint32_t G = gridDim.x, g = blockIdx.x;
    uint32_t t, t_end;
    even_split(s.total_tiles, G, g, t, t_end);
    if (t >= t_end) return;
    if (threadIdx.x < 2) { staged_n[threadIdx.x] = 0; ties[threadIdx.x] = 0; }
    if (s.count > 1) load_prefix(ft, s.first_tile, s.count);       // (a single job: no table to look anything up in)
    else { if (threadIdx.x == 0) ft[0] = 0u; __syncthreads(); }
    for (uint32_t j = find_job(ft, s.count, t); t < t_end; j++) {
        const QJob job = s.job[j];
        const uint32_t j_end = (j + 1 < s.count) ? ft[j + 1] : s.total_tiles;
        const uint32_t k0 = t - ft[j];
        uint32_t k = k0;
        const uint32_t k1 = umin(t_end, j_end) - ft[j];
        t = umin(t_end, j_end);
        const float* __restrict__ x = job.x;
        const uint32_t n = job.n;
        uint32_t* P = job.ws + kOffSpec;
        const bool vec_ok = aligned16_dev(x);
        const uint32_t full = vec_ok ? (n >> 2) / kQTileVec : 0u;
        const uint32_t kf = umin(k1, full);
        // the first tile is requested BEFORE the thresholds are known: their loads (hint words, or the sample histogram of a
        // cold job) are a dependent round trip or two that every workgroup of a persistent grid would otherwise sit out idle
        const float4* xv = reinterpret_cast<const float4*>(x) + threadIdx.x;
        float4 bufa[kQFU], bufb[kQFU];
        auto fetch = [&](float4 (&buf)[kQFU], uint32_t tile) {
            const float4* p = xv + (size_t)tile * kQTileVec;
#pragma unroll
            for (int u = 0; u < kQFU; u++) buf[u] = gload4<PPQHIP_QF_NT != 0>(p + u * kQFBlock);
        };
        if (k < kf) fetch(bufa, k);
        // thresholds: the hint of the previous batch, else from this batch's sample (block-uniform either way)
        const bool hot = hint_valid(job.hint, n, job.k_hi, job.k_lo);
        uint32_t t_hi, t_lo;
        bool enabled;
        if (hot) { t_hi = job.hint[kHTHi]; t_lo = job.hint[kHTLo]; enabled = t_lo <= t_hi; }
        else if (!vec_ok) { t_hi = 0xFFFFFFFFu; t_lo = 0u; enabled = false; }
        else {
            thresholds_from_sample<kQFBlock>(job.ws, n, job.k_hi, job.k_lo, job.cap, scratch, stop, thr);
            enabled = thr[0] != 0u; t_hi = thr[1]; t_lo = thr[2];
            __syncthreads();                                  // thr is rewritten for the next job
        }
        if (k0 == 0 && threadIdx.x == 0) {                    // the owner of the job's first tile publishes them for select A
            // (write-through stores: the line also holds the counters other workgroups add to with device atomics)
            __hip_atomic_store(&P[kPTHi], t_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&P[kPTLo], t_lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&P[kPHot], hot ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&P[kPEnabled], enabled ? 1u : 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (!enabled) continue;                               // nothing to filter by: the job is settled by F1..F3
        const uint32_t span = t_hi - t_lo;                    // key - t_lo > span <=> outside [t_lo, t_hi]
        int tie_hi = 0, tie_lo = 0;                           // wave-uniform
        auto rare = [&](uint32_t key) { qf_rare_key(key, t_hi, staged[0], staged[1], staged_n); };
        if (k < kf) {
            auto consume = [&](const float4 (&buf)[kQFU]) {
#pragma unroll
                for (int u = 0; u < kQFU; u++) {
                    const uint32_t q0 = f2key(buf[u].x), q1 = f2key(buf[u].y), q2 = f2key(buf[u].z), q3 = f2key(buf[u].w);
                    const uint32_t d0 = q0 - t_lo, d1 = q1 - t_lo, d2 = q2 - t_lo, d3 = q3 - t_lo;
                    if ((u & 1) == 0) {    // ties on the thresholds: a lower bound is all select A needs -> one element in eight
                        tie_hi += popc_mask(__builtin_amdgcn_ballot_w64(q0 == t_hi));
                        tie_lo += popc_mask(__builtin_amdgcn_ballot_w64(q0 == t_lo));
                    }
                    if (umax(umax(d0, d1), umax(d2, d3)) > span) {
                        if (d0 > span) rare(q0);
                        if (d1 > span) rare(q1);
                        if (d2 > span) rare(q2);
                        if (d3 > span) rare(q3);
                    }
                }
            };
            for (;;) {
                fetch(bufb, umin(k + 1, kf - 1));
                consume(bufa);
                if (++k >= kf) break;
                fetch(bufa, umin(k + 1, kf - 1));
                consume(bufb);
                if (++k >= kf) break;
            }
        }
        for (; k < k1; k++) {             // ragged tail tile: masked 4-B loads (ties are not counted here: a lower bound)
            const uint32_t e0 = k * kQTileElems + threadIdx.x;
#pragma unroll 4
            for (int r = 0; r < 4 * kQFU; r++) {
                const uint32_t i = e0 + r * kQFBlock;
                if (i < n) {
                    const uint32_t key = f2key(gload1(x + i));
                    if (key - t_lo > span) rare(key);
                }
            }
        }
        if ((threadIdx.x & 63) == 0) {
            if (tie_hi) atomicAdd(&ties[0], (uint32_t)tie_hi);
            if (tie_lo) atomicAdd(&ties[1], (uint32_t)tie_lo);
        }
        __syncthreads();
        const uint32_t shards = q_job_shards(job.tiles), shard = g & (shards - 1u), seg = job.cap / shards;
        if (threadIdx.x < 2) {                                 // reserve this workgroup's slice of the job's lists
            const uint32_t all = staged_n[threadIdx.x];
            staged_base[threadIdx.x] = all ? atomicAdd(&P[kPCnt + 8 * threadIdx.x + shard], umin(all, kQFLocalCap)) : 0u;
            if (all > kQFLocalCap) atomicOr(&P[threadIdx.x ? kPOvfLo : kPOvfHi], 1u);
            if (ties[threadIdx.x]) atomicAdd(&P[kPTie + 8 * threadIdx.x + shard], ties[threadIdx.x]);
        }
        __syncthreads();
        for (int w = 0; w < 2; w++) {
            const uint32_t cnt = umin(staged_n[w], kQFLocalCap), at = staged_base[w];
            uint32_t* list = job.spec + (w ? job.cap : 0u) + shard * seg;
            for (uint32_t i = threadIdx.x; i < cnt; i += kQFBlock)
                if (at + i < seg) list[at + i] = staged[w][i];
        }
        __syncthreads();
        if (threadIdx.x < 2) { staged_n[threadIdx.x] = 0; ties[threadIdx.x] = 0; }
        __syncthreads();
    }
}

// ---- selection inside a key list --------------------------------------------------------------------------------------
// The rank-th smallest (0-based) of the keys of <= 8 list segments (count >= 1 in total, rank < count, every segment 16-B
// aligned); 256 threads, all call.  Radix select on (key - min) over the bits the keys' RANGE actually has (the most
// extreme keys of a tensor share their high bits: the top 12 bits of the full key would pile them into two or three LDS
// counters), <= 3 rounds of <= 12 bits.  Up to `lds_keys` keys are read from global memory ONCE and kept in LDS.
struct KeyLists {
    const uint32_t* base;       // segment i starts at base + i * seg
    uint32_t seg, segments, count;
    uint32_t cnt[kQShards];
};
template <int THREADS, typename F>
__device__ __forceinline__ void list_sweep(const uint32_t* __restrict__ list, uint32_t count, uint32_t at0, uint32_t lt, F&& f) {
    // `lt` = this thread's index among the THREADS threads that sweep this list
    const uint4* lv = reinterpret_cast<const uint4*>(list);
    const uint32_t nv = (count + 3) >> 2;
    for (uint32_t i = lt; i < nv; i += 8 * THREADS) {
        uint4 k[8];
#pragma unroll
        for (int u = 0; u < 8; u++) k[u] = lv[umin(i + u * THREADS, nv - 1)];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const uint32_t at = (i + u * THREADS) << 2;
            if (i + u * THREADS < nv) {
                if (at + 0 < count) f(k[u].x, at0 + at + 0);
                if (at + 1 < count) f(k[u].y, at0 + at + 1);
                if (at + 2 < count) f(k[u].z, at0 + at + 2);
                if (at + 3 < count) f(k[u].w, at0 + at + 3);
            }
        }
    }
}
// all segments at once: THREADS / 8 threads per segment (8 dependent sweeps in a row cost select A 6 us)
template <int THREADS, typename F>
__device__ __forceinline__ void lists_sweep(const KeyLists& L, F&& f) {
    if (L.segments == 1) { list_sweep<THREADS>(L.base, L.cnt[0], 0u, threadIdx.x, f); return; }
    constexpr uint32_t tpg = THREADS / kQShards;                        // segments == kQShards
    const uint32_t grp = threadIdx.x / tpg, lt = threadIdx.x % tpg;
    uint32_t at0 = 0, count = 0;
#pragma unroll
    for (int i = 0; i < kQShards; i++) { if ((uint32_t)i < grp) at0 += L.cnt[i]; if ((uint32_t)i == grp) count = L.cnt[i]; }
    list_sweep<tpg>(L.base + (size_t)grp * L.seg, count, at0, lt, f);
}
template <int THREADS>
__device__ uint32_t list_select(const KeyLists& L, uint32_t rank, uint32_t* keys, uint32_t lds_keys, uint32_t* h, uint32_t* scratch,
                                uint32_t* sel) {
    const uint32_t count = L.count;
    const bool in_lds = count <= lds_keys;
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
    lists_sweep<THREADS>(L, [&](uint32_t key, uint32_t at) { mn = umin(mn, key); mx = umax(mx, key); if (in_lds) keys[at] = key; });
    mn = wave_min_u32(mn); mx = wave_max_u32(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { scratch[threadIdx.x >> 6] = mn; scratch[16 + (threadIdx.x >> 6)] = mx; }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < THREADS / kWave; w++) { mn = umin(mn, scratch[w]); mx = umax(mx, scratch[16 + w]); }
    __syncthreads();
    if (mn == mx) return mn;
    int pos = 32 - __builtin_clz(mx - mn);                  // bits of the range, 1..32
    uint32_t prefix = 0;                                    // the digits chosen so far == (key - mn) >> pos
    while (pos > 0) {
        const int w = pos > 12 ? 12 : pos;
        const int shift = pos - w;
        for (int i = threadIdx.x; i < kQ1; i += THREADS) h[i] = 0;
        __syncthreads();
        auto digit = [&](uint32_t key) {
            const uint32_t d = key - mn;
            const uint32_t head = pos >= 32 ? 0u : d >> pos;
            if (head == prefix) atomicAdd(&h[(d >> shift) & ((1u << w) - 1u)], 1u);
        };
        if (in_lds) { for (uint32_t i = threadIdx.x; i < count; i += THREADS) digit(keys[i]); }
        else lists_sweep<THREADS>(L, [&](uint32_t key, uint32_t) { digit(key); });
        __syncthreads();
        select_bin<THREADS>(h, w > 8 ? kQ1 : kQ3, rank, scratch, sel);
        prefix = (prefix << w) | sel[0];
        rank = sel[1];
        pos = shift;
        __syncthreads();
    }
    return mn + prefix;
}

// the longest list a hint may keep producing: a few thousand keys cost select A nothing, whatever multiple of `wanted`
__device__ __forceinline__ uint32_t q_list_limit(uint32_t wanted, uint32_t cap) { return umin(cap / 2u, umax(16u * wanted + 1024u, 8192u)); }

// ---- select A: one workgroup per (job, side) ---------------------------------------------------------------------------
// The list of the side holds EVERY key beyond the threshold T (unless it overflowed): the `count` most extreme keys of the
// tensor.  Sorted ascending S[0..n):
//   hi:  keys > T are S[n - count .. n):  k >= n - count -> the (k - (n - count))-th smallest listed key
//        else away = (n - count) - k >= 1 positions below them lie the keys == T:  away <= tie -> T
//   lo:  keys < T are S[0 .. count):      k < count -> the k-th smallest listed key;  else away = k - count + 1 <= tie -> T
// Anything else (unlucky sample, overflow, a tie on a value the thresholds do not sit on) is left OPEN for F1..F3.
// The hint of the side is kept when the list was comfortably long, dropped when it was short, overflowing or useless.
constexpr int kQSABlock = 1024;                 // 16 waves: the list sweeps and the LDS rounds are latency chains, 4x the lanes = 1/4 the trips
constexpr uint32_t kQSALdsKeys = 32768;
__global__ __launch_bounds__(kQSABlock) void quantile_select_a_kernel(const QSeq s) {
    __shared__ uint32_t keys[kQSALdsKeys];
    __shared__ uint32_t h[kQ1];
    __shared__ uint32_t scratch[32];
    __shared__ uint32_t sel[2];
    // (the counters are addressed from the sequence's base, not through the job record: both loads leave together)
    const uint32_t* P = s.fixed + (size_t)(blockIdx.x >> 1) * kQWords + kOffSpec;
    const QJob job = s.job[blockIdx.x >> 1];
    const int w = (int)(blockIdx.x & 1u);
    const uint32_t n = job.n, k = w ? job.k_lo : job.k_hi;
    bool done = false, keep = false;
    const uint32_t T = P[w ? kPTLo : kPTHi];
    if (P[kPEnabled] != 0u) {
        KeyLists L;
        L.segments = q_job_shards(job.tiles); L.seg = job.cap / L.segments; L.base = job.spec + (w ? job.cap : 0u);
        uint32_t count = 0, tie = 0;
        bool complete = P[w ? kPOvfLo : kPOvfHi] == 0u;
#pragma unroll
        for (int i = 0; i < kQShards; i++) {
            const uint32_t c = (uint32_t)i < L.segments ? P[kPCnt + 8 * w + i] : 0u;
            L.cnt[i] = c; count += c; tie += P[kPTie + 8 * w + i];
            complete = complete && c <= L.seg;
        }
        L.count = count;
        if (complete) {
            const uint32_t wanted = w ? k + 1u : n - k;             // listed keys the answer needs
            if (count >= wanted) {
                const uint32_t key = list_select<kQSABlock>(L, w ? k : k - (n - count), keys, kQSALdsKeys, h, scratch, sel);
                if (threadIdx.x == 0) job.dest[w] = key2f(key);
                done = true;
                // next batch: the same threshold while the list is neither nearly too short nor needlessly long (a list of
                // a couple of thousand keys costs nothing, whatever multiple of `wanted` it is)
                keep = count - wanted >= (wanted >> 3) + 8u && count <= q_list_limit(wanted, job.cap);
            } else if (wanted - count <= tie) {
                if (threadIdx.x == 0) job.dest[w] = key2f(T);
                done = true;
                keep = true;                                         // the tie value itself: as stationary as the activation
            }
        }
    }
    if (threadIdx.x == 0) {
        uint32_t* S = job.ws + kOffSel + 8 * w;
        S[kSMode] = done ? kModeDone : kModeHist;
        if (!done) atomicAdd(&s.header[kGOpen], 1u);
        if (job.hint != nullptr) {
            uint32_t* H = job.hint;
            H[w ? kHValidLo : kHValidHi] = keep ? 1u : 0u;
            H[w ? kHTLo : kHTHi] = T;
            if (w == 0) {
                H[kHN] = n; H[kHKHi] = job.k_hi; H[kHKLo] = job.k_lo;
                if (done && P[kPHot] != 0u) H[kHUses] += 1u;
            }
        }
    }
}

// ---- how the exact passes spread their work ------------------------------------------------------------------------
// Which jobs are open is not known at launch, so F1..F3 cannot lay the work out as one concatenated tile list the way
// the filter does (a contiguous range per workgroup would land the two or three open jobs of a forward on two or three
// percent of the grid: 180 us for 50 MB).  Instead every workgroup looks at every job (the modes of all sides are
// fetched into LDS once) and takes, of each open job, the slices g', g' + G, .. of 8 tiles, g' = g rotated by a
// per-job offset so that different open jobs start on different workgroups.
constexpr uint32_t kQFSliceTiles = 8;              // 32768 elements (128 KB) per slice
__device__ __forceinline__ void load_modes(uint32_t* modes, const QSeq& s) {
    for (uint32_t i = threadIdx.x; i < 2u * s.count; i += blockDim.x)
        modes[i] = s.fixed[(size_t)(i >> 1) * kQWords + kOffSel + 8 * (i & 1u) + kSMode];      // == job[i >> 1].ws[..], one load
    __syncthreads();
}
__device__ __forceinline__ uint32_t first_slice(uint32_t g, uint32_t G, uint32_t j) { return (g + G - (j * 61u) % G) % G; }

// ---- F1: exact histogram of the top 12 key bits of every job with an open side; tail: bucket + rank per open side ---
constexpr int kQTrash = 64;
// LDS of the three exact passes: they never run at the same time, so the fused kernel below overlays them (a union).
struct F1Lds {
    uint32_t modes[2 * kQMaxJobs];
    int h[kQ1];
    uint32_t scratch[32];
    uint32_t sel[2];
    uint32_t flag;
};
constexpr uint32_t kF2LocalCap = 512;
struct F2Lds {
    uint32_t modes[2 * kQMaxJobs];
    uint32_t h[2 * (kQ2 + kQTrash)];
    uint32_t red[4][kBlock / kWave];
    uint32_t staged[2][kF2LocalCap];
    uint32_t staged_n[2], staged_base[2];
    uint32_t tail_keys[kQCap];          // the tail's LDS copy of a candidate list
    uint32_t scratch[32];
    uint32_t sel[2];
    uint32_t flag;
};
struct F3Lds {
    uint32_t modes[2 * kQMaxJobs];
    uint32_t h[2 * (kQ3 + kQTrash)];
    uint32_t scratch[32];
    uint32_t sel[2];
    uint32_t flag;
};

__device__ __forceinline__ void quantile_f1_body(const QSeq& s, F1Lds& L) {
    uint32_t (&modes)[2 * kQMaxJobs] = L.modes;
    int (&h)[kQ1] = L.h;
    uint32_t (&scratch)[32] = L.scratch;
    uint32_t (&sel)[2] = L.sel;
    uint32_t& flag = L.flag;
    const uint32_t G = gridDim.x, g = blockIdx.x;
    load_modes(modes, s);
    for (int i = threadIdx.x; i < kQ1; i += kBlock) h[i] = 0;
    __syncthreads();
    WaveBinCounter<false, true, true> acc;
    for (uint32_t j = 0; j < s.count; j++) {
        if (modes[2 * j] == kModeDone && modes[2 * j + 1] == kModeDone) continue;
        const QJob job = s.job[j];
        const uint32_t nsl = (job.tiles + kQFSliceTiles - 1) / kQFSliceTiles;
        uint32_t sl = first_slice(g, G, j), mine = 0;
        if (sl >= nsl) continue;
        uint32_t* S_hi = job.ws + kOffSel;
        uint32_t* S_lo = job.ws + kOffSel + 8;
        acc.init(h, kQ1);
        for (; sl < nsl; sl += G) {
            const uint32_t k0 = sl * kQFSliceTiles, k1 = umin(k0 + kQFSliceTiles, job.tiles);
            mine += k1 - k0;
            walk_job_tiles(job.x, job.n, k0, k1,
                           [&](float v, bool in) { acc.elect((int)(f2key(v) >> 20), in); },
                           [&](float v, bool in) { acc.template commit<false>((int)(f2key(v) >> 20), in); });
        }
        acc.flush_hot();
        __syncthreads();
        for (int i = threadIdx.x; i < kQ1; i += kBlock) {
            const int v = h[i];
            if (v) { atomicAdd(&job.ws[kOffH1 + i], (uint32_t)v); h[i] = 0; }
        }
        if (job_ticket(job.ws + kOffTick + 0, mine, job.tiles, &flag)) {
            for (int w = 0; w < 2; w++) {
                uint32_t* S = w ? S_lo : S_hi;
                if (S[kSMode] != kModeDone) {                                    // block-uniform
                    select_bin<kBlock>(job.ws + kOffH1, kQ1, w ? job.k_lo : job.k_hi, scratch, sel);
                    if (threadIdx.x == 0) {
                        const uint32_t top = sel[0];
                        S[kSTop] = top; S[kSRank] = sel[1];
                        S[kSMode] = job.ws[kOffH1 + top] <= kQCap ? kModeCompact : kModeHist;
                        S[kSCount] = 0; S[kSMin] = 0xFFFFFFFFu; S[kSMax] = 0u;
                    }
                }
                __syncthreads();
            }
        }
        __syncthreads();
    }
}

// ---- F2: second radix pass of the open sides.  COMPACT: append the bucket's keys to a candidate list (staged in LDS,
// one reservation per workgroup); else histogram of the middle 12 bits + min / max key of the bucket.
// tail: COMPACT -> finish on the candidate list; all keys of the bucket equal (saturated values) -> done; else the
// 24-bit prefix for F3.
__device__ __forceinline__ void quantile_f2_body(const QSeq& s, F2Lds& L) {
    uint32_t (&modes)[2 * kQMaxJobs] = L.modes;
    uint32_t (&h)[2 * (kQ2 + kQTrash)] = L.h;
    uint32_t (&red)[4][kBlock / kWave] = L.red;
    constexpr uint32_t kLocalCap = kF2LocalCap;
    uint32_t (&staged)[2][kLocalCap] = L.staged;
    uint32_t (&staged_n)[2] = L.staged_n;
    uint32_t (&staged_base)[2] = L.staged_base;
    uint32_t (&tail_keys)[kQCap] = L.tail_keys;
    uint32_t (&scratch)[32] = L.scratch;
    uint32_t (&sel)[2] = L.sel;
    uint32_t& flag = L.flag;
    const uint32_t G = gridDim.x, g = blockIdx.x;
    load_modes(modes, s);
    for (int i = threadIdx.x; i < 2 * (kQ2 + kQTrash); i += kBlock) h[i] = 0;
    if (threadIdx.x < 2) staged_n[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t j = 0; j < s.count; j++) {
        const uint32_t m_hi = modes[2 * j], m_lo = modes[2 * j + 1];
        if (m_hi == kModeDone && m_lo == kModeDone) continue;
        const QJob job = s.job[j];
        const uint32_t nsl = (job.tiles + kQFSliceTiles - 1) / kQFSliceTiles;
        uint32_t sl = first_slice(g, G, j), mine = 0;
        if (sl >= nsl) continue;
        uint32_t* S_hi = job.ws + kOffSel;
        uint32_t* S_lo = job.ws + kOffSel + 8;
        // a finished side must match nothing: 0xFFFFFFFF is no 12-bit prefix
        const uint32_t p_hi = m_hi == kModeDone ? 0xFFFFFFFFu : S_hi[kSTop];
        const uint32_t p_lo = m_lo == kModeDone ? 0xFFFFFFFFu : S_lo[kSTop];
        const bool compact_hi = m_hi == kModeCompact, compact_lo = m_lo == kModeCompact;
        uint32_t* cand_hi = job.ws + kOffCand;
        uint32_t* cand_lo = job.ws + kOffCand + kQCap;
        HotCounter hi_c, lo_c;
        hi_c.init(h, kQ2);
        lo_c.init(h + kQ2 + kQTrash, kQ2);
        uint32_t mn_hi = 0xFFFFFFFFu, mx_hi = 0u, mn_lo = 0xFFFFFFFFu, mx_lo = 0u;
        for (; sl < nsl; sl += G) {
          const uint32_t k0 = sl * kQFSliceTiles, k1 = umin(k0 + kQFSliceTiles, job.tiles);
          mine += k1 - k0;
          walk_job_tiles(job.x, job.n, k0, k1,
                       [&](float v, bool in) {
                           const uint32_t key = f2key(v);
                           hi_c.elect((int)((key >> 8) & 0xFFFu), in && !compact_hi && (key >> 20) == p_hi);
                           lo_c.elect((int)((key >> 8) & 0xFFFu), in && !compact_lo && (key >> 20) == p_lo);
                       },
                       [&](float v, bool in) {
                           const uint32_t key = f2key(v);
                           const uint32_t top = key >> 20;
                           const int mid = (int)((key >> 8) & 0xFFFu);
                           if (in && top == p_hi) {
                               if (compact_hi) {
                                   const uint32_t at = atomicAdd(&staged_n[0], 1u);
                                   if (at < kLocalCap) staged[0][at] = key;
                                   else { const uint32_t gi = atomicAdd(&S_hi[kSCount], 1u); if (gi < kQCap) cand_hi[gi] = key; }
                               } else { hi_c.add(mid); mn_hi = umin(mn_hi, key); mx_hi = umax(mx_hi, key); }
                           }
                           if (in && top == p_lo) {
                               if (compact_lo) {
                                   const uint32_t at = atomicAdd(&staged_n[1], 1u);
                                   if (at < kLocalCap) staged[1][at] = key;
                                   else { const uint32_t gi = atomicAdd(&S_lo[kSCount], 1u); if (gi < kQCap) cand_lo[gi] = key; }
                               } else { lo_c.add(mid); mn_lo = umin(mn_lo, key); mx_lo = umax(mx_lo, key); }
                           }
                       });
        }
        hi_c.flush(); lo_c.flush();
        // workgroup min / max of the bucket keys -> one atomic pair per side
        mn_hi = wave_min_u32(mn_hi); mx_hi = wave_max_u32(mx_hi); mn_lo = wave_min_u32(mn_lo); mx_lo = wave_max_u32(mx_lo);
        const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) { red[0][wid] = mn_hi; red[1][wid] = mx_hi; red[2][wid] = mn_lo; red[3][wid] = mx_lo; }
        __syncthreads();
        if (threadIdx.x < 2) {                                         // reserve this workgroup's slice of the global lists
            const uint32_t cnt = umin(staged_n[threadIdx.x], kLocalCap);
            staged_base[threadIdx.x] = cnt ? atomicAdd(&(threadIdx.x ? S_lo : S_hi)[kSCount], cnt) : 0u;
        }
        __syncthreads();
        for (int w = 0; w < 2; w++) {
            const uint32_t cnt = umin(staged_n[w], kLocalCap), at = staged_base[w];
            uint32_t* cand = w ? cand_lo : cand_hi;
            for (uint32_t i = threadIdx.x; i < cnt; i += kBlock)
                if (at + i < kQCap) cand[at + i] = staged[w][i];
        }
        if (threadIdx.x == 0) {
            for (int w = 1; w < kBlock / kWave; w++) {
                mn_hi = umin(mn_hi, red[0][w]); mx_hi = umax(mx_hi, red[1][w]);
                mn_lo = umin(mn_lo, red[2][w]); mx_lo = umax(mx_lo, red[3][w]);
            }
            if (m_hi == kModeHist && mn_hi <= mx_hi) { atomicMin(&S_hi[kSMin], mn_hi); atomicMax(&S_hi[kSMax], mx_hi); }
            if (m_lo == kModeHist && mn_lo <= mx_lo) { atomicMin(&S_lo[kSMin], mn_lo); atomicMax(&S_lo[kSMax], mx_lo); }
        }
        for (int i = threadIdx.x; i < kQ2; i += kBlock) {
            const uint32_t a = h[i], b = h[kQ2 + kQTrash + i];
            if (a) { if (m_hi == kModeHist) atomicAdd(&job.ws[kOffH2 + i], a); h[i] = 0; }
            if (b) { if (m_lo == kModeHist) atomicAdd(&job.ws[kOffH2 + kQ2 + i], b); h[kQ2 + kQTrash + i] = 0; }
        }
        if (threadIdx.x < kQTrash) { h[kQ2 + threadIdx.x] = 0; h[2 * kQ2 + kQTrash + threadIdx.x] = 0; }
        __syncthreads();
        if (threadIdx.x < 2) staged_n[threadIdx.x] = 0;
        if (job_ticket(job.ws + kOffTick + 1, mine, job.tiles, &flag)) {
            // the tail: h doubles as the selection's scratch (its counters are flushed and zero)
            for (int w = 0; w < 2; w++) {
                uint32_t* S = w ? S_lo : S_hi;
                const uint32_t mode = S[kSMode], top = S[kSTop], rank = S[kSRank];
                if (mode != kModeDone && job.hint != nullptr) {
                    // The exact passes know where the answer lies: leave the NEXT batch a threshold that lists ~1.5x the needed
                    // keys, so that a tensor whose sample misleads the estimate (channel-structured activations) pays for the
                    // three passes once, not every batch.  hist1 / hist2 give the exact number of keys a threshold lists.
                    const uint32_t n = job.n, k = w ? job.k_lo : job.k_hi;
                    const uint32_t inb = job.ws[kOffH1 + top];                 // keys in the bucket of the answer
                    const uint32_t outer = w ? k - rank : n - (k - rank) - inb;   // keys beyond the bucket, on the extreme side
                    const uint32_t wanted = w ? k + 1u : n - k;                // = outer + the bucket's keys from the answer outwards
                    const uint32_t target = wanted + (wanted >> 1) + 32u, limit = q_list_limit(wanted, job.cap);
                    const bool one_value = mode == kModeHist && S[kSMin] == S[kSMax];
                    uint32_t T = 0u, listed = 0xFFFFFFFFu;
                    bool ok = false;
                    if (mode == kModeCompact || one_value) {                   // the whole bucket / the one value it holds
                        const uint32_t V = S[kSMin];
                        listed = outer + inb;
                        if (one_value) { ok = w ? V < 0xFFFFFFFFu : V > 0u; T = w ? V + 1u : V - 1u; }
                        else { ok = w ? top < 0xFFFu : top > 0u; T = w ? (top + 1u) << 20 : (top << 20) - 1u; }
                        ok = ok && listed <= limit;
                        // a tie too heavy to list: the threshold ON the value, select A settles it from the tie count (1 in 8 seen)
                        if (!ok && one_value && inb / 16u >= wanted + 16u) { T = V; ok = true; }
                    } else {
                        const uint32_t need_in = target > outer ? target - outer : 1u;     // >= the bucket's share of `wanted`
                        const uint32_t r = w ? umin(inb, need_in) - 1u : (inb > need_in ? inb - need_in : 0u);
                        select_bin<kBlock>(job.ws + kOffH2 + w * kQ2, kQ2, r, scratch, sel);
                        const uint32_t m = sel[0], p24 = (top << 12) | m;
                        if (w) { listed = outer + (r - sel[1]) + job.ws[kOffH2 + kQ2 + m]; ok = p24 < 0xFFFFFFu; T = (p24 + 1u) << 8; }
                        else { listed = outer + inb - (r - sel[1]); ok = p24 > 0u; T = (p24 << 8) - 1u; }
                        ok = ok && listed <= limit;
                        __syncthreads();
                    }
                    if (threadIdx.x == 0) {
                        uint32_t* H = job.hint;
                        H[w ? kHValidLo : kHValidHi] = ok ? 1u : 0u;
                        H[w ? kHTLo : kHTHi] = T;
                        H[kHN] = n; H[kHKHi] = job.k_hi; H[kHKLo] = job.k_lo;
                    }
                }
                if (mode == kModeCompact) {
                    const uint32_t count = umin(S[kSCount], kQCap);
                    // every candidate shares `top`: the rank inside the bucket is the rank inside the list
                    KeyLists L;
                    L.base = job.ws + kOffCand + w * kQCap; L.seg = kQCap; L.segments = 1; L.count = count; L.cnt[0] = count;
                    const uint32_t key = list_select<kBlock>(L, rank, tail_keys, kQCap, h, scratch, sel);
                    if (threadIdx.x == 0) { job.dest[w] = key2f(key); S[kSMode] = kModeDone; }
                } else if (mode == kModeHist) {
                    if (S[kSMin] == S[kSMax]) {                             // every element of the bucket is the same value
                        if (threadIdx.x == 0) { job.dest[w] = key2f(S[kSMin]); S[kSMode] = kModeDone; }
                    } else {
                        select_bin<kBlock>(job.ws + kOffH2 + w * kQ2, kQ2, rank, scratch, sel);
                        if (threadIdx.x == 0) {
                            S[kSP24] = (top << 12) | sel[0]; S[kSR24] = sel[1];
                            atomicAdd(&s.header[kGOpen3], 1u);
                        }
                    }
                }
                __syncthreads();
            }
            for (int i = threadIdx.x; i < 2 * (kQ2 + kQTrash); i += kBlock) h[i] = 0;     // list_select dirtied it
        }
        __syncthreads();
    }
}

// ---- F3: last 8 bits of the sides still open; tail: pick ----------------------------------------------------------------
__device__ __forceinline__ void quantile_f3_body(const QSeq& s, F3Lds& L) {
    uint32_t (&modes)[2 * kQMaxJobs] = L.modes;
    uint32_t (&h)[2 * (kQ3 + kQTrash)] = L.h;
    uint32_t (&scratch)[32] = L.scratch;
    uint32_t (&sel)[2] = L.sel;
    uint32_t& flag = L.flag;
    const uint32_t G = gridDim.x, g = blockIdx.x;
    load_modes(modes, s);
    for (int i = threadIdx.x; i < 2 * (kQ3 + kQTrash); i += kBlock) h[i] = 0;
    __syncthreads();
    for (uint32_t j = 0; j < s.count; j++) {
        const bool need_hi = modes[2 * j] == kModeHist, need_lo = modes[2 * j + 1] == kModeHist;
        if (!need_hi && !need_lo) continue;
        const QJob job = s.job[j];
        const uint32_t nsl = (job.tiles + kQFSliceTiles - 1) / kQFSliceTiles;
        uint32_t sl = first_slice(g, G, j), mine = 0;
        if (sl >= nsl) continue;
        const uint32_t* S_hi = job.ws + kOffSel;
        const uint32_t* S_lo = job.ws + kOffSel + 8;
        const uint32_t p_hi = S_hi[kSP24], p_lo = S_lo[kSP24];
        HotCounter hi_c, lo_c;
        hi_c.init(h, kQ3);
        lo_c.init(h + kQ3 + kQTrash, kQ3);
        for (; sl < nsl; sl += G) {
          const uint32_t k0 = sl * kQFSliceTiles, k1 = umin(k0 + kQFSliceTiles, job.tiles);
          mine += k1 - k0;
          walk_job_tiles(job.x, job.n, k0, k1,
                       [&](float v, bool in) {
                           const uint32_t key = f2key(v);
                           hi_c.elect((int)(key & 0xFFu), in && need_hi && (key >> 8) == p_hi);
                           lo_c.elect((int)(key & 0xFFu), in && need_lo && (key >> 8) == p_lo);
                       },
                       [&](float v, bool in) {
                           const uint32_t key = f2key(v);
                           const int low = (int)(key & 0xFFu);
                           if (in && need_hi && (key >> 8) == p_hi) hi_c.add(low);
                           if (in && need_lo && (key >> 8) == p_lo) lo_c.add(low);
                       });
        }
        hi_c.flush(); lo_c.flush();
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * (kQ3 + kQTrash); i += kBlock) {
            const uint32_t v = h[i];
            const int side = i >= kQ3 + kQTrash ? 1 : 0, bin = i - side * (kQ3 + kQTrash);
            if (v && bin < kQ3) atomicAdd(&job.ws[kOffH3 + side * kQ3 + bin], v);
            h[i] = 0;
        }
        if (job_ticket(job.ws + kOffTick + 2, mine, job.tiles, &flag)) {
            for (int w = 0; w < 2; w++) {
                const uint32_t* S = w ? S_lo : S_hi;
                if (S[kSMode] == kModeHist) {
                    select_bin<kBlock>(job.ws + kOffH3 + w * kQ3, kQ3, S[kSR24], scratch, sel);
                    if (threadIdx.x == 0) {
                        const uint32_t V = (S[kSP24] << 8) | sel[0];
                        job.dest[w] = key2f(V);
                        // hist3 counts single keys: the answer's exact multiplicity.  A heavy tie on a value no threshold rule
                        // looks at (a clip at 5.3, say) would list itself into an overflow batch after batch; with the threshold
                        // ON the value select A settles it from the tie count (one element in eight is counted: 16x margin).
                        const uint32_t mult = job.ws[kOffH3 + w * kQ3 + sel[0]];
                        const uint32_t wanted = w ? job.k_lo + 1u : job.n - job.k_hi;
                        if (job.hint != nullptr && mult / 16u >= wanted + 16u) {
                            uint32_t* H = job.hint;
                            H[w ? kHTLo : kHTHi] = V;
                            H[w ? kHValidLo : kHValidHi] = 1u;
                        }
                    }
                }
                __syncthreads();
            }
        }
        __syncthreads();
    }
}

// ---- the exact passes as kernels ---------------------------------------------------------------------------------------
// (Round 4 fused the three into ONE cooperative launch -- hipLaunchCooperativeKernel + grid.sync() between the passes, early
// exit when nothing is open.  Correct, all quantile tests green, and SLOWER: a cooperative launch costs ~20 us on this stack
// (B hinted 23.6 -> 41.1 us, Bx32 57.6 -> 76.3 us, in situ 0.75 -> 0.68 of the roofline; profiles/r04_quantile_coop.txt), so
// the passes stay three ordinary launches that return on one load when nothing is open.  Skipping the sample launch and shrinking
// these grids for sequences made of already-used hints was measured too (Bx32 57.6 -> 55.6 us, within run-to-run noise: back to
// back the near-empty launches overlap with the tail of the launch before them) and not kept: a hint that select A drops would
// then leave its job to exact passes on a small grid instead of a resampled filter.)
__global__ __launch_bounds__(kBlock) void quantile_f1_kernel(const QSeq s) {
    __shared__ F1Lds lds;
    if (!s.all_open && s.header[kGOpen] == 0u) return;
    quantile_f1_body(s, lds);
}
__global__ __launch_bounds__(kBlock) void quantile_f2_kernel(const QSeq s) {
    __shared__ F2Lds lds;
    if (!s.all_open && s.header[kGOpen] == 0u) return;
    quantile_f2_body(s, lds);
}
__global__ __launch_bounds__(kBlock) void quantile_f3_kernel(const QSeq s) {
    __shared__ F3Lds lds;
    if (s.header[kGOpen3] == 0u) return;
    quantile_f3_body(s, lds);
}

// ONE small tensor with an extreme q (the calibration call: n (1 - q) wanted keys, a few hundred): the filter + select A settle
// it unless the sample misled them or the thresholds of an old hint became absurd, so the exact passes are a RARE fallback --
// and three launches that each return on one load are then 13 us of device time (4.5 + 4.5 + 3.9 on B) plus three boundaries
// behind a 5 us filter.  For such a sequence the three bodies run back to back in ONE launch of ONE workgroup: it returns on one
// load in the common case (~2 us), and when a side is open it walks the tensor three times alone (G = 1: the ticket of every
// pass is its own; an agent-scope fence + barrier between the passes makes the tail's plain stores visible to the next body's
// loads).  That is ~0.5 ms for 6 MB -- acceptable for something that happens on the first batch of an unlucky observer, not for
// a quantile the filter cannot help (a median): the host only routes here when both wanted counts fit the smallest list.
__global__ __launch_bounds__(kBlock) void quantile_f123_single_kernel(const QSeq s) {
    __shared__ union { F1Lds f1; F2Lds f2; F3Lds f3; } lds;
    if (s.header[kGOpen] == 0u) return;
    quantile_f1_body(s, lds.f1);
    __threadfence(); __syncthreads();
    quantile_f2_body(s, lds.f2);
    __threadfence(); __syncthreads();
    if (*(volatile const uint32_t*)&s.header[kGOpen3] != 0u) quantile_f3_body(s, lds.f3);
}
#ifndef PPQHIP_Q_SINGLE_ELEMS
#define PPQHIP_Q_SINGLE_ELEMS (4ll << 20)
#endif

// ---- ONE hinted tensor: two launches ----------------------------------------------------------------------------------
// What the reference's percentile observer does per tensor per batch (observer/range.py:349 -> CUDA.Quantile -> sort.cu:42-59)
// arrives here as ONE tensor with the hint of its observer.  The general sequence above spends five launches on it -- init,
// sample (returns on one load), filter, select A, F1..F3 (return on one load): 26 us of device time on [1,512,56,56], of which
// the filter's read is 5 -- because each launch decides on the DEVICE what the next one has to do (is the hint usable? did the
// lists settle both sides?), and the host cannot know without a synchronisation.  This path keeps every decision on the device
// and still launches only twice:
//   quantile_hot_filter_kernel   the filter with its arguments by value (no job table, no prefix arrays, no init launch).  The
//                                hint is read by every workgroup; a usable one filters the tensor in one pass and every workgroup
//                                leaves its keys in ITS OWN record (count, tie count, six keys inline; more keys in its slot) --
//                                no reservation atomics, nothing shared, no fences: the kernel boundary publishes the records.
//                                It also publishes the thresholds it used and zeroes the state of the launch behind it.
//   quantile_hot_select_kernel   one workgroup per CU.  The workgroup that ARRIVES first (a ticket; big lists: the first two, one side
//                                each -- never "workgroup 0": see the kernel) reads the records -- thread t holds the keys of filter
//                                workgroup t in registers, sweeps a slot of 33-256 keys itself --, counts ONE histogram round on
//                                fixed bit positions of (key - T - 1), picks the 2^11-key-wide bin of the wanted rank, collects the
//                                handful of keys of that bin and lets one wavefront finish on them (select A's rules decide
//                                settled / keep-the-hint); the other workgroups poll ONE word.
//                                Settled (the common case): that workgroup writes dest and the hint, everybody returns.  Not
//                                settled (no usable hint yet, a list that overflowed or came up short): the SAME launch runs
//                                the exact radix select over the whole tensor -- 12 + 12 + 8 key bits, three levels whose chunks
//                                are handed out through a counter, so nothing waits for a workgroup that is not resident --
//                                and leaves a hint computed from the exact histograms (F2's rule: a threshold that lists what
//                                qh_target asks for; F3's rule: ON a heavily tied answer), so the next batch is settled by the filter.
// The FIRST call on a hint does not come here: it takes the general sequence, which samples its thresholds (quantile_hint_met_before).
// What shaped the select (s_memrealtime stamps, tools/quantile_hot_stamps.py): a single workgroup's chain of barrier-separated
// LDS stages costs 0.3-0.6 us per stage whatever it computes; __shfl-based scans are ds_bpermute round trips (DPP instead); 512
// LDS atomics on one address serialise (one per wavefront instead); values that are wave uniform but live in vector registers
// turn every test into an EXEC-mask branch (readfirstlane); one CU pulls ~64 B/clk, so what it reads must be compact.
// Results are exact in every case, as everywhere in this file: the hint only decides how much is read.
#ifndef PPQHIP_QH_POLL_SLEEP
#define PPQHIP_QH_POLL_SLEEP 2                      // s_sleep argument of the workgroups that wait for the decision
#endif
#ifndef PPQHIP_QH_POLL_FIRST
#define PPQHIP_QH_POLL_FIRST 32                     // .. before their first look
#endif
#ifndef PPQHIP_QH_SELECT_WGS
#define PPQHIP_QH_SELECT_WGS 0                      // grid of the select launch; 0: one workgroup per CU
#endif
#ifdef PPQHIP_QH_TIMING                             // developer builds: s_memrealtime stamps (10 ns) of the selecting workgroup -> ws[32 + i]
#define QH_STAMP(i) do { if (threadIdx.x == 0) { a.ws[32 + (i)] = (uint32_t)wall_clock64(); } } while (0)
#else
#define QH_STAMP(i) do { } while (0)
#endif
constexpr int kQHBlock = 512;                       // both kernels
constexpr uint32_t kQHStage = 2048;                 // keys a workgroup can stage per side (== its slot)
constexpr uint32_t kQHInline = 6;                   // keys per side inside the record
constexpr uint32_t kQHMaxWg = 512;                  // filter grid limit (records, slots)
constexpr uint32_t kQHListMax = 65536;              // longest list a hint may keep producing
constexpr uint32_t kQHWantedMax = 8192;             // the host routes here only when both wanted counts are at most this
constexpr uint32_t kQHThreadKeys = 32;              // keys of one filter workgroup and side a thread of the select holds in registers
constexpr uint32_t kQHThreadMore = 256;             // .. and up to this many it sweeps straight from the workgroup's slot (longer slots: through LDS)
constexpr uint32_t kQHRoomPerWg = 128;              // keys per filter workgroup and side the thresholds may count on (half of that: slots are uneven)
// How long a list the NEXT call's threshold is aimed at.  The wanted keys sit 3.7 sigma out (q = 0.9999): the number of keys beyond a
// FIXED threshold moves with the 14th power of the activation's scale, so a list of 1.5 x the wanted keys -- the shortest, fastest
// choice: six keys per filter workgroup, inline in the records -- is used up by a batch whose scale is 3 % smaller, and the call then
// pays the exact passes (tools/quantile_drift.py: 5 % jitter between batches -> one call in three, 31 us per call instead of 10.7).
// Each side of the hint therefore carries a LEVEL 0..3 (bits 8-9 of its valid word): level 0 aims at 1.5 x wanted, level 3 at the
// geometric middle of [wanted, what the select holds in registers] (as much room below as above), 1 and 2 in between.  A call the
// hint could not settle raises the side to level 3, a list that came within a quarter of failing raises it by one, and every 64th
// settled call lowers it by one: a stationary stream works with the short lists, a restless one with the long ones.
__device__ __forceinline__ uint32_t qh_target(uint32_t wanted, uint32_t wgs, uint32_t level) {
    const uint32_t least = wanted + (wanted >> 1);
    const uint32_t room = umin(kQHRoomPerWg * wgs, kQHListMax);
    const uint32_t middle = (uint32_t)sqrtf((float)room * (float)wanted);
    const uint32_t most = umax(least, umin(middle, room / 2u));
    return least + (most - least) * umin(level, 3u) / 3u + 32u;
}
// workspace layout (uint32 words)
enum { kQHEnabled = 0, kQHTHi = 1, kQHTLo = 2, kQHUses = 3,      // written by the filter's workgroup 0 (uses: hint word 7 as it found it)
       kQHZero0 = 4,                                // first word the filter zeroes
       kQHLoFlag = 4,                               // the lo side's decision, published by its owner: 0 pending, 1 settled, 2 open
       kQHRoleTicket = 10,                          // arrival ticket of the select launch: the first arrival selects (the second: the lo side of a split select)
       kQHLoClaim = 9,                              // who owns the lo side of a split select: 0 nobody yet, 1 the second arrival, 2 the first
       kQHDecision = 7,                             // the hi side's decision, published by its owner: 0 pending, 1 settled, 2 open
       kQHNext = 12,                                // [3] next chunk of each exact level
       kQHDone = 16 };                              // [3] chunks counted per exact level
constexpr uint32_t kQHOffH0 = 64;                                   // hist of key >> 20 (both sides select from it)
constexpr uint32_t kQHOffH1 = kQHOffH0 + kQ1;                       // [2][4096]: (key >> 8) & 0xFFF of the side's bucket
constexpr uint32_t kQHOffH2 = kQHOffH1 + 2 * kQ2;                   // [2][256]: key & 0xFF of the side's 24-bit prefix
constexpr uint32_t kQHZeroEnd = kQHOffH2 + 2 * kQ3;
constexpr uint32_t kQHOffRec = kQHZeroEnd;                          // [kQHMaxWg][2][8]: per side count, tie count, the first six keys (one 64-B line per workgroup)
constexpr uint32_t kQHOffHeads = kQHOffRec + kQHMaxWg * 16;         // [kQHMaxWg][2][32]: the first 32 keys of every slot, contiguous; read when a side holds more than six
constexpr uint32_t kQHOffSlots = kQHOffHeads + kQHMaxWg * 2 * 32;   // [kQHMaxWg][2][kQHStage]: the whole slot, read when it holds more than 32 keys
constexpr size_t kQHWords = (size_t)kQHOffSlots + (size_t)kQHMaxWg * 2 * kQHStage;
static_assert(kQHOffRec % 4 == 0 && kQHOffHeads % 4 == 0 && kQHOffSlots % 4 == 0, "16-B alignment of records and slots");

struct QHot {
    const float* x;
    float* dest;
    uint32_t* hint;
    uint32_t* ws;
    uint32_t n, k_hi, k_lo, wgs;     // wgs: grid of the filter (records to gather)
    uint32_t split, heads, pad0, pad1;   // split: two selecting workgroups, one per side; heads: the slots' first 32 keys are requested with the records
};

template <int K, bool PING, bool NT>
__global__ __launch_bounds__(kQHBlock) void quantile_hot_filter_kernel(const QHot a) {
    __shared__ uint32_t staged[2][kQHStage];
    __shared__ uint32_t staged_n[2], ties[2];
    const uint32_t G = gridDim.x, g = blockIdx.x, n = a.n;
    const uint32_t full_rows = (n >> 2) / kQHBlock;                   // rows of kQHBlock float4
    uint32_t r, r1;
    even_split(full_rows, G, g, r, r1);
    const float4* xv = reinterpret_cast<const float4*>(a.x) + threadIdx.x;
    float4 bufa[K], bufb[K];
    auto fetch = [&](float4 (&buf)[K], uint32_t row) {                 // clamped rows: straight-line loads (see hist_small_kernel)
#pragma unroll
        for (int k = 0; k < K; k++) buf[k] = gload4<NT>(xv + (size_t)umin(row + (uint32_t)k, r1 - 1u) * kQHBlock);
    };
    if (r < r1) fetch(bufa, r);                                        // in flight while the hint is read
    // the hint: ONE scalar load of its eight words (a short-circuit && chain compiles to five dependent round trips)
    const uint32_t* __restrict__ H = a.hint;
    const uint32_t h0 = H[kHValidHi], t_hi = H[kHTHi], h2 = H[kHValidLo], t_lo = H[kHTLo], h4 = H[kHN], h5 = H[kHKHi], h6 = H[kHKLo], h7 = H[kHUses];
    const bool enabled = (((h0 & 0xFFu) == 1u) & ((h2 & 0xFFu) == 1u) & (h4 == n) & (h5 == a.k_hi) & (h6 == a.k_lo) & (t_lo <= t_hi)) != 0;   // the same in every workgroup
    if (threadIdx.x < 2) { staged_n[threadIdx.x] = 0; ties[threadIdx.x] = 0; }
    {   // the state of the launch behind this one: flags, barrier counter, exact histograms (zeroed whether needed or not)
        constexpr uint32_t words = kQHZeroEnd - kQHZero0;
        for (uint32_t i = g * kQHBlock + threadIdx.x; i < words; i += G * kQHBlock) a.ws[kQHZero0 + i] = 0u;
        // (word 0: enabled | the sides' list-length levels, qh_target)
        if (g == 0 && threadIdx.x == 0) *reinterpret_cast<uint4*>(a.ws) = make_uint4(enabled ? (1u | (h0 & 0x300u) | ((h2 & 0x300u) << 8)) : 0u, t_hi, t_lo, h7);
    }
    if (!enabled) return;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // LDS counters are zero; the loads stay in flight
    const uint32_t span = t_hi - t_lo;                                 // key - t_lo > span <=> outside [t_lo, t_hi]
    int tie_hi = 0, tie_lo = 0;                                        // wave uniform
    auto rare = [&](uint32_t key) {
        const int w = key > t_hi ? 0 : 1;
        const uint32_t at = atomicAdd(&staged_n[w], 1u);
        if (at < kQHStage) staged[w][at] = key;
    };
    auto consume = [&](const float4 (&buf)[K], uint32_t cnt) {
#pragma unroll
        for (int k = 0; k < K; k++) {
            if ((uint32_t)k < cnt) {                                   // block uniform
                const uint32_t q0 = f2key(buf[k].x), q1 = f2key(buf[k].y), q2 = f2key(buf[k].z), q3 = f2key(buf[k].w);
                const uint32_t d0 = q0 - t_lo, d1 = q1 - t_lo, d2 = q2 - t_lo, d3 = q3 - t_lo;
                if ((k & 1) == 0) {        // ties on the thresholds: a lower bound is all the select needs -> one element in eight
                    tie_hi += popc_mask(__builtin_amdgcn_ballot_w64(q0 == t_hi));
                    tie_lo += popc_mask(__builtin_amdgcn_ballot_w64(q0 == t_lo));
                }
                if (umax(umax(d0, d1), umax(d2, d3)) > span) {
                    if (d0 > span) rare(q0);
                    if (d1 > span) rare(q1);
                    if (d2 > span) rare(q2);
                    if (d3 > span) rare(q3);
                }
            }
        }
    };
    if (r < r1) {
        if (PING) {
            for (;;) {
                fetch(bufb, r + K);
                consume(bufa, umin((uint32_t)K, r1 - r));
                r += K;
                if (r >= r1) break;
                fetch(bufa, r + K);
                consume(bufb, umin((uint32_t)K, r1 - r));
                r += K;
                if (r >= r1) break;
            }
        } else {
            for (;;) {
                consume(bufa, umin((uint32_t)K, r1 - r));
                r += K;
                if (r >= r1) break;
                fetch(bufa, r);
            }
        }
    }
    if (g == G - 1) {                                                  // the ragged rest: < kQHBlock float4 + n % 4 elements
        for (uint32_t i = full_rows * kQHBlock * 4u + threadIdx.x; i < n; i += kQHBlock) {
            const uint32_t key = f2key(gload1(a.x + i));
            if (key - t_lo > span) rare(key);
        }
    }
    if ((threadIdx.x & 63) == 0) {
        if (tie_hi) atomicAdd(&ties[0], (uint32_t)tie_hi);
        if (tie_lo) atomicAdd(&ties[1], (uint32_t)tie_lo);
    }
    __syncthreads();
    if (threadIdx.x < 16) {                                            // the record of both sides: one 64-B store
        const uint32_t side = threadIdx.x >> 3, j = threadIdx.x & 7u, c = staged_n[side];
        a.ws[kQHOffRec + g * 16u + threadIdx.x] = j == 0 ? c : (j == 1 ? ties[side] : ((j - 2u) < umin(c, kQHInline) ? staged[side][j - 2u] : 0u));
    }
    if (threadIdx.x < 64) {                                            // the heads of both slots: one 256-B store (read when a side holds more than the record does)
        const uint32_t side = threadIdx.x >> 5, i = threadIdx.x & 31u, c = staged_n[side];
        if ((c > kQHInline || a.heads) && i < c) a.ws[kQHOffHeads + g * 64u + threadIdx.x] = staged[side][i];
    }
#pragma unroll
    for (int side = 0; side < 2; side++) {
        const uint32_t c = umin(staged_n[side], kQHStage);
        if (c > 32u) {
            uint32_t* slot = a.ws + kQHOffSlots + ((size_t)g * 2 + side) * kQHStage;
            for (uint32_t i = threadIdx.x; i < c; i += kQHBlock) slot[i] = staged[side][i];
        }
    }
}

constexpr uint32_t kQHBins = 2048;                  // the select's one histogram round: 11-bit digits ..
constexpr int kQHDigitShift = 11;                   // .. of (key - T - 1) >> 11, saturating: bins of 2^-12 relative width over the half binade above T
                                                    // (the answer is the wanted-th largest of ~1.5 x wanted keys: it lies in the dense third next to T)
constexpr uint32_t kQHSurvCap = 2048;               // keys of the chosen bin ("survivors") a wavefront finishes on
constexpr uint32_t kQHBigCap = 8192;                // LDS room per side for the keys of slots longer than that
constexpr uint32_t kQHWaveKeys = kQHSurvCap;
struct QHSelLds {
    uint32_t hist[2][kQHBins + 64];                 // + one trash counter per lane: the adds of a pass are unconditional
    uint32_t surv[2][kQHSurvCap];
    uint32_t big[2][kQHBigCap];
    uint32_t bigdesc[2][kQHMaxWg][2];               // slots copied into `big`: (workgroup << 16 | count), offset
    uint32_t wavehist[2][320];                      // wave_select: 256 counters + 64 trash counters per wave
    uint32_t total[2], tie[2], nsurv[2], nbig[2], nbigdesc[2], flags;
    uint32_t bin[2], rin[2], result[2];
    uint32_t bin_up[2], bin_q[2];                    // re-centring the thresholds: the bin of rank total - target, of rank total / 4
    uint32_t sc[2][8];
};
struct QHExactLds {
    uint32_t h[2 * (kQ1 + kQTrash)];
};

__device__ __forceinline__ uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// The rank-th smallest (0-based) of keys[0..count) in LDS, 1 <= count <= kQHWaveKeys, by ONE wavefront and without a barrier.
// Up to 64 keys: one key per lane, its rank counted against the other lanes' keys (v_readlane).  More: <= 32 keys per lane in
// registers, radix select on (key - min) with 8-bit digits over a 256-counter LDS histogram private to the wave (`hist`: 320
// words, 16-B aligned).  count / rank must be wave uniform (they are made scalar here: values picked by wave index arrive in
// vector registers, and every test on them would become an EXEC-mask branch with its own LDS wait).
__device__ __forceinline__ uint32_t wave_select(const uint32_t* keys, uint32_t count, uint32_t rank, uint32_t* hist) {
    count = rfl(count); rank = rfl(rank);
    const uint32_t lane = threadIdx.x & 63u;
    if (count <= 64u) {
        const uint32_t mine = keys[umin(lane, count - 1u)];
        uint32_t below = 0u;
        for (uint32_t i = 0; i < count; i++) {                          // scalar trip count
            const uint32_t o = (uint32_t)__builtin_amdgcn_readlane((int)mine, (int)i);
            below += (o < mine || (o == mine && i < lane)) ? 1u : 0u;
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(lane < count && below == rank);      // exactly one lane
        return (uint32_t)__builtin_amdgcn_readlane((int)mine, m ? __builtin_ctzll(m) : 0);
    }
    constexpr int S = (int)(kQHWaveKeys / 64u);
    const int slots = (int)((count + 63u) >> 6);        // wave uniform: registers in use
    uint32_t d[S];
    uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
    for (int j = 0; j < S; j++) d[j] = keys[umin((uint32_t)j * 64u + lane, count - 1u)];       // unconditional: one pipelined burst
#pragma unroll
    for (int j = 0; j < S; j++) { mn = umin(mn, d[j]); mx = umax(mx, d[j]); }                   // (clamped slots repeat the last key)
    mn = wave_all_min(mn); mx = wave_all_max(mx);
    if (mn == mx) return mn;
    int pos = 32 - __builtin_clz(mx - mn);
    uint32_t prefix = 0u;
    uint4* hist4 = reinterpret_cast<uint4*>(hist);
    const uint32_t trash = 256u + lane;                 // per-lane counter for "not this round": the adds stay unconditional
    while (pos > 0) {                                   // wave uniform
        const int w = pos > 8 ? 8 : pos, shift = pos - w;
        hist4[lane] = make_uint4(0u, 0u, 0u, 0u);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < S; j++) {
            if (j < slots) {
                const uint32_t dj = d[j] - mn;
                const bool in = (uint32_t)j * 64u + lane < count;
                const uint32_t head = pos >= 32 ? 0u : dj >> pos;
                const bool match = in && head == prefix;
                atomicAdd(&hist[match ? ((dj >> shift) & ((1u << w) - 1u)) : trash], 1u);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const uint4 c = hist4[lane];
        const uint32_t sum = c.x + c.y + c.z + c.w;
        const uint32_t inc = wave_scan_add(sum);
        const uint32_t excl = inc - sum;
        const bool hit = rank >= excl && rank < inc;    // exactly one lane (rank < the number of matching keys)
        uint32_t digit = lane * 4u, rin = rank - excl;
        if (rin >= c.x) { rin -= c.x; digit++; if (rin >= c.y) { rin -= c.y; digit++; if (rin >= c.z) { rin -= c.z; digit++; } } }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
        const int src = m ? __builtin_ctzll(m) : 0;
        digit = (uint32_t)__builtin_amdgcn_readlane((int)digit, src);
        rank = (uint32_t)__builtin_amdgcn_readlane((int)rin, src);
        prefix = (prefix << w) | digit;
        pos = shift;
    }
    return mn + prefix;
}

// Both sides of the filter's records -> settled?  (select A's rules.)  All 512 threads of ONE workgroup call; the results are
// block uniform.  Thread t holds the keys of workgroup t's slots in registers (both sides: the record and the first 32 keys of each
// slot are requested together, ONE round trip; longer slots go through LDS).  The lo side runs on ~key, so that both sides read
// "the rank-th smallest of the keys ABOVE a threshold".  One histogram round on fixed bit positions of (key - T - 1) finds the
// 2^13-key-wide bin of the answer; the keys of that bin are few and one wavefront per side finishes on them.
// What thread t of a selecting workgroup holds of filter workgroup t: its 64-B record.  Requested BEFORE the workgroup knows whether it
// selects at all (see the kernel): one round trip for the arrival ticket, the header and these.  (The heads of big tensors --
// 164 KB -- are fetched by the selecting workgroups only: requested up front they held the ticket back by 3 us.)
struct QHRecs {
    uint4 r4[4];
    uint4 head[kQHThreadKeys / 4];      // split select of a big tensor: the heads of side `head_side` (2: none held)
    uint32_t head_side;
};
// `guess`: the side this workgroup will probably select (split select: the first arrival takes hi, the second lo -- in practice
// workgroups 0 and 1); 2: no guess.  A wrong guess costs the reload inside hot_select_records, nothing else.
__device__ __forceinline__ void hot_load_records(const QHot& a, QHRecs& R, uint32_t guess) {
    R.r4[0] = R.r4[1] = R.r4[2] = R.r4[3] = make_uint4(0u, 0u, 0u, 0u);
    R.head_side = 2u;
    if (threadIdx.x < a.wgs) {
        const uint4* rec = reinterpret_cast<const uint4*>(a.ws + kQHOffRec) + (size_t)threadIdx.x * 4;
        R.r4[0] = rec[0]; R.r4[1] = rec[1]; R.r4[2] = rec[2]; R.r4[3] = rec[3];
    }
    if (a.heads && guess < 2u) {
        R.head_side = guess;
        if (threadIdx.x < a.wgs) {
            const uint4* head = reinterpret_cast<const uint4*>(a.ws + kQHOffHeads) + ((size_t)threadIdx.x * 2 + guess) * (kQHThreadKeys / 4);
#pragma unroll
            for (uint32_t i = 0; i < kQHThreadKeys / 4; i++) R.head[i] = head[i];
        }
    }
}

// keep[w]: the side's valid word for the hint -- 0: drop it, else 1 | level << 8 (qh_target); level[w] / uses: as the filter found them
__device__ __forceinline__ void hot_select_records(const QHot& a, QHRecs& R, const uint32_t (&T)[2], const uint32_t sides, QHSelLds& L, uint32_t (&key_out)[2],
                                                   bool (&done)[2], uint32_t (&keep)[2], uint32_t (&T_next)[2], const uint32_t (&level)[2], const uint32_t uses) {
    const uint32_t n = a.n, t = threadIdx.x, lane = t & 63u;
    uint4 (&r4)[4] = R.r4;
    uint4 k4[2][kQHThreadKeys / 4];
    if (a.heads && t < a.wgs) {                                        // big tensors: the filter wrote every head; this workgroup's sides only
#pragma unroll
        for (int w = 0; w < 2; w++) {
            if (!(sides & (1u << w))) continue;
            if (R.head_side == (uint32_t)w) {                          // (block uniform) requested with the ticket
#pragma unroll
                for (uint32_t i = 0; i < kQHThreadKeys / 4; i++) k4[w][i] = R.head[i];
                continue;
            }
            const uint4* head = reinterpret_cast<const uint4*>(a.ws + kQHOffHeads) + ((size_t)t * 2 + w) * (kQHThreadKeys / 4);
#pragma unroll
            for (uint32_t i = 0; i < kQHThreadKeys / 4; i++) k4[w][i] = head[i];
        }
    }
    if (t < 2) { L.total[t] = 0u; L.tie[t] = 0u; L.nsurv[t] = 0u; L.nbig[t] = 0u; L.nbigdesc[t] = 0u; L.flags = 0u; }
    {
        uint4* z = reinterpret_cast<uint4*>(&L.hist[0][0]);           // (2 x 2112 words = 1056 uint4)
        z[t] = make_uint4(0u, 0u, 0u, 0u); z[kQHBlock + t] = make_uint4(0u, 0u, 0u, 0u);
        if (t < 2u * (kQHBins + 64u) / 4u - 2u * kQHBlock) z[2 * kQHBlock + t] = make_uint4(0u, 0u, 0u, 0u);
    }
    uint32_t cnt[2] = {r4[0].x, r4[2].x}, tie[2] = {r4[0].y, r4[2].y};
    uint32_t c[2] = {umin(cnt[0], kQHStage), umin(cnt[1], kQHStage)};
#pragma unroll
    for (int w = 0; w < 2; w++) {
        if (!(sides & (1u << w))) { c[w] = 0u; tie[w] = 0u; cnt[w] = 0u; continue; }                   // (block uniform: the other workgroup's side)
        // six keys came with the record; a longer slot's first 32 were requested with it (big tensors: `heads`) or are fetched
        // now (a second round trip, only for the threads that need it)
        if (!a.heads) {
            k4[w][0] = make_uint4(r4[2 * w].z, r4[2 * w].w, r4[2 * w + 1].x, r4[2 * w + 1].y);
            k4[w][1] = make_uint4(r4[2 * w + 1].z, r4[2 * w + 1].w, 0u, 0u);
            if (c[w] > kQHInline && c[w] <= kQHThreadKeys) {
                const uint4* head = reinterpret_cast<const uint4*>(a.ws + kQHOffHeads) + ((size_t)t * 2 + w) * (kQHThreadKeys / 4);
#pragma unroll
                for (uint32_t i = 0; i < kQHThreadKeys / 4; i++) k4[w][i] = head[umin(i, (c[w] - 1u) >> 2)];
            }
        }
    }
    __syncthreads();
    QH_STAMP(8);
#pragma unroll
    for (int w = 0; w < 2; w++) {
        // (one LDS atomic per wavefront: 512 adds on one address serialise at ~5 cycles each -- 4 us measured)
        const uint32_t wc = wave_scan_add(c[w]), wt = wave_scan_add(tie[w]);
        if (lane == 63u) { if (wc) atomicAdd(&L.total[w], wc); if (wt) atomicAdd(&L.tie[w], wt); }
        if (cnt[w] > kQHStage) atomicOr(&L.flags, 1u << w);                              // the workgroup could not stage all its keys
        if (c[w] > kQHThreadMore) {
            const uint32_t base = atomicAdd(&L.nbig[w], c[w]);
            if (base + c[w] <= kQHBigCap) {
                const uint32_t at = atomicAdd(&L.nbigdesc[w], 1u);
                L.bigdesc[w][at][0] = (t << 16) | c[w]; L.bigdesc[w][at][1] = base;
            } else atomicOr(&L.flags, 1u << w);
        }
    }
    const uint32_t Tp[2] = {T[0], ~T[1]};
    // every key this thread holds of side w: f(key', valid), key' = the key (hi) / ~key (lo); straight-line code up to the
    // wave's longest slot (a scalar trip count)
    auto for_my_keys = [&](int w, auto f) {
        const uint32_t cw = c[w] <= kQHThreadKeys ? c[w] : 0u;
        const uint32_t cmax = wave_all_max(cw);
        const uint32_t flip = w ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (uint32_t i = 0; i < kQHThreadKeys / 4; i++) {
            if (4u * i < cmax) {
                f(k4[w][i].x ^ flip, 4u * i + 0u < cw); f(k4[w][i].y ^ flip, 4u * i + 1u < cw);
                f(k4[w][i].z ^ flip, 4u * i + 2u < cw); f(k4[w][i].w ^ flip, 4u * i + 3u < cw);
            }
        }
    };
    // a slot of 33 .. kQHThreadMore keys: its thread sweeps it straight from the workspace, sixteen keys per trip (every lane its own
    // lines, L2-resident: a restless stream's lists -- qh_target -- are a few dozen keys per filter workgroup, not six)
    auto for_slot_keys = [&](int w, auto f) __attribute__((always_inline)) {
        const uint32_t cw = (c[w] > kQHThreadKeys && c[w] <= kQHThreadMore) ? c[w] : 0u;
        const uint32_t cmax = wave_all_max(cw);
        if (cmax == 0u) return;
        const uint32_t flip = w ? 0xFFFFFFFFu : 0u;
        const uint4* slot = reinterpret_cast<const uint4*>(a.ws + kQHOffSlots + ((size_t)t * 2 + w) * kQHStage);
#pragma nounroll
        for (uint32_t i = 0; 4u * i < cmax; i += 4u) {                    // (scalar trip count; the slot is kQHStage keys long: no clamp needed)
            const uint4 q0 = slot[i], q1 = slot[i + 1u], q2 = slot[i + 2u], q3 = slot[i + 3u];
            const uint32_t at = 4u * i;
            f(q0.x ^ flip, at + 0u < cw); f(q0.y ^ flip, at + 1u < cw); f(q0.z ^ flip, at + 2u < cw); f(q0.w ^ flip, at + 3u < cw);
            f(q1.x ^ flip, at + 4u < cw); f(q1.y ^ flip, at + 5u < cw); f(q1.z ^ flip, at + 6u < cw); f(q1.w ^ flip, at + 7u < cw);
            f(q2.x ^ flip, at + 8u < cw); f(q2.y ^ flip, at + 9u < cw); f(q2.z ^ flip, at + 10u < cw); f(q2.w ^ flip, at + 11u < cw);
            f(q3.x ^ flip, at + 12u < cw); f(q3.y ^ flip, at + 13u < cw); f(q3.z ^ flip, at + 14u < cw); f(q3.w ^ flip, at + 15u < cw);
        }
    };
    auto digit_of = [&](int w, uint32_t kp) { return umin((kp - Tp[w] - 1u) >> kQHDigitShift, kQHBins - 1u); };
    const uint32_t trash = kQHBins + lane;
    // the histogram round does not wait for the totals (whether a side selects at all is decided behind the next barrier)
#pragma unroll
    for (int w = 0; w < 2; w++)
        if (sides & (1u << w)) {
            for_my_keys(w, [&](uint32_t kp, bool valid) { atomicAdd(&L.hist[w][valid ? digit_of(w, kp) : trash], 1u); });
            for_slot_keys(w, [&](uint32_t kp, bool valid) { atomicAdd(&L.hist[w][valid ? digit_of(w, kp) : trash], 1u); });
        }
    __syncthreads();
    QH_STAMP(9);
    uint32_t nbigkeys[2] = {0u, 0u};
    if (L.nbigdesc[0] | L.nbigdesc[1]) {                               // block uniform: long slots, wavefront v copies the v-th, (v + 8)-th ..
#pragma unroll
        for (int w = 0; w < 2; w++) {
            const uint32_t nb = L.nbigdesc[w];
            for (uint32_t b = t >> 6; b < nb; b += kQHBlock / kWave) {
                const uint32_t g = L.bigdesc[w][b][0] >> 16, cg = L.bigdesc[w][b][0] & 0xFFFFu;
                const uint32_t* slot = a.ws + kQHOffSlots + ((size_t)g * 2 + w) * kQHStage;
                uint32_t* dst = L.big[w] + L.bigdesc[w][b][1];
                for (uint32_t i = lane; i < cg; i += 4u * kWave) {
                    uint32_t k[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; u++) k[u] = slot[umin(i + u * kWave, cg - 1u)];
#pragma unroll
                    for (uint32_t u = 0; u < 4; u++) if (i + u * kWave < cg) dst[i + u * kWave] = k[u];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int w = 0; w < 2; w++) {
            nbigkeys[w] = umin(L.nbig[w], kQHBigCap);
            for (uint32_t i = t; i < nbigkeys[w]; i += kQHBlock) atomicAdd(&L.hist[w][digit_of(w, L.big[w][i] ^ (w ? 0xFFFFFFFFu : 0u))], 1u);
        }
        __syncthreads();
    }
    const uint32_t flags = L.flags;
    uint32_t total[2], rank[2], wanted[2];
    int how[2];                                                        // 0: open, 1: the threshold itself, 2: select
#pragma unroll
    for (int w = 0; w < 2; w++) {
        const uint32_t k = w ? a.k_lo : a.k_hi;
        total[w] = L.total[w];
        wanted[w] = w ? k + 1u : n - k;
        // hi: the (k - (n - total))-th smallest listed key; lo: the k-th smallest = the (total - 1 - k)-th smallest of the ~keys
        rank[w] = w ? total[w] - 1u - k : k - (n - total[w]);
        done[w] = false; keep[w] = 0u; key_out[w] = T[w]; how[w] = 0;
        if ((flags & (1u << w)) || !(sides & (1u << w))) continue;
        if (total[w] >= wanted[w]) {
            // settled by the list.  The hint is KEPT and its threshold re-centred on this batch (below): select A's rule -- drop a hint
            // whose list came out nearly too short or needlessly long -- costs three exact passes on the next batch here.  The level
            // goes up when the list came within a quarter of failing, down on every 64th settled call.
            how[w] = 2; done[w] = true;
            uint32_t lv = level[w];
            if (total[w] - wanted[w] < (wanted[w] >> 2)) lv = umin(lv + 1u, 3u);
            else if ((uses & 63u) == 63u && lv > 0u) lv -= 1u;
            keep[w] = 1u | (lv << 8);
        } else if (wanted[w] - total[w] <= L.tie[w]) { how[w] = 1; done[w] = true; keep[w] = 1u | (level[w] << 8); }     // the tie value itself
    }
    uint32_t target[2];
#pragma unroll
    for (int w = 0; w < 2; w++) target[w] = qh_target(wanted[w], a.wgs, keep[w] >> 8);                  // keys the NEXT list should hold
    if (t < 2) { L.bin_up[t] = 0u; L.bin_q[t] = 0u; }
    {   // the bin of the rank: half h of the workgroup scans side h (thread lt owns bins [8 lt, 8 lt + 8))
        const uint32_t half = rfl(t >> 8), lt = t & 255u, wl = rfl(lt >> 6);
        const uint4* h4 = reinterpret_cast<const uint4*>(&L.hist[half][0]);       // ((kQHBins + 64) * 4 B: 16-B aligned rows)
        const uint4 c0 = h4[2u * lt], c1 = h4[2u * lt + 1u];
        const uint32_t b[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const uint32_t sum = b[0] + b[1] + b[2] + b[3] + b[4] + b[5] + b[6] + b[7];
        const uint32_t inc = wave_scan_add(sum);
        if (lane == 63u) L.sc[half][wl] = inc;
        __syncthreads();
    QH_STAMP(12);
        uint32_t woff = 0u;
#pragma unroll
        for (uint32_t ww = 0; ww < 4; ww++) woff += ww < wl ? L.sc[half][ww] : 0u;
        const uint32_t excl = woff + inc - sum, r = half ? rank[1] : rank[0];
        if ((half ? how[1] : how[0]) == 2 && r >= excl && r < excl + sum) {             // one thread of the half
            uint32_t rin = r - excl, digit = lt * 8u;
#pragma unroll
            for (int j = 0; j < 7; j++) if (digit == lt * 8u + (uint32_t)j && rin >= b[j]) { rin -= b[j]; digit++; }
            L.bin[half] = digit; L.rin[half] = rin;
        }
        // the same prefix sums place two more ranks: total - target (everything from that bin up is what the next list should be)
        // and total / 4 (how densely the keys lie just above the threshold, should the list have to grow)
        const uint32_t tot = half ? total[1] : total[0], tgt = half ? target[1] : target[0];
        if ((half ? how[1] : how[0]) == 2 && sum != 0u) {
            auto place = [&](uint32_t rr) {
                uint32_t rin = rr - excl, digit = lt * 8u;
#pragma unroll
                for (int j = 0; j < 7; j++) if (digit == lt * 8u + (uint32_t)j && rin >= b[j]) { rin -= b[j]; digit++; }
                return digit;
            };
            if (tot > tgt && tot - tgt >= excl && tot - tgt < excl + sum) L.bin_up[half] = place(tot - tgt);
            if (tot < tgt && tot / 4u >= excl && tot / 4u < excl + sum) L.bin_q[half] = place(tot / 4u);
        }
    }
    __syncthreads();
    QH_STAMP(13);
#pragma unroll
    for (int w = 0; w < 2; w++) {                                       // next call's thresholds (on the key' axis, then back)
        T_next[w] = T[w];
        if (how[w] != 2) continue;
        uint32_t Tpn = Tp[w];
        if (total[w] > target[w]) Tpn = Tp[w] + (L.bin_up[w] << kQHDigitShift);         // exact: the hist says how many keys lie above
        else if (total[w] < target[w]) {                                   // extrapolated from the density of the lowest quarter of the list
            const float span = (float)((L.bin_q[w] + 1u) << kQHDigitShift);                 // (< 2^22: exact; the rest is an estimate anyway)
            const float want_more = (float)(target[w] - total[w]) * span * 4.f / (float)umax(total[w], 1u);
            const uint32_t delta = (uint32_t)fminf(want_more, 8.f * span);
            Tpn = Tp[w] > delta ? Tp[w] - delta : 0u;
        }
        T_next[w] = w ? ~Tpn : Tpn;
    }
#pragma unroll
    for (int w = 0; w < 2; w++) {
        if (how[w] != 2) continue;
        const uint32_t bin = L.bin[w];
        // key' lies in the bin <=> key' - (T' + 1 + (bin << shift)) < 2^shift (the saturating last bin: no upper end)
        const uint32_t lo = Tp[w] + 1u + (bin << kQHDigitShift), width = bin == kQHBins - 1u ? 0xFFFFFFFFu - lo : (1u << kQHDigitShift) - 1u;
        auto take = [&](uint32_t kp) { const uint32_t at = atomicAdd(&L.nsurv[w], 1u); if (at < kQHSurvCap) L.surv[w][at] = kp; };
        {
            const uint32_t cw = c[w] <= kQHThreadKeys ? c[w] : 0u;
            const uint32_t cmax = wave_all_max(cw);
            const uint32_t flip = w ? 0xFFFFFFFFu : 0u;
#pragma unroll
            for (uint32_t i = 0; i < kQHThreadKeys / 4; i++) {
                if (4u * i < cmax) {
                    const uint32_t k0 = k4[w][i].x ^ flip, k1 = k4[w][i].y ^ flip, k2 = k4[w][i].z ^ flip, k3 = k4[w][i].w ^ flip;
                    const bool m0 = k0 - lo <= width && 4u * i + 0u < cw, m1 = k1 - lo <= width && 4u * i + 1u < cw;
                    const bool m2 = k2 - lo <= width && 4u * i + 2u < cw, m3 = k3 - lo <= width && 4u * i + 3u < cw;
                    if (m0 | m1 | m2 | m3) {                               // rare: one divergent region per four keys
                        if (m0) take(k0);
                        if (m1) take(k1);
                        if (m2) take(k2);
                        if (m3) take(k3);
                    }
                }
            }
        }
        for_slot_keys(w, [&](uint32_t kp, bool valid) { if (valid && kp - lo <= width) take(kp); });
        for (uint32_t i = t; i < nbigkeys[w]; i += kQHBlock) { const uint32_t kp = L.big[w][i] ^ (w ? 0xFFFFFFFFu : 0u); if (kp - lo <= width) take(kp); }
    }
    __syncthreads();
    QH_STAMP(14);
    {   // wavefront 0 finishes the hi side, wavefront 4 (another SIMD) the lo side
        const uint32_t wid = rfl(t >> 6), side = wid >> 2;
        const uint32_t ns = side ? L.nsurv[1] : L.nsurv[0];
        if ((wid & 3u) == 0u && (side ? how[1] : how[0]) == 2 && ns >= 1u && ns <= kQHSurvCap) {
            const uint32_t kp = wave_select(L.surv[side], ns, side ? L.rin[1] : L.rin[0], L.wavehist[side]);
            if (lane == 0u) L.result[side] = side ? ~kp : kp;
        }
    }
    __syncthreads();
    QH_STAMP(15);
#pragma unroll
    for (int w = 0; w < 2; w++) {
        if (how[w] != 2) continue;
        const uint32_t ns = L.nsurv[w];
        if (ns >= 1u && ns <= kQHSurvCap) key_out[w] = L.result[w];
        else { done[w] = false; keep[w] = 0u; }                        // a bin too crowded for one wavefront (ties, saturation): exact passes
    }
}

// One CHUNK of the tensor: `rows` rows of kQHBlock float4 starting at row c * rows (a multiple of four rows; the last chunk also
// owns the n % 4 elements behind the last float4): on_tile(sample, valid) once per four rows, on_elem(value, valid) for every
// slot -- trip counts are block uniform.  Two register tiles of four rows ping-pong.
template <typename FT, typename FE>
__device__ __forceinline__ void hot_walk_chunk(const float* __restrict__ x, uint32_t n, uint32_t c, uint32_t chunks, uint32_t rows, FT on_tile, FE on_elem) {
    const uint32_t nvec = n >> 2;
    const float4* xv = reinterpret_cast<const float4*>(x);
    const uint32_t v_begin = c * rows * kQHBlock + threadIdx.x, groups = rows / 4u;
    float4 ba[4], bb[4];
    auto fetch = [&](float4 (&b)[4], uint32_t grp) {
        const uint32_t v0 = v_begin + umin(grp, groups - 1u) * 4u * kQHBlock;
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) b[u] = gload4<false>(xv + umin(v0 + u * kQHBlock, nvec - 1u));        // nvec >= 1: n >= 2^18
    };
    auto consume = [&](const float4 (&b)[4], uint32_t grp) {
        const uint32_t v0 = v_begin + grp * 4u * kQHBlock;
        on_tile(b[0].x, v0 < nvec);
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            const bool in = v0 + u * kQHBlock < nvec;
            on_elem(b[u].x, in); on_elem(b[u].y, in); on_elem(b[u].z, in); on_elem(b[u].w, in);
        }
    };
    fetch(ba, 0);
    for (uint32_t grp = 0;;) {
        fetch(bb, grp + 1);
        consume(ba, grp);
        if (++grp >= groups) break;
        fetch(ba, grp + 1);
        consume(bb, grp);
        if (++grp >= groups) break;
    }
    if (c == chunks - 1u) {                                            // block uniform
        const uint32_t i = (nvec << 2) + threadIdx.x;
        const bool in = i < n;
        const float v = in ? gload1(x + i) : 0.f;
        on_tile(v, in);
        on_elem(v, in);
    }
}

// The exact passes do not depend on which workgroups are resident: they hand out their chunks through a counter, so a level is
// complete when its chunks are -- whoever counted them.  (A grid barrier that waits for WORKGROUPS would hang as soon as two of
// these launches, from two streams, share the chip.)
__global__ __launch_bounds__(kQHBlock) void quantile_hot_select_kernel(const QHot a) {
    __shared__ union { QHSelLds s; QHExactLds e; } L;
    __shared__ uint32_t scratch[32], sel[2], bcast[4];
    const uint32_t n = a.n;
    uint32_t* ws = a.ws;
    // The selecting role goes to the workgroup that ARRIVES first (a ticket), the second side of a split select to the second.
    // Rounds of measurements with "workgroup 0 selects, the others wait" ended in launches of 7 .. 55 s: with three queues busy
    // (two of these launches on two streams beside a copy on a third) workgroup 0 of a grid is NOT always resident when its
    // siblings are, and two launches whose pollers hold each other's CUs only move again when the queue scheduler time-slices
    // them (tools/quantile_soak.py single, profiles/r06_quantile_soak.txt).  Nothing here depends on dispatch order now.
#ifdef PPQHIP_QH_TIMING
    const uint32_t stamp0 = (uint32_t)wall_clock64();
#endif
    const uint4 hdr = *reinterpret_cast<const uint4*>(ws);
    if (threadIdx.x == 0) bcast[0] = __hip_atomic_fetch_add(&ws[kQHRoleTicket], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    QHRecs R;
    __syncthreads();
    const uint32_t arrival = bcast[0];
    __syncthreads();
    const uint32_t role = arrival == 0u ? 0u : ((a.split && arrival == 1u) ? 1u : 0xFFFFFFFFu);
    // the records are requested once the role is known, the owner's own side's heads with them.  (Until the round's last day the 16
    // lowest workgroups requested them together with the ticket, "in practice the first to arrive": the stamps say the first arrivals
    // are workgroups 5, 47, 101, 135, 149, 237 .. -- hardly ever one of those; 0 / 8 / 16 / 32 / 64 speculators measured alike.)
    if (role < 2u) hot_load_records(a, R, role);
    else { R.r4[0] = R.r4[1] = R.r4[2] = R.r4[3] = make_uint4(0u, 0u, 0u, 0u); R.head_side = 2u; }
    const bool enabled = hdr.x != 0u;
    const uint32_t T[2] = {hdr.y, hdr.z};
    uint32_t key_sel[2] = {T[0], T[1]};
    bool done_sel[2] = {false, false};
    uint32_t keep_sel[2] = {0u, 0u};                                 // the sides' valid words for the hint (0: drop; 1 | level << 8)
    const uint32_t level[2] = {(hdr.x >> 8) & 3u, (hdr.x >> 16) & 3u};
    uint32_t T_next[2] = {T[0], T[1]};
    // ---- the decisions: every side has an OWNER that selects it, writes its results and publishes one word; everybody else polls ----
    // One selecting workgroup owns both sides; of two, the first arrival owns the hi side and the second the lo side -- if it is
    // there: it claims the side FIRST (a compare-and-swap), and the first arrival, done with its own side, claims the lo side for the
    // exact passes should nobody have (a workgroup that is not resident must never be waited for).  Nothing is handed from one
    // owner to the other: each writes its side of `dest` and of the hint itself (the hi side's owner also the words they share), and an
    // owner whose sides are settled returns at once.  (Until round 6's last day the lo side's result travelled to the first arrival
    // through a flag: 3.4 us of waiting on B x 32, profiles/r06_quantile_select_stamps.txt.)
    uint32_t open_mask;
    uint32_t own = role == 0u ? (a.split ? 1u : 3u) : 0u;            // sides this workgroup owns (block uniform)
    if (role == 1u) {
        if (threadIdx.x == 0) {
            uint32_t expected = 0u;
            bcast[0] = __hip_atomic_compare_exchange_strong(&ws[kQHLoClaim], &expected, 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
        }
        __syncthreads();
        own = bcast[0] ? 2u : 0u;
        __syncthreads();
    }
    if (role == 0u) {
#ifdef PPQHIP_QH_TIMING
        if (threadIdx.x == 0) { a.ws[32] = stamp0; }
#endif
        QH_STAMP(1);
    }
    // ONE call site for both roles (the function is a few thousand instructions, inlined: a second copy costs registers and scratch)
    if (own != 0u && enabled) hot_select_records(a, R, T, own, L.s, key_sel, done_sel, keep_sel, T_next, level, hdr.w);
    if (role == 0u) {
        QH_STAMP(6);
        if (a.split) {                               // is the lo side taken?  If the second arrival has not even started, it stays OPEN and is
            if (threadIdx.x == 0) {                  // this workgroup's: the exact passes settle it (never seen outside a chip shared with other queues)
                uint32_t expected = 0u;
                bcast[0] = __hip_atomic_compare_exchange_strong(&ws[kQHLoClaim], &expected, 2u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) ? 1u : 0u;
            }
            __syncthreads();
            if (bcast[0]) own |= 2u;
            __syncthreads();
        }
    }
    if (own != 0u && threadIdx.x == 0) {
        // the decisions FIRST (a few hundred workgroups are waiting for them; the stores behind them queue in order), then the settled
        // sides' results and hint words -- nobody reads those before the launch ends; open sides: after the exact passes (below)
        if (own & 1u) __hip_atomic_store(&ws[kQHDecision], done_sel[0] ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (own & 2u) __hip_atomic_store(&ws[kQHLoFlag], done_sel[1] ? 1u : 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t* H = a.hint;
        if ((own & 1u) && done_sel[0]) { a.dest[0] = key2f(key_sel[0]); H[kHValidHi] = keep_sel[0]; H[kHTHi] = keep_sel[0] ? T_next[0] : T[0]; }
        if ((own & 2u) && done_sel[1]) { a.dest[1] = key2f(key_sel[1]); H[kHValidLo] = keep_sel[1]; H[kHTLo] = keep_sel[1] ? T_next[1] : T[1]; }
        if (role == 0u) {
            H[kHN] = n; H[kHKHi] = a.k_hi; H[kHKLo] = a.k_lo;
            if (enabled && done_sel[0]) H[kHUses] = hdr.w + 1u;
        }
    }
    {
        const uint32_t mine_open = ((own & 1u) && !done_sel[0] ? 1u : 0u) | ((own & 2u) && !done_sel[1] ? 2u : 0u);
        if (own != 0u && mine_open == 0u) { QH_STAMP(7); return; }   // an owner with nothing open is done: the exact passes (if the other side needs
                                                                     // them) hand their chunks out through a counter, whoever is there takes them
        if (threadIdx.x == 0) {                      // everybody else needs BOTH decisions: one open mask for all who count
            uint32_t dh, dl;
            if (own == 0u) __builtin_amdgcn_s_sleep(PPQHIP_QH_POLL_FIRST);           // the decisions are microseconds away
            while ((dh = __hip_atomic_load(&ws[kQHDecision], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) __builtin_amdgcn_s_sleep(PPQHIP_QH_POLL_SLEEP);
            while ((dl = __hip_atomic_load(&ws[kQHLoFlag], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) __builtin_amdgcn_s_sleep(PPQHIP_QH_POLL_SLEEP);
            bcast[3] = (dh == 2u ? 1u : 0u) | (dl == 2u ? 2u : 0u);
        }
        __syncthreads();
        open_mask = bcast[3];
        if (open_mask == 0u) return;
    }
    // ---- not settled: exact radix select over the whole tensor (12 + 12 + 8 key bits) ----
    // prefix / rank per side after each level; every workgroup arrives at the same numbers from the same global histograms
    uint32_t top[2] = {0u, 0u}, r0k[2] = {0u, 0u}, p24[2] = {0u, 0u}, r24[2] = {0u, 0u}, low[2] = {0u, 0u};
    if (open_mask != 0u) {
        __syncthreads();
        uint32_t* he = L.e.h;
        const uint32_t kk[2] = {a.k_hi, a.k_lo};
        // chunks of 8 .. 256 rows (64 KB .. 2 MB), about four per workgroup, handed out by a counter (a returning device atomic is
        // a 1.8 us round trip: one per 64 KB, not overlapped, held the passes at 3.5 TB/s)
        const uint32_t G = gridDim.x, total_rows = ((n >> 2) + kQHBlock - 1u) / kQHBlock;
        uint32_t chunk_rows = (total_rows + 4u * G - 1u) / (4u * G);
        chunk_rows = umin(256u, umax(8u, (chunk_rows + 3u) & ~3u));
        const uint32_t chunks = (total_rows + chunk_rows - 1u) / chunk_rows;
        for (int level = 0; level < 3; level++) {
            if (role == 0u) QH_STAMP(16 + 4 * level);
            for (uint32_t i = threadIdx.x; i < 2u * (kQ1 + kQTrash); i += kQHBlock) he[i] = 0u;
            __syncthreads();
            uint32_t mine = 0;
            const int shift = level == 1 ? 8 : 0, pshift = level == 1 ? 20 : 8;
            const uint32_t dmask = level == 1 ? 0xFFFu : 0xFFu;
            const int nb = level == 2 ? kQ3 : kQ1;
            // a settled side matches nothing: no prefix has bit 31 set after the shift
            const uint32_t p_hi = (open_mask & 1u) ? (level == 1 ? top[0] : p24[0]) : 0xFFFFFFFFu;
            const uint32_t p_lo = (open_mask & 2u) ? (level == 1 ? top[1] : p24[1]) : 0xFFFFFFFFu;
            WaveBinCounter<false, true, true> acc;
            acc.init(reinterpret_cast<int*>(he), kQ1);
            HotCounter hi_c, lo_c;
            hi_c.init(he, nb);
            lo_c.init(he + kQ1 + kQTrash, nb);
            // every chunk comes from the counter (a chunk owned by a workgroup that is not resident would stall the level); the NEXT
            // ticket is requested before the current chunk is walked, so only the first round trip of a level is exposed
            uint32_t ticket = 0;
            if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(&ws[kQHNext + level], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (;;) {
                if (threadIdx.x == 0) bcast[0] = ticket;
                __syncthreads();
                const uint32_t c = bcast[0];
                __syncthreads();
                if (c >= chunks) break;
                if (threadIdx.x == 0) ticket = __hip_atomic_fetch_add(&ws[kQHNext + level], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                mine++;
                if (level == 0) {
                    hot_walk_chunk(a.x, n, c, chunks, chunk_rows,
                                   [&](float v, bool in) { acc.elect((int)(f2key(v) >> 20), in); },
                                   [&](float v, bool in) { acc.template commit<false>((int)(f2key(v) >> 20), in); });
                } else {
                    hot_walk_chunk(a.x, n, c, chunks, chunk_rows,
                                   [&](float v, bool in) {
                                       const uint32_t key = f2key(v);
                                       hi_c.elect((int)((key >> shift) & dmask), in && (key >> pshift) == p_hi);
                                       lo_c.elect((int)((key >> shift) & dmask), in && (key >> pshift) == p_lo);
                                   },
                                   [&](float v, bool in) {
                                       const uint32_t key = f2key(v);
                                       const int d = (int)((key >> shift) & dmask);
                                       if (in && (key >> pshift) == p_hi) hi_c.add(d);
                                       if (in && (key >> pshift) == p_lo) lo_c.add(d);
                                   });
                }
            }
            if (role == 0u) QH_STAMP(17 + 4 * level);
            if (level == 0) acc.flush_hot(); else { hi_c.flush(); lo_c.flush(); }
            __syncthreads();
            if (mine) {                             // this workgroup's counts -> the global histograms of the level
                for (int i = threadIdx.x; i < nb; i += kQHBlock) {
                    const uint32_t c0 = he[i], c1 = he[kQ1 + kQTrash + i];
                    if (level == 0) { if (c0) atomicAdd(&ws[kQHOffH0 + i], c0); }
                    else {
                        uint32_t* Hg = ws + (level == 1 ? kQHOffH1 : kQHOffH2);
                        if (c0) atomicAdd(&Hg[i], c0);
                        if (c1) atomicAdd(&Hg[nb + i], c1);
                    }
                }
            }
            // the level is complete when all its chunks are counted: publish mine, wait for the rest
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (role == 0u) QH_STAMP(18 + 4 * level);
            if (threadIdx.x == 0) {
                // (what this workgroup published are device atomics, drained above: no L2 write-back to wait for -- agent atomics on both
                //  sides of a hand-off are a valid form, MI355X_MICROARCH.md "Valid forms"; a release fence here was 1.7 us per level)
                if (mine) __hip_atomic_fetch_add(&ws[kQHDone + level], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while (__hip_atomic_load(&ws[kQHDone + level], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < chunks) __builtin_amdgcn_s_sleep(8);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            }
            __syncthreads();
            if (role == 0u) QH_STAMP(19 + 4 * level);
            {   // the bin of each open side's rank: half h of the workgroup scans side h's histogram of the level (both at once)
                const uint32_t half = rfl(threadIdx.x >> 8), lt = threadIdx.x & 255u, wl = rfl(lt >> 6), lane = threadIdx.x & 63u;
                const uint32_t nbins = level == 2 ? (uint32_t)kQ3 : (uint32_t)kQ1, per = nbins / 256u;          // 16 or 1 bins per thread
                const uint32_t* Hs = level == 0 ? ws + kQHOffH0 : (level == 1 ? ws + kQHOffH1 + half * kQ2 : ws + kQHOffH2 + half * kQ3);
                const uint32_t want = level == 0 ? (half ? kk[1] : kk[0]) : (level == 1 ? (half ? r0k[1] : r0k[0]) : (half ? r24[1] : r24[0]));
                // the histogram goes through LDS: 16-B loads at consecutive addresses (thread t reading its 16 bins straight from
                // global memory is sixteen loads of 64 scattered lines each -- this stage took 6.6 us of a 20 us level)
                {
                    const uint4* H4 = reinterpret_cast<const uint4*>(Hs);
                    uint4* he4 = reinterpret_cast<uint4*>(he + half * (kQ1 + kQTrash));
                    for (uint32_t i = lt; i < nbins / 4u; i += 256u) he4[i] = H4[i];
                }
                __syncthreads();
                const uint32_t* hl = he + half * (kQ1 + kQTrash);
                uint32_t b[16], sum = 0u;
#pragma unroll
                for (uint32_t j = 0; j < 16; j++) { b[j] = j < per ? hl[lt * per + j] : 0u; sum += b[j]; }
                const uint32_t inc = wave_scan_add(sum);
                if (lane == 63u) scratch[half * 4u + wl] = inc;
                __syncthreads();
                uint32_t woff = 0u;
#pragma unroll
                for (uint32_t ww = 0; ww < 4; ww++) woff += ww < wl ? scratch[half * 4u + ww] : 0u;
                const uint32_t excl = woff + inc - sum;
                if (want >= excl && want < excl + sum) {                // one thread per half
                    uint32_t rin = want - excl, digit = lt * per;
#pragma unroll
                    for (uint32_t j = 0; j < 15; j++) if (j + 1u < per && digit == lt * per + j && rin >= b[j]) { rin -= b[j]; digit++; }
                    scratch[8u + half * 2u] = digit; scratch[9u + half * 2u] = rin;
                }
                __syncthreads();
#pragma unroll
                for (int w = 0; w < 2; w++) {
                    if (!(open_mask & (1u << w))) continue;
                    const uint32_t digit = scratch[8 + 2 * w], rin = scratch[9 + 2 * w];
                    if (level == 0) { top[w] = digit; r0k[w] = rin; }
                    else if (level == 1) { p24[w] = (top[w] << 12) | digit; r24[w] = rin; }
                    else low[w] = digit;
                }
                __syncthreads();
            }
        }
    }
    if ((own & open_mask) == 0u) return;
    // ---- the owner of a side that went through the exact passes writes its result and its half of the hint ----
    uint32_t out_key[2], out_valid[2], out_T[2];
    // (one body, instantiated per side: as a loop the compiler stopped unrolling it once the select grew, and every array indexed by
    //  the side -- thresholds, keys, valid words -- moved to scratch)
    auto side_out = [&](auto W) __attribute__((always_inline)) {
        constexpr int w = decltype(W)::value;
        if (!(own & open_mask & (1u << w))) { out_key[w] = 0u; out_valid[w] = 0u; out_T[w] = 0u; return; }      // (block uniform; written above, or another owner's)
        // the side went through the exact passes: leave a threshold that works (rules of F2's and F3's tails)
        const uint32_t V = (p24[w] << 8) | low[w];
        out_key[w] = V;
        const uint32_t k = w ? a.k_lo : a.k_hi;
        const uint32_t inb = ws[kQHOffH0 + top[w]];
        const uint32_t outer = w ? k - r0k[w] : n - (k - r0k[w]) - inb;
        const uint32_t wanted = w ? k + 1u : n - k;
        // a hint that was in use and did not settle this side: the stream is restless, the longest lists from here on (qh_target)
        const uint32_t lv = enabled ? 3u : 0u;
        const uint32_t target = qh_target(wanted, a.wgs, lv);
        const uint32_t limit = umin(q_list_limit(wanted, quantile_spec_cap(n)), kQHListMax);
        const uint32_t need_in = target > outer ? target - outer : 1u;
        const uint32_t r = w ? umin(inb, need_in) - 1u : (inb > need_in ? inb - need_in : 0u);
        select_bin<kQHBlock>(ws + kQHOffH1 + w * kQ2, kQ2, r, scratch, sel);
        const uint32_t m = sel[0], q24 = (top[w] << 12) | m;
        uint32_t listed, Tn;
        bool ok;
        if (w) { listed = outer + (r - sel[1]) + ws[kQHOffH1 + kQ2 + m]; ok = q24 < 0xFFFFFFu; Tn = (q24 + 1u) << 8; }
        else { listed = outer + inb - (r - sel[1]); ok = q24 > 0u; Tn = (q24 << 8) - 1u; }
        ok = ok && listed <= limit;
        if (need_in > inb) {
            // the bucket of the answer (1/8 of a binade) does not hold the list this level asks for: whole first-level buckets beyond
            // it, as many as it takes (coarse -- a bucket can double the list -- and exact again after the next settled call)
            uint32_t cum = outer + inb, b = top[w];
            for (int step = 0; step < 16 && cum < target; step++) {
                if (w ? b >= 0xFFEu : b <= 1u) break;
                b = w ? b + 1u : b - 1u;
                cum += ws[kQHOffH0 + b];
            }
            // (never a list the select could not hold even if it were spread evenly: the same data would fail again, and again)
            const uint32_t room = umin(kQHRoomPerWg * a.wgs, kQHListMax);
            if (b != top[w] && cum <= umin(limit, room - room / 4u)) { listed = cum; ok = true; Tn = w ? (b + 1u) << 20 : (b << 20) - 1u; }
        }
        const uint32_t mult = ws[kQHOffH2 + w * kQ3 + low[w]];
        if (mult / 16u >= wanted + 16u) { Tn = V; ok = true; }          // a heavy tie: the threshold ON the value (1 key in 8 is counted)
        out_valid[w] = ok ? (1u | (lv << 8)) : 0u; out_T[w] = Tn;
        __syncthreads();
    };
    side_out(std::integral_constant<int, 0>{});
    side_out(std::integral_constant<int, 1>{});
    if (threadIdx.x == 0) {
        uint32_t* H = a.hint;
        if (own & open_mask & 1u) { a.dest[0] = key2f(out_key[0]); H[kHValidHi] = out_valid[0]; H[kHTHi] = out_T[0]; }
        if (own & open_mask & 2u) { a.dest[1] = key2f(out_key[1]); H[kHValidLo] = out_valid[1]; H[kHTLo] = out_T[1]; }
    }
    QH_STAMP(7);
}

#ifndef PPQHIP_Q_HOT
#define PPQHIP_Q_HOT 1
#endif
#ifndef PPQHIP_QH_SMALL_ELEMS
#define PPQHIP_QH_SMALL_ELEMS (4ll << 20)       // up to here every load of a workgroup's share is issued up front (<= 8 per lane)
#endif
#ifndef PPQHIP_QH_SPLIT_WANTED
#define PPQHIP_QH_SPLIT_WANTED 2048
#endif
#ifndef PPQHIP_QH_HEADS_MIN
#define PPQHIP_QH_HEADS_MIN 4u                 // expected keys per filter workgroup and side from which the heads travel with the records
#endif
#ifndef PPQHIP_QH_NT_ELEMS
#define PPQHIP_QH_NT_ELEMS (48ll << 20)
#endif
static bool quantile_hot_enabled() {
#ifdef PPQHIP_DEV_KNOBS                     // measurement builds only: A/B against the general sequence
    if (const char* e = getenv("PPQHIP_DEV_Q_HOT")) return atoi(e) != 0;
#endif
    return true;
}
// A hint this process hands over for the first time is almost always a fresh one (an observer's first batch), and the two-launch
// path has only its exact passes for a tensor without usable thresholds: three reads behind LDS histograms, 73 us on B / 221 us on
// B x 32 where the general sequence -- sample, thresholds, filter, select -- takes 35 / 73 us and leaves the same kind of hint
// behind.  So the first call on a hint ADDRESS goes through the sequence; every later one through the two launches.  The memo only
// ever chooses between two exact paths: an address met again after its tensor was freed and zeroed costs one call of exact passes,
// a valid hint met for the first time (written by the multi-tensor entry point) costs one call of the sequence from its hint.
static bool quantile_hint_met_before(const uint32_t* hint) {
    static std::mutex lock;
    static std::unordered_set<const void*> met;
    std::lock_guard<std::mutex> guard(lock);
    if (met.size() > (1u << 16)) met.clear();
    return !met.insert((const void*)hint).second;
}
static void quantile_hot_launch(const QHot& a0, hipStream_t s) {
    QHot a = a0;
    const uint32_t full_rows = (a.n >> 2) / kQHBlock;
    uint32_t cap = (uint32_t)num_cu() * 2u;
    if (cap > kQHMaxWg) cap = kQHMaxWg;
    if ((int64_t)a.n <= PPQHIP_QH_SMALL_ELEMS) {
        uint32_t g = (full_rows + 1) / 2;
        if (g > cap) g = cap;
        if (g < 1) g = 1;
        const uint32_t share = (full_rows + g - 1) / g;
        a.wgs = g;
        if (share <= 2) hipLaunchKernelGGL((quantile_hot_filter_kernel<2, false, false>), dim3(g), dim3(kQHBlock), 0, s, a);
        else if (share <= 4) hipLaunchKernelGGL((quantile_hot_filter_kernel<4, false, false>), dim3(g), dim3(kQHBlock), 0, s, a);
        else hipLaunchKernelGGL((quantile_hot_filter_kernel<8, false, false>), dim3(g), dim3(kQHBlock), 0, s, a);
    } else {
        uint32_t g = full_rows / 4;
        if (g > cap) g = cap;
        if (g < 1) g = 1;
        a.wgs = g;
        // a filter workgroup is expected to list ~1.5 x wanted / g keys per side: more than the record holds -> the heads travel with
        // the records; lists of thousands of keys -> the two sides are selected by two workgroups
        const uint32_t wanted = a.n - a.k_hi > a.k_lo + 1u ? a.n - a.k_hi : a.k_lo + 1u;
        a.heads = (wanted + wanted / 2u) / g >= PPQHIP_QH_HEADS_MIN ? 1u : 0u;
        a.split = (wanted >= PPQHIP_QH_SPLIT_WANTED && num_cu() >= 2) ? 1u : 0u;
        if ((int64_t)a.n >= PPQHIP_QH_NT_ELEMS) hipLaunchKernelGGL((quantile_hot_filter_kernel<2, true, true>), dim3(g), dim3(kQHBlock), 0, s, a);
        else hipLaunchKernelGGL((quantile_hot_filter_kernel<2, true, false>), dim3(g), dim3(kQHBlock), 0, s, a);
    }
    // one workgroup per CU -- the exact passes need the chip -- but half of that for tensors of a few MB: 128 tickets and pollers
    // instead of 256 retire 0.4 us earlier on B, and its exact passes are FASTER with them (61 instead of 73 us: fewer flushes into
    // the shared histograms); from B x 8 up the smaller grid costs the exact passes dearly (B x 32: 219 -> 319 us)
    uint32_t gs = (uint32_t)num_cu();
    if ((int64_t)a.n <= PPQHIP_QH_SMALL_ELEMS && gs > 128u) gs = 128u;
    if (PPQHIP_QH_SELECT_WGS > 0) gs = (uint32_t)PPQHIP_QH_SELECT_WGS;
    hipLaunchKernelGGL(quantile_hot_select_kernel, dim3(gs), dim3(kQHBlock), 0, s, a);
}

static int validate(int64_t n, const char* what) {
    if (n <= 0) { set_error("%s: tensor is empty", what); return PPQHIP_ERR_INVALID_VALUE; }
    if (n > 0x7fffffffLL) { set_error("%s: too many elements", what); return PPQHIP_ERR_INVALID_VALUE; }
    return PPQHIP_OK;
}

// index rule of _Quantile_T, sort.cu:13-19: __float2int_rn(num_of_elements * q), clipped to [0, n-1]
static uint32_t quantile_pos(int64_t n, float f) {
    float p = nearbyintf((float)n * f);
    if (!(p > 0.f)) return 0u;                      // also NaN
    if (p >= (float)(n - 1)) return (uint32_t)(n - 1);
    return (uint32_t)p;
}

static int quantile_multi_impl(const ppqhip_quantile_job* jobs, int num_jobs, float q, void* workspace, hipStream_t s,
                               const char* what) {
    uint8_t* prefix = (uint8_t*)workspace;
    uint32_t* fixed = (uint32_t*)(prefix + kQPrefBytes);
    uint32_t* spec_at = fixed + (size_t)num_jobs * kQWords;     // the filter lists live behind all fixed parts
    for (int seq_base = 0; seq_base < num_jobs; seq_base += kQMaxJobs) {
        const int count = (num_jobs - seq_base) < kQMaxJobs ? (num_jobs - seq_base) : kQMaxJobs;
        uint32_t tiles = 0, units = 0;
        int64_t elems = 0;
        uint32_t wanted_max = 0;                          // most keys any side of any job needs listed
        for (int base = 0; base < count; base += kQInitMax) {
            QInitArgs a;
            a.count = (uint32_t)((count - base) < kQInitMax ? (count - base) : kQInitMax);
            a.base = (uint32_t)base; a.tile_base = tiles; a.unit_base = units;
            a.prefix = prefix; a.fixed = fixed + (size_t)(seq_base + base) * kQWords; a.spec = spec_at;
            for (uint32_t k = 0; k < a.count; k++) {
                const ppqhip_quantile_job& src = jobs[seq_base + base + (int)k];
                const int64_t n = src.n;
                elems += n;
                auto pos = [n](float f) -> uint32_t { return quantile_pos(n, f); };
                QUpload& e = a.e[k];
                e.x = src.x; e.dest = src.dest; e.hint = src.hint; e.n = (uint32_t)n; e.k_hi = pos(q); e.k_lo = pos(1 - q); e.pad = 0;
                const uint32_t w_hi = e.n - e.k_hi, w_lo = e.k_lo + 1u;
                if (w_hi > wanted_max) wanted_max = w_hi;
                if (w_lo > wanted_max) wanted_max = w_lo;
                tiles += q_job_tiles(e.n, aligned16(src.x));
                units += q_job_units(e.n);
                spec_at += 2 * (size_t)quantile_spec_cap((uint64_t)n);
            }
            hipLaunchKernelGGL(quantile_init_kernel, dim3(a.count * kQInitSplit), dim3(kBlock), 0, s, a);
        }
        QSeq seq;
        seq.job = (const QJob*)(prefix + kQPrefTable);
        seq.first_tile = (const uint32_t*)(prefix + kQPrefTile);
        seq.first_unit = (const uint32_t*)(prefix + kQPrefUnit);
        seq.header = (uint32_t*)(prefix + kQPrefHeader);
        seq.fixed = fixed + (size_t)seq_base * kQWords;
        seq.count = (uint32_t)count; seq.total_tiles = tiles; seq.total_units = units;
        seq.all_open = elems >= kQSpeculateMinElems ? 0u : 1u;
        const uint32_t cus = (uint32_t)num_cu();
        if (!seq.all_open) {
            uint32_t gs = units < 1024u ? units : 1024u;
            hipLaunchKernelGGL(quantile_sample_kernel, dim3(gs), dim3(kBlock), 0, s, seq);      // returns on one load when no job is cold
            uint32_t gf = tiles / 2;                  // >= 2 tiles per workgroup
            if (gf < 1) gf = 1;
            if (gf > cus * kQFWgPerCu) gf = cus * kQFWgPerCu;
            hipLaunchKernelGGL(quantile_filter_kernel, dim3(gf), dim3(kQFBlock), 0, s, seq);
            hipLaunchKernelGGL(quantile_select_a_kernel, dim3(2 * (uint32_t)count), dim3(kQSABlock), 0, s, seq);
        }
        // (tiny all-open jobs -- below 256 K elements the exact passes ARE the algorithm -- keep their three parallel launches: one
        //  workgroup alone took 119 us instead of 53 for [1,3,224,224])
        if (count == 1 && !seq.all_open && elems <= PPQHIP_Q_SINGLE_ELEMS && wanted_max <= 4096u) {
            hipLaunchKernelGGL(quantile_f123_single_kernel, dim3(1), dim3(kBlock), 0, s, seq);
            continue;
        }
        uint32_t gF = tiles < cus * 4 ? tiles : cus * 4;
        if (gF < 1) gF = 1;
        hipLaunchKernelGGL(quantile_f1_kernel, dim3(gF), dim3(kBlock), 0, s, seq);
        hipLaunchKernelGGL(quantile_f2_kernel, dim3(gF), dim3(kBlock), 0, s, seq);
        hipLaunchKernelGGL(quantile_f3_kernel, dim3(gF), dim3(kBlock), 0, s, seq);
    }
    return finish_launch(what);
}

}  // namespace ppqhip

using namespace ppqhip;

extern "C" {

int64_t ppqhip_quantile_workspace_bytes(int64_t n) {
    // (also what ppqhip_isotone_t asks for: its 16 KB of partials fit the sequence prefix)
    const int64_t seq = (int64_t)kQPrefBytes + ((int64_t)kQWords + 2 * (int64_t)quantile_spec_cap((uint64_t)(n > 0 ? n : 0))) * 4;
    const int64_t hot = (int64_t)kQHWords * 4;       // the two-launch path of one hinted tensor lays the same memory out its own way
    return seq > hot ? seq : hot;
}

int ppqhip_quantile_t(const float* x, int64_t n, float q, float* dest, uint32_t* hint, void* workspace, void* stream) {
    if (int st = validate(n, "quantile_t")) return st;
    if (workspace == nullptr) { set_error("quantile_t: workspace is null"); return PPQHIP_ERR_INVALID_VALUE; }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_QUANTILE, 4.0 * (double)n, s);
#if PPQHIP_Q_HOT
    if (hint != nullptr && dest != nullptr && aligned16(x) && n >= kQSpeculateMinElems && quantile_hot_enabled()) {
        QHot a;
        a.x = x; a.dest = dest; a.hint = hint; a.ws = (uint32_t*)workspace; a.n = (uint32_t)n;
        a.k_hi = quantile_pos(n, q); a.k_lo = quantile_pos(n, 1 - q); a.wgs = 0; a.split = 0; a.heads = 0; a.pad0 = a.pad1 = 0;
        if (a.n - a.k_hi <= kQHWantedMax && a.k_lo + 1u <= kQHWantedMax && quantile_hint_met_before(hint)) {
            quantile_hot_launch(a, s);
            return finish_launch("quantile_t");
        }
    }
#endif
    ppqhip_quantile_job job;
    job.x = x; job.dest = dest; job.hint = hint; job.n = n;
    return quantile_multi_impl(&job, 1, q, workspace, s, "quantile_t");
}

int64_t ppqhip_quantile_multi_workspace_bytes(int num_jobs, int64_t total_elems) {
    // sequence prefix + fixed part per job + the filter lists: sum over jobs of 2 * clamp(n / 128, 16384, 2^20) keys (+ rounding)
    if (num_jobs <= 0) return 0;
    int64_t lists = (total_elems > 0 ? total_elems : 0) / 128;            // sum of n / 128 <= total / 128 ..
    if (lists > (int64_t)num_jobs << 20) lists = (int64_t)num_jobs << 20;       // .. and every list is capped at 2^20 keys
    return (int64_t)kQPrefBytes + ((int64_t)num_jobs * (kQWords + 2 * 16384 + 64) + 2 * lists) * 4;
}

void ppqhip_quantile_debug_layout(int64_t* out) {
    out[0] = (int64_t)kQPrefBytes; out[1] = kQWords; out[2] = kOffSel; out[3] = kOffSpec; out[4] = kOffTick;
    out[5] = (int64_t)kQPrefTable; out[6] = kPCnt; out[7] = kPTie;
}

void ppqhip_quantile_hot_layout(int64_t* out) {
    out[0] = (int64_t)kQHWords; out[1] = kQHOffRec; out[2] = kQHOffHeads; out[3] = kQHOffSlots; out[4] = kQHMaxWg; out[5] = kQHStage;
    out[6] = kQHZeroEnd; out[7] = kQHThreadKeys;
}

int ppqhip_quantile_t_multi(const ppqhip_quantile_job* jobs, int num_jobs, float q, void* workspace, void* stream) {
    if (num_jobs <= 0) return PPQHIP_OK;
    if (jobs == nullptr || workspace == nullptr) {
        set_error("quantile_t_multi: jobs / workspace is null"); return PPQHIP_ERR_INVALID_VALUE;
    }
    double bytes = 0.0;
    for (int k = 0; k < num_jobs; k++) {
        if (int st = validate(jobs[k].n, "quantile_t_multi")) return st;
        if (jobs[k].x == nullptr || jobs[k].dest == nullptr) {
            set_error("quantile_t_multi: job %d has a null pointer", k); return PPQHIP_ERR_INVALID_VALUE;
        }
        bytes += 4.0 * (double)jobs[k].n;
    }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_QUANTILE, bytes, s);
    return quantile_multi_impl(jobs, num_jobs, q, workspace, s, "quantile_t_multi");
}

}  // extern "C"
