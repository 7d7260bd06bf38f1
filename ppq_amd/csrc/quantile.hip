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
// ONE small tensor with an extreme q (a single-tensor call from the reference's percentile observer): 5 launches -- F1 F2 F3 are one
// launch of one workgroup there (quantile_f123_single_kernel).
#include <cmath>
#include <cstdlib>
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
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += (uint32_t)__shfl_xor((int)v, m, 64);
    return v;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = umin(v, (uint32_t)__shfl_xor((int)v, m, 64));
    return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v = umax(v, (uint32_t)__shfl_xor((int)v, m, 64));
    return v;
}
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
    return hint[kHValidHi] == 1u && hint[kHValidLo] == 1u && hint[kHN] == n && hint[kHKHi] == k_hi && hint[kHKLo] == k_lo;
}

// ---- block-wide helpers (THREADS = blockDim.x, a multiple of 64) --------------------------------------------------
// exclusive prefix of v over the workgroup + the total; scratch: THREADS / 64 words.  All threads call this.
template <int THREADS>
__device__ __forceinline__ void block_scan_excl(uint32_t v, uint32_t* scratch, uint32_t& excl, uint32_t& total) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, d, 64);
        if (lane >= d) inc += o;
    }
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

static int validate(int64_t n, const char* what) {
    if (n <= 0) { set_error("%s: tensor is empty", what); return PPQHIP_ERR_INVALID_VALUE; }
    if (n > 0x7fffffffLL) { set_error("%s: too many elements", what); return PPQHIP_ERR_INVALID_VALUE; }
    return PPQHIP_OK;
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
                // index rule of _Quantile_T, sort.cu:13-19: __float2int_rn(num_of_elements * q), clipped to [0, n-1]
                auto pos = [n](float f) -> uint32_t {
                    float p = nearbyintf((float)n * f);
                    if (!(p > 0.f)) return 0u;                      // also NaN
                    if (p >= (float)(n - 1)) return (uint32_t)(n - 1);
                    return (uint32_t)p;
                };
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
    return (int64_t)kQPrefBytes + ((int64_t)kQWords + 2 * (int64_t)quantile_spec_cap((uint64_t)(n > 0 ? n : 0))) * 4;
}

int ppqhip_quantile_t(const float* x, int64_t n, float q, float* dest, uint32_t* hint, void* workspace, void* stream) {
    if (int st = validate(n, "quantile_t")) return st;
    if (workspace == nullptr) { set_error("quantile_t: workspace is null"); return PPQHIP_ERR_INVALID_VALUE; }
    hipStream_t s = (hipStream_t)stream;
    LaunchScope scope(K_QUANTILE, 4.0 * (double)n, s);
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
