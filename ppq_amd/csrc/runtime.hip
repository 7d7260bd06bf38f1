// runtime.hip -- error reporting, introspection and the hipEvent profiling aid of libppq_hip.so.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "common.hpp"

namespace ppqhip {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return PPQHIP_OK;
    set_error("HIP failure in %s: %s", what, hipGetErrorString(e));
    return PPQHIP_ERR_HIP;
}

int finish_launch(const char* what) { return check_hip(hipGetLastError(), what); }

const char* const kKernelNames[K_NUM] = {
    "fq_linear_t", "fq_linear_c", "fq_linear_t_bwd", "fq_linear_c_bwd", "fq_float_t", "fq_float_c",
    "fq_float_bwd", "hist_sym_t", "hist_asym_t", "hist_sym_c", "quantile_t", "isotone_t", "minmax_t",
    "minmax_c", "mse_search", "kl_losses", "tensor_clip", "rounding_loss", "channel_sum", "float_scale_search", "lsq_finish"};

int num_cu() {
    static std::mutex mu;
    static std::map<int, int> per_device;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return kNumCU;
    std::lock_guard<std::mutex> lk(mu);
    auto it = per_device.find(dev);
    if (it != per_device.end()) return it->second;
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = kNumCU;
    if (n > kNumCU) n = kNumCU;          // the persistent accumulators have kNumCU x k rows / slots
    per_device[dev] = n;
    return n;
}

// ---- per-(device, stream) scratch arena ---------------------------------------------------
// Kernels that need a few KiB..MiB of device scratch (two-stage reductions) take it from here:
// launches on one stream are ordered, so one buffer per (device, stream) is race-free.
static std::mutex g_scratch_mu;
static std::map<std::pair<int, hipStream_t>, std::pair<void*, size_t>> g_scratch;

void* scratch(hipStream_t stream, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    auto& slot = g_scratch[{dev, stream}];
    if (slot.second < bytes) {
        // growing means hipMalloc (+ a stream synchronize): neither may happen while `stream` is being captured into a
        // HIP graph.  Callers that capture warm the stream up with one eager run of the same work first (ppq_amd/lsq.py).
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (stream != nullptr && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) {
            set_error("scratch: %zu bytes needed on a stream that is being captured (run the same work once eagerly on this stream first)", bytes);
            return nullptr;
        }
        if (slot.first) {   // the old buffer may still be in use by queued work on this stream
            if (hipStreamSynchronize(stream) != hipSuccess) return nullptr;
            (void)hipFree(slot.first);
            slot = {nullptr, 0};
        }
        size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
        void* p = nullptr;
        if (hipMalloc(&p, want) != hipSuccess) { set_error("scratch: hipMalloc(%zu) failed", want); return nullptr; }
        slot = {p, want};
    }
    return slot.first;
}

// A second arena per (device, stream) whose contract is that it is ALL ZERO between launches: kernels that accumulate into it
// with atomics hand it back zeroed (the one-shot histogram's partial rows: hist_partial_reduce_kernel re-zeroes what it read).
// Zeroed once when allocated; fixed size (the largest user's need), so it never moves under a captured graph.
static std::map<std::pair<int, hipStream_t>, void*> g_zeroed;
void* zeroed_arena(hipStream_t stream, size_t bytes) {
    if (bytes > kZeroedArenaBytes) { set_error("zeroed_arena: %zu bytes asked of a %zu-byte arena", bytes, (size_t)kZeroedArenaBytes); return nullptr; }
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_scratch_mu);
    void*& p = g_zeroed[{dev, stream}];
    if (p == nullptr) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (stream != nullptr && hipStreamIsCapturing(stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return nullptr;   // (callers fall back)
        void* q = nullptr;
        if (hipMalloc(&q, kZeroedArenaBytes) != hipSuccess) { set_error("zeroed_arena: hipMalloc failed"); return nullptr; }
        if (hipMemset(q, 0, kZeroedArenaBytes) != hipSuccess) { (void)hipFree(q); set_error("zeroed_arena: hipMemset failed"); return nullptr; }
        p = q;
    }
    return p;
}

// ---- profiling aid ------------------------------------------------------------------------
struct ProfRecord {
    int id;
    double bytes;
    hipEvent_t start, stop;
};
static std::mutex g_prof_mu;
static bool g_prof_on = false;
static std::vector<ProfRecord> g_records;
static std::vector<hipEvent_t> g_pool;

static hipEvent_t get_event() {
    if (!g_pool.empty()) { hipEvent_t e = g_pool.back(); g_pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
}

LaunchScope::LaunchScope(KernelId id, double bytes, hipStream_t s) : slot(-1), stream(s) {
    if (!g_prof_on) return;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;         // events recorded into a graph cannot be timed: skip
    if (s != nullptr && hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRecord r; r.id = id; r.bytes = bytes; r.start = get_event(); r.stop = get_event();
    (void)hipEventRecord(r.start, stream);
    g_records.push_back(r);
    slot = (int)g_records.size() - 1;
}

LaunchScope::~LaunchScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_records[slot].stop, stream);
}

}  // namespace ppqhip

using namespace ppqhip;

extern "C" {

const char* ppqhip_last_error(void) { return g_err; }

int ppqhip_version(void) { return PPQHIP_ABI_VERSION; }   // include/ppq_hip.h; ppq_amd/_lib.py refuses any other value

int ppqhip_device_arch(char* buf, int n) {
    int dev = 0;
    if (int st = check_hip(hipGetDevice(&dev), "hipGetDevice")) return st;
    hipDeviceProp_t prop;
    if (int st = check_hip(hipGetDeviceProperties(&prop, dev), "hipGetDeviceProperties")) return st;
    snprintf(buf, n, "%s", prop.gcnArchName);
    return PPQHIP_OK;
}

int ppqhip_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return PPQHIP_OK;
}

double ppqhip_prof_event_overhead_us(void* stream, int pairs) {
    // elapsed time of an EMPTY start/stop event pair on this stream: what a bracketed launch reports
    // on top of the kernel's own begin->end duration (timestamp writes by the command processor).
    hipStream_t s = (hipStream_t)stream;
    if (pairs < 1) pairs = 1;
    std::vector<hipEvent_t> ev(2 * (size_t)pairs);
    for (auto& e : ev) if (hipEventCreate(&e) != hipSuccess) return -1.0;
    for (int i = 0; i < pairs; i++) {
        (void)hipEventRecord(ev[2 * i], s);
        (void)hipEventRecord(ev[2 * i + 1], s);
    }
    (void)hipStreamSynchronize(s);
    double total = 0.0;
    for (int i = 0; i < pairs; i++) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]);
        total += ms;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    return total * 1e3 / pairs;
}

int ppqhip_prof_collect(ppqhip_prof_entry* entries, int max_entries) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ppqhip_prof_entry agg[K_NUM];
    memset(agg, 0, sizeof(agg));
    for (int k = 0; k < K_NUM; k++) snprintf(agg[k].name, sizeof(agg[k].name), "%s", kKernelNames[k]);
    for (auto& r : g_records) {
        float ms = 0.f;
        (void)hipEventSynchronize(r.stop);
        (void)hipEventElapsedTime(&ms, r.start, r.stop);
        agg[r.id].launches += 1; agg[r.id].total_ms += ms; agg[r.id].total_bytes += r.bytes;
        g_pool.push_back(r.start); g_pool.push_back(r.stop);
    }
    g_records.clear();
    int n = 0;
    for (int k = 0; k < K_NUM && n < max_entries; k++)
        if (agg[k].launches > 0) entries[n++] = agg[k];
    return n;
}

}  // extern "C"
