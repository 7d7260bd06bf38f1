/* TEST INFRASTRUCTURE ONLY -- C entry points onto the reference's OWN histogram / quantile kernel bodies.
 *
 * `sort_kernels.inc` is NOT in this repository: oracle/Makefile writes it (git-ignored, under oracle/_ref/)
 * by running extract_kernels.awk over $(REF)/ppq/csrc/cuda/sort.cu, i.e. it is the text of the reference's
 * `_Quantile_T`, `_Isotone_T`, `_Histogram_T`, `_Histogram_Asymmetric_T`, `_Histogram_C` (sort.cu:6-40,
 * 75-89, 113-139, 167-185) and nothing else; `common.cuh` resolves to the reference's header (CLIP,
 * KERNEL_LOOP).  ref_kernel_host.h supplies blockIdx / atomicAdd / __float2int_rn for one sequential
 * "thread".  What is restated here, because `<<<...>>>` launchers and thrust do not compile on a host:
 *   - the sort in front of `_Quantile_T` / `_Isotone_T` (sort.cu:49-52, 70: thrust::sort ascending) = std::sort;
 *   - the channel geometry of `Histogram_C` (sort.cu:204-208) is the caller's job (ref_sort.py);
 *   - every kernel runs as a 1 x 1 grid: KERNEL_LOOP then visits every element in order.
 * Output: oracle/_ref/libref_kernels.so (with ref_linear_shim.cc / ref_floating_shim.cc). */
#include "common.cuh"
#include "ref_kernel_host.h"
#include <algorithm>

#include "sort_kernels.inc"

extern "C" {

/* sort.cu:42-59 */
void ref_quantile_t(const float* source, int64_t n, float q, float* dest) {
    std::vector<float> v(source, source + n);
    std::sort(v.begin(), v.end());
    launch(1, 1, 1, [&] { _Quantile_T(v.data(), dest, n, q); });
}

/* The two positions `_Quantile_T` reads (sort.cu:13-14, 17-18), for ANY n up to 2^31 - 1, observed from the kernel
 * itself without materialising n floats: the "tensor" is a PROT_NONE reservation, the first read faults and the
 * fault address is max_pos; that page is then made readable (and filled with its in-page indices, for the case
 * that min_pos lies on the same page) and the kernel runs again for min_pos. */
}  // extern "C"
#include <setjmp.h>
#include <signal.h>
#include <sys/mman.h>
static sigjmp_buf g_jb;
static void* g_fault;
static void on_segv(int, siginfo_t* si, void*) { g_fault = si->si_addr; siglongjmp(g_jb, 1); }
extern "C" {
int ref_quantile_positions(int64_t n, float q, int64_t* pos) {
    const size_t page = 4096, bytes = (((size_t)n * 4 + page - 1) / page) * page;
    char* base = (char*)mmap(nullptr, bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (base == (char*)MAP_FAILED) return -1;
    struct sigaction sa = {}, old_segv, old_bus;
    sa.sa_sigaction = on_segv; sa.sa_flags = SA_SIGINFO | SA_NODEFER; sigemptyset(&sa.sa_mask);
    sigaction(SIGSEGV, &sa, &old_segv); sigaction(SIGBUS, &sa, &old_bus);
    float d[2] = {-1.f, -1.f};
    const float* src = (const float*)base;
    int rc = 0;
    pos[0] = pos[1] = -1;
    if (sigsetjmp(g_jb, 1) == 0) { launch(1, 1, 1, [&] { _Quantile_T(src, d, n, q); }); rc = -2; }   /* must fault */
    else pos[0] = ((char*)g_fault - base) / 4;
    if (rc == 0) {
        char* pg = base + ((size_t)(pos[0] * 4) / page) * page;
        mprotect(pg, page, PROT_READ | PROT_WRITE);
        for (int i = 0; i < 1024; i++) ((float*)pg)[i] = (float)i;
        if (sigsetjmp(g_jb, 1) == 0) {
            launch(1, 1, 1, [&] { _Quantile_T(src, d, n, q); });
            pos[1] = (pg - base) / 4 + (int64_t)d[1];                    /* same page: the kernel returned the in-page index */
        } else pos[1] = ((char*)g_fault - base) / 4;
    }
    sigaction(SIGSEGV, &old_segv, nullptr); sigaction(SIGBUS, &old_bus, nullptr);
    munmap(base, bytes);
    return rc;
}

/* sort.cu:61-73 */
void ref_isotone_t(const float* source, int64_t n, float* dest) {
    std::vector<float> v(source, source + n);
    std::sort(v.begin(), v.end());
    launch(1, 1, 1, [&] { _Isotone_T(v.data(), dest, n); });
}

/* sort.cu:91-111 */
void ref_hist_sym_t(const float* value, int64_t n, int64_t bins, float hist_scale, int clip_outliers, int* hist) {
    launch(1, 1, 1, [&] { _Histogram_T(n, bins, value, hist_scale, clip_outliers != 0, hist); });
}

/* sort.cu:141-165 */
void ref_hist_asym_t(float vmin, float vmax, const float* value, int64_t n, int64_t bins, int clip_outliers, int* hist) {
    launch(1, 1, 1, [&] { _Histogram_Asymmetric_T(vmin, vmax, n, bins, value, clip_outliers != 0, hist); });
}

/* sort.cu:187-218 */
void ref_hist_sym_c(const float* value, int64_t n, int64_t element_per_channel, int num_of_channel, int64_t bins,
                    float hist_scale, int clip_outliers, int* hist) {
    launch(1, 1, 1, [&] { _Histogram_C(n, element_per_channel, num_of_channel, bins, value, hist_scale, clip_outliers != 0, hist); });
}

}  // extern "C"
