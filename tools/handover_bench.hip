// Workgroup-to-workgroup hand-over latency on gfx950 (diagnostic, not part of the product): a chain of G workgroups inside ONE launch, each
// waiting for its predecessor's result, against the same chain as G dependent launches.
//   mode 0: flag (agent-scope release / acquire) + data read after the flag      -- two round trips per link
//   mode 1: the value itself is polled against a NaN sentinel                  -- one round trip per link
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ __launch_bounds__(256) void k_chain_flag(double *data, int *flag, int epoch, int n, int *fail) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ int ok;
    if (tid == 0) { ok = 1;
        if (b > 0) { int spins = 0; while (__hip_atomic_load(&flag[b - 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) { if (++spins > (1 << 20)) { ok = 0; *fail = 1; break; } __builtin_amdgcn_s_sleep(1); } } }
    __syncthreads();
    if (!ok) return;
    double x = 0.0;
    if (tid < n) { x = b > 0 ? __hip_atomic_load(&data[(size_t)(b - 1)*n + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (double)epoch;
        __hip_atomic_store(&data[(size_t)b*n + tid], x + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __syncthreads();
    if (tid == 0) __hip_atomic_store(&flag[b], epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// (a {value, tag} pair needs a 16-byte single-copy-atomic access the language does not offer; the one-round-trip form polls the VALUE itself
//  against a sentinel instead: the slot is NaN until the producer's value -- never NaN -- arrives)
__global__ __launch_bounds__(256) void k_chain_nan(double *data, int n, int *fail) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (tid >= n) return;
    double x = 0.0;
    if (b > 0) { int spins = 0;
        for (;;) { x = __hip_atomic_load(&data[(size_t)(b - 1)*n + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (x == x) break; if (++spins > (1 << 20)) { *fail = 1; return; } __builtin_amdgcn_s_sleep(1); } }
    __hip_atomic_store(&data[(size_t)b*n + tid], x + 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_fill_nan(double *data, size_t n) { size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; if (i < n) data[i] = __builtin_nan(""); }
__global__ __launch_bounds__(256) void k_link(double *data, int b, int n, int epoch) {          // one link of the chain as its own launch
    const int tid = threadIdx.x;
    if (tid < n) { const double x = b > 0 ? data[(size_t)(b - 1)*n + tid] : (double)epoch; data[(size_t)b*n + tid] = x + 1.0; }
}
int main() {
    const int n = 48, reps = 50;
    int *flag, *fail; double *data;
    CHECK(hipMalloc(&flag, 4096*sizeof(int))); CHECK(hipMalloc(&fail, sizeof(int))); CHECK(hipMalloc(&data, 4096*(size_t)n*sizeof(double)));
    CHECK(hipMemset(flag, 0, 4096*sizeof(int))); CHECK(hipMemset(fail, 0, sizeof(int)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    for (int G : {1, 8, 32, 128}) {
        std::vector<float> tf, tn, tl;
        for (int r = 0; r < reps; r++) {
            CHECK(hipEventRecord(e0, st)); hipLaunchKernelGGL(k_chain_flag, dim3(G), dim3(256), 0, st, data, flag, r + 1, n, fail); CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); tf.push_back(ms*1e3f);
        }
        double last = 0; CHECK(hipMemcpy(&last, data + (size_t)(G - 1)*n, sizeof(double), hipMemcpyDeviceToHost));
        for (int r = 0; r < reps; r++) {
            hipLaunchKernelGGL(k_fill_nan, dim3((G*n + 255)/256), dim3(256), 0, st, data, (size_t)G*n);
            CHECK(hipEventRecord(e0, st)); hipLaunchKernelGGL(k_chain_nan, dim3(G), dim3(256), 0, st, data, n, fail); CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); tn.push_back(ms*1e3f);
        }
        double last2 = 0; CHECK(hipMemcpy(&last2, data + (size_t)(G - 1)*n, sizeof(double), hipMemcpyDeviceToHost));
        for (int r = 0; r < reps; r++) {
            CHECK(hipEventRecord(e0, st)); for (int b = 0; b < G; b++) hipLaunchKernelGGL(k_link, dim3(1), dim3(256), 0, st, data, b, n, r); CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); tl.push_back(ms*1e3f);
        }
        std::sort(tf.begin(), tf.end()); std::sort(tn.begin(), tn.end()); std::sort(tl.begin(), tl.end());
        int hf = 0; CHECK(hipMemcpy(&hf, fail, sizeof(int), hipMemcpyDeviceToHost));
        printf("G = %4d links: one launch, flag + data: median %8.2f us (min %.2f)   value polled: %8.2f us (min %.2f)   %d launches: %8.2f us (min %.2f)   [check %.0f %.0f, fail %d]\n",
               G, tf[reps/2], tf[0], tn[reps/2], tn[0], G, tl[reps/2], tl[0], last, last2, hf);
    }
    return 0;
}
