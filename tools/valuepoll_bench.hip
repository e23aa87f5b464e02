// Round 6 companion of ticket_bench.hip (diagnostic, not part of the product): a phase boundary of an LM trial handed over INSIDE a launch by polling the
// consumed values themselves -- no counter, no flag, no fence.  G producer workgroups of 128 threads do `work` dependent fp64 operations and store one
// 64-byte record per thread (what a workgroup of k_linearize leaves); F consumer workgroups (the FIRST F of the grid: dispatched before any producer, so they
// never wait for a workgroup that cannot start) gather R records per thread from all over the record array and sum them (what a thread of k_mid does for its
// landmark).
//   mode 0  two launches: producers with plain stores, then consumers with plain loads (the production shape: k_linearize, k_mid)
//   mode 1  one launch: records NaN before the launch (an earlier kernel), device-coherent stores, consumers poll all R x 8 values with device-coherent
//           loads until none is NaN
//   mode 2  as 1, but a consumer first sleeps until ONE word per record (the last stored) is there and only then loads the rest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ __forceinline__ double ld_co(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_co(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
#define R 5
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
__device__ __forceinline__ void produce(double *rec, int p, int tid, int work, double epoch, bool coherent) {
    double a = epoch;
    for (int k = 0; k < work; k++) a = fma(a, 1.0000001, 1e-9);          // dependent chain: ~8 cycles each
    double *mine = rec + ((size_t)p*128 + tid)*8;
    if (coherent) { for (int k = 0; k < 8; k++) st_co(mine + k, a + k); }
    else for (int k = 0; k < 8; k++) mine[k] = a + k;
}
template <int MODE>
__device__ __forceinline__ void consume(const double *rec, double *out, int c, int tid, int nrec, int *fail) {
    const int j = c*128 + tid;
    int idx[R];
    for (int r = 0; r < R; r++) idx[r] = hash(j*R + r) % (unsigned)nrec;
    double v[R][8];
    if (MODE == 0) { for (int r = 0; r < R; r++) for (int k = 0; k < 8; k++) v[r][k] = rec[(size_t)idx[r]*8 + k]; }
    else {
        if (MODE == 2) {
            for (int spins = 0; spins < (1 << 16); spins++) {
                bool all = true;
                double t[R];
                for (int r = 0; r < R; r++) t[r] = ld_co(rec + (size_t)idx[r]*8 + 7);
                for (int r = 0; r < R; r++) all = all && t[r] == t[r];
                if (all) break;
                __builtin_amdgcn_s_sleep(8);
            }
        }
        int spins = 0;
        for (;;) {
            for (int r = 0; r < R; r++) for (int k = 0; k < 8; k++) v[r][k] = ld_co(rec + (size_t)idx[r]*8 + k);
            bool all = true;
            for (int r = 0; r < R; r++) for (int k = 0; k < 8; k++) all = all && v[r][k] == v[r][k];
            if (all) break;
            if (++spins > (1 << 16)) { *fail = 1; break; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    double s = 0.0;
    for (int r = 0; r < R; r++) for (int k = 0; k < 8; k++) s += v[r][k];
    out[j] = s;
}
__global__ __launch_bounds__(128) void k_prod(double *rec, int work, double epoch) { produce(rec, blockIdx.x, threadIdx.x, work, epoch, false); }
__global__ __launch_bounds__(128) void k_cons(const double *rec, double *out, int nrec, int *fail) { consume<0>(rec, out, blockIdx.x, threadIdx.x, nrec, fail); }
template <int MODE>
__global__ __launch_bounds__(128) void k_fused(double *rec, double *out, int F, int work, double epoch, int nrec, int *fail) {
    if ((int)blockIdx.x < F) consume<MODE>(rec, out, blockIdx.x, threadIdx.x, nrec, fail);
    else produce(rec, blockIdx.x - F, threadIdx.x, work, epoch, true);
}
__global__ void k_nan(double *rec, size_t n) { for (size_t i = blockIdx.x*(size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x*blockDim.x) rec[i] = __builtin_nan(""); }
int main() {
    const int reps = 60;
    double *rec, *out; int *fail;
    CHECK(hipMalloc(&rec, 2048*128*8*sizeof(double))); CHECK(hipMalloc(&out, 256*128*sizeof(double))); CHECK(hipMalloc(&fail, sizeof(int)));
    CHECK(hipMemset(fail, 0, sizeof(int)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    for (int G : {256, 736}) for (int F : {48}) for (int work : {0, 1500, 3000}) for (int mode : {0, 1, 2}) {
        const int nrec = G*128;
        std::vector<float> t, tp;
        for (int r = 0; r < reps; r++) {
            hipLaunchKernelGGL(k_nan, dim3(256), dim3(256), 0, st, rec, (size_t)nrec*8);
            CHECK(hipEventRecord(e0, st));
            if (mode == 0) { hipLaunchKernelGGL(k_prod, dim3(G), dim3(128), 0, st, rec, work, (double)r); hipLaunchKernelGGL(k_cons, dim3(F), dim3(128), 0, st, rec, out, nrec, fail); }
            else if (mode == 1) hipLaunchKernelGGL(k_fused<1>, dim3(G + F), dim3(128), 0, st, rec, out, F, work, (double)r, nrec, fail);
            else hipLaunchKernelGGL(k_fused<2>, dim3(G + F), dim3(128), 0, st, rec, out, F, work, (double)r, nrec, fail);
            CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms*1e3f);
            if (mode == 0) {                                     // the producers alone
                CHECK(hipEventRecord(e0, st)); hipLaunchKernelGGL(k_prod, dim3(G), dim3(128), 0, st, rec, work, (double)r); CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
                CHECK(hipEventElapsedTime(&ms, e0, e1)); tp.push_back(ms*1e3f);
            }
        }
        int f; CHECK(hipMemcpy(&f, fail, sizeof(f), hipMemcpyDeviceToHost));
        std::sort(t.begin(), t.end()); std::sort(tp.begin(), tp.end());
        const char *names[] = { "two launches (plain stores, plain loads)", "one launch, values polled", "one launch, one word per record polled first" };
        printf("G %5d F %3d work %5d  %-46s median %7.2f us  min %7.2f", G, F, work, names[mode], t[t.size()/2], t[0]);
        if (mode == 0) printf("   (producers alone: %6.2f)", tp[tp.size()/2]);
        printf("%s\n", f ? "  (TIME-OUT)" : "");
    }
    return 0;
}
