// "All workgroups of this launch have stored their results" -- what it costs to find that out inside the launch, on gfx950 (diagnostic, not part of the
// product).  G workgroups of 128 threads store 8 doubles per thread each (what a workgroup of k_linearize leaves), then signal; the last F of them (by
// index) wait until all G have signalled and read one record of every other workgroup.
//   mode 0  no signal at all (the stores alone: the floor)
//   mode 1  __threadfence() + atomicAdd on ONE counter; the waiters poll the counter
//   mode 2  as 1 with C counters (workgroup b -> counter b % C, 256 B apart); the waiters poll all C
//   mode 3  __threadfence() + a flag word per workgroup (plain store of the epoch); the waiters poll the G flags
//   mode 4  no fence: the results go out as device-coherent (sc1) stores, s_waitcnt, then the flag as a coherent store; the waiters poll the flags with
//           coherent loads and read the records with coherent loads
//   mode 5  as 1 without the fence (how much of 1 is the fence)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ __forceinline__ unsigned long long ld_co(const unsigned long long *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__global__ __launch_bounds__(128) void k_sig(double *rec, unsigned long long *cnt, unsigned long long *flag, double *out, int mode, int C, int F, unsigned long long epoch, unsigned long long base, int *fail) {
    const int b = blockIdx.x, tid = threadIdx.x, G = gridDim.x;
    double *mine = rec + ((size_t)b*128 + tid)*8;
    if (mode == 4) { for (int k = 0; k < 8; k++) __hip_atomic_store(mine + k, (double)epoch + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_amdgcn_s_waitcnt(0); }
    else for (int k = 0; k < 8; k++) mine[k] = (double)epoch + k;
    __syncthreads();
    if (tid == 0) {
        if (mode == 1 || mode == 2 || mode == 3) __threadfence();
        if (mode == 1 || mode == 5) atomicAdd(cnt, 1ull);
        else if (mode == 2) atomicAdd(cnt + 32*(b % C), 1ull);
        else if (mode == 3) flag[b] = epoch;
        else if (mode == 4) __hip_atomic_store(flag + b, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (b < G - F || mode == 0) return;
    __shared__ int ok; if (tid == 0) ok = 1;
    __syncthreads();
    if (mode == 1 || mode == 5) { if (tid == 0) { int spins = 0; while (ld_co(cnt) - base < (unsigned long long)G) { if (++spins > (1 << 18)) { ok = 0; *fail = 1; break; } __builtin_amdgcn_s_sleep(2); } } }
    else if (mode == 2) { if (tid < C) { const unsigned long long want = (unsigned long long)((G - tid + C - 1)/C); int spins = 0; while (ld_co(cnt + 32*tid) - base*0 < want*epoch) { if (++spins > (1 << 18)) { ok = 0; *fail = 1; break; } __builtin_amdgcn_s_sleep(2); } } }
    else { for (int k = tid; k < G; k += 128) { int spins = 0; while (ld_co(flag + k) != epoch) { if (++spins > (1 << 18)) { ok = 0; *fail = 1; break; } __builtin_amdgcn_s_sleep(2); } } }
    __syncthreads();
    if (mode != 4) __threadfence();
    double s = 0.0;
    for (int k = tid; k < G; k += 128) { const double *p = rec + ((size_t)k*128 + (tid & 127))*8;
        s += mode == 4 ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p; }
    out[(size_t)b*128 + tid] = s;
    if (tid == 0 && s != (double)epoch*((G - tid + 127)/128)) { /* stale read */ if (ok) atomicAdd(fail + 1, 1); }
}
int main() {
    const int reps = 60;
    double *rec, *out; unsigned long long *cnt, *flag; int *fail;
    CHECK(hipMalloc(&rec, 2048*128*8*sizeof(double))); CHECK(hipMalloc(&out, 2048*128*sizeof(double))); CHECK(hipMalloc(&cnt, 64*32*sizeof(unsigned long long)));
    CHECK(hipMalloc(&flag, 4096*sizeof(unsigned long long))); CHECK(hipMalloc(&fail, 2*sizeof(int)));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipStream_t st; CHECK(hipStreamCreate(&st));
    for (int G : {256, 736, 1472}) for (int F : {48}) {
        struct V { int mode, C; const char *name; };
        const V vs[] = { {0, 1, "stores only"}, {5, 1, "atomic, one counter, no fence"}, {1, 1, "fence + atomic, one counter"}, {2, 8, "fence + atomic, 8 counters"}, {2, 32, "fence + atomic, 32 counters"},
                         {3, 1, "fence + flag per workgroup"}, {4, 1, "coherent stores + flag, no fence"} };
        for (const V &v : vs) {
            CHECK(hipMemsetAsync(cnt, 0, 64*32*sizeof(unsigned long long), st)); CHECK(hipMemsetAsync(flag, 0, 4096*sizeof(unsigned long long), st)); CHECK(hipMemsetAsync(fail, 0, 2*sizeof(int), st));
            std::vector<float> t;
            for (int r = 0; r < reps; r++) {
                const unsigned long long epoch = (unsigned long long)r + 1, base = (unsigned long long)r*G;
                CHECK(hipEventRecord(e0, st));
                hipLaunchKernelGGL(k_sig, dim3(G), dim3(128), 0, st, rec, cnt, flag, out, v.mode, v.C, F, epoch, base, fail);
                CHECK(hipEventRecord(e1, st)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); t.push_back(ms*1e3f);
            }
            int f[2]; CHECK(hipMemcpy(f, fail, sizeof(f), hipMemcpyDeviceToHost));
            std::sort(t.begin(), t.end());
            printf("G %5d F %3d  %-36s median %7.2f us  min %7.2f%s%s\n", G, F, v.name, t[t.size()/2], t[0], f[0] ? "  (TIME-OUT)" : "", f[1] ? "  (stale reads)" : "");
        }
    }
    return 0;
}
