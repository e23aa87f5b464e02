// Dependent-chain latency probes for gfx950 (diagnostic, not part of the product): cycles per op measured with s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define N 512
__global__ void k_lat(double *out, long long *cyc, const int *chase, double seed) {
    __shared__ double lds[4096];
    __shared__ int ichase[1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = seed + i;
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) ichase[i] = (i*17 + 5) & 1023;
    __syncthreads();
    long long t0, t1; double x = seed + lane*1e-9, y = 1.0000001; int p = lane;
    if (wave == 0) {
        // 0: fma chain
        t0 = clock64();
#pragma unroll 16
        for (int i = 0; i < N; i++) x = fma(x, y, 1e-9);
        t1 = clock64(); if (lane == 0) cyc[0] = t1 - t0;
        // 1: mul chain
        t0 = clock64();
#pragma unroll 16
        for (int i = 0; i < N; i++) x = x*y;
        t1 = clock64(); if (lane == 0) cyc[1] = t1 - t0;
        // 2: rcp chain
        t0 = clock64();
#pragma unroll 16
        for (int i = 0; i < N; i++) x = __builtin_amdgcn_rcp(x);
        t1 = clock64(); if (lane == 0) cyc[2] = t1 - t0;
        // 3: independent fma throughput (8 chains)
        double a[8]; for (int k = 0; k < 8; k++) a[k] = x + k;
        t0 = clock64();
#pragma unroll 4
        for (int i = 0; i < N/8; i++) {
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = fma(a[k], y, 1e-9);
        }
        t1 = clock64(); if (lane == 0) cyc[3] = t1 - t0;
        for (int k = 0; k < 8; k++) x += a[k];
        // 4: LDS dependent read chain (b32 index chase)
        t0 = clock64();
#pragma unroll 16
        for (int i = 0; i < N; i++) p = ichase[p];
        t1 = clock64(); if (lane == 0) cyc[4] = t1 - t0;
        // 5: LDS write -> read round trip (same wave)
        t0 = clock64();
        for (int i = 0; i < N/4; i++) { lds[lane + 64*(i & 7)] = x; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); x += lds[(lane ^ 1) + 64*(i & 7)]; }
        t1 = clock64(); if (lane == 0) cyc[5] = (t1 - t0)*4;
        // 6: readlane -> valu chain
        t0 = clock64();
#pragma unroll 16
        for (int i = 0; i < N; i++) { int lo = __builtin_amdgcn_readlane(__double2loint(x), i & 63); x = x + (double)lo*1e-30; }
        t1 = clock64(); if (lane == 0) cyc[6] = t1 - t0;
        // 7: mfma f64 16x16x4 dependent chain
        typedef double v4d __attribute__((ext_vector_type(4)));
        v4d c = {x, x, x, x};
        t0 = clock64();
#pragma unroll 16
        for (int i = 0; i < N; i++) c = __builtin_amdgcn_mfma_f64_16x16x4f64(y, 1e-9, c, 0, 0, 0);
        t1 = clock64(); if (lane == 0) cyc[7] = t1 - t0;
        x += c[0] + c[1] + c[2] + c[3];
        // 8: global dependent chase (L2 resident after first touches)
        int g = lane;
        t0 = clock64();
        for (int i = 0; i < N/4; i++) g = chase[g];
        t1 = clock64(); if (lane == 0) cyc[8] = (t1 - t0)*4;
        x += g;
        // 9: uniform 16 x ds_read_b64 batch then use (issue + latency of a batch)
        t0 = clock64();
        for (int i = 0; i < N/16; i++) { double s = 0; 
#pragma unroll
            for (int k = 0; k < 16; k++) s += lds[(i*16 + k*37) & 4095]; x += s; }
        t1 = clock64(); if (lane == 0) cyc[9] = t1 - t0;
        // 10: f32 rcp chain, 11: f64 sqrt chain
        float xf = (float)x;
        t0 = clock64();
#pragma unroll 16
        for (int i = 0; i < N; i++) xf = __builtin_amdgcn_rcpf(xf);
        t1 = clock64(); if (lane == 0) cyc[10] = t1 - t0;
        x += xf;
        t0 = clock64();
#pragma unroll 16
        for (int i = 0; i < N; i++) x = __builtin_amdgcn_rsq(x);
        t1 = clock64(); if (lane == 0) cyc[11] = t1 - t0;
    }
    // 12: workgroup barriers, all waves
    __syncthreads();
    t0 = clock64();
    for (int i = 0; i < N/4; i++) __syncthreads();
    t1 = clock64(); if (threadIdx.x == 0) cyc[12] = (t1 - t0)*4;
    out[threadIdx.x] = x + p;
}
int main() {
    double *out; long long *cyc; int *chase;
    hipMalloc(&out, 8*1024); hipMalloc(&cyc, 8*64); hipMalloc(&chase, 4*65536);
    std::vector<int> h(65536); for (int i = 0; i < 65536; i++) h[i] = (i*4099 + 64*7) & 65535;
    hipMemcpy(chase, h.data(), 4*65536, hipMemcpyHostToDevice);
    const char *nm[] = {"fma_f64 dep", "mul_f64 dep", "rcp_f64 dep", "fma_f64 8 indep chains (per op)", "LDS b32 chase", "LDS write->fence->read", "readlane->valu", "mfma_f64_16x16x4 dep", "global chase (L2)", "16 x ds_read_b64 + sum (per read)", "rcp_f32 dep", "rsq_f64 dep", "barrier"};
    for (int threads : {64, 768}) {
        for (int rep = 0; rep < 2; rep++) { hipMemset(cyc, 0, 8*64); hipLaunchKernelGGL(k_lat, dim3(1), dim3(threads), 0, 0, out, cyc, chase, 1.25); hipDeviceSynchronize(); }
        long long c[64]; hipMemcpy(c, cyc, 8*64, hipMemcpyDeviceToHost);
        printf("---- %d threads (cycles per op, s_memtime ticks)\n", threads);
        for (int i = 0; i < 13; i++) printf("%-36s %8.1f\n", nm[i], (double)c[i]/N);
    }
    return 0;
}
