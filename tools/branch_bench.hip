// Cost of wave-uniform (scalar) branches on gfx950 (diagnostic): taken forward jumps over a block of instructions.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SKIP> __device__ long long probe(int never, double &x) {
    long long t0 = clock64();
    for (int it = 0; it < 64; it++) {
#pragma unroll
        for (int k = 0; k < 16; k++) {
            if (__builtin_expect(never == k + 1000*it, 0)) {          // never true: the body is jumped over (taken branch)
#pragma unroll
                for (int j = 0; j < SKIP; j++) x = fma(x, 1.0000001, (double)j);
                asm volatile("s_nop 0");
            }
            asm volatile("s_nop 0");
        }
    }
    return clock64() - t0;
}
__global__ void k_br(double *out, long long *cyc, int never, int nact) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double x = lane;
    if (wave < nact) {
        long long c;
        c = probe<1>(never, x); if (threadIdx.x == 0) cyc[0] = c;
        c = probe<16>(never, x); if (threadIdx.x == 0) cyc[1] = c;
        c = probe<128>(never, x); if (threadIdx.x == 0) cyc[2] = c;
        // straight-line reference: 1024 s_nop
        long long t0 = clock64();
        for (int it = 0; it < 64; it++) {
#pragma unroll
            for (int k = 0; k < 16; k++) asm volatile("s_nop 0");
        }
        c = clock64() - t0; if (threadIdx.x == 0) cyc[3] = c;
    }
    out[threadIdx.x] = x;
}
int main() {
    double *out; long long *cyc;
    (void)hipMalloc(&out, 8*1024); (void)hipMalloc(&cyc, 8*64);
    for (int nact : {1, 12}) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_br, dim3(1), dim3(768), 0, 0, out, cyc, -5, nact); (void)hipDeviceSynchronize(); }
        long long c[8]; (void)hipMemcpy(c, cyc, 64, hipMemcpyDeviceToHost);
        printf("---- %d waves active: cycles per iteration (compare + branch + s_nop)\n", nact);
        printf("jump over 1 fma      %6.1f\njump over 16 fma     %6.1f\njump over 128 fma    %6.1f\nno branch, s_nop     %6.1f\n", c[0]/1024., c[1]/1024., c[2]/1024., c[3]/1024.);
    }
    return 0;
}
