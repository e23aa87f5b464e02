// LDS issue-rate probes for gfx950 (diagnostic): cycles per LDS instruction for one wave, 16 independent reads in flight,
// written with inline asm so the compiler cannot merge / spill / serialise them.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 32
#define RD16(INS, W) \
    asm volatile( \
        INS " %0, %16\n" INS " %1, %16 offset:" W "*1\n" INS " %2, %16 offset:" W "*2\n" INS " %3, %16 offset:" W "*3\n" \
        INS " %4, %16 offset:" W "*4\n" INS " %5, %16 offset:" W "*5\n" INS " %6, %16 offset:" W "*6\n" INS " %7, %16 offset:" W "*7\n" \
        INS " %8, %16 offset:" W "*8\n" INS " %9, %16 offset:" W "*9\n" INS " %10, %16 offset:" W "*10\n" INS " %11, %16 offset:" W "*11\n" \
        INS " %12, %16 offset:" W "*12\n" INS " %13, %16 offset:" W "*13\n" INS " %14, %16 offset:" W "*14\n" INS " %15, %16 offset:" W "*15\n" \
        "s_waitcnt lgkmcnt(0)\n" \
        : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]), \
          "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15]) : "v"(addr) : "memory")
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE> __device__ long long probe(unsigned addr, double &sink) {
    long long t0 = clock64();
    for (int it = 0; it < REP; it++) {
        if (MODE == 0) { float r[16]; RD16("ds_read_b32", "512"); sink += r[0] + r[15]; }
        if (MODE == 1) { double r[16]; RD16("ds_read_b64", "512"); sink += r[0] + r[15]; }
        if (MODE == 2) { v4f r[16]; RD16("ds_read_b128", "1024"); sink += r[0].x + r[15].y; }
    }
    return clock64() - t0;
}
__global__ void k_lds(double *out, long long *cyc, int nwaves_active) {
    extern __shared__ double lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i*0.5;
    __syncthreads();
    double sink = 0;
    if (wave < nwaves_active) {
        long long c;
        const unsigned base = wave*64;      // byte offsets inside a 64 KB window
        c = probe<0>(base + 0, sink);        if (lane == 0 && wave == 0) cyc[0] = c;      // b32 uniform
        c = probe<0>(base + 4*lane, sink);   if (lane == 0 && wave == 0) cyc[1] = c;      // b32 per-lane consecutive
        c = probe<1>(base + 0, sink);        if (lane == 0 && wave == 0) cyc[2] = c;      // b64 uniform
        c = probe<1>(base + 8*lane, sink);   if (lane == 0 && wave == 0) cyc[3] = c;      // b64 per-lane consecutive
        c = probe<2>(base + 0, sink);        if (lane == 0 && wave == 0) cyc[4] = c;      // b128 uniform
        c = probe<2>(base + 16*lane, sink);  if (lane == 0 && wave == 0) cyc[5] = c;      // b128 per-lane consecutive
        c = probe<1>(base + 8*33*lane, sink); if (lane == 0 && wave == 0) cyc[6] = c;     // b64 stride 33 doubles
        c = probe<1>(base + 8*(lane & 7), sink); if (lane == 0 && wave == 0) cyc[7] = c;  // b64 8 distinct addresses
    }
    out[threadIdx.x] = sink;
}
int main() {
    double *out; long long *cyc;
    (void)hipMalloc(&out, 8*1024); (void)hipMalloc(&cyc, 8*64);
    const char *nm[] = {"b32 uniform", "b32 per-lane", "b64 uniform", "b64 per-lane", "b128 uniform", "b128 per-lane", "b64 stride 33", "b64 8 addresses"};
    (void)hipFuncSetAttribute((const void *)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 140*1024);
    for (int lds_kb : {64, 140}) for (int nw : {1, 4, 12}) {
        for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_lds, dim3(1), dim3(768), lds_kb*1024, 0, out, cyc, nw); (void)hipDeviceSynchronize(); }
        long long c[64]; (void)hipMemcpy(c, cyc, 8*64, hipMemcpyDeviceToHost);
        printf("---- LDS %d KB, %d waves reading concurrently: cycles per ds_read (16 in flight)\n", lds_kb, nw);
        for (int i = 0; i < 8; i++) printf("%-20s %8.1f\n", nm[i], (double)c[i]/(16*REP));
    }
    return 0;
}
