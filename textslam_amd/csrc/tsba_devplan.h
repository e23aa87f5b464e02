// Device-side part of the plan of a LARGE map: the slot pairs of the point landmarks by S block.  (part of the single translation unit tsba.hip)
//
// Block (a, b), a <= b, of the reduced camera system gathers one slot pair (slot of landmark j at pose a, slot of j at pose b) per landmark that both
// poses see, in landmark order (HostPlan::sb_pt_*; the order fixes the summation order of k_schur_*: results are bit-reproducible).  The host built
// these lists in two passes over the 2 M pairs of a 5000-keyframe map on sixteen threads: 9 ms of a 21 ms plan.  Here a wave per block walks pose
// a's slot list (ascending landmarks: the slots are landmark-major) and looks for pose b among each landmark's slots (at most a dozen): the same
// entries in the same order, no sort and no atomics -- count per block, one scan, the same walk again to place.
#pragma once

template <int PASS>
__global__ __launch_bounds__(64) void k_sb_pairs(int n_sb, const int *__restrict__ sb_a, const int *__restrict__ sb_b, const int *__restrict__ pose_ps_off, const int *__restrict__ pose_ps,
                                                 const int *__restrict__ pose_ps_lm, const int *__restrict__ pls_off, const int *__restrict__ pslot_pose, const int *__restrict__ cl,
                                                 int *__restrict__ off, int *__restrict__ s1, int *__restrict__ s2, int *__restrict__ lm) {
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= n_sb) return;
    const int pa = sb_a[b], pb = sb_b[b];
    const int x0 = pose_ps_off[pa], x1 = pose_ps_off[pa + 1];
    int at = PASS ? off[b] : 0;
    for (int xb = x0; xb < x1; xb += 64) {
        const int x = xb + lane; int sa = -1, sb = -1, j = -1;
        if (x < x1) { sa = pose_ps[x]; j = pose_ps_lm[x];
            if (pa == pb) sb = sa;
            else { const int o = pls_off[j], e = pls_off[j + 1];
                for (int q = o; q < e; q++) if (pslot_pose[q] == pb) sb = q; }
            if (sb >= 0 && cl && cl[sa] != cl[sb]) sb = -1; }         // (different clusters of the landmark: that pair belongs to E, not to the band part)
        const unsigned long long m = __ballot(sb >= 0);
        if (PASS && sb >= 0) { const int w = at + __popcll(m & ((1ull << lane) - 1ull)); s1[w] = sa; s2[w] = sb; lm[w] = j; }
        at += __popcll(m);
    }
    if (!PASS && lane == 0) off[b + 1] = at;                          // counts, shifted by one: the scan below turns them into offsets in place
}
// off[0] = 0, off[k + 1] = sum of the counts 0 .. k: one workgroup, a contiguous chunk per thread
__global__ __launch_bounds__(1024) void k_sb_scan(int n, int *off) {
    __shared__ int part[1024];
    const int tid = threadIdx.x, per = (n + 1023)/1024, k0 = tid*per, k1 = min(k0 + per, n);
    int s = 0;
    for (int k = k0; k < k1; k++) s += off[k + 1];
    part[tid] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { const int v = tid >= d ? part[tid - d] : 0; __syncthreads(); part[tid] += v; __syncthreads(); }
    int run = tid ? part[tid - 1] : 0;
    if (tid == 0) off[0] = 0;
    for (int k = k0; k < k1; k++) { run += off[k + 1]; off[k + 1] = run; }
}
