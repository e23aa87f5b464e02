// Reduced camera system of a map with LONG-RANGE coupling: S = M + E, M the band of far_B pose blocks in keyframe order (what the band
// solvers of tsba_band.h / tsba_bandp.h / tsba_bandcre.h factor), E the scattered 6x6 blocks between keyframes further apart (points seen
// again much later, several loop closures -- HostPlan::far_*, tsba_plan.h).  The reference hands any sparsity pattern to Ceres' sparse
// Cholesky (optimizer.cc:1727-1765,1833-1840; GlobalBA runs after every loop closure, loopClosing.cc:587-591); a dense fallback here
// would be (6 n_kf)^2 doubles -- 7.2 GB and ~9 TFLOP per LM trial at 5000 keyframes.  Instead:
//
//     S x = -g   by conjugate gradients preconditioned with M:   z = M^-1 r  is one run of the band solver on the right-hand side r.
//
// The long-range landmarks are a small share of a keyframe's information, so M^-1 S has its spectrum near 1 and a handful of iterations
// reach 1e-10 (relative, in the M^-1 norm); a loop closure adds as many outlying eigenvalues as it couples pose unknowns.  Everything is
// deterministic: partial sums per workgroup, summed in a fixed order by every consumer (no atomics); every workgroup derives the same
// alpha / beta / convergence decision from the same numbers.
//
// Per LM trial (launch_step):  band solve (z_0 = M^-1 (-g))  ->  k_pcg_begin  ->  { k_pcg_matvec  k_pcg_update  band solve  k_pcg_dot } x its
// ->  k_pcg_finish (dp = x).  Vectors live in the compressed row space of S (6 x free poses).  The host stays at most two iterations
// ahead of the device (pinned progress word), so a converged solve wastes two iterations of empty launches.
#pragma once

#define PCG_T 256                           // matvec workgroup: 4 waves, a wave takes PCG_PPW poses
#define PCG_PPW 8
#define PCG_ET 192                          // element-wise kernels: 32 poses x 6 rows per workgroup (the same number of workgroups, so one partial array serves all)

struct PcgState { double rz, rz0; int it, pad; };

__device__ __forceinline__ double pcg_sum_parts(const double *part, int nb, int lane) {     // every lane gets the sum; fixed order
    double s = 0.0;
    for (int k = lane; k < nb; k += 64) s += part[k];
    return wave_sum1(s);
}
__device__ __forceinline__ void pcg_publish(const Work &W, unsigned int seq, int it, int done) {
    if (W.hprog) { W.hprog[1] = ((unsigned long long)seq << 32) | ((unsigned long long)(unsigned int)it << 1) | (done ? 1u : 0u); __threadfence_system(); }
}
// workgroup partial: every thread's v, summed wave by wave in a fixed order, to out[blockIdx.x]
template <int NT>
__device__ __forceinline__ void pcg_block_partial(double v, double *out, double *lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    v = wave_sum1(v);
    if (lane == 0) lds[wave] = v;
    __syncthreads();
    if (tid == 0) { double s = 0.0; for (int k = 0; k < NT/64; k++) s += lds[k]; out[blockIdx.x] = s; }
}

// r_0 = -g, z_0 = M^-1 r_0 (the band solve that just ran), x = 0, p = 0 (beta_0 = 0), partial r.z
__global__ __launch_bounds__(PCG_ET) void k_pcg_begin(Work W) {
    __shared__ double lds[4];
    const LmState *st = W.st;
    if (st->done || st->step_fail) return;
    const int tid = threadIdx.x, a = blockIdx.x*32 + tid/6, k = tid % 6;
    double rzp = 0.0;
    if (a < W.n_kf) { const int ia = W.fidx[a];
        if (ia >= 0) { const int i = 6*ia + k; const double gv = W.g[i], r = -gv, z = -W.Sy[i];
            W.pc_g0[i] = gv; W.pc_r[i] = r; W.pc_x[i] = 0.0; W.pc_p[0][i] = 0.0; W.pc_p[1][i] = 0.0; rzp = r*z; } }
    pcg_block_partial<PCG_ET>(rzp, W.pc_part, lds);
}

// launch `it`: beta from r.z, p = z + beta p, q = S p = (band + long-range blocks) p, partial p.q.  Also where convergence is noticed.
// zp, zs: where the last preconditioner application left z = zs * zp[] (the factorisation's own solve: -W.Sy; the solve phase: its X)
__global__ __launch_bounds__(PCG_T) void k_pcg_matvec(Work W, LevelDev L, int it, unsigned int seq, int B, double tol2, int nbp, const double *zp, double zs) {
    __shared__ double lds[4];
    LmState *st = W.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (st->done || st->step_fail || st->lin_done) { if (blockIdx.x == 0 && tid == 0) pcg_publish(W, seq, it, 1); return; }
    const double rz = pcg_sum_parts(W.pc_part, nbp, lane);
    PcgState *so = W.pcs + ((it + 1) & 1), *sn = W.pcs + (it & 1);
    const double rz0 = it == 0 ? rz : so->rz0, rz_old = it == 0 ? 1.0 : so->rz;
    if (!(rz == rz)) { if (blockIdx.x == 0 && tid == 0) { st->step_fail = 1; pcg_publish(W, seq, it, 1); } return; }
    if (!(rz > tol2*rz0)) {                                  // converged (every workgroup takes the same decision from the same partials)
        if (blockIdx.x == 0 && tid == 0) { st->lin_done = 1; W.pc_stat[0] += it; W.pc_stat[1] += 1; if (it > W.pc_stat[2]) W.pc_stat[2] = it; pcg_publish(W, seq, it, 1); }
        return; }
    const double beta = it == 0 ? 0.0 : rz/rz_old;
    if (blockIdx.x == 0 && tid == 0) { sn->rz = rz; sn->rz0 = rz0; sn->it = it; pcg_publish(W, seq, it, 0); }
    const double *po = W.pc_p[(it + 1) & 1]; double *pn = W.pc_p[it & 1];
    auto pnew = [&](int i) { return fma(beta, po[i], zs*zp[i]); };
    const int nfree = W.nfree[0]; const size_t ldS = (size_t)W.ldS;
    double pq = 0.0;
    for (int u = 0; u < PCG_PPW; u++) {
        const int a = (blockIdx.x*4 + wave)*PCG_PPW + u;
        if (a >= W.n_kf) break;
        const int ia = W.fidx[a];
        if (ia < 0) continue;
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        // lower part of the band incl. the (square) diagonal block: rows 6 ia .. 6 ia + 5, columns from pose block ia - B on
        const int cl = 6*max(ia - B, 0), ncol = 6*ia + 6 - cl;
        for (int cb = 0; cb < ncol; cb += 64) {
            const bool ok = cb + lane < ncol; const int c = ok ? cl + cb + lane : cl;
            const double pv = ok ? pnew(c) : 0.0;
#pragma unroll
            for (int r = 0; r < 6; r++) acc[r] += (c <= 6*ia + r ? W.S[(size_t)(6*ia + r)*ldS + c] : W.S[(size_t)c*ldS + 6*ia + r])*pv;     // (the diagonal block by its lower triangle: what a sharded run exchanges)
        }
        // the transposed part: rows of the pose blocks ia + 1 .. ia + B, columns 6 ia .. 6 ia + 5
        const int r0 = 6*(ia + 1), nrow = 6*min(ia + B, nfree - 1) + 6 - r0;
        for (int rb = 0; rb < nrow; rb += 64) {
            const bool ok = rb + lane < nrow; const int j = ok ? r0 + rb + lane : r0;
            const double pv = ok ? pnew(j) : 0.0;
            const double *row = W.S + (size_t)j*ldS + 6*ia;
#pragma unroll
            for (int k = 0; k < 6; k++) acc[k] += row[k]*pv;
        }
        // long-range blocks of this keyframe: lane l < 36 holds entry (l / 6, l % 6) of a block (rows: far_a, columns: far_b)
        double f0 = 0.0, f1 = 0.0;
        const int e0 = L.far_off[a], e1 = L.far_off[a + 1], fr = lane < 36 ? lane/6 : 0, fc = lane < 36 ? lane % 6 : 0;
        for (int e = e0; e < e1; e++) {
            const int ent = L.far_ent[e], fid = ent >> 1, side = ent & 1;
            const int io = W.fidx[side ? L.far_a[fid] : L.far_b[fid]];
            if (io < 0) continue;
            if (lane < 36) { const double v = W.Sfar[(size_t)fid*36 + lane];
                if (!side) f0 += v*pnew(6*io + fc); else f1 += v*pnew(6*io + fr); }
        }
        double qk = 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) { const double s = wave_sum1(acc[k]); if (lane == k) qk = s; }
        double g0 = 0.0, g1 = 0.0;
#pragma unroll
        for (int j = 0; j < 6; j++) { g0 += __shfl(f0, 6*min(lane, 5) + j, 64); g1 += __shfl(f1, 6*j + min(lane, 5), 64); }
        if (lane < 6) { qk += g0 + g1; const int i = 6*ia + lane; const double pv = pnew(i); pn[i] = pv; W.pc_q[i] = qk; pq += pv*qk; }
    }
    pcg_block_partial<PCG_T>(pq, W.pc_part + nbp, lds);
}

// alpha = r.z / p.q; x += alpha p; r -= alpha q; the next preconditioner application's right-hand side rhs = rs * r (the factorisation
// path solves M y = -g: rs = -1 into W.g; the solve phase takes r itself)
__global__ __launch_bounds__(PCG_ET) void k_pcg_update(Work W, int it, int nbp, double *rhs, double rs) {
    LmState *st = W.st;
    if (st->done || st->step_fail || st->lin_done) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const double pq = pcg_sum_parts(W.pc_part + nbp, nbp, lane);
    if (!(pq > 0.0)) { if (blockIdx.x == 0 && tid == 0) st->step_fail = 1; return; }       // S is positive definite (damped): a breakdown is a failed step
    const double alpha = W.pcs[it & 1].rz/pq;
    const int a = blockIdx.x*32 + tid/6, k = tid % 6;
    if (a >= W.n_kf) return;
    const int ia = W.fidx[a]; if (ia < 0) return;
    const int i = 6*ia + k;
    W.pc_x[i] = fma(alpha, W.pc_p[it & 1][i], W.pc_x[i]);
    const double r = fma(-alpha, W.pc_q[i], W.pc_r[i]);
    W.pc_r[i] = r; rhs[i] = rs*r;
}

// partial r.z after the preconditioner application
__global__ __launch_bounds__(PCG_ET) void k_pcg_dot(Work W, const double *zp, double zs) {
    __shared__ double lds[4];
    const LmState *st = W.st;
    if (st->done || st->step_fail || st->lin_done) return;
    const int tid = threadIdx.x, a = blockIdx.x*32 + tid/6, k = tid % 6;
    double rzp = 0.0;
    if (a < W.n_kf) { const int ia = W.fidx[a]; if (ia >= 0) rzp = W.pc_r[6*ia + k]*(zs*zp[6*ia + k]); }
    pcg_block_partial<PCG_ET>(rzp, W.pc_part, lds);
}

// dp = x by keyframe (0 for constant poses / failed steps), g restored, the trial's flag cleared
__global__ __launch_bounds__(PCG_ET) void k_pcg_finish(Work W, int its_enqueued) {
    LmState *st = W.st;
    if (st->done) return;
    const int tid = threadIdx.x, a = blockIdx.x*32 + tid/6, k = tid % 6;
    const int fail = st->step_fail, conv = st->lin_done;
    if (a < W.n_kf) { const int ia = W.fidx[a];
        W.dp[6*a + k] = (ia >= 0 && !fail) ? W.pc_x[6*ia + k] : 0.0;
        if (ia >= 0 && !fail) W.g[6*ia + k] = W.pc_g0[6*ia + k]; }
    __syncthreads();                                         // (every thread of workgroup 0 has read the flag)
    if (blockIdx.x == 0 && tid == 0) {
        if (!conv && !fail) { W.pc_stat[0] += its_enqueued; W.pc_stat[1] += 1; W.pc_stat[3] += 1; if (its_enqueued > W.pc_stat[2]) W.pc_stat[2] = its_enqueued; }
        st->lin_done = 0; }
}
