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

#define PCG_T 256                           // workgroups of the block kernels: 4 waves, a wave takes PCG_PPW poses
#define PCG_PPW 8
#define PCG_ET 192                          // element-wise kernels: 32 poses x 6 rows per workgroup (the same number of workgroups, so one partial array serves all)

// (struct PcgState: tsba_types.h)

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
__global__ __launch_bounds__(PCG_ET) void k_pcg_begin(Work W, const double *zp, double zs) {
    __shared__ double lds[4];
    const LmState *st = W.st;
    if (st->done || st->step_fail) return;
    const int tid = threadIdx.x, a = blockIdx.x*32 + tid/6, k = tid % 6;
    double rzp = 0.0, rrp = 0.0;
    if (a < W.n_kf) { const int ia = W.fidx[a];
        if (ia >= 0) { const int i = 6*ia + k; const double gv = W.g[i], r = -gv, z = zs*zp[i];
            W.pc_g0[i] = gv; W.pc_r[i] = r; W.pc_x[i] = 0.0; W.pc_p[0][i] = 0.0; W.pc_p[1][i] = 0.0; rzp = r*z; rrp = r*r; } }
    pcg_block_partial<PCG_ET>(rzp, W.pc_part, lds);
    __syncthreads();
    pcg_block_partial<PCG_ET>(rrp, W.pc_part + 2*gridDim.x, lds);
}

// per pass, after the gauge: far_rec[e] = (far_ent[e], row of the entry's other keyframe among the free poses | -1)
__global__ __launch_bounds__(256) void k_far_rows(Work W, LevelDev L, int ne) {
    const int e = blockIdx.x*256 + threadIdx.x;
    if (e >= ne) return;
    const int ent = L.far_ent[e], fid = ent >> 1, oth = (ent & 1) ? L.far_a[fid] : L.far_b[fid];
    L.far_rec[e] = make_int2(ent, W.fidx[oth]);
}

// launch `it`: beta from r.z, p = z + beta p, q = S p = (band + long-range blocks) p, partial p.q.  Also where convergence is noticed.
// zp, zs: where the last preconditioner application left z = zs * zp[] (the factorisation's own solve: -W.Sy; the solve phase: its X).
// A wave per keyframe, PCG_MW keyframes per workgroup; the partial p.q of a workgroup goes to pc_part[pq_off + blockIdx.x].  The kernel is a chain
// of memory round trips, not arithmetic (18 MB at 5000 keyframes): everything a keyframe needs is requested in as few rounds as its index chain allows
// (row of the keyframe, its list of long-range blocks | band rows, columns, the vector, the list's entries | the blocks' other keyframes | their rows |
// the blocks and the vector there) -- with two keyframes one after the other on a wave and the blocks of a keyframe one after the other on 36 lanes
// this was 69 us per launch.
#define PCG_MW 16
__global__ __launch_bounds__(64*PCG_MW) void k_pcg_matvec(Work W, LevelDev L, int it, unsigned int seq, int B, double tol2, int rz_off, int nrz, int pq_off, const double *zp, double zs) {
    __shared__ double lds[PCG_MW];
    LmState *st = W.st;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, a = blockIdx.x*PCG_MW + wave; const bool kf = a < W.n_kf;
    // round 1: the flags, this keyframe's row and list, the partial sums of r.z and the state of the previous iteration -- requested together (pinned: the
    // compiler would issue them one decision at a time)
    const int flags = st->done | st->step_fail | st->lin_done;
    int ia = kf ? W.fidx[a] : -1; const int e0 = kf ? L.far_off[a] : 0, e1 = kf ? L.far_off[a + 1] : 0;
    PcgState *so = W.pcs + ((it + 1) & 1), *sn = W.pcs + (it & 1);
    double pp[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { pp[k] = lane + 64*k < nrz ? W.pc_part[rz_off + lane + 64*k] : 0.0; sv_pin(pp[k]); }
    double so_rz0 = it == 0 ? 0.0 : so->rz0, so_rz = it == 0 ? 1.0 : so->rz, so_best = it == 0 ? 0.0 : so->best; int so_since = it == 0 ? 0 : so->since;
    sv_pin(so_rz0); sv_pin(so_rz); sv_pin(so_best); asm volatile("" : "+v"(so_since)); asm volatile("" : "+v"(ia));
    if (flags) { if (blockIdx.x == 0 && tid == 0) pcg_publish(W, seq, it, 1); return; }
    double rzs = (pp[0] + pp[1]) + (pp[2] + pp[3]);
    for (int k = lane + 256; k < nrz; k += 64) rzs += W.pc_part[rz_off + k];       // (more than 256 partial sums: not with today's sizes)
    const double rz = wave_sum1(rzs);
    const double rz0 = it == 0 ? rz : so_rz0, rz_old = so_rz;
    if (!(rz == rz) || rz < 0.0) {                       // NaN, or r.M^-1 r < 0: the preconditioner is numerically indefinite -- a failed linear solve, the LM loop raises the damping
        if (blockIdx.x == 0 && tid == 0) { st->step_fail = 1; W.pc_stat[5] += 1; pcg_publish(W, seq, it, 1); } return; }
    // where M is so ill-conditioned that the tolerance lies below the rounding noise of M^-1 r (weakly damped trials of maps with loop closures:
    // the drift modes) r.z stops falling: 40 iterations without a gain of a tenth (r.z of conjugate gradients is not monotone: plateaus of a dozen
    // iterations occur on the way down) end the solve with what it has -- the LM step test judges it
    const double best_o = it == 0 ? rz : so_best; const int since_o = so_since;
    const bool gain = rz < 0.9*best_o; const double best = gain ? rz : best_o; const int since = gain ? 0 : since_o + 1;
    if (!(rz > tol2*rz0) || since >= 40) {                   // converged (every workgroup takes the same decision from the same partials)
        if (blockIdx.x == 0 && tid == 0) { st->lin_done = 1; W.pc_stat[0] += it; W.pc_stat[1] += 1; if (it > W.pc_stat[2]) W.pc_stat[2] = it;
            if (rz > tol2*rz0) W.pc_stat[4] += 1;          // ended by stagnation at the attainable accuracy, not by the tolerance
            pcg_publish(W, seq, it, 1); }
        return; }
    const double beta = it == 0 ? 0.0 : rz/rz_old;
    if (blockIdx.x == 0 && tid == 0) { sn->rz = rz; sn->rz0 = rz0; sn->best = best; sn->since = since; sn->it = it; pcg_publish(W, seq, it, 0); }
    const double *po = W.pc_p[(it + 1) & 1]; double *pn = W.pc_p[it & 1];
    const int nfree = W.nfree[0]; const size_t ldS = (size_t)W.ldS;
    double qk = 0.0, pvk = 0.0;
    if (ia >= 0) {
        // round 2: the band.  Lower part incl. the (square) diagonal block: rows 6 ia .. 6 ia + 5, columns from pose block ia - B on (lane: column, up to two);
        // transposed part: rows of the pose blocks ia + 1 .. ia + B, columns 6 ia .. 6 ia + 5 (lane: row, up to two)
        const int cl = 6*max(ia - B, 0), ncol = 6*ia + 6 - cl, r0 = 6*(ia + 1), nrow = 6*min(ia + B, nfree - 1) + 6 - r0;
        double sl[2][6], pl_o[2], pl_z[2], pt_o[2], pt_z[2], st_[2][6];
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const int cc = lane + 64*g; const bool ok = cc < ncol; const int c = cl + (ok ? cc : 0);
            pl_o[g] = ok ? po[c] : 0.0; pl_z[g] = ok ? zp[c] : 0.0;
#pragma unroll
            for (int r = 0; r < 6; r++) sl[g][r] = !ok ? 0.0 : (c <= 6*ia + r ? W.S[(size_t)(6*ia + r)*ldS + c] : W.S[(size_t)c*ldS + 6*ia + r]);     // (the diagonal block by its lower triangle: what a sharded run exchanges)
            const bool okr = cc < nrow; const int j = r0 + (okr ? cc : 0);
            pt_o[g] = okr ? po[j] : 0.0; pt_z[g] = okr ? zp[j] : 0.0;
            const double *row = W.S + (size_t)j*ldS + 6*ia;
#pragma unroll
            for (int k = 0; k < 6; k++) st_[g][k] = okr ? row[k] : 0.0;
        }
        // the long-range blocks of this keyframe, ten at a time: lane (slot, r) holds row r of q's contribution of block `slot`
        const int es = lane/6, r6 = lane - 6*es;
        double fsum = 0.0;
        for (int eb = e0; eb < e1; eb += 10) {
            const int e = eb + es; const bool on = es < 10 && e < e1;
            const int2 rec = on ? L.far_rec[e] : make_int2(0, -1);      // (entry, row of the other keyframe: k_far_rows)
            const int ent = rec.x, fid = ent >> 1, side = ent & 1, io = rec.y;
            double f = 0.0;
            if (io >= 0) {                                      // side 0: this keyframe is far_a (rows of the block), side 1: far_b (columns)
                const double *blk = W.Sfar + (size_t)fid*36 + (side ? r6 : 6*r6); const int stp = side ? 6 : 1;
                double bv[6], xo[6], xz[6];
#pragma unroll
                for (int c = 0; c < 6; c++) { bv[c] = blk[c*stp]; xo[c] = po[6*io + c]; xz[c] = zp[6*io + c]; }
#pragma unroll
                for (int c = 0; c < 6; c++) f = fma(bv[c], fma(beta, xo[c], zs*xz[c]), f);
            }
            double gsum = 0.0;                                  // over the ten slots, in slot order
#pragma unroll
            for (int s2 = 0; s2 < 10; s2++) gsum += __shfl(f, 6*s2 + (lane < 6 ? lane : 0), 64);
            fsum += gsum;
        }
        double acc[6];
#pragma unroll
        for (int r = 0; r < 6; r++) acc[r] = 0.0;
#pragma unroll
        for (int g = 0; g < 2; g++) {
            const double pv = fma(beta, pl_o[g], zs*pl_z[g]), pt = fma(beta, pt_o[g], zs*pt_z[g]);
#pragma unroll
            for (int r = 0; r < 6; r++) acc[r] = fma(sl[g][r], pv, acc[r]);
#pragma unroll
            for (int k = 0; k < 6; k++) acc[k] = fma(st_[g][k], pt, acc[k]);
        }
#pragma unroll
        for (int k = 0; k < 6; k++) { const double s = wave_sum1(acc[k]); if (lane == k) qk = s; }
        if (lane < 6) { qk += fsum; const int i = 6*ia + lane; pvk = fma(beta, po[i], zs*zp[i]); pn[i] = pvk; W.pc_q[i] = qk; }
    }
    const double pq = wave_sum1(lane < 6 ? pvk*qk : 0.0);
    if (lane == 0) lds[wave] = pq;
    __syncthreads();
    if (tid == 0) { double s = 0.0; for (int k = 0; k < PCG_MW; k++) s += lds[k]; W.pc_part[pq_off + blockIdx.x] = s; }
}

// alpha = r.z / p.q; x += alpha p; r -= alpha q; the next preconditioner application's right-hand side rhs = rs * r (the factorisation
// path solves M y = -g: rs = -1 into W.g; the solve phase takes r itself)
__global__ __launch_bounds__(PCG_ET) void k_pcg_update(Work W, int it, int nbp, int pq_off, int npq, double *rhs, double rs) {
    __shared__ double lds[4];
    LmState *st = W.st;
    if (st->done || st->step_fail || st->lin_done) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const double pq = pcg_sum_parts(W.pc_part + pq_off, npq, lane);
    if (!(pq > 0.0)) { if (blockIdx.x == 0 && tid == 0) st->step_fail = 1; return; }       // S is positive definite (damped): a breakdown is a failed step
    const double alpha = W.pcs[it & 1].rz/pq;
    const int a = blockIdx.x*32 + tid/6, k = tid % 6;
    double rrp = 0.0;
    if (a < W.n_kf) { const int ia = W.fidx[a];
        if (ia >= 0) { const int i = 6*ia + k;
            W.pc_x[i] = fma(alpha, W.pc_p[it & 1][i], W.pc_x[i]);
            const double r = fma(-alpha, W.pc_q[i], W.pc_r[i]);
            W.pc_r[i] = r; rhs[i] = rs*r; rrp = r*r; } }
    pcg_block_partial<PCG_ET>(rrp, W.pc_part + 2*nbp, lds);
}
// |r|^2 against |b|^2 right after the update: with an (almost) exact preconditioner -- the low-rank correction of tsba_wb.h -- the first step
// already ends the solve, and the test on r.z would only notice after one more application of M^-1.  One wave.
__global__ __launch_bounds__(64) void k_pcg_rcheck(Work W, int it, int nbp, double tolr2) {
    LmState *st = W.st;
    if (st->done || st->step_fail || st->lin_done) return;
    const double rr = pcg_sum_parts(W.pc_part + 2*nbp, nbp, threadIdx.x);
    if (threadIdx.x != 0) return;
    if (it < 0) { W.pc_part[3*nbp] = rr; return; }             // |b|^2
    if (rr <= tolr2*W.pc_part[3*nbp]) { st->lin_done = 1; W.pc_stat[0] += it + 1; W.pc_stat[1] += 1; if (it + 1 > W.pc_stat[2]) W.pc_stat[2] = it + 1; }
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

// dp = x by keyframe (0 for constant poses / failed steps), g restored.  A solve that ran into the iteration cap without converging is a FAILED
// linear solve: the reference's sparse Cholesky either solves the system or the step is invalid (Ceres: LINEAR_SOLVER_FAILURE -> the trust region
// shrinks; optimizer.cc:1833-1845) -- the trial is flagged step_fail, k_decide halves the radius, and the count goes to tsba_report.pcg_unconverged.
// (lin_done is cleared by k_decide: every workgroup of this kernel reads it.)
__global__ __launch_bounds__(PCG_ET) void k_pcg_finish(Work W, int its_enqueued) {
    LmState *st = W.st;
    if (st->done) return;
    const int tid = threadIdx.x, a = blockIdx.x*32 + tid/6, k = tid % 6;
    const int conv = st->lin_done, fail0 = st->step_fail, fail = fail0 | !conv;
    if (a < W.n_kf) { const int ia = W.fidx[a];
        W.dp[6*a + k] = (ia >= 0 && !fail) ? W.pc_x[6*ia + k] : 0.0;
        if (ia >= 0 && !fail0) W.g[6*ia + k] = W.pc_g0[6*ia + k]; }
    if (blockIdx.x == 0 && tid == 0 && !conv && !fail0) {
        W.pc_stat[0] += its_enqueued; W.pc_stat[1] += 1; W.pc_stat[3] += 1; if (its_enqueued > W.pc_stat[2]) W.pc_stat[2] = its_enqueued;
        st->step_fail = 1; }
}

// =====================================================================================================================================
// Enlarged conjugate gradients (Grigori, Moufawad, Nataf 2016): the residual is split into ECG_T vectors by pose (column c holds the rows of the
// poses a with a mod ECG_T == c), the search space grows by ECG_T directions per iteration -- one application of M^-1 to ECG_T columns costs
// what one column costs (tsba_bandms.h: a lane owns a column), and the outlying eigenvalues that a loop closure or the long-range points add
// to M^-1 S are captured ECG_T at a time: 38 -> 6 iterations on a map with two closures, 25 -> 13 with 1 % long-range points (1500 keyframes).
//   P~ (n x T) search directions (not orthonormalised: every use goes through C = P~^T S P~), Q~ = S P~, R residual block, Z = M^-1 R, x
//   per iteration:  Q~ = S P~ | C = P~^T Q~, D = P~^T R | Y1 = C^-1 D | R -= Q~ Y1, x += P~ (Y1 1) | Z = M^-1 R | E = Q~^T Z, r.z |
//                   Y2 = C^-1 E, convergence | P~ = Z - P~ Y2
// C^-1 by a Cholesky factorisation with a pivot threshold (a direction that has become dependent is dropped for the iteration).  All sums in
// fixed order (per-chunk partials, summed by one workgroup): deterministic.
#define ECG_T 32
#define ECG_CH 32                           // poses per chunk of the Gram kernels (192 rows)

struct EcgBuf { double *P, *Q, *part, *Cm, *Lm, *Y, *y1, *scal; int nchunk; };      // part: [nchunk][2][T*T + 1]; scal: rz0, rz, flags

// R = T(-g) (column = pose mod T), x = 0
__global__ __launch_bounds__(PCG_ET) void k_ecg_begin(Work W, MsBuf M) {
    const LmState *st = W.st;
    if (st->done || st->step_fail) return;
    const int tid = threadIdx.x, a = blockIdx.x*32 + tid/6, k = tid % 6;
    if (a >= W.n_kf) return;
    const int ia = W.fidx[a]; if (ia < 0) return;
    const int i = 6*ia + k; const double gv = W.g[i];
    W.pc_g0[i] = gv; W.pc_x[i] = 0.0;
    for (int c = 0; c < ECG_T; c++) M.R[(size_t)i*ECG_T + c] = (ia % ECG_T) == c ? -gv : 0.0;
}

// Q~ = S P~ for ECG_T columns: one wave per pose, a lane per column; the pose's blocks of S staged in LDS by the wave (broadcast reads)
__global__ __launch_bounds__(256) void k_ecg_matvec(Work W, LevelDev L, int B, EcgBuf E) {
    __shared__ double blk[4][36];
    const LmState *st = W.st;
    if (st->done || st->step_fail || st->lin_done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & (ECG_T - 1), half = lane >> 5;      // two halves of a wave: two neighbour blocks at a time
    const int nfree = W.nfree[0]; const size_t ldS = (size_t)W.ldS;
    for (int u = 0; u < PCG_PPW; u++) {
        const int a = (blockIdx.x*4 + wave)*PCG_PPW + u;
        if (a >= W.n_kf) break;
        const int ia = W.fidx[a];
        if (ia < 0) continue;
        double acc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        // band: neighbour blocks ic = ia - B .. ia + B; lane (half, column): half 0 takes the even offsets, half 1 the odd ones
        const int lo = max(ia - B, 0), hi = min(ia + B, nfree - 1);
        for (int ic = lo + half; ic <= hi; ic += 2) {
            double s[36];
            if (ic <= ia) {
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int cc = 0; cc < 6; cc++) { const int col = 6*ic + cc, row = 6*ia + r;
                        s[6*r + cc] = (ic < ia || col <= row) ? W.S[(size_t)row*ldS + col] : W.S[(size_t)col*ldS + row]; }
            } else {
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int cc = 0; cc < 6; cc++) s[6*r + cc] = W.S[(size_t)(6*ic + cc)*ldS + 6*ia + r];
            }
            double p[6];
#pragma unroll
            for (int cc = 0; cc < 6; cc++) p[cc] = E.P[(size_t)(6*ic + cc)*ECG_T + c];
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int cc = 0; cc < 6; cc++) acc[r] = fma(s[6*r + cc], p[cc], acc[r]);
        }
        // blocks of E of this keyframe
        const int e0 = L.far_off[a], e1 = L.far_off[a + 1];
        for (int e = e0 + half; e < e1; e += 2) {
            const int ent = L.far_ent[e], fid = ent >> 1, side = ent & 1;
            const int io = W.fidx[side ? L.far_a[fid] : L.far_b[fid]];
            if (io < 0) continue;
            const double *sb = W.Sfar + (size_t)fid*36;
            double p[6];
#pragma unroll
            for (int cc = 0; cc < 6; cc++) p[cc] = E.P[(size_t)(6*io + cc)*ECG_T + c];
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int cc = 0; cc < 6; cc++) acc[r] = fma(side ? sb[6*cc + r] : sb[6*r + cc], p[cc], acc[r]);
        }
#pragma unroll
        for (int r = 0; r < 6; r++) { acc[r] += __shfl_xor(acc[r], 32, 64); if (half == 0) E.Q[(size_t)(6*ia + r)*ECG_T + c] = acc[r]; }
    }
    (void)blk;
}

// partial Gram matrices of a chunk of ECG_CH poses: out1 = A^T B1, out2 = A^T B2 (B2 may be null), and the partial of (R 1).(Z 1)
__global__ __launch_bounds__(256) void k_ecg_gram(Work W, const double *A, const double *B1, const double *B2, const double *Rr, const double *Zz, EcgBuf E) {
    __shared__ double sa[32*ECG_T], sb1[32*ECG_T], sb2[32*ECG_T], red[4];
    const LmState *st = W.st;
    if (st->done || st->step_fail || st->lin_done) return;
    const int tid = threadIdx.x, i = tid >> 3, j0 = (tid & 7)*4;                  // thread: row i of the t x t result, columns j0 .. j0 + 3
    double c1[4] = {0, 0, 0, 0}, c2[4] = {0, 0, 0, 0}, rz = 0.0;
    const int a0 = blockIdx.x*ECG_CH;
    // the chunk's rows: the free poses among a0 .. a0 + ECG_CH - 1, 32 rows staged at a time
    for (int sub = 0; sub < 6; sub++) {                                           // 32 poses x 6 rows = 6 slabs of 32 rows (row k of every pose of the chunk)
        __syncthreads();
        for (int e = tid; e < 32*ECG_T; e += 256) { const int p = e/ECG_T, col = e - p*ECG_T, a = a0 + p;
            const int ia = a < W.n_kf ? W.fidx[a] : -1; const size_t idx = ia >= 0 ? (size_t)(6*ia + sub)*ECG_T + col : 0;
            sa[e] = ia >= 0 ? A[idx] : 0.0; sb1[e] = ia >= 0 ? B1[idx] : 0.0; sb2[e] = (ia >= 0 && B2) ? B2[idx] : 0.0; }
        if (Rr && tid < 32) { const int a = a0 + tid, ia = a < W.n_kf ? W.fidx[a] : -1;
            if (ia >= 0) { double sr = 0.0, sz = 0.0; for (int col = 0; col < ECG_T; col++) { sr += Rr[(size_t)(6*ia + sub)*ECG_T + col]; sz += Zz[(size_t)(6*ia + sub)*ECG_T + col]; } rz += sr*sz; } }
        __syncthreads();
        for (int p = 0; p < 32; p++) { const double av = sa[p*ECG_T + i];
#pragma unroll
            for (int q = 0; q < 4; q++) { c1[q] = fma(av, sb1[p*ECG_T + j0 + q], c1[q]); c2[q] = fma(av, sb2[p*ECG_T + j0 + q], c2[q]); } }
    }
    double *o = E.part + (size_t)blockIdx.x*2*(ECG_T*ECG_T + 1);
#pragma unroll
    for (int q = 0; q < 4; q++) { o[i*ECG_T + j0 + q] = c1[q]; o[ECG_T*ECG_T + 1 + i*ECG_T + j0 + q] = c2[q]; }
    rz = wave_sum1(tid < 64 ? rz : 0.0);
    if (tid == 0) o[ECG_T*ECG_T] = rz;
    (void)red;
}

// the t x t algebra, one workgroup of 1024 threads (thread = entry).  mode 1: C, D from the partials, Cholesky of C with a pivot threshold,
// Y = C^-1 D, y1 = Y 1.  mode 2: E from the partials, r.z, convergence, Y = C^-1 E.  mode 0: only r.z of the start (rz0).
__global__ __launch_bounds__(1024) void k_ecg_small(Work W, EcgBuf E, int mode, int it, unsigned int seq, double tol2) {
    __shared__ double Cm[ECG_T*ECG_T], Dm[ECG_T*ECG_T], dinv[ECG_T]; __shared__ int dead[ECG_T];
    LmState *st = W.st;
    const int tid = threadIdx.x, i = tid/ECG_T, j = tid % ECG_T;
    if (st->done || st->step_fail || st->lin_done) { if (tid == 0 && mode == 2) pcg_publish(W, seq, it, 1); return; }
    const size_t ps = 2*(ECG_T*ECG_T + 1);
    double c = 0.0, d = 0.0;
    for (int k = 0; k < E.nchunk; k++) { c += E.part[(size_t)k*ps + tid]; d += E.part[(size_t)k*ps + ECG_T*ECG_T + 1 + tid]; }
    if (mode != 1) {                                            // r.z in fixed order
        double rz = 0.0; if (tid == 0) { for (int k = 0; k < E.nchunk; k++) rz += E.part[(size_t)k*ps + ECG_T*ECG_T]; }
        __shared__ double srz; if (tid == 0) srz = rz; __syncthreads(); rz = srz;
        if (mode == 0) { if (tid == 0) { E.scal[0] = rz; E.scal[1] = rz; E.scal[2] = rz; E.scal[3] = 0.0; } return; }
        const double rz0 = E.scal[0], best_o = E.scal[2]; const int since_o = (int)E.scal[3];
        const bool gain = rz < 0.9*best_o; const int since = gain ? 0 : since_o + 1;
        if (!(rz == rz)) { if (tid == 0) { st->step_fail = 1; pcg_publish(W, seq, it, 1); } return; }
        if (!(rz > tol2*rz0) || since >= 16) { if (tid == 0) { st->lin_done = 1; W.pc_stat[0] += it + 1; W.pc_stat[1] += 1; if (it + 1 > W.pc_stat[2]) W.pc_stat[2] = it + 1; pcg_publish(W, seq, it, 1); } return; }
        __syncthreads();
        if (tid == 0) { E.scal[1] = rz; if (gain) E.scal[2] = rz; E.scal[3] = (double)since; pcg_publish(W, seq, it, 0); }
    }
    if (mode == 1) {
        Cm[tid] = c; Dm[tid] = d;
        __syncthreads();
        { const double cs = 0.5*(Cm[i*ECG_T + j] + Cm[j*ECG_T + i]); __syncthreads(); Cm[tid] = cs; }
        __syncthreads();
        double dmax = 0.0; for (int k = 0; k < ECG_T; k++) dmax = fmax(dmax, Cm[k*ECG_T + k]);
        // right-looking Cholesky, lower triangle in place; a pivot below the threshold drops its direction
        for (int k = 0; k < ECG_T; k++) {
            const double piv = Cm[k*ECG_T + k];
            const bool ok = piv > 1e-12*dmax;
            __syncthreads();
            if (tid == 0) { dead[k] = ok ? 0 : 1; dinv[k] = ok ? 1.0/sqrt(piv) : 0.0; }
            __syncthreads();
            if (j == k && i >= k) Cm[i*ECG_T + k] = i == k ? (ok ? sqrt(piv) : 1.0) : Cm[i*ECG_T + k]*dinv[k];
            __syncthreads();
            if (i > k && j > k && j <= i) Cm[i*ECG_T + j] -= Cm[i*ECG_T + k]*Cm[j*ECG_T + k];
            __syncthreads();
        }
        E.Lm[tid] = Cm[tid];
        if (tid < ECG_T) E.Lm[ECG_T*ECG_T + tid] = (double)dead[tid];
    } else {
        Cm[tid] = E.Lm[tid]; Dm[tid] = c;
        if (tid < ECG_T) dead[tid] = E.Lm[ECG_T*ECG_T + tid] != 0.0;
        __syncthreads();
    }
    // Y = C^-1 D: thread j < T takes right-hand side column j (forward, backward); dropped directions get 0
    __syncthreads();
    if (tid < ECG_T) {
        const int col = tid; double y[ECG_T];
        for (int r = 0; r < ECG_T; r++) { double v = Dm[r*ECG_T + col]; for (int k = 0; k < r; k++) v -= Cm[r*ECG_T + k]*y[k]; y[r] = dead[r] ? 0.0 : v/Cm[r*ECG_T + r]; }
        for (int r = ECG_T - 1; r >= 0; r--) { double v = y[r]; for (int k = r + 1; k < ECG_T; k++) v -= Cm[k*ECG_T + r]*y[k]; y[r] = dead[r] ? 0.0 : v/Cm[r*ECG_T + r]; }
        for (int r = 0; r < ECG_T; r++) E.Y[r*ECG_T + col] = y[r];
    }
    __syncthreads();
    if (mode == 1 && tid < ECG_T) { double s = 0.0; for (int col = 0; col < ECG_T; col++) s += E.Y[tid*ECG_T + col]; E.y1[tid] = s; }
}

// mode 1: R -= Q~ Y, x += P~ y1.   mode 2: P~ = Z - P~ Y (first: P~ = Z).   One wave per row at a time, a lane per column.
__global__ __launch_bounds__(256) void k_ecg_update(Work W, MsBuf M, EcgBuf E, int mode, int first) {
    const LmState *st = W.st;
    if (st->done || st->step_fail || st->lin_done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, c = lane & (ECG_T - 1);
    double y[ECG_T];
#pragma unroll
    for (int k = 0; k < ECG_T; k++) y[k] = first ? 0.0 : E.Y[k*ECG_T + c];
    const double y1 = (mode == 1 && lane < ECG_T) ? E.y1[lane] : 0.0;
    for (int u = 0; u < PCG_PPW; u++) {
        const int a = (blockIdx.x*4 + wave)*PCG_PPW + u;
        if (a >= W.n_kf) break;
        const int ia = W.fidx[a];
        if (ia < 0) continue;
        for (int r = 0; r < 6; r++) {
            const size_t row = (size_t)(6*ia + r)*ECG_T;
            const double src = lane < ECG_T ? (mode == 1 ? E.Q[row + lane] : E.P[row + lane]) : 0.0;      // the row of Q~ / P~: lane k holds entry k
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < ECG_T; k++) acc = fma(readlane_f64(src, k), y[k], acc);
            if (mode == 1) {
                if (lane < ECG_T) M.R[row + c] -= acc;
                const double px = lane < ECG_T ? E.P[row + lane]*y1 : 0.0;
                const double s = wave_sum1(px);
                if (lane == 0) W.pc_x[6*ia + r] += s;
            } else if (lane < ECG_T) E.P[row + c] = M.X[row + c] - acc;
        }
    }
}

// dp = x by keyframe, g restored, flags cleared (as k_pcg_finish)
__global__ __launch_bounds__(PCG_ET) void k_ecg_finish(Work W, int its_enqueued) {
    LmState *st = W.st;
    if (st->done) return;
    const int tid = threadIdx.x, a = blockIdx.x*32 + tid/6, k = tid % 6;
    const int fail = st->step_fail, conv = st->lin_done;
    if (a < W.n_kf) { const int ia = W.fidx[a];
        W.dp[6*a + k] = (ia >= 0 && !fail) ? W.pc_x[6*ia + k] : 0.0; }
    __syncthreads();
    if (blockIdx.x == 0 && tid == 0) {
        if (!conv && !fail) { W.pc_stat[0] += its_enqueued; W.pc_stat[1] += 1; W.pc_stat[3] += 1; if (its_enqueued > W.pc_stat[2]) W.pc_stat[2] = its_enqueued; }
        st->lin_done = 0; }
}
