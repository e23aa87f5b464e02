// Pose-only optimisation (optimizer::PoseOptim -> PyrPoseOptim, optimizer.cc:1060-1327: one free pose, every landmark frozen in its
// host keyframe -- residual rows R3 / R7).  Included by tsba.hip after tsba_solve.h.
//
// The general pipeline spends several dependent launches per LM iteration (schur, solve + back, linearize, mid) and gives the single
// (target, host = frozen) pair of a frame to ONE wave: 3000 scene blocks = 47 serial rounds.  With 6 unknowns there is nothing to
// schedule between the sweeps: an LM step is "sum the sweep's partial sums, decide, solve 6 x 6, sweep at the candidate", run by 7 - 18
// workgroups of 256 threads (256 scene blocks or 32 text features x 8 taps each).
//
//   k_pose_pass (round 5, the path taken where the grid is resident at once): ALL steps of a pass in one launch; the state lives in
//                     every workgroup's registers, the sums are exchanged by polling their values (three buffers; see the kernel).
//   k_pose_iter(k) (one launch per step: devices too small for the grid, tsba_debug_options.pass_launches, rounds 1 - 4):
//                     every workgroup, redundantly and bit-identically:
//                       state_k   <- pst[k & 1]                                      (k = 0: built from W.st / W.pose)
//                       sums(cand) <- sum over workgroups of part[(k + 1) & 1]       (fixed order: deterministic)
//                       Ceres decision for trial k-1 (k = 0: Jacobi scaling + gradient test of the first linearisation)
//                       (M + D/radius) dp = -c  by 6x6 LDL^T in registers, candidate on the quaternion manifold
//                     workgroup 0 writes state_{k+1} -> pst[(k + 1) & 1] and the pinned progress word;
//                     every workgroup then sweeps ITS observations at the candidate -> part[k & 1].
//                     State and partial sums are double-buffered by the launch ordinal, so no workgroup reads what another one writes in
//                     the same launch; a converged pass copies its state forward.
//
// Both share pose_step / pose_sweep_obs.  k_outlier's extra workgroup installs the final state into W.st / W.pose; mu / sigma, participation
// and gauge are k_pose_begin (one launch), the outlier pass is the general path's kernel.
#pragma once

#define POSE_WG 256
#define POSE_NW (POSE_WG/64)
#define POSE_TILE 128                   /* workgroups of partial sums staged in LDS at a time */
#define POSE_LDS (POSE_NW*14*65 + POSE_NW*32)        /* >= POSE_TILE*28 */

struct PoseSums { double M[21], c[6], cost; };
struct PoseState {
    LmState S;
    double M[21], c[6], sig[6], dgs[6], x[7], cand[7], mcc, step2;
    int fail, pad;
};

// What a thread's observation needs besides the pose -- constant over a pass (the landmarks are frozen, the flags change in the outlier pass only): loaded once
// per launch.  Scene block: T12 = T_rw of the point's host, v = ray (2), rho, u, v.  Text tap: T12 = T_wr of the plane's host, v = mu, sigma, feature u, v,
// reference intensity, theta (3).
struct PoseObs { bool on; double T12[12], v[8]; const uint8_t *img; };
__device__ __forceinline__ void pose_obs_load(const Work &W, const LevelDev &L, const double *rho, const double *theta, int b, int nb_sc, PoseObs &O) {
    const int tid = threadIdx.x;
    O.on = false; O.img = nullptr;
#pragma unroll
    for (int q = 0; q < 12; q++) O.T12[q] = 0.0;
#pragma unroll
    for (int q = 0; q < 8; q++) O.v[q] = 0.0;
    if (b < nb_sc) {
        const int c = b*POSE_WG + tid;
        if (c < L.n_sc && (!W.filter_good || W.sgood[L.sc_flag[c]])) {
            const int pt = L.sc_pt[c];
            O.on = true;
#pragma unroll
            for (int q = 0; q < 12; q++) O.T12[q] = W.pt_Trw[12*(size_t)pt + q];
            O.v[0] = W.pt_ray[2*pt]; O.v[1] = W.pt_ray[2*pt+1]; O.v[2] = rho[pt]; O.v[3] = L.sc_uv[2*c]; O.v[4] = L.sc_uv[2*c+1];
        }
    } else {
        const int fi = (b - nb_sc)*(POSE_WG/8) + (tid >> 3), kt = tid & 7;
        if (fi < L.n_pf) {
            const int g = L.pf_g[fi], f = L.pf_f[fi];
            const int4 ra = ((const int4 *)L.tg_rec)[2*g], rb = ((const int4 *)L.tg_rec)[2*g + 1];
            const int tb = ra.x, j = ra.z, fg = rb.w;
            O.v[0] = W.musig[2*tb]; O.v[1] = W.musig[2*tb+1];
            O.v[2] = L.tfeat_uv[2*f]; O.v[3] = L.tfeat_uv[2*f+1]; O.v[4] = L.tfeat_ref[8*(size_t)f + kt];
            O.img = L.img[ra.y];
#pragma unroll
            for (int q = 0; q < 12; q++) O.T12[q] = W.text_Twr[12*(size_t)j + q];
            O.v[5] = theta[3*j]; O.v[6] = theta[3*j+1]; O.v[7] = theta[3*j+2];
            O.on = (!W.filter_good || (W.tobs_good[tb] && W.tfgood[fg + L.tfeat_raw[f]])) && O.v[1] != 0.0;
        }
    }
}
// this workgroup's share of the observations at pose p7 -- 256 scene blocks, or 32 text features x 8 taps: thread t < 28 returns
// the workgroup total of value t (0..20 = sum w J^T J upper / sym6 order, 21..26 = sum w J^T r, 27 = sum rho / 2)
__device__ __forceinline__ double pose_sweep_obs(const Work &W, const LevelDev &L, const double *p7, const PoseObs &O, int b, int nb_sc, double *lds) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; k++) acc[k] = 0.0;
    Pose C; load_pose(p7, C);
    if (b < nb_sc) {
        // ---- scene blocks (R3), one per thread: frozen host, T_rw stored with the point
        if (O.on) {
            PairT T; pair_from_Trw(C, O.T12, T);
            double r[2], jt[2][6], jl[2];
            scene_block(T, C.t, O.v[0], O.v[1], O.v[2], O.v[3], O.v[4],
                        W.K0[0], W.K0[1], W.K0[2], W.K0[3], W.w_sx, W.w_sy, r, jt, jl);
            double wgt; acc[27] = 0.5*huber(r[0]*r[0] + r[1]*r[1], W.huber_s, wgt);
            int q = 0;
#pragma unroll
            for (int a = 0; a < 6; a++)
#pragma unroll
                for (int cc = a; cc < 6; cc++) { acc[q] = wgt*(jt[0][a]*jt[0][cc] + jt[1][a]*jt[1][cc]); q++; }
#pragma unroll
            for (int a = 0; a < 6; a++) acc[21 + a] = wgt*(jt[0][a]*r[0] + jt[1][a]*r[1]);
        }
    } else {
        // ---- photometric blocks (R7): thread = (feature, tap); (group, feature) from the frame's flat feature list
        const int kt = tid & 7;
        double r = 0.0, jt[6] = {0, 0, 0, 0, 0, 0}; const bool good = O.on;
        if (good) {
            PairT T; pair_from_Twr(C, O.T12, T);
            const double th[3] = { O.v[5], O.v[6], O.v[7] };
            const double mx = (O.v[2] + TAP_DX[kt] - L.K[2])/L.K[0], my = (O.v[3] + TAP_DY[kt] - L.K[3])/L.K[1];   // tool.cc:1561
            double jl[3];
            r = text_tap(T, C.t, th, mx, my, L.K[0], L.K[1], L.K[2], L.K[3], O.img, L.img_w, L.img_h,
                         O.v[0], O.v[1], 1.0/O.v[1], O.v[4], W.w_t, true, jt, jl);
        }
        double s8 = r*r;                                        // the block's squared norm: its 8 taps sit on 8 neighbouring lanes
        s8 += __shfl_xor(s8, 1, 64); s8 += __shfl_xor(s8, 2, 64); s8 += __shfl_xor(s8, 4, 64);
        double wgt; const double rho_h = 0.5*huber(s8, W.huber_t, wgt);
        const double wg = good ? wgt : 0.0;
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int cc = a; cc < 6; cc++) { acc[q] = wg*(jt[a]*jt[cc]); q++; }
#pragma unroll
        for (int a = 0; a < 6; a++) acc[21 + a] = wg*(jt[a]*r);
        acc[27] = (good && kt == 0) ? rho_h : 0.0;
    }
    double *xw = lds + POSE_NW*14*65;                        // two transposes of 14 values per wave (LDS stays under 32 KB)
    const double t0 = wave_sum_to_lane_mw<14>(acc, lds + wave*14*65, lane);
    const double t1 = wave_sum_to_lane_mw<14>(acc + 14, lds + wave*14*65, lane);
    if (lane < 14) { xw[wave*32 + lane] = t0; xw[wave*32 + 14 + lane] = t1; }
    __syncthreads();
    double tot = 0.0;
    if (tid < 28) {
#pragma unroll
        for (int w = 0; w < POSE_NW; w++) tot += xw[w*32 + tid];
    }
    return tot;
}
__device__ __forceinline__ double pose_sweep_wg(const Work &W, const LevelDev &L, const double *p7, const double *rho, const double *theta, int b, int nb_sc, double *lds) {
    PoseObs O; pose_obs_load(W, L, rho, theta, b, nb_sc, O);
    return pose_sweep_obs(W, L, p7, O, b, nb_sc, lds);
}

// the state a pass starts from (k = 0)
__device__ __forceinline__ void pose_state_init(const Work &W, PoseState &P) {
    P.S = *W.st;
#pragma unroll
    for (int q = 0; q < 7; q++) { P.x[q] = W.pose[P.S.cur][q]; P.cand[q] = P.x[q]; }
#pragma unroll
    for (int q = 0; q < 21; q++) P.M[q] = 0.0;
#pragma unroll
    for (int q = 0; q < 6; q++) { P.c[q] = 0.0; P.sig[q] = 1.0; P.dgs[q] = 0.0; }
    P.mcc = 0.0; P.step2 = 0.0; P.fail = 0; P.pad = 0;
}
// One step of the state machine, executed redundantly and bit-identically by every workgroup: the sums of the last sweep (src: [G][28]), the Ceres decision for
// the trial they belong to (k = 0: scaling + gradient test of the first linearisation), then the next trial's 6 x 6 solve and candidate.  POLLED: the sums are
// being written by the other workgroups of THIS launch (k_pose_pass): every value is read with device-coherent loads until it is there (NaN = not yet).
__device__ __forceinline__ double pose_poll(const double *p) {
    double v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int spins = 0; v != v && spins < (1 << 17); spins++) { __builtin_amdgcn_s_sleep(1); v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    if (v != v) { v = __builtin_inf(); atomicAdd(&ts_poll_giveups, 1u); }      // (counted: tsba_report.poll_timeouts; an infinite sum fails the trial)
    return v;
}
__device__ __forceinline__ void pose_publish(double *p, double v) { __hip_atomic_store(p, v == v ? v : __builtin_inf(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <bool POLLED>
__device__ __forceinline__ void pose_step(const Work &W, const tsba_options &o, int k, int G, const double *src, bool is_free, PoseState &P, double *lds, double *s_ps, double *s_sum) {
    const int tid = threadIdx.x;
    LmState &S = P.S;
    // ---- sums of the last sweep: 28 values x G workgroups.  All loads in flight at once (staged through LDS), then eight
    // partial sums per value in a fixed order
    {
        const int i = tid % 28, h = tid / 28;                // h < 8 for tid < 224
        double part = 0.0;
        for (int g0 = 0; g0 < G; g0 += POSE_TILE) {
            const int ng = min(POSE_TILE, G - g0);
            const double2 *s2 = (const double2 *)(src + (size_t)g0*28);
            for (int e = tid; e < ng*14; e += POSE_WG) {
                if (POLLED) { const double *sd = (const double *)(s2 + e); ((double2 *)lds)[e] = make_double2(pose_poll(sd), pose_poll(sd + 1)); }
                else ((double2 *)lds)[e] = s2[e];
            }
            __syncthreads();
            if (h < 8) for (int g = h; g < ng; g += 8) part += lds[g*28 + i];
            __syncthreads();
        }
        if (h < 8) s_ps[h*28 + i] = part;
        __syncthreads();
        if (tid < 28) { double t = 0.0;
#pragma unroll
            for (int q = 0; q < 8; q++) t += s_ps[q*28 + tid];
            s_sum[tid] = t; }
        __syncthreads();
    }
    PoseSums Cn;
#pragma unroll
    for (int q = 0; q < 21; q++) Cn.M[q] = s_sum[q];
#pragma unroll
    for (int q = 0; q < 6; q++) Cn.c[q] = s_sum[21 + q];
    Cn.cost = s_sum[27];
    auto scale = [&](bool first) {                           // k_postlin: Jacobi scaling fixed at the first linearisation of the pass
#pragma unroll
        for (int q = 0; q < 6; q++) { const double h = P.M[sym6(q, q)]; if (first) P.sig[q] = 1.0/(1.0 + sqrt(h));
            P.dgs[q] = clampd(P.sig[q]*P.sig[q]*h, W.min_diag, W.max_diag)/(P.sig[q]*P.sig[q]); }
    };
    auto install = [&]() {                                   // the swept point becomes x
#pragma unroll
        for (int q = 0; q < 21; q++) P.M[q] = Cn.M[q];
#pragma unroll
        for (int q = 0; q < 6; q++) P.c[q] = Cn.c[q];
        double g = 0.0, v = 0.0;
        if (is_free) {
#pragma unroll
            for (int q = 0; q < 6; q++) g = fmax(g, fabs(P.c[q]));
#pragma unroll
            for (int q = 0; q < 7; q++) v += P.x[q]*P.x[q];
        }
        S.gmax = g; S.x_norm = sqrt(v);
    };
    if (k == 0) {
        install(); scale(true);
        S.x_cost = Cn.cost; S.cost0 = Cn.cost; S.first = 0; S.need_lin = 0; S.n_lin++;
        if (S.gmax <= o.gradient_tolerance) { S.done = 1; S.term = 3; }
    } else {
        // ---- k_decide for the trial the previous launch prepared
        S.it++;
        S.cand_cost = P.fail ? S.cand_cost : Cn.cost; S.model_change = 0.5*P.mcc; S.step_norm = sqrt(P.step2);
        const double mcc = 0.5*P.mcc;
        if (P.fail || !(mcc > 0.0)) {
            S.step_fail = 0;
            if (++S.invalid >= 5) { S.done = 1; S.term = 5; }
            else S.radius *= 0.5;
        } else {
            S.invalid = 0; S.n_cost++;
            double cost = Cn.cost; if (!(cost == cost)) cost = 1.7976931348623157e308;
            const double cost_change = S.x_cost - cost;
            if (S.step_norm <= o.parameter_tolerance*(S.x_norm + o.parameter_tolerance)) { S.done = 1; S.term = 2; }
            else if (fabs(cost_change) <= o.function_tolerance*S.x_cost) { S.done = 1; S.term = 1; }
            else {
                const double rel = cost_change/mcc;
                if (rel > o.min_relative_decrease) {
#pragma unroll
                    for (int q = 0; q < 7; q++) P.x[q] = P.cand[q];
                    install(); scale(false); S.accepted++; S.n_lin++;
                    S.x_cost = cost;
                    double t = 2.0*rel - 1.0, f = 1.0 - t*t*t; if (f < 1.0/3.0) f = 1.0/3.0;
                    S.radius = fmin(S.radius/f, o.max_radius);
                    S.decrease_factor = 2.0;
                    if (S.gmax <= o.gradient_tolerance) { S.done = 1; S.term = 3; }
                } else {
                    S.radius = S.radius/S.decrease_factor; S.decrease_factor *= 2.0;
                }
            }
        }
        if (!S.done) {
            if (S.it >= S.max_it) { S.done = 1; S.term = 0; }
            else if (S.radius < o.min_radius) { S.done = 1; S.term = 4; }
        }
    }
    if (!S.done) {
        // ---- k_schur + k_solve + k_back for one pose: (M + D/radius) dp = -c, candidate, step norm, model cost change
        const double irad = 1.0/S.radius;
        double dp[6] = {0, 0, 0, 0, 0, 0}; bool fail = S.step_fail != 0 || !is_free;
        if (!fail) {
            double s[21], l[15], d[6], id[6]; bool bad = false;
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = 0; c <= r; c++) s[(r*(r + 1))/2 + c] = P.M[sym6(c, r)] + (r == c ? P.dgs[r]*irad : 0.0);
            ldl6(s, l, d, id, bad);
            if (bad) fail = true;
            else {
                double z[6];
#pragma unroll
                for (int r = 0; r < 6; r++) { double v = P.c[r];
#pragma unroll
                    for (int q = 0; q < r; q++) v -= l[(r*(r - 1))/2 + q]*z[q];
                    z[r] = v; }
#pragma unroll
                for (int r = 0; r < 6; r++) z[r] *= id[r];
#pragma unroll
                for (int r = 5; r >= 0; r--) { double v = z[r];
#pragma unroll
                    for (int q = r + 1; q < 6; q++) v -= l[(q*(q - 1))/2 + r]*z[q];
                    z[r] = v; }
#pragma unroll
                for (int r = 0; r < 6; r++) dp[r] = -z[r];
            }
        }
        double step2 = 0.0, mcc = 0.0;
        if (!fail) {
            double q4[4] = { P.x[0], P.x[1], P.x[2], P.x[3] }, qn[4];
            quat_plus(q4, dp, qn);
#pragma unroll
            for (int q = 0; q < 4; q++) { P.cand[q] = qn[q]; step2 += (qn[q] - q4[q])*(qn[q] - q4[q]); }
#pragma unroll
            for (int q = 0; q < 3; q++) { P.cand[4 + q] = P.x[4 + q] + dp[3 + q]; step2 += dp[3 + q]*dp[3 + q]; }
#pragma unroll
            for (int q = 0; q < 6; q++) { const double lam = P.dgs[q]*irad; mcc += lam*dp[q]*dp[q] - P.c[q]*dp[q]; }
        } else {
#pragma unroll
            for (int q = 0; q < 7; q++) P.cand[q] = P.x[q];
        }
        P.mcc = mcc; P.step2 = step2; P.fail = fail ? 1 : 0;
    }
}

// k = -1: linearisation at the start point (into part[1]); k >= 0: see the header
__global__ __launch_bounds__(POSE_WG) void k_pose_iter(Work W, LevelDev L, tsba_options o, int k, int G) {
    __shared__ double lds[POSE_LDS];                        // the sweep's transposes; before it, the staged partial sums
    __shared__ double s_ps[8*28];
    __shared__ double s_sum[32];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int nb_sc = (L.n_sc + POSE_WG - 1)/POSE_WG;
    const bool is_free = W.fidx[0] >= 0;                    // (a constant pose leaves every block out of the reduced program)
    if (k < 0) {
        const LmState *st = W.st;
        if (st->done) return;
        const int cur = st->cur;
        double tot = 0.0;
        if (is_free) tot = pose_sweep_wg(W, L, W.pose[cur], W.rho[cur], W.theta[cur], b, nb_sc, lds);
        if (tid < 28) W.ppart[(size_t)(1*G + b)*28 + tid] = tot;
        return;
    }
    // ---- state
    PoseState P;
    if (k == 0) pose_state_init(W, P);
    else P = W.pst[k & 1];
    PoseState *Pn = W.pst + ((k + 1) & 1);
    LmState &S = P.S;
    if (S.done) { if (b == 0 && tid == 0) *Pn = P; return; }
    pose_step<false>(W, o, k, G, W.ppart + (size_t)((k + 1) & 1)*G*28, is_free, P, lds, s_ps, s_sum);
    if (b == 0 && tid == 0) {
        *Pn = P;
        if (W.hprog) *W.hprog = ((unsigned long long)W.pass_seq << 32) | ((unsigned long long)(unsigned)S.it << 1) | (unsigned long long)(S.done != 0);
    }
    if (S.done || P.fail) return;
    // ---- speculative linearisation at the candidate (skipped after a failed step, as the general path does)
    const double tot = pose_sweep_wg(W, L, P.cand, W.rho[S.cur], W.theta[S.cur], b, nb_sc, lds);
    if (tid < 28) W.ppart[(size_t)((k & 1)*G + b)*28 + tid] = tot;
}


// The whole pass in ONE launch (round 5): the steps of k_pose_iter in a loop, and where the launches were, nothing but the sums themselves.  Every workgroup
// carries the state in registers -- they all compute the same bits, so they all leave the loop in the same iteration without telling each other --; the sums of
// step k go to buffer k mod 3 of W.ppart as device-coherent stores and are read by polling every value until it is no longer NaN (a NaN sum is published as
// +inf: the trial fails either way).  A workgroup in step k has seen every other workgroup's sums of step k - 1, so all of them have finished reading the
// buffer of step k - 2: that is the one it resets (its own 28 values) for step k + 1; k_pose_begin fills all three with NaNs before the pass.  A wait that runs
// into its bound is counted (tsba_report.poll_timeouts) and fails the trial; the launch is used only where all G workgroups (a few tens: 256 scene blocks or
// 32 text features each) are resident at once (tsba.hip: grid_resident).  Saved: a launch boundary per LM step and the two steps of empty launches the host
// stays ahead -- C3 0.28 -> 0.2x ms; a counter barrier (fence + atomic + polling the counter, then reading the sums) in the same place gave 0.232.
__global__ __launch_bounds__(POSE_WG) void k_pose_pass(Work W, LevelDev L, tsba_options o, int G, int max_k) {
    __shared__ double lds[POSE_LDS];
    __shared__ double s_ps[8*28];
    __shared__ double s_sum[32];
    const int tid = threadIdx.x, b = blockIdx.x;
    const int nb_sc = (L.n_sc + POSE_WG - 1)/POSE_WG;
    const bool is_free = W.fidx[0] >= 0;
    if (W.st->done) return;                                  // (uniform: written before this launch)
    const int cur = W.st->cur;
    PoseObs O; pose_obs_load(W, L, W.rho[cur], W.theta[cur], b, nb_sc, O);      // (once: every step's sweep runs from registers; a text tap still fetches its pixels)
    {   // linearisation at the start point ("step -1": buffer 2)
        double tot = 0.0;
        if (is_free) tot = pose_sweep_obs(W, L, W.pose[cur], O, b, nb_sc, lds);
        if (tid < 28) pose_publish(&W.ppart[(size_t)(2*G + b)*28 + tid], tot);
    }
    PoseState P;
    pose_state_init(W, P);
    LmState &S = P.S;
    for (int k = 0; k <= max_k; k++) {
        __syncthreads();                                     // (lds / s_sum of the last sweep and step are free)
        pose_step<true>(W, o, k, G, W.ppart + (size_t)((k + 2) % 3)*G*28, is_free, P, lds, s_ps, s_sum);
        if (b == 0 && tid == 0 && W.hprog) *W.hprog = ((unsigned long long)W.pass_seq << 32) | ((unsigned long long)(unsigned)S.it << 1) | (unsigned long long)(S.done != 0);
        if (S.done) break;
        if (tid < 28) __hip_atomic_store(&W.ppart[(size_t)(((k + 1) % 3)*G + b)*28 + tid], __builtin_nan(""), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (step k - 2's: read by everybody)
        __syncthreads();
        // speculative linearisation at the candidate; after a failed step (no candidate) nobody looks at the sums, but everybody waits for them: zeros
        double tot = 0.0;
        if (!P.fail) tot = pose_sweep_obs(W, L, P.cand, O, b, nb_sc, lds);
        // (release: this thread's NaN reset of the NEXT-BUT-ONE buffer above is visible before these sums are -- a reader that has seen everybody's sums of step k
        // may then rely on the reset: nothing but timing ordered the two relaxed stores before; round-5 advisor.  The reset was issued a whole sweep ago: the fence waits for nothing)
        if (tid < 28) { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); pose_publish(&W.ppart[(size_t)((k % 3)*G + b)*28 + tid], tot); }
    }
    if (b == 0 && tid == 0) W.pst[0] = P;                    // (k_outlier installs it)
}

// pass start of the pose-only path in one launch: workgroups 0 .. n_tg-1 = k_musigma (the pose is the same in both parameter
// buffers here); workgroup n_tg = k_pass_reset + k_participation + k_gauge for one keyframe
__global__ __launch_bounds__(MS_THREADS) void k_pose_begin(Work W, LevelDev L, double radius0, int max_it, const uint8_t *kf_initial, double *sums_nan, int n_nan, LmState *log_prev) {
    if ((int)blockIdx.x < L.n_tg) { musigma_wg(W, L, blockIdx.x, W.pose[0], W.theta[0]); return; }
    __shared__ int cnt_s, cnt_t;
    const int tid = threadIdx.x;
    if (sums_nan) for (int e = tid; e < n_nan; e += MS_THREADS) sums_nan[e] = __builtin_nan("");      // k_pose_pass: "not there yet"
    if (tid == 0) { cnt_s = 0; cnt_t = 0; }
    __syncthreads();
    int ns = 0, nt = 0;
    for (int c = tid; c < L.n_sc; c += MS_THREADS) ns += (!W.filter_good || W.sgood[L.sc_flag[c]]) ? 1 : 0;
    for (int fi = tid; fi < L.n_pf; fi += MS_THREADS) {
        const int g = L.pf_g[fi], f = L.pf_f[fi], tb = L.tg_rec[8*g], fg = L.tg_rec[8*g + 7];
        nt += (!W.filter_good || (W.tobs_good[tb] && W.tfgood[fg + L.tfeat_raw[f]])) ? 1 : 0;
    }
    if (ns) atomicAdd(&cnt_s, ns);
    if (nt) atomicAdd(&cnt_t, nt);
    __syncthreads();
    if (tid == 0) {
        LmState *s = W.st;                                   // (every field but cur / n_lin / n_cost, which carry over)
        if (log_prev) *log_prev = *s;                        // the pass before this one left its final state here only (its outlier counts were still being added when it was installed)
        s->radius = radius0; s->decrease_factor = 2.0; s->x_cost = 0; s->x_norm = 0; s->cand_cost = 0; s->model_change = 0;
        s->step_norm = 0; s->gmax = 0; s->cost0 = 0;
        s->done = 0; s->need_lin = 1; s->first = 1; s->it = 0; s->accepted = 0; s->term = 0; s->invalid = 0; s->max_it = max_it;
        s->step_fail = 0; s->lcur = 0; s->lin_done = 0; s->pad2 = 0;
        s->ns_active = cnt_s; s->nt_active = cnt_t; s->n_bad_scene = 0; s->n_bad_tfeat = 0; s->n_bad_text = 0;
        const int in = (cnt_s > 0 || cnt_t > 0) ? 1 : 0, cst = (kf_initial[0] && in) ? 1 : 0;    // optimizer.cc:1562-1588 for one keyframe
        W.kf_in[0] = in; W.kf_const[0] = cst;
        W.fidx[0] = (in && !cst) ? 0 : -1; *W.nfree = (in && !cst) ? 1 : 0;
        if (W.hprog) { *W.hprog = (unsigned long long)W.pass_seq << 32; __threadfence_system(); }
    }
}
