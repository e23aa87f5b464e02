// The rest of an LM trial: k_back, k_decide, the split (multi-GPU) kernels, the outlier pass, test-hook evaluation kernels.  (part of the single translation unit tsba.hip: included there, in this order)
#pragma once
// ---- landmark back-substitution + candidate parameters.  256-thread blocks: BK_PT points (four threads each) | BK_TX texts (sixteen threads each) | 256 poses.
// The threads of a landmark take its slots in turn (slot o + t, o + t + T, ...), so a point of up to 24 slots and a text of up to 32 (48 where three are in flight) are
// one round of requests; the partial sums meet in a fixed order (xor shuffles: every thread of the landmark gets the same bits) and the landmark's
// first thread writes the result.  (One thread per landmark walked a text's twenty slots three at a time: seven dependent rounds.)
// POLL (k_solve_back): the block runs in the launch that solves the reduced system.  Everything that does not depend on the pose step -- offsets, state,
// the landmark's own terms and the records of its first round -- is requested first; then the block polls the step (W.dp, NaN until the solver
// workgroup publishes it; W.dp[N]: 0 solved / 1 failed) into LDS and goes on from there.  Same arithmetic in the same order either way.
#define BK_TP 4                             // threads of a point / of a text
#define BK_TT 16
#define BK_PT (256/BK_TP)
#define BK_TX (256/BK_TT)
static inline int back_blocks_pt(int n_pt) { return (n_pt + BK_PT - 1)/BK_PT; }
static inline int back_blocks_tx(int n_text) { return (n_text + BK_TX - 1)/BK_TX; }
template <bool POLL>
__device__ __forceinline__ void back_body(const Work &W, const LevelDev &L, int nb_pt, int nb_tx, int b, int tid, bool live, double *red, double *dps) {
    LmState *st = W.st;
    const bool is_pt = live && b < nb_pt, is_tx = live && !is_pt && b < nb_pt + nb_tx;
    const int T = is_pt ? BK_TP : BK_TT, part = is_pt ? tid % BK_TP : tid % BK_TT;           // threads of a landmark, this thread's place among them
    const int j = is_pt ? b*BK_PT + tid/BK_TP : is_tx ? (b - nb_pt)*BK_TX + tid/BK_TT : 0;
    const bool lm = (is_pt && j < W.n_pt) || (is_tx && j < W.n_text);
    // static offsets / slot poses of this thread's landmark first: in flight together with the LM state
    int o = 0, e = 0, act_ = 0, a0[3] = {0, 0, 0};
    if (lm && is_pt) { o = L.pls_off[j]; e = L.pls_off[j+1]; act_ = W.act_pt[j];
#pragma unroll
        for (int u = 0; u < 3; u++) a0[u] = part + BK_TP*u < 6 ? L.pt_pose6[6*(size_t)j + part + BK_TP*u] : 0; }       // (the poses of a point's first six slots: this thread's share of them)
    else if (lm) { o = L.tls_off[j]; e = L.tls_off[j+1]; act_ = W.act_tx[j]; }
    if (st->done) return;
    const int cur = st->cur;
    const double irad = 1.0/st->radius;
    bool fail = st->step_fail;
    const LinBuf &B = W.lb[st->lcur];
    const double *dp = W.dp;
    const bool on = lm && e > o && act_;
    auto slot_of = [&](int k) { return o + part + T*k; };        // this thread's k-th slot (valid while < e)
    // POLL: the first round of records (a point's six, a text's two), then the step
    double wq[6][6], tq[12], rh0 = 0.0, Vj0 = 0.0, Dj0 = 0.0, bj0 = 0.0; int aq[6] = {0, 0, 0, 0, 0, 0}, pfi = -1;      // (tq: a text's b | V | damping, a pose's parameters)
    if (POLL) {
#pragma unroll
        for (int u = 0; u < 6; u++)
#pragma unroll
            for (int k = 0; k < 6; k++) wq[u][k] = 0.0;
#pragma unroll
        for (int k = 0; k < 12; k++) tq[k] = 0.0;
        if (is_pt && lm) { rh0 = W.rho[cur][j];
            if (on) { VDB_LOADB(B, j, W.n_pt, Vj0, Dj0, bj0);
#pragma unroll
                for (int u = 0; u < 6; u++) { const int sl = min(slot_of(u), e - 1); aq[u] = part + BK_TP*u < 6 ? a0[u < 3 ? u : 0] : L.pslot_pose[sl];
#pragma unroll
                    for (int k = 0; k < 6; k++) wq[u][k] = B.w_pt[(size_t)sl*PT_REC + k]; } } }
        else if (is_tx && on) {
#pragma unroll
            for (int u = 0; u < 2; u++) { const int sl = min(slot_of(u), e - 1); aq[u] = L.tslot_pose[sl];
#pragma unroll
                for (int k = 0; k < 18; k++) wq[3*u + k/6][k % 6] = B.w_tx[(size_t)sl*TX_REC + k]; }
#pragma unroll
            for (int k = 0; k < 3; k++) { tq[k] = B.b_tx[(size_t)k*W.n_text + j]; tq[9 + k] = B.dgs_tx[(size_t)k*W.n_text + j]; }
#pragma unroll
            for (int k = 0; k < 6; k++) tq[3 + k] = B.V_tx[(size_t)k*W.n_text + j]; }
        else if (live && !is_pt && !is_tx) { const int a = (b - nb_pt - nb_tx)*256 + tid;          // a pose: its parameters, damping and gradient rows
            if (a < W.n_kf) { pfi = W.fidx[a];
#pragma unroll
                for (int k = 0; k < 7; k++) tq[k] = W.pose[cur][7*a + k];
#pragma unroll
                for (int k = 0; k < 6; k++) { wq[0][k] = B.dgs_p[6*a + k]; wq[1][k] = B.bp[6*a + k]; } } }
#pragma unroll
        for (int k = 0; k < 12; k++) asm volatile("" : "+v"(tq[k]));
#pragma unroll
        for (int u = 0; u < 6; u++) {
#pragma unroll
            for (int k = 0; k < 6; k++) asm volatile("" : "+v"(wq[u][k]));
            asm volatile("" : "+v"(aq[u])); }
        asm volatile("" : "+v"(rh0)); asm volatile("" : "+v"(Vj0)); asm volatile("" : "+v"(Dj0)); asm volatile("" : "+v"(bj0));
        const int nd = W.N + 1, t768 = threadIdx.x;
        for (int k0 = 0; k0 < nd; k0 += (int)blockDim.x) { const int k = k0 + t768;
            if (k < nd) { double v = co_load(&W.dp[k]);
                for (int spins = 0; v != v && spins < (1 << 16); spins++) { __builtin_amdgcn_s_sleep(1); v = co_load(&W.dp[k]); }
                if (v != v) atomicAdd(&ts_poll_giveups, 1u);
                dps[k] = v == v ? v : __builtin_inf(); } }
        __syncthreads();
        fail = fail || dps[W.N] != 0.0;                         // (the assembly's flag, read with the state, or the solver's: what k_back reads after both launches)
        dp = dps;
    }
    double step2 = 0.0, mcc = 0.0;
    if (is_pt) {
        if (lm) {
            double rh = POLL ? rh0 : W.rho[cur][j], d = 0.0;
            if (!fail && on) {
                double Vj, Dj, bj;
                if (POLL) { Vj = Vj0; Dj = Dj0; bj = bj0; } else VDB_LOADB(B, j, W.n_pt, Vj, Dj, bj);
                double acc = 0.0;
                auto pt_round = [&](int k0, const int (&a)[6], const double (&w)[6][6]) {     // six of this thread's slots; dp is 0 for constant / absent poses
                    if (POLL) {                                                     // (the step is in LDS: no batch of requests to keep in registers)
#pragma unroll
                        for (int u = 0; u < 6; u++) { const double *du = dp + 6*a[u];
#pragma unroll
                            for (int k = 0; k < 6; k++) acc = slot_of(k0 + u) < e ? fma(w[u][k], du[k], acc) : acc; }
                        return; }
                    double dpv[6][6];
#pragma unroll
                    for (int u = 0; u < 6; u++)
#pragma unroll
                        for (int k = 0; k < 6; k++) dpv[u][k] = dp[6*a[u] + k];
#pragma unroll
                    for (int u = 0; u < 6; u++)
#pragma unroll
                        for (int k = 0; k < 6; k++) acc = slot_of(k0 + u) < e ? fma(w[u][k], dpv[u][k], acc) : acc; };
                if (POLL) pt_round(0, aq, wq);
                for (int k0 = POLL ? 6 : 0; slot_of(k0) < e; k0 += 6) {
                    int a[6]; double w[6][6];
#pragma unroll
                    for (int u = 0; u < 6; u++) a[u] = part + BK_TP*(k0 + u) < 6 ? a0[(k0 + u) % 3] : L.pslot_pose[min(slot_of(k0 + u), e - 1)];
#pragma unroll
                    for (int u = 0; u < 6; u++)
#pragma unroll
                        for (int k = 0; k < 6; k++) w[u][k] = B.w_pt[(size_t)(min(slot_of(k0 + u), e - 1))*PT_REC + k];
                    pt_round(k0, a, w);
                }
#pragma unroll
                for (int x = 1; x < BK_TP; x <<= 1) acc += __shfl_xor(acc, x, 64);  // the threads' sums: the same bits in every one of them
                acc = bj + acc;
                const double lam = Dj*irad;
                d = -acc/(Vj + lam);
                if (part == 0) { step2 = d*d; mcc = lam*d*d - bj*d; }
            }
            if (part == 0) W.rho[cur ^ 1][j] = rh + d;
        }
    } else if (is_tx) {
        if (lm) {
            double d[3] = {0,0,0};
            if (!fail && on) {
                double acc[3] = { 0.0, 0.0, 0.0 };
                auto tx_slot = [&](int a, const double *w) {
#pragma unroll
                    for (int k = 0; k < 6; k++) { const double dk = dp[6*a + k]; acc[0] = fma(w[k*3], dk, acc[0]); acc[1] = fma(w[k*3 + 1], dk, acc[1]); acc[2] = fma(w[k*3 + 2], dk, acc[2]); } };
                if (POLL) {
#pragma unroll
                    for (int u = 0; u < 2; u++) if (slot_of(u) < e) { double w[18];
#pragma unroll
                        for (int k = 0; k < 18; k++) w[k] = wq[3*u + k/6][k % 6];
                        tx_slot(aq[u], w); } }
                constexpr int TB = POLL ? 2 : 3;                                    // slots in flight (the sums run slot by slot either way)
                for (int k0 = POLL ? 2 : 0; slot_of(k0) < e; k0 += TB) {
                    int a[TB]; double w[TB][18];
#pragma unroll
                    for (int u = 0; u < TB; u++) { const int sl = min(slot_of(k0 + u), e - 1); a[u] = L.tslot_pose[sl];
#pragma unroll
                        for (int k = 0; k < 18; k++) w[u][k] = B.w_tx[(size_t)sl*TX_REC + k]; }
#pragma unroll
                    for (int u = 0; u < TB; u++) if (slot_of(k0 + u) < e) tx_slot(a[u], w[u]);
                }
#pragma unroll
                for (int k = 0; k < 3; k++) { double t = acc[k];
#pragma unroll
                    for (int x = 1; x < BK_TT; x <<= 1) t += __shfl_xor(t, x, 64);              // the threads' sums
                    acc[k] = (POLL ? tq[k] : B.b_tx[(size_t)k*W.n_text + j]) + t; }
                double Vd[6], Vi[6], lam[3];
#pragma unroll
                for (int k = 0; k < 6; k++) Vd[k] = POLL ? tq[3 + k] : B.V_tx[(size_t)k*W.n_text + j];
#pragma unroll
                for (int k = 0; k < 3; k++) lam[k] = (POLL ? tq[9 + k] : B.dgs_tx[(size_t)k*W.n_text + j])*irad;
                Vd[0] += lam[0]; Vd[3] += lam[1]; Vd[5] += lam[2];
                if (inv_sym3(Vd, Vi)) {
                    d[0] = -(Vi[0]*acc[0] + Vi[1]*acc[1] + Vi[2]*acc[2]);
                    d[1] = -(Vi[1]*acc[0] + Vi[3]*acc[1] + Vi[4]*acc[2]);
                    d[2] = -(Vi[2]*acc[0] + Vi[4]*acc[1] + Vi[5]*acc[2]);
                    if (part == 0) for (int k = 0; k < 3; k++) { step2 += d[k]*d[k]; mcc += lam[k]*d[k]*d[k] - (POLL ? tq[k] : B.b_tx[(size_t)k*W.n_text + j])*d[k]; }
                }
            }
            if (part == 0) for (int k = 0; k < 3; k++) W.theta[cur ^ 1][3*j + k] = W.theta[cur][3*j + k] + d[k];
        }
    } else if (live) {
        int a = (b - nb_pt - nb_tx)*256 + tid;
        if (a < W.n_kf) {
            double xl[7];
#pragma unroll
            for (int k = 0; k < 7; k++) xl[k] = POLL ? tq[k] : W.pose[cur][7*a + k];
            const double *x = xl; double *c = W.pose[cur ^ 1] + 7*a;
            if (!fail && (POLL ? pfi : W.fidx[a]) >= 0) {
                double d[6];
#pragma unroll
                for (int k = 0; k < 6; k++) d[k] = dp[6*a + k];
                double q[4] = { x[0], x[1], x[2], x[3] }, qn[4];
                quat_plus(q, d, qn);
                for (int k = 0; k < 4; k++) { c[k] = qn[k]; step2 += (qn[k] - q[k])*(qn[k] - q[k]); }
                for (int k = 0; k < 3; k++) { c[4 + k] = x[4 + k] + d[3 + k]; step2 += d[3 + k]*d[3 + k]; }
                for (int k = 0; k < 6; k++) { const double lam = (POLL ? wq[0][k] : B.dgs_p[6*a + k])*irad; mcc += lam*d[k]*d[k] - (POLL ? wq[1][k] : B.bp[6*a + k])*d[k]; }
            } else for (int k = 0; k < 7; k++) c[k] = x[k];
        }
    }
    // the block's two partial sums: 256 values each, in block_sum<256>'s order (a block is four waves of its workgroup)
    red[tid] = step2; __syncthreads();
    if (tid < 64) { double sv = ((red[tid] + red[tid + 64]) + red[tid + 128]) + red[tid + 192]; sv = wave_sum1(sv); if (tid == 0 && live) W.partial[2*b] = sv; }
    __syncthreads();
    red[tid] = mcc; __syncthreads();
    if (tid < 64) { double sv = ((red[tid] + red[tid + 64]) + red[tid + 128]) + red[tid + 192]; sv = wave_sum1(sv); if (tid == 0 && live) W.partial[2*b + 1] = sv; }
}
__global__ __launch_bounds__(256) void k_back(Work W, LevelDev L, int nb_pt, int nb_tx) {
    __shared__ double red[256];
    back_body<false>(W, L, nb_pt, nb_tx, blockIdx.x, threadIdx.x, true, red, nullptr);
}
// ---- the reduced system of a small window and the back-substitution in ONE launch: workgroup 0 is k_solve_t, every other workgroup three blocks of k_back
// that wait for the pose step where k_back would wait for its launch (6.97 us per trial on C4, most of it the launch, the state round trip and the
// records' round trip behind it).  The step reaches the waiting blocks ~0.65 us after the solver stores it (tools/handover_bench.hip).
template <bool RL = true>
__global__ __launch_bounds__(SOLVE_THREADS) void k_solve_back(Work W, LevelDev L, int nb_pt, int nb_tx, int nb_all) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if (blockIdx.x != 0) {
        const int sub = threadIdx.x >> 8, b = 3*((int)blockIdx.x - 1) + sub;
        back_body<true>(W, L, nb_pt, nb_tx, b, threadIdx.x & 255, b < nb_all, smem + 256*sub, smem + 768);
        return; }
    solve_body<false, true, RL>(W, 0, smem);
}

// ---- the decision on one LM trial (Ceres 1.x TrustRegionMinimizer / LevenbergMarquardtStrategy semantics) on the state s -- the state in device memory
// (k_decide) or a workgroup's private copy of it (k_schur_t with the decision of the previous trial inside, tsba_kernels_schur.h).
// Returns the trace verdict: 1 accepted / 0 rejected / -1 invalid step / 2 tolerance exit on this trial.
template <class S>
__device__ __forceinline__ double lm_decide(S &s, double cost, double step2, double mcc, double gmax_c, double xn_c, const tsba_options &o) {
    double verdict = 0.0;
    mcc *= 0.5;                                   // model_cost_change = 1/2 dx^T (Lambda dx - g)
    s.it++;
    s.cand_cost = cost; s.model_change = mcc; s.step_norm = sqrt(step2);
    if (s.step_fail || !(mcc > 0.0)) {            // invalid step (LevenbergMarquardtStrategy::StepIsInvalid)
        s.step_fail = 0; verdict = -1.0;
        if (++s.invalid >= 5) { s.done = 1; s.term = 5; return verdict; }
        s.radius *= 0.5;
    } else {
        s.invalid = 0; s.n_cost++;
        if (!(cost == cost)) cost = 1.7976931348623157e308;
        if (s.step_norm <= o.parameter_tolerance*(s.x_norm + o.parameter_tolerance)) { s.done = 1; s.term = 2; return 2.0; }
        double cost_change = s.x_cost - cost;
        if (fabs(cost_change) <= o.function_tolerance*s.x_cost) { s.done = 1; s.term = 1; return 2.0; }
        double rel = cost_change/mcc;
        if (rel > o.min_relative_decrease) {      // accept: the speculative linearisation becomes the current one
            s.cur ^= 1; s.lcur ^= 1; s.accepted++; s.n_lin++;
            s.x_cost = cost; s.x_norm = sqrt(xn_c); s.gmax = gmax_c;
            double t = 2.0*rel - 1.0, f = 1.0 - t*t*t; if (f < 1.0/3.0) f = 1.0/3.0;
            s.radius = fmin(s.radius/f, o.max_radius);
            s.decrease_factor = 2.0; verdict = 1.0;
            if (gmax_c <= o.gradient_tolerance) { s.done = 1; s.term = 3; return verdict; }
        } else {
            s.radius = s.radius/s.decrease_factor; s.decrease_factor *= 2.0;
        }
    }
    if (s.it >= s.max_it) { s.done = 1; s.term = 0; }
    else if (s.radius < o.min_radius) { s.done = 1; s.term = 4; }
    return verdict;
}
// what a decision leaves for the host and the tests: the trace record of the trial (tsba_debug_lm_trace: 32 bytes) and the pinned progress word
__device__ __forceinline__ void lm_decide_publish(const Work &W, const LmState &s, double cost, double mcc_half, double verdict) {
    if (W.trace && s.it >= 1 && s.it <= TSBA_TRACE_CAP) {
        double *t = W.trace + 4*((size_t)W.trace_pass*TSBA_TRACE_CAP + s.it - 1);
        t[0] = verdict < 0.0 ? __longlong_as_double(0x7ff8000000000000LL) : (cost == cost ? cost : 1.7976931348623157e308); t[1] = mcc_half; t[2] = s.radius; t[3] = verdict; }
    if (W.hprog) { *W.hprog = ((unsigned long long)W.pass_seq << 32) | ((unsigned long long)s.it << 1) | (s.done ? 1u : 0u); __threadfence_system(); }
}
// ---- step quality and trust-region update (Ceres 1.x TrustRegionMinimizer / LevenbergMarquardtStrategy semantics)
__global__ __launch_bounds__(256) void k_decide(Work W, LevelDev L, int nb_back, int nb_lm, tsba_options o, int multi, int npp) {
    LmState *st = W.st;
    if (W.dp_poll) for (int k = threadIdx.x; k <= W.N; k += 256) W.dp[k] = __builtin_nan("");       // (k_solve_back: "not there yet" for the blocks that poll the next step)
    if (st->done) return;
    __shared__ double red[5*256], xch[256];
    const int tid = threadIdx.x;
    // the candidate was linearised speculatively into lb[lcur^1]: its cost, gradient and diagonals are already there
    const LinBuf &Bc = W.lb[st->lcur ^ 1];
    double gmax_c, xn_c, cost;
    double step2 = 0.0, mcc = 0.0;
#ifdef TSBA_SOLVE_STAMPS
    const long long s0_ = clock64(); long long s1_ = 0, s2_ = 0, s3_ = 0;
#endif
    if (!multi) {
        double o5[5];
        postlin_fused(W, L, Bc, W.pose[st->cur ^ 1], false, nb_lm, nb_back, red, xch, o5, npp);
        gmax_c = o5[0]; xn_c = o5[1]; cost = o5[2]; step2 = o5[3]; mcc = o5[4];
#ifdef TSBA_SOLVE_STAMPS
        s1_ = s2_ = s3_ = clock64();
#endif
    } else {                                      // k_sums_multi + all-reduce already produced the global sums
        double gp, xp;
        if (npp) pose_parts_multi(W, npp, red, gp, xp); else pose_scale(W, Bc, W.cb, W.cb + W.N, W.pose[st->cur ^ 1], false, red, gp, xp);
        const double *sc = W.cb + 2*(size_t)W.N;
        cost = sc[0]; xn_c = sc[1] + xp; step2 = sc[2]; mcc = sc[3]; gmax_c = fmax(W.cbm[0], gp);
    }
    if (tid) return;
    st->lin_done = 0;                             // (the iterative reduced-system solve of this trial is over: tsba_pcg.h)
    const double verdict = lm_decide(*st, cost, step2, mcc, gmax_c, xn_c, o);
    lm_decide_publish(W, *st, cost, st->model_change, verdict);
#ifdef TSBA_SOLVE_STAMPS
    W.dbg[32] = s1_ - s0_; W.dbg[33] = s2_ - s1_; W.dbg[34] = s3_ - s2_; W.dbg[35] = clock64() - s3_;
#endif
}

// ================================================================== multi-GPU (global BA sharded by landmark over RCCL)
// stage A: local sums into the all-reduce buffer hb = [Hd | bp | scal] and gm.  spec: candidate LinBuf (also folds the
// k_back partial sums: landmark blocks are owned by exactly one rank, the replicated pose blocks count on rank 0 only)
__global__ __launch_bounds__(256) void k_sums_multi(Work W, LevelDev L, int spec, int nb_lm, int nb_back, int nb_back_lm, int skip_pose) {
    LmState *st = W.st;
    if (st->done) return;
    if (!spec && !st->need_lin) return;
    __shared__ double red[256];
    const LinBuf &B = W.lb[spec ? (st->lcur ^ 1) : st->lcur];
    double gl, xl, cost;
    sums_local(W, L, B, W.cb, W.cb + W.N, nb_lm, red, gl, xl, cost, skip_pose != 0);
    double step2 = 0.0, mcc = 0.0;
    if (spec) for (int k = threadIdx.x; k < nb_back; k += 256)
        if (k < nb_back_lm || W.rank == 0) { step2 += W.partial[2*k]; mcc += W.partial[2*k + 1]; }
    step2 = block_sum<256>(step2, red); mcc = block_sum<256>(mcc, red);
    if (threadIdx.x == 0) { double *sc = W.cb + 2*(size_t)W.N; sc[0] = cost; sc[1] = xl; sc[2] = step2; sc[3] = mcc; W.cbm[0] = gl; }
}
// reduced camera system: add the pose damping once, after the all-reduce of the partial S
__global__ void k_damp_multi(Work W) {
    LmState *st = W.st;
    if (st->done) return;
    const LinBuf &B = W.lb[st->lcur];
    const double irad = 1.0/st->radius;
    int a = blockIdx.x*blockDim.x + threadIdx.x;
    if (a >= W.n_kf) return;
    int ia = W.fidx[a]; if (ia < 0) return;
    for (int k = 0; k < 6; k++) W.S[(size_t)(6*ia + k)*W.ldS + 6*ia + k] += B.dgs_p[6*a + k]*irad;
}
// Band storage keeps every row of S in a skewed window of LDB = band + 2 x 96 - 1 doubles (room for the wide-band Cholesky's diagonal
// blocks): 251 columns at a band of 60 rows, of which a row holds at most band + 6 entries of the lower triangle.  The ranks exchange
// only those: pack -> one all-reduce of N (band + 6) doubles (15.8 MB instead of 60 MB at 5000 keyframes) -> unpack.
__global__ __launch_bounds__(256) void k_band_pack(Work W, double *buf, int wp, int unpack) {
    const LmState *st = W.st;
    if (st->done) return;
    const int n = 6*(*W.nfree) + (W.ring ? wp - 6 : 0);          // (ring: + the ghost rows behind the last free pose, wp = band + 6)
    const long long tot = (long long)n*wp;
    const size_t ldS = (size_t)W.ldS;
    for (long long e = (long long)blockIdx.x*256 + threadIdx.x; e < tot; e += (long long)gridDim.x*256) {
        const int i = (int)(e/wp), k = (int)(e - (long long)i*wp), c = i - wp + 1 + k;     // row i, columns i - wp + 1 .. i
        if (c < 0) { if (!unpack) buf[e] = 0.0; continue; }
        if (unpack) W.S[(size_t)i*ldS + c] = buf[e]; else buf[e] = W.S[(size_t)i*ldS + c];
    }
}
// landmark parameters live on their owner: delta = x - x0 on the owner, 0 elsewhere (all-reduced, then x = x0 + delta)
__global__ void k_delta_multi(Work W, const double *rho0, const double *theta0, int apply) {
    const int cur = W.st->cur;
    int j = blockIdx.x*blockDim.x + threadIdx.x;
    if (j < W.n_pt) {
        if (!apply) W.dl_pt[j] = (W.pt_host[j] >= 0 && tsba_shard_of(W.pt_host[j], 0, W.n_kf, W.world) == W.rank) ? W.rho[cur][j] - rho0[j] : 0.0;   // (a frozen landmark does not move)
        else W.rho[cur][j] = rho0[j] + W.dl_pt[j];
    } else if (j < W.n_pt + 3*W.n_text) {
        int k = j - W.n_pt, t = k/3;
        if (!apply) W.dl_tx[k] = (W.text_host[t] >= 0 && tsba_shard_of(W.text_host[t], 0, W.n_kf, W.world) == W.rank) ? W.theta[cur][k] - theta0[k] : 0.0;
        else W.theta[cur][k] = theta0[k] + W.dl_tx[k];
    }
}
__global__ void k_kfin_multi(Work W) {               // kf_in was summed over ranks: back to a flag
    int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k < W.n_kf) W.kf_in[k] = W.kf_in[k] != 0;
}

// ---- outlier pass on loss-corrected residuals, optimizer.cc:1609-1686 / :1228-1305
// pfin != nullptr (pose-only path): the pass's result still lives in the PoseState -- pose from there, and one extra workgroup
// installs it into W.st / W.pose (field by field: the counters of this very kernel are being updated by atomics)
// (one wave of 64 lanes per block b: k_outlier launches a workgroup per wave, k_pass_end -- tsba_kernels_pass.h -- four waves per workgroup next to
// other roles)
__device__ __forceinline__ void outlier_wave(const Work &W, const LevelDev &L, const int b, const int lane, double chi2_mono, double chi2_text, double bad_ratio,
                                             int do_scene, int do_text, const PoseState *pfin) {
    LmState *st = W.st;
    const int selc = pfin ? 0 : st->cur;
    const double *pose = pfin ? pfin->x : W.pose[selc], *rho = W.rho[selc], *theta = W.theta[selc];
    if (st->nt_active < 50) chi2_mono += 4.0;
    const int nb_sc = (L.n_sc + 63) >> 6;
    if (b < nb_sc) {
        // scene: one candidate per lane (a frame's single (target, frozen host) pair would otherwise be one wave's serial loop)
        if (!do_scene) return;
        const int c = b*64 + lane;
        int nbad = 0;
        if (c < L.n_sc && (!W.filter_good || W.sgood[L.sc_flag[c]])) {
            const int pt = L.sc_pt[c], i = L.sc_kf[c], h = W.pt_host[pt];
            Pose C; load_pose(pose + 7*i, C);
            PairT T;
            if (h >= 0) { Pose Hs; load_pose(pose + 7*h, Hs); pair_from_poses(C, Hs, T); }
            else pair_from_Trw(C, W.pt_Trw + 12*(size_t)pt, T);
            double r[2];
            scene_residual(T, C.t, W.pt_ray[2*pt], W.pt_ray[2*pt+1], rho[pt], L.sc_uv[2*c], L.sc_uv[2*c+1],
                           W.K0[0], W.K0[1], W.K0[2], W.K0[3], W.w_sx, W.w_sy, r);
            double wgt; huber(r[0]*r[0] + r[1]*r[1], W.huber_s, wgt);
            double sc = sqrt(wgt);
            double ex = r[0]*sc/W.w_sx, ey = r[1]*sc/W.w_sy;
            if (ex*ex > chi2_mono || ey*ey > chi2_mono) { W.sgood[L.sc_flag[c]] = 0; nbad++; }
        }
        nbad = (int)wave_sum1((double)nbad);
        if (lane == 0 && nbad) atomicAdd(&st->n_bad_scene, nbad);
    } else if (b < nb_sc + L.n_tg) {
        // text: one (KF, text) observation per workgroup, lane = (feature lane >> 3, tap lane & 7), 8 features per round
        if (!do_text) return;
        const int g = b - nb_sc;
        const int tb = L.tg_tobs[g], i = L.tg_kf[g], j = L.tg_text[g], h = W.text_host[j];
        if (W.filter_good && !W.tobs_good[tb]) return;
        const double mu = W.musig[2*tb], sigma = W.musig[2*tb+1];
        Pose C; load_pose(pose + 7*i, C);
        PairT T;
        if (h >= 0) { Pose Hs; load_pose(pose + 7*h, Hs); pair_from_poses(C, Hs, T); }
        else pair_from_Twr(C, W.text_Twr + 12*(size_t)j, T);
        const double th[3] = { theta[3*j], theta[3*j+1], theta[3*j+2] };
        const uint8_t *img = L.img[i];
        const int fg = W.tobs_fgood_off[tb];
        const int k = lane & 7, f0 = L.tfeat_off[j], f1 = L.tfeat_off[j+1];
        int nblk = 0, nbad = 0;
        for (int fb = f0; fb < f1; fb += 8) {                          // (uniform trip count: the shuffles need all 8 lanes of a feature)
            const int f = fb + (lane >> 3);
            const bool in = f < f1 && (!W.filter_good || W.tfgood[fg + L.tfeat_raw[min(f, f1 - 1)]]);
            double r = 0.0;
            if (in && sigma != 0.0) {                                   // sigma == 0: residuals are 0, never an outlier
                const double fu = L.tfeat_uv[2*f], fv = L.tfeat_uv[2*f+1];
                double jt[6], jl[3];
                double mx = (fu + TAP_DX[k] - L.K[2])/L.K[0], my = (fv + TAP_DY[k] - L.K[3])/L.K[1];
                r = text_tap(T, C.t, th, mx, my, L.K[0], L.K[1], L.K[2], L.K[3], img, L.img_w, L.img_h, mu, sigma, 1.0/sigma,
                             L.tfeat_ref[8*(size_t)f + k], W.w_t, false, jt, jl);
            }
            double s = r*r;
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            double wgt; huber(s, W.huber_t, wgt);
            const double sc = sqrt(wgt);
            int bad = (in && sigma != 0.0 && fabs(r*sc/W.w_t) > chi2_text) ? 1 : 0;
            bad |= __shfl_xor(bad, 1, 64); bad |= __shfl_xor(bad, 2, 64); bad |= __shfl_xor(bad, 4, 64);
            if (k == 0 && in) { nblk++; if (bad) { W.tfgood[fg + L.tfeat_raw[f]] = 0; nbad++; } }
        }
        nblk = (int)wave_sum1((double)nblk); nbad = (int)wave_sum1((double)nbad);
        if (lane == 0 && nblk > 0) {
            if (nbad) atomicAdd(&st->n_bad_tfeat, nbad);
            if ((double)nbad/(double)nblk > bad_ratio) { W.tobs_good[tb] = 0; atomicAdd(&st->n_bad_text, 1); }
        }
    } else if (pfin) {
        if (lane < 7) { W.pose[0][lane] = pfin->x[lane]; W.pose[1][lane] = pfin->x[lane]; }
        if (lane == 0) {
            const LmState &S = pfin->S;
            st->radius = S.radius; st->decrease_factor = S.decrease_factor; st->x_cost = S.x_cost; st->x_norm = S.x_norm;
            st->cand_cost = S.cand_cost; st->model_change = S.model_change; st->step_norm = S.step_norm; st->gmax = S.gmax; st->cost0 = S.cost0;
            st->done = S.done; st->need_lin = S.need_lin; st->first = S.first; st->it = S.it; st->accepted = S.accepted;
            st->term = S.term; st->invalid = S.invalid; st->max_it = S.max_it; st->step_fail = S.step_fail; st->lcur = S.lcur;
            st->n_lin = S.n_lin; st->n_cost = S.n_cost;
        }
    }
}

__global__ __launch_bounds__(64) void k_outlier(Work W, LevelDev L, double chi2_mono, double chi2_text, double bad_ratio,
                                                int do_scene, int do_text, const PoseState *pfin) {
    outlier_wave(W, L, blockIdx.x, threadIdx.x, chi2_mono, chi2_text, bad_ratio, do_scene, do_text, pfin);
}

// ---- information matrix V (6 values) of one text plane at the end of a pass: ceres::Covariance runs after every pyramid pass of
// PyrThetaOptim and the last successful one is kept (optimizer.cc:2219-2238)
__global__ void k_record_vtx(Work W, int text, double *out6) {
    const LinBuf &B = W.lb[W.st->lcur];
    if (threadIdx.x < 6 && blockIdx.x == 0) out6[threadIdx.x] = B.V_tx[(size_t)threadIdx.x*W.n_text + text];
}
// ---- test hook: explicit residuals and Jacobians of every block, written at the reference's block order
__global__ void k_eval_scene(Work W, LevelDev L, const int *out_idx, double *resid, double *jac) {
    int c = blockIdx.x*blockDim.x + threadIdx.x; if (c >= L.n_sc) return;
    int oi = out_idx[c]; if (oi < 0) return;
    const double *pose = W.pose[0], *rho = W.rho[0];
    // pair of this candidate
    int i = L.sc_kf[c], pt = L.sc_pt[c], h = W.pt_host[pt];
    Pose C; load_pose(pose + 7*i, C);
    PairT T;
    if (h >= 0) { Pose Hs; load_pose(pose + 7*h, Hs); pair_from_poses(C, Hs, T); } else pair_from_Trw(C, W.pt_Trw + 12*(size_t)pt, T);
    double r[2], jt[2][6], jl[2];
    scene_block(T, C.t, W.pt_ray[2*pt], W.pt_ray[2*pt+1], rho[pt], L.sc_uv[2*c], L.sc_uv[2*c+1], W.K0[0], W.K0[1], W.K0[2], W.K0[3], W.w_sx, W.w_sy, r, jt, jl);
    resid[2*oi] = r[0]; resid[2*oi+1] = r[1];
    if (jac) for (int k = 0; k < 2; k++) {
        double *row = jac + (size_t)oi*26 + k*13;
        for (int a = 0; a < 6; a++) row[a] = jt[k][a];
        if (h >= 0) {
            for (int cc = 0; cc < 3; cc++) {
                row[6 + cc] = -(jt[k][0]*T.Rcr[cc] + jt[k][1]*T.Rcr[3 + cc] + jt[k][2]*T.Rcr[6 + cc]);
                row[9 + cc] = -(jt[k][3]*T.Rcr[cc] + jt[k][4]*T.Rcr[3 + cc] + jt[k][5]*T.Rcr[6 + cc]);
            }
            row[12] = jl[k];
        } else for (int a = 6; a < 13; a++) row[a] = 0.0;
    }
}
__global__ void k_eval_text(Work W, LevelDev L, int nblk, const int *blk_g, const int *blk_f, int ns, double *resid, double *jac) {
    int q = blockIdx.x*blockDim.x + threadIdx.x; if (q >= nblk) return;
    int g = blk_g[q], f = blk_f[q];
    const double *pose = W.pose[0], *theta = W.theta[0];
    const int tb = L.tg_tobs[g], i = L.tg_kf[g], j = L.tg_text[g], h = W.text_host[j];
    const double mu = W.musig[2*tb], sigma = W.musig[2*tb+1];
    Pose C; load_pose(pose + 7*i, C);
    PairT T;
    if (h >= 0) { Pose Hs; load_pose(pose + 7*h, Hs); pair_from_poses(C, Hs, T); } else pair_from_Twr(C, W.text_Twr + 12*(size_t)j, T);
    const double th[3] = { theta[3*j], theta[3*j+1], theta[3*j+2] };
    const double fu = L.tfeat_uv[2*f], fv = L.tfeat_uv[2*f+1];
    double *rout = resid + 2*(size_t)ns + 8*(size_t)q;
    double *jout = jac ? jac + 26*(size_t)ns + 120*(size_t)q : nullptr;
    for (int k = 0; k < 8; k++) {
        double jt[6] = {0,0,0,0,0,0}, jl[3] = {0,0,0}, r = 0.0;
        if (sigma != 0.0) {
            double mx = (fu + TAP_DX[k] - L.K[2])/L.K[0], my = (fv + TAP_DY[k] - L.K[3])/L.K[1];
            r = text_tap(T, C.t, th, mx, my, L.K[0], L.K[1], L.K[2], L.K[3], L.img[i], L.img_w, L.img_h, mu, sigma, 1.0/sigma,
                         L.tfeat_ref[8*(size_t)f + k], W.w_t, true, jt, jl);
        }
        rout[k] = r;
        if (jout) {
            double *row = jout + k*15;
            for (int a = 0; a < 6; a++) row[a] = jt[a];
            if (h >= 0) {
                for (int cc = 0; cc < 3; cc++) {
                    row[6 + cc] = -(jt[0]*T.Rcr[cc] + jt[1]*T.Rcr[3 + cc] + jt[2]*T.Rcr[6 + cc]);
                    row[9 + cc] = -(jt[3]*T.Rcr[cc] + jt[4]*T.Rcr[3 + cc] + jt[5]*T.Rcr[6 + cc]);
                }
                row[12] = jl[0]; row[13] = jl[1]; row[14] = jl[2];
            } else for (int a = 6; a < 15; a++) row[a] = 0.0;
        }
    }
}

