// Assembly of the reduced camera system: k_schur_t, k_schur_quad.  (part of the single translation unit tsba.hip: included there, in this order)
#pragma once
// ---- reduced camera system.  grid = n_sb (one workgroup per 6x6 block) + n_kf (reduced gradient), 256 threads.
// The (slot, slot, landmark) gather lists of a diagonal block hold ~1000 entries: four waves, and per wave the indices and
// operands of four entries in flight before the first multiply (two dependent global round trips per 1024 entries).
#define SCHUR_U 4
#define SCHUR_KEEP_KF 32                    // keyframes whose damping / gradient rows a workgroup of k_schur_t<4> keeps in LDS when it takes the previous trial's decision itself
// SCHUR_NW waves per workgroup: 4 for windows (a diagonal block gathers ~1000 slot pairs), 1 for large maps (54 k blocks of ~60 slot
// pairs each at 5000 keyframes: three idle waves per block and their hand-off were most of the 0.64 ms)
// dec (windows on one GPU, SCHUR_NW = 4: 256 threads as k_decide): the launch takes the DECISION on the previous trial first -- every workgroup,
// redundantly and bit-identically, sums what the candidate's linearisation left (postlin_fused: cost, step, model cost change, the poses' diagonal and
// gradient rows), applies the Ceres rules to its private copy of the state and goes on with the outcome; workgroup 0 also stores the rows and writes the
// new state to W.st_next (the host passes that copy as W.st from the next launch on).  k_decide as a launch of its own was 9.0 us + a gap per trial;
// inside this launch its round trip runs next to the list fetches of the assembly.
struct SchurDec { int on, nb_back, nb_lm; tsba_options o; };
template <class S> __device__ __forceinline__ double lm_decide(S &s, double cost, double step2, double mcc, double gmax_c, double xn_c, const tsba_options &o);
__device__ __forceinline__ void lm_decide_publish(const Work &W, const LmState &s, double cost, double mcc_half, double verdict);
__device__ __forceinline__ void lm_state_store(LmState *n, const LmState &s) {       // every field but step_fail
    n->radius = s.radius; n->decrease_factor = s.decrease_factor; n->x_cost = s.x_cost; n->x_norm = s.x_norm; n->cand_cost = s.cand_cost;
    n->model_change = s.model_change; n->step_norm = s.step_norm; n->gmax = s.gmax; n->cost0 = s.cost0;
    n->cur = s.cur; n->done = s.done; n->need_lin = s.need_lin; n->first = s.first; n->it = s.it; n->accepted = s.accepted; n->term = s.term;
    n->invalid = s.invalid; n->max_it = s.max_it; n->lcur = s.lcur; n->lin_done = s.lin_done;
    n->ns_active = s.ns_active; n->nt_active = s.nt_active; n->n_bad_scene = s.n_bad_scene; n->n_bad_tfeat = s.n_bad_tfeat; n->n_bad_text = s.n_bad_text; n->pad2 = s.pad2;
    n->n_lin = s.n_lin; n->n_cost = s.n_cost;
}
template <int SCHUR_NW>
__global__ __launch_bounds__(64*SCHUR_NW) void k_schur_t(Work W, LevelDev L, int multi, int b0, SchurDec dec) {
    constexpr int SCHUR_T = 64*SCHUR_NW;
#ifdef MID_STAMPS                           // (make-time experiment, tools/mid_stamps.sh: cycles of a workgroup by kind -- diagonal S block, off-diagonal S block, gradient -- into W.dbg[40..63])
    const long long ss_t0 = clock64(); int ss_kind = 2;      // (kind 2's slots are shared with k_musigma's stamps: not recorded)
#define SCHUR_STAMP(slot) do { if (threadIdx.x == 0 && ss_kind < 1) atomicAdd((unsigned long long *)&W.dbg[40 + 8*ss_kind + (slot)], (unsigned long long)(clock64() - ss_t0)); } while (0)
#else
#define SCHUR_STAMP(slot) do { } while (0)
#endif
    // Round 6 (stamps, tools/mid_stamps.sh: a diagonal block's workgroup spent 18 k cycles on the decision and then walked NINE dependent round trips -- block ->
    // offsets / rows -> slot-pair indices -> records, the same again for the planes' slot pairs, offsets -> ranges for the tail): everything that does not depend
    // on the decision (the static lists of the plan, the rows of the free poses) is requested BEFORE it -- the block's own entries with the LM state, what hangs
    // off them with the decision's first loads -- and is there when the decision is.
    struct Pre { int a, c, ia, ic, pt0, pt1, tx0, tx1, pab, pba, t0, t1, h0, h1, ps0, ps1, ts0, ts1; int s1[2*SCHUR_U], s2[2*SCHUR_U], j[2*SCHUR_U], xs1, xs2, xj; } pre;      // (two rounds of a diagonal block's slot pairs: ~1100 entries on 256 threads)
    constexpr bool PRE = SCHUR_NW == 4;
    const int pre_kind = !PRE || b0 != 0 ? 0 : ((int)blockIdx.x < L.n_sb ? 1 : ((int)blockIdx.x - L.n_sb < W.n_kf ? 2 : 0));
    if constexpr (PRE) {
        const int b = (int)blockIdx.x;                             // (windows: b0 == 0, no XCD remapping)
        if (pre_kind == 1) {
            pre.a = L.sb_a[b]; pre.c = L.sb_b[b]; pre.pt0 = L.sb_pt_off[b]; pre.pt1 = L.sb_pt_off[b+1]; pre.tx0 = L.sb_tx_off[b]; pre.tx1 = L.sb_tx_off[b+1];
            pre.pab = L.sb_pab[b]; pre.pba = L.sb_pba[b];
            pre.t0 = L.sb_rng[4*b]; pre.t1 = L.sb_rng[4*b + 1]; pre.h0 = L.sb_rng[4*b + 2]; pre.h1 = L.sb_rng[4*b + 3];
        } else if (pre_kind == 2) {
            pre.a = b - L.n_sb; pre.ia = W.fidx[pre.a];
            pre.ps0 = L.pose_ps_off[pre.a]; pre.ps1 = L.pose_ps_off[pre.a+1]; pre.ts0 = L.pose_ts_off[pre.a]; pre.ts1 = L.pose_ts_off[pre.a+1];
            pre.t0 = L.pose_t_off[pre.a]; pre.t1 = L.pose_t_off[pre.a+1]; pre.h0 = L.pose_h_off[pre.a]; pre.h1 = L.pose_h_off[pre.a+1];
        }
    }
    LmState *st = W.st;
    if (st->done) {                                             // (a finished pass: the launches the host still had in flight -- the other copy of the state has to say so too)
        if (SCHUR_NW == 4 && dec.on && blockIdx.x == 0 && threadIdx.x == 0) { const LmState s = *st; lm_state_store(W.st_next, s); }
        return; }
    __shared__ double lds[SCHUR_NW > 1 ? 3*36*64 : 36*65];     // waves 1..3 hand their partial blocks to wave 0, which then transposes (36*65 <= 3*36*64)
    __shared__ double dsh_r; __shared__ int dsh_i[3];
    __shared__ double tls[SCHUR_NW == 4 ? 21*65 : 1];          // a diagonal block's tail: the 21 distinct entries x (target range | host range), see below
    int lcur_ = st->lcur; double radius_ = st->radius; bool fresh = false;      // fresh: the current linearisation is the candidate this launch has just accepted
    if constexpr (PRE) {                                            // (second stage: in flight together with the decision's first loads)
        const int tid = threadIdx.x;
        if (pre_kind == 1) {
            pre.ia = W.fidx[pre.a]; pre.ic = W.fidx[pre.c];
#pragma unroll
            for (int u = 0; u < 2*SCHUR_U; u++) { const int qc = max(min(pre.pt0 + u*SCHUR_T + tid, pre.pt1 - 1), 0);
                pre.s1[u] = L.sb_pt_s1[qc]; pre.s2[u] = L.sb_pt_s2[qc]; pre.j[u] = L.sb_pt_lm[qc]; }
            { const int qc = max(min(pre.tx0 + tid, pre.tx1 - 1), 0); pre.xs1 = L.sb_tx_s1[qc]; pre.xs2 = L.sb_tx_s2[qc]; pre.xj = L.sb_tx_lm[qc]; }
        } else if (pre_kind == 2) {
#pragma unroll
            for (int u = 0; u < 2*SCHUR_U; u++) { const int qc = max(min(pre.ps0 + u*SCHUR_T + tid, pre.ps1 - 1), 0); pre.s1[u] = L.pose_ps[qc]; pre.j[u] = L.pose_ps_lm[qc]; }
            { const int qc = max(min(pre.ts0 + tid, pre.ts1 - 1), 0); pre.xs1 = L.pose_ts[qc]; pre.xj = L.pose_ts_lm[qc]; }
        }
    }
    const bool use_pre = PRE && b0 == 0;
    bool first_ = false;                                        // dec.on == 2: this is the pass's first linearisation (the Jacobi scales are being fixed by it)
    if constexpr (SCHUR_NW == 4) { if (dec.on == 2) {
        // The first trial of a pass (round 6): no decision to take -- workgroup 0 does what k_postlin did as a launch of its own on the first linearisation
        // (the poses' rows and Jacobi scales, cost, gradient test, the state into the other copy); every workgroup forms the rows it needs of its own pose itself
        first_ = st->first != 0;
        if (blockIdx.x == 0) {
            double o5[5];
            const bool lin = st->need_lin != 0;
            if (lin) postlin_fused(W, L, W.lb[lcur_], W.pose[st->cur], first_, dec.nb_lm, 0, lds, lds + 5*256, o5, 0, true, nullptr, false);
            if (W.dp_poll) for (int k = threadIdx.x; k <= W.N; k += SCHUR_T) W.dp[k] = __builtin_nan("");     // (k_solve_back: "not there yet")
            if (threadIdx.x == 0) {
                LmState s = *st;
                if (lin) {
                    s.x_cost = o5[2]; s.x_norm = sqrt(o5[1]); s.gmax = o5[0];
                    if (s.first) s.cost0 = o5[2];
                    s.first = 0; s.need_lin = 0; s.n_lin++;
                    if (s.gmax <= dec.o.gradient_tolerance) { s.done = 1; s.term = 3; }
                }
                lm_state_store(W.st_next, s);
                if (W.hprog) { *W.hprog = ((unsigned long long)W.pass_seq << 32) | ((unsigned long long)s.it << 1) | (s.done ? 1u : 0u); __threadfence_system(); }
                dsh_i[0] = s.done;
            }
            __syncthreads();
            if (dsh_i[0]) return;
        }
        fresh = true;
        st = W.st_next;                                         // (where this trial's failure flag lives)
    } else if (dec.on) {
        // Round 6: workgroup 0 takes the FULL decision (and stores the state, the poses' rows, the trace); every other workgroup takes it from the partials of
        // cost / step / model cost change alone -- the same sums in the same order, so accept / reject, the trust region and the buffer that becomes current
        // come out the same bits; what it skips (the poses' gradient max and |x|^2 over 75 KB of pair products that 230 workgroups were all pulling through the
        // L2 at once: 18 - 24 k cycles of a 38 k-cycle kernel, tools/mid_stamps.sh) only decides the gradient-tolerance exit, which workgroup 0 keeps -- a
        // workgroup that misses it assembles a block nobody reads.  The rows a workgroup needs of the accepted candidate (its pose's damping / gradient) it
        // forms itself from the ranges it reads anyway, with the arithmetic of postlin_fused.
        double o5[5];
        const bool full = blockIdx.x == 0;
        // (Measured and dropped: the block's tail -- pose-pair products / gradient row -- formed for BOTH outcomes before the decision, in flight with its partials.
        // A diagonal block's 36 lanes read 36 different rows of the pair products with every request: 2 x 48 requests x 36 cache lines through one compute
        // unit's vector cache cost 9 k cycles in front of the decision against the 4.5 k they take behind the gather: 15.9 against 14.0 us.)
        postlin_fused(W, L, W.lb[lcur_ ^ 1], W.pose[st->cur ^ 1], false, dec.nb_lm, dec.nb_back, lds, lds + 5*256, o5, 0, full, nullptr, !full);
        if (blockIdx.x == 0 && W.dp_poll) for (int k = threadIdx.x; k <= W.N; k += SCHUR_T) W.dp[k] = __builtin_nan("");     // (k_solve_back: "not there yet")
        if (threadIdx.x == 0) {
            LmState s = *st;
            s.lin_done = 0;
            const double verdict = lm_decide(s, o5[2], o5[3], o5[4], full ? o5[0] : __builtin_inf(), full ? o5[1] : 0.0, dec.o);      // (no gradient-tolerance exit from the light decision)
            dsh_i[0] = s.done; dsh_i[1] = s.lcur; dsh_i[2] = s.lcur != lcur_; dsh_r = s.radius;
            if (blockIdx.x == 0) {                               // the new state, every field but step_fail (zero since k_mid of the last trial; the workgroups of this launch may raise it)
                lm_state_store(W.st_next, s);
                lm_decide_publish(W, s, o5[2], s.model_change, verdict);
            }
        }
        __syncthreads();
        if (dsh_i[0]) return;
        lcur_ = dsh_i[1]; radius_ = dsh_r; fresh = dsh_i[2] != 0;
        st = W.st_next;                                         // (where this trial's failure flag lives)
    } }
    // (b0 > 0: large maps take the S blocks through k_schur_quad and only the gradient part here, a grid of a multiple of 8 workgroups in
    // which the workgroups of ONE XCD -- workgroup i runs on XCD i mod 8 -- take neighbouring poses: the slot records of a landmark sit next
    // to each other, one per observing pose, and neighbouring poses observe the same landmarks)
    const int bx = b0 > 0 ? ((int)blockIdx.x & 7)*((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int b = bx + b0, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef MID_STAMPS
    ss_kind = b < L.n_sb ? (L.sb_a[b] == L.sb_b[b] ? 0 : 1) : 2;
#endif
    SCHUR_STAMP(0);                                             // (the decision on the previous trial is taken)
    const double radius = radius_, irad = 1.0/radius;
    const LinBuf &B = W.lb[lcur_];
    if (b < L.n_sb) {
        const int a = use_pre ? pre.a : L.sb_a[b], c = use_pre ? pre.c : L.sb_b[b];
        const int ia = use_pre ? pre.ia : W.fidx[a], ic = use_pre ? pre.ic : W.fidx[c];           // rows / columns of S exist for free poses only
        if (ia < 0 || ic < 0) return;
        double acc[36];
#pragma unroll
        for (int k = 0; k < 36; k++) acc[k] = 0.0;
        const int pt0 = use_pre ? pre.pt0 : L.sb_pt_off[b], pt1 = use_pre ? pre.pt1 : L.sb_pt_off[b+1];
        // Round 6: the TAIL's operands are requested here, in flight with the gather, laid out so that a request touches a few cache lines instead of 36: a
        // diagonal block's 36 lanes used to read 36 different rows of the pair products with every one of 48 requests (1700 line accesses through one
        // vector cache, 4.5 k cycles behind the gather).  Now lane q of waves 1 - 3 reads element q of the pose's target range (lanes 0 - 31) or host range
        // (32 - 63) of seven of the 21 distinct rows each -- contiguous: 2 + 2 lines per request --, the values cross through LDS, and lane (r, c) of wave 0
        // adds its row up in index order: the same sums in the same order as range_sum<24>(target) + range_sum<24>(host).
        bool tq = false; double tl[7] = {0,0,0,0,0,0,0}, tsg = 1.0, tdg = 0.0, to0 = 0.0, to1 = 0.0; bool t_on = false;
        if constexpr (SCHUR_NW == 4) if (use_pre && pre.t1 - pre.t0 <= 32 && pre.h1 - pre.h0 <= 32) {      // (uniform: windows of at most 32 keyframes)
            tq = true;
            const double *out = B.pairOut;
            if (a == c) {
                if (wave > 0) {
                    const bool ht = lane < 32; const int pos = ht ? lane : lane - 32;
                    t_on = ht ? pos < pre.t1 - pre.t0 : pos < pre.h1 - pre.h0;
                    const size_t q = t_on ? (size_t)((ht ? pre.t0 : pre.h0) + pos) : 0;
#pragma unroll
                    for (int i = 0; i < 7; i++) tl[i] = out[(size_t)((ht ? 0 : 63) + 7*(wave - 1) + i)*L.n_pair + q];
                } else if (lane < 36 && lane/6 == lane % 6) { tsg = W.sig_p[6*a + lane/6]; tdg = B.dgs_p[6*a + lane/6]; }
            } else if (wave == 0 && lane < 36) {
                const int r = lane/6, cc = lane % 6;
                to0 = out[(size_t)(27 + r*6 + cc)*L.n_pair + max(pre.pab, 0)]; to1 = out[(size_t)(27 + cc*6 + r)*L.n_pair + max(pre.pba, 0)];
            }
        }
        for (int base = pt0; base < pt1; base += SCHUR_T*SCHUR_U) {
            int s1[SCHUR_U], s2[SCHUR_U], j[SCHUR_U]; bool ok[SCHUR_U];
#pragma unroll
            for (int u = 0; u < SCHUR_U; u++) {
                const int q = base + u*SCHUR_T + tid; ok[u] = q < pt1;
                const int qc = min(q, pt1 - 1);
                if (use_pre && base == pt0) { s1[u] = pre.s1[u]; s2[u] = pre.s2[u]; j[u] = pre.j[u]; }
                else if (use_pre && base == pt0 + SCHUR_T*SCHUR_U) { s1[u] = pre.s1[SCHUR_U + u]; s2[u] = pre.s2[SCHUR_U + u]; j[u] = pre.j[SCHUR_U + u]; }
                else { s1[u] = L.sb_pt_s1[qc]; s2[u] = L.sb_pt_s2[qc]; j[u] = L.sb_pt_lm[qc]; }
            }
            double w1[SCHUR_U][6], w2[SCHUR_U][6], Vv[SCHUR_U], dg[SCHUR_U];
            if (SCHUR_NW == 4 && a == c) {                      // a diagonal block: both slots of an entry are the landmark's ONE slot at this pose -- one record, not
                                                                // two requests for it (a third of the block's cache-line accesses; a second slot at the same pose is read on its own)
#pragma unroll
                for (int u = 0; u < SCHUR_U; u++) {
                    VDB_LOAD(B, j[u], W.n_pt, Vv[u], dg[u]);
#pragma unroll
                    for (int k = 0; k < 6; k++) w1[u][k] = B.w_pt[(size_t)(s1[u])*PT_REC + k];
                }
#pragma unroll
                for (int u = 0; u < SCHUR_U; u++) {
                    if (s2[u] != s1[u]) {
#pragma unroll
                        for (int k = 0; k < 6; k++) w2[u][k] = B.w_pt[(size_t)(s2[u])*PT_REC + k];
                    } else {
#pragma unroll
                        for (int k = 0; k < 6; k++) w2[u][k] = w1[u][k];
                    }
                }
            } else {
#pragma unroll
            for (int u = 0; u < SCHUR_U; u++) {
                VDB_LOAD(B, j[u], W.n_pt, Vv[u], dg[u]);
#pragma unroll
                for (int k = 0; k < 6; k++) { w1[u][k] = B.w_pt[(size_t)(s1[u])*PT_REC + k]; w2[u][k] = B.w_pt[(size_t)(s2[u])*PT_REC + k]; }
            }
            }
#pragma unroll
            for (int u = 0; u < SCHUR_U; u++) {
                const double vinv = ok[u] ? ts_rcp(Vv[u] + dg[u]*irad) : 0.0;     // (v_rcp + Newton: an IEEE division is ~35 instructions per slot pair)
#pragma unroll
                for (int r = 0; r < 6; r++) {
                    const double wr = w1[u][r]*vinv;
#pragma unroll
                    for (int cc = 0; cc < 6; cc++) acc[r*6 + cc] += wr*w2[u][cc];
                }
            }
        }
        const int tx0 = use_pre ? pre.tx0 : L.sb_tx_off[b], tx1 = use_pre ? pre.tx1 : L.sb_tx_off[b+1];
        for (int q = tx0 + tid; q < tx1; q += SCHUR_T) {
            const bool fq = use_pre && q == tx0 + tid;
            const int s1 = fq ? pre.xs1 : L.sb_tx_s1[q], s2 = fq ? pre.xs2 : L.sb_tx_s2[q], j = fq ? pre.xj : L.sb_tx_lm[q];
            double Vd[6], Vi[6];
#pragma unroll
            for (int k = 0; k < 6; k++) Vd[k] = B.V_tx[(size_t)k*W.n_text + j];
            Vd[0] += B.dgs_tx[j]*irad; Vd[3] += B.dgs_tx[(size_t)W.n_text + j]*irad; Vd[5] += B.dgs_tx[(size_t)2*W.n_text + j]*irad;
            double W1[18], W2[18];
#pragma unroll
            for (int k = 0; k < 18; k++) { W1[k] = B.w_tx[(size_t)(s1)*TX_REC + k]; W2[k] = B.w_tx[(size_t)(s2)*TX_REC + k]; }
            if (!inv_sym3(Vd, Vi)) { st->step_fail = 1; continue; }
#pragma unroll
            for (int r = 0; r < 6; r++) {
                double t0 = W1[r*3]*Vi[0] + W1[r*3+1]*Vi[1] + W1[r*3+2]*Vi[2];
                double t1 = W1[r*3]*Vi[1] + W1[r*3+1]*Vi[3] + W1[r*3+2]*Vi[4];
                double t2 = W1[r*3]*Vi[2] + W1[r*3+1]*Vi[4] + W1[r*3+2]*Vi[5];
#pragma unroll
                for (int cc = 0; cc < 6; cc++) acc[r*6 + cc] += t0*W2[cc*3] + t1*W2[cc*3+1] + t2*W2[cc*3+2];
            }
        }
        SCHUR_STAMP(1);                                         // (the slot pairs gathered and multiplied)
        // operands of the tail, independent of the sums: issued before the reduction
        double tail = 0.0;
        if (tq) {
            if (a == c && wave > 0) {
#pragma unroll
                for (int i = 0; i < 7; i++) tls[(7*(wave - 1) + i)*65 + lane] = t_on ? tl[i] : 0.0;
            } else if (a != c && wave == 0 && lane < 36) {
                if (pre.pab >= 0) tail -= to0;
                if (pre.pba >= 0) tail -= to1;
            }
        } else if (wave == 0 && lane < 36) {
            const int r = lane/6, cc = lane % 6;
            const double *out = B.pairOut;
            if (a == c) {
                const double *rt = out + (size_t)sym6(r, cc)*L.n_pair, *rh = out + (size_t)(63 + sym6(r, cc))*L.n_pair;
                const double sgd = r == cc ? W.sig_p[6*a + r] : 1.0, dgm = r == cc ? B.dgs_p[6*a + r] : 0.0;      // (requested with the ranges)
                tail = use_pre ? range_sum<24>(rt, pre.t0, pre.t1) + range_sum<24>(rh, pre.h0, pre.h1)
                               : range_sum<24>(rt, L.pose_t_off[a], L.pose_t_off[a+1]) + range_sum<24>(rh, L.pose_h_off[a], L.pose_h_off[a+1]);
                // the damping of a candidate this launch has just accepted: postlin_fused's expression on this block's own diagonal sums (workgroup 0 stores the same values for the launches to come)
                if (r == cc && !multi) { const double sg_ = first_ ? 1.0/(1.0 + sqrt(tail)) : sgd;
                    tail += (SCHUR_NW == 4 && fresh ? clampd(sg_*sg_*tail, W.min_diag, W.max_diag)/(sg_*sg_) : dgm)*irad; }      // multi-GPU: added once after the all-reduce
            } else {
                int pab = use_pre ? pre.pab : L.sb_pab[b], pba = use_pre ? pre.pba : L.sb_pba[b];
                if (pab >= 0) tail -= out[(size_t)(27 + r*6 + cc)*L.n_pair + pab];        // -(M Q)       target a, host c
                if (pba >= 0) tail -= out[(size_t)(27 + cc*6 + r)*L.n_pair + pba];        // -(M Q)^T     target c, host a
            }
        }
        if (SCHUR_NW > 1) {
            if (wave > 0) {
#pragma unroll
                for (int k = 0; k < 36; k++) lds[((wave - 1)*36 + k)*64 + lane] = acc[k];
            }
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int k = 0; k < 36; k++) acc[k] += (lds[k*64 + lane] + lds[(36 + k)*64 + lane]) + lds[(72 + k)*64 + lane];
                if (tq && a == c && lane < 36) {                // the diagonal block's tail from the values waves 1 - 3 left in LDS
                    const int r = lane/6, cc = lane % 6;
                    const double *row = tls + sym6(r, cc)*65;
                    double st_ = 0.0, sh_ = 0.0;
#pragma unroll 8
                    for (int q = 0; q < 32; q++) st_ += row[q];
#pragma unroll 8
                    for (int q = 0; q < 32; q++) sh_ += row[32 + q];
                    tail = st_ + sh_;
                    if (r == cc && !multi) { const double sg_ = first_ ? 1.0/(1.0 + sqrt(tail)) : tsg;      // (postlin_fused's expressions)
                        tail += (fresh ? clampd(sg_*sg_*tail, W.min_diag, W.max_diag)/(sg_*sg_) : tdg)*irad; }
                }
            }
            __syncthreads();
        }
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < 36; k++) lds[k*65 + lane] = acc[k];          // transpose: lane l < 36 sums entry l over the 64 lanes
        }
        __syncthreads();
        if (wave > 0) return;
        SCHUR_STAMP(2);                                         // (tail operands there, the four waves' sums met)
        double tot = 0.0;
        if (lane < 36) {
            const double *row = lds + lane*65;
#pragma unroll 16
            for (int k = 0; k < 64; k++) tot += row[k];
        }
        if (lane < 36) {
            const int r = lane/6, cc = lane % 6;
            const double v = tail - tot;
            const size_t ldS = (size_t)W.ldS;                // (sb_a <= sb_b: the first store is the upper triangle, which band storage does not hold)
            // band storage holds the lower triangle: the block goes to the row of the pose that comes LATER in S (with a plan order
            // that need not be the larger keyframe index)
            int ja = ia, jc = ic;
            if (W.ring) { const int nf = W.nfree[0], r0 = W.nfree[1];       // closure block (a pose of the loop's first separator against a far one): the ghost row
                if (ia - ic > W.ring_b && ic >= r0 && ic < r0 + W.ring_b) jc += nf - r0; else if (ic - ia > W.ring_b && ia >= r0 && ia < r0 + W.ring_b) ja += nf - r0; }
            const bool a_later = ja > jc;
            const int fq = L.sb_far ? L.sb_far[b] : -1;     // a block outside the band (long-range coupling): to the compact list, rows = the earlier keyframe a
            if (fq >= 0) W.Sfar[(size_t)fq*36 + r*6 + cc] = v;
            else {
            if (a == c || !W.band || a_later) W.S[(size_t)(6*ja + r)*ldS + 6*jc + cc] = v;
            if (a != c && (!W.band || !a_later)) W.S[(size_t)(6*jc + cc)*ldS + 6*ja + r] = v;
            }
        }
        SCHUR_STAMP(3);
#ifdef MID_STAMPS
        if (threadIdx.x == 0) atomicAdd((unsigned long long *)&W.dbg[40 + 8*ss_kind + 7], 1ull);
#endif
    } else {
        const int a = b - L.n_sb;
        if (a >= W.n_kf) return;                               // (the gradient-only grid is rounded up to a multiple of 8)
        const int ia = use_pre ? pre.ia : W.fidx[a];
        if (ia < 0) return;
        double acc[6] = {0,0,0,0,0,0};
        const int ps0 = use_pre ? pre.ps0 : L.pose_ps_off[a], ps1 = use_pre ? pre.ps1 : L.pose_ps_off[a+1];
        for (int base = ps0; base < ps1; base += SCHUR_T*SCHUR_U) {
            int s[SCHUR_U], j[SCHUR_U]; bool ok[SCHUR_U];
#pragma unroll
            for (int u = 0; u < SCHUR_U; u++) {
                const int q = base + u*SCHUR_T + tid; ok[u] = q < ps1;
                const int qc = min(q, ps1 - 1);
                if (use_pre && base == ps0) { s[u] = pre.s1[u]; j[u] = pre.j[u]; }
                else if (use_pre && base == ps0 + SCHUR_T*SCHUR_U) { s[u] = pre.s1[SCHUR_U + u]; j[u] = pre.j[SCHUR_U + u]; }
                else { s[u] = L.pose_ps[qc]; j[u] = L.pose_ps_lm[qc]; }
            }
            double w[SCHUR_U][6], bb[SCHUR_U], Vv[SCHUR_U], dg[SCHUR_U];
#pragma unroll
            for (int u = 0; u < SCHUR_U; u++) {
                VDB_LOADB(B, j[u], W.n_pt, Vv[u], dg[u], bb[u]);
#pragma unroll
                for (int k = 0; k < 6; k++) w[u][k] = B.w_pt[(size_t)(s[u])*PT_REC + k];
            }
#pragma unroll
            for (int u = 0; u < SCHUR_U; u++) {
                const double f = ok[u] ? bb[u]*ts_rcp(Vv[u] + dg[u]*irad) : 0.0;
#pragma unroll
                for (int k = 0; k < 6; k++) acc[k] += w[u][k]*f;
            }
        }
        const int ts0 = use_pre ? pre.ts0 : L.pose_ts_off[a], ts1 = use_pre ? pre.ts1 : L.pose_ts_off[a+1];
        for (int q = ts0 + tid; q < ts1; q += SCHUR_T) {
            const bool fq = use_pre && q == ts0 + tid;
            const int s = fq ? pre.xs1 : L.pose_ts[q], j = fq ? pre.xj : L.pose_ts_lm[q];
            double Vd[6], Vi[6];
#pragma unroll
            for (int k = 0; k < 6; k++) Vd[k] = B.V_tx[(size_t)k*W.n_text + j];
            Vd[0] += B.dgs_tx[j]*irad; Vd[3] += B.dgs_tx[(size_t)W.n_text + j]*irad; Vd[5] += B.dgs_tx[(size_t)2*W.n_text + j]*irad;
            if (!inv_sym3(Vd, Vi)) { st->step_fail = 1; continue; }
            double b0 = B.b_tx[j], b1 = B.b_tx[(size_t)W.n_text + j], b2 = B.b_tx[(size_t)2*W.n_text + j];
            double f0 = Vi[0]*b0 + Vi[1]*b1 + Vi[2]*b2, f1 = Vi[1]*b0 + Vi[3]*b1 + Vi[4]*b2, f2 = Vi[2]*b0 + Vi[4]*b1 + Vi[5]*b2;
#pragma unroll
            for (int k = 0; k < 6; k++)
                acc[k] += B.w_tx[(size_t)(s)*TX_REC + (k*3)]*f0 + B.w_tx[(size_t)(s)*TX_REC + (k*3 + 1)]*f1 + B.w_tx[(size_t)(s)*TX_REC + (k*3 + 2)]*f2;
        }
        double bpv = 0.0;
        if (tid < 6) {
            if (multi) bpv = B.bp_loc[6*a + tid];
            else if (SCHUR_NW == 4 && fresh) {              // the gradient row of a candidate this launch has just accepted: postlin_fused's sums for this pose
                const double *out = B.pairOut; const size_t np = L.n_pair;
                bpv = range_sum<24>(out + (size_t)(21 + tid)*np, L.pose_t_off[a], L.pose_t_off[a+1]) - range_sum<24>(out + (size_t)(84 + tid)*np, L.pose_h_off[a], L.pose_h_off[a+1]);
            } else bpv = B.bp[6*a + tid];
        }
#pragma unroll
        for (int k = 0; k < 6; k++) acc[k] = wave_sum1(acc[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 6; k++) lds[wave*6 + k] = acc[k];
        }
        __syncthreads();
        if (tid < 6) W.g[6*ia + tid] = bpv - (SCHUR_NW > 1 ? (((lds[tid] + lds[6 + tid]) + lds[12 + tid]) + lds[18 + tid]) : lds[tid]);
        SCHUR_STAMP(3);
    }
}

// Large maps: FOUR S blocks per wave, 16 lanes each.  At 5000 keyframes a block gathers ~70 slot pairs: a whole wave per block left most
// load slots empty and paid a 64-lane reduction (36 LDS writes + 64 reads) per block; here a 16-lane group walks its block's list 64
// entries per round trip (4 in flight per lane), the 36 sums of a group are transposed through a 36 x 17 LDS tile and every lane
// finishes up to three entries of the block (tail: pose-pair products, damping) and stores them.  Same sums, same order within a lane;
// the order ACROSS lanes differs from k_schur_t<1> (16 partial sums instead of 64), which the tests' tolerances cover.
#ifndef SCHURQ_U
#define SCHURQ_U 1                          // list entries per lane in flight (k_schur_quad): 1 -> 128 registers, four waves per SIMD (101 us with 2 / three waves, 97 us with 1 at 5000 keyframes; 113 us with 4 / two waves)
#endif
// TEXT = false: a level without text planes (the reference's GlobalBA) -- the plane part (3x3 inverse, 18-value records) sets the kernel's
// register count (214: two waves per SIMD); without it three fit.
// The reduced gradient of one pose by one wave (the gradient branch of k_schur_t<1>): g_a = b_a - sum_j W_aj (V_j + lambda_j)^-1 b_j over the pose's slots.
template <bool TEXT>
__device__ __forceinline__ void schur_grad_wave(const Work &W, const LevelDev &L, LmState *st, const LinBuf &B, double irad, int multi, int a, int lane) {
    const int ia = W.fidx[a];
    if (ia < 0) return;
    double acc[6] = {0,0,0,0,0,0};
    const int ps0 = L.pose_ps_off[a], ps1 = L.pose_ps_off[a+1];
    for (int base = ps0; base < ps1; base += 64*SCHUR_U) {
        int s[SCHUR_U], j[SCHUR_U]; bool ok[SCHUR_U];
#pragma unroll
        for (int u = 0; u < SCHUR_U; u++) { const int q = base + u*64 + lane; ok[u] = q < ps1; const int qc = min(q, ps1 - 1); s[u] = L.pose_ps[qc]; j[u] = L.pose_ps_lm[qc]; }
        double w[SCHUR_U][6], bb[SCHUR_U], Vv[SCHUR_U], dg[SCHUR_U];
#pragma unroll
        for (int u = 0; u < SCHUR_U; u++) {
            VDB_LOADB(B, j[u], W.n_pt, Vv[u], dg[u], bb[u]);
#pragma unroll
            for (int k = 0; k < 6; k++) w[u][k] = B.w_pt[(size_t)(s[u])*PT_REC + k];
        }
#pragma unroll
        for (int u = 0; u < SCHUR_U; u++) {
            const double f = ok[u] ? bb[u]*ts_rcp(Vv[u] + dg[u]*irad) : 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) acc[k] += w[u][k]*f;
        }
    }
    if (TEXT) for (int q = L.pose_ts_off[a] + lane; q < L.pose_ts_off[a+1]; q += 64) {
        const int s = L.pose_ts[q], j = L.pose_ts_lm[q];
        double Vd[6], Vi[6];
#pragma unroll
        for (int k = 0; k < 6; k++) Vd[k] = B.V_tx[(size_t)k*W.n_text + j];
        Vd[0] += B.dgs_tx[j]*irad; Vd[3] += B.dgs_tx[(size_t)W.n_text + j]*irad; Vd[5] += B.dgs_tx[(size_t)2*W.n_text + j]*irad;
        if (!inv_sym3(Vd, Vi)) { st->step_fail = 1; continue; }
        double b0 = B.b_tx[j], b1 = B.b_tx[(size_t)W.n_text + j], b2 = B.b_tx[(size_t)2*W.n_text + j];
        double f0 = Vi[0]*b0 + Vi[1]*b1 + Vi[2]*b2, f1 = Vi[1]*b0 + Vi[3]*b1 + Vi[4]*b2, f2 = Vi[2]*b0 + Vi[4]*b1 + Vi[5]*b2;
#pragma unroll
        for (int k = 0; k < 6; k++)
            acc[k] += B.w_tx[(size_t)(s)*TX_REC + (k*3)]*f0 + B.w_tx[(size_t)(s)*TX_REC + (k*3 + 1)]*f1 + B.w_tx[(size_t)(s)*TX_REC + (k*3 + 2)]*f2;
    }
    const double bpv = lane < 6 ? (multi ? B.bp_loc[6*a + lane] : B.bp[6*a + lane]) : 0.0;
    double mine = 0.0;
#pragma unroll
    for (int k = 0; k < 6; k++) { const double sk = wave_sum1(acc[k]); if (lane == k) mine = sk; }
    if (lane < 6) W.g[6*ia + lane] = bpv - mine;
}

// grid = nq workgroups for the S blocks (four per wave) + ng for the reduced gradient (one pose per wave), both multiples of 8: blocks and gradient
// are independent, and as two launches the second (22 us at 5000 keyframes) waited for the first (85 us)
template <bool TEXT>
__global__ __launch_bounds__(64) void k_schur_quad(Work W, LevelDev L, int multi, int nq, int ng) {
    LmState *st = W.st;
    if (st->done) return;
    __shared__ double lds[4*12*17];                             // a third of a group's 36 sums at a time: 6.5 KB, the registers set the occupancy
    if ((int)blockIdx.x >= nq) {                                // gradient part: the workgroups of ONE XCD take neighbouring poses (as k_schur_t with b0 > 0)
        const int i = (int)blockIdx.x - nq, a = (i & 7)*(ng >> 3) + (i >> 3);
        if (a < W.n_kf) schur_grad_wave<TEXT>(W, L, st, W.lb[st->lcur], 1.0/st->radius, multi, a, (int)threadIdx.x);
        return;
    }
    const int lane = threadIdx.x, grp = lane >> 4, sub = lane & 15;
    // workgroups are handed to the 8 XCDs round-robin (workgroup i -> XCD i mod 8), and an XCD's L2 does not see the others': neighbouring S
    // blocks read the same landmarks' records, so the workgroups of ONE XCD take a contiguous range of blocks (the kernel is bound by
    // L2 -> L1 line fills; with neighbouring blocks on eight different XCDs every record crossed the fabric up to eight times)
    const int per = nq >> 3, wg = ((int)blockIdx.x & 7)*per + ((int)blockIdx.x >> 3);     // (nq is a multiple of 8 workgroups)
    const int b = 4*wg + grp;
    const bool have = b < L.n_sb;
    const int bc = have ? b : L.n_sb - 1;
    const double irad = 1.0/st->radius;
    const LinBuf &B = W.lb[st->lcur];
    const int a = L.sb_a[bc], c = L.sb_b[bc];
    const int ia = W.fidx[a], ic = W.fidx[c];
    const bool live = have && ia >= 0 && ic >= 0;               // rows / columns of S exist for free poses only
    double acc[36];
#pragma unroll
    for (int k = 0; k < 36; k++) acc[k] = 0.0;
    const int pt0 = L.sb_pt_off[bc], pt1 = live ? L.sb_pt_off[bc+1] : pt0;
    for (int base = pt0; base < pt1; base += 16*SCHURQ_U) {
        int s1[SCHURQ_U], s2[SCHURQ_U], j[SCHURQ_U]; bool ok[SCHURQ_U];
#pragma unroll
        for (int u = 0; u < SCHURQ_U; u++) {
            const int q = base + u*16 + sub; ok[u] = q < pt1;
            const int qc = min(q, pt1 - 1);
            s1[u] = L.sb_pt_s1[qc]; s2[u] = L.sb_pt_s2[qc]; j[u] = L.sb_pt_lm[qc];
        }
        double w1[SCHURQ_U][6], w2[SCHURQ_U][6], Vv[SCHURQ_U], dg[SCHURQ_U];
#pragma unroll
        for (int u = 0; u < SCHURQ_U; u++) {
            VDB_LOAD(B, j[u], W.n_pt, Vv[u], dg[u]);
#pragma unroll
            for (int k = 0; k < 6; k++) { w1[u][k] = B.w_pt[(size_t)(s1[u])*PT_REC + k]; w2[u][k] = B.w_pt[(size_t)(s2[u])*PT_REC + k]; }
        }
#pragma unroll
        for (int u = 0; u < SCHURQ_U; u++) {
            const double vinv = ok[u] ? ts_rcp(Vv[u] + dg[u]*irad) : 0.0;
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const double wr = w1[u][r]*vinv;
#pragma unroll
                for (int cc = 0; cc < 6; cc++) acc[r*6 + cc] += wr*w2[u][cc];
            }
        }
    }
    if (TEXT && live) for (int q = L.sb_tx_off[bc] + sub; q < L.sb_tx_off[bc+1]; q += 16) {
        const int s1 = L.sb_tx_s1[q], s2 = L.sb_tx_s2[q], j = L.sb_tx_lm[q];
        double Vd[6], Vi[6];
#pragma unroll
        for (int k = 0; k < 6; k++) Vd[k] = B.V_tx[(size_t)k*W.n_text + j];
        Vd[0] += B.dgs_tx[j]*irad; Vd[3] += B.dgs_tx[(size_t)W.n_text + j]*irad; Vd[5] += B.dgs_tx[(size_t)2*W.n_text + j]*irad;
        // (t = W1 Vi first, then W2 three values at a time: W1, W2 and acc live together cost the kernel a wave per SIMD)
        double tv[18];
        {
            double W1[18];
#pragma unroll
            for (int k = 0; k < 18; k++) W1[k] = B.w_tx[(size_t)(s1)*TX_REC + k];
            if (!inv_sym3(Vd, Vi)) { st->step_fail = 1; continue; }
#pragma unroll
            for (int r = 0; r < 6; r++) {
                tv[r*3] = W1[r*3]*Vi[0] + W1[r*3+1]*Vi[1] + W1[r*3+2]*Vi[2];
                tv[r*3+1] = W1[r*3]*Vi[1] + W1[r*3+1]*Vi[3] + W1[r*3+2]*Vi[4];
                tv[r*3+2] = W1[r*3]*Vi[2] + W1[r*3+1]*Vi[4] + W1[r*3+2]*Vi[5];
            }
        }
#pragma unroll
        for (int cc = 0; cc < 6; cc++) {
            const double x0 = B.w_tx[(size_t)(s2)*TX_REC + cc*3], x1 = B.w_tx[(size_t)(s2)*TX_REC + cc*3 + 1], x2 = B.w_tx[(size_t)(s2)*TX_REC + cc*3 + 2];
#pragma unroll
            for (int r = 0; r < 6; r++) acc[r*6 + cc] += tv[r*3]*x0 + tv[r*3+1]*x1 + tv[r*3+2]*x2;
        }
    }
    // tails of this lane's entries o = sub, 12 + sub, 24 + sub (sub < 12), independent of the sums: issued before the reduction
    double tail[3] = {0.0, 0.0, 0.0};
    if (live && sub < 12) {
        const double *out = B.pairOut;
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int o = sub + 12*t;
            const int r = o/6, cc = o - 6*r;
            if (a == c) {
                const double *rt = out + (size_t)sym6(r, cc)*L.n_pair, *rh = out + (size_t)(63 + sym6(r, cc))*L.n_pair;
                tail[t] = range_sum<8>(rt, L.pose_t_off[a], L.pose_t_off[a+1]) + range_sum<8>(rh, L.pose_h_off[a], L.pose_h_off[a+1]);     // (8 in flight: a keyframe of a large map has ~8 pairs each way; 24 cost the kernel a wave per SIMD)
                if (r == cc && !multi) tail[t] += B.dgs_p[6*a + r]*irad;      // multi-GPU: added once after the all-reduce
            } else {
                const int pab = L.sb_pab[bc], pba = L.sb_pba[bc];
                if (pab >= 0) tail[t] -= out[(size_t)(27 + r*6 + cc)*L.n_pair + pab];        // -(M Q)       target a, host c
                if (pba >= 0) tail[t] -= out[(size_t)(27 + cc*6 + r)*L.n_pair + pba];        // -(M Q)^T     target c, host a
            }
        }
    }
    double *tile = lds + grp*12*17;
    const size_t ldS = (size_t)W.ldS;
    int ja = ia, jc = ic;
    if (W.ring) { const int nf = W.nfree[0], r0 = W.nfree[1];               // closure block (a pose of the loop's first separator against a far one): the ghost row
        if (ia - ic > W.ring_b && ic >= r0 && ic < r0 + W.ring_b) jc += nf - r0; else if (ic - ia > W.ring_b && ia >= r0 && ia < r0 + W.ring_b) ja += nf - r0; }
    const bool a_later = ja > jc;
    const int fq = (live && L.sb_far) ? L.sb_far[bc] : -1;     // a block outside the band (long-range coupling): to the compact list
#pragma unroll
    for (int t = 0; t < 3; t++) {
#pragma unroll
        for (int k = 0; k < 12; k++) tile[k*17 + sub] = acc[12*t + k];
        __syncthreads();                                        // (one wave: an s_barrier of one wave)
        if (live && sub < 12) {
            const int o = sub + 12*t;
            const double *row = tile + sub*17;
            double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
            for (int q = 0; q < 16; q += 4) { s0 += row[q]; s1 += row[q + 1]; s2 += row[q + 2]; s3 += row[q + 3]; }
            const double v = tail[t] - ((s0 + s1) + (s2 + s3));
            const int r = o/6, cc = o - 6*r;
            if (fq >= 0) W.Sfar[(size_t)fq*36 + o] = v;
            else {
            if (a == c || !W.band || a_later) W.S[(size_t)(6*ja + r)*ldS + 6*jc + cc] = v;
            if (a != c && (!W.band || !a_later)) W.S[(size_t)(6*jc + cc)*ldS + 6*ja + r] = v;
            }
        }
        __syncthreads();
    }
}

