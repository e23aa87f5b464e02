// libtsba.so -- MI355X (gfx950) bundle adjustment / pose optimisation for TextSLAM's optimizer:: hot path.
// C ABI in include/tsba.h.  One translation unit: kernels + Levenberg-Marquardt driver + ABI.
//
// Kernel sequence of one LM iteration (all launched back-to-back on one stream, no host sync; the LM state machine
// lives in device memory and every kernel starts by reading it):
//   k_schur   one wave per 6x6 block of the reduced camera system S (gather over landmark slot pairs) + reduced gradient
//   k_solve   one workgroup: blocked Cholesky of S in LDS, pose step
//   k_back    landmark back-substitution, candidate parameters x (+) dx, step norm, model cost change
//   k_linearize<COST>  candidate cost (residuals + Huber only)
//   k_decide  step quality, trust-region update, accept / reject, convergence tests (Ceres 1.x semantics)
//   k_linearize<FULL>  (only after an accepted step) residual + analytic Jacobian + IRLS weight + per-pair / per-group
//                      J^T J, J^T r, J^T J_landmark sums: one wave per (target KF, host KF) pair of scene observations,
//                      one wave per (KF, text) observation of up to 64 photometric blocks
//   k_mid     per landmark V, b and the host-pose column of W; per pair the host-side products
//   k_postlin pose diagonal / gradient, Jacobi scaling, cost, gradient tolerance
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types only: the library is dlopen()ed by tsba_comm_init, single-GPU use never touches RCCL
#include "../../include/tsba.h"
#include "tsba_device.h"
#include "tsraster.h"
#include "tsba_plan.h"

// ------------------------------------------------------------------------------------------------ device structs
struct LmState {
    double radius, decrease_factor, x_cost, x_norm, cand_cost, model_change, step_norm, gmax, cost0;
    int cur, done, need_lin, first, it, accepted, term, invalid, max_it, step_fail, lcur, lin_done;      // lin_done: the iterative reduced-system solve of this trial has converged (tsba_pcg.h): the remaining preconditioner launches return at once
    int ns_active, nt_active, n_bad_scene, n_bad_tfeat, n_bad_text, pad2;
    long long n_lin, n_cost;
};

struct LevelDev {            // device copies of HostPlan + per-level inputs
    int level, n_sc, n_pair, n_tg, n_pslot, n_tslot, n_sb, n_tfeat, bw_rows;      // bw_rows: rows of S below a pose block that can be non-zero
    double K[4];             // K_l
    int img_w, img_h;
    const uint8_t *const *img;      // [n_kf] device pointers
    const int *sc_obs, *sc_kf, *sc_pt, *sc_flag, *sc_slot; const double *sc_uv;
    const int *pair_i, *pair_h, *pair_hpos, *pair_sc_off, *pair_tg_off, *pair_tg;
    const int *tg_tobs, *tg_kf, *tg_text, *tg_pair, *tg_slot, *tg_rec, *tg_ppos, *pt_pose6, *pt_pair4;
    const int *pls_off, *pslot_pose, *pslot_pair, *pslot_lm, *tls_off, *tslot_pose, *tslot_pair, *tslot_lm;
    const int *sb_a, *sb_b, *sb_pab, *sb_pba, *sb_pt_off, *sb_pt_s1, *sb_pt_s2, *sb_pt_lm, *sb_tx_off, *sb_tx_s1, *sb_tx_s2, *sb_tx_lm;
    const int *pose_t_off, *pose_t, *pose_h_off, *pose_h, *pose_ps_off, *pose_ps, *pose_ps_lm, *pose_ts_off, *pose_ts, *pose_ts_lm;
    const int *tfeat_off, *tfeat_raw; const double *tfeat_uv, *tfeat_ref;
    const int *pf_g, *pf_f; int n_pf;   // pose-only path: flat (group, feature) list of the frame's text features
    const int *kf_order;                // nullptr: the rows of S follow the keyframe index; else kf_order[i] = keyframe at position i (tsba_plan.h: rcm_order)
    // band + long-range coupling (HostPlan::far_* / fb_*, tsba_pcg.h): nullptr / 0 unless the plan split the reduced system into M (the sb_* lists) + E.
    // sb_far: nullptr in this view; in the view of E that launch_schur derives (sb_* = the fb_* lists) the index of every block in W.Sfar
    const int *sb_far, *far_a, *far_b, *far_off, *far_ent; int n_far, far_B;
    const int *fb_id, *fb_pab, *fb_pba, *fb_pt_off, *fb_pt_s1, *fb_pt_s2, *fb_pt_lm, *fb_tx_off, *fb_tx_s1, *fb_tx_s2, *fb_tx_lm;
};

#define PT_REC 8
#define TX_REC 28
struct LinBuf {              // everything one linearisation produces
    double *pairM, *pairCost, *pairR, *pairOut, *tgM, *tgCost;
    double *w_pt;                       // per point slot, one 64-byte record: w[0..5] | v | b   (PT_REC doubles; the host slot of a landmark
                                        // holds its host column -sum Q^T w in [0..5], formed by k_mid from w and the pair's R_cr)
    double *V_pt, *b_pt, *dgs_pt;       // per point: V, b, clamp(sigma^2 V)/sigma^2  (lambda = dgs / radius)
    double *w_tx;                       // per plane slot, one 224-byte record: W[0..17] | V6 [18..23] | b3 [24..26]   (TX_REC doubles)
    double *V_tx, *b_tx, *dgs_tx;       // per plane: V [6][n], b [3][n], dgs [3][n]
    double *Hd, *bp, *dgs_p;            // per pose: diag(H_pp), gradient, dgs.  Hd | bp | scal[8] are one allocation (hb):
    double *bp_loc;                     // multi-GPU: this rank's part of bp (the reduced gradient is assembled from it)
    double *lmpart;                     // per k_mid block: (gradient max, |x|^2) of its landmarks, cost of its pairs (+ their text groups)
};

struct PoseState;
struct Work {                // device work buffers (sized for the largest level)
    int n_kf, n_pt, n_text, n_tobs, N;      // N = 6 n_kf
    int rank, world;                        // landmark shard of this process (global BA over RCCL), 0 / 1 otherwise
    double K0[4];
    double w_sx, w_sy, w_t, huber_s, huber_t;
    int filter_good;
    double min_diag, max_diag;
    // parameters: double-buffered (x = buf[cur], candidate = buf[cur^1])
    double *pose[2], *rho[2], *theta[2];
    const double *pt_ray; const int *pt_host; const double *pt_Trw;
    const int *text_host; const double *text_Twr; const double *text_box;
    const int *tobs_kf, *tobs_text, *tobs_fgood_off;
    uint8_t *sgood, *tobs_good, *tfgood;
    double *musig;                      // [n_tobs][2]
    int *kf_in, *kf_const, *act_pt, *act_tx;
    int *fidx, *nfree;                  // compressed index of the free poses in S / g
    double *cb, *cbm;                   // multi-GPU exchange buffers: cb = [Hd 6n | bp 6n | cost, |x_lm|^2, step^2, mcc] (sum), cbm = gradient max (max)
    long long *dbg;                     // [64] cycle stamps of instrumented kernels (debug)
    double *LDbuf;                      // diagonal of the inverse diagonal factors (large-system Cholesky)
    int ldS, band;                      // S(i,j) = S[i*ldS + j]; band: S holds only the band of the reduced camera matrix (large systems)
    int ring;                           // 1: ring-shaped co-visibility (one loop closure, tsba_plan.h): the closure blocks -- the loop's first poses S against its last --
                                        // live in ghost rows behind the last free pose (row = nfree + row - first row of S; nfree[1] = first row of S)
    int ring_g, ring_b, ring_k0;        // interiors of the loop (worst case, a power of two); band = separator size in pose blocks; first keyframe of the loop
    double *Sy;                         // right-hand-side row of the large-system solver (row n of the small one lives in LDS)
    unsigned long long *hprog;          // pinned host word (seq << 32 | it << 1 | done): lets the host stop enqueuing a converged pass
    unsigned int pass_seq;
    // linearisation outputs, double-buffered: lb[lcur] belongs to x, lb[lcur^1] to the LM candidate (speculative)
    LinBuf lb[2];
    double *sig_pt, *sig_tx, *sig_p;    // Jacobi column scales, fixed at the first linearisation of a pass
    double *S, *g, *dp, *dl_pt, *dl_tx;
    double *partial;                    // [nblocks_back][2]
    int *cntpart;                       // per k_participation workgroup: active scene blocks, active text blocks
    double *posepart;                   // large maps: per k_pose_sums workgroup (21 poses): gradient max, |x|^2
    LmState *st;
    PoseState *pst; double *ppart;      // pose-only path (tsba_pose.h): double-buffered state, [2][G][28] partial sums
    // band + long-range blocks, preconditioned conjugate gradients (tsba_pcg.h): the blocks outside the band [n_far][36] (rows: the earlier keyframe),
    // the iteration's vectors in the compressed row space of S, per-workgroup partial sums [2][workgroups], double-buffered scalars, statistics
    double *Sfar, *pc_x, *pc_r, *pc_p[2], *pc_q, *pc_g0, *pc_part;
    struct PcgState *pcs; int *pc_stat;
};

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(c, std::string(#x) + ": " + hipGetErrorString(e_)); return TSBA_ERR_DEVICE; } } while (0)

// ------------------------------------------------------------------------------------------------ kernels
__device__ __forceinline__ double wave_sum1(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int NT>
__device__ __forceinline__ double block_sum(double v, double *lds) {     // deterministic (fixed order), NT threads, all get the result
    const int t = threadIdx.x;
    lds[t] = v; __syncthreads();
    if (t < 64) {
        double s = lds[t];
#pragma unroll
        for (int k = 64; k < NT; k += 64) s += lds[t + k];
        s = wave_sum1(s);
        if (t == 0) lds[0] = s;
    }
    __syncthreads();
    const double r = lds[0]; __syncthreads();
    return r;
}
template <int NT>
__device__ __forceinline__ double block_max(double v, double *lds) {
    const int t = threadIdx.x;
    lds[t] = v; __syncthreads();
    if (t < 64) {
        double s = lds[t];
#pragma unroll
        for (int k = 64; k < NT; k += 64) s = fmax(s, lds[t + k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s = fmax(s, __shfl_xor(s, o, 64));
        if (t == 0) lds[0] = s;
    }
    __syncthreads();
    const double r = lds[0]; __syncthreads();
    return r;
}

// Sum over a variable-length gather list with U entries (index, then value) in flight per round trip instead of one:
// val(idx) is evaluated for clamped indices and masked, the summation order is the list order.
template <int U, class F>
__device__ __forceinline__ double gather_sum(const int *list, int n, F &&val) {
    double s = 0.0;
    for (int base = 0; base < n; base += U) {
        int idx[U]; double v[U];
#pragma unroll
        for (int u = 0; u < U; u++) idx[u] = list[min(base + u, n - 1)];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = val(idx[u]);
#pragma unroll
        for (int u = 0; u < U; u++) s += base + u < n ? v[u] : 0.0;
    }
    return s;
}

// contiguous range with U loads in flight per round trip, summed in index order
template <int U>
__device__ __forceinline__ double range_sum(const double *v, int i0, int i1) {
    double s = 0.0;
    for (int base = i0; base < i1; base += U) {
        double x[U];
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = v[min(base + u, i1 - 1)];
#pragma unroll
        for (int u = 0; u < U; u++) s += base + u < i1 ? x[u] : 0.0;
    }
    return s;
}

// ---- start point of a solve: parameters (both buffers), inlier flags, LM state
struct ResetSrc { const double *pose0, *rho0, *theta0; const uint8_t *sg0, *tg0, *tf0; long long n_pose, n_rho, n_theta, n_sg, n_tg, n_tf; };
__global__ __launch_bounds__(256) void k_reset_state(Work W, ResetSrc A) {
    const long long t = (long long)blockIdx.x*blockDim.x + threadIdx.x, n = (long long)gridDim.x*blockDim.x;
    for (long long k = t; k < A.n_pose; k += n) { const double v = A.pose0[k]; W.pose[0][k] = v; W.pose[1][k] = v; }
    for (long long k = t; k < A.n_rho; k += n) { const double v = A.rho0[k]; W.rho[0][k] = v; W.rho[1][k] = v; }
    for (long long k = t; k < A.n_theta; k += n) { const double v = A.theta0[k]; W.theta[0][k] = v; W.theta[1][k] = v; }
    for (long long k = t; k < A.n_sg; k += n) W.sgood[k] = A.sg0[k];
    for (long long k = t; k < A.n_tg; k += n) W.tobs_good[k] = A.tg0[k];
    for (long long k = t; k < A.n_tf; k += n) W.tfgood[k] = A.tf0[k];
    if (t == 0) memset(W.st, 0, sizeof(LmState));
}

// ---- pass initialisation
__global__ void k_pass_reset(Work W, double radius0, int max_it) {
    LmState *s = W.st;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int cur = s->cur; long long nl = s->n_lin, nc = s->n_cost;
        memset(s, 0, sizeof(LmState));
        s->cur = cur; s->n_lin = nl; s->n_cost = nc;
        s->radius = radius0; s->decrease_factor = 2.0; s->need_lin = 1; s->first = 1; s->max_it = max_it;
        if (W.hprog) { *W.hprog = (unsigned long long)W.pass_seq << 32; __threadfence_system(); }
    }
    int t = blockIdx.x*blockDim.x + threadIdx.x, n = gridDim.x*blockDim.x;
    for (int k = t; k < W.n_kf; k += n) { W.kf_in[k] = 0; W.kf_const[k] = 0; }
    for (int k = t; k < W.n_pt; k += n) W.act_pt[k] = 0;
    for (int k = t; k < W.n_text; k += n) W.act_tx[k] = 0;
}

// which candidates are active (good flags), which keyframes participate (FLAG_KFIN, optimizer.cc:1410-1411,1428,1514-1515)
__global__ __launch_bounds__(256) void k_participation(Work W, LevelDev L, int partials) {
    // workgroups 0 .. nb_sc-1: one scene candidate per thread; the rest: one (KF, text) group per WAVE, its features on the lanes
    // (a thread walking the 64 features of a group alone was most of this kernel's 14 us)
    __shared__ int cnt_s, cnt_t;
    if (threadIdx.x == 0) { cnt_s = 0; cnt_t = 0; }
    __syncthreads();
    const int nb_sc = (L.n_sc + 255) >> 8, lane = threadIdx.x & 63;
    if ((int)blockIdx.x < nb_sc) {
        const int t = blockIdx.x*256 + threadIdx.x;
        bool act = false;
        if (t < L.n_sc) {
            act = !W.filter_good || W.sgood[L.sc_flag[t]];
            if (act) {
                int pt = L.sc_pt[t], h = W.pt_host[pt];
                W.kf_in[L.sc_kf[t]] = 1;
                if (h >= 0) { W.kf_in[h] = 1; W.act_pt[pt] = 1; }
            }
        }
        const int nw = __popcll(__ballot(act));
        if (lane == 0 && nw) atomicAdd(&cnt_s, nw);
    } else {
        const int g = (blockIdx.x - nb_sc)*4 + (threadIdx.x >> 6);
        if (g < L.n_tg) {
            const int tb = L.tg_tobs[g], j = L.tg_text[g];
            if (!W.filter_good || W.tobs_good[tb]) {
                const int f0 = L.tfeat_off[j], f1 = L.tfeat_off[j+1], fg = W.tobs_fgood_off[tb];
                int cnt = 0;
                for (int f = f0 + lane; f < f1; f += 64) if (!W.filter_good || W.tfgood[fg + L.tfeat_raw[f]]) cnt++;
                cnt = (int)wave_sum1((double)cnt);
                if (lane == 0 && cnt > 0) {
                    const int h = W.text_host[j];
                    W.kf_in[L.tg_kf[g]] = 1;
                    if (h >= 0) { W.kf_in[h] = 1; W.act_tx[j] = 1; }
                    atomicAdd(&cnt_t, cnt);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // single GPU: per-workgroup partials, summed by the gauge kernel; multi-GPU: the counts are all-reduced before the gauge
        // kernel runs, so they go straight to the state
        if (partials) { W.cntpart[2*blockIdx.x] = cnt_s; W.cntpart[2*blockIdx.x + 1] = cnt_t; }
        else { if (cnt_s) atomicAdd(&W.st->ns_active, cnt_s); if (cnt_t) atomicAdd(&W.st->nt_active, cnt_t); }
    }
}
// block counts of k_participation -> LM state (called by the gauge kernels' first wave / all threads)
__device__ __forceinline__ void sum_counts(const Work &W, int ncp, int tid, int nthreads, int *lds2 /* [2] zeroed */) {
    int a = 0, b = 0;
    for (int k = tid; k < ncp; k += nthreads) { a += W.cntpart[2*k]; b += W.cntpart[2*k + 1]; }
    if (a) atomicAdd(&lds2[0], a);
    if (b) atomicAdd(&lds2[1], b);
}
// gauge fixing, optimizer.cc:1562-1588 / :1825-1830
__global__ void k_gauge(Work W, const uint8_t *kf_initial, int state, int ncp, const int *order) {
    __shared__ int s_cnt2[2];
    if (threadIdx.x == 0) { s_cnt2[0] = 0; s_cnt2[1] = 0; }
    __syncthreads();
    if (ncp) { sum_counts(W, ncp, threadIdx.x, blockDim.x, s_cnt2); __syncthreads(); if (threadIdx.x == 0) { W.st->ns_active = s_cnt2[0]; W.st->nt_active = s_cnt2[1]; } }
    if (threadIdx.x || blockIdx.x) return;
    int cnt = 0;
    for (int k = 0; k < W.n_kf; k++) { if (kf_initial[k] && W.kf_in[k]) W.kf_const[k] = 1; cnt += W.kf_in[k]; }
    if (state == TSBA_STATE_LOCAL && cnt > 3) {
        int fixed = 0;
        for (int k = 0; k < W.n_kf && fixed < 3; k++) if (W.kf_in[k]) { W.kf_const[k] = 1; fixed++; }
    }
    int nf = 0;                                                // rows of S: free poses in keyframe order, or in the plan's order
    int row0 = 0;
    for (int i = 0; i < W.n_kf; i++) { const int k = order ? order[i] : i; if (i == W.ring_k0) row0 = nf; W.fidx[k] = (W.kf_in[k] && !W.kf_const[k]) ? nf++ : -1; }
    W.nfree[0] = nf; W.nfree[1] = row0;                         // (ring maps: free poses before the loop's first keyframe)
}

// windows of up to 64 keyframes: one lane per keyframe, ballots instead of the serial walk (7.8 -> ~2 us per pass)
__global__ __launch_bounds__(64) void k_gauge_wave(Work W, const uint8_t *kf_initial, int state, int ncp) {
    __shared__ int s_cnt2[2];
    const int k = threadIdx.x;
    if (k == 0) { s_cnt2[0] = 0; s_cnt2[1] = 0; }
    __syncthreads();
    if (ncp) { sum_counts(W, ncp, k, 64, s_cnt2); __syncthreads(); if (k == 0) { W.st->ns_active = s_cnt2[0]; W.st->nt_active = s_cnt2[1]; } }
    const bool on = k < W.n_kf;
    const int in = on ? W.kf_in[k] : 0, ini = on ? kf_initial[k] : 0;
    const unsigned long long m_in = __ballot(in != 0);
    int cst = (ini && in) ? 1 : 0;
    if (state == TSBA_STATE_LOCAL && __popcll(m_in) > 3) {
        const int before = __popcll(m_in & ((1ull << k) - 1));      // participating keyframes with a smaller index
        if (in && before < 3) cst = 1;                               // the first three of them are held constant
    }
    const bool fre = in && !cst;
    const unsigned long long m_free = __ballot(fre);
    if (on) { W.kf_const[k] = cst; W.fidx[k] = fre ? __popcll(m_free & ((1ull << k) - 1)) : -1; }
    if (k == 0) { W.nfree[0] = __popcll(m_free); W.nfree[1] = 0; }
}
// the same for large maps: 1024 threads, consecutive keyframes per thread, one block-wide exclusive scan for the compressed indices
// (the single-thread walk above costs 1.4 ms at 5000 keyframes)
__global__ __launch_bounds__(1024) void k_gauge_par(Work W, const uint8_t *kf_initial, int state, int ncp, const int *order) {
    __shared__ int s_scan[1024]; __shared__ int s_first[3]; __shared__ int s_cnt; __shared__ int s_cnt2[2];
    if (threadIdx.x == 0) { s_cnt2[0] = 0; s_cnt2[1] = 0; }
    __syncthreads();
    if (ncp) { sum_counts(W, ncp, threadIdx.x, 1024, s_cnt2); __syncthreads(); if (threadIdx.x == 0) { W.st->ns_active = s_cnt2[0]; W.st->nt_active = s_cnt2[1]; } }
    const int tid = threadIdx.x, per = (W.n_kf + 1023)/1024, k0 = tid*per, k1 = min(W.n_kf, k0 + per);
    if (tid == 0) {                                           // STATE_LOCAL: the first three participating keyframes are held constant
        int f = 0; s_first[0] = s_first[1] = s_first[2] = -1;
        if (state == TSBA_STATE_LOCAL) for (int k = 0; k < W.n_kf && f < 3; k++) if (W.kf_in[k]) s_first[f++] = k;
    }
    int cin = 0;
    for (int k = k0; k < k1; k++) cin += W.kf_in[k];
    s_scan[tid] = cin; __syncthreads();
    for (int d = 512; d > 0; d >>= 1) { if (tid < d) s_scan[tid] += s_scan[tid + d]; __syncthreads(); }
    if (tid == 0) s_cnt = s_scan[0];
    __syncthreads();
    const bool fix3 = state == TSBA_STATE_LOCAL && s_cnt > 3;
    int nfree = 0;                                             // from here on a thread's range is a range of POSITIONS (= keyframes without a plan order)
    for (int i = k0; i < k1; i++) {
        const int k = order ? order[i] : i;
        int cst = (kf_initial[k] && W.kf_in[k]) ? 1 : 0;
        if (fix3 && (k == s_first[0] || k == s_first[1] || k == s_first[2])) cst = 1;
        W.kf_const[k] = cst;
        nfree += (W.kf_in[k] && !cst) ? 1 : 0;
    }
    __syncthreads();
    s_scan[tid] = nfree; __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { const int t = tid >= d ? s_scan[tid - d] : 0; __syncthreads(); s_scan[tid] += t; __syncthreads(); }
    int at = s_scan[tid] - nfree;                              // exclusive prefix
    if (tid == 0) W.nfree[1] = 0;
    __syncthreads();
    for (int i = k0; i < k1; i++) { const int k = order ? order[i] : i; if (i == W.ring_k0 && i > 0) W.nfree[1] = at; W.fidx[k] = (W.kf_in[k] && !W.kf_const[k]) ? at++ : -1; }
    if (tid == 1023) W.nfree[0] = s_scan[1023];
}

// ---- mu / sigma of a projected text box: tool::GetProjText x4 + tool::CalTextinfo (src/tool.cc:1178-1262,1655-1728)
// with cv::fillPoly's scan conversion (boundary Bresenham lines + 16.16 fixed-point scanline spans).  One workgroup per
// (KF, text) observation; the polygon mask of the clamped bounding box lives in LDS as a bit field.
#define MS_THREADS 256
__device__ __forceinline__ void musigma_wg(const Work &W, const LevelDev &L, const int g, const double *pose, const double *theta) {
    __shared__ unsigned mask[MS_MASK_WORDS];
    __shared__ unsigned hist[256];
    __shared__ int s_xy[8], s_bb[4];
    __shared__ double s_red[MS_THREADS];
    const int tid = threadIdx.x;
    int tb = L.tg_tobs[g], kf = L.tg_kf[g], j = L.tg_text[g], h = W.text_host[j];
    if (W.filter_good && !W.tobs_good[tb]) { if (tid == 0) { W.musig[2*tb] = 0; W.musig[2*tb+1] = 0; } return; }
    const int w = L.img_w, hh = L.img_h;
    __shared__ int s_c[16];
    if (tid < 4) {                                            // one box corner per lane (the serial walk over the four cost ~1.5 us of divisions)
        const int b = tid;
        Pose C; load_pose(pose + 7*kf, C);
        PairT T;
        if (h >= 0) { Pose Hs; load_pose(pose + 7*h, Hs); pair_from_poses(C, Hs, T); }
        else pair_from_Twr(C, W.text_Twr + 12*j, T);
        double th[3] = { theta[3*j], theta[3*j+1], theta[3*j+2] };
        double mx = W.text_box[(j*4 + b)*2], my = W.text_box[(j*4 + b)*2 + 1];
        double invz = -(mx*th[0] + my*th[1] + th[2]);
        double m[3] = { mx, my, 1.0 }, Rm[3]; mat3_vec(T.Rcr, m, Rm);
        double X = Rm[0]/invz + T.tq[0] + C.t[0], Y = Rm[1]/invz + T.tq[1] + C.t[1], Z = Rm[2]/invz + T.tq[2] + C.t[2];
        double cu = L.K[0]*X/Z + L.K[2], cv = L.K[1]*Y/Z + L.K[3];
        s_xy[2*b] = (int)cu; s_xy[2*b+1] = (int)cv;
        // the reference updates xMax / xMin only on strict improvement, starting from -1 / w + 1: a corner that does not improve
        // contributes nothing -- the same as taking max / min over the corners that do
        s_c[4*b] = cu > -1.0 ? (int)ceil(cu) : -1;            // candidate for xMax (initial value -1)
        s_c[4*b + 1] = cu < (double)(w + 1) ? (int)floor(cu) : w + 1;
        s_c[4*b + 2] = cv > -1.0 ? (int)ceil(cv) : -1;
        s_c[4*b + 3] = cv < (double)(hh + 1) ? (int)floor(cv) : hh + 1;
    }
    __syncthreads();
    if (tid == 0) {
        int xMax = max(max(s_c[0], s_c[4]), max(s_c[8], s_c[12])), xMin = min(min(s_c[1], s_c[5]), min(s_c[9], s_c[13]));
        int yMax = max(max(s_c[2], s_c[6]), max(s_c[10], s_c[14])), yMin = min(min(s_c[3], s_c[7]), min(s_c[11], s_c[15]));
        if (xMin < 0) xMin = 0;
        if (xMin >= w) xMin = w - 1;
        if (yMin < 0) yMin = 0;
        if (yMin >= hh) yMin = hh - 1;
        if (xMax >= w) xMax = w - 1;
        if (xMax < 0) xMax = 0;
        if (yMax >= hh) yMax = hh - 1;
        if (yMax < 0) yMax = 0;
        s_bb[0] = xMin; s_bb[1] = xMax; s_bb[2] = yMin; s_bb[3] = yMax;
    }
    for (int k = tid; k < min((w*hh + 31) >> 5, MS_MASK_WORDS); k += MS_THREADS) mask[k] = 0;
    hist[tid] = 0;
    __syncthreads();
    const int xMin = s_bb[0], xMax = s_bb[1], yMin = s_bb[2], yMax = s_bb[3];
    raster_quad(mask, s_xy, w, hh, tid, MS_THREADS);
    __syncthreads();
    // histogram of masked pixels inside the clamped bounding box (tool.cc:1217-1232)
    const uint8_t *img = L.img[kf];
    int bw = xMax - xMin + 1, bh = yMax - yMin + 1;
    {   // four pixels per thread and round with their loads in flight together; (x, y) advance without a division per pixel
        const int npx = bw*bh, dx = MS_THREADS % bw, dy = MS_THREADS / bw;
        int x = tid % bw, y = tid / bw;
        for (int k0 = tid; k0 < npx; k0 += 4*MS_THREADS) {
            int bit[4]; bool in[4]; unsigned px[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                bit[u] = (yMin + y)*w + xMin + x;
                in[u] = k0 + u*MS_THREADS < npx && (mask[bit[u] >> 5] & (1u << (bit[u] & 31)));
                x += dx; y += dy; if (x >= bw) { x -= bw; y++; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) px[u] = in[u] ? img[bit[u]] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) if (in[u]) atomicAdd(&hist[px[u]], 1u);
        }
    }
    __syncthreads();
    double cnt = (double)hist[tid], sum = (double)hist[tid]*(double)tid;
    double n = block_sum<MS_THREADS>(cnt, s_red), sm = block_sum<MS_THREADS>(sum, s_red);
    if (n < 2.0) { if (tid == 0) { W.musig[2*tb] = 0; W.musig[2*tb+1] = 0; } return; }
    double mu = sm/n;
    double d = (double)tid - mu;
    double ss = block_sum<MS_THREADS>((double)hist[tid]*d*d, s_red);
    if (tid == 0) { W.musig[2*tb] = mu; W.musig[2*tb+1] = sqrt(ss/(n - 1.0)); }
}

__global__ __launch_bounds__(MS_THREADS) void k_musigma(Work W, LevelDev L) {
    musigma_wg(W, L, blockIdx.x, W.pose[W.st->cur], W.theta[W.st->cur]);
}

// ---- text label image of one keyframe (optimizer::ShowBAReproj_TextBox -> tool::TextBoxWithFill, optimizer.cc:2508-2582,
// tool.cc:2103-2166): background -1, then every text observation of the keyframe in observation order fills its projected quad
// with its rank; later quads overwrite earlier ones, so ONE workgroup walks the observations sequentially (a keyframe sees a few
// dozen planes) and only the rasterisation of each quad is parallel.
#define LBL_THREADS 1024
__global__ __launch_bounds__(LBL_THREADS) void k_label(Work W, int kf, int w, int hh, double fx, double fy, double cx, double cy, float *out) {
    __shared__ unsigned mask[MS_MASK_WORDS];
    __shared__ int s_xy[8], s_bb[4];
    const int tid = threadIdx.x;
    const double *pose = W.pose[W.st->cur], *theta = W.theta[W.st->cur];
    for (int k = tid; k < w*hh; k += LBL_THREADS) out[k] = -1.0f;
    int rank = 0;
    for (int t = 0; t < W.n_tobs; t++) {
        if (W.tobs_kf[t] != kf) continue;                   // (uniform)
        const int j = W.tobs_text[t], h = W.text_host[j];
        if (tid == 0) {
            Pose C; load_pose(pose + 7*kf, C);
            PairT T;
            if (h >= 0) { Pose Hs; load_pose(pose + 7*h, Hs); pair_from_poses(C, Hs, T); }
            else pair_from_Twr(C, W.text_Twr + 12*j, T);
            const double th[3] = { theta[3*j], theta[3*j+1], theta[3*j+2] };
            int xMin = w, xMax = -1, yMin = hh, yMax = -1;
            for (int b = 0; b < 4; b++) {
                const double mx = W.text_box[(j*4 + b)*2], my = W.text_box[(j*4 + b)*2 + 1];
                const double invz = -(mx*th[0] + my*th[1] + th[2]);
                double m[3] = { mx, my, 1.0 }, Rm[3]; mat3_vec(T.Rcr, m, Rm);
                const double X = Rm[0]/invz + T.tq[0] + C.t[0], Y = Rm[1]/invz + T.tq[1] + C.t[1], Z = Rm[2]/invz + T.tq[2] + C.t[2];
                const double cu = fx*X/Z + cx, cv = fy*Y/Z + cy;
                const int iu = (int)cu, iv = (int)cv;                 // cv::Point(double, double): truncation
                s_xy[2*b] = iu; s_xy[2*b+1] = iv;
                xMin = min(xMin, iu); xMax = max(xMax, iu); yMin = min(yMin, iv); yMax = max(yMax, iv);
            }
            s_bb[0] = max(xMin, 0); s_bb[1] = min(xMax, w - 1); s_bb[2] = max(yMin, 0); s_bb[3] = min(yMax, hh - 1);
        }
        for (int k = tid; k < MS_MASK_WORDS; k += LBL_THREADS) mask[k] = 0;
        __syncthreads();
        raster_quad(mask, s_xy, w, hh, tid, LBL_THREADS);
        __syncthreads();
        const int x0 = s_bb[0], x1 = s_bb[1], y0 = s_bb[2], y1 = s_bb[3];     // the filled set lies inside the corners' bounding box
        const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
        if (bw > 0 && bh > 0)
            for (int k = tid; k < bw*bh; k += LBL_THREADS) {
                const int x = x0 + k % bw, y = y0 + k / bw, bit = y*w + x;
                if (mask[bit >> 5] & (1u << (bit & 31))) out[bit] = (float)rank;
            }
        rank++;
        __syncthreads();
    }
}

// ---- linearisation / cost.  grid = n_pair (scene waves) + n_tg (text waves), 64 threads each.
#define MODE_FULL 0
#define MODE_COST 1
// transpose-sum of N <= 28 per-lane values of ONE wave inside a two-wave workgroup: lane l (< N) returns the total of acc[l].
// reg: this wave's 28*65 doubles.  Both waves of the workgroup must call it (two workgroup barriers).
template <int N>
__device__ __forceinline__ double wave_sum_to_lane_mw(const double *acc, double *reg, int lane) {
#pragma unroll
    for (int i = 0; i < N; i++) reg[i*65 + lane] = acc[i];
    __syncthreads();
    double s = 0.0;
    if (lane < N) {
        const double *row = reg + lane*65;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int q = 0; q < 64; q += 4) { s0 += row[q]; s1 += row[q + 1]; s2 += row[q + 2]; s3 += row[q + 3]; }
        s = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    return s;
}
// the same for FOUR 16-lane groups per wave (small pairs share a wave): lane (group g, sub s) returns the group totals of acc[s] and
// acc[s + 16] (the latter only for s + 16 < N).  Both waves of the workgroup must call it.
template <int N>
__device__ __forceinline__ void wave_sum_groups16_mw(const double *acc, double *reg, int lane, double &t0, double &t1) {
#pragma unroll
    for (int i = 0; i < N; i++) reg[i*65 + lane] = acc[i];
    __syncthreads();
    const int g16 = lane & 48, sub = lane & 15;
    {
        const double *row = reg + sub*65 + g16;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int q = 0; q < 16; q += 4) { s0 += row[q]; s1 += row[q + 1]; s2 += row[q + 2]; s3 += row[q + 3]; }
        t0 = (s0 + s1) + (s2 + s3);
    }
    t1 = 0.0;
    if (sub + 16 < N) {
        const double *row = reg + (sub + 16)*65 + g16;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int q = 0; q < 16; q += 4) { s0 += row[q]; s1 += row[q + 1]; s2 += row[q + 2]; s3 += row[q + 3]; }
        t1 = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
}
// 128-thread workgroups: a scene workgroup takes two (target, host) pairs (one per wave) -- or, PPW = 4 for maps whose pairs hold a
// dozen scene blocks (thousands of keyframes: a 64-lane wave per pair was 85 % idle), eight pairs, one per 16-lane group; a text workgroup one (KF, text)
// observation with every feature on TWO lanes (4 taps each): the text lanes' instruction stream (~450 instructions per tap at
// one instruction per ~4.5 cycles) is what bounds the kernel.  <= 256 VGPRs so that all ~730 workgroups of C4 are resident at once.
#define LIN_TPL 4                        // photometric taps per lane: a feature's 8 taps sit on 8 / LIN_TPL neighbouring lanes.  4 = two lanes per
                                         // feature, 128-thread workgroups.  2 (four lanes per feature, 256 threads) was measured in round 2: the C4
                                         // level-0 launch went from 12.6 to 14.8 us -- the 55-value workgroup reduction is paid per wave, and
                                         // halving a lane's tap loop does not pay for twice the waves
#define LIN_T (64*(8/LIN_TPL))           // 64 features per text workgroup
#define LIN_NWV (LIN_T/64)
#define MID_U 4                          // slot records of a point that k_mid keeps in flight per round trip
// TEXT = false: levels without text planes (the reference's GlobalBA): the scene path alone needs far fewer registers than the text path.
template <int MODE, int PPW = 1, bool TEXT = true>
__global__ __launch_bounds__(LIN_T, TEXT ? 2 : 3) void k_linearize(Work W, LevelDev L, int spec) {
    // spec = 0: linearise at x (pass start); spec = 1: speculative linearisation at the LM candidate, into the other LinBuf
    const LmState *st = W.st;
    constexpr int NWV = LIN_T/64, TPL = LIN_TPL, LPF = 8/LIN_TPL;   // waves per workgroup; taps per lane; lanes per feature
    __shared__ double lds[NWV*28*65 + NWV*64];
    __shared__ unsigned s_px[TEXT ? TPL*LIN_T : 1];          // the text path's pixel quads (four bytes): indexed by tap at run time (not registers)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *reg = lds + wave*28*65, *xw = lds + NWV*28*65;
    // static indices of this workgroup first: in flight together with the LM state
    constexpr int LPP = 64/PPW;                              // lanes per pair
    const int nb_sc = (L.n_pair + NWV*PPW - 1)/(NWV*PPW);
    const int sub = PPW == 1 ? lane : (lane & (LPP - 1));
    // (scene-only launches: the workgroups of ONE XCD -- workgroup i runs on XCD i mod 8 -- take neighbouring pairs: the 64-byte slot records
    // of a landmark's observers share 128-byte lines, and neighbouring pairs observe the same landmarks.  With text groups the same mapping
    // was measured SLOWER on C4 (13.6 vs 13.0 us): the groups are the heavy workgroups and sit at the end of the index range -- contiguous
    // ranges put all of them on the last two XCDs.)  The grid is a multiple of 8 workgroups either way.
    // (Dispatching the text groups FIRST -- lowest workgroup indices -- was no better either: 13.2 us.)
    const int bq = TEXT ? (int)blockIdx.x : ((int)blockIdx.x & 7)*((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    if (bq >= nb_sc + (TEXT ? L.n_tg : 0)) return;
    const int pr = (NWV*bq + wave)*PPW + (PPW == 1 ? 0 : lane/LPP), prc = min(pr, max(L.n_pair - 1, 0));
    int pi = 0, ph = 0, pbeg = 0, pend = 0, tgpp = 0; int4 ra = {0, 0, 0, 0}, rb = {0, 0, 0, 0};
    if (bq < nb_sc) { pi = L.pair_i[prc]; ph = L.pair_h[prc]; pbeg = L.pair_sc_off[prc]; pend = pr < L.n_pair ? L.pair_sc_off[prc+1] : pbeg; }
    else if (TEXT) { ra = ((const int4 *)L.tg_rec)[2*(bq - nb_sc)]; rb = ((const int4 *)L.tg_rec)[2*(bq - nb_sc) + 1]; tgpp = L.tg_ppos[bq - nb_sc]; }   // one static record per group
    if (st->done) return;
    if (!spec && !st->need_lin) return;
    if (spec && st->step_fail) return;
    const int sel = spec ? (st->cur ^ 1) : st->cur;
    const LinBuf &B = W.lb[spec ? (st->lcur ^ 1) : st->lcur];
    const double *pose = W.pose[sel], *rho = W.rho[sel], *theta = W.theta[sel];
    const int b = bq;
    if (b < nb_sc) {
        // ---------------- scene observations of pair (i, h)
        const int i = pi, h = ph;
        Pose C; load_pose(pose + 7*i, C);
        PairT T;
        if (h >= 0) { Pose Hs; load_pose(pose + 7*h, Hs); pair_from_poses(C, Hs, T); }
        const bool fixed = (h < 0) && W.kf_const[i];           // all parameter blocks constant: not in the reduced program
        double acc[28];
#pragma unroll
        for (int k = 0; k < 28; k++) acc[k] = 0.0;
        const int beg = pbeg, end = pend;
#pragma unroll 1
        for (int c = beg + sub; c < end; c += LPP) {
            const int slot = L.sc_slot[c], pt = L.sc_pt[c];
            const bool act = !fixed && (!W.filter_good || W.sgood[L.sc_flag[c]]);
            // the slot record of an inactive candidate is zeros: one store sequence for both cases (a second, branchy one
            // costs the kernel ~170 VGPRs of live ranges)
            double wv[8];
#pragma unroll
            for (int k = 0; k < 8; k++) wv[k] = 0.0;
            if (act) {
                if (h < 0) pair_from_Trw(C, W.pt_Trw + 12*(size_t)pt, T);
                const double mx = W.pt_ray[2*pt], my = W.pt_ray[2*pt+1], rh = rho[pt];
                const double uo = L.sc_uv[2*c], vo = L.sc_uv[2*c+1];
                double r[2], jt[2][6], jl[2];
                scene_block(T, C.t, mx, my, rh, uo, vo, W.K0[0], W.K0[1], W.K0[2], W.K0[3], W.w_sx, W.w_sy, r, jt, jl);
                double wgt; acc[27] += 0.5*huber(r[0]*r[0] + r[1]*r[1], W.huber_s, wgt);
                int q = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int cc = a; cc < 6; cc++) { acc[q] += wgt*(jt[0][a]*jt[0][cc] + jt[1][a]*jt[1][cc]); q++; }
#pragma unroll
                for (int a = 0; a < 6; a++) acc[21 + a] += wgt*(jt[0][a]*r[0] + jt[1][a]*r[1]);
#pragma unroll
                for (int a = 0; a < 6; a++) wv[a] = wgt*(jt[0][a]*jl[0] + jt[1][a]*jl[1]);
                wv[6] = wgt*(jl[0]*jl[0] + jl[1]*jl[1]);
                wv[7] = wgt*(jl[0]*r[0] + jl[1]*r[1]);
            }
            if (slot >= 0) {                         // (the host column -Q^T w is a function of w and the pair's R_cr: k_mid forms it)
#pragma unroll
                for (int k = 0; k < 8; k++) B.w_pt[(size_t)(slot)*PT_REC + k] = wv[k];
            }
        }
        if (MODE == MODE_COST) {
            double cs = acc[27];
            if (PPW == 1) cs = wave_sum1(cs);
            else {
#pragma unroll
                for (int o = LPP/2; o > 0; o >>= 1) cs += __shfl_xor(cs, o, LPP);
            }
            if (sub == 0 && pr < L.n_pair) B.pairCost[pr] = cs;
        } else if (PPW > 1) {
            double t0, t1;
            wave_sum_groups16_mw<28>(acc, reg, lane, t0, t1);
            if (pr < L.n_pair) {
                B.pairM[(size_t)sub*L.n_pair + pr] = t0;                       // values 0 .. 15
                if (sub + 16 < 27) B.pairM[(size_t)(sub + 16)*L.n_pair + pr] = t1;
                else if (sub + 16 == 27) B.pairCost[pr] = t1;
                if (h >= 0 && sub == 0) {
#pragma unroll
                    for (int k = 0; k < 9; k++) B.pairR[(size_t)k*L.n_pair + pr] = T.Rcr[k];
                }
            }
        } else {
            double tot = wave_sum_to_lane_mw<28>(acc, reg, lane);
            if (pr < L.n_pair) {
                if (lane < 27) B.pairM[(size_t)lane*L.n_pair + pr] = tot;
                else if (lane == 27) B.pairCost[pr] = tot;
                if (h >= 0 && lane == 0) {
#pragma unroll
                    for (int k = 0; k < 9; k++) B.pairR[(size_t)k*L.n_pair + pr] = T.Rcr[k];
                }
            }
        }
    } else if constexpr (TEXT) {
        // ---------------- photometric blocks of one (KF, text) observation: thread = (feature tid / LPF, tap group tid % LPF)
        const int g = b - nb_sc;
        const int tb = ra.x, i = ra.y, j = ra.z, h = ra.w, slot = rb.x, f0 = rb.y, f1 = rb.z, fg = rb.w;
        const double mu = W.musig[2*tb], sigma = W.musig[2*tb+1];
        const bool act_g = (!W.filter_good || W.tobs_good[tb]) && !((h < 0) && W.kf_const[i]) && sigma != 0.0;
        // static data of this thread's first feature: fetched together with the level-2 operands, not after them
        const int fl = tid/LPF, tp = tid % LPF;
        int f = f0 + fl, raw = 0; double fu = 0.0, fv = 0.0, refv[TPL];
#pragma unroll
        for (int k = 0; k < TPL; k++) refv[k] = 0.0;
        if (f1 > f0) {
            const int fc = min(f, f1 - 1);
            raw = L.tfeat_raw[fc]; fu = L.tfeat_uv[2*fc]; fv = L.tfeat_uv[2*fc+1];
#pragma unroll
            for (int k = 0; k < TPL; k++) refv[k] = L.tfeat_ref[8*(size_t)fc + TPL*tp + k];
        }
        PairT T;
        // poses / plane / image pointer do not wait for the activity test (h is known from the record)
        Pose C; load_pose(pose + 7*i, C);
        if (h >= 0) { Pose Hs; load_pose(pose + 7*h, Hs); pair_from_poses(C, Hs, T); }
        else pair_from_Twr(C, W.text_Twr + 12*(size_t)j, T);
        const double th[3] = { theta[3*j], theta[3*j+1], theta[3*j+2] };
        const uint8_t *img = L.img[i];
        const double inv_sigma = 1.0/sigma;
        const double ifx = 1.0/L.K[0], ify = 1.0/L.K[1];
        double tot = 0.0;                       // lane l < 55 of wave 0: running total of value l
        // chunks of 64 features (one chunk unless the plane has more).  One accumulator set only: the weighted block of the
        // thread's half feature is reduced per chunk, so that it stays inside the architectural VGPRs
        for (int fb = f0; fb == f0 || fb < f1; fb += 64, f += 64) {
            double blk[55];
#pragma unroll
            for (int k = 0; k < 55; k++) blk[k] = 0.0;
            if (act_g) {                                               // (uniform; the shuffle below needs both lanes of a feature)
                const bool in = f < f1;
                if (fb != f0 && in) {
                    raw = L.tfeat_raw[f]; fu = L.tfeat_uv[2*f]; fv = L.tfeat_uv[2*f+1];
#pragma unroll
                    for (int k = 0; k < TPL; k++) refv[k] = L.tfeat_ref[8*(size_t)f + TPL*tp + k];
                }
                const uint8_t good = in ? (W.filter_good ? W.tfgood[fg + raw] : (uint8_t)1) : (uint8_t)0;   // in flight with the pixel fetches
                // the pixel-pair fetches of all the thread's taps in flight before the first residual; the quads wait in LDS so that
                // the residual loop can stay rolled (unrolled, its live state does not fit 256 VGPRs and spills to scratch)
#pragma unroll
                for (int k = 0; k < TPL; k++) {
                    const int kt = TPL*tp + k;
                    const double mx = (fu + TAP_DX[kt] - L.K[2])*ifx, my = (fv + TAP_DY[kt] - L.K[3])*ify;   // tool.cc:1561
                    const TapPx q = tap_fetch(T, C.t, th, mx, my, L.K[0], L.K[1], L.K[2], L.K[3], img, L.img_w, L.img_h);
                    s_px[k*LIN_T + tid] = (unsigned)q.I00 | ((unsigned)q.I01 << 8) | ((unsigned)q.I10 << 16) | ((unsigned)q.I11 << 24);
                }
                double s = 0.0;
#pragma unroll 1
                for (int k = 0; k < TPL; k++) {
                    const int kt = TPL*tp + k;
                    const double mx = (fu + TAP_DX[kt] - L.K[2])*ifx, my = (fv + TAP_DY[kt] - L.K[3])*ify;
                    const unsigned q4 = s_px[k*LIN_T + tid];
                    const TapPx pxk = { (int)(q4 & 0xff), (int)((q4 >> 8) & 0xff), (int)((q4 >> 16) & 0xff), (int)(q4 >> 24) };
                    double rf = refv[0];
#pragma unroll
                    for (int q = 1; q < TPL; q++) if (k == q) rf = refv[q];
                    double jt[6], jl[3];
                    double r = text_tap_px(T, C.t, th, mx, my, L.K[0], L.K[1], L.K[2], L.K[3], pxk, L.img_w, L.img_h,
                                           mu, sigma, inv_sigma, rf, W.w_t, true, jt, jl);
                    s += r*r;
                    int q = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++)
#pragma unroll
                        for (int cc = a; cc < 6; cc++) { blk[q] += jt[a]*jt[cc]; q++; }
#pragma unroll
                    for (int a = 0; a < 6; a++) blk[21 + a] += jt[a]*r;
#pragma unroll
                    for (int a = 0; a < 6; a++)
#pragma unroll
                        for (int cc = 0; cc < 3; cc++) blk[27 + a*3 + cc] += jt[a]*jl[cc];
                    blk[45] += jl[0]*jl[0]; blk[46] += jl[0]*jl[1]; blk[47] += jl[0]*jl[2];
                    blk[48] += jl[1]*jl[1]; blk[49] += jl[1]*jl[2]; blk[50] += jl[2]*jl[2];
                    blk[51] += jl[0]*r; blk[52] += jl[1]*r; blk[53] += jl[2]*r;
                }
                double s8 = s;                                      // the block's squared norm: its 8 taps sit on LPF neighbouring lanes
#pragma unroll
                for (int q = 1; q < LPF; q <<= 1) s8 += __shfl_xor(s8, q, 64);
                double wgt; const double rho_h = 0.5*huber(s8, W.huber_t, wgt);
                const double wg = good ? wgt : 0.0;
#pragma unroll
                for (int k = 0; k < 54; k++) blk[k] *= wg;
                blk[54] = (good && tp == 0) ? rho_h : 0.0;
            }
            // 55 sums over the workgroup's threads: per wave two transposes (28 + 27 values), then the waves (fixed order)
            const double t0 = wave_sum_to_lane_mw<28>(blk, reg, lane);
            const double t1 = wave_sum_to_lane_mw<27>(blk + 28, reg, lane);
            if (lane < 28) xw[wave*64 + lane] = t0;
            if (lane < 27) xw[wave*64 + 28 + lane] = t1;
            __syncthreads();
            if (lane < 55) { double part = xw[lane];
#pragma unroll
                for (int q = 1; q < NWV; q++) part += xw[q*64 + lane];
                tot += part; }
            __syncthreads();
        }
        if (wave > 0) return;
        // wave 0 alone from here (LDS accesses of one wave are ordered; the fence keeps the compiler honest)
        if (lane < 27) B.tgM[(size_t)lane*L.n_tg + tgpp] = tot;          // pair-major rank: k_mid sums a contiguous range
        else if (lane < 45) { if (slot >= 0) B.w_tx[(size_t)(slot)*TX_REC + (lane - 27)] = tot; }
        else if (lane < 54) { if (slot >= 0) B.w_tx[(size_t)(slot)*TX_REC + 18 + (lane - 45)] = tot; }
        else if (lane == 54) B.tgCost[tgpp] = tot;                       // pair-major rank as well: k_mid adds it to its pair's cost
        // (the host column of W, -blkdiag(R,R)^T W, is formed by k_mid from W and the pair's R_cr; an inactive group leaves W = 0)
    }
}

// ---- per landmark: V, b, host column of W (= -sum Q^T w);  per pair: host-side products.  256-thread blocks.
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
__global__ __launch_bounds__(256) void k_mid(Work W, LevelDev L, int nb_pt, int nb_tx, int spec) {
    const LmState *st = W.st;
    const int b = blockIdx.x;
    // static offsets of this thread's landmark / pair first: in flight together with the LM state
    int o = 0, e = 0, tq0 = 0, tq1 = 0, ph_ = -1, hp_ = -1, act_ = 0, pr0[MID_U] = {0, 0, 0, 0};
    if (b < nb_pt) { const int j = b*256 + threadIdx.x; if (j < W.n_pt) { o = L.pls_off[j]; e = L.pls_off[j+1]; act_ = W.act_pt[j];
#pragma unroll
        for (int u = 0; u < MID_U; u++) pr0[u] = L.pt_pair4[MID_U*(size_t)j + u]; } }
    else if (b < nb_pt + nb_tx) { const int j = (b - nb_pt)*256 + threadIdx.x; if (j < W.n_text) { o = L.tls_off[j]; e = L.tls_off[j+1]; act_ = W.act_tx[j]; } }
    else { const int p = (b - nb_pt - nb_tx)*256 + threadIdx.x; if (p < L.n_pair) { tq0 = L.pair_tg_off[p]; tq1 = L.pair_tg_off[p+1]; ph_ = L.pair_h[p]; hp_ = L.pair_hpos[p]; } }
    if (st->done) return;
    if (!spec && !st->need_lin) return;
    if (spec && st->step_fail) return;
    const LinBuf &B = W.lb[spec ? (st->lcur ^ 1) : st->lcur];
    const int sel = spec ? (st->cur ^ 1) : st->cur;
    __shared__ double red[256];
    const double *rho_x = W.rho[sel], *theta_x = W.theta[sel];
    const size_t np_ = L.n_pair;
    double gm = 0.0, xn = 0.0, cs = 0.0;                        // cs: cost of this thread's pair and of its text groups
    if (b < nb_pt) {
        const int j = b*256 + threadIdx.x;
        if (e > o) {
            double acc[8] = {0,0,0,0,0,0,0,0};                       // V, b, host column -sum Q^T w
            for (int s0 = o; s0 < e - 1; s0 += MID_U) {              // MID_U slot records (and their pairs' R_cr) in flight per round trip
                int pr[MID_U]; double v[MID_U][8], R[MID_U][9];
#pragma unroll
                for (int u = 0; u < MID_U; u++) pr[u] = s0 == o ? pr0[u] : L.pslot_pair[min(s0 + u, e - 2)];
#pragma unroll
                for (int u = 0; u < MID_U; u++) {
#pragma unroll
                    for (int k = 0; k < 8; k++) v[u][k] = B.w_pt[(size_t)(min(s0 + u, e - 2))*PT_REC + k];
#pragma unroll
                    for (int k = 0; k < 9; k++) R[u][k] = B.pairR[(size_t)k*np_ + pr[u]];
                }
#pragma unroll
                for (int u = 0; u < MID_U; u++) if (s0 + u < e - 1) {
                    double qa[3], qc[3]; mat3T_vec(R[u], v[u], qa); mat3T_vec(R[u], v[u] + 3, qc);
                    acc[0] += v[u][6]; acc[1] += v[u][7];
#pragma unroll
                    for (int a = 0; a < 3; a++) { acc[2 + a] += -qa[a]; acc[5 + a] += -qc[a]; }
                }
            }
#pragma unroll
            for (int k = 0; k < 6; k++) B.w_pt[(size_t)(e - 1)*PT_REC + k] = acc[2 + k];
            const double V = acc[0];
            B.V_pt[j] = V; B.b_pt[j] = acc[1];
            if (st->first) W.sig_pt[j] = 1.0/(1.0 + sqrt(V));
            const double sg = W.sig_pt[j];
            B.dgs_pt[j] = clampd(sg*sg*V, W.min_diag, W.max_diag)/(sg*sg);
            if (act_) { gm = fabs(acc[1]); xn = rho_x[j]*rho_x[j]; }
        }
    } else if (b < nb_pt + nb_tx) {
        const int j = (b - nb_pt)*256 + threadIdx.x;
        if (e > o) {
            double acc[27];                                          // V6, b3, host column -blkdiag(R,R)^T W (18)
#pragma unroll
            for (int k = 0; k < 27; k++) acc[k] = 0.0;
            for (int s0 = o; s0 < e - 1; s0 += 2) {                  // 2 slot records in flight per round trip
                int pr[2]; double v[2][27], R[2][9];
#pragma unroll
                for (int u = 0; u < 2; u++) pr[u] = L.tslot_pair[min(s0 + u, e - 2)];
#pragma unroll
                for (int u = 0; u < 2; u++) {
#pragma unroll
                    for (int k = 0; k < 27; k++) v[u][k] = B.w_tx[(size_t)(min(s0 + u, e - 2))*TX_REC + k];
#pragma unroll
                    for (int k = 0; k < 9; k++) R[u][k] = B.pairR[(size_t)k*np_ + pr[u]];
                }
#pragma unroll
                for (int u = 0; u < 2; u++) if (s0 + u < e - 1) {
#pragma unroll
                    for (int k = 0; k < 9; k++) acc[k] += v[u][18 + k];
#pragma unroll
                    for (int half = 0; half < 2; half++)
#pragma unroll
                        for (int rr = 0; rr < 3; rr++)
#pragma unroll
                            for (int cc = 0; cc < 3; cc++)
                                acc[9 + (half*3 + rr)*3 + cc] += -(R[u][0*3 + rr]*v[u][(half*3 + 0)*3 + cc] + R[u][1*3 + rr]*v[u][(half*3 + 1)*3 + cc] + R[u][2*3 + rr]*v[u][(half*3 + 2)*3 + cc]);
                }
            }
#pragma unroll
            for (int k = 0; k < 18; k++) B.w_tx[(size_t)(e - 1)*TX_REC + k] = acc[9 + k];
#pragma unroll
            for (int k = 0; k < 6; k++) B.V_tx[(size_t)k*W.n_text + j] = acc[k];
#pragma unroll
            for (int k = 0; k < 3; k++) B.b_tx[(size_t)k*W.n_text + j] = acc[6 + k];
            const double dv[3] = { acc[0], acc[3], acc[5] };
#pragma unroll
            for (int k = 0; k < 3; k++) {
                if (st->first) W.sig_tx[(size_t)k*W.n_text + j] = 1.0/(1.0 + sqrt(dv[k]));
                const double sg = W.sig_tx[(size_t)k*W.n_text + j];
                B.dgs_tx[(size_t)k*W.n_text + j] = clampd(sg*sg*dv[k], W.min_diag, W.max_diag)/(sg*sg);
            }
            if (act_) for (int k = 0; k < 3; k++) { gm = fmax(gm, fabs(acc[6 + k])); xn += theta_x[3*j + k]*theta_x[3*j + k]; }
        }
    } else {
        const int p = (b - nb_pt - nb_tx)*256 + threadIdx.x;
        if (p < L.n_pair) {
            double M[21], c[6];
#pragma unroll
            for (int k = 0; k < 21; k++) M[k] = B.pairM[(size_t)k*L.n_pair + p];
#pragma unroll
            for (int k = 0; k < 6; k++) c[k] = B.pairM[(size_t)(21 + k)*L.n_pair + p];
            cs = B.pairCost[p];
            for (int q = tq0; q < tq1; q++) {          // (stored in pair-major order by k_linearize)
#pragma unroll
                for (int k = 0; k < 21; k++) M[k] += B.tgM[(size_t)k*L.n_tg + q];
#pragma unroll
                for (int k = 0; k < 6; k++) c[k] += B.tgM[(size_t)(21 + k)*L.n_tg + q];
                cs += B.tgCost[q];
            }
            double *out = B.pairOut;      // [90][n_pair]: M(21) c(6) MQ(36) by pair | QMQ(21) Qc(6) by host-major rank
#pragma unroll
            for (int k = 0; k < 21; k++) out[(size_t)k*L.n_pair + p] = M[k];
#pragma unroll
            for (int k = 0; k < 6; k++) out[(size_t)(21 + k)*L.n_pair + p] = c[k];
            if (ph_ >= 0) {
                const int hp = hp_;                        // rows 63..89 are stored host-major
                double R[9];
#pragma unroll
                for (int k = 0; k < 9; k++) R[k] = B.pairR[(size_t)k*L.n_pair + p];
                double Mf[36];
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int cc = 0; cc < 6; cc++) Mf[r*6 + cc] = M[sym6(r, cc)];
                double MQ[36];                         // M * blkdiag(R,R)
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int half = 0; half < 2; half++)
#pragma unroll
                        for (int cc = 0; cc < 3; cc++)
                            MQ[r*6 + half*3 + cc] = Mf[r*6 + half*3]*R[cc] + Mf[r*6 + half*3 + 1]*R[3 + cc] + Mf[r*6 + half*3 + 2]*R[6 + cc];
#pragma unroll
                for (int k = 0; k < 36; k++) out[(size_t)(27 + k)*L.n_pair + p] = MQ[k];
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int cc = r; cc < 6; cc++) {
                        const int hr = r/3, rr = r % 3;
                        double v = R[0*3 + rr]*MQ[(hr*3 + 0)*6 + cc] + R[1*3 + rr]*MQ[(hr*3 + 1)*6 + cc] + R[2*3 + rr]*MQ[(hr*3 + 2)*6 + cc];
                        out[(size_t)(63 + sym6(r, cc))*L.n_pair + hp] = v;
                    }
                double a[3], d[3]; mat3T_vec(R, c, a); mat3T_vec(R, c + 3, d);
                out[(size_t)84*L.n_pair + hp] = a[0]; out[(size_t)85*L.n_pair + hp] = a[1]; out[(size_t)86*L.n_pair + hp] = a[2];
                out[(size_t)87*L.n_pair + hp] = d[0]; out[(size_t)88*L.n_pair + hp] = d[1]; out[(size_t)89*L.n_pair + hp] = d[2];
            }
        }
    }
    gm = block_max<256>(gm, red); xn = block_sum<256>(xn, red);
    if (b >= nb_pt + nb_tx) cs = block_sum<256>(cs, red);       // (uniform) the cost as per-block partials: k_postlin / k_decide add a few hundred
                                                                // numbers instead of walking 40 k pairs at 5000 keyframes (50 us of one workgroup)
    if (threadIdx.x == 0) { B.lmpart[3*b] = gm; B.lmpart[3*b + 1] = xn; B.lmpart[3*b + 2] = cs; }
}

// ---- after a linearisation (256 threads of one block), in two stages so that a multi-GPU run can all-reduce in between:
//   sums_local : pose diagonal / gradient from the pair sums (-> B.Hd, B.bp), landmark gradient max / |x|^2, cost
//   pose_scale : Jacobi scaling, LM diagonal, gradient max and |x|^2 of the free poses (from the possibly all-reduced Hd / bp)
__device__ void sums_local(const Work &W, const LevelDev &L, const LinBuf &B, double *dHd, double *dbp, int nb_lm, double *red,
                           double &gmax_lm, double &xn_lm, double &cost, bool skip_pose = false) {
    const int tid = threadIdx.x;
    gmax_lm = 0.0; xn_lm = 0.0; cost = 0.0;
    const double *out = B.pairOut;
    const size_t np = L.n_pair;
    for (int task = tid; task < (skip_pose ? 0 : 12*W.n_kf); task += 256) {          // (pose, component): diag H (6) | b (6)   (large maps: k_pose_sums_raw did it)
        const int a = task/12, k = task - 12*a;
        const int t0 = L.pose_t_off[a], t1 = L.pose_t_off[a+1], h0 = L.pose_h_off[a], h1 = L.pose_h_off[a+1];
        if (k < 6) {
            const double h = range_sum<24>(out + (size_t)sym6(k, k)*np, t0, t1) + range_sum<24>(out + (size_t)(63 + sym6(k, k))*np, h0, h1);
            dHd[6*a + k] = h;
        } else {
            const double g = range_sum<24>(out + (size_t)(21 + k - 6)*np, t0, t1) - range_sum<24>(out + (size_t)(84 + k - 6)*np, h0, h1);
            dbp[6*a + k - 6] = g; B.bp_loc[6*a + k - 6] = g;
        }
    }
    __threadfence_block();                                             // pose_scale reads these through other threads
    for (int k = tid; k < nb_lm; k += 256) { gmax_lm = fmax(gmax_lm, B.lmpart[3*k]); xn_lm += B.lmpart[3*k + 1]; cost += B.lmpart[3*k + 2]; }
    gmax_lm = block_max<256>(gmax_lm, red); xn_lm = block_sum<256>(xn_lm, red); cost = block_sum<256>(cost, red);
}
__device__ void pose_scale(const Work &W, const LinBuf &B, const double *sHd, const double *sbp, const double *pose, bool first,
                           double *red, double &gmax_p, double &xn_p) {
    const int tid = threadIdx.x;
    gmax_p = 0.0; xn_p = 0.0;
    for (int a = tid; a < W.n_kf; a += 256) {
        const bool fre = W.fidx[a] >= 0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const double h = sHd[6*a + k], g = sbp[6*a + k];
            B.Hd[6*a + k] = h; B.bp[6*a + k] = g;                 // (multi-GPU: the all-reduced values replace the local ones)
            if (first) W.sig_p[6*a + k] = 1.0/(1.0 + sqrt(h));
            const double sg = W.sig_p[6*a + k];
            B.dgs_p[6*a + k] = clampd(sg*sg*h, W.min_diag, W.max_diag)/(sg*sg);
            if (fre) gmax_p = fmax(gmax_p, fabs(g));
        }
        if (fre) for (int k = 0; k < 7; k++) xn_p += pose[7*a + k]*pose[7*a + k];
    }
    gmax_p = block_max<256>(gmax_p, red); xn_p = block_sum<256>(xn_p, red);
}
// Single-GPU path of k_postlin / k_decide: everything one linearisation contributes to the LM decision, with the independent
// loads of all parts issued before the first wait and ONE five-value block reduction (a global round trip from this lone
// workgroup costs ~0.6 us, a block reduction ~0.3 us: the old sequence had a dozen of the former and seven of the latter).
//   out5 = { max |gradient|, |x|^2, cost, step^2 (nb_back partials), model cost change (nb_back partials) }   (thread 0)
__device__ void postlin_fused(const Work &W, const LevelDev &L, const LinBuf &B, const double *pose, bool first, int nb_lm, int nb_back,
                              double *red /*[5*256]*/, double *xch /*[252]*/, double out5[5], int npp = 0) {
    const int tid = threadIdx.x;
    double gmax = 0.0, xn = 0.0, cost = 0.0, step2 = 0.0, mcc = 0.0;
#ifdef TSBA_SOLVE_STAMPS
    long long q0_ = clock64(), q1_ = 0, q2_ = 0, q3_ = 0, q4_ = 0;
#endif
    {   // landmark / cost / step partials (one entry per k_mid / k_back workgroup): three per thread in flight, the (rare) rest in a plain loop
        double lc[3], lg[3], lx[3], ps[3], pm[3];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int k = tid + 256*u;
            lg[u] = B.lmpart[3*min(k, max(nb_lm - 1, 0))]; lx[u] = B.lmpart[3*min(k, max(nb_lm - 1, 0)) + 1]; lc[u] = B.lmpart[3*min(k, max(nb_lm - 1, 0)) + 2];
            ps[u] = W.partial[2*min(k, max(nb_back - 1, 0))]; pm[u] = W.partial[2*min(k, max(nb_back - 1, 0)) + 1];
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int k = tid + 256*u;
            if (k < nb_lm) { gmax = fmax(gmax, lg[u]); xn += lx[u]; cost += lc[u]; }
            if (k < nb_back) { step2 += ps[u]; mcc += pm[u]; }
        }
        for (int k = tid + 768; k < nb_lm; k += 256) { gmax = fmax(gmax, B.lmpart[3*k]); xn += B.lmpart[3*k + 1]; cost += B.lmpart[3*k + 2]; }
        for (int k = tid + 768; k < nb_back; k += 256) { step2 += W.partial[2*k]; mcc += W.partial[2*k + 1]; }
    }
#ifdef TSBA_SOLVE_STAMPS
    q1_ = clock64();
#endif
    // poses, 21 per round: thread (pose, component) sums one entry of diag(H_pp) (6) or of the gradient (6) over the pose's
    // pairs -- target side by pair, host side host-major, both contiguous -- then the six diag threads finish the pose
    const double *out = B.pairOut; const size_t np = L.n_pair;
    // (large maps: k_pose_sums did the per-pose work on many workgroups; only its partials are left to add)
    for (int k = tid; k < npp; k += 256) { gmax = fmax(gmax, W.posepart[2*k]); xn += W.posepart[2*k + 1]; }
    for (int a0 = 0; a0 < (npp > 0 ? 0 : W.n_kf); a0 += 21) {
        const int al = tid/12, k = tid - 12*al, a = a0 + al;
        const bool on = tid < 252 && a < W.n_kf;
        const int ac = min(a, W.n_kf - 1);
        const int t0 = L.pose_t_off[ac], t1 = L.pose_t_off[ac+1], h0 = L.pose_h_off[ac], h1 = L.pose_h_off[ac+1];
        const int kk = k < 6 ? k : k - 6;
        const double sgp = first ? 0.0 : W.sig_p[6*ac + kk]; const int fre = W.fidx[ac];
        const double px = pose[7*ac + kk], px6 = pose[7*ac + 6];
        const double *rt = out + (size_t)(k < 6 ? sym6(kk, kk) : 21 + kk)*np, *rh = out + (size_t)(k < 6 ? 63 + sym6(kk, kk) : 84 + kk)*np;
        const double vt = range_sum<24>(rt, t0, t1), vh = range_sum<24>(rh, h0, h1);
        const double val = k < 6 ? vt + vh : vt - vh;
#ifdef TSBA_SOLVE_STAMPS
        q2_ = clock64();
#endif
        if (on) xch[tid] = val;
        __syncthreads();
        if (on && k < 6) {
            const double h = val, g = xch[tid + 6];
            B.Hd[6*a + k] = h; B.bp[6*a + k] = g; B.bp_loc[6*a + k] = g;
            double sg = sgp;
            if (first) { sg = 1.0/(1.0 + sqrt(h)); W.sig_p[6*a + k] = sg; }
            B.dgs_p[6*a + k] = clampd(sg*sg*h, W.min_diag, W.max_diag)/(sg*sg);
            if (fre >= 0) { gmax = fmax(gmax, fabs(g)); xn += px*px + (k == 0 ? px6*px6 : 0.0); }
        }
        __syncthreads();
    }
#ifdef TSBA_SOLVE_STAMPS
    q3_ = clock64();
#endif
    // one reduction for the five values: wave w reduces value w, wave 0 also value 4
    red[tid] = gmax; red[256 + tid] = xn; red[512 + tid] = cost; red[768 + tid] = step2; red[1024 + tid] = mcc;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    auto reduce_one = [&](int v) -> double {
        const double *r = red + 256*v;
        double x;
        if (v == 0) {
            x = fmax(fmax(r[lane], r[lane + 64]), fmax(r[lane + 128], r[lane + 192]));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o, 64));
        } else { x = (r[lane] + r[lane + 64]) + (r[lane + 128] + r[lane + 192]); x = wave_sum1(x); }
        return x;
    };
    const double x0 = reduce_one(wave), x4 = wave == 0 ? reduce_one(4) : 0.0;
    __syncthreads();
    if (lane == 0) { red[256*wave] = x0; if (wave == 0) red[1024] = x4; }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 5; v++) out5[v] = red[256*v];
#ifdef TSBA_SOLVE_STAMPS
    q4_ = clock64();
    if (tid == 0) { W.dbg[40] = q1_ - q0_; W.dbg[41] = q2_ - q1_; W.dbg[42] = q3_ - q2_; W.dbg[43] = q4_ - q3_; }
#endif
}
// The pose part of postlin_fused for large maps (hundreds of keyframes and more): 21 poses per workgroup instead of 21 per ROUND of
// the single postlin / decide workgroup (238 rounds, 1.1 ms per LM iteration at 5000 keyframes).
__global__ __launch_bounds__(256) void k_pose_sums(Work W, LevelDev L, int spec) {
    const LmState *st = W.st;
    if (st->done) return;
    if (!spec && !st->need_lin) return;
    if (spec && st->step_fail) return;
    __shared__ double xch[256], red[256];
    const LinBuf &B = W.lb[spec ? (st->lcur ^ 1) : st->lcur];
    const double *pose = W.pose[spec ? (st->cur ^ 1) : st->cur];
    const bool first = !spec && st->first != 0;
    const int tid = threadIdx.x;
    const double *out = B.pairOut; const size_t np = L.n_pair;
    const int al = tid/12, k = tid - 12*al, a = blockIdx.x*21 + al;
    const bool on = tid < 252 && a < W.n_kf;
    const int ac = min(a, W.n_kf - 1);
    const int t0 = L.pose_t_off[ac], t1 = L.pose_t_off[ac+1], h0 = L.pose_h_off[ac], h1 = L.pose_h_off[ac+1];
    const int kk = k < 6 ? k : k - 6;
    const double sgp = first ? 0.0 : W.sig_p[6*ac + kk]; const int fre = W.fidx[ac];
    const double px = pose[7*ac + kk], px6 = pose[7*ac + 6];
    const double *rt = out + (size_t)(k < 6 ? sym6(kk, kk) : 21 + kk)*np, *rh = out + (size_t)(k < 6 ? 63 + sym6(kk, kk) : 84 + kk)*np;
    const double vt = range_sum<24>(rt, t0, t1), vh = range_sum<24>(rh, h0, h1);
    const double val = k < 6 ? vt + vh : vt - vh;
    if (on) xch[tid] = val;
    __syncthreads();
    double gmax = 0.0, xn = 0.0;
    if (on && k < 6) {
        const double h = val, g = xch[tid + 6];
        B.Hd[6*a + k] = h; B.bp[6*a + k] = g; B.bp_loc[6*a + k] = g;
        double sg = sgp;
        if (first) { sg = 1.0/(1.0 + sqrt(h)); W.sig_p[6*a + k] = sg; }
        B.dgs_p[6*a + k] = clampd(sg*sg*h, W.min_diag, W.max_diag)/(sg*sg);
        if (fre >= 0) { gmax = fabs(g); xn = px*px + (k == 0 ? px6*px6 : 0.0); }
    }
    gmax = block_max<256>(gmax, red); xn = block_sum<256>(xn, red);
    if (tid == 0) { W.posepart[2*blockIdx.x] = gmax; W.posepart[2*blockIdx.x + 1] = xn; }
}
// The same in a sharded (multi-GPU) run, in two stages around the all-reduce of the exchange buffer cb = [Hd | bp | scalars]:
//   k_pose_sums_raw    this rank's part of diag(H_pp) and of the pose gradient, 21 poses per workgroup  -> cb, B.bp_loc
//   k_pose_scale_multi from the all-reduced cb: B.Hd / B.bp, Jacobi scale (first linearisation), LM diagonal, per-workgroup partials of
//                      the gradient maximum and |x|^2 of the free poses -> W.posepart
// (one workgroup walking 5000 poses cost 0.7 ms per linearisation: more than everything the sharding saves)
__global__ __launch_bounds__(256) void k_pose_sums_raw(Work W, LevelDev L, int spec) {
    const LmState *st = W.st;
    if (st->done) return;
    if (!spec && !st->need_lin) return;
    if (spec && st->step_fail) return;
    const LinBuf &B = W.lb[spec ? (st->lcur ^ 1) : st->lcur];
    const int tid = threadIdx.x;
    const double *out = B.pairOut; const size_t np = L.n_pair;
    const int al = tid/12, k = tid - 12*al, a = blockIdx.x*21 + al;
    if (tid >= 252 || a >= W.n_kf) return;
    const int t0 = L.pose_t_off[a], t1 = L.pose_t_off[a+1], h0 = L.pose_h_off[a], h1 = L.pose_h_off[a+1];
    const int kk = k < 6 ? k : k - 6;
    const double *rt = out + (size_t)(k < 6 ? sym6(kk, kk) : 21 + kk)*np, *rh = out + (size_t)(k < 6 ? 63 + sym6(kk, kk) : 84 + kk)*np;
    const double vt = range_sum<24>(rt, t0, t1), vh = range_sum<24>(rh, h0, h1);
    if (k < 6) W.cb[6*a + kk] = vt + vh;
    else { const double g = vt - vh; W.cb[W.N + 6*a + kk] = g; B.bp_loc[6*a + kk] = g; }
}
__global__ __launch_bounds__(256) void k_pose_scale_multi(Work W, int spec) {
    const LmState *st = W.st;
    if (st->done) return;
    if (!spec && !st->need_lin) return;
    if (spec && st->step_fail) return;
    __shared__ double red[256];
    const LinBuf &B = W.lb[spec ? (st->lcur ^ 1) : st->lcur];
    const double *pose = W.pose[spec ? (st->cur ^ 1) : st->cur];
    const bool first = !spec && st->first != 0;
    const int tid = threadIdx.x, al = tid/6, k = tid - 6*al, a = blockIdx.x*21 + al;
    double gmax = 0.0, xn = 0.0;
    if (tid < 126 && a < W.n_kf) {
        const double h = W.cb[6*a + k], g = W.cb[W.N + 6*a + k];
        B.Hd[6*a + k] = h; B.bp[6*a + k] = g;
        double sg;
        if (first) { sg = 1.0/(1.0 + sqrt(h)); W.sig_p[6*a + k] = sg; } else sg = W.sig_p[6*a + k];
        B.dgs_p[6*a + k] = clampd(sg*sg*h, W.min_diag, W.max_diag)/(sg*sg);
        if (W.fidx[a] >= 0) { gmax = fabs(g); const double px = pose[7*a + k]; xn = px*px; if (k == 0) { const double p6 = pose[7*a + 6]; xn += p6*p6; } }
    }
    gmax = block_max<256>(gmax, red); xn = block_sum<256>(xn, red);
    if (tid == 0) { W.posepart[2*blockIdx.x] = gmax; W.posepart[2*blockIdx.x + 1] = xn; }
}
// the pose part of k_postlin / k_decide in a sharded run on a large map: the partials k_pose_scale_multi left
__device__ void pose_parts_multi(const Work &W, int npp, double *red, double &gmax_p, double &xn_p) {
    gmax_p = 0.0; xn_p = 0.0;
    for (int k = threadIdx.x; k < npp; k += 256) { gmax_p = fmax(gmax_p, W.posepart[2*k]); xn_p += W.posepart[2*k + 1]; }
    gmax_p = block_max<256>(gmax_p, red); xn_p = block_sum<256>(xn_p, red);
}
__global__ __launch_bounds__(256) void k_postlin(Work W, LevelDev L, double grad_tol, int nb_lm, int multi, int npp) {
    LmState *st = W.st;
    if (st->done || !st->need_lin) return;
    __shared__ double red[5*256], xch[256];
    double gmax, xn, cost;
    const LinBuf &B = W.lb[st->lcur];
    if (!multi) { double o5[5]; postlin_fused(W, L, B, W.pose[st->cur], st->first != 0, nb_lm, 0, red, xch, o5, npp); gmax = o5[0]; xn = o5[1]; cost = o5[2]; }
    else { double gp, xp;
           if (npp) pose_parts_multi(W, npp, red, gp, xp); else pose_scale(W, B, W.cb, W.cb + W.N, W.pose[st->cur], st->first != 0, red, gp, xp);
           const double *sc = W.cb + 2*(size_t)W.N; cost = sc[0]; xn = sc[1] + xp; gmax = fmax(W.cbm[0], gp); }
    if (threadIdx.x == 0) {
        st->x_cost = cost; st->x_norm = sqrt(xn); st->gmax = gmax;
        if (st->first) st->cost0 = cost;
        st->first = 0; st->need_lin = 0; st->n_lin++;
        if (gmax <= grad_tol) { st->done = 1; st->term = 3; }
        if (W.hprog) { *W.hprog = ((unsigned long long)W.pass_seq << 32) | ((unsigned long long)st->it << 1) | (st->done ? 1u : 0u); __threadfence_system(); }
    }
}

// ---- reduced camera system.  grid = n_sb (one workgroup per 6x6 block) + n_kf (reduced gradient), 256 threads.
// The (slot, slot, landmark) gather lists of a diagonal block hold ~1000 entries: four waves, and per wave the indices and
// operands of four entries in flight before the first multiply (two dependent global round trips per 1024 entries).
#define SCHUR_U 4
// SCHUR_NW waves per workgroup: 4 for windows (a diagonal block gathers ~1000 slot pairs), 1 for large maps (54 k blocks of ~60 slot
// pairs each at 5000 keyframes: three idle waves per block and their hand-off were most of the 0.64 ms)
template <int SCHUR_NW>
__global__ __launch_bounds__(64*SCHUR_NW) void k_schur_t(Work W, LevelDev L, int multi, int b0) {
    constexpr int SCHUR_T = 64*SCHUR_NW;
    LmState *st = W.st;
    if (st->done) return;
    __shared__ double lds[SCHUR_NW > 1 ? 3*36*64 : 36*65];     // waves 1..3 hand their partial blocks to wave 0, which then transposes (36*65 <= 3*36*64)
    // (b0 > 0: large maps take the S blocks through k_schur_quad and only the gradient part here, a grid of a multiple of 8 workgroups in
    // which the workgroups of ONE XCD -- workgroup i runs on XCD i mod 8 -- take neighbouring poses: the slot records of a landmark sit next
    // to each other, one per observing pose, and neighbouring poses observe the same landmarks)
    const int bx = b0 > 0 ? ((int)blockIdx.x & 7)*((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int b = bx + b0, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double radius = st->radius, irad = 1.0/radius;
    const LinBuf &B = W.lb[st->lcur];
    if (b < L.n_sb) {
        const int a = L.sb_a[b], c = L.sb_b[b];
        const int ia = W.fidx[a], ic = W.fidx[c];           // rows / columns of S exist for free poses only
        if (ia < 0 || ic < 0) return;
        double acc[36];
#pragma unroll
        for (int k = 0; k < 36; k++) acc[k] = 0.0;
        const int pt0 = L.sb_pt_off[b], pt1 = L.sb_pt_off[b+1];
        for (int base = pt0; base < pt1; base += SCHUR_T*SCHUR_U) {
            int s1[SCHUR_U], s2[SCHUR_U], j[SCHUR_U]; bool ok[SCHUR_U];
#pragma unroll
            for (int u = 0; u < SCHUR_U; u++) {
                const int q = base + u*SCHUR_T + tid; ok[u] = q < pt1;
                const int qc = min(q, pt1 - 1);
                s1[u] = L.sb_pt_s1[qc]; s2[u] = L.sb_pt_s2[qc]; j[u] = L.sb_pt_lm[qc];
            }
            double w1[SCHUR_U][6], w2[SCHUR_U][6], Vv[SCHUR_U], dg[SCHUR_U];
#pragma unroll
            for (int u = 0; u < SCHUR_U; u++) {
                Vv[u] = B.V_pt[j[u]]; dg[u] = B.dgs_pt[j[u]];
#pragma unroll
                for (int k = 0; k < 6; k++) { w1[u][k] = B.w_pt[(size_t)(s1[u])*PT_REC + k]; w2[u][k] = B.w_pt[(size_t)(s2[u])*PT_REC + k]; }
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
        for (int q = L.sb_tx_off[b] + tid; q < L.sb_tx_off[b+1]; q += SCHUR_T) {
            const int s1 = L.sb_tx_s1[q], s2 = L.sb_tx_s2[q], j = L.sb_tx_lm[q];
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
        // operands of the tail, independent of the sums: issued before the reduction
        double tail = 0.0;
        if (wave == 0 && lane < 36) {
            const int r = lane/6, cc = lane % 6;
            const double *out = B.pairOut;
            if (a == c) {
                const double *rt = out + (size_t)sym6(r, cc)*L.n_pair, *rh = out + (size_t)(63 + sym6(r, cc))*L.n_pair;
                tail = range_sum<24>(rt, L.pose_t_off[a], L.pose_t_off[a+1]) + range_sum<24>(rh, L.pose_h_off[a], L.pose_h_off[a+1]);
                if (r == cc && !multi) tail += B.dgs_p[6*a + r]*irad;      // multi-GPU: added once after the all-reduce
            } else {
                int pab = L.sb_pab[b], pba = L.sb_pba[b];
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
            }
            __syncthreads();
        }
        if (wave == 0) {
#pragma unroll
            for (int k = 0; k < 36; k++) lds[k*65 + lane] = acc[k];          // transpose: lane l < 36 sums entry l over the 64 lanes
        }
        __syncthreads();
        if (wave > 0) return;
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
    } else {
        const int a = b - L.n_sb;
        if (a >= W.n_kf) return;                               // (the gradient-only grid is rounded up to a multiple of 8)
        const int ia = W.fidx[a];
        if (ia < 0) return;
        double acc[6] = {0,0,0,0,0,0};
        const int ps0 = L.pose_ps_off[a], ps1 = L.pose_ps_off[a+1];
        for (int base = ps0; base < ps1; base += SCHUR_T*SCHUR_U) {
            int s[SCHUR_U], j[SCHUR_U]; bool ok[SCHUR_U];
#pragma unroll
            for (int u = 0; u < SCHUR_U; u++) {
                const int q = base + u*SCHUR_T + tid; ok[u] = q < ps1;
                const int qc = min(q, ps1 - 1);
                s[u] = L.pose_ps[qc]; j[u] = L.pose_ps_lm[qc];
            }
            double w[SCHUR_U][6], bb[SCHUR_U], Vv[SCHUR_U], dg[SCHUR_U];
#pragma unroll
            for (int u = 0; u < SCHUR_U; u++) {
                bb[u] = B.b_pt[j[u]]; Vv[u] = B.V_pt[j[u]]; dg[u] = B.dgs_pt[j[u]];
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
        for (int q = L.pose_ts_off[a] + tid; q < L.pose_ts_off[a+1]; q += SCHUR_T) {
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
        const double bpv = tid < 6 ? (multi ? B.bp_loc[6*a + tid] : B.bp[6*a + tid]) : 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) acc[k] = wave_sum1(acc[k]);
        if (lane == 0) {
#pragma unroll
            for (int k = 0; k < 6; k++) lds[wave*6 + k] = acc[k];
        }
        __syncthreads();
        if (tid < 6) W.g[6*ia + tid] = bpv - (SCHUR_NW > 1 ? (((lds[tid] + lds[6 + tid]) + lds[12 + tid]) + lds[18 + tid]) : lds[tid]);
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
template <bool TEXT>
__global__ __launch_bounds__(64) void k_schur_quad(Work W, LevelDev L, int multi) {
    LmState *st = W.st;
    if (st->done) return;
    __shared__ double lds[4*12*17];                             // a third of a group's 36 sums at a time: 6.5 KB, the registers set the occupancy
    const int lane = threadIdx.x, grp = lane >> 4, sub = lane & 15;
    // workgroups are handed to the 8 XCDs round-robin (workgroup i -> XCD i mod 8), and an XCD's L2 does not see the others': neighbouring S
    // blocks read the same landmarks' records, so the workgroups of ONE XCD take a contiguous range of blocks (the kernel is bound by
    // L2 -> L1 line fills; with neighbouring blocks on eight different XCDs every record crossed the fabric up to eight times)
    const int per = (int)gridDim.x >> 3, wg = ((int)blockIdx.x & 7)*per + ((int)blockIdx.x >> 3);     // (the grid is a multiple of 8 workgroups)
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
            Vv[u] = B.V_pt[j[u]]; dg[u] = B.dgs_pt[j[u]];
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

#include "tsba_solve.h"
#include "tsba_chol.h"
#include "tsba_band.h"
#include "tsba_bandp.h"
#include "tsba_bandcr.h"
#include "tsba_bandcre.h"
#include "tsba_bandms.h"
#include "tsba_pcg.h"
#include "tsba_pose.h"

// ---- landmark back-substitution + candidate parameters.  256-thread blocks: points | texts | poses
__global__ __launch_bounds__(256) void k_back(Work W, LevelDev L, int nb_pt, int nb_tx) {
    LmState *st = W.st;
    const int b = blockIdx.x, tid = threadIdx.x;
    // static offsets / slot poses of this thread's landmark first: in flight together with the LM state
    int o = 0, e = 0, act_ = 0, a0[6] = {0, 0, 0, 0, 0, 0};
    if (b < nb_pt) { const int j = b*256 + tid; if (j < W.n_pt) { o = L.pls_off[j]; e = L.pls_off[j+1]; act_ = W.act_pt[j];
#pragma unroll
        for (int u = 0; u < 6; u++) a0[u] = L.pt_pose6[6*(size_t)j + u]; } }
    else if (b < nb_pt + nb_tx) { const int j = (b - nb_pt)*256 + tid; if (j < W.n_text) { o = L.tls_off[j]; e = L.tls_off[j+1]; act_ = W.act_tx[j]; } }
    if (st->done) return;
    __shared__ double red[256];
    const int cur = st->cur;
    const double irad = 1.0/st->radius;
    const bool fail = st->step_fail;
    const LinBuf &B = W.lb[st->lcur];
    double step2 = 0.0, mcc = 0.0;
    if (b < nb_pt) {
        int j = b*256 + tid;
        if (j < W.n_pt) {
            double rh = W.rho[cur][j], d = 0.0;
            if (!fail && e > o && act_) {
                double acc = B.b_pt[j];
                for (int s0 = o; s0 < e; s0 += 6) {                                 // 6 slots in flight; dp is 0 for constant / absent poses
                    int a[6]; double w[6][6], dpv[6][6];
#pragma unroll
                    for (int u = 0; u < 6; u++) a[u] = s0 == o ? a0[u] : L.pslot_pose[min(s0 + u, e - 1)];
#pragma unroll
                    for (int u = 0; u < 6; u++)
#pragma unroll
                        for (int k = 0; k < 6; k++) { w[u][k] = B.w_pt[(size_t)(min(s0 + u, e - 1))*PT_REC + k]; dpv[u][k] = W.dp[6*a[u] + k]; }
#pragma unroll
                    for (int u = 0; u < 6; u++)
#pragma unroll
                        for (int k = 0; k < 6; k++) acc += s0 + u < e ? w[u][k]*dpv[u][k] : 0.0;
                }
                const double lam = B.dgs_pt[j]*irad;
                d = -acc/(B.V_pt[j] + lam);
                step2 = d*d; mcc = lam*d*d - B.b_pt[j]*d;
            }
            W.rho[cur ^ 1][j] = rh + d;
        }
    } else if (b < nb_pt + nb_tx) {
        int j = (b - nb_pt)*256 + tid;
        if (j < W.n_text) {
            double d[3] = {0,0,0};
            if (!fail && e > o && act_) {
                double acc[3] = { B.b_tx[j], B.b_tx[(size_t)W.n_text + j], B.b_tx[(size_t)2*W.n_text + j] };
                for (int s0 = o; s0 < e; s0 += 3) {                                 // 3 slots in flight
                    int a[3]; double w[3][18], dpv[3][6];
#pragma unroll
                    for (int u = 0; u < 3; u++) a[u] = L.tslot_pose[min(s0 + u, e - 1)];
#pragma unroll
                    for (int u = 0; u < 3; u++) {
#pragma unroll
                        for (int k = 0; k < 18; k++) w[u][k] = B.w_tx[(size_t)(min(s0 + u, e - 1))*TX_REC + k];
#pragma unroll
                        for (int k = 0; k < 6; k++) dpv[u][k] = W.dp[6*a[u] + k];
                    }
#pragma unroll
                    for (int u = 0; u < 3; u++) if (s0 + u < e) {
#pragma unroll
                        for (int k = 0; k < 6; k++) { acc[0] += w[u][k*3]*dpv[u][k]; acc[1] += w[u][k*3 + 1]*dpv[u][k]; acc[2] += w[u][k*3 + 2]*dpv[u][k]; }
                    }
                }
                double Vd[6], Vi[6], lam[3];
#pragma unroll
                for (int k = 0; k < 6; k++) Vd[k] = B.V_tx[(size_t)k*W.n_text + j];
#pragma unroll
                for (int k = 0; k < 3; k++) lam[k] = B.dgs_tx[(size_t)k*W.n_text + j]*irad;
                Vd[0] += lam[0]; Vd[3] += lam[1]; Vd[5] += lam[2];
                if (inv_sym3(Vd, Vi)) {
                    d[0] = -(Vi[0]*acc[0] + Vi[1]*acc[1] + Vi[2]*acc[2]);
                    d[1] = -(Vi[1]*acc[0] + Vi[3]*acc[1] + Vi[4]*acc[2]);
                    d[2] = -(Vi[2]*acc[0] + Vi[4]*acc[1] + Vi[5]*acc[2]);
                    for (int k = 0; k < 3; k++) { step2 += d[k]*d[k]; mcc += lam[k]*d[k]*d[k] - B.b_tx[(size_t)k*W.n_text + j]*d[k]; }
                }
            }
            for (int k = 0; k < 3; k++) W.theta[cur ^ 1][3*j + k] = W.theta[cur][3*j + k] + d[k];
        }
    } else {
        int a = (b - nb_pt - nb_tx)*256 + tid;
        if (a < W.n_kf) {
            const double *x = W.pose[cur] + 7*a; double *c = W.pose[cur ^ 1] + 7*a;
            if (!fail && W.fidx[a] >= 0) {
                double d[6];
#pragma unroll
                for (int k = 0; k < 6; k++) d[k] = W.dp[6*a + k];
                double q[4] = { x[0], x[1], x[2], x[3] }, qn[4];
                quat_plus(q, d, qn);
                for (int k = 0; k < 4; k++) { c[k] = qn[k]; step2 += (qn[k] - q[k])*(qn[k] - q[k]); }
                for (int k = 0; k < 3; k++) { c[4 + k] = x[4 + k] + d[3 + k]; step2 += d[3 + k]*d[3 + k]; }
                for (int k = 0; k < 6; k++) { const double lam = B.dgs_p[6*a + k]*irad; mcc += lam*d[k]*d[k] - B.bp[6*a + k]*d[k]; }
            } else for (int k = 0; k < 7; k++) c[k] = x[k];
        }
    }
    step2 = block_sum<256>(step2, red); mcc = block_sum<256>(mcc, red);
    if (tid == 0) { W.partial[2*b] = step2; W.partial[2*b + 1] = mcc; }
}

// ---- step quality and trust-region update (Ceres 1.x TrustRegionMinimizer / LevenbergMarquardtStrategy semantics)
__global__ __launch_bounds__(256) void k_decide(Work W, LevelDev L, int nb_back, int nb_lm, tsba_options o, int multi, int npp) {
    LmState *st = W.st;
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
    [&]() {
    mcc *= 0.5;                                   // model_cost_change = 1/2 dx^T (Lambda dx - g)
    st->it++;
    st->cand_cost = cost; st->model_change = mcc; st->step_norm = sqrt(step2);
    if (st->step_fail || !(mcc > 0.0)) {          // invalid step (LevenbergMarquardtStrategy::StepIsInvalid)
        st->step_fail = 0;
        if (++st->invalid >= 5) { st->done = 1; st->term = 5; return; }
        st->radius *= 0.5;
    } else {
        st->invalid = 0; st->n_cost++;
        if (!(cost == cost)) cost = 1.7976931348623157e308;
        if (st->step_norm <= o.parameter_tolerance*(st->x_norm + o.parameter_tolerance)) { st->done = 1; st->term = 2; return; }
        double cost_change = st->x_cost - cost;
        if (fabs(cost_change) <= o.function_tolerance*st->x_cost) { st->done = 1; st->term = 1; return; }
        double rel = cost_change/mcc;
        if (rel > o.min_relative_decrease) {      // accept: the speculative linearisation becomes the current one
            st->cur ^= 1; st->lcur ^= 1; st->accepted++; st->n_lin++;
            st->x_cost = cost; st->x_norm = sqrt(xn_c); st->gmax = gmax_c;
            double t = 2.0*rel - 1.0, f = 1.0 - t*t*t; if (f < 1.0/3.0) f = 1.0/3.0;
            st->radius = fmin(st->radius/f, o.max_radius);
            st->decrease_factor = 2.0;
            if (gmax_c <= o.gradient_tolerance) { st->done = 1; st->term = 3; return; }
        } else {
            st->radius = st->radius/st->decrease_factor; st->decrease_factor *= 2.0;
        }
    }
    if (st->it >= st->max_it) { st->done = 1; st->term = 0; }
    else if (st->radius < o.min_radius) { st->done = 1; st->term = 4; }
    }();
    if (W.hprog) { *W.hprog = ((unsigned long long)W.pass_seq << 32) | ((unsigned long long)st->it << 1) | (st->done ? 1u : 0u); __threadfence_system(); }
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
__global__ __launch_bounds__(64) void k_outlier(Work W, LevelDev L, double chi2_mono, double chi2_text, double bad_ratio,
                                                int do_scene, int do_text, const PoseState *pfin) {
    LmState *st = W.st;
    const int b = blockIdx.x, lane = threadIdx.x;
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

// ------------------------------------------------------------------------------------------------ host side
static int pose_grid(const LevelDev &D) { return std::max(1, (D.n_sc + 255)/256 + (D.n_pf + 31)/32); }    // workgroups of k_pose_iter
struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
};
struct PassRecord { LmState st; };

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool uploaded = false;
    tsba_options opt;
    int n_kf = 0, n_pt = 0, n_text = 0, n_tobs = 0, n_sgood = 0, n_tfgood = 0, n_levels = 0;
    Work W;
    std::vector<HostPlan> hplan;                  // per level (only levels used by the options are built)
    std::vector<LevelDev> lev;
    std::vector<int> lev_built;                   // level l is on the device (plan lists, reference features, image table)
    // one-shot calls on small windows stage a level when its pass begins (the coarse passes run while the plan of level 0 is still being built)
    std::vector<std::thread> planners;            // planners[l]: the host thread that builds level l's plan (joined by stage_level)
    std::vector<int> lev_planned;                 // a plan of level l was started for this upload
    const tsba_problem *stage_p = nullptr;        // the caller's problem while levels may still be staged from it (one-shot calls only)
    std::vector<int> ic_slot; bool use_img_cache = false;      // plane cache: slot of every keyframe of this upload
    std::atomic<int> plan_done[TSBA_MAX_LEVELS];  // set by a plan thread when its plan is complete: a level is staged ahead of its pass only when that costs no wait
    hipStream_t copy_stream = nullptr; hipEvent_t ev_stage[TSBA_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};
    bool stage_async = false; int lev_wait[TSBA_MAX_LEVELS] = {0, 0, 0, 0};     // levels staged during a solve go over the copy stream; their pass waits for the event
    // restart copies
    double *pose0 = nullptr, *rho0 = nullptr, *theta0 = nullptr; uint8_t *sgood0 = nullptr, *tobs_good0 = nullptr, *tfgood0 = nullptr;
    uint8_t *kf_initial = nullptr;
    LmState *st_host = nullptr;                   // pinned: per-pass snapshots
    LmState *st_log = nullptr;                    // device [MAX passes]
    int nb_back_max = 0;
    size_t lds_limit = 0;
    std::vector<uint8_t *> img_dev[TSBA_MAX_LEVELS];
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // RCCL (global BA sharded over the GPUs of one node): one process per GPU, communicator created by tsba_comm_init
    bool pose_only = false;                       // one keyframe, every landmark frozen in its host: the fused pose-only LM kernel applies
    double *S_alloc = nullptr; size_t S_count = 0;
    int S_up = CH_NB;                               // band storage: columns stored right of the diagonal + 1
    bool S_stale = true;                            // band storage: entries of another pass may be left in S (cleared before the next assembly)
    double *S_xchg = nullptr; int xchg_wp = 0;      // multi-GPU, band storage: packed band rows for the exchange (k_band_pack)
    bool sep_cr = false;                  // separator system by cyclic reduction on the compact block pool (tsba_bandcr.h)
    int band_parts = 1; double *Lb = nullptr, *Tbuf = nullptr, *Bpart = nullptr, *Ssep = nullptr, *Lcol_sep = nullptr, *CRcontrib = nullptr, *CRfac = nullptr; int nsep_ld = 0; Work Wsep;   // partitioned band solver (tsba_bandp.h)
    double *Lcol = nullptr; int band_stream = 0;  // streaming band solver (tsba_band.h): L by block column; 1 = every built level fits it     // storage behind W.S (dense or band)
    float *lbl_dev = nullptr, *lbl_host = nullptr; size_t lbl_cap = 0;   // text label image staging
    unsigned long long *hprog = nullptr; unsigned int pass_seq = 0;   // pinned progress word written by k_postlin / k_decide
    std::vector<struct Slab> slabs; int cur_slab = 0;
    int run_slab = 0; size_t run_off = 0, run_len = 0;   // pending contiguous host-to-device range
    int cur_bw_rows = 1 << 30;                     // band bound of the level being solved (set by launch_pass_init)
    // device cache of keyframe pyramid planes (tsba_problem.kf_id): one slot per keyframe, all levels of a keyframe contiguous
    struct ImgCache { uint8_t *dev = nullptr, *stage = nullptr; size_t stage_cap = 0, slot = 0, lvl_off[TSBA_MAX_LEVELS] = {0,0,0,0};
                      int w[TSBA_MAX_LEVELS] = {0,0,0,0}, h[TSBA_MAX_LEVELS] = {0,0,0,0}; unsigned lvl_mask = 0;
                      long long id[TSBA_IMG_CACHE_KF]; unsigned long long used[TSBA_IMG_CACHE_KF]; bool full[TSBA_IMG_CACHE_KF]; unsigned long long tick = 0;
                      long long hits = 0, misses = 0; } ic;
    EcgBuf ecg{}; double *ecg_alloc = nullptr; size_t ecg_bytes = 0;      // enlarged conjugate gradients (tsba_pcg.h)
    MsBuf ms{}; double *ms_alloc = nullptr; size_t ms_bytes = 0; int ms_cap = 0;      // multi-right-hand-side solve phase of the partitioned band solver (tsba_bandms.h)
    int cov_text = -1; double *cov_log = nullptr;     // tsba_theta_optim: V of this plane at the end of every pass [TSBA_MAX_LEVELS][6]
    int far_B = 0, n_far = 0, pcg_parts = 0; unsigned int pcg_seq = 0;      // band + long-range blocks (tsba_pcg.h): band of M in pose blocks, blocks outside it, partial sums per vector kernel
    int rank = 0, world = 1; bool force_multi = false;
    tsba_debug_options dbg{};                      // test / diagnostics switches (tsba_debug_set), all zero in production
    struct LocalGroup *lgroup = nullptr;           // in-process communicator (tsba_comm_init_local)
    bool in_solve = false, has_token = false;      // local group: this rank is inside tsba_solve / holds the group's device token
    size_t x_acc = 0, x_trial = 0, x_lin = 0, x_pass = 0;   // bytes handed to collectives: running total; last LM trial / linearisation / pass set-up
    decltype(&ncclCommCount) p_count = nullptr;
    void *rccl_so = nullptr; ncclComm_t comm = nullptr;
    decltype(&ncclGetUniqueId) p_getid = nullptr; decltype(&ncclCommInitRank) p_init = nullptr;
    decltype(&ncclAllReduce) p_allreduce = nullptr; decltype(&ncclCommDestroy) p_destroy = nullptr;
    decltype(&ncclGetErrorString) p_errstr = nullptr;
};

static void set_err(Ctx *c, const std::string &s) { c->err = s; }
static bool is_multi(const Ctx *c);

// Device memory comes from a few large slabs (bump allocation, 256-byte aligned) that persist across uploads: the ~200 arrays
// of a problem cost no hipMalloc / hipFree / hipMemset each (one memset per slab and upload), and host data is staged through a
// pinned mirror of the slab so that consecutive uploads leave as ONE host-to-device copy (a plan is ~50 arrays per level).
struct Slab { char *dev = nullptr, *host = nullptr; size_t size = 0, used = 0; };
static void flush_run(Ctx *c) {
    if (c->run_len) { Slab &sl = c->slabs[c->run_slab];
        hipMemcpyAsync(sl.dev + c->run_off, sl.host + c->run_off, c->run_len, hipMemcpyHostToDevice, c->stage_async && c->copy_stream ? c->copy_stream : c->stream); c->run_len = 0; }
}
static int slab_take(Ctx *c, size_t bytes, int *slab, size_t *off) {
    for (;;) {
        if (c->cur_slab < (int)c->slabs.size()) { Slab &sl = c->slabs[c->cur_slab];
            if (sl.size - sl.used >= bytes) { *slab = c->cur_slab; *off = sl.used; sl.used += bytes; return 0; }
            c->cur_slab++; continue; }
        Slab sl; sl.size = std::max<size_t>(bytes, (size_t)64 << 20);
        hipError_t e = hipMalloc((void **)&sl.dev, sl.size);
        if (e != hipSuccess) { set_err(c, std::string("hipMalloc: ") + hipGetErrorString(e)); return TSBA_ERR_DEVICE; }
        hipMemsetAsync(sl.dev, 0, sl.size, c->stream);
        if (c->stage_async) hipStreamSynchronize(c->stream);     // (a slab born during a solve: the copy stream must not overtake its clearing)
        c->slabs.push_back(sl);
    }
}
template <typename T>
static int dev_alloc(Ctx *c, T **out, size_t n) {
    const size_t bytes = (std::max<size_t>(n, 1)*sizeof(T) + 255) & ~(size_t)255;
    int si; size_t off; int rc = slab_take(c, bytes, &si, &off); if (rc) return rc;
    *out = (T *)(c->slabs[si].dev + off); return 0;
}
template <typename T>
static int dev_upload(Ctx *c, const T **out, const T *src, size_t n) {
    const size_t bytes = (std::max<size_t>(n, 1)*sizeof(T) + 255) & ~(size_t)255;
    int si; size_t off; int rc = slab_take(c, bytes, &si, &off); if (rc) return rc;
    Slab &sl = c->slabs[si];
    *out = (const T *)(sl.dev + off);
    if (!n || !src) return 0;
    if (!sl.host) { hipError_t e = hipHostMalloc((void **)&sl.host, sl.size, hipHostMallocDefault);
        if (e != hipSuccess) { set_err(c, std::string("hipHostMalloc: ") + hipGetErrorString(e)); return TSBA_ERR_DEVICE; } }
    if (c->run_len && (c->run_slab != si || c->run_off + c->run_len != off)) flush_run(c);
    if (!c->run_len) { c->run_slab = si; c->run_off = off; }
    memcpy(sl.host + off, src, n*sizeof(T));
    if (bytes > n*sizeof(T)) memset(sl.host + off + n*sizeof(T), 0, bytes - n*sizeof(T));
    c->run_len = off + bytes - c->run_off;
    return 0;
}
template <typename T>
static int dev_upload_vec(Ctx *c, const T **out, const std::vector<T> &v) { return dev_upload(c, out, v.data(), v.size()); }

static void join_planners(Ctx *c) { for (auto &t : c->planners) if (t.joinable()) t.join(); c->planners.clear(); }
static void free_problem(Ctx *c) {
    join_planners(c); c->stage_p = nullptr;
    hipStreamSynchronize(c->stream);
    // the slabs stay (tsba_destroy frees them): zero what the last problem used, restart the bump allocation
    for (Slab &sl : c->slabs) { if (sl.used) hipMemsetAsync(sl.dev, 0, sl.used, c->stream); sl.used = 0; }
    c->cur_slab = 0; c->run_len = 0;
    c->uploaded = false; c->lev.clear(); c->lev_built.clear();      // (the host plans keep their storage for the next upload: HostPlan::recycle)
    for (int l = 0; l < TSBA_MAX_LEVELS; l++) c->img_dev[l].clear();
}

extern "C" {

void tsba_default_options_local(tsba_options *o) {
    memset(o, 0, sizeof(*o));
    o->w_sx = o->w_sy = 1.0/1.2; o->w_t = 1.0/0.2;                  // optimizer.cc:1350-1351
    o->huber_scene = sqrt(5.991); o->huber_text = 3.0;              // :1369, :1454
    o->n_passes = 3;
    for (int i = 0; i < 3; i++) { o->levels[i] = 2 - i; o->its[i] = 10; o->chi2_mono[i] = 12.25; o->chi2_text[i] = i == 2 ? 0.95 : 0.5; }  // :282-289
    o->text_bad_ratio = 0.99; o->state = TSBA_STATE_LOCAL; o->outlier_scene = o->outlier_text = 1;
    o->use_text = 1; o->filter_good = 1; o->text_jacobian = 0;
    o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32; o->min_relative_decrease = 1e-3;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->min_diagonal = 1e-6; o->max_diagonal = 1e32; o->lm_shard = 0; o->lm_nshard = 1;
}
void tsba_default_options_pose(tsba_options *o) { tsba_default_options_local(o); o->state = TSBA_STATE_NOTREACHWIN; }
void tsba_default_options_global(tsba_options *o) {
    tsba_default_options_local(o);
    o->w_sx = o->w_sy = o->w_t = 1.0; o->n_passes = 1; o->levels[0] = 0; o->its[0] = 20; o->chi2_mono[0] = 18.0;   // :411-414
    o->state = TSBA_STATE_GLOBAL; o->outlier_scene = o->outlier_text = 0; o->use_text = 0; o->filter_good = 0;
}

void tsba_default_options_init(tsba_options *o) {                    // optimizer.cc:960-1056
    tsba_default_options_local(o);
    o->w_sx = o->w_sy = o->w_t = 1.0; o->huber_scene = 3.0; o->huber_text = 3.0;
    o->n_passes = 4; for (int i = 0; i < 4; i++) { o->levels[i] = 3 - i; o->its[i] = 10; }
    o->state = TSBA_STATE_NOTREACHWIN; o->outlier_scene = o->outlier_text = 0; o->filter_good = 0;
}
void tsba_default_options_landmarker(tsba_options *o) {              // optimizer.cc:531-541,1861,1873,1922
    tsba_default_options_local(o);
    o->w_sx = o->w_sy = o->w_t = 1.0; o->huber_scene = sqrt(5.991); o->huber_text = 2.0;
    o->n_passes = 4; for (int i = 0; i < 4; i++) { o->levels[i] = 3 - i; o->its[i] = 50; o->chi2_mono[i] = 18.0; o->chi2_text[i] = 1.5; }
    o->state = TSBA_STATE_NOTREACHWIN; o->outlier_scene = 1; o->outlier_text = 0;
}
void tsba_default_options_theta(tsba_options *o) {                   // optimizer.cc:610-615,2176,2203-2209
    tsba_default_options_local(o);
    o->w_sx = o->w_sy = o->w_t = 1.0; o->huber_text = 1e300;          // LossFunction* = nullptr
    for (int i = 0; i < 3; i++) o->its[i] = 50;
    o->state = TSBA_STATE_NOTREACHWIN; o->outlier_scene = o->outlier_text = 0; o->filter_good = 0;
}

int tsba_create(void **ctx, int device) {
    if (!ctx) return TSBA_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return TSBA_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return TSBA_ERR_DEVICE;
    Ctx *c = new Ctx(); c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return TSBA_ERR_DEVICE; }
    for (int l = 0; l < TSBA_MAX_LEVELS; l++) c->plan_done[l].store(0);
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) c->copy_stream = nullptr;     // (optional: staging then shares the compute stream)
    for (int l = 0; l < TSBA_MAX_LEVELS; l++) if (hipEventCreateWithFlags(&c->ev_stage[l], hipEventDisableTiming) != hipSuccess) c->ev_stage[l] = nullptr;
    hipDeviceProp_t prop;
    const bool ok = hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess
        && hipHostMalloc((void **)&c->st_host, sizeof(LmState)*TSBA_MAX_LEVELS, 0) == hipSuccess
        && hipMalloc((void **)&c->st_log, sizeof(LmState)*TSBA_MAX_LEVELS) == hipSuccess
        && hipGetDeviceProperties(&prop, device) == hipSuccess;
    if (!ok) {                                      // nothing half-built leaves this function
        if (c->ev0) hipEventDestroy(c->ev0); if (c->ev1) hipEventDestroy(c->ev1);
        if (c->st_host) hipHostFree(c->st_host); if (c->st_log) hipFree(c->st_log);
        hipStreamDestroy(c->stream); delete c; return TSBA_ERR_DEVICE;
    }
    if (hipHostMalloc((void **)&c->hprog, 64, hipHostMallocDefault) != hipSuccess) c->hprog = nullptr; else *c->hprog = 0;   // (optional: early-exit polling only)
    c->lds_limit = prop.sharedMemPerBlock;       // 64 KiB default static limit; dynamic up to 160 KiB on gfx950
    if (c->lds_limit < 160*1024) c->lds_limit = 160*1024;
    *ctx = c; return TSBA_OK;
}
static void lgroup_forget(Ctx *c);              // (defined with LocalGroup below)
int tsba_destroy(void *ctx) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    hipSetDevice(c->device);
    lgroup_forget(c);
    free_problem(c);
    hipStreamSynchronize(c->stream);
    for (Slab &sl : c->slabs) { hipFree(sl.dev); if (sl.host) hipHostFree(sl.host); }
    c->slabs.clear();
    if (c->comm && c->p_destroy) c->p_destroy(c->comm);
    hipHostFree(c->st_host); hipFree(c->st_log); if (c->hprog) hipHostFree(c->hprog);
    if (c->lbl_dev) hipFree(c->lbl_dev); if (c->lbl_host) hipHostFree(c->lbl_host);
    if (c->ic.dev) hipFree(c->ic.dev); if (c->ic.stage) hipHostFree(c->ic.stage);
    if (c->ms_alloc) hipFree(c->ms_alloc);
    if (c->ecg_alloc) hipFree(c->ecg_alloc);
    for (int l = 0; l < TSBA_MAX_LEVELS; l++) if (c->ev_stage[l]) hipEventDestroy(c->ev_stage[l]);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    hipEventDestroy(c->ev0); hipEventDestroy(c->ev1); hipStreamDestroy(c->stream);
    delete c; return TSBA_OK;
}
const char *tsba_last_error(void *ctx) { return ctx ? ((Ctx *)ctx)->err.c_str() : "null ctx"; }

// Every index the plan builder and the kernels dereference is range-checked here, once, in O(problem size): a bad index from the
// adapter becomes TSBA_ERR_ARG instead of a host out-of-bounds read or a GPU memory fault (the library never aborts the process).
static int check_problem(Ctx *c, const tsba_problem *p, const tsba_options *o) {
    auto bad = [&](const char *what) { set_err(c, std::string("invalid problem: ") + what); return TSBA_ERR_ARG; };
    if (!p || !o) return bad("null problem / options");
    if (p->n_kf <= 0 || p->n_levels < 1 || p->n_levels > TSBA_MAX_LEVELS) return bad("n_kf / n_levels");
    if (p->n_pt < 0 || p->n_text < 0 || p->n_tobs < 0 || p->n_sgood < 0) return bad("negative count");
    if (o->n_passes < 1 || o->n_passes > TSBA_MAX_LEVELS) return bad("n_passes");
    for (int i = 0; i < o->n_passes; i++) if (o->levels[i] < 0 || o->levels[i] >= p->n_levels) return bad("pass level out of range");
    if (o->text_jacobian != 0) { set_err(c, "text_jacobian=1 (numeric diff) is oracle-only"); return TSBA_ERR_ARG; }
    if (o->lm_nshard < 1 || o->lm_shard < 0 || o->lm_shard >= o->lm_nshard) return bad("lm_shard / lm_nshard");
    if (!p->pose) return bad("pose is null");
    if (p->n_pt > 0 && (!p->rho || !p->pt_ray || !p->pt_host)) return bad("null point array");
    if (p->n_text > 0 && (!p->theta || !p->text_host || !p->text_box_ray)) return bad("null text-plane array");
    if (p->n_sgood > 0 && !p->sgood) return bad("sgood is null");
    bool frozen = false;
    for (int j = 0; j < p->n_pt; j++) { const int h = p->pt_host[j]; if (h < -1 || h >= p->n_kf) return bad("pt_host out of range"); frozen |= h < 0; }
    if (frozen && !p->pt_host_Trw) return bad("pt_host_Trw is null but a point has a frozen host");
    frozen = false;
    for (int j = 0; j < p->n_text; j++) { const int h = p->text_host[j]; if (h < -1 || h >= p->n_kf) return bad("text_host out of range"); frozen |= h < 0; }
    if (frozen && !p->text_host_Twr) return bad("text_host_Twr is null but a plane has a frozen host");
    if (p->n_tobs > 0) {
        if (!p->tobs_kf || !p->tobs_text || !p->tobs_good || !p->tobs_fgood_off) return bad("null text-observation array");
        if (p->tobs_fgood_off[0] < 0) return bad("tobs_fgood_off[0] < 0");
        for (int t = 0; t < p->n_tobs; t++) {
            if (p->tobs_kf[t] < 0 || p->tobs_kf[t] >= p->n_kf) return bad("tobs_kf out of range");
            if (p->tobs_text[t] < 0 || p->tobs_text[t] >= p->n_text) return bad("tobs_text out of range");
            if (p->tobs_fgood_off[t+1] < p->tobs_fgood_off[t]) return bad("tobs_fgood_off not monotone");
        }
        if (p->tobs_fgood_off[p->n_tobs] > 0 && !p->tfgood) return bad("tfgood is null");
    }
    std::vector<char> seen(p->n_levels, 0);
    std::vector<int32_t> maxraw;
    for (int i = 0; i < o->n_passes; i++) {
        const int l = o->levels[i]; if (seen[l]) continue; seen[l] = 1;
        const int ns = p->n_sobs[l];
        if (ns < 0) return bad("n_sobs < 0");
        if (ns > 0) {
            if (!p->sobs_kf[l] || !p->sobs_pt[l] || !p->sobs_flag[l] || !p->sobs_uv0[l]) return bad("null scene-observation array");
            const int32_t *kf = p->sobs_kf[l], *pt = p->sobs_pt[l], *fl = p->sobs_flag[l];
            for (int s = 0; s < ns; s++) {
                if ((unsigned)kf[s] >= (unsigned)p->n_kf) return bad("sobs_kf out of range");
                if ((unsigned)pt[s] >= (unsigned)p->n_pt) return bad("sobs_pt out of range");
                if ((unsigned)fl[s] >= (unsigned)p->n_sgood) return bad("sobs_flag out of range");
            }
        }
        if (p->n_text > 0 && p->tfeat_off[l]) {             // (a level without text features passes tfeat_off = NULL)
            const int32_t *off = p->tfeat_off[l]; const int nf = p->n_tfeat[l];
            if (nf < 0 || off[0] < 0 || off[p->n_text] > nf) return bad("tfeat_off out of range");
            for (int j = 0; j < p->n_text; j++) if (off[j+1] < off[j]) return bad("tfeat_off not monotone");
            if (off[p->n_text] > off[0] && (!p->tfeat_raw[l] || !p->tfeat_uv[l] || !p->tfeat_ref[l])) return bad("null text-feature array");
            maxraw.assign(p->n_text, -1);                  // a feature's flag is tfgood[tobs_fgood_off[t] + raw]: raw must fit every observation's span
            for (int j = 0; j < p->n_text; j++) for (int f = off[j]; f < off[j+1]; f++) {
                const int r = p->tfeat_raw[l][f]; if (r < 0) return bad("tfeat_raw < 0"); maxraw[j] = std::max(maxraw[j], (int32_t)r); }
            for (int t = 0; t < p->n_tobs; t++)
                if (maxraw[p->tobs_text[t]] >= p->tobs_fgood_off[t+1] - p->tobs_fgood_off[t]) return bad("tfeat_raw exceeds the observation's flag span");
        }
        if (o->use_text && p->n_tobs > 0)
            if (!p->img[l] || p->img_w[l] <= 0 || p->img_h[l] <= 0 || p->img_w[l]*p->img_h[l] > MS_MASK_WORDS*32) { set_err(c, "missing image level or image larger than 640x480"); return TSBA_ERR_ARG; }
    }
    return 0;
}

static int stage_level(Ctx *c, const tsba_problem *p, int l, double *t_plan, double *t_img);
// lazy: the caller (a one-shot entry point) runs the solve right away and keeps *p alive until it returns -- on small windows only the first
// pass's level is staged here, the others when their pass begins (tsba_solve), so that the coarse passes run on the device while the host
// still builds and stages the plan of level 0 (the largest: ~1 ms of a 20-keyframe window's cold call)
static int upload_impl(void *ctx, const tsba_problem *p, const tsba_options *o, bool lazy) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    hipSetDevice(c->device);
    int rc = check_problem(c, p, o); if (rc) return rc;
    const bool tdbg = c->dbg.verbose != 0;
    auto tu0 = std::chrono::steady_clock::now(); double t_plan = 0.0, t_img = 0.0;
    free_problem(c);
    auto tu1 = std::chrono::steady_clock::now();
    c->opt = *o;
    if (c->world > 1) { c->opt.lm_shard = c->rank; c->opt.lm_nshard = c->world; }
    o = &c->opt;
    c->n_kf = p->n_kf; c->n_pt = p->n_pt; c->n_text = p->n_text; c->n_tobs = p->n_tobs; c->n_sgood = p->n_sgood; c->n_levels = p->n_levels;
    c->n_tfgood = p->n_tobs > 0 ? p->tobs_fgood_off[p->n_tobs] : 0;
    Work &W = c->W; memset(&W, 0, sizeof(W));
    W.n_kf = p->n_kf; W.n_pt = p->n_pt; W.n_text = p->n_text; W.n_tobs = p->n_tobs; W.N = 6*p->n_kf;
    for (int k = 0; k < 4; k++) W.K0[k] = p->K[k];
    W.w_sx = o->w_sx; W.w_sy = o->w_sy; W.w_t = o->w_t; W.huber_s = o->huber_scene; W.huber_t = o->huber_text;
    W.filter_good = o->filter_good; W.min_diag = o->min_diagonal; W.max_diag = o->max_diagonal;
    W.rank = c->rank; W.world = c->world;
    // the per-level plans are independent of each other and of the uploads below: one host thread per level builds them while this
    // thread stages the parameter / observation arrays (C4: 1.6 ms of plan construction in sequence -> the largest level, overlapped)
    c->hplan.resize(p->n_levels); c->lev.resize(p->n_levels); c->lev_built.assign(p->n_levels, 0); c->lev_planned.assign(p->n_levels, 0);
    c->planners.clear(); c->planners.resize(p->n_levels);
    std::vector<std::thread> &planners = c->planners;
    struct Joiner { Ctx *c; bool armed; ~Joiner() { if (armed) join_planners(c); } } joiner{c, true};   // on the error returns
    int n_lev_used = 0; { std::vector<char> sn(p->n_levels, 0); for (int q = 0; q < o->n_passes; q++) if (!sn[o->levels[q]]) { sn[o->levels[q]] = 1; n_lev_used++; } }
    const bool small_window = solve_lds_doubles(W.N)*sizeof(double) <= 160*1024 - 64;
    const bool defer = lazy && small_window && p->n_kf > 1 && n_lev_used > 1 && !is_multi(c);
    for (int l = 0; l < TSBA_MAX_LEVELS; l++) { c->plan_done[l].store(0); c->lev_wait[l] = 0; }
    {   auto tp0 = std::chrono::steady_clock::now();
        std::vector<char> seen(p->n_levels, 0);
        const int n_lev = n_lev_used;
        const bool reorder = !c->dbg.no_kf_reorder;
        // ring maps (one loop closure): a single-level, single-GPU solve through the partitioned solver with the cyclic-reduction separator tree
        const int ring_max = (n_lev == 1 && !c->dbg.no_ring && !c->dbg.no_band_stream && c->dbg.sep_solver != 1 && c->dbg.sep_solver != 3 && c->dbg.band_parts != 1) ? CR_SMAX/6 : 0;
        // maps with long-range coupling (several loop closures, points seen again much later): band + blocks outside it, preconditioned conjugate gradients
        const int far_max = (n_lev == 1 && c->dbg.far_solver != 1 && !c->dbg.no_band_stream && (!c->dbg.no_ring || c->dbg.far_solver == 2)) ? CR_SMAX/6 : 0;     // (no_ring asks for the reordering path)
        const bool far_force = c->dbg.far_solver == 2;
        for (int ps = o->n_passes - 1; ps >= 0; ps--) { const int l = o->levels[ps]; if (seen[l]) continue;
            bool later = false; for (int q = 0; q < ps; q++) later |= o->levels[q] == l;       // (a level used by an earlier pass is started with that pass)
            if (later) continue;
            seen[l] = 1;
            HostPlan *H = &c->hplan[l];
            c->lev_planned[l] = 1;
            std::atomic<int> *done = &c->plan_done[l];
            // the levels of the later passes first (the largest plans); a deferring call builds the first pass's (small) plan on this thread:
            // it is needed at once, and a thread's start costs as much as that plan
            if (defer && ps == 0) { build_plan(p, o, l, *H, tdbg, reorder, ring_max, far_max, far_force); done->store(1); continue; }
            planners[l] = std::thread([p, o, l, H, tdbg, reorder, ring_max, far_max, far_force, done]() { build_plan(p, o, l, *H, tdbg, reorder, ring_max, far_max, far_force); done->store(1, std::memory_order_release); }); }
        t_plan += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count();
    }
#define UP(dst, src, n) do { rc = dev_upload(c, &(dst), (src), (size_t)(n)); if (rc) return rc; } while (0)
#define AL(dst, n) do { rc = dev_alloc(c, &(dst), (size_t)(n)); if (rc) return rc; } while (0)
    const double *cd; const uint8_t *cu;
    UP(cd, p->pose, 7*(size_t)p->n_kf); c->pose0 = (double *)cd;
    UP(cd, p->rho, p->n_pt); c->rho0 = (double *)cd;
    UP(cd, p->theta, 3*(size_t)p->n_text); c->theta0 = (double *)cd;
    UP(cu, p->sgood, p->n_sgood); c->sgood0 = (uint8_t *)cu;
    UP(cu, p->tobs_good, p->n_tobs); c->tobs_good0 = (uint8_t *)cu;
    UP(cu, p->tfgood, c->n_tfgood); c->tfgood0 = (uint8_t *)cu;
    std::vector<uint8_t> ki(p->n_kf, 0); if (p->kf_initial) memcpy(ki.data(), p->kf_initial, p->n_kf);
    UP(cu, ki.data(), p->n_kf); c->kf_initial = (uint8_t *)cu;
    for (int b = 0; b < 2; b++) { AL(W.pose[b], 7*(size_t)p->n_kf); AL(W.rho[b], p->n_pt); AL(W.theta[b], 3*(size_t)p->n_text); }
    AL(W.sgood, p->n_sgood); AL(W.tobs_good, p->n_tobs); AL(W.tfgood, c->n_tfgood);
    UP(W.pt_ray, p->pt_ray, 2*(size_t)p->n_pt); UP(W.pt_host, p->pt_host, p->n_pt);
    { std::vector<double> z; const double *src = p->pt_host_Trw; if (!src) { z.assign(12*(size_t)p->n_pt, 0.0); src = z.data(); } UP(W.pt_Trw, src, 12*(size_t)p->n_pt); }
    UP(W.text_host, p->text_host, p->n_text);
    { std::vector<double> z; const double *src = p->text_host_Twr; if (!src) { z.assign(12*(size_t)p->n_text, 0.0); src = z.data(); } UP(W.text_Twr, src, 12*(size_t)p->n_text); }
    UP(W.text_box, p->text_box_ray, 8*(size_t)p->n_text);
    UP(W.tobs_kf, p->tobs_kf, p->n_tobs); UP(W.tobs_text, p->tobs_text, p->n_tobs); UP(W.tobs_fgood_off, p->tobs_fgood_off, (size_t)p->n_tobs + 1);
    AL(W.musig, 2*(size_t)p->n_tobs);
    AL(W.kf_in, p->n_kf); AL(W.kf_const, p->n_kf); AL(W.act_pt, p->n_pt); AL(W.act_tx, p->n_text);
    AL(W.fidx, p->n_kf); AL(W.nfree, 2); AL(W.dbg, 64); AL(W.LDbuf, 32*((size_t)p->n_kf + BAND_BW_MAX/6 + 1));     // (+ the ghost blocks of a ring map)
    // ---- plane cache (tsba_problem.kf_id): the keyframes of this call get their slots; the planes of those not seen before are staged and copied
    std::vector<int> &ic_slot = c->ic_slot; ic_slot.clear();
    bool &use_img_cache = c->use_img_cache; use_img_cache = false;
    if (p->kf_id && !o->img_on_device && o->use_text && p->n_tobs > 0 && p->n_kf <= TSBA_IMG_CACHE_KF) {
        Ctx::ImgCache &IC = c->ic;
        unsigned mask = 0; size_t off = 0, lo[TSBA_MAX_LEVELS] = {0,0,0,0}; bool same = IC.dev != nullptr;
        { std::vector<char> seen(p->n_levels, 0);
          for (int ps = 0; ps < o->n_passes; ps++) { const int l = o->levels[ps]; if (seen[l] || !p->img[l]) continue; seen[l] = 1; mask |= 1u << l; }
          for (int l = 0; l < p->n_levels; l++) if (mask >> l & 1) { lo[l] = off; off += ((size_t)p->img_w[l]*p->img_h[l] + 255) & ~(size_t)255;
              same = same && IC.w[l] == p->img_w[l] && IC.h[l] == p->img_h[l]; } }
        same = same && IC.lvl_mask == mask && IC.slot == off;
        if (mask && off) {
            if (!same) {                                     // new geometry (or first use): an empty cache of TSBA_IMG_CACHE_KF slots
                hipStreamSynchronize(c->stream);
                if (IC.dev) { hipFree(IC.dev); IC.dev = nullptr; }
                if (hipMalloc((void **)&IC.dev, off*TSBA_IMG_CACHE_KF) != hipSuccess) { IC.dev = nullptr; set_err(c, "hipMalloc (plane cache)"); return TSBA_ERR_DEVICE; }
                IC.slot = off; IC.lvl_mask = mask;
                for (int l = 0; l < TSBA_MAX_LEVELS; l++) { IC.lvl_off[l] = lo[l]; IC.w[l] = l < p->n_levels ? p->img_w[l] : 0; IC.h[l] = l < p->n_levels ? p->img_h[l] : 0; }
                for (int q = 0; q < TSBA_IMG_CACHE_KF; q++) { IC.id[q] = 0; IC.used[q] = 0; IC.full[q] = false; }
            }
            IC.tick++;
            ic_slot.assign((size_t)p->n_kf, -1);
            std::vector<int> miss;
            for (int k = 0; k < p->n_kf; k++) { for (int q = 0; q < TSBA_IMG_CACHE_KF; q++) if (IC.full[q] && IC.id[q] == p->kf_id[k]) { ic_slot[(size_t)k] = q; IC.used[q] = IC.tick; break; }
                if (ic_slot[(size_t)k] < 0) miss.push_back(k); }
            for (int k : miss) { int best = -1;              // least recently used slot that this call does not use
                for (int q = 0; q < TSBA_IMG_CACHE_KF; q++) if (IC.used[q] != IC.tick && (best < 0 || !IC.full[q] || (IC.full[best] && IC.used[q] < IC.used[best]))) { best = q; if (!IC.full[q]) break; }
                IC.id[best] = p->kf_id[k]; IC.full[best] = true; IC.used[best] = IC.tick; ic_slot[(size_t)k] = best; }
            IC.hits += p->n_kf - (long long)miss.size(); IC.misses += (long long)miss.size();
            if (!miss.empty()) {
                const size_t need = miss.size()*off;
                if (IC.stage_cap < need) { if (IC.stage) hipHostFree(IC.stage); IC.stage = nullptr; IC.stage_cap = 0;
                    if (hipHostMalloc((void **)&IC.stage, need, hipHostMallocDefault) != hipSuccess) { set_err(c, "hipHostMalloc (plane cache staging)"); return TSBA_ERR_DEVICE; }
                    IC.stage_cap = need; }
                for (size_t m = 0; m < miss.size(); m++) { const int k = miss[m];
                    for (int l = 0; l < p->n_levels; l++) if (mask >> l & 1) { if (!p->img[l][k]) { set_err(c, "null image pointer"); return TSBA_ERR_ARG; }
                        memcpy(IC.stage + m*off + lo[l], p->img[l][k], (size_t)p->img_w[l]*p->img_h[l]); }
                    hipMemcpyAsync(IC.dev + (size_t)ic_slot[(size_t)k]*off, IC.stage + m*off, off, hipMemcpyHostToDevice, c->stream); }
            }
            use_img_cache = true;
        }
    }
    // ---- per-level plans
    size_t mx_pair = 1, mx_tg = 1, mx_pslot = 1, mx_tslot = 1, mx_cnt = 1;
    for (int ps = 0; ps < o->n_passes; ps++) {
        const int l = o->levels[ps]; if (c->lev_built[l]) continue;
        if (defer && ps > 0) {                     // staged when its pass begins: buffers by what the level can hold at most
            const size_t nsc = (size_t)p->n_sobs[l], ntg = (size_t)p->n_tobs;
            mx_pair = std::max(mx_pair, std::min((size_t)p->n_kf*((size_t)p->n_kf + 1), nsc + ntg)); mx_tg = std::max(mx_tg, ntg);
            mx_pslot = std::max(mx_pslot, nsc + (size_t)p->n_pt); mx_tslot = std::max(mx_tslot, ntg + (size_t)p->n_text); mx_cnt = std::max(mx_cnt, nsc + ntg);
            continue; }
        rc = stage_level(c, p, l, &t_plan, &t_img); if (rc) return rc;
        const LevelDev &D = c->lev[l];
        mx_pair = std::max(mx_pair, (size_t)D.n_pair); mx_tg = std::max(mx_tg, (size_t)D.n_tg);
        mx_pslot = std::max(mx_pslot, (size_t)D.n_pslot); mx_tslot = std::max(mx_tslot, (size_t)D.n_tslot); mx_cnt = std::max(mx_cnt, (size_t)D.n_sc + D.n_tg);
    }
    c->stage_p = defer ? p : nullptr;
    c->nb_back_max = (p->n_pt + 255)/256 + (p->n_text + 255)/256 + (p->n_kf + 255)/256;
    {   bool po = p->n_kf == 1;
        for (int j = 0; po && j < p->n_pt; j++) po = p->pt_host[j] < 0;
        for (int j = 0; po && j < p->n_text; j++) po = p->text_host[j] < 0;
        c->pose_only = po && !c->dbg.no_pose_kernel;
        W.pst = nullptr; W.ppart = nullptr;
        if (c->pose_only) {
            size_t gmax = 1;
            for (int l = 0; l < p->n_levels; l++) if (c->lev_built[l]) gmax = std::max(gmax, (size_t)pose_grid(c->lev[l]));
            AL(W.pst, 2); AL(W.ppart, 2*28*gmax);
        } }
    for (int b = 0; b < 2; b++) {
        LinBuf &B = W.lb[b];
        AL(B.pairM, 27*mx_pair); AL(B.pairCost, mx_pair); AL(B.pairR, 9*mx_pair); AL(B.pairOut, 90*mx_pair);
        AL(B.tgM, 27*mx_tg); AL(B.tgCost, mx_tg);
        AL(B.w_pt, PT_REC*mx_pslot); AL(B.V_pt, p->n_pt); AL(B.b_pt, p->n_pt); AL(B.dgs_pt, p->n_pt);
        AL(B.w_tx, TX_REC*mx_tslot); AL(B.V_tx, 6*(size_t)p->n_text); AL(B.b_tx, 3*(size_t)p->n_text); AL(B.dgs_tx, 3*(size_t)p->n_text);
        AL(B.Hd, W.N); AL(B.bp, W.N); AL(B.bp_loc, W.N); AL(B.dgs_p, W.N);
        AL(B.lmpart, 3*((size_t)c->nb_back_max + mx_pair/256 + 2));
    }
    AL(W.sig_pt, p->n_pt); AL(W.sig_tx, 3*(size_t)p->n_text); AL(W.sig_p, W.N);
    AL(W.cb, 2*(size_t)W.N + 8); AL(W.cbm, 1);
    {   // reduced camera matrix: dense for the LDS solver; for the large-system Cholesky only its band (rows overlap in a skewed
        // view: S(i,j) = base[i*(LDB-1) + j], LDB = band + 96 columns of the diagonal block's upper triangle, where the inverse
        // diagonal factors are kept) -- 80 MB instead of 7.2 GB at 5000 keyframes, and what the ranks all-reduce
        const int use_lds_ = solve_lds_doubles(W.N)*sizeof(double) <= 160*1024 - 64;        // (as solve_lds_bytes)
        int bwmax = 0; for (int l = 0; l < p->n_levels; l++) if (c->lev_built[l]) bwmax = std::max(bwmax, c->lev[l].bw_rows);
        // band storage: row i holds the columns [i - Wb, i + up) (skewed view S(i, j) = base[i (LDB - 1) + j]).  The blocked Cholesky of
        // tsba_chol.h writes 96-wide blocks on both sides of the diagonal (up = CH_NB, Wb = band + CH_NB - 1); the streaming / partitioned
        // solvers read the band only (up = 6: the diagonal pose block is stored square) -- 72 instead of 251 columns per row at a band of
        // 60, and the band is cleared before every Schur assembly (60 MB per LM trial at 5000 keyframes with the wide rows)
        const bool stream_ok = bwmax >= 6 && bwmax <= BAND_BW_MAX && band_chunk_blocks(bwmax) > 0 && !c->dbg.no_band_stream;
        int ring = 0, ring_k0 = 0; for (int l = 0; l < p->n_levels; l++) if (c->lev_built[l] && c->hplan[l].ring) { ring = 1; ring_k0 = c->hplan[l].ring_k0; }     // (a ring plan is only built for single-level solves)
        int ring_G = 0;
        const size_t nrow = (size_t)W.N + (ring ? bwmax : 0);                                                 // + the ghost rows of the first separator
        c->S_up = stream_ok ? 6 : CH_NB;
        const size_t LDB = stream_ok ? (size_t)bwmax + 12 : (size_t)bwmax + 2*CH_NB - 1;
        if (use_lds_ || (size_t)bwmax + 2*CH_NB - 1 >= (size_t)W.N) { c->S_count = (size_t)(W.N + 1)*W.N; AL(c->S_alloc, c->S_count); W.S = c->S_alloc; W.ldS = W.N; W.band = 0; }
        else { c->S_count = nrow*LDB + LDB; AL(c->S_alloc, c->S_count); W.S = c->S_alloc + (LDB - c->S_up); W.ldS = (int)LDB - 1; W.band = 1; }
        W.ring = 0; W.ring_g = 0; W.ring_b = 0; W.ring_k0 = -1;
        c->Lcol = nullptr; c->band_stream = 0; c->sep_cr = false;
        if (W.band && stream_ok) {
            AL(c->Lcol, ((size_t)p->n_kf + bwmax/6 + 1)*bwmax*6); c->band_stream = 1;
            // substructuring: P interiors on P workgroups + a separator system (again a band, 2 bw - 6 wide)
            // number of interiors: the interiors run in parallel (n_kf / P blocks each, ~3.5 us per block, 5 us once the border makes the
            // panel waves take two rounds), the separator system is sequential again ((P - 1) B blocks at ~4.5 us, 5.5 us when its band
            // exceeds 115 rows): the sum is smallest near sqrt(n_kf t_f / (B t_s))
            const int Bq = bwmax/6;
            const double t_f = bwmax > 57 ? 5.0 : 3.5, t_s = 2*bwmax - 6 > 115 ? 5.5 : 4.5;
            int P = (int)lround(sqrt((double)p->n_kf*t_f/((double)std::max(Bq, 1)*t_s)));
            bool want_cr = false;
            if (bwmax <= CR_SMAX && c->dbg.sep_solver != 1) {
                // separator system by cyclic reduction (tsba_bandcre.h): its cost grows with log2(P) only (~45 us per level: one elimination
                // and one back-substitution launch; 130 us with the three kernels of round 1), so many more, shorter interiors pay.
                // Measured at 5000 keyframes / band 10 (ms per 20-iteration solve): P = 64 / 80 / 96 / 112 / 127 / 150 -> 20.6 / 20.3 / 19.4 /
                // 18.6 / 18.0 / 18.8 (150: an eighth level)
                double best = 1e300; int bestP = P;
                for (int q = 4; q <= BANDP_MAXP; q++) {
                    if ((p->n_kf - (q - 1)*Bq)/q < 2*Bq + 8) break;          // (the kernels need 2 B + 2 blocks per interior)
                    int lev = 1; for (int hh = 1; hh < q - 1; hh <<= 1) lev++;
                    const double cost = (double)p->n_kf/q*t_f + 45.0*lev;
                    if (cost < best) { best = cost; bestP = q; }
                }
                // (few, long interiors -- some hundred keyframes -- are still cheaper with the sequential separator solve: compare)
                const int Ps = std::max(1, std::min(P, BANDP_MAXP));
                const double cost_seq = (double)p->n_kf/Ps*t_f + (double)(Ps - 1)*Bq*t_s;
                if (best < cost_seq) { P = bestP; want_cr = true; }
            }
            if (c->dbg.sep_solver >= 2 && bwmax <= CR_SMAX) want_cr = true;
            if (c->dbg.band_parts > 0) P = c->dbg.band_parts;
            P = std::max(1, std::min(P, BANDP_MAXP));
            if (ring) {                       // ring: a power of two interiors in the loop (the separator tree ends in its first separator and the ghost), cyclic reduction only
                const int cap = c->dbg.band_parts > 0 ? c->dbg.band_parts : 128, nloop = p->n_kf - ring_k0;
                int Pr = 4; while (2*Pr <= cap && (nloop - 2*Pr*Bq)/(2*Pr) >= 2*Bq + 8) Pr *= 2;
                ring_G = Pr;
                int Pt = 0;                   // a tail before the loop: interiors of about the loop's size
                if (ring_k0 > 0) { const int ql = (nloop - Pr*Bq)/Pr; Pt = std::max(1, std::min(std::min(RING_OFF - 1, BANDP_MAXP - Pr), (ring_k0 + ql/2)/(ql + Bq)));
                    while (Pt > 1 && (ring_k0 - (Pt - 1)*Bq)/Pt < 2*Bq + 8) Pt--; }
                P = Pr + Pt; want_cr = true;
            }
            while (!ring && P > 1 && (p->n_kf - (P - 1)*Bq)/P < ((c->dbg.band_parts > 0 || want_cr) ? 2*Bq + 2 : 4*Bq + 4)) P--;     // (2 B + 2: the least the kernels take; the sequential separator solve pays only for interiors of a few bands)
            if (P > 1 && bandp_chunk_blocks(bwmax) > 0 && 2*bwmax - 6 <= BAND_BW_MAX && band_chunk_blocks(2*bwmax - 6) > 0) {
                const int nsepb = cr_mmax(ring, P, ring_G);                     // separator labels (ring: the last one is the ghost of the loop's first separator)
                const int nsep = nsepb*bwmax, bws = 2*bwmax - 6;
                c->nsep_ld = nsep;
                AL(c->Lb, ((size_t)p->n_kf + bwmax/6 + 1)*bwmax*6); AL(c->Tbuf, (size_t)P*((size_t)4*bwmax*bwmax + 2*bwmax));
                AL(c->Bpart, (size_t)P*BANDP_NS*((size_t)bwmax*bwmax + bwmax));
                c->sep_cr = want_cr && P >= 4;
                if (c->sep_cr) { AL(c->Ssep, cr_pool_blocks(nsepb)*(size_t)bwmax*bwmax); AL(c->CRcontrib, (size_t)nsepb*cre_contrib_doubles(bwmax)); AL(c->CRfac, (size_t)nsepb*cre_rec_doubles(bwmax)); }
                else AL(c->Ssep, (size_t)nsep*nsep + nsep);
                AL(c->Lcol_sep, (size_t)(nsep/6 + 1)*bws*6);
                Work &Ws = c->Wsep; memset(&Ws, 0, sizeof(Ws));
                Ws.N = nsep; Ws.n_kf = 0; Ws.S = c->Ssep; Ws.ldS = nsep; Ws.band = 1; Ws.st = nullptr;       // (st is set at launch: W.st is allocated below)
                AL(Ws.Sy, nsep); AL(Ws.g, nsep); AL(Ws.dp, nsep); AL(Ws.LDbuf, 32*(size_t)(nsep/6 + 1)); AL(Ws.nfree, 1); AL(Ws.fidx, 1);
                c->band_parts = P;
                W.ring = (ring && c->sep_cr) ? 1 : 0; W.ring_g = ring_G; W.ring_b = bwmax/6; W.ring_k0 = W.ring ? ring_k0 : -1;
            } else c->band_parts = 1;
        }
        if (ring && !W.ring) { set_err(c, "ring-shaped map: the partitioned band solver is not available for this plan"); return TSBA_ERR_STATE; }
        AL(W.Sy, W.N + BAND_BW_MAX);                            // (+ the ghost rows of a ring map)
        c->far_B = 0; c->n_far = 0; c->pcg_parts = 0;
        for (int l = 0; l < p->n_levels; l++) if (c->lev_built[l] && c->lev[l].far_B > 0) { c->far_B = c->lev[l].far_B; c->n_far = std::max(c->n_far, c->lev[l].n_far); }
        if (c->far_B > 0) {
            if (!W.band || !c->band_stream) { set_err(c, "map with long-range coupling: the band solvers are not available for this plan"); return TSBA_ERR_STATE; }
            c->pcg_parts = (p->n_kf + 31)/32;
            AL(W.Sfar, 36*(size_t)std::max(c->n_far, 1));
            AL(W.pc_x, W.N); AL(W.pc_r, W.N); AL(W.pc_p[0], W.N); AL(W.pc_p[1], W.N); AL(W.pc_q, W.N); AL(W.pc_g0, W.N);
            AL(W.pc_part, 2*(size_t)c->pcg_parts); AL(W.pcs, 2); AL(W.pc_stat, 4);
        }
        c->S_xchg = nullptr; c->xchg_wp = 0;
        if (W.band && is_multi(c)) { c->xchg_wp = std::min(W.N, bwmax + 6); AL(c->S_xchg, ((size_t)W.N + bwmax)*c->xchg_wp); }
    }
    AL(W.g, W.N); AL(W.dp, W.N); AL(W.dl_pt, p->n_pt); AL(W.dl_tx, 3*(size_t)p->n_text);
    AL(W.partial, 2*(size_t)c->nb_back_max);
    AL(W.posepart, 2*((size_t)p->n_kf/21 + 2));
    AL(W.cntpart, 2*(mx_cnt/4 + mx_cnt/256 + 4));
    AL(W.st, 1);
    AL(c->cov_log, 6*TSBA_MAX_LEVELS);
    flush_run(c);
    auto tu2 = std::chrono::steady_clock::now();
    if (hipStreamSynchronize(c->stream) != hipSuccess) { set_err(c, "upload sync failed"); return TSBA_ERR_DEVICE; }
    if (tdbg) { auto tu3 = std::chrono::steady_clock::now(); auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[tsba_upload] free %.2f ms, host total %.2f ms (waiting for the plan threads %.2f ms, image section %.2f ms), final sync %.2f ms\n", ms(tu0, tu1), ms(tu1, tu2), t_plan, t_img, ms(tu2, tu3)); }
    c->uploaded = true;
    joiner.armed = false;                          // (deferred levels: their plan threads are joined by stage_level / free_problem)
    if (!c->stage_p) join_planners(c);
    return TSBA_OK;
}
int tsba_upload(void *ctx, const tsba_problem *p, const tsba_options *o) { return upload_impl(ctx, p, o, false); }

// One pyramid level onto the device: the plan's lists (its host thread is joined here), the level's reference features, the table of image planes.
static int stage_level(Ctx *c, const tsba_problem *p, int l, double *t_plan, double *t_img) {
    const tsba_options *o = &c->opt; int rc;
    {   auto tp0 = std::chrono::steady_clock::now();            // in pass order: the coarse levels are ready first and are staged while level 0 is still being built
        if (l < (int)c->planners.size() && c->planners[l].joinable()) c->planners[l].join();
        if (t_plan) *t_plan += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count(); }
    HostPlan &H = c->hplan[l];
    LevelDev &D = c->lev[l]; memset(&D, 0, sizeof(D));
    D.level = l; D.n_sc = H.n_sc(); D.n_pair = H.n_pair(); D.n_tg = H.n_tg(); D.n_pslot = H.n_pslot(); D.n_tslot = H.n_tslot(); D.n_sb = H.n_sb(); D.bw_rows = 6*H.bw_pose;
    for (int k = 0; k < 4; k++) { double v = p->K[k]; for (int q = 0; q < l; q++) v *= 0.5; D.K[k] = v; }
    D.img_w = p->img_w[l]; D.img_h = p->img_h[l];
#define UV(field) do { rc = dev_upload_vec(c, &D.field, H.field); if (rc) return rc; } while (0)
    UV(sc_obs); UV(sc_kf); UV(sc_pt); UV(sc_flag); UV(sc_slot); UV(sc_uv);
    UV(tg_rec); UV(tg_ppos); UV(pt_pose6); UV(pt_pair4); UV(pair_i); UV(pair_h); UV(pair_hpos); UV(pair_sc_off); UV(pair_tg_off); UV(pair_tg);
    UV(tg_tobs); UV(tg_kf); UV(tg_text); UV(tg_pair); UV(tg_slot);
    UV(pf_g); UV(pf_f); D.n_pf = (int)H.pf_g.size();
    if (!H.kf_order.empty()) UV(kf_order); else D.kf_order = nullptr;
    D.far_B = H.far_B; D.n_far = H.n_far();
    D.sb_far = nullptr;
    if (H.far_B > 0) { UV(far_a); UV(far_b); UV(far_off); UV(far_ent); UV(fb_id); UV(fb_pab); UV(fb_pba); UV(fb_pt_off); UV(fb_pt_s1); UV(fb_pt_s2); UV(fb_pt_lm);
        UV(fb_tx_off); UV(fb_tx_s1); UV(fb_tx_s2); UV(fb_tx_lm); }
    UV(pls_off); UV(pslot_pose); UV(pslot_pair); UV(pslot_lm); UV(tls_off); UV(tslot_pose); UV(tslot_pair); UV(tslot_lm);
    UV(sb_a); UV(sb_b); UV(sb_pab); UV(sb_pba); UV(sb_pt_off); UV(sb_pt_s1); UV(sb_pt_s2); UV(sb_pt_lm); UV(sb_tx_off); UV(sb_tx_s1); UV(sb_tx_s2); UV(sb_tx_lm);
    UV(pose_t_off); UV(pose_t); UV(pose_h_off); UV(pose_h); UV(pose_ps_off); UV(pose_ps); UV(pose_ps_lm); UV(pose_ts_off); UV(pose_ts); UV(pose_ts_lm);
#undef UV
    if (p->n_text > 0 && p->tfeat_off[l]) {
        D.n_tfeat = p->n_tfeat[l];
        UP(D.tfeat_off, p->tfeat_off[l], (size_t)p->n_text + 1); UP(D.tfeat_raw, p->tfeat_raw[l], p->n_tfeat[l]);
        UP(D.tfeat_uv, p->tfeat_uv[l], 2*(size_t)p->n_tfeat[l]); UP(D.tfeat_ref, p->tfeat_ref[l], 8*(size_t)p->n_tfeat[l]);
    } else { std::vector<int32_t> z((size_t)p->n_text + 1, 0); UP(D.tfeat_off, z.data(), z.size()); }
    auto ti0 = std::chrono::steady_clock::now();
    if (o->use_text && p->n_tobs > 0 && p->img[l]) {
        std::vector<const uint8_t *> ptrs(p->n_kf, nullptr);
        size_t npx = (size_t)p->img_w[l]*p->img_h[l];
        const bool cached = c->use_img_cache;
        for (int k = 0; k < p->n_kf; k++) {                 // through the pinned staging mirror: one copy for the whole level
            if (!p->img[l][k]) { set_err(c, "null image pointer"); return TSBA_ERR_ARG; }
            if (o->img_on_device) { ptrs[k] = p->img[l][k]; continue; }   // resident pyramid plane (tsframe_level_ptr): used in place
            if (cached) { ptrs[k] = c->ic.dev + (size_t)c->ic_slot[(size_t)k]*c->ic.slot + c->ic.lvl_off[l]; continue; }   // the keyframe's slot of the plane cache (filled by the upload)
            rc = dev_upload(c, &ptrs[k], p->img[l][k], npx); if (rc) return rc;
        }
        const uint8_t *const *dptr = nullptr;
        rc = dev_upload(c, &dptr, (const uint8_t *const *)ptrs.data(), ptrs.size()); if (rc) return rc;
        D.img = (const uint8_t *const *)dptr;
    }
    if (t_img) *t_img += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti0).count();
    flush_run(c);
    if (c->stage_async && c->copy_stream && c->ev_stage[l]) { hipEventRecord(c->ev_stage[l], c->copy_stream); c->lev_wait[l] = 1; }      // the level's pass waits for this copy
    c->lev_built[l] = 1;
    return TSBA_OK;
}
// during a solve: levels of later passes whose plans are complete go to the device now, over the copy stream, next to the running pass
static int stage_ahead(Ctx *c, int ps) {
    if (!c->stage_p) return TSBA_OK;
    for (int q = ps + 1; q < c->opt.n_passes; q++) { const int l = c->opt.levels[q];
        if (c->lev_built[l] || !c->lev_planned[l]) continue;
        if (!c->plan_done[l].load(std::memory_order_acquire)) break;           // in pass order
        c->stage_async = true; const int rc = stage_level(c, c->stage_p, l, nullptr, nullptr); c->stage_async = false;
        if (rc) return rc; }
    return TSBA_OK;
}

static int reset_state(Ctx *c) {                 // one launch instead of ten small copies (each ~2.5 us on the stream)
    Work &W = c->W;
    ResetSrc A = { (const double *)c->pose0, (const double *)c->rho0, (const double *)c->theta0,
                   (const uint8_t *)c->sgood0, (const uint8_t *)c->tobs_good0, (const uint8_t *)c->tfgood0,
                   (long long)7*c->n_kf, (long long)c->n_pt, (long long)3*c->n_text, (long long)c->n_sgood, (long long)c->n_tobs, (long long)c->n_tfgood };
    long long mx = std::max(std::max(A.n_pose, A.n_rho), std::max(std::max(A.n_theta, A.n_sg), std::max(A.n_tg, A.n_tf)));
    const int nb = (int)std::min<long long>(1024, std::max<long long>(1, (mx + 255)/256));
    hipLaunchKernelGGL(k_reset_state, dim3(nb), dim3(256), 0, c->stream, W, A);
    return 0;
}

static bool is_multi(const Ctx *c) { return c->world > 1 || c->force_multi; }

// In-process communicator (tsba_comm_init_local): `world` contexts of one process, one host thread each.  A collective is
// stream-sync -> device-to-host -> barrier -> rank 0 reduces in rank order (deterministic) -> barrier -> host-to-device.
// It exists so that the N > 1 code path -- sharded upload, split kernel sequence, every exchange -- runs under the test-suite on a
// one-GPU box; production multi-GPU runs use RCCL (tsba_comm_init).
struct LocalGroup {
    int world = 1; bool broken = false;
    std::vector<Ctx *> members;                    // contexts that joined (tsba_comm_init_local): their lgroup is cleared when the group goes away
    std::mutex m; std::condition_variable cv; int arrived = 0; unsigned long long gen = 0;
    std::vector<std::vector<char>> stage; std::vector<char> result;
    // The ranks of a group usually share ONE device.  A rank holds this token while it has kernels in flight (from the moment it leaves a
    // collective until its stream has drained at the next one), so that the ranks' launches do not overlap on the device: kernel
    // durations under a profiler are then those of a rank that has the GPU to itself, as in a real multi-GPU run.
    std::mutex gpu_token;
    void fail() { std::lock_guard<std::mutex> lk(m); broken = true; cv.notify_all(); }     // a member gives up: the others must not wait for it
    bool barrier() {                               // false: a member never arrived (it failed before the collective) -- do not hang
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const unsigned long long g = gen;
        if (++arrived == world) { arrived = 0; gen++; cv.notify_all(); return true; }
        if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return gen != g || broken; })) { broken = true; cv.notify_all(); return false; }
        return !broken;
    }
};
static void lgroup_forget(Ctx *c) {              // the context leaves its in-process group (destroyed, or joined to an RCCL communicator)
    if (!c->lgroup) return;
    { std::lock_guard<std::mutex> lk(c->lgroup->m); for (Ctx *&m : c->lgroup->members) if (m == c) m = nullptr; }
    c->lgroup = nullptr;
}
static void local_allreduce(Ctx *c, void *buf, size_t count, ncclDataType_t dt, ncclRedOp_t op) {
    LocalGroup *G = c->lgroup;
    const size_t bytes = count*(dt == ncclDouble ? sizeof(double) : sizeof(int));
    auto fail = [&](const char *what) { c->err = std::string("ncclAllReduce (local group): ") + what; G->fail(); };
    if (hipStreamSynchronize(c->stream) != hipSuccess) { fail("stream"); return; }
    std::vector<char> &mine = G->stage[c->rank];
    mine.resize(bytes);
    if (hipMemcpy(mine.data(), buf, bytes, hipMemcpyDeviceToHost) != hipSuccess) { fail("device-to-host"); return; }
    if (c->has_token) { c->has_token = false; G->gpu_token.unlock(); }          // this rank's kernels have drained: the next rank may run
    struct Retake { Ctx *c; LocalGroup *G; ~Retake() { if (c->in_solve && !c->has_token) { G->gpu_token.lock(); c->has_token = true; } } } retake{c, G};
    if (!G->barrier()) { fail("a rank did not arrive"); return; }
    if (c->rank == 0) {
        G->result = G->stage[0];
        for (int r = 1; r < G->world; r++) {
            if (G->stage[r].size() != bytes) { G->fail(); break; }              // ranks disagree on the count: a layout bug, never sum garbage
            if (dt == ncclDouble) { double *a = (double *)G->result.data(); const double *b = (const double *)G->stage[r].data();
                if (op == ncclMax) for (size_t k = 0; k < count; k++) a[k] = a[k] > b[k] ? a[k] : b[k]; else for (size_t k = 0; k < count; k++) a[k] += b[k]; }
            else { int *a = (int *)G->result.data(); const int *b = (const int *)G->stage[r].data(); for (size_t k = 0; k < count; k++) a[k] += b[k]; }
        }
    }
    if (!G->barrier()) { fail("ranks disagree on the element count, or a rank did not arrive"); return; }
    if (hipMemcpy(buf, G->result.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) fail("host-to-device");
}
static void allreduce(Ctx *c, void *buf, size_t count, ncclDataType_t dt, ncclRedOp_t op) {
    c->x_acc += count*(dt == ncclDouble ? sizeof(double) : sizeof(int));
    if (c->lgroup) { local_allreduce(c, buf, count, dt, op); return; }
    if (!c->comm) return;                          // force_multi without a communicator: exercises the split kernels only
    ncclResult_t r = c->p_allreduce(buf, buf, count, dt, op, c->comm, c->stream);
    if (r != ncclSuccess) c->err = std::string("ncclAllReduce: ") + c->p_errstr(r);
}
static void launch_pass_init(Ctx *c, const LevelDev &D, int pass) {
    const size_t x0 = c->x_acc;
    struct XP { Ctx *c; size_t x0; ~XP() { c->x_pass = c->x_acc - x0; } } xp{c, x0};
    c->cur_bw_rows = D.bw_rows; c->S_stale = true;             // (a new pass: other free poses, other entries of S)
    c->W.hprog = c->hprog; c->W.pass_seq = ++c->pass_seq;
    Work &W = c->W; const tsba_options &o = c->opt;
    hipLaunchKernelGGL(k_pass_reset, dim3(64), dim3(256), 0, c->stream, W, o.initial_radius, o.its[pass]);
    int n = D.n_sc + D.n_tg;
    const int npb = (D.n_sc + 255)/256 + (D.n_tg + 3)/4;                    // k_participation workgroups: scene candidates | text groups
    const int ncp = (n > 0 && !is_multi(c)) ? npb : 0;                      // count partials (single GPU)
    if (n > 0) hipLaunchKernelGGL(k_participation, dim3(npb), dim3(256), 0, c->stream, W, D, ncp ? 1 : 0);
    if (is_multi(c)) {                             // participation and block counts are global properties
        allreduce(c, W.kf_in, c->n_kf, ncclInt32, ncclSum);
        allreduce(c, &W.st->ns_active, 2, ncclInt32, ncclSum);
        hipLaunchKernelGGL(k_kfin_multi, dim3((c->n_kf + 255)/256), dim3(256), 0, c->stream, W);
    }
    if (c->n_kf <= 64) hipLaunchKernelGGL(k_gauge_wave, dim3(1), dim3(64), 0, c->stream, W, (const uint8_t *)c->kf_initial, o.state, ncp);
    else if (c->n_kf > 256) hipLaunchKernelGGL(k_gauge_par, dim3(1), dim3(1024), 0, c->stream, W, (const uint8_t *)c->kf_initial, o.state, ncp, D.kf_order);
    else hipLaunchKernelGGL(k_gauge, dim3(1), dim3(64), 0, c->stream, W, (const uint8_t *)c->kf_initial, o.state, ncp, D.kf_order);
    if (D.n_tg > 0) hipLaunchKernelGGL(k_musigma, dim3(D.n_tg), dim3(MS_THREADS), 0, c->stream, W, D);
}
static int pose_parts(const Ctx *c) { return c->n_kf > 126 ? (c->n_kf + 20)/21 : 0; }    // k_pose_sums workgroups (0: the pose sums stay in k_postlin / k_decide)
// pairs with a dozen scene blocks (large maps): four pairs per wave
static bool lin_small_pairs(const Ctx *c, const LevelDev &D) { return !c->dbg.no_small_pairs && D.n_pair > 0 && (long long)D.n_sc <= 24LL*D.n_pair; }
static void launch_linearize(Ctx *c, const LevelDev &D, int spec) {
    struct XL { Ctx *c; size_t x0; ~XL() { c->x_lin = c->x_acc - x0; } } xl{c, c->x_acc};
    Work &W = c->W;
    int nb_pt = (c->n_pt + 255)/256, nb_tx = (c->n_text + 255)/256, nb_pr = (D.n_pair + 255)/256, nb_kf = (c->n_kf + 255)/256;
    if (D.n_pair + D.n_tg > 0) {
        if (lin_small_pairs(c, D) && D.n_tg == 0) hipLaunchKernelGGL((k_linearize<MODE_FULL, 4, false>), dim3((((D.n_pair + 4*LIN_NWV - 1)/(4*LIN_NWV) + 7)/8)*8), dim3(LIN_T), 0, c->stream, W, D, spec);
        else if (lin_small_pairs(c, D)) hipLaunchKernelGGL((k_linearize<MODE_FULL, 4>), dim3((((D.n_pair + 4*LIN_NWV - 1)/(4*LIN_NWV) + D.n_tg + 7)/8)*8), dim3(LIN_T), 0, c->stream, W, D, spec);
        else hipLaunchKernelGGL((k_linearize<MODE_FULL, 1>), dim3((((D.n_pair + LIN_NWV - 1)/LIN_NWV + D.n_tg + 7)/8)*8), dim3(LIN_T), 0, c->stream, W, D, spec);
    }
    hipLaunchKernelGGL(k_mid, dim3(nb_pt + nb_tx + nb_pr), dim3(256), 0, c->stream, W, D, nb_pt, nb_tx, spec);
    const int multi = is_multi(c);
    const int npp = pose_parts(c);
    if (multi) {
        // local sums -> exchange buffer -> all-reduce; the consumer (k_postlin / k_decide) installs them into the right LinBuf
        if (npp) hipLaunchKernelGGL(k_pose_sums_raw, dim3(npp), dim3(256), 0, c->stream, W, D, spec);
        hipLaunchKernelGGL(k_sums_multi, dim3(1), dim3(256), 0, c->stream, W, D, spec, nb_pt + nb_tx + nb_pr, nb_pt + nb_tx + nb_kf, nb_pt + nb_tx, npp);
        allreduce(c, W.cb, 2*(size_t)W.N + 8, ncclDouble, ncclSum);
        allreduce(c, W.cbm, 1, ncclDouble, ncclMax);
        if (npp) hipLaunchKernelGGL(k_pose_scale_multi, dim3(npp), dim3(256), 0, c->stream, W, spec);
    } else if (npp) hipLaunchKernelGGL(k_pose_sums, dim3(npp), dim3(256), 0, c->stream, W, D, spec);
    if (!spec) hipLaunchKernelGGL(k_postlin, dim3(1), dim3(256), 0, c->stream, W, D, c->opt.gradient_tolerance, nb_pt + nb_tx + nb_pr, multi, npp);
}
static int solve_lds_bytes(Ctx *c, int *use_lds) {
    size_t bytes = solve_lds_doubles(c->W.N)*sizeof(double);                                // worst case: every pose free
    *use_lds = bytes <= 160*1024 - 64;                                                      // gfx950: 160 KB of LDS per workgroup
    return *use_lds ? (int)bytes : 0;
}
static void launch_schur(Ctx *c, const LevelDev &D, int multi) {
    if (c->n_kf > 126 && !c->dbg.no_schur_quad) {               // large maps: four S blocks per wave, then one wave per pose for the reduced gradient
        if (D.n_sb > 0) { const int nq = (((D.n_sb + 3)/4 + 7)/8)*8;           // (a multiple of 8 workgroups: the kernel's XCD-aware block mapping)
            if (D.n_tg > 0) hipLaunchKernelGGL(k_schur_quad<true>, dim3(nq), dim3(64), 0, c->stream, c->W, D, multi);
            else hipLaunchKernelGGL(k_schur_quad<false>, dim3(nq), dim3(64), 0, c->stream, c->W, D, multi); }
        hipLaunchKernelGGL(k_schur_t<1>, dim3(((c->n_kf + 7)/8)*8), dim3(64), 0, c->stream, c->W, D, multi, D.n_sb);
    } else if (c->n_kf > 126) hipLaunchKernelGGL(k_schur_t<1>, dim3(D.n_sb + c->n_kf), dim3(64), 0, c->stream, c->W, D, multi, 0);
    else hipLaunchKernelGGL(k_schur_t<4>, dim3(D.n_sb + c->n_kf), dim3(256), 0, c->stream, c->W, D, multi, 0);
    if (D.far_B > 0 && D.n_far > 0) {            // the blocks of E (what couples different clusters of a landmark): the same kernels on the fb_* lists, stored to W.Sfar
        LevelDev E = D;
        E.n_sb = D.n_far; E.sb_a = D.far_a; E.sb_b = D.far_b; E.sb_pab = D.fb_pab; E.sb_pba = D.fb_pba; E.sb_far = D.fb_id;
        E.sb_pt_off = D.fb_pt_off; E.sb_pt_s1 = D.fb_pt_s1; E.sb_pt_s2 = D.fb_pt_s2; E.sb_pt_lm = D.fb_pt_lm;
        E.sb_tx_off = D.fb_tx_off; E.sb_tx_s1 = D.fb_tx_s1; E.sb_tx_s2 = D.fb_tx_s2; E.sb_tx_lm = D.fb_tx_lm;
        if (c->n_kf > 126 && !c->dbg.no_schur_quad) { const int nq = (((E.n_sb + 3)/4 + 7)/8)*8;
            if (D.n_tg > 0) hipLaunchKernelGGL(k_schur_quad<true>, dim3(nq), dim3(64), 0, c->stream, c->W, E, multi);
            else hipLaunchKernelGGL(k_schur_quad<false>, dim3(nq), dim3(64), 0, c->stream, c->W, E, multi); }
        else if (c->n_kf > 126) hipLaunchKernelGGL(k_schur_t<1>, dim3(E.n_sb), dim3(64), 0, c->stream, c->W, E, multi, 0);
        else hipLaunchKernelGGL(k_schur_t<4>, dim3(E.n_sb), dim3(256), 0, c->stream, c->W, E, multi, 0);
    }
}
// kernels with more than 64 KB of dynamic LDS need the attribute once per process
static int set_solver_attrs(Ctx *c) {
    int use_lds; int lds = solve_lds_bytes(c, &use_lds);
    if (use_lds) CK(hipFuncSetAttribute((const void *)k_solve_t<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    else {
        CK(hipFuncSetAttribute((const void *)k_band_solve, hipFuncAttributeMaxDynamicSharedMemorySize, 156*1024));
        CK(hipFuncSetAttribute((const void *)k_bandp_factor, hipFuncAttributeMaxDynamicSharedMemorySize, 156*1024));
        CK(hipFuncSetAttribute((const void *)k_cr_pivot, hipFuncAttributeMaxDynamicSharedMemorySize, 156*1024));
        CK(hipFuncSetAttribute((const void *)k_cr_update, hipFuncAttributeMaxDynamicSharedMemorySize, 156*1024));
        CK(hipFuncSetAttribute((const void *)k_cr_back, hipFuncAttributeMaxDynamicSharedMemorySize, 156*1024));
        CK(hipFuncSetAttribute((const void *)k_cre_elim, hipFuncAttributeMaxDynamicSharedMemorySize, 156*1024));
        CK(hipFuncSetAttribute((const void *)k_bandp_sepf, hipFuncAttributeMaxDynamicSharedMemorySize, 156*1024));
        CK(hipFuncSetAttribute((const void *)k_cre_back, hipFuncAttributeMaxDynamicSharedMemorySize, 156*1024));
        CK(hipFuncSetAttribute((const void *)k_ms_cre_back, hipFuncAttributeMaxDynamicSharedMemorySize, 159*1024));
        CK(hipFuncSetAttribute((const void *)k_ms_cre_fwd, hipFuncAttributeMaxDynamicSharedMemorySize, 100*1024));
        CK(hipFuncSetAttribute((const void *)k_ms_cre_root, hipFuncAttributeMaxDynamicSharedMemorySize, 100*1024));
        CK(hipFuncSetAttribute((const void *)k_bandp_backsub<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156*1024));
        CK(hipFuncSetAttribute((const void *)k_bandp_backsub<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 156*1024));
        CK(hipFuncSetAttribute((const void *)k_band_backsub<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 156*1024));
        CK(hipFuncSetAttribute((const void *)k_band_backsub<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 156*1024));
        CK(hipFuncSetAttribute((const void *)k_band_backsub<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 156*1024));
        CK(hipFuncSetAttribute((const void *)k_solve_t<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(solve_diag_lds_doubles()*sizeof(double))));
        CK(hipFuncSetAttribute((const void *)k_chol_panel, hipFuncAttributeMaxDynamicSharedMemorySize, (CH_NB + 64)*(CH_NB + 1)*(int)sizeof(double)));
        CK(hipFuncSetAttribute((const void *)k_chol_update, hipFuncAttributeMaxDynamicSharedMemorySize, 2*64*(CH_NB + 1)*(int)sizeof(double)));
        CK(hipFuncSetAttribute((const void *)k_chol_backsub, hipFuncAttributeMaxDynamicSharedMemorySize, (CH_NB*(CH_NB + 1) + 2*CH_NB + 8*CH_NB)*(int)sizeof(double)));
    }
    return 0;
}
// dense solve of the reduced camera system: LDS kernel for small windows, multi-workgroup blocked Cholesky otherwise
static void launch_solve(Ctx *c) {
    Work &W = c->W;
    int use_lds; int lds = solve_lds_bytes(c, &use_lds);
    if (use_lds) { hipLaunchKernelGGL(k_solve_t<false>, dim3(1), dim3(SOLVE_THREADS), lds, c->stream, W, 0); return; }
    if (c->band_stream && c->band_parts > 1) {      // partitioned: interiors in parallel + separator system (tsba_bandp.h)
        const int bwp = std::max(6, c->cur_bw_rows), cbp = bandp_chunk_blocks(bwp), P = c->band_parts;
        const int bwsep = 2*bwp - 6, cbs = band_chunk_blocks(bwsep);
        Work &Ws = c->Wsep; Ws.st = W.st; Ws.ldS = (P - 1)*bwp; Ws.N = (P - 1)*bwp;
        if (!c->sep_cr) hipMemsetAsync(c->Ssep, 0, sizeof(double)*((size_t)Ws.ldS*Ws.ldS + Ws.ldS), c->stream);
        hipLaunchKernelGGL(k_bandp_factor, dim3(P), dim3(SOLVE_THREADS), (int)(bandp_lds_doubles(bwp, cbp)*sizeof(double)), c->stream, W, bwp, cbp, P, c->Lcol, c->Lb, c->Tbuf);
        if (c->sep_cr && (W.ring || (c->dbg.sep_solver != 3 && c->dbg.sep_solver != 4)))      // block pool: border products + separator assembly in one launch (4: the three launches, for A/B runs)
            hipLaunchKernelGGL(k_bandp_sepf, dim3(W.ring ? P + 1 : P - 1), dim3(BSF_T), (int)(bandp_sepf_lds_doubles()*sizeof(double)), c->stream, W, bwp, P, (const double *)c->Tbuf, (const double *)c->Lb, c->Ssep, Ws.g, Ws.nfree);
        else {
        hipMemsetAsync(c->Bpart, 0, sizeof(double)*(size_t)P*BANDP_NS*((size_t)bwp*bwp + bwp), c->stream);        // (slices of short interiors stay empty)
        hipLaunchKernelGGL(k_bandp_border, dim3(P, BANDP_NS), dim3(256), (int)((2*(size_t)BANDP_JC*bwp*6 + 6*BANDP_JC)*sizeof(double)), c->stream, W, bwp, P, (const double *)c->Lb, c->Bpart);
        hipLaunchKernelGGL(k_bandp_sep, dim3(P - 1), dim3(256), 0, c->stream, W, bwp, P, (const double *)c->Tbuf, (const double *)c->Bpart, c->Ssep, Ws.ldS, Ws.g, Ws.nfree, (int)c->sep_cr);
        }
        if (c->sep_cr) {                  // separator system by block cyclic reduction (tsba_bandcr.h): log2(P - 1) levels
            const int mmax = cr_mmax(W.ring, P, W.ring_g);
            int mlev = mmax;                  // levels h < mlev.  Ring: the loop's separators need h <= G/2 (the root and the ghost are merged at the root, no level for them),
            if (W.ring) { mlev = W.ring_g;    // a tail's separator RING_OFF - j the level of the lowest set bit of j (j < the number of tail interiors)
                for (int hh = 1; hh < P - W.ring_g; hh <<= 1) mlev = std::max(mlev, 2*hh); }
            const int lab0 = W.ring && P > W.ring_g ? RING_OFF - (P - W.ring_g) + 1 : 0;          // lowest separator label (a ring with a tail counts down from RING_OFF)
            const int lp = (int)(cr_pivot_lds_doubles(bwp)*sizeof(double)), lu = (int)(cr_update_lds_doubles(bwp)*sizeof(double)), lb = (int)(cr_back_lds_doubles(bwp)*sizeof(double));
            int htop = 1;
            if (c->dbg.sep_solver != 3 || W.ring) {      // one launch per level (tsba_bandcre.h); 3: the pivot / update / back kernels of tsba_bandcr.h
                const int le = (int)(cre_elim_lds_doubles(bwp)*sizeof(double)), lbk = (int)(cre_back_lds_doubles(bwp)*sizeof(double));
                auto pivots = [&](int h, int &kb) { kb = lab0/(2*h); const int klast = (mmax - 1 - h)/(2*h); return std::max(0, klast - kb + 1); };     // pivots (2 k + 1) h, k = kb ..
                for (int h = 1; h < mlev; h <<= 1) {
                    int kb; const int npiv = pivots(h, kb); if (npiv <= 0) { htop = h; continue; }
                    const int K = std::max(1, std::min(TSBA_CRE_KMAX, 224/npiv));     // workgroups per pivot (they share its product and stores)
                    hipLaunchKernelGGL(k_cre_elim, dim3(npiv*K), dim3(CRE_T), le, c->stream, W, Ws, bwp, P, h, 0, K, kb, c->CRcontrib, c->CRfac); htop = h; }
                hipLaunchKernelGGL(k_cre_elim, dim3(1), dim3(CRE_T), le, c->stream, W, Ws, bwp, P, 0, W.ring ? 2 : 1, 1, 0, c->CRcontrib, c->CRfac);
                for (int h = htop; h >= 1; h >>= 1) { int kb; const int npiv = pivots(h, kb); if (npiv > 0) hipLaunchKernelGGL(k_cre_back, dim3(npiv), dim3(CRE_BT), lbk, c->stream, W, Ws, bwp, P, h, kb, (const double *)c->CRfac); }
            } else {
            for (int h = 1; h < mmax; h <<= 1) {
                const int npiv = (mmax + 2*h - 1)/(2*h);           // >= the pivots (2k + 1) h < m; workgroups past the end return
                hipLaunchKernelGGL(k_cr_pivot, dim3(npiv), dim3(CR_T), lp, c->stream, W, Ws, bwp, P, h, 0);
                hipLaunchKernelGGL(k_cr_update, dim3(2*npiv + 1), dim3(CR_T), lu, c->stream, W, Ws, bwp, P, h, npiv);
                htop = h;
            }
            hipLaunchKernelGGL(k_cr_pivot, dim3(1), dim3(CR_T), lp, c->stream, W, Ws, bwp, P, 0, 1);
            hipLaunchKernelGGL(k_cr_back, dim3(1), dim3(CR_T), lb, c->stream, W, Ws, bwp, P, 0, 1);
            for (int h = htop; h >= 1; h >>= 1)
                hipLaunchKernelGGL(k_cr_back, dim3((mmax + 2*h - 1)/(2*h)), dim3(CR_T), lb, c->stream, W, Ws, bwp, P, h, 0);
            }
        } else {
        const int ldss = (int)(band_lds_doubles(bwsep, cbs)*sizeof(double)), nus = (bwsep + 63)/64;
        hipLaunchKernelGGL(k_band_solve, dim3(1), dim3(SOLVE_THREADS), ldss, c->stream, Ws, bwsep, cbs, c->Lcol_sep);
        if (nus <= 1) hipLaunchKernelGGL(k_band_backsub<1>, dim3(1), dim3(BAND_BS_T), ldss, c->stream, Ws, bwsep, (const double *)c->Lcol_sep);
        else if (nus == 2) hipLaunchKernelGGL(k_band_backsub<2>, dim3(1), dim3(BAND_BS_T), ldss, c->stream, Ws, bwsep, (const double *)c->Lcol_sep);
        else hipLaunchKernelGGL(k_band_backsub<3>, dim3(1), dim3(BAND_BS_T), ldss, c->stream, Ws, bwsep, (const double *)c->Lcol_sep);
        }
        const int nup = (bwp + 63)/64, ldsp = (int)((2*(size_t)BAND_CK*(2*(size_t)bwp*6 + 32) + 6*BAND_RINGB + 2*bwp + 64)*sizeof(double));
        if (nup <= 1) hipLaunchKernelGGL(k_bandp_backsub<1>, dim3(P), dim3(BAND_BS_T), ldsp, c->stream, W, bwp, P, (const double *)c->Lcol, (const double *)c->Lb, (const double *)Ws.Sy);
        else hipLaunchKernelGGL(k_bandp_backsub<2>, dim3(P), dim3(BAND_BS_T), ldsp, c->stream, W, bwp, P, (const double *)c->Lcol, (const double *)c->Lb, (const double *)Ws.Sy);
        hipLaunchKernelGGL(k_bandp_dp, dim3((W.n_kf + 255)/256), dim3(256), 0, c->stream, W);
        return;
    }
    if (c->band_stream) {                                          // narrow band: one workgroup streams down the band (tsba_band.h)
        const int bws = std::max(6, c->cur_bw_rows), cb = band_chunk_blocks(bws);
        if (c->dbg.verbose) fprintf(stderr, "[launch_solve] band stream bw %d cb %d lds %zu B\n", bws, cb, band_lds_doubles(bws, cb)*sizeof(double));
        hipLaunchKernelGGL(k_band_solve, dim3(1), dim3(SOLVE_THREADS), (int)(band_lds_doubles(bws, cb)*sizeof(double)), c->stream, W, bws, cb, c->Lcol);
        const int nu = (bws + 63)/64, ldsb = (int)(band_lds_doubles(bws, cb)*sizeof(double));      // tasks per lane of the back substitution
        if (nu <= 1) hipLaunchKernelGGL(k_band_backsub<1>, dim3(1), dim3(BAND_BS_T), ldsb, c->stream, W, bws, (const double *)c->Lcol);
        else if (nu == 2) hipLaunchKernelGGL(k_band_backsub<2>, dim3(1), dim3(BAND_BS_T), ldsb, c->stream, W, bws, (const double *)c->Lcol);
        else hipLaunchKernelGGL(k_band_backsub<3>, dim3(1), dim3(BAND_BS_T), ldsb, c->stream, W, bws, (const double *)c->Lcol);
        return;
    }
    const int N = W.N;                                             // worst case: every keyframe free
    const int bw = std::min(c->cur_bw_rows, N);                    // band of the reduced camera matrix (rows below a pose block)
    hipLaunchKernelGGL(k_chol_rhs, dim3((N + 255)/256), dim3(256), 0, c->stream, W);
    const int lds_diag = (int)(solve_diag_lds_doubles()*sizeof(double));
    const int lds_panel = (CH_NB + 64)*(CH_NB + 1)*(int)sizeof(double);
    const int lds_upd = 2*64*(CH_NB + 1)*(int)sizeof(double);
    for (int j0 = 0; j0 < N; j0 += CH_NB) {
        hipLaunchKernelGGL(k_solve_t<true>, dim3(1), dim3(SOLVE_THREADS), lds_diag, c->stream, W, j0);
        // the host only knows the worst case n = N; a shorter last block (nb < NB) still has the rhs row below it
        const int wr = std::max(0, std::min(bw, N - (j0 + 6)));        // band rows below the block, + 1 for the rhs row
        hipLaunchKernelGGL(k_chol_panel, dim3(wr/64 + 1), dim3(CH_T), lds_panel, c->stream, W, j0, bw);
        const int nt = (wr + 1 + 63)/64;
        if (wr > 0) hipLaunchKernelGGL(k_chol_update, dim3(nt*(nt + 1)/2), dim3(CH_T), lds_upd, c->stream, W, j0, bw);
    }
    const int lds_bs = (CH_NB*(CH_NB + 1) + 2*CH_NB + 8*CH_NB)*(int)sizeof(double);
    hipLaunchKernelGGL(k_chol_backsub, dim3(1), dim3(1024), lds_bs, c->stream, W, bw);
}

// ---- solve phase of the partitioned band solver for T right-hand sides (tsba_bandms.h): needs the factor of the last launch_solve of this level
static bool ms_available(const Ctx *c) { return c->band_stream && c->band_parts > 1 && c->sep_cr && !c->W.ring && c->dbg.sep_solver != 3; }
static int ms_reserve(Ctx *c, int T) {           // buffers for T columns (kept until a larger request or another problem size)
    const size_t n6 = (size_t)c->W.N, labels = (size_t)cr_mmax(0, c->band_parts, 0) + 1, sdim = (size_t)std::max(6, c->cur_bw_rows);
    const size_t per = 4*n6 + 5*labels*sdim, need = per*(size_t)T*sizeof(double);
    if (need > c->ms_bytes) { if (c->ms_alloc) { hipStreamSynchronize(c->stream); hipFree(c->ms_alloc); } c->ms_alloc = nullptr; c->ms_bytes = 0;
        if (hipMalloc((void **)&c->ms_alloc, need) != hipSuccess) { set_err(c, "hipMalloc (multi-right-hand-side buffers)"); return TSBA_ERR_DEVICE; }
        c->ms_bytes = need; }
    double *q = c->ms_alloc; MsBuf &M = c->ms; M.T = T;
    M.R = q; q += n6*T; M.Wm = q; q += n6*T; M.V = q; q += n6*T; M.X = q; q += n6*T;
    M.G = q; q += labels*sdim*T; M.Z = q; q += labels*sdim*T; M.Xs = q; q += labels*sdim*T; M.Cg = q;
    c->ms_cap = T;
    return TSBA_OK;
}
static void launch_ms_solve(Ctx *c) {            // M.R -> M.X
    Work &W = c->W; const MsBuf &M = c->ms;
    const int bwp = std::max(6, c->cur_bw_rows), P = c->band_parts, B = bwp/6, ncg = (M.T + 63)/64;
    Work &Ws = c->Wsep; Ws.st = W.st;
    const size_t ldsf = ms_cre_lds_doubles(bwp, 1)*sizeof(double), ldsb = (ms_cre_lds_doubles(bwp, 3) + 8*(size_t)(bwp + 2))*sizeof(double);
    hipLaunchKernelGGL(k_ms_fwd_int, dim3(P, ncg), dim3(64), 0, c->stream, W, bwp, P, (const double *)c->Lcol, M);
    hipLaunchKernelGGL(k_ms_sep_rhs, dim3((P - 1)*B, ncg), dim3(64), 0, c->stream, W, bwp, P, (const double *)c->Lcol, (const double *)c->Lb, M);
    const int mmax = cr_mmax(0, P, 0);
    auto pivots = [&](int h, int &kb) { kb = 0; const int klast = (mmax - 1 - h)/(2*h); return mmax - 1 - h < 0 ? 0 : std::max(0, klast + 1); };
    int htop = 0;
    for (int h = 1; h < mmax; h <<= 1) { int kb; const int npiv = pivots(h, kb); if (npiv <= 0) continue;
        hipLaunchKernelGGL(k_ms_cre_fwd, dim3(npiv, ncg), dim3(MS_CT), ldsf, c->stream, W, Ws, bwp, P, h, kb, (const double *)c->CRfac, M); htop = h; }
    hipLaunchKernelGGL(k_ms_cre_root, dim3(1, ncg), dim3(256), ldsf, c->stream, W, Ws, bwp, P, (const double *)c->CRfac, M);
    for (int h = htop; h >= 1; h >>= 1) { int kb; const int npiv = pivots(h, kb);
        if (npiv > 0) hipLaunchKernelGGL(k_ms_cre_back, dim3(npiv, ncg), dim3(MS_CT), ldsb, c->stream, W, Ws, bwp, P, h, kb, (const double *)c->CRfac, M); }
    hipLaunchKernelGGL(k_ms_back_border, dim3(c->n_kf, ncg), dim3(64), 0, c->stream, W, bwp, P, (const double *)c->Lb, M);
    hipLaunchKernelGGL(k_ms_back_int, dim3(P, ncg), dim3(64), 0, c->stream, W, bwp, P, (const double *)c->Lcol, M);
}

// The reduced system of one LM trial: a direct solve, or -- band + long-range blocks -- conjugate gradients preconditioned with the band
// solver (tsba_pcg.h).  The host enqueues iteration k only once the device has reached iteration k - 2 (pinned progress word), so a solve
// that converges wastes two iterations of empty launches; every rank of a sharded run iterates on its own copy of the summed system.
static void launch_solve_full(Ctx *c, const LevelDev &D) {
    launch_solve(c);
    if (D.far_B <= 0) return;
    Work &W = c->W;
    const int nbp = c->pcg_parts, B = std::max(6, c->cur_bw_rows)/6;
    const int cap = c->dbg.pcg_max_it > 0 ? c->dbg.pcg_max_it : 200;
    const double tol = c->dbg.pcg_tol_exp > 0 ? pow(10.0, -(double)c->dbg.pcg_tol_exp) : 1e-10, tol2 = tol*tol;
    const unsigned int seq = ++c->pcg_seq;
    // Enlarged conjugate gradients on the solve phase of the band solver (ECG_T columns per application of M^-1): the default where that phase exists
    // (measured at 5000 keyframes, ms per solve, single vector / enlarged: two loop closures 410 / 303, three closures at 3000 keyframes 231 / 229, 1 % long-range
    // points 247 / 320: the block iteration pays where the coupling outside the band is a few hundred blocks -- outlying eigenvalues, which it captures 32 at a
    // time -- and loses where it is spread over the map)
    const bool want_block = c->dbg.pcg_block == 2 || (c->dbg.pcg_block == 0 && D.n_far <= 4096);
    if (ms_available(c) && want_block && ms_reserve(c, std::max(ECG_T, c->ms_cap)) == TSBA_OK) {
        const int nch = (c->n_kf + ECG_CH - 1)/ECG_CH; const size_t n6 = (size_t)W.N;
        const size_t need = (2*n6*ECG_T + (size_t)nch*2*(ECG_T*ECG_T + 1) + 4*(size_t)ECG_T*ECG_T + 4*ECG_T + 16)*sizeof(double);
        bool ok = true;
        if (need > c->ecg_bytes) { if (c->ecg_alloc) { hipStreamSynchronize(c->stream); hipFree(c->ecg_alloc); } c->ecg_alloc = nullptr; c->ecg_bytes = 0;
            ok = hipMalloc((void **)&c->ecg_alloc, need) == hipSuccess; if (ok) c->ecg_bytes = need; }
        if (ok) {
            EcgBuf &E = c->ecg; double *q = c->ecg_alloc;
            E.P = q; q += n6*ECG_T; E.Q = q; q += n6*ECG_T; E.part = q; q += (size_t)nch*2*(ECG_T*ECG_T + 1); E.Cm = q; q += ECG_T*ECG_T; E.Lm = q; q += ECG_T*ECG_T + ECG_T;
            E.Y = q; q += ECG_T*ECG_T; E.y1 = q; q += ECG_T; E.scal = q; E.nchunk = nch;
            const int Tk = c->ms.T; c->ms.T = ECG_T; const MsBuf M = c->ms;
            auto finished_e = [&](int it) {
                if (!c->hprog || it < 2) return false;
                const auto tw = std::chrono::steady_clock::now();
                for (int spin = 0;; spin++) {
                    const unsigned long long w = ((volatile unsigned long long *)c->hprog)[1];
                    if ((unsigned int)(w >> 32) == seq) { if (w & 1) return true; if ((int)((w & 0xffffffffu) >> 1) + 2 >= it) return false; }
                    if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - tw > std::chrono::seconds(5)) return false;
                }
            };
            hipLaunchKernelGGL(k_ecg_begin, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, M);
            launch_ms_solve(c);
            hipLaunchKernelGGL(k_ecg_gram, dim3(nch), dim3(256), 0, c->stream, W, (const double *)M.X, (const double *)M.X, (const double *)nullptr, (const double *)M.R, (const double *)M.X, E);
            hipLaunchKernelGGL(k_ecg_small, dim3(1), dim3(1024), 0, c->stream, W, E, 0, 0, seq, tol2);
            hipLaunchKernelGGL(k_ecg_update, dim3(nbp), dim3(256), 0, c->stream, W, M, E, 2, 1);
            int it = 0;
            for (; it < cap; it++) {
                if (finished_e(it)) break;
                hipLaunchKernelGGL(k_ecg_matvec, dim3(nbp), dim3(256), 0, c->stream, W, D, B, E);
                hipLaunchKernelGGL(k_ecg_gram, dim3(nch), dim3(256), 0, c->stream, W, (const double *)E.P, (const double *)E.Q, (const double *)M.R, (const double *)nullptr, (const double *)nullptr, E);
                hipLaunchKernelGGL(k_ecg_small, dim3(1), dim3(1024), 0, c->stream, W, E, 1, it, seq, tol2);
                hipLaunchKernelGGL(k_ecg_update, dim3(nbp), dim3(256), 0, c->stream, W, M, E, 1, 0);
                launch_ms_solve(c);
                hipLaunchKernelGGL(k_ecg_gram, dim3(nch), dim3(256), 0, c->stream, W, (const double *)E.Q, (const double *)M.X, (const double *)nullptr, (const double *)M.R, (const double *)M.X, E);
                hipLaunchKernelGGL(k_ecg_small, dim3(1), dim3(1024), 0, c->stream, W, E, 2, it, seq, tol2);
                hipLaunchKernelGGL(k_ecg_update, dim3(nbp), dim3(256), 0, c->stream, W, M, E, 2, 0);
            }
            hipLaunchKernelGGL(k_ecg_finish, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, it);
            c->ms.T = Tk;
            return;
        }
    }
    hipLaunchKernelGGL(k_pcg_begin, dim3(nbp), dim3(PCG_ET), 0, c->stream, W);
    auto finished = [&](int it) {                                  // true: the device reported convergence (or the end of the pass); else waits until it is within two iterations
        if (!c->hprog || it < 2) return false;
        const auto tw = std::chrono::steady_clock::now();
        for (int spin = 0;; spin++) {
            const unsigned long long w = ((volatile unsigned long long *)c->hprog)[1];
            if ((unsigned int)(w >> 32) == seq) { if (w & 1) return true; if ((int)((w & 0xffffffffu) >> 1) + 2 >= it) return false; }
            if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - tw > std::chrono::seconds(5)) return false;      // never hang on it
        }
    };
    // M^-1 on the residual: the solve phase of the partitioned band solver on the factor this trial's first solve left (tsba_bandms.h); where
    // that is not available (a single interior, the sequential separator solve) the factorisation is run again with the residual as right-hand side
    // (measured at 5000 keyframes, one column: 1.3 ms per application against 0.57 ms for the factorisation re-run -- the solve phase pays for 64
    // columns whether it has them or not; it is the default only for the block variants.  pcg_refactor = 2 selects it for the single-vector iteration)
    const bool ms = ms_available(c) && c->dbg.pcg_refactor == 2 && ms_reserve(c, std::max(1, c->ms_cap)) == TSBA_OK;
    const double *zp = W.Sy; double zs = -1.0;
    int it = 0;
    for (; it < cap; it++) {
        if (finished(it)) break;
        hipLaunchKernelGGL(k_pcg_matvec, dim3(nbp), dim3(PCG_T), 0, c->stream, W, D, it, seq, B, tol2, nbp, zp, zs);
        if (ms) { hipLaunchKernelGGL(k_pcg_update, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, it, nbp, c->ms.R, 1.0);
            const int Tk = c->ms.T; c->ms.T = 1; launch_ms_solve(c); c->ms.T = Tk; zp = c->ms.X; zs = 1.0; }
        else { hipLaunchKernelGGL(k_pcg_update, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, it, nbp, W.g, -1.0); launch_solve(c); }
        hipLaunchKernelGGL(k_pcg_dot, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, zp, zs);
    }
    hipLaunchKernelGGL(k_pcg_finish, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, it);
}

// one LM iteration: reduced system -> pose step -> back-substitution / candidate -> speculative linearisation at the
// candidate -> decision (on acceptance the speculative LinBuf simply becomes the current one)
static void launch_step(Ctx *c, const LevelDev &D) {
    struct XT { Ctx *c; size_t x0; ~XT() { c->x_trial = c->x_acc - x0; } } xt{c, c->x_acc};
    Work &W = c->W;
    int nb_pt = (c->n_pt + 255)/256, nb_tx = (c->n_text + 255)/256, nb_kf = (c->n_kf + 255)/256, nb_pr = (D.n_pair + 255)/256;
    int use_lds; int lds = solve_lds_bytes(c, &use_lds);
    // block-sparse S: what no block of the plan covers must read as zero.  The streaming / partitioned band solvers leave S intact and a pass
    // writes the same entries in every trial (the free poses are fixed at its start), so the band is cleared once per pass; the in-place
    // Cholesky of the wide-band path needs it before every assembly -- and so does a sharded run (a rank assembles only its own blocks; the
    // other entries hold the sums the last exchange unpacked)
    if ((int64_t)D.n_sb < (int64_t)c->n_kf*(c->n_kf + 1)/2 && (!c->band_stream || c->S_stale || is_multi(c))) {
        hipMemsetAsync(c->S_alloc, 0, sizeof(double)*c->S_count, c->stream); c->S_stale = false;
        if (D.far_B > 0) hipMemsetAsync(W.Sfar, 0, sizeof(double)*36*(size_t)std::max(D.n_far, 1), c->stream); }
    launch_schur(c, D, (int)is_multi(c));
    if (is_multi(c)) {                             // one exchange per LM trial: the reduced normal equations
        if (c->S_xchg) {                           // band storage: only the band's entries travel
            const size_t nx = ((size_t)W.N + (W.ring ? c->xchg_wp - 6 : 0))*c->xchg_wp;
            const int nbp = (int)std::min<size_t>(2048, (nx + 255)/256);
            hipLaunchKernelGGL(k_band_pack, dim3(nbp), dim3(256), 0, c->stream, W, c->S_xchg, c->xchg_wp, 0);
            allreduce(c, c->S_xchg, nx, ncclDouble, ncclSum);
            hipLaunchKernelGGL(k_band_pack, dim3(nbp), dim3(256), 0, c->stream, W, c->S_xchg, c->xchg_wp, 1);
        } else allreduce(c, c->S_alloc, c->S_count, ncclDouble, ncclSum);
        if (D.far_B > 0 && D.n_far > 0) allreduce(c, W.Sfar, 36*(size_t)D.n_far, ncclDouble, ncclSum);      // the blocks outside the band
        allreduce(c, W.g, W.N, ncclDouble, ncclSum);
        hipLaunchKernelGGL(k_damp_multi, dim3((c->n_kf + 255)/256), dim3(256), 0, c->stream, W);
    }
    launch_solve_full(c, D);
    hipLaunchKernelGGL(k_back, dim3(nb_pt + nb_tx + nb_kf), dim3(256), 0, c->stream, W, D, nb_pt, nb_tx);
    launch_linearize(c, D, 1);
    hipLaunchKernelGGL(k_decide, dim3(1), dim3(256), 0, c->stream, W, D, nb_pt + nb_tx + nb_kf, nb_pt + nb_tx + nb_pr, c->opt, (int)is_multi(c), pose_parts(c));
}

int tsba_solve(void *ctx, tsba_report *r) {
    Ctx *c = (Ctx *)ctx; if (!c || !r) return TSBA_ERR_ARG;
    if (!c->uploaded) { set_err(c, "no problem uploaded"); return TSBA_ERR_STATE; }
    hipSetDevice(c->device);
    memset(r, 0, sizeof(*r));
    const tsba_options &o = c->opt;
    int use_lds; int lds = solve_lds_bytes(c, &use_lds);
    { int rca = set_solver_attrs(c); if (rca) return rca; }
    struct Token { Ctx *c; Token(Ctx *c_) : c(c_) { if (c->lgroup) { c->in_solve = true; c->lgroup->gpu_token.lock(); c->has_token = true; } }
                   ~Token() { if (c->lgroup) { c->in_solve = false; if (c->has_token) { hipStreamSynchronize(c->stream); c->has_token = false; c->lgroup->gpu_token.unlock(); } } } } token(c);
    auto t0 = std::chrono::steady_clock::now();
    int rc = reset_state(c); if (rc) return rc;
    if (c->far_B > 0) hipMemsetAsync(c->W.pc_stat, 0, 4*sizeof(int), c->stream);
    for (int ps = 0; ps < o.n_passes; ps++) {
        if (!c->lev_built[o.levels[ps]]) {            // a level the upload left for now (one-shot call on a small window): its plan is ready or nearly so
            if (!c->stage_p) { set_err(c, "level not staged"); return TSBA_ERR_STATE; }
            c->stage_async = true; rc = stage_level(c, c->stage_p, o.levels[ps], nullptr, nullptr); c->stage_async = false; if (rc) return rc; }
        if (c->lev_wait[o.levels[ps]]) { hipStreamWaitEvent(c->stream, c->ev_stage[o.levels[ps]], 0); c->lev_wait[o.levels[ps]] = 0; }
        const LevelDev &D = c->lev[o.levels[ps]];
        const bool pose_path = c->pose_only && !is_multi(c);
        if (pose_path) {                                     // k_pass_reset + k_participation + k_gauge + k_musigma in one launch
            c->W.hprog = c->hprog; c->W.pass_seq = ++c->pass_seq;
            hipLaunchKernelGGL(k_pose_begin, dim3(D.n_tg + 1), dim3(MS_THREADS), 0, c->stream, c->W, D, o.initial_radius, o.its[ps], (const uint8_t *)c->kf_initial);
        } else launch_pass_init(c, D, ps);
        // The kernels of an LM iteration return at once when the pass has converged, but each still costs a launch (~4 us):
        // the host reads the pinned progress word and stays at most two iterations ahead of the device -- no API call, no
        // synchronisation -- so a pass that converges early wastes two iterations of empty launches instead of all the rest.
        auto converged = [&](int it) {
            if (!c->hprog || is_multi(c)) return false;
            const auto tw = std::chrono::steady_clock::now();
            for (int spin = 0;; spin++) {
                const unsigned long long w = *(volatile unsigned long long *)c->hprog;
                if ((unsigned int)(w >> 32) == c->W.pass_seq) { if (w & 1) return true; if ((int)((w & 0xffffffffu) >> 1) + 2 > it) break; }
                else if (it < 2) break;                              // the device has not reached this pass yet
                if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - tw > std::chrono::seconds(2)) break;   // never hang on it
            }
            return false;
        };
        if (pose_path) {                                     // PoseOptim: one launch per LM iteration (tsba_pose.h)
            const int G = pose_grid(D);
            hipLaunchKernelGGL(k_pose_iter, dim3(G), dim3(POSE_WG), 0, c->stream, c->W, D, o, -1, G);
            int k_last = 0;
            for (int k = 0; k <= o.its[ps]; k++) {           // launch k decides trial k - 1 and prepares trial k
                if (k >= 1 && converged(k - 1)) break;
                hipLaunchKernelGGL(k_pose_iter, dim3(G), dim3(POSE_WG), 0, c->stream, c->W, D, o, k, G);
                k_last = k;
            }
            // outlier pass + installation of the pass's result (one extra workgroup) in one launch
            hipLaunchKernelGGL(k_outlier, dim3((D.n_sc + 63)/64 + D.n_tg + 1), dim3(64), 0, c->stream, c->W, D, o.chi2_mono[ps], o.chi2_text[ps],
                               o.text_bad_ratio, o.outlier_scene, o.outlier_text, (const PoseState *)(c->W.pst + ((k_last + 1) & 1)));
            CK(hipMemcpyAsync(c->st_log + ps, c->W.st, sizeof(LmState), hipMemcpyDeviceToDevice, c->stream));
            continue;
        }
        launch_linearize(c, D, 0);
        for (int it = 0; it < o.its[ps]; it++) {
            if (converged(it)) break;
            launch_step(c, D);
            if (it >= 1) { rc = stage_ahead(c, ps); if (rc) return rc; }      // (with two iterations queued the device does not run dry while the host stages)
        }
        if (o.outlier_scene || o.outlier_text)
            if (D.n_sc + D.n_tg > 0) hipLaunchKernelGGL(k_outlier, dim3((D.n_sc + 63)/64 + D.n_tg), dim3(64), 0, c->stream, c->W, D,
                                                          o.chi2_mono[ps], o.chi2_text[ps], o.text_bad_ratio, o.outlier_scene, o.outlier_text, (const PoseState *)nullptr);
        if (c->cov_text >= 0 && c->cov_text < c->n_text && ps < TSBA_MAX_LEVELS) hipLaunchKernelGGL(k_record_vtx, dim3(1), dim3(64), 0, c->stream, c->W, c->cov_text, c->cov_log + 6*ps);
        CK(hipMemcpyAsync(c->st_log + ps, c->W.st, sizeof(LmState), hipMemcpyDeviceToDevice, c->stream));
    }
    if (c->world > 1) {                           // every landmark was optimised by its owner only
        int nl = c->n_pt + 3*c->n_text;
        if (nl > 0) {
            hipLaunchKernelGGL(k_delta_multi, dim3((nl + 255)/256), dim3(256), 0, c->stream, c->W, (const double *)c->rho0, (const double *)c->theta0, 0);
            if (c->n_pt) allreduce(c, c->W.dl_pt, c->n_pt, ncclDouble, ncclSum);
            if (c->n_text) allreduce(c, c->W.dl_tx, 3*(size_t)c->n_text, ncclDouble, ncclSum);
            hipLaunchKernelGGL(k_delta_multi, dim3((nl + 255)/256), dim3(256), 0, c->stream, c->W, (const double *)c->rho0, (const double *)c->theta0, 1);
        }
    }
    CK(hipMemcpyAsync(c->st_host, c->st_log, sizeof(LmState)*o.n_passes, hipMemcpyDeviceToHost, c->stream));
    CK(hipStreamSynchronize(c->stream));
    CK(hipGetLastError());
    if (!c->err.empty() && c->err.rfind("ncclAllReduce", 0) == 0) return TSBA_ERR_COMM;
    auto t1 = std::chrono::steady_clock::now();
    r->t_solve_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    r->n_passes = o.n_passes;
    long long prev_lin = 0, prev_cost = 0;
    for (int ps = 0; ps < o.n_passes; ps++) {
        const LmState &s = c->st_host[ps];
        r->iters[ps] = s.it; r->accepted[ps] = s.accepted; r->termination[ps] = s.term;
        r->cost0[ps] = s.cost0; r->cost1[ps] = s.x_cost;
        r->n_sblock[ps] = s.ns_active; r->n_tblock[ps] = s.nt_active;
        r->n_bad_scene[ps] = s.n_bad_scene; r->n_bad_tfeat[ps] = s.n_bad_tfeat; r->n_bad_text[ps] = s.n_bad_text;
        long long evals = (s.n_lin - prev_lin) + (s.n_cost - prev_cost);
        r->n_resid_evals += evals*(2LL*s.ns_active + 8LL*s.nt_active);
        prev_lin = s.n_lin; prev_cost = s.n_cost;
        if (s.term == 5) r->status = TSBA_ERR_NUMERIC;
    }
    return TSBA_OK;
}

int tsba_download(void *ctx, tsba_problem *p) {
    Ctx *c = (Ctx *)ctx; if (!c || !p) return TSBA_ERR_ARG;
    if (!c->uploaded) { set_err(c, "no problem uploaded"); return TSBA_ERR_STATE; }
    hipSetDevice(c->device);
    LmState st; CK(hipMemcpy(&st, c->W.st, sizeof(st), hipMemcpyDeviceToHost));
    int cur = st.cur & 1;
    CK(hipMemcpy(p->pose, c->W.pose[cur], sizeof(double)*7*c->n_kf, hipMemcpyDeviceToHost));
    if (c->n_pt) CK(hipMemcpy(p->rho, c->W.rho[cur], sizeof(double)*c->n_pt, hipMemcpyDeviceToHost));
    if (c->n_text) CK(hipMemcpy(p->theta, c->W.theta[cur], sizeof(double)*3*c->n_text, hipMemcpyDeviceToHost));
    if (c->n_sgood) CK(hipMemcpy(p->sgood, c->W.sgood, c->n_sgood, hipMemcpyDeviceToHost));
    if (c->n_tobs) CK(hipMemcpy(p->tobs_good, c->W.tobs_good, c->n_tobs, hipMemcpyDeviceToHost));
    if (c->n_tfgood) CK(hipMemcpy(p->tfgood, c->W.tfgood, c->n_tfgood, hipMemcpyDeviceToHost));
    return TSBA_OK;
}

static int one_shot(void *ctx, tsba_problem *p, const tsba_options *o, tsba_report *r) {
    auto t0 = std::chrono::steady_clock::now();
    int rc = upload_impl(ctx, p, o, true); if (rc) return rc;
    auto t1 = std::chrono::steady_clock::now();
    rc = tsba_solve(ctx, r);
    { Ctx *c = (Ctx *)ctx; join_planners(c); c->stage_p = nullptr;      // (*p is the caller's: nothing may be staged from it after this call)
      for (int l = 0; l < (int)c->lev_planned.size(); l++) if (c->lev_planned[l] && !c->lev_built[l]) c->uploaded = false; }   // a failed solve left a level unstaged: upload again before anything else
    if (rc) return rc;
    auto t2 = std::chrono::steady_clock::now();
    rc = tsba_download(ctx, p); if (rc) return rc;
    auto t3 = std::chrono::steady_clock::now();
    r->t_upload_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    r->t_download_ms = std::chrono::duration<double, std::milli>(t3 - t2).count();
    return r->status;
}
int tsba_local_ba(void *ctx, tsba_problem *p, const tsba_options *o, tsba_report *r) { return one_shot(ctx, p, o, r); }
int tsba_pose_optim(void *ctx, tsba_problem *p, const tsba_options *o, tsba_report *r) {
    if (p && p->n_kf != 1) { if (ctx) set_err((Ctx *)ctx, "tsba_pose_optim needs n_kf == 1"); return TSBA_ERR_ARG; }
    return one_shot(ctx, p, o, r);
}
int tsba_global_ba(void *ctx, tsba_problem *p, const tsba_options *o, tsba_report *r) { return one_shot(ctx, p, o, r); }
int tsba_theta_optim(void *ctx, tsba_problem *p, const tsba_options *o, int text, double cov[9], tsba_report *r) {
    Ctx *c = (Ctx *)ctx;
    if (!c || !p || !o || !cov || !r || text < 0 || text >= p->n_text) return TSBA_ERR_ARG;
    c->cov_text = text;
    int rc = one_shot(ctx, p, o, r);
    c->cov_text = -1;
    if (rc) return rc;
    // Information matrix of theta[text] = V of the linearisation at the end of a pass (undamped, loss-corrected J^T J).  The reference
    // runs ceres::Covariance after EVERY pyramid pass and keeps the last one that succeeds (optimizer.cc:2219-2238: thetaVariance is only
    // overwritten when Compute returns true): the passes are tried from the last to the first.  A pass fails when V is not positive
    // definite or its reciprocal condition number is below 1e-14 (Ceres' min_reciprocal_condition_number; recalled, oracle/RECALLED.md).
    // No pass succeeds: as the reference (PyrThetaOptim still returns true, optimizer.cc:2224-2241) not an error, cov[] untouched.
    double Vall[6*TSBA_MAX_LEVELS];
    CK(hipMemcpy(Vall, c->cov_log, sizeof(Vall), hipMemcpyDeviceToHost));
    r->cov_valid = 0;
    for (int ps = std::min(o->n_passes, TSBA_MAX_LEVELS) - 1; ps >= 0; ps--) {
        const double *V = Vall + 6*ps;
        const double a = V[0], b = V[1], cc = V[2], e = V[3], f = V[4], i = V[5];
        const double A = e*i - f*f, B = -(b*i - cc*f), C = b*f - cc*e, det = a*A + b*B + cc*C;
        if (!(det > 0.0) || !(a > 0.0) || !(a*e - b*b > 0.0)) continue;
        // eigenvalues of the symmetric 3x3 (trigonometric form): reciprocal condition number
        const double q = (a + e + i)/3.0, p1 = b*b + cc*cc + f*f, p2 = (a - q)*(a - q) + (e - q)*(e - q) + (i - q)*(i - q) + 2.0*p1, pp = sqrt(p2/6.0);
        double lmin = q, lmax = q;
        if (pp > 0.0) { const double ip = 1.0/pp, b00 = (a - q)*ip, b01 = b*ip, b02 = cc*ip, b11 = (e - q)*ip, b12 = f*ip, b22 = (i - q)*ip;
            double hr = 0.5*(b00*(b11*b22 - b12*b12) - b01*(b01*b22 - b12*b02) + b02*(b01*b12 - b11*b02));
            hr = hr < -1.0 ? -1.0 : (hr > 1.0 ? 1.0 : hr);
            const double phi = acos(hr)/3.0; lmax = q + 2.0*pp*cos(phi); lmin = q + 2.0*pp*cos(phi + 2.0943951023931953); }
        if (!(lmin > 1e-14*lmax)) continue;
        const double id = 1.0/det;
        cov[0] = A*id; cov[1] = B*id; cov[2] = C*id; cov[3] = B*id; cov[4] = (a*i - cc*cc)*id; cov[5] = -(a*f - b*cc)*id;
        cov[6] = C*id; cov[7] = cov[5]; cov[8] = (a*e - b*b)*id;
        r->cov_valid = 1;
        break;
    }
    return TSBA_OK;
}

int tsba_eval(void *ctx, const tsba_problem *p, const tsba_options *o, int level,
              double *resid, double *jac, double *musigma, int64_t *ns_out, int64_t *nt_out) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    if (!p || level < 0 || level >= p->n_levels) return TSBA_ERR_ARG;
    tsba_options oo = *o; oo.n_passes = 1; oo.levels[0] = level;
    int rc = tsba_upload(ctx, p, &oo); if (rc) return rc;
    rc = reset_state(c); if (rc) return rc;
    const LevelDev &D = c->lev[level]; const HostPlan &H = c->hplan[level];
    launch_pass_init(c, D, 0);
    // reference block order: scene candidates by observation index, then text blocks by (tobs, feature)
    std::vector<int> out_idx(H.n_sc(), -1);
    { std::vector<int> by_obs(p->n_sobs[level], -1);
      for (int q = 0; q < H.n_sc(); q++) by_obs[H.sc_obs[q]] = q;
      int n = 0;
      for (int s = 0; s < p->n_sobs[level]; s++) { int q = by_obs[s]; if (q < 0) continue;
          if (oo.filter_good && !p->sgood[H.sc_flag[q]]) continue; out_idx[q] = n++; }
      *ns_out = n; }
    std::vector<int> bg, bf;
    for (int g = 0; g < H.n_tg(); g++) {
        int tb = H.tg_tobs[g], j = H.tg_text[g];
        if (oo.filter_good && !p->tobs_good[tb]) continue;
        for (int f = p->tfeat_off[level][j]; f < p->tfeat_off[level][j+1]; f++) {
            if (oo.filter_good && !p->tfgood[p->tobs_fgood_off[tb] + p->tfeat_raw[level][f]]) continue;
            bg.push_back(g); bf.push_back(f);
        }
    }
    *nt_out = (int64_t)bg.size();
    int64_t ns = *ns_out, nt = *nt_out;
    if (resid || jac) {
        const int *d_oi, *d_bg, *d_bf; double *d_r, *d_j = nullptr;
        rc = dev_upload_vec(c, &d_oi, out_idx); if (rc) return rc;
        rc = dev_upload_vec(c, &d_bg, bg); if (rc) return rc;
        rc = dev_upload_vec(c, &d_bf, bf); if (rc) return rc;
        rc = dev_alloc(c, &d_r, (size_t)(2*ns + 8*nt)); if (rc) return rc;
        if (jac) { rc = dev_alloc(c, &d_j, (size_t)(26*ns + 120*nt)); if (rc) return rc; }
        flush_run(c);                                  // staged uploads leave as one copy
        if (H.n_sc() > 0) hipLaunchKernelGGL(k_eval_scene, dim3((H.n_sc() + 255)/256), dim3(256), 0, c->stream, c->W, D, d_oi, d_r, d_j);
        if (nt > 0) hipLaunchKernelGGL(k_eval_text, dim3(((int)nt + 255)/256), dim3(256), 0, c->stream, c->W, D, (int)nt, d_bg, d_bf, (int)ns, d_r, d_j);
        CK(hipStreamSynchronize(c->stream)); CK(hipGetLastError());
        if (resid) CK(hipMemcpy(resid, d_r, sizeof(double)*(size_t)(2*ns + 8*nt), hipMemcpyDeviceToHost));
        if (jac) CK(hipMemcpy(jac, d_j, sizeof(double)*(size_t)(26*ns + 120*nt), hipMemcpyDeviceToHost));
    }
    CK(hipStreamSynchronize(c->stream));
    if (musigma && p->n_tobs) CK(hipMemcpy(musigma, c->W.musig, sizeof(double)*2*p->n_tobs, hipMemcpyDeviceToHost));
    return TSBA_OK;
}

// debug / test aid: first linearisation of pass 0 + reduced system for `radius`; copies S (N x N), g (N), cost, kf flags
int tsba_debug_reduced_system(void *ctx, double radius, double *S, double *g, double *cost, int32_t *kf_free, double *dp) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    if (!c->uploaded) return TSBA_ERR_STATE;
    hipSetDevice(c->device);
    int rc = reset_state(c); if (rc) return rc;
    tsba_options saved = c->opt; c->opt.initial_radius = radius;
    const LevelDev &D = c->lev[c->opt.levels[0]];
    { int rca = set_solver_attrs(c); if (rca) return rca; }
    launch_pass_init(c, D, 0);
    launch_linearize(c, D, 0);
    Work &W = c->W;
    if ((int64_t)D.n_sb < (int64_t)c->n_kf*(c->n_kf + 1)/2) { hipMemsetAsync(c->S_alloc, 0, sizeof(double)*c->S_count, c->stream); c->S_stale = false;
        if (D.far_B > 0) hipMemsetAsync(W.Sfar, 0, sizeof(double)*36*(size_t)std::max(D.n_far, 1), c->stream); }
    // split (multi-GPU) sequence: this shard's PARTIAL S and g, before any exchange and without the pose damping (which is added
    // once after the all-reduce) -- the parts of all shards sum to the unsharded system; dp is not computed
    launch_schur(c, D, (int)is_multi(c));
    if (!is_multi(c)) launch_solve_full(c, D); else hipMemsetAsync(W.dp, 0, sizeof(double)*W.N, c->stream);
    c->opt = saved;
    CK(hipStreamSynchronize(c->stream)); CK(hipGetLastError());
    if (S) {
        if (!W.band) CK(hipMemcpy(S, W.S, sizeof(double)*(size_t)W.N*W.N, hipMemcpyDeviceToHost));
        else { std::vector<double> hb(c->S_count); CK(hipMemcpy(hb.data(), c->S_alloc, sizeof(double)*c->S_count, hipMemcpyDeviceToHost));
            const long long N = W.N, LDB = W.ldS + 1, Wb = LDB - c->S_up;                // band -> dense (entries outside the band are zero)
            for (long long i = 0; i < N; i++) for (long long j = 0; j < N; j++)
                S[i*N + j] = (j >= i - Wb && j <= i + c->S_up - 1) ? hb[(size_t)(Wb + i*(LDB - 1) + j)] : 0.0;
            if (W.ring) {                     // the loop-closure blocks: ghost row 6 nfree + r stands for row r of the first poses (lower triangle: (late pose, early pose))
                int nfr[2] = {0, 0}; CK(hipMemcpy(nfr, W.nfree, 2*sizeof(int), hipMemcpyDeviceToHost));
                const long long n6 = 6LL*nfr[0], r06 = 6LL*nfr[1], ng = std::min<long long>(6LL*W.ring_b, N);
                for (long long r = 0; r < ng && n6 + r < (long long)(c->S_count/LDB); r++) for (long long j = std::max(0LL, n6 + r - Wb); j < n6; j++) {
                    const double v = hb[(size_t)(Wb + (n6 + r)*(LDB - 1) + j)]; if (v != 0.0 && j > r06 + r) S[j*N + r06 + r] = v; }
            }
            if (D.far_B > 0 && D.n_far > 0) {        // the blocks outside the band (lower triangle: rows of the later keyframe)
                const HostPlan &H = c->hplan[D.level];
                std::vector<double> hf(36*(size_t)D.n_far); std::vector<int> fi(c->n_kf);
                CK(hipMemcpy(hf.data(), W.Sfar, sizeof(double)*hf.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(fi.data(), W.fidx, sizeof(int)*c->n_kf, hipMemcpyDeviceToHost));
                for (int q = 0; q < D.n_far; q++) { const long long ia = fi[H.far_a[q]], ic = fi[H.far_b[q]]; if (ia < 0 || ic < 0) continue;
                    for (int r = 0; r < 6; r++) for (int cc = 0; cc < 6; cc++) S[(6*ic + cc)*N + 6*ia + r] = hf[36*(size_t)q + 6*r + cc]; }
            } }
    }
    if (g) CK(hipMemcpy(g, W.g, sizeof(double)*W.N, hipMemcpyDeviceToHost));
    if (dp) CK(hipMemcpy(dp, W.dp, sizeof(double)*W.N, hipMemcpyDeviceToHost));
    LmState st; CK(hipMemcpy(&st, W.st, sizeof(st), hipMemcpyDeviceToHost));
    if (cost) *cost = st.x_cost;
    if (kf_free) { std::vector<int> in(c->n_kf), cs(c->n_kf);
        CK(hipMemcpy(in.data(), W.kf_in, sizeof(int)*c->n_kf, hipMemcpyDeviceToHost)); CK(hipMemcpy(cs.data(), W.kf_const, sizeof(int)*c->n_kf, hipMemcpyDeviceToHost));
        for (int k = 0; k < c->n_kf; k++) kf_free[k] = in[k] && !cs[k]; }
    return TSBA_OK;
}

// The same for LARGE maps, where the dense (6 n_kf)^2 copy is not an option (7.2 GB at 5000 keyframes): the band of the
// compressed (free-pose) system in LAPACK lower-band storage, ab[(i - j)*n + j] = S(i, j) for j <= i <= j + bw -- what
// scipy.linalg.solveh_banded(lower=True) takes.  Call once with ab = NULL to get n (rows) and bw, then with buffers.
int tsba_debug_reduced_band(void *ctx, double radius, int32_t *n_out, int32_t *bw_out, double *ab, double *g, double *dp) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    if (!c->uploaded) return TSBA_ERR_STATE;
    if (!c->W.band) { set_err(c, "the uploaded problem keeps a dense reduced system: use tsba_debug_reduced_system"); return TSBA_ERR_STATE; }
    if (c->W.ring) { set_err(c, "ring-shaped map: the loop-closure blocks live in ghost rows outside the band (tsba_debug_set no_ring for the reordered band)"); return TSBA_ERR_STATE; }
    int rc = tsba_debug_reduced_system(ctx, radius, nullptr, nullptr, nullptr, nullptr, nullptr); if (rc) return rc;
    Work &W = c->W;
    int nfree = 0; CK(hipMemcpy(&nfree, W.nfree, sizeof(int), hipMemcpyDeviceToHost));
    const long long n = 6LL*nfree, LDB = W.ldS + 1, Wb = LDB - c->S_up;
    const int bw = std::max(6, c->lev[c->opt.levels[0]].bw_rows) + 5;              // rows below the diagonal that can be non-zero (block-aligned band)
    if (n_out) *n_out = (int32_t)n; if (bw_out) *bw_out = bw;
    if (ab) {
        std::vector<double> hb(c->S_count); CK(hipMemcpy(hb.data(), c->S_alloc, sizeof(double)*c->S_count, hipMemcpyDeviceToHost));
        for (long long d = 0; d <= bw; d++) for (long long j = 0; j < n; j++) { const long long i = j + d;
            ab[d*n + j] = (i < n && j >= i - Wb) ? hb[(size_t)(Wb + i*(LDB - 1) + j)] : 0.0; }
    }
    if (g) CK(hipMemcpy(g, W.g, sizeof(double)*n, hipMemcpyDeviceToHost));
    if (dp) CK(hipMemcpy(dp, W.dp, sizeof(double)*W.N, hipMemcpyDeviceToHost));
    return TSBA_OK;
}

int tsba_time_linearize(void *ctx, int level, int n, double *avg_ms, double *algo_bytes) {
    Ctx *c = (Ctx *)ctx; if (!c || n <= 0) return TSBA_ERR_ARG;
    if (!c->uploaded || level < 0 || level >= c->n_levels || !c->lev_built[level]) { set_err(c, "level not uploaded"); return TSBA_ERR_STATE; }
    hipSetDevice(c->device);
    int rc = reset_state(c); if (rc) return rc;
    const LevelDev &D = c->lev[level];
    int ps = 0; for (int k = 0; k < c->opt.n_passes; k++) if (c->opt.levels[k] == level) ps = k;
    launch_pass_init(c, D, ps);
    launch_linearize(c, D, 0);                                 // warm-up (also leaves need_lin = 0)
    CK(hipStreamSynchronize(c->stream));
    LmState st; CK(hipMemcpy(&st, c->W.st, sizeof(st), hipMemcpyDeviceToHost));
    st.need_lin = 1; st.done = 0; st.first = 0;
    CK(hipMemcpy(c->W.st, &st, sizeof(st), hipMemcpyHostToDevice));   // k_linearize never clears need_lin itself
    CK(hipEventRecord(c->ev0, c->stream));
    for (int k = 0; k < n; k++) {
        if (lin_small_pairs(c, D) && D.n_tg == 0) hipLaunchKernelGGL((k_linearize<MODE_FULL, 4, false>), dim3((((D.n_pair + 4*LIN_NWV - 1)/(4*LIN_NWV) + 7)/8)*8), dim3(LIN_T), 0, c->stream, c->W, D, 0);
        else if (lin_small_pairs(c, D)) hipLaunchKernelGGL((k_linearize<MODE_FULL, 4>), dim3((((D.n_pair + 4*LIN_NWV - 1)/(4*LIN_NWV) + D.n_tg + 7)/8)*8), dim3(LIN_T), 0, c->stream, c->W, D, 0);
        else hipLaunchKernelGGL((k_linearize<MODE_FULL, 1>), dim3((((D.n_pair + LIN_NWV - 1)/LIN_NWV + D.n_tg + 7)/8)*8), dim3(LIN_T), 0, c->stream, c->W, D, 0);
    }
    CK(hipEventRecord(c->ev1, c->stream));
    CK(hipEventSynchronize(c->ev1));
    float ms = 0; CK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (avg_ms) *avg_ms = (double)ms/n;
    if (algo_bytes) {   // 44 B / scene block, 128 B / text block, 16 B / (KF,text) pair, parameters once
        int npairs_text = 0; std::vector<int> kin;
        (void)kin;
        for (int g = 0; g < D.n_tg; g++) npairs_text++;
        *algo_bytes = 44.0*st.ns_active + 128.0*st.nt_active + 16.0*npairs_text + 56.0*c->n_kf + 8.0*c->n_pt + 24.0*c->n_text;
    }
    return TSBA_OK;
}

int tsba_text_label_image(void *ctx, int kf, int level, float *out) {
    Ctx *c = (Ctx *)ctx; if (!c || !out) return TSBA_ERR_ARG;
    if (!c->uploaded) { set_err(c, "no problem uploaded"); return TSBA_ERR_STATE; }
    if (kf < 0 || kf >= c->n_kf || level < 0 || level >= c->n_levels || !c->lev_built[level]) { set_err(c, "keyframe / level out of range or level not uploaded"); return TSBA_ERR_ARG; }
    hipSetDevice(c->device);
    const LevelDev &D = c->lev[level];
    if (D.img_w <= 0 || D.img_h <= 0 || (size_t)D.img_w*D.img_h > (size_t)MS_MASK_WORDS*32) { set_err(c, "no image geometry for this level"); return TSBA_ERR_ARG; }
    const size_t npx = (size_t)D.img_w*D.img_h;
    if (c->lbl_cap < npx) { if (c->lbl_dev) hipFree(c->lbl_dev); if (c->lbl_host) hipHostFree(c->lbl_host); c->lbl_cap = 0;
        CK(hipMalloc((void **)&c->lbl_dev, npx*sizeof(float))); CK(hipHostMalloc((void **)&c->lbl_host, npx*sizeof(float), hipHostMallocDefault)); c->lbl_cap = npx; }
    hipLaunchKernelGGL(k_label, dim3(1), dim3(LBL_THREADS), 0, c->stream, c->W, kf, D.img_w, D.img_h, D.K[0], D.K[1], D.K[2], D.K[3], c->lbl_dev);
    CK(hipMemcpyAsync(c->lbl_host, c->lbl_dev, npx*sizeof(float), hipMemcpyDeviceToHost, c->stream));
    CK(hipStreamSynchronize(c->stream)); CK(hipGetLastError());
    memcpy(out, c->lbl_host, npx*sizeof(float));
    return TSBA_OK;
}

// which kernels the uploaded problem runs through (so that a test can assert that it exercises the path it means to):
// out[0] reduced system in LDS (k_solve_t / k_solve_col)   [1] band storage   [2] streaming band solver   [3] interiors P
// [4] separator system by cyclic reduction   [5] band rows   [6] four pairs per wave in the linearisation of the first pass's level
// [7] fused pose-only kernel   [8] one-wave Schur blocks + k_pose_sums (large maps)   [9] world size   [10] rank
// [11..14] size of this rank's plan of the first pass's level: (target, host) pairs, S blocks, scene candidates, point slots
// [15] the rows of S follow a reverse Cuthill-McKee order of the keyframes instead of the keyframe index
int tsba_debug_solver_info(void *ctx, int32_t *out, int n) {
    Ctx *c = (Ctx *)ctx; if (!c || !out || n < 16) return TSBA_ERR_ARG;
    if (!c->uploaded) return TSBA_ERR_STATE;
    int use_lds; solve_lds_bytes(c, &use_lds);
    int bwmax = 0; for (int l = 0; l < c->n_levels; l++) if (c->lev_built[l]) bwmax = std::max(bwmax, c->lev[l].bw_rows);
    out[0] = use_lds; out[1] = c->W.band; out[2] = c->band_stream; out[3] = c->band_stream ? c->band_parts : 0; out[4] = c->sep_cr ? 1 : 0; out[5] = bwmax;
    out[6] = lin_small_pairs(c, c->lev[c->opt.levels[0]]) ? 1 : 0; out[7] = c->pose_only ? 1 : 0; out[8] = c->n_kf > 126 ? 1 : 0;
    out[9] = c->world; out[10] = c->rank;
    { const LevelDev &D0 = c->lev[c->opt.levels[0]]; out[11] = D0.n_pair; out[12] = D0.n_sb; out[13] = D0.n_sc; out[14] = D0.n_pslot; out[15] = D0.kf_order ? 1 : 0; }
    if (n >= 17) out[16] = c->W.ring;
    if (n >= 19) { out[17] = c->far_B; out[18] = c->n_far; }
    return TSBA_OK;
}
// Iterative reduced-system solves of the last tsba_solve on a map with long-range coupling (tsba_pcg.h): out[0] conjugate-gradient iterations in
// total, [1] reduced systems solved (LM trials), [2] most iterations of one system, [3] systems that hit the iteration cap.  Zeros otherwise.
int tsba_debug_pcg_stats(void *ctx, int32_t out[4]) {
    Ctx *c = (Ctx *)ctx; if (!c || !out) return TSBA_ERR_ARG;
    if (!c->uploaded) return TSBA_ERR_STATE;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (c->far_B <= 0) return TSBA_OK;
    hipSetDevice(c->device); CK(hipStreamSynchronize(c->stream));
    CK(hipMemcpy(out, c->W.pc_stat, 4*sizeof(int32_t), hipMemcpyDeviceToHost));
    return TSBA_OK;
}
// The 6x6 blocks outside the band after tsba_debug_reduced_system / a solve: keyframes a < b of block q and its 36 values (row-major, rows = a);
// the number of blocks is solver_info [18].  Any output may be NULL.
int tsba_debug_far_blocks(void *ctx, int32_t *a, int32_t *b, double *blocks) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    if (!c->uploaded || c->far_B <= 0) return TSBA_ERR_STATE;
    const LevelDev &D = c->lev[c->opt.levels[0]]; const HostPlan &H = c->hplan[D.level];
    hipSetDevice(c->device); CK(hipStreamSynchronize(c->stream));
    if (a) memcpy(a, H.far_a.data(), sizeof(int32_t)*H.far_a.size());
    if (b) memcpy(b, H.far_b.data(), sizeof(int32_t)*H.far_b.size());
    if (blocks && D.n_far > 0) CK(hipMemcpy(blocks, c->W.Sfar, sizeof(double)*36*(size_t)D.n_far, hipMemcpyDeviceToHost));
    return TSBA_OK;
}
// plane cache of the context (tsba_problem.kf_id): keyframes found on the device / copied, over the context's lifetime
int tsba_debug_img_cache_stats(void *ctx, int64_t out[2]) {
    Ctx *c = (Ctx *)ctx; if (!c || !out) return TSBA_ERR_ARG;
    out[0] = c->ic.hits; out[1] = c->ic.misses; return TSBA_OK;
}
// Test hook of the multi-right-hand-side solve phase (tsba_bandms.h): M X = R with the band factor the last tsba_debug_reduced_system / solve left
// behind.  R, X: [6 nfree][T] row-major (compressed free-pose rows).  TSBA_ERR_STATE unless the problem runs through the partitioned band
// solver with the cyclic-reduction separator system on a chain.
int tsba_debug_multi_solve(void *ctx, int T, const double *R, double *X) {
    Ctx *c = (Ctx *)ctx; if (!c || T < 1 || !R || !X) return TSBA_ERR_ARG;
    if (!c->uploaded || !ms_available(c)) { if (c) set_err(c, "multi-right-hand-side solve: needs the partitioned band solver with cyclic reduction on a chain"); return TSBA_ERR_STATE; }
    hipSetDevice(c->device);
    int nfree = 0; CK(hipMemcpy(&nfree, c->W.nfree, sizeof(int), hipMemcpyDeviceToHost));
    int rc = ms_reserve(c, T); if (rc) return rc;
    { int rca = set_solver_attrs(c); if (rca) return rca; }
    LmState st; CK(hipMemcpy(&st, c->W.st, sizeof(st), hipMemcpyDeviceToHost));
    st.done = 0; st.step_fail = 0; st.lin_done = 0; CK(hipMemcpy(c->W.st, &st, sizeof(st), hipMemcpyHostToDevice));
    CK(hipMemcpy(c->ms.R, R, sizeof(double)*6*(size_t)nfree*T, hipMemcpyHostToDevice));
    launch_ms_solve(c);
    CK(hipStreamSynchronize(c->stream)); CK(hipGetLastError());
    CK(hipMemcpy(X, c->ms.X, sizeof(double)*6*(size_t)nfree*T, hipMemcpyDeviceToHost));
    return TSBA_OK;
}
// row block of every keyframe in the compressed reduced system of the last pass set-up (-1: constant / not participating); with a
// plan order (reverse Cuthill-McKee, tsba_plan.h) this is not monotone in the keyframe index
int tsba_debug_row_of_kf(void *ctx, int32_t *rowblk) {
    Ctx *c = (Ctx *)ctx; if (!c || !rowblk) return TSBA_ERR_ARG;
    if (!c->uploaded) return TSBA_ERR_STATE;
    hipSetDevice(c->device); CK(hipStreamSynchronize(c->stream));
    CK(hipMemcpy(rowblk, c->W.fidx, sizeof(int32_t)*c->n_kf, hipMemcpyDeviceToHost));
    return TSBA_OK;
}
// host only (no device needed): the plan's band bound of one level, with or without the keyframe reordering; order_out [n_kf] gets the
// row order (identity when the plan keeps the keyframe order)
int tsba_debug_plan_band(const tsba_problem *p, const tsba_options *o, int level, int reorder, int32_t *bw_pose, int32_t *order_out) {
    if (!p || !o || !bw_pose || level < 0 || level >= p->n_levels) return TSBA_ERR_ARG;
    HostPlan H; build_plan(p, o, level, H, false, reorder != 0);
    *bw_pose = H.bw_pose;
    if (order_out) for (int k = 0; k < p->n_kf; k++) order_out[k] = H.kf_order.empty() ? k : H.kf_order[(size_t)k];
    return TSBA_OK;
}
int tsba_debug_plan_time(const tsba_problem *p, const tsba_options *o, int level, int reps, double *avg_ms) {   // host only: plan construction
    if (!p || !o || reps == 0) return TSBA_ERR_ARG;
    const bool laps = reps < 0; if (laps) reps = -reps;           // reps < 0: one recycled plan object (as a context does), lap times of the last build on stderr
    HostPlan R;
    if (laps) build_plan(p, o, level, R, false, true, CR_SMAX/6);
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; k++) { if (laps) build_plan(p, o, level, R, k == reps - 1, true, CR_SMAX/6); else { HostPlan H; build_plan(p, o, level, H); } }
    *avg_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()/reps;
    return TSBA_OK;
}
int tsba_debug_time_solve(void *ctx, int n, double *avg_ms) {     // n back-to-back launches of the dense solve on the last S, g
    Ctx *c = (Ctx *)ctx; if (!c || n <= 0 || !c->uploaded) return TSBA_ERR_ARG;
    hipSetDevice(c->device);
    CK(hipStreamSynchronize(c->stream));
    LmState st; CK(hipMemcpy(&st, c->W.st, sizeof(st), hipMemcpyDeviceToHost));
    st.done = 0; st.step_fail = 0;
    CK(hipMemcpy(c->W.st, &st, sizeof(st), hipMemcpyHostToDevice));
    launch_solve(c);
    CK(hipEventRecord(c->ev0, c->stream));
    for (int k = 0; k < n; k++) launch_solve(c);
    CK(hipEventRecord(c->ev1, c->stream));
    CK(hipEventSynchronize(c->ev1));
    float ms = 0; CK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *avg_ms = (double)ms/n;
    return TSBA_OK;
}

int tsba_debug_copy_S(void *ctx, double *out) {      // (6 n_kf + 1) x (6 n_kf): factored S and the rhs row after a solve
    Ctx *c = (Ctx *)ctx; if (!c || !c->uploaded) return TSBA_ERR_STATE;
    hipSetDevice(c->device); hipStreamSynchronize(c->stream);
    const Work &W = c->W; const long long N = W.N;
    if (!W.band) { if (hipMemcpy(out, W.S, sizeof(double)*(size_t)N*N, hipMemcpyDeviceToHost) != hipSuccess) return TSBA_ERR_DEVICE; }
    else { std::vector<double> hb(c->S_count); if (hipMemcpy(hb.data(), c->S_alloc, sizeof(double)*c->S_count, hipMemcpyDeviceToHost) != hipSuccess) return TSBA_ERR_DEVICE;
        const long long LDB = W.ldS + 1, Wb = LDB - c->S_up;
        for (long long i = 0; i < N; i++) for (long long j = 0; j < N; j++)
            out[i*N + j] = (j >= i - Wb && j <= i + c->S_up - 1) ? hb[(size_t)(Wb + i*(LDB - 1) + j)] : 0.0; }
    // row N = the rhs row: of the large-system solver if that ran, else unused (the LDS solver keeps it on chip)
    return hipMemcpy(out + (size_t)N*N, W.Sy, sizeof(double)*(size_t)N, hipMemcpyDeviceToHost) == hipSuccess ? 0 : TSBA_ERR_DEVICE;
}

int tsba_debug_band_factor(void *ctx, double *lcol, long long n_lcol, double *ldbuf, long long n_ld) {      // test hook: streaming band solver's factor
    Ctx *c = (Ctx *)ctx; if (!c || !c->uploaded || !c->Lcol) return TSBA_ERR_STATE;
    hipSetDevice(c->device); hipStreamSynchronize(c->stream);
    if (hipMemcpy(lcol, c->Lcol, sizeof(double)*n_lcol, hipMemcpyDeviceToHost) != hipSuccess) return TSBA_ERR_DEVICE;
    return hipMemcpy(ldbuf, c->W.LDbuf, sizeof(double)*n_ld, hipMemcpyDeviceToHost) == hipSuccess ? 0 : TSBA_ERR_DEVICE;
}
// host-side index arithmetic of the partitioned band solver, for the CPU test-suite (no device needed):
// out5 = { P, a, b, has_left, has_right } of interior p;  block index of (br, bc) in the cyclic-reduction pool and the pool size
// host-only: FNV-1a over the Schur slot-pair lists of the plan of `level`, built with `threads` host threads in the parallel sections
// (0 = the production choice): the plan must not depend on the number of threads
static int tsba_plan_checksum_ring = 0;       // ring_max_blocks the checksum hook builds its plan with (knob 2)
void tsba_debug_plan_knob(int which, int value) { if (which == 0) tsba_plan_threads = value; else if (which == 1) tsba_plan_mark_mt = value; else if (which == 2) tsba_plan_checksum_ring = value; else if (which == 3) tsba_plan_pin = value; }   // host-only measurement knobs
} // extern "C"
static unsigned long long plan_checksum(const HostPlan &H) {           // over EVERY list of the plan
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](const std::vector<int32_t> &v) { for (int32_t x : v) { h ^= (unsigned int)x; h *= 1099511628211ull; } h ^= v.size(); h *= 1099511628211ull; };
    mix(H.sb_a); mix(H.sb_b); mix(H.sb_pt_off); mix(H.sb_pt_s1); mix(H.sb_pt_s2); mix(H.sb_pt_lm); mix(H.sb_tx_off); mix(H.sb_tx_s1); mix(H.sb_tx_s2); mix(H.sb_tx_lm);
    for (const std::vector<int32_t> *v : { &H.kf_order, &H.sc_obs, &H.sc_kf, &H.sc_pt, &H.sc_flag, &H.sc_slot, &H.pair_i, &H.pair_h, &H.pair_hpos, &H.pair_sc_off, &H.pair_tg_off, &H.pair_tg,
                                           &H.tg_tobs, &H.tg_kf, &H.tg_text, &H.tg_pair, &H.tg_slot, &H.pt_pose6, &H.pt_pair4, &H.tg_ppos, &H.pf_g, &H.pf_f, &H.tg_rec,
                                           &H.pls_off, &H.pslot_pose, &H.pslot_pair, &H.pslot_lm, &H.tls_off, &H.tslot_pose, &H.tslot_pair, &H.tslot_lm, &H.sb_pab, &H.sb_pba,
                                           &H.pose_t_off, &H.pose_t, &H.pose_h_off, &H.pose_h, &H.pose_ps_off, &H.pose_ps, &H.pose_ps_lm, &H.pose_ts_off, &H.pose_ts, &H.pose_ts_lm }) mix(*v);
    if (H.far_B > 0) { h ^= (unsigned long long)H.far_B; h *= 1099511628211ull;
        for (const std::vector<int32_t> *v : { &H.far_a, &H.far_b, &H.far_off, &H.far_ent, &H.fb_id, &H.fb_pab, &H.fb_pba, &H.fb_pt_off, &H.fb_pt_s1, &H.fb_pt_s2, &H.fb_pt_lm, &H.fb_tx_off, &H.fb_tx_s1, &H.fb_tx_s2, &H.fb_tx_lm }) mix(*v); }
    for (double x : H.sc_uv) { unsigned long long u; memcpy(&u, &x, 8); h ^= u; h *= 1099511628211ull; }
    h ^= (unsigned long long)(H.bw_pose*4 + H.ring*2) + 8ull*(unsigned)H.ring_k0; h *= 1099511628211ull;
    return h;
}
extern "C" {
unsigned long long tsba_debug_plan_checksum(const tsba_problem *p, const tsba_options *o, int level, int threads) {
    if (!p || !o || level < 0 || level >= p->n_levels) return 0;
    const int saved = tsba_plan_threads; tsba_plan_threads = threads;
    HostPlan H; build_plan(p, o, level, H, false, true, tsba_plan_checksum_ring);
    tsba_plan_threads = saved;
    return plan_checksum(H);
}
// the plan of (p, o, level) built into a plan object that held the plan of (warm, ow, warm_level) before -- as a context does from call to call
unsigned long long tsba_debug_plan_checksum_recycled(const tsba_problem *warm, const tsba_options *ow, int warm_level, int warm_threads,
                                                     const tsba_problem *p, const tsba_options *o, int level, int threads) {
    if (!warm || !ow || !p || !o || level < 0 || level >= p->n_levels || warm_level < 0 || warm_level >= warm->n_levels) return 0;
    const int saved = tsba_plan_threads;
    HostPlan H;
    tsba_plan_threads = warm_threads; build_plan(warm, ow, warm_level, H, false, true, tsba_plan_checksum_ring);
    tsba_plan_threads = threads; build_plan(p, o, level, H, false, true, tsba_plan_checksum_ring);
    tsba_plan_threads = saved;
    return plan_checksum(H);
}
// host-only: the band + long-range split of the plan of `level` (tsba_plan.h: HostPlan::far_* / fb_*) when bands of up to far_max_blocks pose blocks are allowed.
// out[0] band of the preconditioner M in pose blocks (0: no split: ring, reordering or plain band), [1] 6x6 blocks of E (all ranks'), [2] those this rank
// contributes to, [3] the plan's band bound, [4] ring, [5] keyframes reordered, [6] checksum of the block positions (low 31 bits), [7] slot pairs of E.
// Checks what the split promises -- every block of M within the band, every slot pair of a landmark in exactly one of M / E, the positions of E
// sorted -- and returns TSBA_ERR_STATE if not.
int tsba_debug_plan_far(const tsba_problem *p, const tsba_options *o, int level, int far_max_blocks, int force, int ring_max_blocks, int32_t out[8]) {
    if (!p || !o || !out || level < 0 || level >= p->n_levels) return TSBA_ERR_ARG;
    HostPlan H; build_plan(p, o, level, H, false, true, ring_max_blocks, far_max_blocks, force != 0);
    int mine = 0; long long npairs = 0;
    if (H.far_B > 0) {
        for (int q = 0; q < H.n_sb(); q++) if (H.sb_b[q] - H.sb_a[q] > H.far_B) return TSBA_ERR_STATE;
        for (int q = 0; q < H.n_far(); q++) { if (H.far_a[q] >= H.far_b[q]) return TSBA_ERR_STATE;
            if (q > 0 && !(H.far_a[q-1] < H.far_a[q] || (H.far_a[q-1] == H.far_a[q] && H.far_b[q-1] < H.far_b[q]))) return TSBA_ERR_STATE;
            const bool has = H.fb_pt_off[q+1] > H.fb_pt_off[q] || H.fb_tx_off[q+1] > H.fb_tx_off[q] || H.fb_pab[q] >= 0 || H.fb_pba[q] >= 0; mine += has;
            for (int e = H.fb_pt_off[q]; e < H.fb_pt_off[q+1]; e++) if (H.pslot_pose[H.fb_pt_s1[e]] != H.far_a[q] || H.pslot_pose[H.fb_pt_s2[e]] != H.far_b[q] || H.pslot_lm[H.fb_pt_s1[e]] != H.pslot_lm[H.fb_pt_s2[e]]) return TSBA_ERR_STATE; }
        npairs = (long long)H.fb_pt_s1.size() + (long long)H.fb_tx_s1.size();
        // slot pairs with pose(s1) <= pose(s2): those of M + those of E = all of them
        long long all = 0; for (int j = 0; j < p->n_pt; j++) { const long long n = H.pls_off[j+1] - H.pls_off[j]; all += n*(n + 1)/2; }
        for (int j = 0; j < p->n_text; j++) { const long long n = H.tls_off[j+1] - H.tls_off[j]; all += n*(n + 1)/2; }
        if ((long long)H.sb_pt_s1.size() + (long long)H.sb_tx_s1.size() + npairs != all) return TSBA_ERR_STATE;
    }
    unsigned long long h = 1469598103934665603ull;
    for (const std::vector<int32_t> *v : { &H.far_a, &H.far_b, &H.far_off, &H.far_ent }) for (int32_t x : *v) { h ^= (unsigned int)x; h *= 1099511628211ull; }
    out[0] = H.far_B; out[1] = H.n_far(); out[2] = mine; out[3] = H.bw_pose; out[4] = H.ring; out[5] = H.kf_order.empty() ? 0 : 1; out[6] = (int32_t)(h & 0x7fffffffu);
    out[7] = (int32_t)npairs;
    return TSBA_OK;
}
// host-only: does the plan of `level` take the ring path (one loop closure between the last and the first keyframes) when separators of up
// to ring_max_blocks pose blocks are allowed?  Returns 1 / 0 (< 0: error); *bw_pose = the band of the plan either way
int tsba_debug_plan_ring(const tsba_problem *p, const tsba_options *o, int level, int ring_max_blocks, int32_t *bw_pose) {
    if (!p || !o || !bw_pose || level < 0 || level >= p->n_levels) return TSBA_ERR_ARG;
    HostPlan H; build_plan(p, o, level, H, false, true, ring_max_blocks);
    *bw_pose = H.bw_pose;
    return H.ring ? 1 + 16*H.ring_k0 : 0;     // (the first keyframe of the loop in the upper bits: 0 = the whole trajectory)
}
void tsba_debug_bandp_part_ring(int nb, int B, int Pmax, int p, int *out5) { const BandpPart r = bandp_part_ring(nb - B, 0, B, Pmax, Pmax, p); out5[0] = r.P; out5[1] = r.a; out5[2] = r.b; out5[3] = r.has_left; out5[4] = r.has_right; }
// ring with a tail: nf free poses, the loop starts at free row row0; out8 = P, a, b, has_left, has_right, G, Pt, label of the left separator
void tsba_debug_bandp_part_ring2(int nf, int row0, int B, int Pmax, int Gmax, int p, int *out8) { const BandpPart r = bandp_part_ring(nf, row0, B, Pmax, Gmax, p);
    out8[0] = r.P; out8[1] = r.a; out8[2] = r.b; out8[3] = r.has_left; out8[4] = r.has_right; out8[5] = r.G; out8[6] = r.Pt; out8[7] = r.lblL; }
void tsba_debug_bandp_part(int nb, int B, int Pmax, int p, int *out5) { const BandpPart r = bandp_part(nb, B, Pmax, p); out5[0] = r.P; out5[1] = r.a; out5[2] = r.b; out5[3] = r.has_left; out5[4] = r.has_right; }
long long tsba_debug_cr_blk_index(int mmax, int br, int bc) { return (long long)cr_blk_index(mmax, br, bc); }
long long tsba_debug_cr_pool_blocks(int mmax) { return (long long)cr_pool_blocks(mmax); }
int tsba_debug_stamps(void *ctx, long long *out64) {
    Ctx *c = (Ctx *)ctx; if (!c || !c->uploaded) return TSBA_ERR_STATE;
    hipSetDevice(c->device); hipStreamSynchronize(c->stream);
    return hipMemcpy(out64, c->W.dbg, 64*sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : TSBA_ERR_DEVICE;
}

static int load_rccl(Ctx *c) {
    if (c->rccl_so) return 0;
    // RCCL must sit on the SAME HIP runtime as this library: streams and device pointers do not cross runtimes.  A process can hold
    // two (a PyTorch wheel bundles its own libamdhip64 + librccl next to /opt/rocm's, and which one this library is bound to
    // depends on the load order), and a bare dlopen("librccl.so.1") returns whichever copy was loaded first.  So: find the
    // runtime our own HIP calls resolve to and take the librccl next to it, by full path.
    std::string tried;
    Dl_info di;
    if (dladdr((void *)&hipStreamSynchronize, &di) && di.dli_fname) {
        std::string dir(di.dli_fname); const size_t sl = dir.rfind('/'); dir = sl == std::string::npos ? std::string(".") : dir.substr(0, sl);
        for (const char *n : { "/librccl.so.1", "/librccl.so" }) {
            const std::string path = dir + n; tried += path + " ";
            c->rccl_so = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL); if (c->rccl_so) break; }
    }
    if (!c->rccl_so) for (const char *n : { "librccl.so.1", "librccl.so" }) { tried += std::string(n) + " "; c->rccl_so = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (c->rccl_so) break; }
    if (!c->rccl_so) { set_err(c, std::string("dlopen(librccl) failed, tried: ") + tried + ": " + dlerror()); return TSBA_ERR_COMM; }
    c->p_getid = (decltype(c->p_getid))dlsym(c->rccl_so, "ncclGetUniqueId");
    c->p_init = (decltype(c->p_init))dlsym(c->rccl_so, "ncclCommInitRank");
    c->p_allreduce = (decltype(c->p_allreduce))dlsym(c->rccl_so, "ncclAllReduce");
    c->p_destroy = (decltype(c->p_destroy))dlsym(c->rccl_so, "ncclCommDestroy");
    c->p_errstr = (decltype(c->p_errstr))dlsym(c->rccl_so, "ncclGetErrorString");
    c->p_count = (decltype(c->p_count))dlsym(c->rccl_so, "ncclCommCount");
    if (!c->p_getid || !c->p_init || !c->p_allreduce || !c->p_destroy || !c->p_errstr) { set_err(c, "librccl: missing symbols"); return TSBA_ERR_COMM; }
    return 0;
}
int tsba_comm_unique_id(void *ctx, void *id128) {
    Ctx *c = (Ctx *)ctx; if (!c || !id128) return TSBA_ERR_ARG;
    int rc = load_rccl(c); if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t r = c->p_getid(&id);
    if (r != ncclSuccess) { set_err(c, std::string("ncclGetUniqueId: ") + c->p_errstr(r)); return TSBA_ERR_COMM; }
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return TSBA_OK;
}
void *tsba_local_group_create(int world) {
    if (world < 1) return nullptr;
    LocalGroup *G = new LocalGroup(); G->world = world; G->stage.resize(world); return G;
}
void tsba_local_group_destroy(void *group) {
    LocalGroup *G = (LocalGroup *)group; if (!G) return;
    G->fail();
    for (Ctx *m : G->members) if (m && m->lgroup == G) { m->lgroup = nullptr; m->rank = 0; m->world = 1; m->uploaded = false; }   // (a resident problem was sharded for the group)
    delete G;
}
int tsba_comm_init_local(void *ctx, void *group, int rank, int world) {
    Ctx *c = (Ctx *)ctx; LocalGroup *G = (LocalGroup *)group;
    if (!c || !G || world != G->world || rank < 0 || rank >= world) return TSBA_ERR_ARG;
    hipSetDevice(c->device);
    free_problem(c);                               // any resident problem was sharded for the old world size
    c->lgroup = G; c->rank = rank; c->world = world;
    { std::lock_guard<std::mutex> lk(G->m); G->members.push_back(c); }
    return TSBA_OK;
}
int tsba_comm_stats(void *ctx, int32_t *ranks, int64_t bytes[3]) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    if (ranks) { int n = c->lgroup ? c->lgroup->world : 1;
        if (c->comm && c->p_count && c->p_count(c->comm, &n) != ncclSuccess) return TSBA_ERR_COMM;
        *ranks = n; }
    if (bytes) { bytes[0] = (int64_t)c->x_trial; bytes[1] = (int64_t)c->x_lin; bytes[2] = (int64_t)c->x_pass; }
    return TSBA_OK;
}
int tsba_debug_set(void *ctx, const tsba_debug_options *d) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    if (d) c->dbg = *d; else memset(&c->dbg, 0, sizeof(c->dbg));
    return TSBA_OK;
}
int tsba_comm_init(void *ctx, const void *id128, int rank, int world) {
    Ctx *c = (Ctx *)ctx; if (!c || world < 1 || rank < 0 || rank >= world) return TSBA_ERR_ARG;
    hipSetDevice(c->device);
    lgroup_forget(c);                              // (an RCCL communicator replaces an in-process group)
    if (!id128) { c->force_multi = world >= 1; c->rank = 0; c->world = 1; return TSBA_OK; }   // test hook: split kernels, no communicator
    int rc = load_rccl(c); if (rc) return rc;
    ncclUniqueId id; memcpy(&id, id128, 128);
    ncclResult_t r = c->p_init(&c->comm, world, id, rank);
    if (r != ncclSuccess) { set_err(c, std::string("ncclCommInitRank: ") + c->p_errstr(r)); return TSBA_ERR_COMM; }
    c->rank = rank; c->world = world;
    free_problem(c);                               // any resident problem was sharded for the old world size
    return TSBA_OK;
}

} // extern "C"
