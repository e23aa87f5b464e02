// Device-side structs of libtsba.so: LM state, per-level plan views, linearisation buffers, work buffers.  (part of the single translation unit tsba.hip: included there, in this order)
#pragma once
typedef double v2d __attribute__((ext_vector_type(2)));
// Kernels whose workgroups poll values that other workgroups of the same launch publish (k_solve_back, k_sv_cre_tree, k_sv_tree_back, k_cre_back_tree, and the
// roles of k_lin_mid) bound their polling; a thread that gives up counts itself here.  A solve reports the difference over its duration
// (tsba_report.poll_timeouts): a give-up fails the linear solve of its LM trial -- this says that it was a wait and not the numbers.  ONE counter per process:
// a solve's figure also counts the give-ups of other contexts that solved at the same time (tracking's PoseOptim beside a mapping BA), and it counts threads,
// not events -- it answers "did anything give up while this solve ran", which is what the callers (and the pose-only retry in tsba_solve) ask.
__device__ unsigned int ts_poll_giveups;
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
    const int *tg_tobs, *tg_kf, *tg_text, *tg_pair, *tg_slot, *tg_rec, *tg_ppos, *pt_pose6, *pt_pair4, *tx_pair8;      // pt_pair4: PT_PAIRN (6) entries per point
    const int *pls_off, *pslot_pose, *pslot_pair, *pslot_lm, *tls_off, *tslot_pose, *tslot_pair, *tslot_lm;
    const int *sb_rng;       // [n_sb][4]: a diagonal block's (pose_t_off[a], pose_t_off[a+1], pose_h_off[a], pose_h_off[a+1])
    const int *sb_a, *sb_b, *sb_pab, *sb_pba, *sb_pt_off, *sb_pt_s1, *sb_pt_s2, *sb_pt_lm, *sb_tx_off, *sb_tx_s1, *sb_tx_s2, *sb_tx_lm;
    const int *pose_t_off, *pose_t, *pose_h_off, *pose_h, *pose_ps_off, *pose_ps, *pose_ps_lm, *pose_ts_off, *pose_ts, *pose_ts_lm;
    const int *tfeat_off, *tfeat_raw; const double *tfeat_uv, *tfeat_ref;
    const int *pf_g, *pf_f; int n_pf;   // pose-only path: flat (group, feature) list of the frame's text features
    const int *kf_order;                // nullptr: the rows of S follow the keyframe index; else kf_order[i] = keyframe at position i (tsba_plan.h: rcm_order)
    // band + long-range coupling (HostPlan::far_* / fb_*, tsba_pcg.h): nullptr / 0 unless the plan split the reduced system into M (the sb_* lists) + E.
    // sb_far: nullptr in this view; in the view of E that launch_schur derives (sb_* = the fb_* lists) the index of every block in W.Sfar
    const int *sb_far, *far_a, *far_b, *far_off, *far_ent; int n_far, far_B;
    int n_far_ent; int2 *far_rec;       // [n_far_ent = entries of far_ent]: (the entry, the row of the block's OTHER keyframe among the free poses or -1): filled by k_far_rows once the pass's gauge is fixed, so that the matrix-vector product of tsba_pcg.h finds a block in one hop instead of three
    const int *wb_kf, *wb_idx; int n_wb;   // keyframes touched by E when they are few (tsba_wb.h)
    const int *fb_id, *fb_pab, *fb_pba, *fb_pt_off, *fb_pt_s1, *fb_pt_s2, *fb_pt_lm, *fb_tx_off, *fb_tx_s1, *fb_tx_s2, *fb_tx_lm;
};

#define TSBA_TRACE_CAP 64
#define PT_REC 8
#define PT_VDB 4
// layout experiments (A/B builds): -DTSBA_VDB_SOA keeps V | dgs | b as three arrays [3][n_pt], -DTSBA_PAIRR_SOA keeps the pair rotations as [9][n_pair]
#ifdef TSBA_VDB_SOA
#define VDB_LOAD(B, j, npt, V_, D_) do { V_ = (B).vdb_pt[(size_t)(j)]; D_ = (B).vdb_pt[(size_t)(npt) + (j)]; } while (0)
#define VDB_LOADB(B, j, npt, V_, D_, B_) do { V_ = (B).vdb_pt[(size_t)(j)]; D_ = (B).vdb_pt[(size_t)(npt) + (j)]; B_ = (B).vdb_pt[2*(size_t)(npt) + (j)]; } while (0)
#define VDB_STORE(B, j, npt, V_, D_, B_) do { (B).vdb_pt[(size_t)(j)] = V_; (B).vdb_pt[(size_t)(npt) + (j)] = D_; (B).vdb_pt[2*(size_t)(npt) + (j)] = B_; } while (0)
#else
#define VDB_LOAD(B, j, npt, V_, D_) do { const v2d vd_ = *(const v2d *)((B).vdb_pt + PT_VDB*(size_t)(j)); V_ = vd_.x; D_ = vd_.y; } while (0)
#define VDB_LOADB(B, j, npt, V_, D_, B_) do { const double *rec_ = (B).vdb_pt + PT_VDB*(size_t)(j); const v2d vd_ = *(const v2d *)rec_; V_ = vd_.x; D_ = vd_.y; B_ = rec_[2]; } while (0)
#define VDB_STORE(B, j, npt, V_, D_, B_) do { double *rec_ = (B).vdb_pt + PT_VDB*(size_t)(j); ((v2d *)rec_)[0] = v2d{V_, D_}; rec_[2] = B_; } while (0)
#endif
#ifdef TSBA_PAIRR_SOA
#define PAIRR(B, p, k, np) (B).pairR[(size_t)(k)*(np) + (p)]
#else
#define PAIRR(B, p, k, np) (B).pairR[9*(size_t)(p) + (k)]
#endif
#define TX_REC 28
struct LinBuf {              // everything one linearisation produces
    double *pairM, *pairCost, *pairR, *pairOut, *tgM, *tgCost;
    double *w_pt;                       // per point slot, one 64-byte record: w[0..5] | v | b   (PT_REC doubles; the host slot of a landmark
                                        // holds its host column -sum Q^T w in [0..5], formed by k_mid from w and the pair's R_cr)
    double *vdb_pt;                     // per point ONE 32-byte record: V | clamp(sigma^2 V)/sigma^2 (lambda = that / radius) | b | -   (three separate arrays were three gathers
                                        // per slot pair in the Schur kernels, which are bound by the number of lines they pull)
    double *w_tx;                       // per plane slot, one 224-byte record: W[0..17] | V6 [18..23] | b3 [24..26]   (TX_REC doubles)
    double *V_tx, *b_tx, *dgs_tx;       // per plane: V [6][n], b [3][n], dgs [3][n]
    double *Hd, *bp, *dgs_p;            // per pose: diag(H_pp), gradient, dgs.  Hd | bp | scal[8] are one allocation (hb):
    double *bp_loc;                     // multi-GPU: this rank's part of bp (the reduced gradient is assembled from it)
    double *lmpart;                     // per k_mid block: (gradient max, |x|^2) of its landmarks, cost of its pairs (+ their text groups)
};

struct PoseState;
// state of the conjugate-gradient iteration (tsba_pcg.h), double-buffered by the iteration's parity
struct PcgState { double rz, rz0, best; int it, since; };      // best: smallest r.z so far; since: iterations since it improved by a tenth (stagnation at the attainable accuracy)
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
    double *trace; int trace_pass;      // per LM trial of the pass being solved: candidate cost, model cost change, radius after the decision, decision (tsba_debug_lm_trace; [TSBA_MAX_LEVELS][TSBA_TRACE_CAP][4])
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
    int dp_poll;                        // small windows: the back-substitution runs in the solver's launch and polls dp [N] + the failure flag dp[N] (k_solve_back); k_postlin / k_decide leave NaN there
    double *partial;                    // [nblocks_back][2]
    int *cntpart;                       // per k_participation workgroup: active scene blocks, active text blocks
    double *posepart;                   // large maps: per k_pose_sums workgroup (21 poses): gradient max, |x|^2
    unsigned int *poll0;                // ts_poll_giveups at the start of the solve (k_reset_state)
    LmState *st, *st_next;              // st_next (windows, single GPU; else null): the copy of the state that k_schur_t writes when it takes the previous trial's decision itself -- the host swaps the two after that launch
    PoseState *pst; double *ppart;      // pose-only path (tsba_pose.h): double-buffered state, [2][G][28] partial sums
    // band + long-range blocks, preconditioned conjugate gradients (tsba_pcg.h): the blocks outside the band [n_far][36] (rows: the earlier keyframe),
    // the iteration's vectors in the compressed row space of S, per-workgroup partial sums [2][workgroups], double-buffered scalars, statistics
    double *Sfar, *pc_x, *pc_r, *pc_p[2], *pc_q, *pc_g0, *pc_part;
    struct PcgState *pcs; int *pc_stat;
};

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { set_err(c, std::string(#x) + ": " + hipGetErrorString(e_)); return TSBA_ERR_DEVICE; } } while (0)

