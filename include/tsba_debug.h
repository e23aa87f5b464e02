/*
 * tsba_debug.h -- test and diagnostics hooks of libtsba.so.  NOT part of the drop-in surface (include/tsba.h): nothing here replaces a
 * function of the reference.  The parity tests use these to compare intermediate results (reduced camera system, Levenberg-Marquardt
 * trace) with the CPU oracle, to force one of several exact solver paths, and bench.py uses the timing hooks for its roofline object.
 */
#ifndef TSBA_DEBUG_H
#define TSBA_DEBUG_H
#include "tsba.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Debug aid: first linearisation of pass 0 of the uploaded problem and the damped reduced camera system for
 * `radius`: S [(6 n_kf)^2] (identity rows for constant / absent poses), g [6 n_kf], cost, kf_free [n_kf], dp [6 n_kf]
 * (the pose step S dp = -g).  Any output may be NULL. */
int  tsba_debug_reduced_system(void *ctx, double radius, double *S, double *g, double *cost, int32_t *kf_free, double *dp);

/* The same for large maps (band storage of the reduced system; the dense copy would be 7.2 GB at 5000 keyframes): n = 6 x free poses,
 * bw = sub-diagonals kept, ab[(i - j)*n + j] = S(i, j) for j <= i <= j + bw (LAPACK lower band, rows of the COMPRESSED free-pose
 * system), g [n], dp [6 n_kf] (by keyframe, 0 for constant poses).  ab / g / dp may be NULL (first call: sizes only). */
int  tsba_debug_reduced_band(void *ctx, double radius, int32_t *n, int32_t *bw, double *ab, double *g, double *dp);
/* Row block of every keyframe in the compressed reduced system of the last pass set-up (-1: constant / not participating).  Not
 * monotone in the keyframe index when the plan reordered the keyframes (solver_info [15]). */
int  tsba_debug_row_of_kf(void *ctx, int32_t *rowblk);

/* Which kernels the uploaded problem runs through, so that a test can assert it exercises the path it means to.  out[16]:
 * [0] reduced system solved in LDS  [1] band storage of S  [2] streaming band solver  [3] interiors P of the partitioned solver
 * [4] separator system by cyclic reduction  [5] band rows  [6] four (target, host) pairs per wave in the linearisation
 * [7] fused pose-only kernel  [8] large-map Schur / pose-sum kernels  [9] world size  [10] rank
 * [11..14] this rank's plan of the first pass's level: (target, host) pairs, S blocks, scene candidates, point slots
 * [15] rows of S in reverse Cuthill-McKee order of the keyframes (wide envelopes: loop closures)
 * [16] (n >= 17) ring-shaped map solved with ghost rows for the first separator (one loop closure between the last and the first keyframes)
 * [17] (n >= 19) long-range coupling: band of the preconditioner in pose blocks (0: direct solve)  [18] 6x6 blocks outside the band */
int  tsba_debug_solver_info(void *ctx, int32_t *out, int n);
/* Maps with long-range coupling: the conjugate-gradient solves of the last tsba_solve.  out[0] iterations in total, [1] reduced systems
 * solved (LM trials), [2] most iterations of one system, [3] systems that hit the iteration cap. */
int  tsba_debug_pcg_stats(void *ctx, int32_t out[4]);
/* Test hook: M X = R for T right-hand sides with the band factor of the last solve / tsba_debug_reduced_system (the solve phase that the
 * iterative and low-rank solvers of maps with long-range coupling run on).  R, X: [6 x free poses][T], row-major. */
int  tsba_debug_multi_solve(void *ctx, int T, const double *R, double *X);   /* T = -1: one column through the single-vector solve phase */
int  tsba_debug_sv_lmax(int n_kf, int B, int Pmax);                          /* host only: the bound on an interior's length the solve phase sizes its LDS with */
/* Plane cache of the context (tsba_problem.kf_id): out[0] keyframes whose planes were found on the device, out[1] keyframes copied. */
int  tsba_debug_img_cache_stats(void *ctx, int64_t out[2]);
/* The 6x6 blocks of the reduced system outside the band (solver_info [18] of them) as left by tsba_debug_reduced_system / the last solve:
 * keyframes a < b of every block and its 36 values, row-major, rows = keyframe a.  Any output may be NULL. */
int  tsba_debug_far_blocks(void *ctx, int32_t *a, int32_t *b, double *blocks);

/* Average duration (ms) of the linearisation kernel (residual + Jacobian + robust weight + normal-
 * equation accumulation) over n launches on the library's stream, measured with HIP events.
 * Requires an uploaded problem; `level` selects the pass.  Also returns the algorithmic bytes of one launch. */
int  tsba_time_linearize(void *ctx, int level, int n, double *avg_ms, double *algo_bytes);

/* Average duration (ms) of the reduced-system solve (every kernel between the Schur complement and the landmark back-substitution)
 * over n repetitions on the S, g left by the last solve / tsba_debug_reduced_system; HIP events on the library's stream. */
int  tsba_debug_time_solve(void *ctx, int n, double *avg_ms);

/* Test / diagnostics switches of one context (never read from the environment; all zero = production behaviour).  They select
 * between solver paths that are all exact -- none of them changes what is computed, only by which kernels. */
typedef struct tsba_debug_options {
    int32_t band_parts;        /* > 0: number of interiors of the partitioned band solver (1 = single-workgroup streaming solver) */
    int32_t sep_solver;        /* separator system: 0 cost model, 1 sequential streaming solver, 2 block cyclic reduction (one launch per level), 3 cyclic reduction by the pivot / update / back kernels of round 1, 4 as 2 with the separator system assembled by the border / sep kernels instead of the fused one */
    int32_t no_band_stream;    /* 1: large systems through the wide-band multi-workgroup Cholesky even when the band is narrow */
    int32_t no_pose_kernel;    /* 1: PoseOptim through the general pipeline instead of the fused pose-only kernel */
    int32_t no_small_pairs;    /* 1: never put four (target, host) pairs on one wave of the linearisation */
    int32_t verbose;           /* 1: host-side timing of upload / plan construction on stderr */
    int32_t no_kf_reorder;     /* 1: keep the rows of S in keyframe order even when the envelope is wide (loop closures) */
    int32_t no_schur_quad;     /* 1: large maps assemble S with one wave per 6x6 block (k_schur_t<1>) instead of four blocks per wave */
    int32_t no_ring;           /* 1: a ring-shaped map (one loop closure between the last and the first keyframes) through the reordering path instead of the ghost-row partition */
    int32_t far_solver;        // maps with long-range coupling (band part + blocks between a landmark's clusters, tsba_pcg.h): 0 by the plan's rule (when no keyframe order brings the envelope within the band solvers' reach), 1 never (reordering / wide-band Cholesky as before), 2 whenever the map is eligible, 3 as 2 without the low-rank correction for loop closures (tsba_wb.h: A/B runs of the plain iterations)
    int32_t pcg_max_it;        // > 0: iteration cap of the conjugate gradients (default 200)
    int32_t pcg_tol_exp;       // > 0: relative tolerance 10^-pcg_tol_exp of the conjugate gradients in the M^-1 norm (default 10)
    int32_t pcg_refactor;      // preconditioner of the single-vector iteration: 0 the single-vector solve phase (tsba_bandsv.h) where it exists, else the factorisation re-run with the residual as right-hand side; 1: always the re-run; 2: the many-column solve phase with one column; 3: as 0 with r.z by its own kernel instead of inside the solve phase (A/B runs)
    int32_t pcg_block;         // 0 / 1: the single-vector iteration, 2: enlarged conjugate gradients (32 columns per preconditioner application, tsba_pcg.h) where the many-column solve phase of the band solver exists
    int32_t solve_variant;     // reduced system of a small window (one workgroup, S in LDS): 0 the two-panel-wave schedule (tsba_solve.h), 1 the look-ahead schedule with a separate diagonal wave (tsba_solve_la.h: built in round 4 and measured SLOWER, 39.3 against 33.2 us on C4; kept for A/B runs), 2 as 1 with the update tiles handed out in wave order, 3 the production solver with the back-substitution and the step decision as launches of their own (k_solve_t + k_back + k_decide per trial, production until round 4) instead of in the solver's launch (k_solve_back) and in the next trial's k_schur_t, 4 as 3 with every 6x6 diagonal block factored from a scratch copy in LDS by every lane of the panel waves (production until round 5) instead of in place across six lanes with v_readlane broadcasts (bit-identical), 5 as 0 (production) with that scratch-copy factorisation
    int32_t sv_per_level;      // bit 0: the separator tree of the single-vector solve phase as one launch per level (k_sv_cre_fwd / _top / _back, production until round 4) instead of one launch for the whole tree (k_sv_cre_tree); bit 1: the back substitution of the factorisation's own solve on maps with long-range blocks as a launch per level (k_cre_back) instead of one launch through the solve phase's products (k_cre_back_tree); bit 2: the update step of a conjugate-gradient iteration (alpha; x, r) as a launch of its own (k_pcg_update) instead of inside the first kernel of the preconditioner application; bit 3: the interiors' back substitution of the solve phase as a launch of its own (k_sv_back_int) instead of in the tree's launch (k_sv_tree_back): A/B runs, bit-identity / parity tests
    int32_t host_pair_lists;   // 0: on the device for problems of at least 4096 scene observations, on the host below; 2: on the device at any size; 1: the slot pairs of the S blocks are built on the host (as every map did until round 4 and every window until round 6) instead of on the device (tsba_devplan.h): A/B runs, list comparison; and single-frame problems (tsba_pose_optim) get their plan from the generic builder on plan threads, the later passes' levels staged during the solve (until round 6), instead of build_plan_single_frame on the calling thread
    int32_t pass_launches;     // 1: a window's pass begins and ends with the launches of rounds 1-4 (k_pass_reset, k_participation, k_gauge_wave, k_musigma | k_outlier, state copy) instead of k_pass_begin | k_pass_end (tsba_kernels_pass.h); PoseOptim: a launch per LM step (k_pose_iter) instead of one per pass (k_pose_pass): A/B runs, agreement tests
    int32_t trial_launches;    // windows: 0 production (k_linearize, k_mid<256> as two launches); 2: the round-5 experiment k_lin_mid -- k_mid inside the speculative linearisation's launch, its last workgroups to finish taking k_mid's blocks of 128 -- measured SLOWER (40.7 against 13.4 + 10.6 us: 736 workgroups signalling completion cost more than the kernel boundary); 1: the two launches with k_mid's blocks of 128 (the experiment's bit-identical comparison partner)
    int32_t assume_cus;        // > 0: the residency test of the kernels whose workgroups poll each other (k_solve_back, k_sv_cre_tree, k_sv_tree_back, k_cre_back_tree) assumes a device of this many compute units (tests: a device too small for the grid takes the launch-per-step path)
    int32_t lds_poison;        // 1 / 2 / 3: before every launch of tsba_solve the LDS of every compute unit is filled with NaNs / 1e300 / 0x5a bytes (what another context's kernels may leave there): results must not change
} tsba_debug_options;
int  tsba_debug_set(void *ctx, const tsba_debug_options *d);   /* d == NULL: back to production behaviour; applies to the next upload */


/* The reduced system of the first linearisation as 6x6 blocks keyed by KEYFRAME pairs, whatever the storage behind it (band rows in any
 * keyframe order, the ghost rows of a ring map, the blocks outside the band of a map with long-range coupling): block q couples keyframes
 * kf_r[q], kf_c[q] and holds S(rows of kf_r, columns of kf_c), row-major; a keyframe pair may appear more than once (the parts add).
 * First call with kf_r == NULL: *nblk = number of blocks; second call with buffers.  g_kf, dp_kf [6 n_kf] by keyframe (0 for constant
 * poses), cost = the cost at the linearisation point.  Band storage only (maps of more than 31 keyframes). */
int  tsba_debug_reduced_blocks(void *ctx, double radius, int32_t *nblk, int32_t *kf_r, int32_t *kf_c, double *val, double *g_kf, double *dp_kf, double *cost);
/* Per LM trial of pass `pass` of the last solve: out[4 k] = candidate cost (NaN: invalid step), [4 k + 1] = model cost change,
 * [4 k + 2] = radius after the decision, [4 k + 3] = 1 accepted / 0 rejected / -1 invalid step / 2 tolerance exit on this trial.
 * Returns the number of trials recorded or a negative error.  (The fused pose-only kernel keeps no trace.) */
int  tsba_debug_lm_trace(void *ctx, int pass, double *out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* TSBA_DEBUG_H */
