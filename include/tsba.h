/*
 * tsba.h -- C ABI of the MI355X-native bundle-adjustment / pose-optimisation back-end.
 *
 * Drop-in boundary for TextSLAM's optimizer:: hot path.  Every entry point replaces the
 * numeric core of one public optimizer method of the reference (citations are relative to
 * the TextSLAM tree):
 *
 *   tsba_local_ba     <- optimizer::LocalBundleAdjustment   src/optimizer.cc:197-331  (PyrBA :1330-1698)
 *   tsba_pose_optim   <- optimizer::PoseOptim               src/optimizer.cc:135-195  (PyrPoseOptim :1060-1327)
 *   tsba_global_ba    <- optimizer::GlobalBA                src/optimizer.cc:334-453  (PyrGlobalBA :1701-1851)
 *   tsba_eval         <- ceres::Problem::Evaluate as used at src/optimizer.cc:1228-1231, 1609-1612
 *
 * The reference methods traffic in Eigen / OpenCV / keyframe* object graphs; the header-only
 * adapter shown in INTEGRATION.md gathers those into the flat, plain-pointer description
 * below (row B1 of SURVEY.md section 8a) and scatters the results back (rows O2/O3).
 *
 * Conventions
 *   - pose[7] = (qw,qx,qy,qz, tx,ty,tz) of T_cw (world -> camera)      optimizer.cc:84-90
 *   - scene point = host KF r + ray (mx,my,1) + inverse depth rho       mapPts.cc:49-69
 *   - text plane  = host KF r + theta (= n/d in the host camera frame)   ModelTool.hpp:164-171
 *   - K_l = K / 2^l with K_l(2,2) = 1                                    optimizer.cc:43-52
 *   - all arithmetic fp64 (as the reference); images uint8, row stride == width
 *   - every pointer is a HOST pointer owned by the caller; the library keeps device mirrors in ctx
 *   - return 0 = OK, negative = error (the library never calls exit())
 *   - one ctx per calling thread; calls are synchronous
 */
#ifndef TSBA_H
#define TSBA_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSBA_MAX_LEVELS 4
#define TSBA_NTAP 8            /* INTERVAL8 pattern, tool.cc:1550-1561 */

/* error codes */
#define TSBA_OK              0
#define TSBA_ERR_ARG        -1
#define TSBA_ERR_DEVICE     -2   /* no HIP device / HIP runtime error */
#define TSBA_ERR_NUMERIC    -3   /* Cholesky breakdown, NaN */
#define TSBA_ERR_STATE      -4   /* call order (no problem uploaded ...) */
#define TSBA_ERR_COMM       -5   /* RCCL failure */

/* BAStatus, setting.h */
#define TSBA_STATE_NOTREACHWIN 0
#define TSBA_STATE_LOCAL       1
#define TSBA_STATE_GLOBAL      2

typedef struct tsba_problem {
    int32_t n_kf, n_pt, n_text;
    int32_t n_levels;                 /* number of pyramid levels described below (1..4) */
    double  K[4];                     /* fx, fy, cx, cy of level 0 */

    /* ---- parameters (in/out) ---- */
    double  *pose;                    /* [n_kf][7] */
    double  *rho;                     /* [n_pt] */
    double  *theta;                   /* [n_text][3] */

    /* ---- gauge (row L4): KFs with mnId 0/1, optimizer.cc:274-275 ---- */
    const uint8_t *kf_initial;        /* [n_kf] 1 = "InitialIdx" keyframe */

    /* ---- scene points ---- */
    const double  *pt_ray;            /* [n_pt][2] mx,my   (mapPts::GetRaydir, z == 1) */
    const int32_t *pt_host;           /* [n_pt] KF index of the host, or -1: host outside the
                                         window => landmark and host pose frozen (vMapPtOptim=false) */
    const double  *pt_host_Trw;       /* [n_pt][12] row-major 3x4 T_rw of the host (RefKF->mTcw);
                                         read only where pt_host < 0; may be NULL if none */
    /* ---- text planes ---- */
    const int32_t *text_host;         /* [n_text] KF index or -1 (vMapTextOptim=false) */
    const double  *text_host_Twr;     /* [n_text][12] row-major 3x4 T_wr (RefKF->mTwc); read where text_host < 0 */
    const double  *text_box_ray;      /* [n_text][4][2] vTextDeteRay, mapText.cc:87-90 */

    /* ---- scene observations, per level, in the reference's residual order
     *      (KF-major, then vSceneObv2d[level] order; optimizer.cc:1366-1435) ---- */
    int32_t        n_sobs[TSBA_MAX_LEVELS];
    const int32_t *sobs_kf[TSBA_MAX_LEVELS];    /* [n_sobs] target KF index */
    const int32_t *sobs_pt[TSBA_MAX_LEVELS];    /* [n_sobs] point index */
    const int32_t *sobs_flag[TSBA_MAX_LEVELS];  /* [n_sobs] index into sgood[] (KF offset + IdxToRaw) */
    const double  *sobs_uv0[TSBA_MAX_LEVELS];   /* [n_sobs][2] LEVEL-0 pixel (SceneUse0Pyr, optimizer.cc:1336) */
    int32_t        n_sgood;
    uint8_t       *sgood;                       /* [n_sgood] vObvGoodPts of all KFs, concatenated; in/out */

    /* ---- text reference features per level (mapText::vRefFeature[level]), grouped by text ---- */
    int32_t        n_tfeat[TSBA_MAX_LEVELS];
    const int32_t *tfeat_off[TSBA_MAX_LEVELS];  /* [n_text+1] CSR: features of text j are [off[j], off[j+1]) */
    const int32_t *tfeat_raw[TSBA_MAX_LEVELS];  /* [n_tfeat] IdxToRaw (index of the level-0 feature) */
    const double  *tfeat_uv[TSBA_MAX_LEVELS];   /* [n_tfeat][2] feature centre (u,v) in the HOST level-l image;
                                                   the 8 tap rays are ((u+dx-cx_l)/fx_l,(v+dy-cy_l)/fy_l,1), tool.cc:1550-1566 */
    const double  *tfeat_ref[TSBA_MAX_LEVELS];  /* [n_tfeat][8] neighbourNInten (normalised host intensities) */

    /* ---- text observations: (KF, text) pairs, KF-major (GetStateTextObvs order) ---- */
    int32_t        n_tobs;
    const int32_t *tobs_kf;           /* [n_tobs] */
    const int32_t *tobs_text;         /* [n_tobs] */
    uint8_t       *tobs_good;         /* [n_tobs] vObvGoodTexts; in/out */
    const int32_t *tobs_fgood_off;    /* [n_tobs+1] offsets into tfgood (one flag per LEVEL-0 feature of the text) */
    uint8_t       *tfgood;            /* [tobs_fgood_off[n_tobs]] vObvGoodTextFeats; in/out */

    /* ---- images: level l of KF k = img[l][k], continuous uint8, img_w[l] x img_h[l] ---- */
    const uint8_t *const *img[TSBA_MAX_LEVELS]; /* img[l] = array of n_kf pointers (may be NULL if no text) */
    int32_t        img_w[TSBA_MAX_LEVELS], img_h[TSBA_MAX_LEVELS];

    /* ---- optional: identity of the keyframes (keyframe::mnId).  A keyframe's pyramid planes never change after its creation, and
     * LocalBundleAdjustment is called once per new keyframe on the last 20 (tracking.cc:828-842: 19 of them were in the last call).
     * With kf_id != NULL the context keeps the planes of the keyframes it has seen on the device (up to TSBA_IMG_CACHE_KF of them, least
     * recently used out first) and a call copies only the planes of keyframes it has not seen -- the caller promises that equal ids mean
     * equal images at every level.  NULL: every call copies every plane.  Ignored with tsba_options.img_on_device. */
    const int64_t *kf_id;             /* [n_kf] or NULL */
} tsba_problem;
#define TSBA_IMG_CACHE_KF 64

typedef struct tsba_options {
    /* residual weights and robust kernels, optimizer.cc:1350-1351,1369,1454 */
    double  w_sx, w_sy, w_t;
    double  huber_scene, huber_text;
    /* pyramid passes, optimizer.cc:282-289 (local), :174-186 (pose), :411-414 (global) */
    int32_t n_passes;
    int32_t levels[TSBA_MAX_LEVELS];
    int32_t its[TSBA_MAX_LEVELS];
    double  chi2_mono[TSBA_MAX_LEVELS];
    double  chi2_text[TSBA_MAX_LEVELS];
    double  text_bad_ratio;          /* 0.99 */
    int32_t state;                   /* TSBA_STATE_* (gauge rule, optimizer.cc:1571-1588) */
    int32_t outlier_scene, outlier_text;   /* O1 on/off (off when bFlag_rapid) */
    int32_t use_text;                /* FLAG_TEXT */
    int32_t filter_good;             /* 1: skip observations whose good-flag is 0 (local/pose); 0: global BA */
    int32_t text_jacobian;           /* 0 analytic bilinear (HIP path), 1 Ceres CENTRAL numeric diff (oracle only) */
    /* Levenberg-Marquardt constants = Ceres 1.x defaults */
    double  initial_radius, max_radius, min_radius;
    double  min_relative_decrease;
    double  function_tolerance, gradient_tolerance, parameter_tolerance;
    double  min_diagonal, max_diagonal;
    /* multi-GPU (global BA): this rank keeps the residual blocks of the landmarks hosted in keyframes [lm_shard, lm_shard + 1) * n_kf /
     * lm_nshard (co-visibility is local in keyframe index: the rank's pairs and S blocks stay near its own range); the observations of
     * a frozen landmark go with their target keyframe */
    int32_t lm_shard, lm_nshard;
    /* 1: the pointers in tsba_problem.img are DEVICE pointers on this context's GPU (tsframe_level_ptr planes, include/tsframe.h:
     * the pyramid frame::GetPyrMat left in HBM) -- the upload reads them in place, no host round trip; they must stay valid and
     * unchanged until the last solve on the upload.  0: host pointers, copied. */
    int32_t img_on_device;
    /* 1: the host threads that build the index plan of a LARGE map (> 100 k observations: up to 15 short-lived workers) are pinned to the CPUs
     * that share the calling thread's last-level cache for the duration of the build (5000 keyframes on a two-socket host: 21 -> 13 ms).
     * 0 (default): the library leaves scheduling alone.  The calling thread's own affinity is never changed either way. */
    int32_t host_plan_pin;
} tsba_options;

typedef struct tsba_report {
    int32_t status;
    int32_t n_passes;
    int32_t iters[TSBA_MAX_LEVELS];        /* LM iterations taken (incl. unsuccessful) */
    int32_t accepted[TSBA_MAX_LEVELS];     /* successful steps */
    int32_t termination[TSBA_MAX_LEVELS];  /* 0 max-iter, 1 function tol, 2 parameter tol, 3 gradient tol, 4 radius, 5 failure */
    double  cost0[TSBA_MAX_LEVELS], cost1[TSBA_MAX_LEVELS];
    int64_t n_sblock[TSBA_MAX_LEVELS], n_tblock[TSBA_MAX_LEVELS];   /* residual blocks in the problem */
    int64_t n_resid_evals;                 /* scalar residuals evaluated, one count per LM trial step + linearisation */
    int32_t n_bad_scene[TSBA_MAX_LEVELS], n_bad_tfeat[TSBA_MAX_LEVELS], n_bad_text[TSBA_MAX_LEVELS];
    double  t_upload_ms, t_solve_ms, t_download_ms;
    int32_t cov_valid;                     /* tsba_theta_optim: 1 = cov[] was written, 0 = singular information matrix (cov untouched) */
    /* How the reduced camera system S dx = -g was solved (the reference hands it to Ceres' sparse Cholesky, optimizer.cc:1833-1840, which
     * either solves it or fails the step).  Every path below is an exact solve except TSBA_SOLVER_BAND_PCG, which iterates to a relative
     * tolerance of 1e-10; an iterative solve that does not get there is treated as a FAILED linear solve (the LM loop shrinks the trust
     * region, as Ceres does on LINEAR_SOLVER_FAILURE) and counted in pcg_unconverged -- an unconverged step is never accepted. */
    int32_t solver_path;                   /* TSBA_SOLVER_* of the last pass */
    int32_t pcg_iterations;                /* TSBA_SOLVER_BAND_PCG / _BAND_LOWRANK: conjugate-gradient iterations over all LM trials */
    int32_t pcg_systems;                   /* reduced systems solved iteratively (LM trials) */
    int32_t pcg_max_iterations;            /* most iterations one system took */
    int32_t pcg_unconverged;               /* systems that hit the iteration cap: their LM trial was rejected as an invalid step */
    int32_t pcg_stagnated;                 /* systems that ended at the attainable accuracy (rounding noise of M^-1 r) short of the tolerance */
    int32_t poll_timeouts;                 /* threads of kernels whose workgroups hand results over inside one launch (k_solve_back, the separator tree of the solve
                                            * phase) that gave up waiting: each fails the linear solve of its LM trial (the trial is rejected as an invalid step, as
                                            * LINEAR_SOLVER_FAILURE in Ceres) -- 0 unless the device was kept from running the launch's workgroups for tens of ms */
    int32_t reserved_[2];
} tsba_report;
#define TSBA_SOLVER_LDS          0   /* window of <= 31 keyframes: blocked LDL^T in one workgroup's LDS */
#define TSBA_SOLVER_DENSE        1   /* multi-workgroup blocked Cholesky on the dense / wide-band matrix */
#define TSBA_SOLVER_BAND         2   /* streaming band solver, one workgroup */
#define TSBA_SOLVER_BAND_PART    3   /* partitioned band solver, separator system sequential */
#define TSBA_SOLVER_BAND_CR      4   /* partitioned band solver, separator system by block cyclic reduction */
#define TSBA_SOLVER_RING         5   /* as 4 on a ring (one loop closure): ghost rows, merged root */
#define TSBA_SOLVER_BAND_LOWRANK 6   /* band part + exact low-rank correction for a few loop closures (one or two refinement iterations) */
#define TSBA_SOLVER_BAND_PCG     7   /* band part as preconditioner of conjugate gradients (scattered long-range observations) */
#define TSBA_SOLVER_POSE         8   /* one free pose: 6x6 in registers (tsba_pose_optim) */

/* Reference defaults for the three public methods. */
void tsba_default_options_local (tsba_options *o);   /* levels 2,1,0 x10, chi2 12.25 / .5 .5 .5(.95 at 0) */
void tsba_default_options_pose  (tsba_options *o);
void tsba_default_options_global(tsba_options *o);   /* level 0, 20 its, unweighted, scene only */
/* The remaining optimizer:: entry points are the same solver with other constants (all poses constant = mark every
 * keyframe in kf_initial; "no loss" = a huge Huber delta):
 *   init        optimizer::InitBA -> PyrIniBA (optimizer.cc:960-1056): 2 KFs, the host at identity constant (kf_initial = {1,0}),
 *               auto_IniBAScene / nume_IniBAText = unweighted R1 / R6, Huber 3 / 3, levels 3,2,1,0 x 10, no flags, no outlier pass
 *   landmarker  optimizer::OptimizeLandmarker -> PyrLandmarkers (:456-562,1853-2168): poses constant, auto_RhoScene / nume_thetaText
 *               = unweighted R1 / R6 with only rho / theta free, Huber sqrt(5.991) / 2, levels 3,2,1,0 x 50, scene outlier pass chi2 18
 *   theta       optimizer::ThetaOptimMultiFs -> PyrThetaOptim (:565-624,2170-2242): theta of ONE plane, poses constant, no loss,
 *               levels 2,1,0 x 50 (Ceres default), then the 3x3 covariance of theta */
void tsba_default_options_init      (tsba_options *o);
void tsba_default_options_landmarker(tsba_options *o);
void tsba_default_options_theta     (tsba_options *o);

/* ---- context ---- */
/* The layout of tsba_problem / tsba_options / tsba_report this header describes.  A caller built against another header must not hand its structs to the
 * library: compare tsba_abi_version() with TSBA_ABI_VERSION once at start-up (adapter/tsba_gather.hpp does).  Bumped whenever a struct changes size or a
 * field changes meaning (round 5: 5 -- tsba_report.poll_timeouts took one of the reserved words, the size is unchanged). */
#define TSBA_ABI_VERSION 5
int  tsba_abi_version(void);
int  tsba_create (void **ctx, int device);   /* TSBA_ERR_DEVICE if no gfx950 device is usable */
int  tsba_destroy(void *ctx);
const char *tsba_last_error(void *ctx);

/* ---- one-shot calls (upload + solve + download), the optimizer:: replacements ---- */
int  tsba_local_ba  (void *ctx, tsba_problem *p, const tsba_options *o, tsba_report *r);
int  tsba_pose_optim(void *ctx, tsba_problem *p, const tsba_options *o, tsba_report *r);  /* n_kf==1, all landmarks frozen */
int  tsba_global_ba (void *ctx, tsba_problem *p, const tsba_options *o, tsba_report *r);
/* optimizer::ThetaOptimMultiFs: solve (normally with tsba_default_options_theta) and return the covariance of theta[text]
 * = (J^T J)^-1 of its residual blocks at the solution (ceres::Covariance, optimizer.cc:2219-2238), row-major 3x3.
 * A singular information matrix is NOT an error, as in the reference: PyrThetaOptim returns true and leaves thetaVariance
 * as it was when Covariance::Compute fails (optimizer.cc:2224-2241) -- the call returns TSBA_OK, cov[] is left untouched and
 * r->cov_valid = 0 (1 when cov[] was written). */
int  tsba_theta_optim(void *ctx, tsba_problem *p, const tsba_options *o, int text, double cov[9], tsba_report *r);

/* Text label image of keyframe `kf` at pyramid level `level` for the state left by the last solve on this context (one-shot
 * entry points leave it too).  Replaces the label part of optimizer::ShowBAReproj_TextBox (src/optimizer.cc:2508-2582 ->
 * tool::TextBoxWithFill / GetTextLabelMask, src/tool.cc:2103-2166): background -1, then every text observation of the keyframe,
 * in the order of tobs_*, fills its projected quad (cv::Point truncation, cv::fillPoly scan conversion) with its rank among the
 * keyframe's observations; later quads overwrite earlier ones.  out: img_h[level] * img_w[level] floats (the CV_32F Mat that
 * optimizer::UpdateTrackedTextBA, optimizer.cc:2246-2386, reads).  The drawing of box outlines / ids stays in the host. */
int  tsba_text_label_image(void *ctx, int kf, int level, float *out);

/* ---- staged calls (bench / repeated solves with the problem resident in HBM) ---- */
int  tsba_upload  (void *ctx, const tsba_problem *p, const tsba_options *o);
int  tsba_solve   (void *ctx, tsba_report *r);        /* restarts from the uploaded parameters every call */
int  tsba_download(void *ctx, tsba_problem *p);       /* parameters + good flags of the last solve */

/* ---- test hooks ---- */
/* Residuals (+ optional Jacobians) of every residual block of pyramid level `level`, in the
 * reference's block order (scene blocks first, optimizer.cc:1609-1612).  Blocks are the ones the
 * reference would add (good-flag filtering per options.filter_good).
 *   resid    [2*ns + 8*nt]             raw residuals (no loss applied)
 *   jac      scene block: 2 x 13 row-major  (target d_rot3,t3 | host d_rot3,t3 | rho)   tangent space of
 *            text  block: 8 x 15 row-major  (target 6 | host 6 | theta3)                 ceres::QuaternionParameterization
 *            frozen-host blocks fill only the target 6 columns (others 0).  May be NULL.
 *   musigma  [n_tobs][2] mu, sigma used (may be NULL)
 *   ns, nt   out: number of scene / text blocks
 */
int  tsba_eval(void *ctx, const tsba_problem *p, const tsba_options *o, int level,
               double *resid, double *jac, double *musigma, int64_t *ns, int64_t *nt);

/* Test hooks that look inside a solve (reduced system, solver-path switches, kernel timing) are declared in tsba_debug.h; nothing of
 * the reference's surface needs them. */

/* ---- multi-GPU: RCCL communicator for tsba_global_ba (one process per GPU) ---- */
/* One process per GPU.  Rank 0 calls tsba_comm_unique_id and broadcasts the 128 bytes (e.g. torch.distributed); every rank
 * then calls tsba_comm_init.  Afterwards tsba_upload keeps only this rank's landmarks (those hosted in its keyframe range, all
 * poses replicated) and tsba_solve all-reduces the reduced normal equations S (its band), g and a few scalars once per LM trial.
 * id128 == NULL selects the split (multi-GPU) kernel sequence without a communicator (single-process test hook). */
int  tsba_comm_unique_id(void *ctx, void *id128);
/* Loads RCCL into the process (once; tsba_comm_unique_id / tsba_comm_init do it themselves otherwise).  Call it at start-up, BEFORE other host threads launch kernels, in a process
 * that will use a communicator: loading a HIP library registers its code objects with the runtime, and a launch on another thread at that moment is not safe (observed beside a looping
 * extractor and a looping bundle adjustment: "invalid device function", a segmentation fault inside the other thread's launch).  TSBA_OK, or TSBA_ERR_COMM when no librccl is found. */
int  tsba_comm_load(void);
int  tsba_comm_init(void *ctx, const void *id128, int rank, int world);
/* What the last solve handed to collectives on this rank: ranks = size of the communicator (ncclCommCount; 1 without one),
 * bytes[0] per LM trial (reduced normal equations + the sums of the speculative linearisation), bytes[1] per linearisation,
 * bytes[2] per pass set-up. */
int  tsba_comm_stats(void *ctx, int32_t *ranks, int64_t bytes[3]);
/* The same rank / world protocol WITHOUT RCCL, for `world` contexts of ONE process (one host thread per context, on the same or on
 * different GPUs): the collectives go through host memory, summed in rank order.  Test hook: the N > 1 code path on a one-GPU box.
 * group: from tsba_local_group_create(world); every member thread calls tsba_comm_init_local, then upload / solve in lockstep. */
void *tsba_local_group_create(int world);
void  tsba_local_group_destroy(void *group);
int  tsba_comm_init_local(void *ctx, void *group, int rank, int world);

#ifdef __cplusplus
}
#endif
#endif /* TSBA_H */
