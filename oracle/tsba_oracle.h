/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU oracle for the TextSLAM bundle-adjustment hot path: a plain-C, fp64, single-threaded
 * restatement of the reference algorithm (cost functors + the Ceres-1.x Levenberg-Marquardt
 * behaviour the reference relies on).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (textslam_amd/) never does.
 *
 * PARITY UNPINNED: the reference ships no tests / golden vectors for this path and its numerics
 * live in un-vendored Ceres / Eigen / OpenCV that do not exist in the build container, so this
 * restatement could not be checked against the reference binary.  Ceres / OpenCV behaviours
 * encoded here are recalled from their published sources and cited inline.
 */
#ifndef TSBA_ORACLE_H
#define TSBA_ORACLE_H
#include "../include/tsba.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Same contract as tsba_eval (include/tsba.h).  options.text_jacobian selects analytic (0) or the
 * reference's Ceres CENTRAL numeric differentiation (1) for text blocks. */
int tsba_oracle_eval(const tsba_problem *p, const tsba_options *o, int level,
                     double *resid, double *jac, double *musigma, int64_t *ns, int64_t *nt);

/* Full solve (all pyramid passes, outlier passes, write-back into p) -- restates
 * LocalBundleAdjustment / PoseOptim / GlobalBA depending on the options. */
int tsba_oracle_solve(tsba_problem *p, const tsba_options *o, tsba_report *r);

/* mu / sigma of the uint8 intensities inside a projected quad: tool::CalTextinfo + CalStatistics
 * (src/tool.cc:1178-1262) with cv::fillPoly's scan conversion.  corners = 4 x (u,v).
 * Returns 1 if (mu,sigma) valid and sigma != 0, else 0 (sigma is set to 0). */
int tsba_oracle_musigma(const uint8_t *img, int w, int h, const double *corners, double *mu, double *sigma);

/* Rasterised mask of cv::fillPoly (boundary lines + scanline interior) for integer vertices; mask[h*w] in {0,1}. */
void tsba_oracle_fillpoly4(int w, int h, const int *xy, uint8_t *mask);

/* Reduced camera system of the first linearisation of a pass (debug aid for the HIP path):
 *   free_idx [n_kf]       out: column-block index of each KF in S, or -1 (fixed / not participating)
 *   S        [(6nf)^2]    out: Schur complement INCLUDING the LM damping for `radius` (row-major)
 *   g        [6nf]        out: reduced gradient  (b_p - W V^-1 b_l), sign convention: S * dx = -g
 *   Hpp/bp   [(6nf)^2],[6nf] out: undamped pose block and pose gradient (may be NULL)
 *   cost     out: 1/2 sum rho(|r|^2) over non-fixed blocks
 * Returns nf (>=0) or negative error. */
int tsba_oracle_reduced_system(const tsba_problem *p, const tsba_options *o, int level, double radius,
                               int32_t *free_idx, double *S, double *g, double *Hpp, double *bp, double *cost);

/* One rank's contribution to the reduced normal equations of a landmark-sharded global BA (see tsba_oracle.c). */
int tsba_oracle_partial_system(const tsba_problem *p, const tsba_options *o, int level, double radius,
                               int32_t *free_idx, double *S, double *g, double *Hd, double *cost);

/* Covariance of theta[text] (all other parameters constant) at the current parameters, ceres::Covariance semantics. */
/* text label image of a keyframe (optimizer::ShowBAReproj_TextBox -> tool::TextBoxWithFill); out: h*w floats */
int tsba_oracle_label_image(const tsba_problem *p, int kf, int level, float *out);

int tsba_oracle_theta_cov(const tsba_problem *p, const tsba_options *o, int level, int text, double cov[9]);

int tsba_oracle_theta_optim(tsba_problem *p, const tsba_options *o, tsba_report *r, int text, double cov[9]);

/* Reference option sets: kind 0 = LocalBundleAdjustment, 1 = PoseOptim, 2 = GlobalBA. */
void tsba_oracle_default_options(tsba_options *o, int kind);

#ifdef __cplusplus
}
#endif
#endif
