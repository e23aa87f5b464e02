/* tsloop.h -- C ABI of the loop-closure optimisers (libtsloop.so, gfx950).  SURVEY.md 8f rank 4.
 *
 *   optimizer::OptimizeSim3   /root/reference/src/optimizer.cc:626-731  (auto_sim.h, auto_siminv.h)   -> tsloop_optimize_sim3
 *   optimizer::OptimizeLoop   /root/reference/src/optimizer.cc:733-957  (numer_loop_ver2.h, ModelTool.hpp:354-432 logSim3) -> tsloop_optimize_loop
 *
 * Same Levenberg-Marquardt semantics as the BA library (Ceres 1.x TrustRegionMinimizer + LevenbergMarquardtStrategy with Jacobi
 * scaling), fp64.  All functions return 0 on success, a negative TSLOOP_ERR_* otherwise. */
#ifndef TSLOOP_H
#define TSLOOP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TSLOOP_OK 0
#define TSLOOP_ERR_ARG (-1)
#define TSLOOP_ERR_DEVICE (-2)
#define TSLOOP_ERR_NUMERIC (-3)

typedef struct tsloop_options {
    int32_t max_it;                    /* options.max_num_iterations = 20, optimizer.cc:676 */
    int32_t pad;
    double  huber_delta;               /* HuberLoss(sqrt(10)), optimizer.cc:661 */
    double  thresh_outlier;            /* 4.0 px, optimizer.cc:629 */
    /* Ceres 1.x defaults */
    double  initial_radius, max_radius, min_radius, min_relative_decrease;
    double  function_tolerance, gradient_tolerance, parameter_tolerance, min_diagonal, max_diagonal;
} tsloop_options;

typedef struct tsloop_report {
    int32_t iters, accepted, termination;   /* termination: 0 max-iter, 1 function tol, 2 parameter tol, 3 gradient tol, 4 radius, 5 failure */
    int32_t n_inlier;                       /* return value of OptimizeSim3 */
    double  cost0, cost1;
    double  t_ms;
} tsloop_report;

/* Sim3 between two keyframes from n 3D-2D matches in both directions (optimizer::OptimizeSim3).
 * P1 / P2: vFeat1[i].posObv / vFeat2[i].posObv (the matched point in camera-1 / camera-2 coordinates), [n][3];
 * uv1 / uv2: vFeat1[i].obv2d.pt / vFeat2[i].obv2d.pt, [n][2] float;  inlier: vbInliers, [n], in/out;
 * sim: Sim12 = (qw qx qy qz | t | s), in/out (q is normalised on entry as the reference does). */
typedef struct tsloop_sim3_problem {
    int32_t n, pad;
    const double *P1, *P2;
    const float  *uv1, *uv2;
    uint8_t *inlier;
    double K[4];                            /* fx fy cx cy (K1 = K2 = K, optimizer.cc:633-634) */
    double sim[8];
} tsloop_sim3_problem;

/* Sim3 pose graph over the keyframes of the map (optimizer::OptimizeLoop).
 * pose: [n_kf][8] = (qw qx qy qz | t | s) per keyframe, the initial values of optimizer.cc:745-778 (vScwIni), in/out;
 * fixed: [n_kf], 1 = SetParameterBlockConstant (keyframes 0, 1 and the loop keyframe, :861-869);
 * one residual block per connection e: keyframes (edge_i[e], edge_j[e]) in AddResidualBlock order and the measured Sji = meas[e]
 * (q | t | s), both the normal (:788-820) and the loop (:823-858) connections.  The map update that follows the solve in the
 * reference (SetPose, rho *= s, theta *= s, :884-956) stays with the caller. */
typedef struct tsloop_graph_problem {
    int32_t n_kf, n_edge;
    double *pose;
    const uint8_t *fixed;
    const int32_t *edge_i, *edge_j;
    const double *meas;
} tsloop_graph_problem;

void tsloop_default_options_sim3(tsloop_options *o);
void tsloop_default_options_loop(tsloop_options *o);    /* 20 iterations, no loss (huber_delta / thresh_outlier unused) */
int  tsloop_create(int device, void **ctx);          /* TSLOOP_ERR_DEVICE without a usable GPU: there is no CPU path */
void tsloop_destroy(void *ctx);
const char *tsloop_last_error(void *ctx);
int  tsloop_optimize_sim3(void *ctx, tsloop_sim3_problem *p, const tsloop_options *o, tsloop_report *r);
int  tsloop_optimize_loop(void *ctx, tsloop_graph_problem *p, const tsloop_options *o, tsloop_report *r);   /* r->n_inlier unused */

#ifdef __cplusplus
}
#endif
#endif
