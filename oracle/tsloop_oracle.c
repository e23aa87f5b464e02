/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  PARITY UNPINNED (Ceres / Eigen are not in this image; the LM constants and the
 * trust-region loop are the Ceres 1.x behaviour recalled in SURVEY.md 8c, as in oracle/tsba_oracle.c).
 *
 * Plain-C restatement of optimizer::OptimizeSim3 (src/optimizer.cc:626-731): the cost functors auto_sim (include/auto_sim.h:28-58)
 * and auto_siminv (include/auto_siminv.h:28-66) over the parameter blocks q (4, ceres::QuaternionParameterization), t (3), s (1),
 * one shared HuberLoss(sqrt(10)), Ceres LM with max 20 iterations, then the 4-pixel inlier test.
 * Derivatives are formed the way Ceres' AutoDiff + LocalParameterization does: ambient Jacobian d f / d (q, t, s) of the functor
 * (ceres::QuaternionRotatePoint normalises q, rotation.h:525-562) times the 4x3 plus-Jacobian -- a different route from the HIP
 * kernel's closed tangent-space forms, so that the two check each other. */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "../include/tsloop.h"

static void quat_plus_jacobian(const double x[4], double J[12]) {      /* ceres::QuaternionParameterization::ComputeJacobian, 4x3 */
    J[0] = -x[1]; J[1]  = -x[2]; J[2]  = -x[3];
    J[3] =  x[0]; J[4]  =  x[3]; J[5]  = -x[2];
    J[6] = -x[3]; J[7]  =  x[0]; J[8]  =  x[1];
    J[9] =  x[2]; J[10] = -x[1]; J[11] =  x[0];
}
static void quat_plus(const double x[4], const double d[3], double o[4]) {   /* ceres::QuaternionParameterization::Plus */
    double nd = sqrt(d[0]*d[0] + d[1]*d[1] + d[2]*d[2]);
    if (nd > 0.0) {
        double s = sin(nd)/nd;
        double z[4] = { cos(nd), s*d[0], s*d[1], s*d[2] };
        o[0] = z[0]*x[0] - z[1]*x[1] - z[2]*x[2] - z[3]*x[3];
        o[1] = z[0]*x[1] + z[1]*x[0] + z[2]*x[3] - z[3]*x[2];
        o[2] = z[0]*x[2] - z[1]*x[3] + z[2]*x[0] + z[3]*x[1];
        o[3] = z[0]*x[3] + z[1]*x[2] - z[2]*x[1] + z[3]*x[0];
    } else { o[0] = x[0]; o[1] = x[1]; o[2] = x[2]; o[3] = x[3]; }
}
/* R(q / |q|) p and its derivative with respect to the four (unnormalised) quaternion components: 3x4, row-major */
static void rotate_with_jac(const double q_[4], const double p[3], double out[3], double J[12]) {
    const double n = sqrt(q_[0]*q_[0] + q_[1]*q_[1] + q_[2]*q_[2] + q_[3]*q_[3]);
    const double w = q_[0]/n, x = q_[1]/n, y = q_[2]/n, z = q_[3]/n;
    const double R[9] = { 1 - 2*(y*y + z*z), 2*(x*y - w*z), 2*(x*z + w*y),
                          2*(x*y + w*z), 1 - 2*(x*x + z*z), 2*(y*z - w*x),
                          2*(x*z - w*y), 2*(y*z + w*x), 1 - 2*(x*x + y*y) };
    for (int i = 0; i < 3; i++) out[i] = R[3*i]*p[0] + R[3*i+1]*p[1] + R[3*i+2]*p[2];
    /* d (R(u) p) / d u for the unit quaternion u = (w, x, y, z) */
    double Ju[12];
    Ju[0] = 2*(-z*p[1] + y*p[2]);  Ju[1] = 2*(y*p[1] + z*p[2]);             Ju[2]  = 2*(-2*y*p[0] + x*p[1] + w*p[2]); Ju[3]  = 2*(-2*z*p[0] - w*p[1] + x*p[2]);
    Ju[4] = 2*(z*p[0] - x*p[2]);   Ju[5] = 2*(y*p[0] - 2*x*p[1] - w*p[2]);  Ju[6]  = 2*(x*p[0] + z*p[2]);             Ju[7]  = 2*(w*p[0] - 2*z*p[1] + y*p[2]);
    Ju[8] = 2*(-y*p[0] + x*p[1]);  Ju[9] = 2*(z*p[0] + w*p[1] - 2*x*p[2]);  Ju[10] = 2*(-w*p[0] + z*p[1] - 2*y*p[2]); Ju[11] = 2*(x*p[0] + y*p[1]);
    /* chain through u = q / |q|: du/dq = (I - u u^T) / |q| */
    const double u[4] = { w, x, y, z };
    for (int i = 0; i < 3; i++) {
        double dot = 0; for (int k = 0; k < 4; k++) dot += Ju[4*i + k]*u[k];
        for (int k = 0; k < 4; k++) J[4*i + k] = (Ju[4*i + k] - dot*u[k])/n;
    }
}

/* one match: residuals r[4] = (auto_sim, auto_siminv) and ambient Jacobians J[4][8] over (q, t, s) */
static void sim3_match(const double x[8], const double P1[3], const double P2[3], const float uv1[2], const float uv2[2], const double K[4],
                       double r[4], double J[32]) {
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3], s = x[7];
    memset(J, 0, sizeof(double)*32);
    {   /* auto_sim: P1' = s R(q) P2 + t */
        double A[3], JA[12]; rotate_with_jac(x, P2, A, JA);
        const double X = s*A[0] + x[4], Y = s*A[1] + x[5], Z = s*A[2] + x[6];
        r[0] = X/Z*fx + cx - (double)uv1[0]; r[1] = Y/Z*fy + cy - (double)uv1[1];
        const double dX[2][3] = { { fx/Z, 0, -fx*X/(Z*Z) }, { 0, fy/Z, -fy*Y/(Z*Z) } };
        for (int a = 0; a < 2; a++) {
            for (int k = 0; k < 4; k++) J[8*a + k] = s*(dX[a][0]*JA[k] + dX[a][1]*JA[4 + k] + dX[a][2]*JA[8 + k]);
            for (int k = 0; k < 3; k++) J[8*a + 4 + k] = dX[a][k];
            J[8*a + 7] = dX[a][0]*A[0] + dX[a][1]*A[1] + dX[a][2]*A[2];
        }
    }
    {   /* auto_siminv: P2' = R(q)^-1 P1 / s - R(q)^-1 t / s */
        const double qi[4] = { x[0], -x[1], -x[2], -x[3] };
        double A[3], JA[12], B[3], JB[12];
        rotate_with_jac(qi, P1, A, JA); rotate_with_jac(qi, x + 4, B, JB);
        const double X = A[0]/s - B[0]/s, Y = A[1]/s - B[1]/s, Z = A[2]/s - B[2]/s;
        r[2] = X/Z*fx + cx - (double)uv2[0]; r[3] = Y/Z*fy + cy - (double)uv2[1];
        const double dX[2][3] = { { fx/Z, 0, -fx*X/(Z*Z) }, { 0, fy/Z, -fy*Y/(Z*Z) } };
        /* R(qinv) as a function of t: d B / d t = R(qinv) (columns = rotate the unit vectors) */
        double Rt[9];
        for (int k = 0; k < 3; k++) { double e[3] = { 0, 0, 0 }, o[3], jj[12]; e[k] = 1; rotate_with_jac(qi, e, o, jj); Rt[k] = o[0]; Rt[3 + k] = o[1]; Rt[6 + k] = o[2]; }
        for (int a = 0; a < 2; a++) {
            for (int k = 0; k < 4; k++) {
                const double sg = k == 0 ? 1.0 : -1.0;              /* d qinv / d q = diag(1, -1, -1, -1) */
                double v = 0; for (int i = 0; i < 3; i++) v += dX[a][i]*(JA[4*i + k] - JB[4*i + k])/s;
                J[8*(2 + a) + k] = sg*v;
            }
            for (int k = 0; k < 3; k++) { double v = 0; for (int i = 0; i < 3; i++) v += dX[a][i]*(-Rt[3*i + k]/s); J[8*(2 + a) + 4 + k] = v; }
            J[8*(2 + a) + 7] = -(dX[a][0]*X + dX[a][1]*Y + dX[a][2]*Z)/s;
        }
    }
}

/* ceres::HuberLoss + Corrector (rho'' <= 0: scale residual and Jacobian by sqrt(rho')) */
static double huber(double s, double delta, double *scale) {
    double b = delta*delta;
    if (s > b) { double r = sqrt(s); double rho1 = delta/r; if (rho1 < DBL_MIN) rho1 = DBL_MIN; *scale = sqrt(rho1); return 2.0*delta*r - b; }
    *scale = 1.0; return s;
}
static int chol_solve_dense(double *A, int n, double *b) {        /* lower Cholesky in place, then solve; 0 ok */
    for (int j = 0; j < n; j++) {
        double d = A[j*n + j]; for (int k = 0; k < j; k++) d -= A[j*n + k]*A[j*n + k];
        if (!(d > 0.0)) return 1;
        d = sqrt(d); A[j*n + j] = d;
        for (int i = j + 1; i < n; i++) { double v = A[i*n + j]; for (int k = 0; k < j; k++) v -= A[i*n + k]*A[j*n + k]; A[i*n + j] = v/d; }
    }
    for (int i = 0; i < n; i++) { double v = b[i]; for (int k = 0; k < i; k++) v -= A[i*n + k]*b[k]; b[i] = v/A[i*n + i]; }
    for (int i = n - 1; i >= 0; i--) { double v = b[i]; for (int k = i + 1; k < n; k++) v -= A[k*n + i]*b[k]; b[i] = v/A[i*n + i]; }
    return 0;
}

typedef struct { const tsloop_sim3_problem *p; double delta; } sim3_ctx;
/* robustified normal equations in the 7-dim tangent space: H = sum w J^T J, g = sum w J^T r, cost = sum rho / 2 */
static void sim3_linearize(const sim3_ctx *c, const double x[8], double H[49], double g[7], double *cost) {
    const tsloop_sim3_problem *p = c->p;
    memset(H, 0, sizeof(double)*49); memset(g, 0, sizeof(double)*7); *cost = 0;
    double Jp[12]; quat_plus_jacobian(x, Jp);
    for (int i = 0; i < p->n; i++) {
        if (!p->inlier[i]) continue;
        double r[4], Ja[32], Jl[28];
        sim3_match(x, p->P1 + 3*i, p->P2 + 3*i, p->uv1 + 2*i, p->uv2 + 2*i, p->K, r, Ja);
        for (int a = 0; a < 4; a++) {
            for (int k = 0; k < 3; k++) { double v = 0; for (int m = 0; m < 4; m++) v += Ja[8*a + m]*Jp[3*m + k]; Jl[7*a + k] = v; }
            for (int k = 0; k < 4; k++) Jl[7*a + 3 + k] = Ja[8*a + 4 + k];
        }
        for (int b = 0; b < 2; b++) {                               /* two residual blocks per match, the loss is applied per block */
            double sc; const double s2 = r[2*b]*r[2*b] + r[2*b+1]*r[2*b+1];
            *cost += 0.5*huber(s2, c->delta, &sc);
            for (int a = 2*b; a < 2*b + 2; a++) {
                const double ra = r[a]*sc;
                for (int k = 0; k < 7; k++) { const double jk = Jl[7*a + k]*sc; g[k] += jk*ra; for (int m = 0; m < 7; m++) H[7*k + m] += jk*Jl[7*a + m]*sc; }
            }
        }
    }
}
static double sim3_cost(const sim3_ctx *c, const double x[8]) {
    const tsloop_sim3_problem *p = c->p; double cost = 0;
    for (int i = 0; i < p->n; i++) {
        if (!p->inlier[i]) continue;
        double r[4], Ja[32], sc;
        sim3_match(x, p->P1 + 3*i, p->P2 + 3*i, p->uv1 + 2*i, p->uv2 + 2*i, p->K, r, Ja);
        cost += 0.5*huber(r[0]*r[0] + r[1]*r[1], c->delta, &sc) + 0.5*huber(r[2]*r[2] + r[3]*r[3], c->delta, &sc);
    }
    return cost;
}
static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

void tsloop_oracle_default_options_sim3(tsloop_options *o) {
    memset(o, 0, sizeof(*o));
    o->max_it = 20; o->huber_delta = sqrt(10.0); o->thresh_outlier = 4.0;
    o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32; o->min_relative_decrease = 1e-3;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8; o->min_diagonal = 1e-6; o->max_diagonal = 1e32;
}

int tsloop_oracle_optimize_sim3(tsloop_sim3_problem *p, const tsloop_options *o, tsloop_report *rep) {
    sim3_ctx c = { p, o->huber_delta };
    double x[8], cand[8];
    {   const double *q = p->sim; const double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);    /* q = q.normalized(), optimizer.cc:639 */
        for (int k = 0; k < 4; k++) x[k] = q[k]/n;
        for (int k = 4; k < 8; k++) x[k] = p->sim[k]; }
    memset(rep, 0, sizeof(*rep));
    double H[49], g[7], sc[7], dg[7], x_cost;
    sim3_linearize(&c, x, H, g, &x_cost); rep->cost0 = x_cost;
    for (int k = 0; k < 7; k++) sc[k] = 1.0/(1.0 + sqrt(H[8*k]));
    double x_norm = 0; for (int k = 0; k < 8; k++) x_norm += x[k]*x[k]; x_norm = sqrt(x_norm);
    double radius = o->initial_radius, decrease_factor = 2.0; int invalid = 0, term = 0, it = 0, accepted = 0;
    double gmax = 0; for (int k = 0; k < 7; k++) if (fabs(g[k]) > gmax) gmax = fabs(g[k]);
    int nact = 0; for (int i = 0; i < p->n; i++) nact += p->inlier[i] ? 1 : 0;
    if (nact == 0) { term = 5; goto done; }
    if (gmax <= o->gradient_tolerance) { term = 3; goto done; }
    while (1) {
        if (it >= o->max_it) { term = 0; break; }
        if (radius < o->min_radius) { term = 4; break; }
        it++;
        double A[49], y[7], d[7];
        for (int k = 0; k < 7; k++) dg[k] = clampd(sc[k]*sc[k]*H[8*k], o->min_diagonal, o->max_diagonal);
        for (int k = 0; k < 7; k++) { for (int m = 0; m < 7; m++) A[7*k + m] = sc[k]*H[7*k + m]*sc[m]; A[8*k] += dg[k]/radius; y[k] = -sc[k]*g[k]; }
        const int rc = chol_solve_dense(A, 7, y);
        double model_change = -1;
        if (!rc) {
            for (int k = 0; k < 7; k++) d[k] = sc[k]*y[k];
            model_change = 0;
            for (int k = 0; k < 7; k++) { double hd = 0; for (int m = 0; m < 7; m++) hd += H[7*k + m]*d[m]; model_change -= d[k]*(g[k] + 0.5*hd); }
        }
        if (rc || !(model_change > 0)) { if (++invalid >= 5) { term = 5; break; } radius *= 0.5; continue; }
        invalid = 0;
        quat_plus(x, d, cand); for (int k = 0; k < 4; k++) cand[4 + k] = x[4 + k] + d[3 + k];
        double c_cost = sim3_cost(&c, cand); if (!(c_cost == c_cost)) c_cost = DBL_MAX;
        double step = 0; for (int k = 0; k < 8; k++) step += (cand[k] - x[k])*(cand[k] - x[k]); step = sqrt(step);
        if (step <= o->parameter_tolerance*(x_norm + o->parameter_tolerance)) { term = 2; break; }
        const double cost_change = x_cost - c_cost;
        if (fabs(cost_change) <= o->function_tolerance*x_cost) { term = 1; break; }
        const double rel = cost_change/model_change;
        if (rel > o->min_relative_decrease) {
            memcpy(x, cand, sizeof(x)); accepted++;
            x_norm = 0; for (int k = 0; k < 8; k++) x_norm += x[k]*x[k]; x_norm = sqrt(x_norm);
            sim3_linearize(&c, x, H, g, &x_cost);
            double t = 2.0*rel - 1.0, f = 1.0 - t*t*t; if (f < 1.0/3.0) f = 1.0/3.0;
            radius = radius/f; if (radius > o->max_radius) radius = o->max_radius;
            decrease_factor = 2.0;
            gmax = 0; for (int k = 0; k < 7; k++) if (fabs(g[k]) > gmax) gmax = fabs(g[k]);
            if (gmax <= o->gradient_tolerance) { term = 3; break; }
        } else { radius = radius/decrease_factor; decrease_factor *= 2.0; }
    }
done:
    rep->iters = it; rep->accepted = accepted; rep->termination = term; rep->cost1 = x_cost;
    /* result + inlier check, optimizer.cc:682-729: q12.normalized(), T12 = [s R | t], T21 = T12^-1 */
    {   const double n = sqrt(x[0]*x[0] + x[1]*x[1] + x[2]*x[2] + x[3]*x[3]);
        for (int k = 0; k < 4; k++) p->sim[k] = x[k]/n;
        for (int k = 4; k < 8; k++) p->sim[k] = x[k]; }
    const double *q = p->sim, s = p->sim[7], *t = p->sim + 4;
    const double w = q[0], qx = q[1], qy = q[2], qz = q[3];
    const double R[9] = { 1 - 2*(qy*qy + qz*qz), 2*(qx*qy - w*qz), 2*(qx*qz + w*qy), 2*(qx*qy + w*qz), 1 - 2*(qx*qx + qz*qz), 2*(qy*qz - w*qx),
                          2*(qx*qz - w*qy), 2*(qy*qz + w*qx), 1 - 2*(qx*qx + qy*qy) };
    int ninl = 0;
    for (int i = 0; i < p->n; i++) {
        if (!p->inlier[i]) continue;
        const double *P2 = p->P2 + 3*i, *P1 = p->P1 + 3*i;
        double X[3], Y[3], D[3] = { P1[0] - t[0], P1[1] - t[1], P1[2] - t[2] };
        for (int a = 0; a < 3; a++) X[a] = s*(R[3*a]*P2[0] + R[3*a+1]*P2[1] + R[3*a+2]*P2[2]) + t[a];
        for (int a = 0; a < 3; a++) Y[a] = (R[a]*D[0] + R[3 + a]*D[1] + R[6 + a]*D[2])/s;               /* T21 = [R^T / s | -R^T t / s] */
        const double e1x = p->K[0]*X[0]/X[2] + p->K[2] - (double)p->uv1[2*i], e1y = p->K[1]*X[1]/X[2] + p->K[3] - (double)p->uv1[2*i+1];
        const double e2x = p->K[0]*Y[0]/Y[2] + p->K[2] - (double)p->uv2[2*i], e2y = p->K[1]*Y[1]/Y[2] + p->K[3] - (double)p->uv2[2*i+1];
        if (fabs(e1x) >= o->thresh_outlier || fabs(e1y) >= o->thresh_outlier || fabs(e2x) >= o->thresh_outlier || fabs(e2y) >= o->thresh_outlier) { p->inlier[i] = 0; continue; }
        ninl++;
    }
    rep->n_inlier = ninl;
    return term == 5 ? TSLOOP_ERR_NUMERIC : TSLOOP_OK;
}

/* test hook: residuals and tangent-space Jacobian (4 x 7) of one match */
void tsloop_oracle_sim3_eval(const double x[8], const double P1[3], const double P2[3], const float uv1[2], const float uv2[2], const double K[4],
                             double r[4], double Jl[28]) {
    double Ja[32], Jp[12]; quat_plus_jacobian(x, Jp);
    sim3_match(x, P1, P2, uv1, uv2, K, r, Ja);
    for (int a = 0; a < 4; a++) {
        for (int k = 0; k < 3; k++) { double v = 0; for (int m = 0; m < 4; m++) v += Ja[8*a + m]*Jp[3*m + k]; Jl[7*a + k] = v; }
        for (int k = 0; k < 4; k++) Jl[7*a + 3 + k] = Ja[8*a + 4 + k];
    }
}
void tsloop_oracle_quat_plus(const double x[4], const double d[3], double o[4]) { quat_plus(x, d, o); }

/* ================================================================================================================================
 * optimizer::OptimizeLoop (src/optimizer.cc:733-957): Sim3 pose graph over all keyframes.  Parameter blocks per keyframe q (4,
 * QuaternionParameterization), t (3), s (1); one residual block per (i, j) connection = numer_loop_ver2 (include/numer_loop_ver2.h:
 * 22-72): logSim3( S_ji(measured) * S_i * S_j^-1 ) (include/ModelTool.hpp:354-432), NumericDiffCostFunction<CENTRAL, 7, 4,3,1, 4,3,1>,
 * no loss; keyframes 0, 1 and the loop keyframe constant (:861-869); LM, 20 iterations (:871-877).  The map update that follows
 * (:884-956: SetPose, rho *= s, theta *= s) is host bookkeeping of the caller and not part of this restatement. */
static void q_mul(const double a[4], const double b[4], double o[4]) {           /* Eigen quaternion product, (w, x, y, z) */
    o[0] = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
    o[1] = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
    o[2] = a[0]*b[2] + a[2]*b[0] + a[3]*b[1] - a[1]*b[3];
    o[3] = a[0]*b[3] + a[3]*b[0] + a[1]*b[2] - a[2]*b[1];
}
static void q_rot(const double q[4], const double v[3], double o[3]) {           /* Eigen QuaternionBase::_transformVector */
    double uv[3] = { q[2]*v[2] - q[3]*v[1], q[3]*v[0] - q[1]*v[2], q[1]*v[1] - q[2]*v[0] };
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q[0]*uv[0] + (q[2]*uv[2] - q[3]*uv[1]);
    o[1] = v[1] + q[0]*uv[1] + (q[3]*uv[0] - q[1]*uv[2]);
    o[2] = v[2] + q[0]*uv[2] + (q[1]*uv[1] - q[2]*uv[0]);
}
static void q_norm(const double q[4], double o[4]) { const double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]); for (int k = 0; k < 4; k++) o[k] = q[k]/n; }
static void q_to_R(const double q[4], double R[9]) {                             /* Eigen toRotationMatrix (no normalisation) */
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2*x, ty = 2*y, tz = 2*z, twx = tx*w, twy = ty*w, twz = tz*w, txx = tx*x, txy = ty*x, txz = tz*x, tyy = ty*y, tyz = tz*y, tzz = tz*z;
    R[0] = 1 - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy; R[3] = txy + twz; R[4] = 1 - (txx + tzz); R[5] = tyz - twx; R[6] = txz - twy; R[7] = tyz + twx; R[8] = 1 - (txx + tyy);
}
static int solve3_lu(const double W_[9], const double b_[3], double x[3]) {      /* 3x3 partial-pivot LU (Eigen .lu().solve) */
    double A[9], b[3]; memcpy(A, W_, sizeof(A)); memcpy(b, b_, sizeof(b));
    for (int c = 0; c < 3; c++) {
        int piv = c; for (int r = c + 1; r < 3; r++) if (fabs(A[3*r + c]) > fabs(A[3*piv + c])) piv = r;
        if (piv != c) { for (int k = 0; k < 3; k++) { double t = A[3*c + k]; A[3*c + k] = A[3*piv + k]; A[3*piv + k] = t; } double t = b[c]; b[c] = b[piv]; b[piv] = t; }
        for (int r = c + 1; r < 3; r++) { const double f = A[3*r + c]/A[3*c + c]; for (int k = c; k < 3; k++) A[3*r + k] -= f*A[3*c + k]; b[r] -= f*b[c]; }
    }
    for (int r = 2; r >= 0; r--) { double v = b[r]; for (int k = r + 1; k < 3; k++) v -= A[3*r + k]*x[k]; x[r] = v/A[3*r + r]; }
    return 0;
}
static void log_sim3(const double rq[4], const double t[3], double s, double res[7]) {      /* ModelTool.hpp:354-432 */
    const double sigma = log(s), eps = 0.00001;
    double R[9]; q_to_R(rq, R);
    const double d = 0.5*(R[0] + R[4] + R[8] - 1);
    const double dR[3] = { R[7] - R[5], R[2] - R[6], R[3] - R[1] };               /* deltaR */
    double omega[3], A, B, Cc;
    if (fabs(sigma) < eps) {
        Cc = 1;
        if (d > 1 - eps) { for (int k = 0; k < 3; k++) omega[k] = 0.5*dR[k]; A = 1./2.; B = 1./6.; }
        else { const double th = acos(d), th2 = th*th, f = th/(2*sqrt(1 - d*d)); for (int k = 0; k < 3; k++) omega[k] = f*dR[k];
               A = (1 - cos(th))/th2; B = (th - sin(th))/(th2*th); }
    } else {
        Cc = (s - 1)/sigma;
        if (d > 1 - eps) { const double s2 = sigma*sigma; for (int k = 0; k < 3; k++) omega[k] = 0.5*dR[k];
               A = ((sigma - 1)*s + 1)/s2; B = ((0.5*s2 - sigma + 1)*s)/(s2*sigma); }
        else { const double th = acos(d), f = th/(2*sqrt(1 - d*d)); for (int k = 0; k < 3; k++) omega[k] = f*dR[k];
               const double th2 = th*th, a = s*sin(th), b = s*cos(th), c = th2 + sigma*sigma;
               A = (a*sigma + (1 - b)*th)/(th*c); B = (Cc - ((b - 1)*sigma + a*th)/c)*1./th2; }
    }
    const double O[9] = { 0, -omega[2], omega[1], omega[2], 0, -omega[0], -omega[1], omega[0], 0 };
    double O2[9]; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double v = 0; for (int k = 0; k < 3; k++) v += O[3*i + k]*O[3*k + j]; O2[3*i + j] = v; }
    double Wm[9]; for (int k = 0; k < 9; k++) Wm[k] = A*O[k] + B*O2[k] + ((k == 0 || k == 4 || k == 8) ? Cc : 0.0);
    double ups[3]; solve3_lu(Wm, t, ups);
    for (int k = 0; k < 3; k++) { res[k] = omega[k]; res[3 + k] = ups[k]; }
    res[6] = sigma;
}
/* numer_loop_ver2::operator() */
static void pg_residual(const double x1[8], const double x2[8], const double m[8], double res[7]) {
    double q1[4], q2[4]; q_norm(x1, q1); q_norm(x2, q2);
    const double qw2[4] = { q2[0], -q2[1], -q2[2], -q2[3] };
    const double sc = -1./x2[7], ts[3] = { sc*x2[4], sc*x2[5], sc*x2[6] };
    double tw2[3]; q_rot(qw2, ts, tw2);
    const double sw2 = 1./x2[7];
    double q12[4]; q_mul(q1, qw2, q12);
    double rt[3]; q_rot(q1, tw2, rt);
    const double t12[3] = { x1[7]*rt[0] + x1[4], x1[7]*rt[1] + x1[5], x1[7]*rt[2] + x1[6] }, s12 = x1[7]*sw2;
    double rq[4]; q_mul(m, q12, rq);
    double mt[3]; q_rot(m, t12, mt);
    const double rtt[3] = { m[7]*mt[0] + m[4], m[7]*mt[1] + m[5], m[7]*mt[2] + m[6] };
    log_sim3(rq, rtt, m[7]*s12, res);
}
/* tangent-space Jacobian blocks (7 x 7) of one edge by Ceres CENTRAL numeric differentiation of the ambient blocks, then the plus-Jacobian */
static void pg_jacobians(const double x1_[8], const double x2_[8], const double m[8], double J1[49], double J2[49]) {
    const double min_step = sqrt(DBL_EPSILON);
    for (int which = 0; which < 2; which++) {
        double x1[8], x2[8], amb[56], rp[7], rm[7], PJ[12];
        memcpy(x1, x1_, sizeof(x1)); memcpy(x2, x2_, sizeof(x2));
        double *x = which == 0 ? x1 : x2;
        for (int j = 0; j < 8; j++) {
            const double x0 = x[j]; double delta = fabs(x0)*1e-6; if (delta < min_step) delta = min_step;
            x[j] = x0 + delta; pg_residual(x1, x2, m, rp);
            x[j] = x0 - delta; pg_residual(x1, x2, m, rm);
            x[j] = x0;
            const double inv = (1.0/delta)/2;
            for (int k = 0; k < 7; k++) amb[8*k + j] = (rp[k] - rm[k])*inv;
        }
        quat_plus_jacobian(which == 0 ? x1_ : x2_, PJ);
        double *J = which == 0 ? J1 : J2;
        for (int k = 0; k < 7; k++) {
            for (int c = 0; c < 3; c++) { double v = 0; for (int a = 0; a < 4; a++) v += amb[8*k + a]*PJ[3*a + c]; J[7*k + c] = v; }
            for (int c = 0; c < 4; c++) J[7*k + 3 + c] = amb[8*k + 4 + c];
        }
    }
}
void tsloop_oracle_pg_eval(const double x1[8], const double x2[8], const double m[8], double res[7], double J1[49], double J2[49]) {
    pg_residual(x1, x2, m, res); pg_jacobians(x1, x2, m, J1, J2);
}

typedef tsloop_graph_problem tsloop_graph_problem_o;

static double pg_cost(const tsloop_graph_problem_o *p, const double *x) {
    double c = 0;
    for (int e = 0; e < p->n_edge; e++) { double r[7]; pg_residual(x + 8*p->edge_i[e], x + 8*p->edge_j[e], p->meas + 8*e, r); for (int k = 0; k < 7; k++) c += 0.5*r[k]*r[k]; }
    return c;
}
static void pg_linearize(const tsloop_graph_problem_o *p, const double *x, const int *fidx, int n, double *H, double *g, double *cost) {
    memset(H, 0, sizeof(double)*(size_t)n*n); memset(g, 0, sizeof(double)*n); *cost = 0;
    for (int e = 0; e < p->n_edge; e++) {
        const int a = p->edge_i[e], b = p->edge_j[e], fa = fidx[a], fb = fidx[b];
        double r[7], J1[49], J2[49];
        pg_residual(x + 8*a, x + 8*b, p->meas + 8*e, r); pg_jacobians(x + 8*a, x + 8*b, p->meas + 8*e, J1, J2);
        for (int k = 0; k < 7; k++) *cost += 0.5*r[k]*r[k];
        const double *Js[2] = { J1, J2 }; const int fs[2] = { fa, fb };
        for (int u = 0; u < 2; u++) { if (fs[u] < 0) continue;
            for (int c = 0; c < 7; c++) { double v = 0; for (int k = 0; k < 7; k++) v += Js[u][7*k + c]*r[k]; g[7*fs[u] + c] += v; }
            for (int w = 0; w < 2; w++) { if (fs[w] < 0) continue;
                for (int c = 0; c < 7; c++) for (int d = 0; d < 7; d++) { double v = 0; for (int k = 0; k < 7; k++) v += Js[u][7*k + c]*Js[w][7*k + d]; H[(size_t)(7*fs[u] + c)*n + 7*fs[w] + d] += v; } } }
    }
}

int tsloop_oracle_optimize_loop(tsloop_graph_problem_o *p, const tsloop_options *o, tsloop_report *rep) {
    const int N = p->n_kf;
    int *fidx = (int *)malloc(sizeof(int)*(N + 1)); int nf = 0;
    for (int k = 0; k < N; k++) fidx[k] = p->fixed[k] ? -1 : nf++;
    const int n = 7*nf;
    double *x = (double *)malloc(sizeof(double)*8*(size_t)(N + 1)), *cand = (double *)malloc(sizeof(double)*8*(size_t)(N + 1));
    memcpy(x, p->pose, sizeof(double)*8*(size_t)N);
    double *H = (double *)malloc(sizeof(double)*((size_t)n*n + 1)), *A = (double *)malloc(sizeof(double)*((size_t)n*n + 1));
    double *g = (double *)calloc(n + 1, sizeof(double)), *sc = (double *)calloc(n + 1, sizeof(double)), *y = (double *)calloc(n + 1, sizeof(double)), *d = (double *)calloc(n + 1, sizeof(double));
    memset(rep, 0, sizeof(*rep));
    double x_cost; pg_linearize(p, x, fidx, n, H, g, &x_cost); rep->cost0 = x_cost;
    for (int k = 0; k < n; k++) sc[k] = 1.0/(1.0 + sqrt(H[(size_t)k*n + k]));
    #define XNORM(xx, out) do { double v_ = 0; for (int k_ = 0; k_ < N; k_++) if (fidx[k_] >= 0) for (int c_ = 0; c_ < 8; c_++) v_ += (xx)[8*k_ + c_]*(xx)[8*k_ + c_]; out = sqrt(v_); } while (0)
    double x_norm; XNORM(x, x_norm);
    double radius = o->initial_radius, decrease_factor = 2.0; int invalid = 0, term = 0, it = 0, accepted = 0;
    double gmax = 0; for (int k = 0; k < n; k++) if (fabs(g[k]) > gmax) gmax = fabs(g[k]);
    if (n == 0 || p->n_edge == 0) { term = 5; goto done; }
    if (gmax <= o->gradient_tolerance) { term = 3; goto done; }
    while (1) {
        if (it >= o->max_it) { term = 0; break; }
        if (radius < o->min_radius) { term = 4; break; }
        it++;
        for (int k = 0; k < n; k++) { for (int m = 0; m < n; m++) A[(size_t)k*n + m] = sc[k]*H[(size_t)k*n + m]*sc[m];
            A[(size_t)k*n + k] += clampd(sc[k]*sc[k]*H[(size_t)k*n + k], o->min_diagonal, o->max_diagonal)/radius; y[k] = -sc[k]*g[k]; }
        const int rc = chol_solve_dense(A, n, y);
        double model_change = -1;
        if (!rc) {
            for (int k = 0; k < n; k++) d[k] = sc[k]*y[k];
            model_change = 0;
            for (int k = 0; k < n; k++) { double hd = 0; for (int m = 0; m < n; m++) hd += H[(size_t)k*n + m]*d[m]; model_change -= d[k]*(g[k] + 0.5*hd); }
        }
        if (rc || !(model_change > 0)) { if (++invalid >= 5) { term = 5; break; } radius *= 0.5; continue; }
        invalid = 0;
        memcpy(cand, x, sizeof(double)*8*(size_t)N);
        for (int k = 0; k < N; k++) if (fidx[k] >= 0) { const double *dk = d + 7*fidx[k]; quat_plus(x + 8*k, dk, cand + 8*k); for (int c = 0; c < 4; c++) cand[8*k + 4 + c] = x[8*k + 4 + c] + dk[3 + c]; }
        double c_cost = pg_cost(p, cand); if (!(c_cost == c_cost)) c_cost = DBL_MAX;
        double step = 0; for (int k = 0; k < N; k++) if (fidx[k] >= 0) for (int c = 0; c < 8; c++) step += (cand[8*k + c] - x[8*k + c])*(cand[8*k + c] - x[8*k + c]);
        step = sqrt(step);
        if (step <= o->parameter_tolerance*(x_norm + o->parameter_tolerance)) { term = 2; break; }
        const double cost_change = x_cost - c_cost;
        if (fabs(cost_change) <= o->function_tolerance*x_cost) { term = 1; break; }
        const double rel = cost_change/model_change;
        if (rel > o->min_relative_decrease) {
            memcpy(x, cand, sizeof(double)*8*(size_t)N); accepted++;
            XNORM(x, x_norm);
            pg_linearize(p, x, fidx, n, H, g, &x_cost);
            double t = 2.0*rel - 1.0, f = 1.0 - t*t*t; if (f < 1.0/3.0) f = 1.0/3.0;
            radius = radius/f; if (radius > o->max_radius) radius = o->max_radius;
            decrease_factor = 2.0;
            gmax = 0; for (int k = 0; k < n; k++) if (fabs(g[k]) > gmax) gmax = fabs(g[k]);
            if (gmax <= o->gradient_tolerance) { term = 3; break; }
        } else { radius = radius/decrease_factor; decrease_factor *= 2.0; }
    }
done:
    rep->iters = it; rep->accepted = accepted; rep->termination = term; rep->cost1 = x_cost;
    memcpy(p->pose, x, sizeof(double)*8*(size_t)N);
    free(fidx); free(x); free(cand); free(H); free(A); free(g); free(sc); free(y); free(d);
    return term == 5 ? TSLOOP_ERR_NUMERIC : TSLOOP_OK;
}
