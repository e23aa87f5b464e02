// optimizer::OptimizeLoop (src/optimizer.cc:733-957) -- Sim3 pose graph.  Included by tsloop.hip.
//
// 7 tangent dimensions per free keyframe, one 7-row residual block per connection: logSim3(S_ji * S_i * S_j^-1)
// (numer_loop_ver2.h:22-72, ModelTool.hpp:354-432), Jacobians by Ceres' CENTRAL numeric differentiation of the ambient blocks
// (numeric_diff.h: step = max(sqrt(eps), 1e-6 |x|)) followed by the quaternion plus-Jacobian -- the reference uses
// NumericDiffCostFunction, and so does this path: 33 residual evaluations per connection, one thread per connection.
// Normal equations H = sum J^T J are assembled without atomics (a gather per keyframe / per keyframe pair over host-built
// incidence lists: deterministic), damped in unscaled form H + S^-1 clamp(S^2 diag H) S^-1 / radius, and solved by the BA
// library's dense solvers, reused as they are (tsba_solve.h: one workgroup in LDS up to 186 unknowns; tsba_chol.h: multi-workgroup
// blocked Cholesky on the matrix cores beyond that) on the system padded to a multiple of 6.
// LM state lives on the device; the host enqueues max_it iterations of a fixed kernel sequence without synchronising.
#pragma once

struct LmState {                      // (done / step_fail are what the solver kernels look at)
    int done, step_fail;
    int first, need_lin, it, accepted, term, invalid, max_it, cur;
    double radius, decrease_factor, x_cost, x_norm, gmax, cost0;
};
struct Work { int N, n_kf; double *S; int ldS; double *Sy, *g, *dp, *LDbuf; int *fidx, *nfree; long long *dbg; LmState *st; };
#include "tsba_solve.h"
#include "tsba_chol.h"

struct PgDev {
    int n_kf, n_edge, nf, n, n6, npair;
    double *x[2];                      // [n_kf][8]: x = x[cur], candidate = x[cur ^ 1]
    const int *fidx_kf;                // [n_kf] compressed index of a free keyframe, -1 = constant
    const int *free_kf;                // [nf] keyframe of a compressed index
    const int *ei, *ej; const double *meas;
    double *r, *J1, *J2;               // per edge: residual (7), tangent Jacobians (49 each, row-major 7 x 7) at x
    double *cost_e, *ccost_e, *q_e;    // per edge: cost at x, cost at the candidate, |J delta|^2
    const int *kf_off, *kf_inc;        // per free keyframe: incident (edge << 1 | side)
    const int *pair_a, *pair_b, *pair_off, *pair_e;   // per pair of free keyframes (compressed a > b): edges (edge << 1 | side of a)
    double *sc;                        // Jacobi scale per unknown, fixed at the first linearisation
    double *part;                      // per free keyframe: gradient max, |x|^2, step^2, g^T delta
    Work W;
};

// ---- Eigen-style quaternion helpers (w, x, y, z)
__device__ __forceinline__ void pq_mul(const double a[4], const double b[4], double o[4]) {
    o[0] = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3];
    o[1] = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
    o[2] = a[0]*b[2] + a[2]*b[0] + a[3]*b[1] - a[1]*b[3];
    o[3] = a[0]*b[3] + a[3]*b[0] + a[1]*b[2] - a[2]*b[1];
}
__device__ __forceinline__ void pq_rot(const double q[4], const double v[3], double o[3]) {       // QuaternionBase::_transformVector
    double uv[3] = { q[2]*v[2] - q[3]*v[1], q[3]*v[0] - q[1]*v[2], q[1]*v[1] - q[2]*v[0] };
    uv[0] += uv[0]; uv[1] += uv[1]; uv[2] += uv[2];
    o[0] = v[0] + q[0]*uv[0] + (q[2]*uv[2] - q[3]*uv[1]);
    o[1] = v[1] + q[0]*uv[1] + (q[3]*uv[0] - q[1]*uv[2]);
    o[2] = v[2] + q[0]*uv[2] + (q[1]*uv[1] - q[2]*uv[0]);
}
__device__ __forceinline__ void pq_norm(const double q[4], double o[4]) {
    const double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
#pragma unroll
    for (int k = 0; k < 4; k++) o[k] = q[k]/n;
}
// logSim3, ModelTool.hpp:354-432 (W.lu().solve(t): 3x3 elimination with partial pivoting)
__device__ void pg_log_sim3(const double rq[4], const double t[3], double s, double res[7]) {
    const double sigma = log(s), eps = 0.00001;
    const double w = rq[0], x = rq[1], y = rq[2], z = rq[3];
    const double tx = 2*x, ty = 2*y, tz = 2*z, twx = tx*w, twy = ty*w, twz = tz*w, txx = tx*x, txy = ty*x, txz = tz*x, tyy = ty*y, tyz = tz*y, tzz = tz*z;
    const double R0 = 1 - (tyy + tzz), R1 = txy - twz, R2 = txz + twy, R3 = txy + twz, R4 = 1 - (txx + tzz), R5 = tyz - twx, R6 = txz - twy, R7 = tyz + twx, R8 = 1 - (txx + tyy);
    const double d = 0.5*(R0 + R4 + R8 - 1);
    const double dR[3] = { R7 - R5, R2 - R6, R3 - R1 };
    double om[3], A, B, Cc;
    if (fabs(sigma) < eps) {
        Cc = 1;
        if (d > 1 - eps) { for (int k = 0; k < 3; k++) om[k] = 0.5*dR[k]; A = 1./2.; B = 1./6.; }
        else { const double th = acos(d), th2 = th*th, f = th/(2*sqrt(1 - d*d)); for (int k = 0; k < 3; k++) om[k] = f*dR[k];
               A = (1 - cos(th))/th2; B = (th - sin(th))/(th2*th); }
    } else {
        Cc = (s - 1)/sigma;
        if (d > 1 - eps) { const double s2 = sigma*sigma; for (int k = 0; k < 3; k++) om[k] = 0.5*dR[k];
               A = ((sigma - 1)*s + 1)/s2; B = ((0.5*s2 - sigma + 1)*s)/(s2*sigma); }
        else { const double th = acos(d), f = th/(2*sqrt(1 - d*d)); for (int k = 0; k < 3; k++) om[k] = f*dR[k];
               const double th2 = th*th, a = s*sin(th), b = s*cos(th), c = th2 + sigma*sigma;
               A = (a*sigma + (1 - b)*th)/(th*c); B = (Cc - ((b - 1)*sigma + a*th)/c)*1./th2; }
    }
    const double O[9] = { 0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0 };
    double Wm[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) { double v = 0;
#pragma unroll
            for (int k = 0; k < 3; k++) v += O[3*i + k]*O[3*k + j];
            Wm[3*i + j] = A*O[3*i + j] + B*v + (i == j ? Cc : 0.0); }
    double b3[3] = { t[0], t[1], t[2] }, ups[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        int piv = c;
#pragma unroll
        for (int r = c + 1; r < 3; r++) if (fabs(Wm[3*r + c]) > fabs(Wm[3*piv + c])) piv = r;
        if (piv != c) {
#pragma unroll
            for (int k = 0; k < 3; k++) { const double tt = Wm[3*c + k]; Wm[3*c + k] = Wm[3*piv + k]; Wm[3*piv + k] = tt; }
            const double tt = b3[c]; b3[c] = b3[piv]; b3[piv] = tt; }
#pragma unroll
        for (int r = c + 1; r < 3; r++) { const double f = Wm[3*r + c]/Wm[3*c + c];
#pragma unroll
            for (int k = c; k < 3; k++) Wm[3*r + k] -= f*Wm[3*c + k];
            b3[r] -= f*b3[c]; }
    }
#pragma unroll
    for (int r = 2; r >= 0; r--) { double v = b3[r];
#pragma unroll
        for (int k = r + 1; k < 3; k++) v -= Wm[3*r + k]*ups[k];
        ups[r] = v/Wm[3*r + r]; }
#pragma unroll
    for (int k = 0; k < 3; k++) { res[k] = om[k]; res[3 + k] = ups[k]; }
    res[6] = sigma;
}
// numer_loop_ver2::operator()
__device__ void pg_residual(const double x1[8], const double x2[8], const double m[8], double res[7]) {
    double q1[4], q2[4]; pq_norm(x1, q1); pq_norm(x2, q2);
    const double qw2[4] = { q2[0], -q2[1], -q2[2], -q2[3] };
    const double sc = -1./x2[7], ts[3] = { sc*x2[4], sc*x2[5], sc*x2[6] };
    double tw2[3]; pq_rot(qw2, ts, tw2);
    const double sw2 = 1./x2[7];
    double q12[4]; pq_mul(q1, qw2, q12);
    double rt[3]; pq_rot(q1, tw2, rt);
    const double t12[3] = { x1[7]*rt[0] + x1[4], x1[7]*rt[1] + x1[5], x1[7]*rt[2] + x1[6] }, s12 = x1[7]*sw2;
    double rq[4]; pq_mul(m, q12, rq);
    double mt[3]; pq_rot(m, t12, mt);
    const double rtt[3] = { m[7]*mt[0] + m[4], m[7]*mt[1] + m[5], m[7]*mt[2] + m[6] };
    pg_log_sim3(rq, rtt, m[7]*s12, res);
}

// ---- per connection: residual and both tangent Jacobians at x (numeric, CENTRAL), cost
__global__ __launch_bounds__(64) void k_pg_linearize(PgDev P) {
    const LmState *st = P.W.st;
    if (st->done || !st->need_lin) return;
    const int e = blockIdx.x*64 + threadIdx.x;
    if (e >= P.n_edge) return;
    const double *X = P.x[st->cur];
    const int a = P.ei[e], b = P.ej[e];
    double x1[8], x2[8], m[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { x1[k] = X[8*a + k]; x2[k] = X[8*b + k]; m[k] = P.meas[8*(size_t)e + k]; }
    double r[7]; pg_residual(x1, x2, m, r);
    double c = 0.0;
#pragma unroll
    for (int k = 0; k < 7; k++) { P.r[7*(size_t)e + k] = r[k]; c += 0.5*r[k]*r[k]; }
    P.cost_e[e] = c;
    const double min_step = 1.4901161193847656e-08;           // sqrt(DBL_EPSILON)
#pragma unroll 1
    for (int which = 0; which < 2; which++) {
        if (P.fidx_kf[which == 0 ? a : b] < 0) continue;       // constant block: no Jacobian
        double *x = which == 0 ? x1 : x2;
        double amb[7][8];
#pragma unroll 1
        for (int j = 0; j < 8; j++) {
            const double x0 = x[j]; double delta = fabs(x0)*1e-6; if (delta < min_step) delta = min_step;
            double rp[7], rm[7];
            x[j] = x0 + delta; pg_residual(x1, x2, m, rp);
            x[j] = x0 - delta; pg_residual(x1, x2, m, rm);
            x[j] = x0;
            const double inv = (1.0/delta)/2;
#pragma unroll
            for (int k = 0; k < 7; k++) amb[k][j] = (rp[k] - rm[k])*inv;
        }
        // ceres::QuaternionParameterization::ComputeJacobian (4 x 3)
        const double PJ[12] = { -x[1], -x[2], -x[3],  x[0], x[3], -x[2],  -x[3], x[0], x[1],  x[2], -x[1], x[0] };
        double *J = (which == 0 ? P.J1 : P.J2) + 49*(size_t)e;
#pragma unroll
        for (int k = 0; k < 7; k++) {
#pragma unroll
            for (int cc = 0; cc < 3; cc++) J[7*k + cc] = amb[k][0]*PJ[cc] + amb[k][1]*PJ[3 + cc] + amb[k][2]*PJ[6 + cc] + amb[k][3]*PJ[9 + cc];
#pragma unroll
            for (int cc = 0; cc < 4; cc++) J[7*k + 3 + cc] = amb[k][4 + cc];
        }
    }
}

// ---- normal equations: blocks 0 .. nf-1 = diagonal block + gradient of a free keyframe (also Jacobi scale, damping, gradient max,
// |x|^2), blocks nf .. nf+npair-1 = off-diagonal block of a keyframe pair (lower triangle: a > b), last block = padding rows.
__global__ __launch_bounds__(64) void k_pg_assemble(PgDev P, tsloop_options o) {
    LmState *st = P.W.st;
    if (st->done) return;
    const int b = blockIdx.x, t = threadIdx.x, ld = P.W.ldS;
    double *S = P.W.S;
    if (b < P.nf) {
        __shared__ double Hd[49], gd[7];
        if (t < 56) {
            const int row = t < 49 ? t/7 : t - 49, col = t < 49 ? t - 7*(t/7) : -1;     // t < 49: H(row, col); else g(row)
            double v = 0.0;
            for (int q = P.kf_off[b]; q < P.kf_off[b + 1]; q++) {
                const int e = P.kf_inc[q] >> 1, side = P.kf_inc[q] & 1;
                const double *J = (side == 0 ? P.J1 : P.J2) + 49*(size_t)e;
                if (col >= 0) {
#pragma unroll
                    for (int k = 0; k < 7; k++) v += J[7*k + row]*J[7*k + col];
                } else {
                    const double *r = P.r + 7*(size_t)e;
#pragma unroll
                    for (int k = 0; k < 7; k++) v += J[7*k + row]*r[k];
                }
            }
            if (col >= 0) Hd[t] = v; else gd[row] = v;
        }
        __syncthreads();
        if (t < 7) {
            const int k = 7*b + t; const double h = Hd[8*t];
            if (st->first) P.sc[k] = 1.0/(1.0 + sqrt(h));
            P.W.g[k] = gd[t];
        }
        __syncthreads();
        if (t < 49) {
            const int row = t/7, col = t - 7*row, k = 7*b + row;
            double v = Hd[t];
            if (row == col) { const double s = P.sc[k]; v += fmin(fmax(s*s*v, o.min_diagonal), o.max_diagonal)/(s*s)/st->radius; }
            S[(size_t)k*ld + 7*b + col] = v;                      // both triangles: a 6-wide block of the solver can straddle two keyframes
        }
        if (t == 0) {
            double gm = 0.0, xn = 0.0;
            for (int k = 0; k < 7; k++) gm = fmax(gm, fabs(gd[k]));
            const double *x = P.x[st->cur] + 8*P.free_kf[b];
            for (int k = 0; k < 8; k++) xn += x[k]*x[k];
            P.part[4*b] = gm; P.part[4*b + 1] = xn;
        }
    } else if (b < P.nf + P.npair) {
        const int p = b - P.nf, fa = P.pair_a[p], fb = P.pair_b[p];             // fa > fb
        if (t < 49) {
            const int row = t/7, col = t - 7*row;
            double v = 0.0;
            for (int q = P.pair_off[p]; q < P.pair_off[p + 1]; q++) {
                const int e = P.pair_e[q] >> 1, side_a = P.pair_e[q] & 1;
                const double *Ja = (side_a == 0 ? P.J1 : P.J2) + 49*(size_t)e, *Jb = (side_a == 0 ? P.J2 : P.J1) + 49*(size_t)e;
#pragma unroll
                for (int k = 0; k < 7; k++) v += Ja[7*k + row]*Jb[7*k + col];
            }
            S[(size_t)(7*fa + row)*ld + 7*fb + col] = v;
            S[(size_t)(7*fb + col)*ld + 7*fa + row] = v;         // (symmetric copy: a 6-wide diagonal block of the solver can straddle two keyframes)
        }
    } else {
        for (int k = P.n + t; k < P.n6; k += 64) { S[(size_t)k*ld + k] = 1.0; P.W.g[k] = 0.0; }
    }
}

template <int NT>
__device__ __forceinline__ double pg_block_sum(double v, double *lds) {      // fixed order, all threads get the result
    const int t = threadIdx.x;
    lds[t] = v; __syncthreads();
    for (int s = NT/2; s > 0; s >>= 1) { if (t < s) lds[t] += lds[t + s]; __syncthreads(); }
    const double r = lds[0]; __syncthreads();
    return r;
}
template <int NT>
__device__ __forceinline__ double pg_block_max(double v, double *lds) {
    const int t = threadIdx.x;
    lds[t] = v; __syncthreads();
    for (int s = NT/2; s > 0; s >>= 1) { if (t < s) lds[t] = fmax(lds[t], lds[t + s]); __syncthreads(); }
    const double r = lds[0]; __syncthreads();
    return r;
}

// ---- after a (re)linearisation: cost, |x|, gradient max, gradient tolerance; every iteration: the loop-top tests of Ceres
__global__ __launch_bounds__(1024) void k_pg_pre(PgDev P, tsloop_options o) {
    __shared__ double lds[1024];
    LmState *st = P.W.st;
    if (st->done) return;
    const int t = threadIdx.x;
    if (st->need_lin) {
        double c = 0.0, xn = 0.0, gm = 0.0;
        for (int e = t; e < P.n_edge; e += 1024) c += P.cost_e[e];
        for (int b = t; b < P.nf; b += 1024) { gm = fmax(gm, P.part[4*b]); xn += P.part[4*b + 1]; }
        c = pg_block_sum<1024>(c, lds); xn = pg_block_sum<1024>(xn, lds); gm = pg_block_max<1024>(gm, lds);
        if (t == 0) {
            st->x_cost = c; st->x_norm = sqrt(xn); st->gmax = gm;
            if (st->first) st->cost0 = c;
            st->first = 0; st->need_lin = 0;
            if (gm <= o.gradient_tolerance) { st->done = 1; st->term = 3; }
        }
        __syncthreads();
    }
    if (t == 0 && !st->done) {
        if (st->it >= st->max_it) { st->done = 1; st->term = 0; }
        else if (st->radius < o.min_radius) { st->done = 1; st->term = 4; }
        else st->it++;
    }
}

// ---- candidate on the manifold, step^2 and g^T delta per keyframe
__global__ __launch_bounds__(256) void k_pg_candidate(PgDev P) {
    const LmState *st = P.W.st;
    if (st->done) return;
    const int b = blockIdx.x*256 + threadIdx.x;
    if (b < P.nf) {
        const int kf = P.free_kf[b];
        const double *x = P.x[st->cur] + 8*kf; double *c = P.x[st->cur ^ 1] + 8*kf;
        double d[7];
#pragma unroll
        for (int k = 0; k < 7; k++) d[k] = P.W.dp[7*b + k];
        double q[4] = { x[0], x[1], x[2], x[3] }, qn[4];
        quat_plus(q, d, qn);
        double s2 = 0.0, gd = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) { c[k] = qn[k]; s2 += (qn[k] - q[k])*(qn[k] - q[k]); }
#pragma unroll
        for (int k = 0; k < 4; k++) { c[4 + k] = x[4 + k] + d[3 + k]; s2 += d[3 + k]*d[3 + k]; }
#pragma unroll
        for (int k = 0; k < 7; k++) gd += P.W.g[7*b + k]*d[k];
        P.part[4*b + 2] = s2; P.part[4*b + 3] = gd;
    }
    // constant keyframes: the candidate buffer carries the same values (written once at upload)
}
// ---- per connection: |J delta|^2 and the cost at the candidate
__global__ __launch_bounds__(64) void k_pg_trial(PgDev P) {
    const LmState *st = P.W.st;
    if (st->done) return;
    const int e = blockIdx.x*64 + threadIdx.x;
    if (e >= P.n_edge) return;
    const int a = P.ei[e], b = P.ej[e], fa = P.fidx_kf[a], fb = P.fidx_kf[b];
    double v[7] = { 0, 0, 0, 0, 0, 0, 0 };
    if (fa >= 0) { const double *J = P.J1 + 49*(size_t)e;
#pragma unroll
        for (int k = 0; k < 7; k++)
#pragma unroll
            for (int c = 0; c < 7; c++) v[k] += J[7*k + c]*P.W.dp[7*fa + c]; }
    if (fb >= 0) { const double *J = P.J2 + 49*(size_t)e;
#pragma unroll
        for (int k = 0; k < 7; k++)
#pragma unroll
            for (int c = 0; c < 7; c++) v[k] += J[7*k + c]*P.W.dp[7*fb + c]; }
    double q = 0.0;
#pragma unroll
    for (int k = 0; k < 7; k++) q += v[k]*v[k];
    P.q_e[e] = q;
    const double *X = P.x[st->cur ^ 1];
    double x1[8], x2[8], m[8], r[7];
#pragma unroll
    for (int k = 0; k < 8; k++) { x1[k] = X[8*a + k]; x2[k] = X[8*b + k]; m[k] = P.meas[8*(size_t)e + k]; }
    pg_residual(x1, x2, m, r);
    double c = 0.0;
#pragma unroll
    for (int k = 0; k < 7; k++) c += 0.5*r[k]*r[k];
    P.ccost_e[e] = c;
}
// ---- Ceres' step acceptance (trust_region_minimizer.cc), as k_decide of the BA library
__global__ __launch_bounds__(1024) void k_pg_decide(PgDev P, tsloop_options o) {
    __shared__ double lds[1024];
    LmState *st = P.W.st;
    if (st->done) return;
    const int t = threadIdx.x;
    double s2 = 0.0, gd = 0.0, qq = 0.0, cc = 0.0;
    for (int b = t; b < P.nf; b += 1024) { s2 += P.part[4*b + 2]; gd += P.part[4*b + 3]; }
    for (int e = t; e < P.n_edge; e += 1024) { qq += P.q_e[e]; cc += P.ccost_e[e]; }
    s2 = pg_block_sum<1024>(s2, lds); gd = pg_block_sum<1024>(gd, lds); qq = pg_block_sum<1024>(qq, lds); cc = pg_block_sum<1024>(cc, lds);
    if (t != 0) return;
    // dp = -(solution of S y = g) already: delta = dp.  model_cost_change = -(J d)^T (r + J d / 2) = -g^T d - |J d|^2 / 2
    const double model_change = -gd - 0.5*qq;
    if (st->step_fail || !(model_change > 0.0)) {
        st->step_fail = 0;
        if (++st->invalid >= 5) { st->done = 1; st->term = 5; return; }
        st->radius *= 0.5; return;
    }
    st->invalid = 0;
    double c_cost = cc; if (!(c_cost == c_cost)) c_cost = 1.7976931348623157e308;
    const double step = sqrt(s2);
    if (step <= o.parameter_tolerance*(st->x_norm + o.parameter_tolerance)) { st->done = 1; st->term = 2; return; }
    const double cost_change = st->x_cost - c_cost;
    if (fabs(cost_change) <= o.function_tolerance*st->x_cost) { st->done = 1; st->term = 1; return; }
    const double rel = cost_change/model_change;
    if (rel > o.min_relative_decrease) {
        st->cur ^= 1; st->accepted++; st->need_lin = 1; st->x_cost = c_cost;
        double tt = 2.0*rel - 1.0, f = 1.0 - tt*tt*tt; if (f < 1.0/3.0) f = 1.0/3.0;
        st->radius = fmin(st->radius/f, o.max_radius); st->decrease_factor = 2.0;
    } else { st->radius = st->radius/st->decrease_factor; st->decrease_factor *= 2.0; }
}
// after an accepted step the constant keyframes of the new candidate buffer are already right (both buffers carry them); nothing to do.
