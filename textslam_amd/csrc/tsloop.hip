// libtsloop.so -- loop-closure optimisers on gfx950 (include/tsloop.h).
//
// optimizer::OptimizeSim3 (src/optimizer.cc:626-731): 7 degrees of freedom, a few hundred matches, two 2-row residual blocks per match.
// The whole Levenberg-Marquardt solve is ONE launch of one 256-thread workgroup: the problem is far too small to spread (a sweep is
// one round of ~400 instructions per lane), so the only thing worth optimising is the number of dependent launches -- zero.  Per
// iteration: sweep (residuals, closed-form tangent-space Jacobians, Huber weights; 36 sums = 28 J^T W J + 7 J^T W r + cost through LDS
// transposes), Jacobi-scaled damped 7x7 LDL^T in registers (every thread, redundantly), candidate on the manifold, cost sweep,
// Ceres' accept / reject / radius logic -- then the 4-pixel inlier test of the reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <chrono>
#include <string>
#include <algorithm>
#include "../../include/tsloop.h"
#include "tsba_device.h"
#include <vector>
#include <map>
#include "tsloop_graph.h"

#define SIM_T 256
#define SIM_NW (SIM_T/64)

struct LCtx {
    int device = 0; hipStream_t stream = nullptr; std::string err;
    uint8_t *h_stage = nullptr; size_t h_cap = 0; uint8_t *d_buf = nullptr; size_t d_cap = 0;
};
#define CKL(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { c->err = std::string(#x) + ": " + hipGetErrorString(e_); return TSLOOP_ERR_DEVICE; } } while (0)

struct Sim3Dev { int n; const double *P1, *P2; const float *uv1, *uv2; uint8_t *inlier; double K[4]; double *sim; tsloop_report *rep; };

// residuals (4) and tangent-space Jacobians (4 x 7: delta(3) | t(3) | s) of one match.  Ceres' quaternion Plus perturbs on the left with
// the half angle: R+ = Exp(2 delta) R.
//   auto_sim    X = s R P2 + t:        dX/ddelta = -2 [s R P2]x,   dX/dt = I,        dX/ds = R P2
//   auto_siminv Y = R^T (P1 - t) / s:  dY/ddelta = (2/s) R^T [P1 - t]x,  dY/dt = -R^T / s,  dY/ds = -Y / s
__device__ __forceinline__ void sim3_match(const double R[9], const double t[3], double s, const double P1[3], const double P2[3],
                                           double u1, double v1, double u2, double v2, const double K[4], bool want_j, double r[4], double J[4][7]) {
    const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
    {
        double A[3]; mat3_vec(R, P2, A);
        const double sA[3] = { s*A[0], s*A[1], s*A[2] };
        const double X = sA[0] + t[0], Y = sA[1] + t[1], Z = sA[2] + t[2], iz = 1.0/Z;
        r[0] = X*iz*fx + cx - u1; r[1] = Y*iz*fy + cy - v1;
        if (want_j) {
            const double d0[3] = { fx*iz, 0.0, -fx*X*iz*iz }, d1[3] = { 0.0, fy*iz, -fy*Y*iz*iz };
            // row * (-2 [sA]x): (d x sA)-type products; [v]x = [[0,-vz,vy],[vz,0,-vx],[-vy,vx,0]]
            J[0][0] = -2.0*(d0[1]*sA[2] - d0[2]*sA[1]); J[0][1] = -2.0*(d0[2]*sA[0] - d0[0]*sA[2]); J[0][2] = -2.0*(d0[0]*sA[1] - d0[1]*sA[0]);
            J[1][0] = -2.0*(d1[1]*sA[2] - d1[2]*sA[1]); J[1][1] = -2.0*(d1[2]*sA[0] - d1[0]*sA[2]); J[1][2] = -2.0*(d1[0]*sA[1] - d1[1]*sA[0]);
#pragma unroll
            for (int k = 0; k < 3; k++) { J[0][3 + k] = d0[k]; J[1][3 + k] = d1[k]; }
            J[0][6] = d0[0]*A[0] + d0[2]*A[2]; J[1][6] = d1[1]*A[1] + d1[2]*A[2];
        }
    }
    {
        const double D[3] = { P1[0] - t[0], P1[1] - t[1], P1[2] - t[2] }, is = 1.0/s;
        double E[3]; mat3T_vec(R, D, E);
        const double X = E[0]*is, Y = E[1]*is, Z = E[2]*is, iz = 1.0/Z;
        r[2] = X*iz*fx + cx - u2; r[3] = Y*iz*fy + cy - v2;
        if (want_j) {
            const double d0[3] = { fx*iz, 0.0, -fx*X*iz*iz }, d1[3] = { 0.0, fy*iz, -fy*Y*iz*iz };
            // g = R d (so that d^T R^T M = (R d)^T M); d^T (2/s) R^T [D]x = (2/s) (g x D)^T ... with [D]x v = D x v: g^T [D]x = (g x D)^T
            double g0[3], g1[3]; mat3_vec(R, d0, g0); mat3_vec(R, d1, g1);
            const double c2 = 2.0*is;
            J[2][0] = c2*(g0[1]*D[2] - g0[2]*D[1]); J[2][1] = c2*(g0[2]*D[0] - g0[0]*D[2]); J[2][2] = c2*(g0[0]*D[1] - g0[1]*D[0]);
            J[3][0] = c2*(g1[1]*D[2] - g1[2]*D[1]); J[3][1] = c2*(g1[2]*D[0] - g1[0]*D[2]); J[3][2] = c2*(g1[0]*D[1] - g1[1]*D[0]);
#pragma unroll
            for (int k = 0; k < 3; k++) { J[2][3 + k] = -g0[k]*is; J[3][3 + k] = -g1[k]*is; }
            J[2][6] = -(d0[0]*X + d0[2]*Z)*is; J[3][6] = -(d1[1]*Y + d1[2]*Z)*is;
        }
    }
}

// sums of NV <= 18 per-thread values over the workgroup: thread v < NV returns the total (fixed order: deterministic)
template <int NV>
__device__ __forceinline__ double wg_sum_to_lane(const double *acc, double *lds /* SIM_NW*18*65 + SIM_NW*32 */) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *reg = lds + wave*18*65, *xw = lds + SIM_NW*18*65;
#pragma unroll
    for (int i = 0; i < NV; i++) reg[i*65 + lane] = acc[i];
    __syncthreads();
    if (lane < NV) {
        const double *row = reg + lane*65;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int q = 0; q < 64; q += 4) { s0 += row[q]; s1 += row[q + 1]; s2 += row[q + 2]; s3 += row[q + 3]; }
        xw[wave*32 + lane] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    double tot = 0.0;
    if (tid < NV) {
#pragma unroll
        for (int w = 0; w < SIM_NW; w++) tot += xw[w*32 + tid];
    }
    __syncthreads();
    return tot;
}

__device__ __forceinline__ constexpr int sym7(int r, int c) { return r <= c ? r*7 - r*(r - 1)/2 + (c - r) : c*7 - c*(c - 1)/2 + (r - c); }

__global__ __launch_bounds__(SIM_T) void k_sim3_lm(Sim3Dev P, tsloop_options o) {
    __shared__ double lds[SIM_NW*18*65 + SIM_NW*32];
    __shared__ double s_tot[40];
    const int tid = threadIdx.x;
    double x[8];
    {   const double n = sqrt(P.sim[0]*P.sim[0] + P.sim[1]*P.sim[1] + P.sim[2]*P.sim[2] + P.sim[3]*P.sim[3]);      // q = q.normalized(), optimizer.cc:639
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = P.sim[k]/n;
#pragma unroll
        for (int k = 4; k < 8; k++) x[k] = P.sim[k]; }
    // full = true: M (28, upper, sym7 order), c (7), cost -> s_tot[0..35]; false: cost only -> s_tot[35].  All threads get them.
    auto sweep = [&](const double *xx, bool full) {
        double R[9]; quat_to_R(xx, R);
        double acc[36];
#pragma unroll
        for (int k = 0; k < 36; k++) acc[k] = 0.0;
#pragma unroll 1
        for (int i = tid; i < P.n; i += SIM_T) {
            if (!P.inlier[i]) continue;
            const double P1[3] = { P.P1[3*i], P.P1[3*i+1], P.P1[3*i+2] }, P2[3] = { P.P2[3*i], P.P2[3*i+1], P.P2[3*i+2] };
            double r[4], J[4][7];
            sim3_match(R, xx + 4, xx[7], P1, P2, (double)P.uv1[2*i], (double)P.uv1[2*i+1], (double)P.uv2[2*i], (double)P.uv2[2*i+1], P.K, full, r, J);
#pragma unroll
            for (int b = 0; b < 2; b++) {                             // one robust weight per residual block
                double w; acc[35] += 0.5*huber(r[2*b]*r[2*b] + r[2*b+1]*r[2*b+1], o.huber_delta, w);
                if (full) {
                    int q = 0;
#pragma unroll
                    for (int a = 0; a < 7; a++)
#pragma unroll
                        for (int cc = a; cc < 7; cc++) { acc[q] += w*(J[2*b][a]*J[2*b][cc] + J[2*b+1][a]*J[2*b+1][cc]); q++; }
#pragma unroll
                    for (int a = 0; a < 7; a++) acc[28 + a] += w*(J[2*b][a]*r[2*b] + J[2*b+1][a]*r[2*b+1]);
                }
            }
        }
        if (full) {
            const double t0 = wg_sum_to_lane<18>(acc, lds), t1 = wg_sum_to_lane<18>(acc + 18, lds);
            if (tid < 18) { s_tot[tid] = t0; s_tot[18 + tid] = t1; }
        } else {
            const double t1 = wg_sum_to_lane<1>(acc + 35, lds);
            if (tid == 0) s_tot[35] = t1;
        }
        __syncthreads();
    };
    int nact = 0;
    for (int i = 0; i < P.n; i++) nact += P.inlier[i] ? 1 : 0;          // (uniform, tiny)
    double M[28], c[7], sc[7], x_cost, x_norm;
    auto install = [&]() {
#pragma unroll
        for (int k = 0; k < 28; k++) M[k] = s_tot[k];
#pragma unroll
        for (int k = 0; k < 7; k++) c[k] = s_tot[28 + k];
        x_cost = s_tot[35];
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 8; k++) v += x[k]*x[k];
        x_norm = sqrt(v);
        __syncthreads();
    };
    sweep(x, true); install();
#pragma unroll
    for (int k = 0; k < 7; k++) sc[k] = 1.0/(1.0 + sqrt(M[sym7(k, k)]));
    const double cost0 = x_cost;
    double radius = o.initial_radius, decrease_factor = 2.0; int invalid = 0, term = 0, it = 0, accepted = 0;
    auto gmax_of = [&]() { double g = 0.0;
#pragma unroll
        for (int k = 0; k < 7; k++) g = fmax(g, fabs(c[k])); return g; };
    bool stop = false;
    if (nact == 0) { term = 5; stop = true; }
    else if (gmax_of() <= o.gradient_tolerance) { term = 3; stop = true; }
    while (!stop) {
        if (it >= o.max_it) { term = 0; break; }
        if (radius < o.min_radius) { term = 4; break; }
        it++;
        // (S M S + D / radius) y = -S c, LDL^T without pivoting (positive definite by construction), d = S y
        double A[28], y[7], d[7]; bool bad = false;
#pragma unroll
        for (int a = 0; a < 7; a++)
#pragma unroll
            for (int b = a; b < 7; b++) A[sym7(a, b)] = sc[a]*M[sym7(a, b)]*sc[b];
#pragma unroll
        for (int a = 0; a < 7; a++) { const double dg = fmin(fmax(sc[a]*sc[a]*M[sym7(a, a)], o.min_diagonal), o.max_diagonal); A[sym7(a, a)] += dg/radius; y[a] = -sc[a]*c[a]; }
        double L[7][7], dd[7];
#pragma unroll
        for (int j = 0; j < 7; j++) {
            double v = A[sym7(j, j)];
#pragma unroll
            for (int k = 0; k < j; k++) v -= L[j][k]*L[j][k]*dd[k];
            if (!(v > 0.0)) { bad = true; v = 1.0; }
            dd[j] = v;
#pragma unroll
            for (int i = j + 1; i < 7; i++) { double u = A[sym7(j, i)];
#pragma unroll
                for (int k = 0; k < j; k++) u -= L[i][k]*L[j][k]*dd[k];
                L[i][j] = u/v; }
        }
#pragma unroll
        for (int i = 0; i < 7; i++) {
#pragma unroll
            for (int k = 0; k < i; k++) y[i] -= L[i][k]*y[k]; }
#pragma unroll
        for (int i = 0; i < 7; i++) y[i] /= dd[i];
#pragma unroll
        for (int i = 6; i >= 0; i--) {
#pragma unroll
            for (int k = i + 1; k < 7; k++) y[i] -= L[k][i]*y[k]; }
        double model_change = -1.0;
        if (!bad) {
#pragma unroll
            for (int k = 0; k < 7; k++) d[k] = sc[k]*y[k];
            model_change = 0.0;                                        // -(J d)^T (r + J d / 2) = -d^T (c + M d / 2)
#pragma unroll
            for (int k = 0; k < 7; k++) { double hd = 0.0;
#pragma unroll
                for (int m = 0; m < 7; m++) hd += M[sym7(k, m)]*d[m];
                model_change -= d[k]*(c[k] + 0.5*hd); }
        }
        if (bad || !(model_change > 0.0)) { if (++invalid >= 5) { term = 5; break; } radius *= 0.5; continue; }
        invalid = 0;
        double cand[8];
        quat_plus(x, d, cand);
#pragma unroll
        for (int k = 0; k < 4; k++) cand[4 + k] = x[4 + k] + d[3 + k];
        sweep(cand, true);                                             // speculative: the linearisation at the candidate (cost in s_tot[35])
        double c_cost = s_tot[35]; if (!(c_cost == c_cost)) c_cost = 1.7976931348623157e308;
        double step = 0.0;
#pragma unroll
        for (int k = 0; k < 8; k++) step += (cand[k] - x[k])*(cand[k] - x[k]);
        step = sqrt(step);
        if (step <= o.parameter_tolerance*(x_norm + o.parameter_tolerance)) { term = 2; break; }
        const double cost_change = x_cost - c_cost;
        if (fabs(cost_change) <= o.function_tolerance*x_cost) { term = 1; break; }
        const double rel = cost_change/model_change;
        if (rel > o.min_relative_decrease) {
#pragma unroll
            for (int k = 0; k < 8; k++) x[k] = cand[k];
            install(); accepted++;
            double t = 2.0*rel - 1.0, f = 1.0 - t*t*t; if (f < 1.0/3.0) f = 1.0/3.0;
            radius = fmin(radius/f, o.max_radius); decrease_factor = 2.0;
            if (gmax_of() <= o.gradient_tolerance) { term = 3; break; }
        } else { radius = radius/decrease_factor; decrease_factor *= 2.0; __syncthreads(); }
    }
    __syncthreads();
    // result (q12.normalized()) and the inlier test, optimizer.cc:682-729
    {   const double n = sqrt(x[0]*x[0] + x[1]*x[1] + x[2]*x[2] + x[3]*x[3]);
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = x[k]/n; }
    double R[9]; quat_to_R(x, R);
    int ninl = 0;
    for (int i = tid; i < P.n; i += SIM_T) {
        if (!P.inlier[i]) continue;
        const double P1[3] = { P.P1[3*i], P.P1[3*i+1], P.P1[3*i+2] }, P2[3] = { P.P2[3*i], P.P2[3*i+1], P.P2[3*i+2] };
        double r[4], J[4][7];
        sim3_match(R, x + 4, x[7], P1, P2, (double)P.uv1[2*i], (double)P.uv1[2*i+1], (double)P.uv2[2*i], (double)P.uv2[2*i+1], P.K, false, r, J);
        if (fabs(r[0]) >= o.thresh_outlier || fabs(r[1]) >= o.thresh_outlier || fabs(r[2]) >= o.thresh_outlier || fabs(r[3]) >= o.thresh_outlier) P.inlier[i] = 0;
        else ninl++;
    }
    double nv = (double)ninl;
    const double tot = wg_sum_to_lane<1>(&nv, lds);
    if (tid == 0) {
#pragma unroll
        for (int k = 0; k < 8; k++) P.sim[k] = x[k];
        P.rep->iters = it; P.rep->accepted = accepted; P.rep->termination = term; P.rep->n_inlier = (int)tot; P.rep->cost0 = cost0; P.rep->cost1 = x_cost;
    }
}

// ------------------------------------------------------------------------------------------------ host side
extern "C" {

void tsloop_default_options_sim3(tsloop_options *o) {
    memset(o, 0, sizeof(*o));
    o->max_it = 20; o->huber_delta = sqrt(10.0); o->thresh_outlier = 4.0;
    o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32; o->min_relative_decrease = 1e-3;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8; o->min_diagonal = 1e-6; o->max_diagonal = 1e32;
}
int tsloop_create(int device, void **ctx) {
    if (!ctx) return TSLOOP_ERR_ARG;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || device < 0 || device >= nd) return TSLOOP_ERR_DEVICE;       // no GPU: fail loudly, no CPU path
    LCtx *c = new LCtx(); c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) { delete c; return TSLOOP_ERR_DEVICE; }
    *ctx = c; return TSLOOP_OK;
}
void tsloop_destroy(void *ctx) {
    LCtx *c = (LCtx *)ctx; if (!c) return;
    hipSetDevice(c->device);
    if (c->d_buf) hipFree(c->d_buf);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}
const char *tsloop_last_error(void *ctx) { LCtx *c = (LCtx *)ctx; return c ? c->err.c_str() : "null context"; }

int tsloop_optimize_sim3(void *ctx, tsloop_sim3_problem *p, const tsloop_options *o, tsloop_report *r) {
    LCtx *c = (LCtx *)ctx;
    if (!c || !p || !o || !r || p->n < 0 || (p->n > 0 && (!p->P1 || !p->P2 || !p->uv1 || !p->uv2 || !p->inlier))) return TSLOOP_ERR_ARG;
    if (!(p->sim[7] > 0.0)) { c->err = "scale must be positive"; return TSLOOP_ERR_ARG; }
    hipSetDevice(c->device);
    const auto t0 = std::chrono::steady_clock::now();
    const size_t n = (size_t)p->n;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_P1 = 0, o_P2 = o_P1 + al(24*n), o_u1 = o_P2 + al(24*n), o_u2 = o_u1 + al(8*n), o_sim = o_u2 + al(8*n), o_rep = o_sim + al(64),
                 o_inl = o_rep + al(sizeof(tsloop_report)), tot = o_inl + al(n);
    if (tot > c->h_cap) { if (c->h_stage) hipHostFree(c->h_stage); c->h_stage = nullptr; c->h_cap = 0; CKL(hipHostMalloc((void **)&c->h_stage, tot, hipHostMallocDefault)); c->h_cap = tot; }
    if (tot > c->d_cap) { if (c->d_buf) hipFree(c->d_buf); c->d_buf = nullptr; c->d_cap = 0; CKL(hipMalloc((void **)&c->d_buf, tot)); c->d_cap = tot; }
    uint8_t *h = c->h_stage, *d = c->d_buf;
    memcpy(h + o_P1, p->P1, 24*n); memcpy(h + o_P2, p->P2, 24*n); memcpy(h + o_u1, p->uv1, 8*n); memcpy(h + o_u2, p->uv2, 8*n);
    memcpy(h + o_sim, p->sim, 64); memset(h + o_rep, 0, sizeof(tsloop_report)); memcpy(h + o_inl, p->inlier, n);
    CKL(hipMemcpyAsync(d, h, tot, hipMemcpyHostToDevice, c->stream));                 // one staged copy in, one out
    Sim3Dev D; D.n = p->n; D.P1 = (const double *)(d + o_P1); D.P2 = (const double *)(d + o_P2); D.uv1 = (const float *)(d + o_u1); D.uv2 = (const float *)(d + o_u2);
    D.inlier = d + o_inl; memcpy(D.K, p->K, sizeof(D.K)); D.sim = (double *)(d + o_sim); D.rep = (tsloop_report *)(d + o_rep);
    hipLaunchKernelGGL(k_sim3_lm, dim3(1), dim3(SIM_T), 0, c->stream, D, *o);
    CKL(hipMemcpyAsync(h + o_sim, d + o_sim, tot - o_sim, hipMemcpyDeviceToHost, c->stream));
    CKL(hipStreamSynchronize(c->stream)); CKL(hipGetLastError());
    memcpy(p->sim, h + o_sim, 64); memcpy(r, h + o_rep, sizeof(tsloop_report)); memcpy(p->inlier, h + o_inl, n);
    r->t_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return r->termination == 5 ? TSLOOP_ERR_NUMERIC : TSLOOP_OK;
}

void tsloop_default_options_loop(tsloop_options *o) {
    tsloop_default_options_sim3(o);
    o->huber_delta = 0.0; o->thresh_outlier = 0.0;            // no loss function (nullptr, optimizer.cc:818,856), no inlier test
}

int tsloop_optimize_loop(void *ctx, tsloop_graph_problem *p, const tsloop_options *o, tsloop_report *r) {
    LCtx *c = (LCtx *)ctx;
    if (!c || !p || !o || !r || p->n_kf <= 0 || p->n_edge < 0 || !p->pose || !p->fixed || (p->n_edge > 0 && (!p->edge_i || !p->edge_j || !p->meas))) return TSLOOP_ERR_ARG;
    for (int e = 0; e < p->n_edge; e++)
        if (p->edge_i[e] < 0 || p->edge_i[e] >= p->n_kf || p->edge_j[e] < 0 || p->edge_j[e] >= p->n_kf || p->edge_i[e] == p->edge_j[e]) { c->err = "bad edge index"; return TSLOOP_ERR_ARG; }
    for (int k = 0; k < p->n_kf; k++) if (!(p->pose[8*k + 7] > 0.0)) { c->err = "scale must be positive"; return TSLOOP_ERR_ARG; }
    hipSetDevice(c->device);
    const auto t0 = std::chrono::steady_clock::now();
    memset(r, 0, sizeof(*r));
    const int N = p->n_kf, E = p->n_edge;
    // ---- host plan: compressed free keyframes, incidence lists, keyframe pairs
    std::vector<int> fidx(N), free_kf; int nf = 0;
    for (int k = 0; k < N; k++) { fidx[k] = p->fixed[k] ? -1 : nf++; if (!p->fixed[k]) free_kf.push_back(k); }
    if (nf == 0 || E == 0) { r->termination = 5; return TSLOOP_ERR_NUMERIC; }
    std::vector<int> kf_off(nf + 1, 0), kf_inc;
    for (int e = 0; e < E; e++) { if (fidx[p->edge_i[e]] >= 0) kf_off[fidx[p->edge_i[e]] + 1]++; if (fidx[p->edge_j[e]] >= 0) kf_off[fidx[p->edge_j[e]] + 1]++; }
    for (int k = 0; k < nf; k++) kf_off[k + 1] += kf_off[k];
    kf_inc.resize(kf_off[nf]);
    { std::vector<int> cur(kf_off.begin(), kf_off.end() - 1);
      for (int e = 0; e < E; e++) { const int fa = fidx[p->edge_i[e]], fb = fidx[p->edge_j[e]];
          if (fa >= 0) kf_inc[cur[fa]++] = e << 1; if (fb >= 0) kf_inc[cur[fb]++] = (e << 1) | 1; } }
    std::map<std::pair<int,int>, std::vector<int>> pm;              // (a > b) -> edges (edge << 1 | side of a)
    for (int e = 0; e < E; e++) { const int fa = fidx[p->edge_i[e]], fb = fidx[p->edge_j[e]];
        if (fa < 0 || fb < 0) continue;
        if (fa > fb) pm[{fa, fb}].push_back(e << 1); else pm[{fb, fa}].push_back((e << 1) | 1); }
    std::vector<int> pair_a, pair_b, pair_off(1, 0), pair_e;
    for (auto &kv : pm) { pair_a.push_back(kv.first.first); pair_b.push_back(kv.first.second); for (int v : kv.second) pair_e.push_back(v); pair_off.push_back((int)pair_e.size()); }
    const int npair = (int)pair_a.size(), n = 7*nf, n6 = (n + 5)/6*6, nb6 = n6/6;
    // ---- device buffers (one allocation)
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0; auto take = [&](size_t bytes) { const size_t o_ = off; off += al(bytes); return o_; };
    const size_t o_x0 = take(64*(size_t)N), o_x1 = take(64*(size_t)N), o_fidx = take(4*(size_t)N), o_free = take(4*(size_t)nf), o_ei = take(4*(size_t)E), o_ej = take(4*(size_t)E),
                 o_meas = take(64*(size_t)E), o_kfoff = take(4*(size_t)(nf + 1)), o_kfinc = take(4*kf_inc.size() + 4), o_pa = take(4*(size_t)npair + 4), o_pb = take(4*(size_t)npair + 4),
                 o_poff = take(4*(size_t)(npair + 1)), o_pe = take(4*pair_e.size() + 4), o_sfidx = take(4*(size_t)nb6), o_nfree = take(4), o_st = take(sizeof(LmState)),
                 o_host_end = off,                                       // everything up to here is uploaded
                 o_r = take(56*(size_t)E), o_J1 = take(392*(size_t)E), o_J2 = take(392*(size_t)E), o_ce = take(8*(size_t)E), o_cce = take(8*(size_t)E), o_qe = take(8*(size_t)E),
                 o_sc = take(8*(size_t)n6), o_part = take(32*(size_t)nf), o_g = take(8*(size_t)n6), o_dp = take(8*(size_t)n6), o_Sy = take(8*(size_t)n6),
                 o_LD = take(8*(size_t)std::max(n6, 32*nb6)), o_dbg = take(8*64), o_S = take(8*((size_t)n6 + 1)*n6), tot = off;
    if (o_host_end > c->h_cap) { if (c->h_stage) hipHostFree(c->h_stage); c->h_stage = nullptr; c->h_cap = 0; CKL(hipHostMalloc((void **)&c->h_stage, o_host_end, hipHostMallocDefault)); c->h_cap = o_host_end; }
    if (tot > c->d_cap) { if (c->d_buf) hipFree(c->d_buf); c->d_buf = nullptr; c->d_cap = 0; CKL(hipMalloc((void **)&c->d_buf, tot)); c->d_cap = tot; }
    uint8_t *h = c->h_stage, *d = c->d_buf;
    memcpy(h + o_x0, p->pose, 64*(size_t)N); memcpy(h + o_x1, p->pose, 64*(size_t)N);
    memcpy(h + o_fidx, fidx.data(), 4*(size_t)N); memcpy(h + o_free, free_kf.data(), 4*(size_t)nf);
    memcpy(h + o_ei, p->edge_i, 4*(size_t)E); memcpy(h + o_ej, p->edge_j, 4*(size_t)E); memcpy(h + o_meas, p->meas, 64*(size_t)E);
    memcpy(h + o_kfoff, kf_off.data(), 4*(size_t)(nf + 1)); memcpy(h + o_kfinc, kf_inc.data(), 4*kf_inc.size());
    memcpy(h + o_pa, pair_a.data(), 4*(size_t)npair); memcpy(h + o_pb, pair_b.data(), 4*(size_t)npair); memcpy(h + o_poff, pair_off.data(), 4*(size_t)(npair + 1)); memcpy(h + o_pe, pair_e.data(), 4*pair_e.size());
    { int *sf = (int *)(h + o_sfidx); for (int k = 0; k < nb6; k++) sf[k] = k; *(int *)(h + o_nfree) = nb6; }
    LmState st0; memset(&st0, 0, sizeof(st0));
    st0.first = 1; st0.need_lin = 1; st0.max_it = o->max_it; st0.radius = o->initial_radius; st0.decrease_factor = 2.0;
    memcpy(h + o_st, &st0, sizeof(st0));
    CKL(hipMemcpyAsync(d, h, o_host_end, hipMemcpyHostToDevice, c->stream));
    PgDev P; memset(&P, 0, sizeof(P));
    P.n_kf = N; P.n_edge = E; P.nf = nf; P.n = n; P.n6 = n6; P.npair = npair;
    P.x[0] = (double *)(d + o_x0); P.x[1] = (double *)(d + o_x1); P.fidx_kf = (const int *)(d + o_fidx); P.free_kf = (const int *)(d + o_free);
    P.ei = (const int *)(d + o_ei); P.ej = (const int *)(d + o_ej); P.meas = (const double *)(d + o_meas);
    P.r = (double *)(d + o_r); P.J1 = (double *)(d + o_J1); P.J2 = (double *)(d + o_J2); P.cost_e = (double *)(d + o_ce); P.ccost_e = (double *)(d + o_cce); P.q_e = (double *)(d + o_qe);
    P.kf_off = (const int *)(d + o_kfoff); P.kf_inc = (const int *)(d + o_kfinc); P.pair_a = (const int *)(d + o_pa); P.pair_b = (const int *)(d + o_pb);
    P.pair_off = (const int *)(d + o_poff); P.pair_e = (const int *)(d + o_pe); P.sc = (double *)(d + o_sc); P.part = (double *)(d + o_part);
    Work &W = P.W; W.N = n6; W.n_kf = nb6; W.S = (double *)(d + o_S); W.ldS = n6; W.Sy = (double *)(d + o_Sy); W.g = (double *)(d + o_g); W.dp = (double *)(d + o_dp);
    W.LDbuf = (double *)(d + o_LD); W.fidx = (int *)(d + o_sfidx); W.nfree = (int *)(d + o_nfree); W.dbg = (long long *)(d + o_dbg); W.st = (LmState *)(d + o_st);
    // ---- solver selection as in the BA library
    const size_t lds_small = solve_lds_doubles(n6)*sizeof(double);
    const bool use_lds = lds_small <= 160*1024 - 64;
    const int lds_diag = (int)(solve_diag_lds_doubles()*sizeof(double)), lds_panel = (CH_NB + 64)*(CH_NB + 1)*(int)sizeof(double), lds_upd = 2*64*(CH_NB + 1)*(int)sizeof(double),
              lds_bs = (CH_NB*(CH_NB + 1) + 2*CH_NB + 8*CH_NB)*(int)sizeof(double);
    if (use_lds) CKL(hipFuncSetAttribute((const void *)k_solve_t<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_small));
    else {
        CKL(hipFuncSetAttribute((const void *)k_solve_t<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_diag));
        CKL(hipFuncSetAttribute((const void *)k_chol_panel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_panel));
        CKL(hipFuncSetAttribute((const void *)k_chol_update, hipFuncAttributeMaxDynamicSharedMemorySize, lds_upd));
        CKL(hipFuncSetAttribute((const void *)k_chol_backsub, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bs));
    }
    auto solve = [&]() {
        if (use_lds) { hipLaunchKernelGGL(k_solve_t<false>, dim3(1), dim3(SOLVE_THREADS), (int)lds_small, c->stream, W, 0); return; }
        const int bw = n6;                                             // dense: every row below a block can be non-zero
        hipLaunchKernelGGL(k_chol_rhs, dim3((n6 + 255)/256), dim3(256), 0, c->stream, W);
        for (int j0 = 0; j0 < n6; j0 += CH_NB) {
            hipLaunchKernelGGL(k_solve_t<true>, dim3(1), dim3(SOLVE_THREADS), lds_diag, c->stream, W, j0);
            const int wr = std::max(0, std::min(bw, n6 - (j0 + 6)));
            hipLaunchKernelGGL(k_chol_panel, dim3(wr/64 + 1), dim3(CH_T), lds_panel, c->stream, W, j0, bw);
            const int nt = (wr + 1 + 63)/64;
            if (wr > 0) hipLaunchKernelGGL(k_chol_update, dim3(nt*(nt + 1)/2), dim3(CH_T), lds_upd, c->stream, W, j0, bw);
        }
        hipLaunchKernelGGL(k_chol_backsub, dim3(1), dim3(1024), lds_bs, c->stream, W, bw);
    };
    // ---- LM: max_it rounds + one to close the last accepted step (gradient test, iteration limit)
    for (int it = 0; it <= o->max_it; it++) {
        hipLaunchKernelGGL(k_pg_linearize, dim3((E + 63)/64), dim3(64), 0, c->stream, P);
        CKL(hipMemsetAsync(W.S, 0, sizeof(double)*(size_t)n6*n6, c->stream));
        hipLaunchKernelGGL(k_pg_assemble, dim3(nf + npair + 1), dim3(64), 0, c->stream, P, *o);
        hipLaunchKernelGGL(k_pg_pre, dim3(1), dim3(1024), 0, c->stream, P, *o);
        if (it == o->max_it) break;
        solve();
        hipLaunchKernelGGL(k_pg_candidate, dim3((nf + 255)/256), dim3(256), 0, c->stream, P);
        hipLaunchKernelGGL(k_pg_trial, dim3((E + 63)/64), dim3(64), 0, c->stream, P);
        hipLaunchKernelGGL(k_pg_decide, dim3(1), dim3(1024), 0, c->stream, P, *o);
    }
    LmState st;
    CKL(hipMemcpyAsync(h + o_st, d + o_st, sizeof(LmState), hipMemcpyDeviceToHost, c->stream));
    CKL(hipStreamSynchronize(c->stream)); CKL(hipGetLastError());
    memcpy(&st, h + o_st, sizeof(st));
    CKL(hipMemcpy(p->pose, st.cur ? P.x[1] : P.x[0], 64*(size_t)N, hipMemcpyDeviceToHost));
    r->iters = st.it; r->accepted = st.accepted; r->termination = st.term; r->cost0 = st.cost0; r->cost1 = st.x_cost;
    r->t_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return st.term == 5 ? TSLOOP_ERR_NUMERIC : TSLOOP_OK;
}

}  // extern "C"
