// Device-side math of the BA hot path (gfx950, wave64).  fp64 throughout, as the reference.
//
// Residual models follow TextSLAM's cost functors (include/auto_BAScene.h:28-87, auto_PoseOptimScene.h:29-88,
// nume_BAText.h:28-94, nume_PoseOptimText.h:28-79, ModelTool.hpp:164-171); Jacobians are the analytic tangent-space
// forms of ceres::QuaternionParameterization (left perturbation, half angle), SURVEY.md Appendix A.
//
// Design note (not in the reference): with P = R_cr (X_r - t_r) + t_c the host-pose Jacobian of every block is
//     J_host = -J_target * blkdiag(R_cr, R_cr)
// so a block only ever forms J_target^T J_target, J_target^T r and J_target^T J_landmark; everything that involves the
// host pose is recovered once per (target, host) keyframe pair from the pair sums.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TS_DEV __device__ __forceinline__

struct Pose {
    double R[9];
    double t[3];
};

// 1/x: v_rcp_f64 (2^-23) + two Newton steps instead of the IEEE division sequence (v_div_scale x2, v_rcp, 6 fma, v_div_fmas,
// v_div_fixup ~ 130 cycles of quarter-rate instructions per wave).  The linearisation is bound by instruction issue and a
// photometric tap has four of these.  Result within 1 ulp; x = 0 / inf / nan fall back to the seed (inf / 0 / nan as IEEE).
TS_DEV double ts_rcp(double x) {
    const double r0 = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r0, 1.0), r = fma(r0, e, r0);
    e = fma(-x, r, 1.0); r = fma(r, e, r);
    return r == r ? r : r0;
}

TS_DEV void quat_to_R(const double q_[4], double R[9]) {       // Eigen normalized() + toRotationMatrix()
    double in = ts_rcp(sqrt(q_[0]*q_[0] + q_[1]*q_[1] + q_[2]*q_[2] + q_[3]*q_[3]));
    double w = q_[0]*in, x = q_[1]*in, y = q_[2]*in, z = q_[3]*in;
    double tx = 2*x, ty = 2*y, tz = 2*z;
    double twx = tx*w, twy = ty*w, twz = tz*w, txx = tx*x, txy = ty*x, txz = tz*x, tyy = ty*y, tyz = tz*y, tzz = tz*z;
    R[0] = 1-(tyy+tzz); R[1] = txy-twz;     R[2] = txz+twy;
    R[3] = txy+twz;     R[4] = 1-(txx+tzz); R[5] = tyz-twx;
    R[6] = txz-twy;     R[7] = tyz+twx;     R[8] = 1-(txx+tyy);
}
TS_DEV void load_pose(const double *__restrict__ p, Pose &P) {
    double q[4] = { p[0], p[1], p[2], p[3] };
    quat_to_R(q, P.R);
    P.t[0] = p[4]; P.t[1] = p[5]; P.t[2] = p[6];
}
TS_DEV void mat3_mulT(const double A[9], const double B[9], double C[9]) {     // A * B^T
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[i*3+j] = A[i*3]*B[j*3] + A[i*3+1]*B[j*3+1] + A[i*3+2]*B[j*3+2];
}
TS_DEV void mat3_mul(const double A[9], const double B[9], double C[9]) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C[i*3+j] = A[i*3]*B[j] + A[i*3+1]*B[3+j] + A[i*3+2]*B[6+j];
}
TS_DEV void mat3_vec(const double A[9], const double v[3], double o[3]) {
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = A[i*3]*v[0] + A[i*3+1]*v[1] + A[i*3+2]*v[2];
}
TS_DEV void mat3T_vec(const double A[9], const double v[3], double o[3]) {     // A^T v
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = A[i]*v[0] + A[3+i]*v[1] + A[6+i]*v[2];
}

// Relative transform of a (target, host) pair: P - t_c = Rcr * X_r + tq,   tq = -Rcr t_r   (host in window)
//                                                                          tq =  Rc t_wr   (frozen text host, T_wr given)
struct PairT {
    double Rcr[9];
    double tq[3];
};
TS_DEV void pair_from_poses(const Pose &C, const Pose &Hst, PairT &T) {        // T_cr = T_cw * T_rw^-1
    mat3_mulT(C.R, Hst.R, T.Rcr);
    double tmp[3]; mat3_vec(T.Rcr, Hst.t, tmp);
    T.tq[0] = -tmp[0]; T.tq[1] = -tmp[1]; T.tq[2] = -tmp[2];
}
TS_DEV void pair_from_Trw(const Pose &C, const double *__restrict__ Trw, PairT &T) {    // frozen scene host: T_rw (3x4)
    Pose Hst;
#pragma unroll
    for (int i = 0; i < 3; i++) { Hst.R[i*3] = Trw[i*4]; Hst.R[i*3+1] = Trw[i*4+1]; Hst.R[i*3+2] = Trw[i*4+2]; Hst.t[i] = Trw[i*4+3]; }
    pair_from_poses(C, Hst, T);
}
TS_DEV void pair_from_Twr(const Pose &C, const double *__restrict__ Twr, PairT &T) {    // frozen text host: T_cr = T_cw * T_wr
    double Rwr[9], twr[3];
#pragma unroll
    for (int i = 0; i < 3; i++) { Rwr[i*3] = Twr[i*4]; Rwr[i*3+1] = Twr[i*4+1]; Rwr[i*3+2] = Twr[i*4+2]; twr[i] = Twr[i*4+3]; }
    mat3_mul(C.R, Rwr, T.Rcr);
    mat3_vec(C.R, twr, T.tq);
}

// Huber (ceres::HuberLoss) on s = |r|^2: returns rho(s), *w = rho'(s) (the IRLS weight; Corrector scales r and J by sqrt(w))
TS_DEV double huber(double s, double delta, double &w) {
    double b = delta*delta;
    if (s > b) { double r = sqrt(s); w = delta*ts_rcp(r); return 2.0*delta*r - b; }
    w = 1.0; return s;
}

// ---- scene reprojection block: residual, target-pose Jacobian rows (2x6), inverse-depth Jacobian (2)
TS_DEV void scene_block(const PairT &T, const double tc[3], double mx, double my, double rho, double u_obs, double v_obs,
                        double fx, double fy, double cx, double cy, double wx, double wy,
                        double r[2], double jt[2][6], double jl[2]) {
    double ir = ts_rcp(rho);
    double X[3] = { ir*mx, ir*my, ir };
    double Pm[3]; mat3_vec(T.Rcr, X, Pm);
    Pm[0] += T.tq[0]; Pm[1] += T.tq[1]; Pm[2] += T.tq[2];
    double Px = Pm[0] + tc[0], Py = Pm[1] + tc[1], Pz = Pm[2] + tc[2];
    double iz = ts_rcp(Pz);
    r[0] = (fx*Px*iz + cx - u_obs)*wx;
    r[1] = (fy*Py*iz + cy - v_obs)*wy;
    double a0[3] = { wx*fx*iz, 0.0, -wx*fx*Px*iz*iz };
    double a1[3] = { 0.0, wy*fy*iz, -wy*fy*Py*iz*iz };
    // J_target row = [ 2 (Pm x a)^T | a^T ]
    jt[0][0] = 2.0*(Pm[1]*a0[2] - Pm[2]*a0[1]); jt[0][1] = 2.0*(Pm[2]*a0[0] - Pm[0]*a0[2]); jt[0][2] = 2.0*(Pm[0]*a0[1] - Pm[1]*a0[0]);
    jt[0][3] = a0[0]; jt[0][4] = a0[1]; jt[0][5] = a0[2];
    jt[1][0] = 2.0*(Pm[1]*a1[2] - Pm[2]*a1[1]); jt[1][1] = 2.0*(Pm[2]*a1[0] - Pm[0]*a1[2]); jt[1][2] = 2.0*(Pm[0]*a1[1] - Pm[1]*a1[0]);
    jt[1][3] = a1[0]; jt[1][4] = a1[1]; jt[1][5] = a1[2];
    // d/d rho: -a^T Rcr m / rho^2, with Rcr m / rho = Pm - tq
    double Rm[3] = { (Pm[0] - T.tq[0])*ir, (Pm[1] - T.tq[1])*ir, (Pm[2] - T.tq[2])*ir };
    jl[0] = -(a0[0]*Rm[0] + a0[2]*Rm[2]);
    jl[1] = -(a1[1]*Rm[1] + a1[2]*Rm[2]);
}
TS_DEV void scene_residual(const PairT &T, const double tc[3], double mx, double my, double rho, double u_obs, double v_obs,
                           double fx, double fy, double cx, double cy, double wx, double wy, double r[2]) {
    double ir = ts_rcp(rho);
    double X[3] = { ir*mx, ir*my, ir };
    double Pm[3]; mat3_vec(T.Rcr, X, Pm);
    double Px = Pm[0] + T.tq[0] + tc[0], Py = Pm[1] + T.tq[1] + tc[1], Pz = Pm[2] + T.tq[2] + tc[2];
    double iz = ts_rcp(Pz);
    r[0] = (fx*Px*iz + cx - u_obs)*wx;
    r[1] = (fy*Py*iz + cy - v_obs)*wy;
}

// INTERVAL8 pattern, src/tool.cc:1550-1557
__device__ __constant__ double TAP_DX[8] = { 0, 2, 1, 0, -1, -2, -1, 0 };
__device__ __constant__ double TAP_DY[8] = { 0, 0, -1, -2, -1, 0, 1, 2 };

// bilinear tap with the reference's in/out rule (nume_BAText.h:67-82) + bilinear gradient
TS_DEV double bilinear_tap(const uint8_t *__restrict__ img, int w, int h, double u, double v, double &gu, double &gv) {
    double uf = floor(u), vf = floor(v);
    int iu = (int)uf, iv = (int)vf;
    gu = 0.0; gv = 0.0;
    if (iu < 0 || iv < 0 || (int)ceil(u) >= w || (int)ceil(v) >= h) return 0.0;
    const uint8_t *p = img + (size_t)iv*w + iu;
    double su = u - uf, sv = v - vf;
    double I00 = p[0];
    double I01 = (iu + 1 < w) ? (double)p[1] : 0.0;
    double I10 = (iv + 1 < h) ? (double)p[w] : 0.0;
    double I11 = (iu + 1 < w && iv + 1 < h) ? (double)p[w + 1] : 0.0;
    gu = (1.0 - sv)*(I01 - I00) + sv*(I11 - I10);
    gv = (1.0 - su)*(I10 - I00) + su*(I11 - I01);
    return (1.0 - su)*(1.0 - sv)*I00 + su*(1.0 - sv)*I01 + (1.0 - su)*sv*I10 + su*sv*I11;
}

// one photometric tap, split so that the image fetches of all taps of a feature can be in flight together:
//   tap_fetch   projects the tap and loads its 2x2 pixel neighbourhood (zeros outside the image: same rule as bilinear_tap)
//   text_tap_px redoes the (cheap, bit-identical) projection and finishes residual, target-pose row (6), theta row (3)
struct TapPx { int I00, I01, I10, I11; };
TS_DEV void tap_project(const PairT &T, const double tc[3], const double th[3], double mx, double my,
                        double fx, double fy, double cx, double cy, double Rm[3], double &is, double Pm[3], double P[3], double &u, double &v) {
    const double s = -(mx*th[0] + my*th[1] + th[2]);         // rho(m) = -m^T theta, ModelTool.hpp:167
    const double m[3] = { mx, my, 1.0 };
    mat3_vec(T.Rcr, m, Rm);
    is = ts_rcp(s);
    Pm[0] = Rm[0]*is + T.tq[0]; Pm[1] = Rm[1]*is + T.tq[1]; Pm[2] = Rm[2]*is + T.tq[2];
    P[0] = Pm[0] + tc[0]; P[1] = Pm[1] + tc[1]; P[2] = Pm[2] + tc[2];
    const double iz = ts_rcp(P[2]);                          // one reciprocal for both coordinates (and the Jacobian)
    u = fx*P[0]*iz + cx; v = fy*P[1]*iz + cy;
}
TS_DEV TapPx tap_fetch(const PairT &T, const double tc[3], const double th[3], double mx, double my,
                       double fx, double fy, double cx, double cy, const uint8_t *__restrict__ img, int w, int h) {
    double Rm[3], is, Pm[3], P[3], u, v;
    tap_project(T, tc, th, mx, my, fx, fy, cx, cy, Rm, is, Pm, P, u, v);
    TapPx px = { 0, 0, 0, 0 };
    const double uf = floor(u), vf = floor(v);
    const int iu = (int)uf, iv = (int)vf;
    if (iu < 0 || iv < 0 || (int)ceil(u) >= w || (int)ceil(v) >= h) return px;
    // two 2-byte fetches instead of four 1-byte ones (the images are one slab: the byte after a row end is mapped memory)
    const uint8_t *p = img + (size_t)iv*w + iu;
    typedef unsigned short u16u __attribute__((aligned(1)));
    const unsigned r0 = *(const u16u *)p;
    const unsigned r1 = (iv + 1 < h) ? (unsigned)*(const u16u *)(p + w) : 0u;
    const bool in_u = iu + 1 < w;
    px.I00 = r0 & 0xff; px.I01 = in_u ? (r0 >> 8) : 0;
    px.I10 = r1 & 0xff; px.I11 = in_u ? (r1 >> 8) : 0;
    return px;
}
TS_DEV double text_tap_px(const PairT &T, const double tc[3], const double th[3], double mx, double my,
                          double fx, double fy, double cx, double cy, const TapPx &px, int w, int h,
                          double mu, double sigma, double inv_sigma, double ref, double wT, bool want_j, double jt[6], double jl[3]) {
    double Rm[3], is, Pm[3], P[3], u, v;
    tap_project(T, tc, th, mx, my, fx, fy, cx, cy, Rm, is, Pm, P, u, v);
    double gu = 0.0, gv = 0.0, I = 0.0;
    const double uf = floor(u), vf = floor(v);
    const int iu = (int)uf, iv = (int)vf;
    if (!(iu < 0 || iv < 0 || (int)ceil(u) >= w || (int)ceil(v) >= h)) {
        const double su = u - uf, sv = v - vf;
        const double I00 = px.I00, I01 = px.I01, I10 = px.I10, I11 = px.I11;
        gu = (1.0 - sv)*(I01 - I00) + sv*(I11 - I10);
        gv = (1.0 - su)*(I10 - I00) + su*(I11 - I01);
        I = (1.0 - su)*(1.0 - sv)*I00 + su*(1.0 - sv)*I01 + (1.0 - su)*sv*I10 + su*sv*I11;
    }
    const double r = ((I - mu)*inv_sigma - ref)*wT;         // nume_BAText.h:86-87
    if (want_j) {
        const double Px = P[0], Py = P[1], Pz = P[2];
        double iz = ts_rcp(Pz);
        double g0 = wT*inv_sigma*gu, g1 = wT*inv_sigma*gv;
        double a[3] = { g0*fx*iz, g1*fy*iz, -(g0*fx*Px + g1*fy*Py)*iz*iz };
        jt[0] = 2.0*(Pm[1]*a[2] - Pm[2]*a[1]); jt[1] = 2.0*(Pm[2]*a[0] - Pm[0]*a[2]); jt[2] = 2.0*(Pm[0]*a[1] - Pm[1]*a[0]);
        jt[3] = a[0]; jt[4] = a[1]; jt[5] = a[2];
        double c = (a[0]*Rm[0] + a[1]*Rm[1] + a[2]*Rm[2])*is*is;
        jl[0] = c*mx; jl[1] = c*my; jl[2] = c;
    }
    return r;
}
TS_DEV double text_tap(const PairT &T, const double tc[3], const double th[3], double mx, double my,
                       double fx, double fy, double cx, double cy, const uint8_t *__restrict__ img, int w, int h,
                       double mu, double sigma, double inv_sigma, double ref, double wT, bool want_j, double jt[6], double jl[3]) {
    const TapPx px = tap_fetch(T, tc, th, mx, my, fx, fy, cx, cy, img, w, h);
    return text_tap_px(T, tc, th, mx, my, fx, fy, cx, cy, px, w, h, mu, sigma, inv_sigma, ref, wT, want_j, jt, jl);
}

// ceres::QuaternionParameterization::Plus
TS_DEV void quat_plus(const double x[4], const double d[3], double o[4]) {
    double nd = sqrt(d[0]*d[0] + d[1]*d[1] + d[2]*d[2]);
    if (nd > 0.0) {
        double s = sin(nd)/nd, c = cos(nd);
        double z1 = s*d[0], z2 = s*d[1], z3 = s*d[2];
        o[0] = c*x[0] - z1*x[1] - z2*x[2] - z3*x[3];
        o[1] = c*x[1] + z1*x[0] + z2*x[3] - z3*x[2];
        o[2] = c*x[2] - z1*x[3] + z2*x[0] + z3*x[1];
        o[3] = c*x[3] + z1*x[2] - z2*x[1] + z3*x[0];
    } else { o[0] = x[0]; o[1] = x[1]; o[2] = x[2]; o[3] = x[3]; }
}

// symmetric 6x6 packed upper index: (r,c) r<=c -> r*6 - r*(r-1)/2 + (c-r)
TS_DEV constexpr int sym6(int r, int c) { return r <= c ? r*6 - r*(r-1)/2 + (c - r) : c*6 - c*(c-1)/2 + (r - c); }

// inverse of a symmetric 3x3 (v = xx,xy,xz,yy,yz,zz); returns false if not positive definite
TS_DEV bool inv_sym3(const double v[6], double o[6]) {
    double a = v[0], b = v[1], c = v[2], e = v[3], f = v[4], i = v[5];
    double A = e*i - f*f, B = -(b*i - c*f), Cc = b*f - c*e;
    double det = a*A + b*B + c*Cc;
    if (!(det > 0.0) || !(a > 0.0) || !(a*e - b*b > 0.0)) return false;
    double id = 1.0/det;
    o[0] = A*id; o[1] = B*id; o[2] = Cc*id; o[3] = (a*i - c*c)*id; o[4] = -(a*f - b*c)*id; o[5] = (a*e - b*b)*id;
    return true;
}

// Wave-wide sum of N per-lane accumulators through an LDS transpose: lane l (< N) returns the total of acc[l].
// lds must hold N*65 doubles (row stride 65 keeps both phases conflict-free for ds_write_b64 / ds_read_b64).
template <int N>
TS_DEV double wave_sum_to_lane(const double (&acc)[N], double *lds, int lane) {
#pragma unroll
    for (int i = 0; i < N; i++) lds[i*65 + lane] = acc[i];
    __syncthreads();                             // single-wave workgroups: an s_barrier of one wave
    double s = 0.0;
    if (lane < N) {
        const double *row = lds + lane*65;
#pragma unroll 16
        for (int k = 0; k < 64; k++) s += row[k];
    }
    __syncthreads();
    return s;
}
