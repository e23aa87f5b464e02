/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  See tsba_oracle.h (PARITY UNPINNED).
 *
 * Plain-C fp64 restatement of TextSLAM's BA / pose-optimisation path:
 *   residual models    include/auto_BAScene.h:28-87, auto_BASceneNW.h:28-84, auto_PoseOptimScene.h:29-88,
 *                      nume_BAText.h:28-94, nume_PoseOptimText.h:28-79, ModelTool.hpp:164-171, rotation.h:525-562
 *   problem assembly   src/optimizer.cc:1060-1327 (PyrPoseOptim), :1330-1698 (PyrBA), :1701-1851 (PyrGlobalBA)
 *   mu / sigma         src/tool.cc:1178-1262 (+ cv::fillPoly scan conversion, recalled from OpenCV 3.x drawing.cpp)
 *   solver             Ceres 1.x trust_region_minimizer.cc / levenberg_marquardt_strategy.cc / corrector.cc /
 *                      numeric_diff.h / local_parameterization.cc behaviour (recalled)
 * Written for clarity, not speed: one residual block at a time, explicit Jacobian blocks, explicit
 * normal equations, landmark elimination, dense Cholesky.
 */
#include "tsba_oracle.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>

/* ------------------------------------------------------------------ small linear algebra */
static void quat_normalize(const double q[4], double o[4]) {          /* Eigen::Quaterniond::normalized() */
    double n = sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
    o[0] = q[0]/n; o[1] = q[1]/n; o[2] = q[2]/n; o[3] = q[3]/n;
}
static void quat_to_R(const double q[4], double R[9]) {               /* Eigen toRotationMatrix(), q = (w,x,y,z) */
    double w = q[0], x = q[1], y = q[2], z = q[3];
    double tx = 2*x, ty = 2*y, tz = 2*z;
    double twx = tx*w, twy = ty*w, twz = tz*w, txx = tx*x, txy = ty*x, txz = tz*x, tyy = ty*y, tyz = tz*y, tzz = tz*z;
    R[0] = 1-(tyy+tzz); R[1] = txy-twz;     R[2] = txz+twy;
    R[3] = txy+twz;     R[4] = 1-(txx+tzz); R[5] = tyz-twx;
    R[6] = txz-twy;     R[7] = tyz+twx;     R[8] = 1-(txx+tyy);
}
static void R_to_quat(const double R[9], double q[4]) {               /* Eigen Quaternion(Matrix3) */
    double t = R[0] + R[4] + R[8];
    double c[4]; /* x y z w */
    if (t > 0) {
        t = sqrt(t + 1.0);
        c[3] = 0.5*t; t = 0.5/t;
        c[0] = (R[7] - R[5])*t; c[1] = (R[2] - R[6])*t; c[2] = (R[3] - R[1])*t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i*3+i]) i = 2;
        int j = (i+1)%3, k = (j+1)%3;
        t = sqrt(R[i*3+i] - R[j*3+j] - R[k*3+k] + 1.0);
        c[i] = 0.5*t; t = 0.5/t;
        c[3] = (R[k*3+j] - R[j*3+k])*t;
        c[j] = (R[j*3+i] + R[i*3+j])*t;
        c[k] = (R[k*3+i] + R[i*3+k])*t;
    }
    q[0] = c[3]; q[1] = c[0]; q[2] = c[1]; q[3] = c[2];
}
static void ceres_quat_rotate(const double q[4], const double pt[3], double r[3]) {  /* rotation.h:525-562 */
    double scale = 1.0 / sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
    double u[4] = { scale*q[0], scale*q[1], scale*q[2], scale*q[3] };
    double t2 = u[0]*u[1], t3 = u[0]*u[2], t4 = u[0]*u[3], t5 = -u[1]*u[1], t6 = u[1]*u[2];
    double t7 = u[1]*u[3], t8 = -u[2]*u[2], t9 = u[2]*u[3], t1 = -u[3]*u[3];
    r[0] = 2*((t8 + t1)*pt[0] + (t6 - t4)*pt[1] + (t3 + t7)*pt[2]) + pt[0];
    r[1] = 2*((t4 + t6)*pt[0] + (t5 + t1)*pt[1] + (t9 - t2)*pt[2]) + pt[1];
    r[2] = 2*((t7 - t3)*pt[0] + (t2 + t9)*pt[1] + (t5 + t8)*pt[2]) + pt[2];
}
static void mat3_mul(const double A[9], const double B[9], double C[9]) {
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
        C[i*3+j] = A[i*3+0]*B[0*3+j] + A[i*3+1]*B[1*3+j] + A[i*3+2]*B[2*3+j];
}
static void mat3_mulT(const double A[9], const double B[9], double C[9]) {   /* A * B^T */
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++)
        C[i*3+j] = A[i*3+0]*B[j*3+0] + A[i*3+1]*B[j*3+1] + A[i*3+2]*B[j*3+2];
}
static void mat3_vec(const double A[9], const double v[3], double o[3]) {
    for (int i = 0; i < 3; i++) o[i] = A[i*3]*v[0] + A[i*3+1]*v[1] + A[i*3+2]*v[2];
}
static void skew(const double v[3], double S[9]) {
    S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
static void level_K(const tsba_problem *p, int level, double K[4]) {   /* optimizer.cc:43-52: K_l = K_{l-1} * (1/scale) */
    K[0] = p->K[0]; K[1] = p->K[1]; K[2] = p->K[2]; K[3] = p->K[3];
    for (int l = 0; l < level; l++) { K[0] *= 0.5; K[1] *= 0.5; K[2] *= 0.5; K[3] *= 0.5; }
}
static const double TAP_DX[8] = { 0, 2, 1, 0, -1, -2, -1, 0 };           /* tool.cc:1550-1557 INTERVAL8 */
static const double TAP_DY[8] = { 0, 0, -1, -2, -1, 0, 1, 2 };

/* ------------------------------------------------------------------ cost functors (literal) */
/* auto_BAScene::operator() / auto_BASceneNW (w = 1) / auto_PoseOptimScene (host quaternion from T_rw) */
static void f_scene(const double qcw[4], const double tcw[3], const double qrw[4], const double trw[3], double rho,
                    const double ray[3], const double K[4], double wx, double wy, double u_obs, double v_obs, double r[2]) {
    double qwr[4] = { qrw[0], -qrw[1], -qrw[2], -qrw[3] };
    double qcr[4];
    qcr[0] = qcw[0]*qwr[0] - qcw[1]*qwr[1] - qcw[2]*qwr[2] - qcw[3]*qwr[3];
    qcr[1] = qcw[0]*qwr[1] + qcw[1]*qwr[0] + qcw[2]*qwr[3] - qcw[3]*qwr[2];
    qcr[2] = qcw[0]*qwr[2] - qcw[1]*qwr[3] + qcw[2]*qwr[0] + qcw[3]*qwr[1];
    qcr[3] = qcw[0]*qwr[3] + qcw[1]*qwr[2] - qcw[2]*qwr[1] + qcw[3]*qwr[0];
    double tmp[3], tcr[3];
    ceres_quat_rotate(qcr, trw, tmp);
    tcr[0] = -tmp[0] + tcw[0]; tcr[1] = -tmp[1] + tcw[1]; tcr[2] = -tmp[2] + tcw[2];
    double pp[3] = { 1.0/rho*ray[0], 1.0/rho*ray[1], 1.0/rho*ray[2] };
    double qp[3];
    ceres_quat_rotate(qcr, pp, qp);
    double u = K[0]*(qp[0] + tcr[0])/(qp[2] + tcr[2]) + K[2];
    double v = K[1]*(qp[1] + tcr[1])/(qp[2] + tcr[2]) + K[3];
    r[0] = (u - u_obs)*wx;
    r[1] = (v - v_obs)*wy;
}

/* bilinear tap with the reference's in/out rule (nume_BAText.h:67-82, tool.cc:1150-1176);
 * also returns the bilinear gradient (Appendix A of SURVEY.md) */
static double bilinear(const uint8_t *img, int w, int h, double u, double v, double *gu, double *gv) {
    int uf = (int)floor(u), vf = (int)floor(v), uc = (int)ceil(u), vc = (int)ceil(v);
    if (gu) { *gu = 0; *gv = 0; }
    if (uf < 0 || vf < 0 || uc >= w || vc >= h) return 0.0;
    const uint8_t *ptr = img + (size_t)vf*w + uf;
    double su = u - uf, sv = v - vf;
    double wtl = (1.0 - su)*(1.0 - sv), wtr = su*(1.0 - sv), wbl = (1.0 - su)*sv, wbr = su*sv;
    /* the reference reads ptr[1], ptr[stride], ptr[stride+1] even when their weight is 0 (uc==uf);
       uc<w / vc<h only guarantees ceil() is inside -- when u is an integer ptr[1] may be one past the row end but
       still inside the buffer except on the very last pixel; guard that single case by weight */
    double I00 = ptr[0];
    double I01 = (uf + 1 < w) ? ptr[1] : 0.0;
    double I10 = (vf + 1 < h) ? ptr[w] : 0.0;
    double I11 = (uf + 1 < w && vf + 1 < h) ? ptr[w + 1] : 0.0;
    if (gu) {
        *gu = (1.0 - sv)*(I01 - I00) + sv*(I11 - I10);
        *gv = (1.0 - su)*(I10 - I00) + su*(I11 - I01);
    }
    return wtl*I00 + wtr*I01 + wbl*I10 + wbr*I11;
}

/* shared tail of nume_BAText / nume_PoseOptimText: 8 taps through T_cr (R,t), plane theta */
static void f_text_taps(const double Rcr[9], const double tcr[3], const double theta[3], const double fu, const double fv,
                        const double Kl[4], const uint8_t *img, int w, int h, const double ref[8],
                        double mu, double sigma, double wT, double r[8]) {
    for (int k = 0; k < 8; k++) {
        double ray[3] = { (fu + TAP_DX[k] - Kl[2])/Kl[0], (fv + TAP_DY[k] - Kl[3])/Kl[1], 1.0 };  /* tool.cc:1561 */
        double rho = -(ray[0]*theta[0] + ray[1]*theta[1] + ray[2]*theta[2]);                    /* ModelTool.hpp:167 */
        double Rr[3]; mat3_vec(Rcr, ray, Rr);
        double P[3] = { Rr[0]/rho + tcr[0], Rr[1]/rho + tcr[1], Rr[2]/rho + tcr[2] };
        double u = Kl[0]*P[0]/P[2] + Kl[2];
        double v = Kl[1]*P[1]/P[2] + Kl[3];
        double I = bilinear(img, w, h, u, v, NULL, NULL);
        if (sigma != 0) r[k] = ((I - mu)/sigma - ref[k])*wT; else r[k] = 0.0;
    }
}
/* nume_BAText::operator() (include/nume_BAText.h:28-94) */
static void f_ba_text(const double qcw_[4], const double tcw[3], const double qrw_[4], const double trw[3], const double theta[3],
                      double fu, double fv, const double Kl[4], const uint8_t *img, int w, int h, const double ref[8],
                      double mu, double sigma, double wT, double r[8]) {
    double qcw[4], qrw[4], Rcw[9], Rrw[9], Rcr[9], tcr[3], tmp[3];
    quat_normalize(qcw_, qcw); quat_to_R(qcw, Rcw);
    quat_normalize(qrw_, qrw); quat_to_R(qrw, Rrw);
    mat3_mulT(Rcw, Rrw, Rcr);                     /* T_cr = T_cw * T_rw^-1 */
    mat3_vec(Rcr, trw, tmp);
    tcr[0] = tcw[0] - tmp[0]; tcr[1] = tcw[1] - tmp[1]; tcr[2] = tcw[2] - tmp[2];
    f_text_taps(Rcr, tcr, theta, fu, fv, Kl, img, w, h, ref, mu, sigma, wT, r);
}
/* nume_PoseOptimText::operator() (include/nume_PoseOptimText.h:28-79): T_cr = T_cw * T_wr */
static void f_pose_text(const double q_[4], const double t[3], const double Twr[12], const double theta[3],
                        double fu, double fv, const double Kl[4], const uint8_t *img, int w, int h, const double ref[8],
                        double mu, double sigma, double wT, double r[8]) {
    double q[4], Rcw[9], Rwr[9], twr[3], Rcr[9], tcr[3], tmp[3];
    quat_normalize(q_, q); quat_to_R(q, Rcw);
    for (int i = 0; i < 3; i++) { Rwr[i*3] = Twr[i*4]; Rwr[i*3+1] = Twr[i*4+1]; Rwr[i*3+2] = Twr[i*4+2]; twr[i] = Twr[i*4+3]; }
    mat3_mul(Rcw, Rwr, Rcr);
    mat3_vec(Rcw, twr, tmp);
    tcr[0] = tmp[0] + t[0]; tcr[1] = tmp[1] + t[1]; tcr[2] = tmp[2] + t[2];
    f_text_taps(Rcr, tcr, theta, fu, fv, Kl, img, w, h, ref, mu, sigma, wT, r);
}

/* ------------------------------------------------------------------ cv::fillPoly restatement (recalled) */
typedef struct { long long x, dx; int y0, y1; } poly_edge;

static int clip_line(long long W, long long H, long long *x1, long long *y1, long long *x2, long long *y2) {
    long long right = W - 1, bottom = H - 1;
    if (W <= 0 || H <= 0) return 0;
    int c1 = (*x1 < 0) + (*x1 > right)*2 + (*y1 < 0)*4 + (*y1 > bottom)*8;
    int c2 = (*x2 < 0) + (*x2 > right)*2 + (*y2 < 0)*4 + (*y2 > bottom)*8;
    if ((c1 & c2) == 0 && (c1 | c2) != 0) {
        long long a;
        if (c1 & 12) { a = c1 < 8 ? 0 : bottom; *x1 += (long long)((double)(a - *y1)*(double)(*x2 - *x1)/(double)(*y2 - *y1)); *y1 = a; c1 = (*x1 < 0) + (*x1 > right)*2; }
        if (c2 & 12) { a = c2 < 8 ? 0 : bottom; *x2 += (long long)((double)(a - *y2)*(double)(*x2 - *x1)/(double)(*y2 - *y1)); *y2 = a; c2 = (*x2 < 0) + (*x2 > right)*2; }
        if ((c1 & c2) == 0 && (c1 | c2) != 0) {
            if (c1) { a = c1 == 1 ? 0 : right; *y1 += (long long)((double)(a - *x1)*(double)(*y2 - *y1)/(double)(*x2 - *x1)); *x1 = a; c1 = 0; }
            if (c2) { a = c2 == 1 ? 0 : right; *y2 += (long long)((double)(a - *x2)*(double)(*y2 - *y1)/(double)(*x2 - *x1)); *x2 = a; c2 = 0; }
        }
    }
    return (c1 | c2) == 0;
}
/* cv::LineIterator (8-connected, left_to_right) */
static void draw_line8(int W, int H, long long x1, long long y1, long long x2, long long y2, uint8_t *mask) {
    if ((unsigned long long)x1 >= (unsigned long long)W || (unsigned long long)x2 >= (unsigned long long)W ||
        (unsigned long long)y1 >= (unsigned long long)H || (unsigned long long)y2 >= (unsigned long long)H) {
        if (!clip_line(W, H, &x1, &y1, &x2, &y2)) return;
    }
    long long dx = x2 - x1, dy = y2 - y1;
    if (dx < 0) { dx = -dx; dy = -dy; x1 = x2; y1 = y2; }          /* left to right: start from the leftmost end */
    long long sx = 1, sy = dy < 0 ? -1 : 1;
    if (dy < 0) dy = -dy;
    int steep = dy > dx;
    long long major = steep ? dy : dx, minor = steep ? dx : dy;
    long long err = major - (minor + minor), plusDelta = major + major, minusDelta = -(minor + minor);
    long long count = major + 1;
    long long x = x1, y = y1;
    for (long long i = 0; i < count; i++) {
        if (x >= 0 && x < W && y >= 0 && y < H) mask[y*W + x] = 1;
        int neg = err < 0;
        err += minusDelta + (neg ? plusDelta : 0);
        /* minusStep = step along the major axis, plusStep = step along the minor axis */
        if (steep) { y += sy; if (neg) x += sx; } else { x += sx; if (neg) y += sy; }
    }
}
static int cmp_edges(const void *a, const void *b) {
    const poly_edge *e1 = (const poly_edge *)a, *e2 = (const poly_edge *)b;
    if (e1->y0 != e2->y0) return e1->y0 < e2->y0 ? -1 : 1;
    if (e1->x != e2->x) return e1->x < e2->x ? -1 : 1;
    if (e1->dx != e2->dx) return e1->dx < e2->dx ? -1 : 1;
    return 0;
}
void tsba_oracle_fillpoly4(int W, int H, const int *xy, uint8_t *mask) {
    const int XY_SHIFT = 16; const long long XY_ONE = 1 << 16;
    poly_edge edges[4]; int ne = 0;
    memset(mask, 0, (size_t)W*H);
    long long p0x = (long long)xy[6]*XY_ONE, p0y = xy[7];         /* (x * 2^16, not x << 16: vertices left of the image are negative) */
    for (int i = 0; i < 4; i++) {
        long long p1x = (long long)xy[2*i]*XY_ONE, p1y = xy[2*i+1];
        draw_line8(W, H, (p0x + (XY_ONE >> 1)) >> XY_SHIFT, p0y, (p1x + (XY_ONE >> 1)) >> XY_SHIFT, p1y, mask);
        if (p0y != p1y) {
            poly_edge e;
            if (p0y < p1y) { e.y0 = (int)p0y; e.y1 = (int)p1y; e.x = p0x; }
            else           { e.y0 = (int)p1y; e.y1 = (int)p0y; e.x = p1x; }
            e.dx = (p1x - p0x)/(p1y - p0y);
            edges[ne++] = e;
        }
        p0x = p1x; p0y = p1y;
    }
    if (ne < 2) return;
    int y_max = -2147483647, y_min = 2147483647;
    for (int i = 0; i < ne; i++) { if (edges[i].y0 < y_min) y_min = edges[i].y0; if (edges[i].y1 > y_max) y_max = edges[i].y1; }
    if (y_max < 0 || y_min >= H) return;
    qsort(edges, ne, sizeof(poly_edge), cmp_edges);
    if (y_max > H) y_max = H;
    /* Scanline: for each row, the active edges (y0 <= y < y1) at their current x, sorted by x, paired off.
       x advances by dx per row from y0 exactly as the incremental loop of FillEdgeCollection does. */
    for (int y = y_min; y < y_max; y++) {
        long long xs[4]; int na = 0;
        for (int i = 0; i < ne; i++)
            if (edges[i].y0 <= y && y < edges[i].y1) xs[na++] = edges[i].x + (long long)(y - edges[i].y0)*edges[i].dx;
        for (int i = 1; i < na; i++) { long long v = xs[i]; int j = i - 1; while (j >= 0 && xs[j] > v) { xs[j+1] = xs[j]; j--; } xs[j+1] = v; }
        if (y < 0) continue;
        for (int i = 0; i + 1 < na; i += 2) {
            int xa = (int)((xs[i] + XY_ONE - 1) >> XY_SHIFT), xb = (int)(xs[i+1] >> XY_SHIFT);
            if (xa < W && xb >= 0) {
                if (xa < 0) xa = 0;
                if (xb >= W) xb = W - 1;
                for (int x = xa; x <= xb; x++) mask[(size_t)y*W + x] = 1;
            }
        }
    }
}

/* tool::CalTextinfo + CalStatistics, src/tool.cc:1178-1262 */
int tsba_oracle_musigma(const uint8_t *img, int w, int h, const double *c, double *mu, double *sigma) {
    int xy[8];
    int xMin = w + 1, xMax = -1, yMin = h + 1, yMax = -1;
    for (int i = 0; i < 4; i++) {
        double cu = c[2*i], cv = c[2*i+1];
        xy[2*i] = (int)cu; xy[2*i+1] = (int)cv;                 /* cv::Point(double,double): truncation */
        if (cu > xMax) xMax = (int)ceil(cu);
        if (cu < xMin) xMin = (int)floor(cu);
        if (cv > yMax) yMax = (int)ceil(cv);
        if (cv < yMin) yMin = (int)floor(cv);
    }
    if (xMin < 0) xMin = 0;
    if (xMin >= w) xMin = w - 1;
    if (yMin < 0) yMin = 0;
    if (yMin >= h) yMin = h - 1;
    if (xMax >= w) xMax = w - 1;
    if (xMax < 0) xMax = 0;
    if (yMax >= h) yMax = h - 1;
    if (yMax < 0) yMax = 0;
    uint8_t *mask = (uint8_t *)malloc((size_t)w*h);
    tsba_oracle_fillpoly4(w, h, xy, mask);
    double sum = 0; long long n = 0;
    for (int r = yMin; r <= yMax; r++) for (int cc = xMin; cc <= xMax; cc++)
        if (mask[(size_t)r*w + cc]) { sum += img[(size_t)r*w + cc]; n++; }
    /* reference: n == 0 leaves mu/sigma uninitialised (UB); n == 1 gives sqrt(0/0).  Restated as "invalid => sigma = 0"
       which makes every residual of the pair 0 (nume_BAText.h:85-90). */
    if (n < 2) { free(mask); *mu = 0; *sigma = 0; return 0; }
    double m = sum/(double)n, ss = 0;
    for (int r = yMin; r <= yMax; r++) for (int cc = xMin; cc <= xMax; cc++)
        if (mask[(size_t)r*w + cc]) { double d = img[(size_t)r*w + cc] - m; ss += d*d; }
    free(mask);
    *mu = m; *sigma = sqrt(ss/(double)(n - 1));
    return *sigma != 0;
}

/* ------------------------------------------------------------------ residual blocks */
enum { BLK_SCENE_BA = 0, BLK_SCENE_POSE = 1, BLK_TEXT_BA = 2, BLK_TEXT_POSE = 3 };
typedef struct {
    int type, kf, host, lm;     /* lm = point index or text index */
    int nres;                   /* 2 or 8 */
    int src;                    /* scene: observation index in sobs[level]; text: feature index in tfeat[level] */
    int tobs;                   /* text: (KF,text) pair index */
    int fixed;                  /* all parameter blocks constant => not in the reduced program */
} blk_t;

typedef struct {
    const tsba_problem *p; const tsba_options *o;
    int level; double K0[4], Kl[4];
    blk_t *blk; int nblk, ns, nt;
    double *mu, *sigma;         /* per tobs */
    int *tobs_size;             /* blocks per tobs (vSizeEachObj) */
    uint8_t *kf_in, *kf_const;  /* FLAG_KFIN, constant */
    int *free_idx; int nf;      /* KF -> column block */
    int *pt_lm, *tx_lm; int nlm;/* landmark -> compact index (or -1) */
} pass_t;

static void pose_Rt(const double *pose, double R[9], double t[3]) {
    double q[4]; quat_normalize(pose, q); quat_to_R(q, R); t[0] = pose[4]; t[1] = pose[5]; t[2] = pose[6];
}

/* projected text box + mu/sigma for pair t, at the given parameters (GetProjText x4 + CalTextinfo) */
static void pair_musigma(const pass_t *P, int t, const double *pose, const double *theta, double *mu, double *sigma, double corners[8]) {
    const tsba_problem *p = P->p;
    int kf = p->tobs_kf[t], j = p->tobs_text[t], host = p->text_host[j];
    double Rc[9], tc[3], Rcr[9], tcr[3], tmp[3];
    pose_Rt(pose + 7*kf, Rc, tc);
    if (host >= 0) {                              /* tool.cc:1687-1728 */
        double Rr[9], tr[3]; pose_Rt(pose + 7*host, Rr, tr);
        mat3_mulT(Rc, Rr, Rcr); mat3_vec(Rcr, tr, tmp);
        tcr[0] = tc[0] - tmp[0]; tcr[1] = tc[1] - tmp[1]; tcr[2] = tc[2] - tmp[2];
    } else {                                      /* tool.cc:1655-1685 */
        const double *T = p->text_host_Twr + 12*j; double Rwr[9], twr[3];
        for (int i = 0; i < 3; i++) { Rwr[i*3] = T[i*4]; Rwr[i*3+1] = T[i*4+1]; Rwr[i*3+2] = T[i*4+2]; twr[i] = T[i*4+3]; }
        mat3_mul(Rc, Rwr, Rcr); mat3_vec(Rc, twr, tmp);
        tcr[0] = tmp[0] + tc[0]; tcr[1] = tmp[1] + tc[1]; tcr[2] = tmp[2] + tc[2];
    }
    const double *th = theta + 3*j;
    for (int b = 0; b < 4; b++) {
        double ray[3] = { p->text_box_ray[(j*4 + b)*2], p->text_box_ray[(j*4 + b)*2 + 1], 1.0 };
        double invz = -(ray[0]*th[0] + ray[1]*th[1] + ray[2]*th[2]);
        double Rr[3]; mat3_vec(Rcr, ray, Rr);
        double X = Rr[0]/invz + tcr[0], Y = Rr[1]/invz + tcr[1], Z = Rr[2]/invz + tcr[2];
        corners[2*b]   = P->Kl[0]*X/Z + P->Kl[2];
        corners[2*b+1] = P->Kl[1]*Y/Z + P->Kl[3];
    }
    tsba_oracle_musigma(p->img[P->level][kf], p->img_w[P->level], p->img_h[P->level], corners, mu, sigma);
}

/* Text label image of keyframe kf at pyramid level `level` from the parameters in p (optimizer::ShowBAReproj_TextBox,
   src/optimizer.cc:2508-2582 -> tool::TextBoxWithFill / GetTextLabelMask, src/tool.cc:2103-2166): background -1, then every
   text observation of the keyframe, in observation order, fills its projected quad (corners truncated by cv::Point, cv::fillPoly)
   with its rank among the keyframe's observations; later quads overwrite earlier ones. */
int tsba_oracle_label_image(const tsba_problem *p, int kf, int level, float *out) {
    if (!p || !out || kf < 0 || kf >= p->n_kf || level < 0 || level >= p->n_levels) return TSBA_ERR_ARG;
    pass_t P; memset(&P, 0, sizeof(P)); P.p = p; P.level = level; level_K(p, level, P.Kl);
    const int w = p->img_w[level], h = p->img_h[level];
    for (size_t k = 0; k < (size_t)w*h; k++) out[k] = -1.0f;
    uint8_t *mask = (uint8_t *)malloc((size_t)w*h);
    int rank = 0;
    for (int t = 0; t < p->n_tobs; t++) {
        if (p->tobs_kf[t] != kf) continue;
        const int j = p->tobs_text[t], host = p->text_host[j];
        double Rc[9], tc[3], Rcr[9], tcr[3], tmp[3], corners[8]; int xy[8];
        pose_Rt(p->pose + 7*kf, Rc, tc);
        if (host >= 0) { double Rr[9], tr[3]; pose_Rt(p->pose + 7*host, Rr, tr);
            mat3_mulT(Rc, Rr, Rcr); mat3_vec(Rcr, tr, tmp);
            tcr[0] = tc[0] - tmp[0]; tcr[1] = tc[1] - tmp[1]; tcr[2] = tc[2] - tmp[2];
        } else { const double *T = p->text_host_Twr + 12*j; double Rwr[9], twr[3];
            for (int i = 0; i < 3; i++) { Rwr[i*3] = T[i*4]; Rwr[i*3+1] = T[i*4+1]; Rwr[i*3+2] = T[i*4+2]; twr[i] = T[i*4+3]; }
            mat3_mul(Rc, Rwr, Rcr); mat3_vec(Rc, twr, tmp);
            tcr[0] = tmp[0] + tc[0]; tcr[1] = tmp[1] + tc[1]; tcr[2] = tmp[2] + tc[2]; }
        const double *th = p->theta + 3*j;
        for (int b = 0; b < 4; b++) {
            double ray[3] = { p->text_box_ray[(j*4 + b)*2], p->text_box_ray[(j*4 + b)*2 + 1], 1.0 };
            double invz = -(ray[0]*th[0] + ray[1]*th[1] + ray[2]*th[2]);
            double Rr[3]; mat3_vec(Rcr, ray, Rr);
            double X = Rr[0]/invz + tcr[0], Y = Rr[1]/invz + tcr[1], Z = Rr[2]/invz + tcr[2];
            corners[2*b] = P.Kl[0]*X/Z + P.Kl[2]; corners[2*b+1] = P.Kl[1]*Y/Z + P.Kl[3];
            xy[2*b] = (int)corners[2*b]; xy[2*b+1] = (int)corners[2*b+1];            /* cv::Point(double, double): truncation */
        }
        tsba_oracle_fillpoly4(w, h, xy, mask);
        for (size_t k = 0; k < (size_t)w*h; k++) if (mask[k]) out[k] = (float)rank;
        rank++;
    }
    free(mask);
    return TSBA_OK;
}

static int pass_build(pass_t *P, const tsba_problem *p, const tsba_options *o, int level) {
    memset(P, 0, sizeof(*P));
    P->p = p; P->o = o; P->level = level;
    level_K(p, 0, P->K0); level_K(p, level, P->Kl);
    int cap = p->n_sobs[level] + (o->use_text ? 1 : 0);
    if (o->use_text) for (int t = 0; t < p->n_tobs; t++) { int j = p->tobs_text[t]; cap += p->tfeat_off[level][j+1] - p->tfeat_off[level][j]; }
    P->blk = (blk_t *)calloc((size_t)cap + 1, sizeof(blk_t));
    P->mu = (double *)calloc((size_t)p->n_tobs + 1, sizeof(double));
    P->sigma = (double *)calloc((size_t)p->n_tobs + 1, sizeof(double));
    P->tobs_size = (int *)calloc((size_t)p->n_tobs + 1, sizeof(int));
    P->kf_in = (uint8_t *)calloc((size_t)p->n_kf, 1); P->kf_const = (uint8_t *)calloc((size_t)p->n_kf, 1);
    P->free_idx = (int *)malloc(sizeof(int)*(size_t)p->n_kf);
    P->pt_lm = (int *)malloc(sizeof(int)*(size_t)(p->n_pt + 1)); P->tx_lm = (int *)malloc(sizeof(int)*(size_t)(p->n_text + 1));
    for (int i = 0; i < p->n_pt; i++) P->pt_lm[i] = -1;
    for (int i = 0; i < p->n_text; i++) P->tx_lm[i] = -1;
    int n = 0;
    /* A) scene points, optimizer.cc:1366-1435 / :1731-1765 */
    for (int s = 0; s < p->n_sobs[level]; s++) {
        if (o->filter_good && !p->sgood[p->sobs_flag[level][s]]) continue;
        int kf = p->sobs_kf[level][s], pt = p->sobs_pt[level][s], host = p->pt_host[pt];
        blk_t b; memset(&b, 0, sizeof(b));
        b.kf = kf; b.host = host; b.lm = pt; b.nres = 2; b.src = s; b.tobs = -1;
        if (host >= 0) { if (host == kf) continue; b.type = BLK_SCENE_BA; P->kf_in[kf] = 1; P->kf_in[host] = 1; }
        else { b.type = BLK_SCENE_POSE; P->kf_in[kf] = 1; }
        P->blk[n++] = b;
    }
    P->ns = n;
    /* B) text objects, optimizer.cc:1447-1557 */
    if (o->use_text) for (int t = 0; t < p->n_tobs; t++) {
        P->tobs_size[t] = 0;
        if (o->filter_good && !p->tobs_good[t]) continue;
        int kf = p->tobs_kf[t], j = p->tobs_text[t], host = p->text_host[j];
        if (host >= 0 && host == kf) continue;
        double corners[8];
        pair_musigma(P, t, p->pose, p->theta, &P->mu[t], &P->sigma[t], corners);
        for (int f = p->tfeat_off[level][j]; f < p->tfeat_off[level][j+1]; f++) {
            if (o->filter_good && !p->tfgood[p->tobs_fgood_off[t] + p->tfeat_raw[level][f]]) continue;
            blk_t b; memset(&b, 0, sizeof(b));
            b.kf = kf; b.host = host; b.lm = j; b.nres = 8; b.src = f; b.tobs = t;
            b.type = host >= 0 ? BLK_TEXT_BA : BLK_TEXT_POSE;
            P->kf_in[kf] = 1; if (host >= 0) P->kf_in[host] = 1;
            P->blk[n++] = b; P->tobs_size[t]++;
        }
    }
    P->nblk = n; P->nt = n - P->ns;
    /* gauge, optimizer.cc:1562-1588 / :1825-1830 */
    for (int k = 0; k < p->n_kf; k++) if (p->kf_initial && p->kf_initial[k] && P->kf_in[k]) P->kf_const[k] = 1;
    if (o->state == TSBA_STATE_LOCAL) {
        int cnt = 0; for (int k = 0; k < p->n_kf; k++) cnt += P->kf_in[k];
        if (cnt > 3) { int fixed = 0; for (int k = 0; k < p->n_kf && fixed < 3; k++) if (P->kf_in[k]) { P->kf_const[k] = 1; fixed++; } }
    }
    P->nf = 0;
    for (int k = 0; k < p->n_kf; k++) P->free_idx[k] = (P->kf_in[k] && !P->kf_const[k]) ? P->nf++ : -1;
    P->nlm = 0;
    for (int i = 0; i < n; i++) {
        blk_t *b = &P->blk[i];
        if (b->type == BLK_SCENE_BA) { if (P->pt_lm[b->lm] < 0) P->pt_lm[b->lm] = P->nlm++; }
        else if (b->type == BLK_TEXT_BA) { if (P->tx_lm[b->lm] < 0) P->tx_lm[b->lm] = P->nlm++; }
        else b->fixed = P->kf_const[b->kf];
    }
    return 0;
}
static void pass_free(pass_t *P) {
    free(P->blk); free(P->mu); free(P->sigma); free(P->tobs_size); free(P->kf_in); free(P->kf_const);
    free(P->free_idx); free(P->pt_lm); free(P->tx_lm);
}

/* host pose of a frozen scene block as (q,t): auto_PoseOptimScene.h:41-43 */
static void frozen_scene_host(const tsba_problem *p, int pt, double q[4], double t[3]) {
    const double *T = p->pt_host_Trw + 12*pt; double R[9], qq[4];
    for (int i = 0; i < 3; i++) { R[i*3] = T[i*4]; R[i*3+1] = T[i*4+1]; R[i*3+2] = T[i*4+2]; t[i] = T[i*4+3]; }
    R_to_quat(R, qq); quat_normalize(qq, q);
}

/* raw residual of one block at (pose, rho, theta) */
static void blk_residual(const pass_t *P, const blk_t *b, const double *pose, const double *rho, const double *theta, double r[8]) {
    const tsba_problem *p = P->p; const tsba_options *o = P->o; int l = P->level;
    if (b->type == BLK_SCENE_BA || b->type == BLK_SCENE_POSE) {
        double ray[3] = { p->pt_ray[2*b->lm], p->pt_ray[2*b->lm+1], 1.0 };
        const double *uv = p->sobs_uv0[l] + 2*b->src;
        if (b->type == BLK_SCENE_BA)
            f_scene(pose + 7*b->kf, pose + 7*b->kf + 4, pose + 7*b->host, pose + 7*b->host + 4, rho[b->lm], ray, P->K0, o->w_sx, o->w_sy, uv[0], uv[1], r);
        else {
            double q[4], t[3]; frozen_scene_host(p, b->lm, q, t);
            f_scene(pose + 7*b->kf, pose + 7*b->kf + 4, q, t, rho[b->lm], ray, P->K0, o->w_sx, o->w_sy, uv[0], uv[1], r);
        }
    } else {
        const double *fuv = p->tfeat_uv[l] + 2*b->src, *ref = p->tfeat_ref[l] + 8*b->src;
        const uint8_t *img = p->img[l][b->kf]; int w = p->img_w[l], h = p->img_h[l];
        if (b->type == BLK_TEXT_BA)
            f_ba_text(pose + 7*b->kf, pose + 7*b->kf + 4, pose + 7*b->host, pose + 7*b->host + 4, theta + 3*b->lm,
                      fuv[0], fuv[1], P->Kl, img, w, h, ref, P->mu[b->tobs], P->sigma[b->tobs], o->w_t, r);
        else
            f_pose_text(pose + 7*b->kf, pose + 7*b->kf + 4, p->text_host_Twr + 12*b->lm, theta + 3*b->lm,
                        fuv[0], fuv[1], P->Kl, img, w, h, ref, P->mu[b->tobs], P->sigma[b->tobs], o->w_t, r);
    }
}

/* ceres::QuaternionParameterization::ComputeJacobian: 4x3, row-major */
static void quat_plus_jacobian(const double x[4], double J[12]) {
    J[0] = -x[1]; J[1]  = -x[2]; J[2]  = -x[3];
    J[3] =  x[0]; J[4]  =  x[3]; J[5]  = -x[2];
    J[6] = -x[3]; J[7]  =  x[0]; J[8]  =  x[1];
    J[9] =  x[2]; J[10] = -x[1]; J[11] =  x[0];
}
/* ceres::QuaternionParameterization::Plus */
static void quat_plus(const double x[4], const double d[3], double o[4]) {
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

/* Ceres NumericDiffCostFunction<CENTRAL> on the raw parameter blocks of a text block, then the quaternion
 * manifold: numeric_diff.h (relative_step_size 1e-6, min step sqrt(eps)).  Output tangent-space blocks. */
static void text_jac_numeric(const pass_t *P, const blk_t *b, const double *pose, const double *rho, const double *theta,
                             double Jt[48], double Jh[48], double Jl[24]) {
    const tsba_problem *p = P->p;
    size_t npose = 7*(size_t)p->n_kf, nth = 3*(size_t)p->n_text;
    double *pz = (double *)malloc(sizeof(double)*(npose + nth)); double *tz = pz + npose;
    memcpy(pz, pose, sizeof(double)*npose); memcpy(tz, theta, sizeof(double)*nth);
    const double min_step = sqrt(DBL_EPSILON);
    double amb_t[8*7], amb_h[8*7], rp[8], rm[8];
    for (int which = 0; which < 3; which++) {
        double *x; int n; double *out; int ld;
        if (which == 0) { x = pz + 7*b->kf; n = 7; out = amb_t; ld = 7; }
        else if (which == 1) { if (b->type != BLK_TEXT_BA) continue; x = pz + 7*b->host; n = 7; out = amb_h; ld = 7; }
        else { if (b->type != BLK_TEXT_BA) continue; x = tz + 3*b->lm; n = 3; out = Jl; ld = 3; }
        for (int j = 0; j < n; j++) {
            double x0 = x[j], delta = fabs(x0)*1e-6; if (delta < min_step) delta = min_step;
            x[j] = x0 + delta; blk_residual(P, b, pz, rho, tz, rp);
            x[j] = x0 - delta; blk_residual(P, b, pz, rho, tz, rm);
            x[j] = x0;
            double inv = (1.0/delta)/2;
            for (int k = 0; k < 8; k++) out[k*ld + j] = (rp[k] - rm[k])*inv;
        }
    }
    double PJ[12];
    quat_plus_jacobian(pose + 7*b->kf, PJ);
    for (int k = 0; k < 8; k++) {
        for (int c = 0; c < 3; c++) {
            double s = 0; for (int a = 0; a < 4; a++) s += amb_t[k*7 + a]*PJ[a*3 + c];
            Jt[k*6 + c] = s; Jt[k*6 + 3 + c] = amb_t[k*7 + 4 + c];
        }
    }
    if (b->type == BLK_TEXT_BA) {
        quat_plus_jacobian(pose + 7*b->host, PJ);
        for (int k = 0; k < 8; k++) for (int c = 0; c < 3; c++) {
            double s = 0; for (int a = 0; a < 4; a++) s += amb_h[k*7 + a]*PJ[a*3 + c];
            Jh[k*6 + c] = s; Jh[k*6 + 3 + c] = amb_h[k*7 + 4 + c];
        }
    } else { memset(Jh, 0, sizeof(double)*48); memset(Jl, 0, sizeof(double)*24); }
    free(pz);
}

/* dr/dP (1x3) -> tangent blocks, SURVEY.md Appendix A.  G_t = [-2[P-t_c]x | I], G_h = [2 R_cr [y]x | -R_cr] */
static void chain_pose(const double drdP[3], const double Pmtc[3], const double Rcr[9], const double y[3], int has_host,
                       double *Jt_row, double *Jh_row) {
    double S[9]; skew(Pmtc, S);
    for (int c = 0; c < 3; c++) {
        Jt_row[c] = -2.0*(drdP[0]*S[0*3+c] + drdP[1]*S[1*3+c] + drdP[2]*S[2*3+c]);
        Jt_row[3+c] = drdP[c];
    }
    if (has_host) {
        double Sy[9], RS[9]; skew(y, Sy); mat3_mul(Rcr, Sy, RS);
        for (int c = 0; c < 3; c++) {
            Jh_row[c] = 2.0*(drdP[0]*RS[0*3+c] + drdP[1]*RS[1*3+c] + drdP[2]*RS[2*3+c]);
            Jh_row[3+c] = -(drdP[0]*Rcr[0*3+c] + drdP[1]*Rcr[1*3+c] + drdP[2]*Rcr[2*3+c]);
        }
    } else for (int c = 0; c < 6; c++) Jh_row[c] = 0;
}

/* residual + tangent-space Jacobian of one block.  Jt/Jh: nres x 6, Jl: nres x 3 (scene: column 0 only), row-major */
static void blk_eval(const pass_t *P, const blk_t *b, const double *pose, const double *rho, const double *theta,
                     double r[8], double Jt[48], double Jh[48], double Jl[24]) {
    const tsba_problem *p = P->p; const tsba_options *o = P->o; int l = P->level;
    blk_residual(P, b, pose, rho, theta, r);
    if (!Jt) return;
    memset(Jt, 0, sizeof(double)*48); memset(Jh, 0, sizeof(double)*48); memset(Jl, 0, sizeof(double)*24);
    double Rc[9], tc[3]; pose_Rt(pose + 7*b->kf, Rc, tc);
    if (b->type == BLK_SCENE_BA || b->type == BLK_SCENE_POSE) {
        double Rr[9], tr[3];
        if (b->type == BLK_SCENE_BA) pose_Rt(pose + 7*b->host, Rr, tr);
        else { double q[4]; frozen_scene_host(p, b->lm, q, tr); quat_to_R(q, Rr); }
        double Rcr[9]; mat3_mulT(Rc, Rr, Rcr);
        double m[3] = { p->pt_ray[2*b->lm], p->pt_ray[2*b->lm+1], 1.0 }, rh = rho[b->lm];
        double y[3] = { m[0]/rh - tr[0], m[1]/rh - tr[1], m[2]/rh - tr[2] };
        double Pm[3]; mat3_vec(Rcr, y, Pm);                     /* P - t_c */
        double Pc[3] = { Pm[0] + tc[0], Pm[1] + tc[1], Pm[2] + tc[2] };
        double fx = P->K0[0], fy = P->K0[1];
        double A[2][3] = { { o->w_sx*fx/Pc[2], 0, -o->w_sx*fx*Pc[0]/(Pc[2]*Pc[2]) }, { 0, o->w_sy*fy/Pc[2], -o->w_sy*fy*Pc[1]/(Pc[2]*Pc[2]) } };
        double Rm[3]; mat3_vec(Rcr, m, Rm);
        for (int k = 0; k < 2; k++) {
            chain_pose(A[k], Pm, Rcr, y, b->type == BLK_SCENE_BA, Jt + 6*k, Jh + 6*k);
            if (b->type == BLK_SCENE_BA) Jl[3*k] = -(A[k][0]*Rm[0] + A[k][1]*Rm[1] + A[k][2]*Rm[2])/(rh*rh);
        }
    } else {
        if (o->text_jacobian == 1) { text_jac_numeric(P, b, pose, rho, theta, Jt, Jh, Jl); return; }
        double sigma = P->sigma[b->tobs];
        if (sigma == 0) return;
        const double *fuv = p->tfeat_uv[l] + 2*b->src; const double *th = theta + 3*b->lm;
        const uint8_t *img = p->img[l][b->kf]; int w = p->img_w[l], h = p->img_h[l];
        double Rcr[9], tr[3] = {0,0,0}, Rwr[9], twr[3];
        if (b->type == BLK_TEXT_BA) { double Rr[9]; pose_Rt(pose + 7*b->host, Rr, tr); mat3_mulT(Rc, Rr, Rcr); }
        else {
            const double *T = p->text_host_Twr + 12*b->lm;
            for (int i = 0; i < 3; i++) { Rwr[i*3] = T[i*4]; Rwr[i*3+1] = T[i*4+1]; Rwr[i*3+2] = T[i*4+2]; twr[i] = T[i*4+3]; }
            mat3_mul(Rc, Rwr, Rcr);
        }
        for (int k = 0; k < 8; k++) {
            double m[3] = { (fuv[0] + TAP_DX[k] - P->Kl[2])/P->Kl[0], (fuv[1] + TAP_DY[k] - P->Kl[3])/P->Kl[1], 1.0 };
            double s = -(m[0]*th[0] + m[1]*th[1] + m[2]*th[2]);
            double Pm[3], y[3];
            if (b->type == BLK_TEXT_BA) { y[0] = m[0]/s - tr[0]; y[1] = m[1]/s - tr[1]; y[2] = m[2]/s - tr[2]; mat3_vec(Rcr, y, Pm); }
            else { double Xr[3] = { m[0]/s, m[1]/s, m[2]/s }, Xw[3]; mat3_vec(Rwr, Xr, Xw); Xw[0] += twr[0]; Xw[1] += twr[1]; Xw[2] += twr[2]; mat3_vec(Rc, Xw, Pm); y[0] = y[1] = y[2] = 0; }
            double Pc[3] = { Pm[0] + tc[0], Pm[1] + tc[1], Pm[2] + tc[2] };
            double u = P->Kl[0]*Pc[0]/Pc[2] + P->Kl[2], v = P->Kl[1]*Pc[1]/Pc[2] + P->Kl[3];
            double gu, gv; bilinear(img, w, h, u, v, &gu, &gv);
            double g0 = o->w_t/sigma*gu, g1 = o->w_t/sigma*gv;
            double drdP[3] = { g0*P->Kl[0]/Pc[2], g1*P->Kl[1]/Pc[2], -(g0*P->Kl[0]*Pc[0] + g1*P->Kl[1]*Pc[1])/(Pc[2]*Pc[2]) };
            chain_pose(drdP, Pm, Rcr, y, b->type == BLK_TEXT_BA, Jt + 6*k, Jh + 6*k);
            if (b->type == BLK_TEXT_BA) {
                double Rm[3]; mat3_vec(Rcr, m, Rm);
                double c = (drdP[0]*Rm[0] + drdP[1]*Rm[1] + drdP[2]*Rm[2])/(s*s);
                Jl[3*k] = c*m[0]; Jl[3*k+1] = c*m[1]; Jl[3*k+2] = c*m[2];
            }
        }
    }
}

/* ceres::HuberLoss + Corrector (corrector.cc: rho'' <= 0 => scale residual and Jacobian by sqrt(rho')) */
static double huber(double s, double delta, double *scale) {
    double b = delta*delta;
    if (s > b) { double r = sqrt(s); double rho1 = delta/r; if (rho1 < DBL_MIN) rho1 = DBL_MIN; *scale = sqrt(rho1); return 2.0*delta*r - b; }
    *scale = 1.0; return s;
}

/* ------------------------------------------------------------------ public: eval */
int tsba_oracle_eval(const tsba_problem *p, const tsba_options *o, int level,
                     double *resid, double *jac, double *musigma, int64_t *ns, int64_t *nt) {
    if (!p || !o || level < 0 || level >= p->n_levels) return TSBA_ERR_ARG;
    pass_t P; pass_build(&P, p, o, level);
    double *rp = resid, *jp = jac;
    for (int i = 0; i < P.nblk; i++) {
        blk_t *b = &P.blk[i]; double r[8], Jt[48], Jh[48], Jl[24];
        blk_eval(&P, b, p->pose, p->rho, p->theta, r, jac ? Jt : NULL, Jh, Jl);
        if (resid) { memcpy(rp, r, sizeof(double)*b->nres); rp += b->nres; }
        if (jac) {
            int nc = b->nres == 2 ? 13 : 15, nl = b->nres == 2 ? 1 : 3;
            for (int k = 0; k < b->nres; k++) {
                for (int c = 0; c < 6; c++) { jp[k*nc + c] = Jt[k*6 + c]; jp[k*nc + 6 + c] = Jh[k*6 + c]; }
                for (int c = 0; c < nl; c++) jp[k*nc + 12 + c] = Jl[k*3 + c];
            }
            jp += b->nres*nc;
        }
    }
    if (musigma) for (int t = 0; t < p->n_tobs; t++) { musigma[2*t] = P.mu[t]; musigma[2*t+1] = P.sigma[t]; }
    if (ns) *ns = P.ns;
    if (nt) *nt = P.nt;
    pass_free(&P);
    return TSBA_OK;
}

/* ------------------------------------------------------------------ normal equations */
typedef struct { int col; double W[18]; } lm_entry;     /* 6 x d block of J_pose^T J_lm, row-major 6 x 3 */
/* 6x6 blocks of a symmetric block-sparse matrix (lower triangle incl. the diagonal blocks), found through an open-addressing hash on
 * (block row, block column): the storage of H_pp / S for maps whose co-visibility graph is neither small nor a band (loop closures,
 * long-range observations at thousands of keyframes).  Blocks keep the order in which they were first touched. */
typedef struct { int nb, cap; long long *key; int *slot; int nblk, cblk; int *br, *bc; double *val; } bsp_t;
static void bsp_init(bsp_t *B, int nb) { memset(B, 0, sizeof(*B)); B->nb = nb; B->cap = 1 << 12; B->key = (long long *)malloc(sizeof(long long)*(size_t)B->cap); B->slot = (int *)malloc(sizeof(int)*(size_t)B->cap);
    for (int i = 0; i < B->cap; i++) B->key[i] = -1;
    B->cblk = 1 << 10; B->br = (int *)malloc(sizeof(int)*(size_t)B->cblk); B->bc = (int *)malloc(sizeof(int)*(size_t)B->cblk); B->val = (double *)malloc(sizeof(double)*36*(size_t)B->cblk); }
static void bsp_free(bsp_t *B) { free(B->key); free(B->slot); free(B->br); free(B->bc); free(B->val); memset(B, 0, sizeof(*B)); }
static void bsp_rehash(bsp_t *B) {
    int ncap = B->cap*2; long long *nk = (long long *)malloc(sizeof(long long)*(size_t)ncap); int *ns = (int *)malloc(sizeof(int)*(size_t)ncap);
    for (int i = 0; i < ncap; i++) nk[i] = -1;
    for (int i = 0; i < B->cap; i++) if (B->key[i] >= 0) { unsigned long long h = (unsigned long long)B->key[i]*0x9E3779B97F4A7C15ull; int q = (int)(h >> 40) & (ncap - 1);
        while (nk[q] >= 0) q = (q + 1) & (ncap - 1);
        nk[q] = B->key[i]; ns[q] = B->slot[i]; }
    free(B->key); free(B->slot); B->key = nk; B->slot = ns; B->cap = ncap;
}
static double *bsp_block(bsp_t *B, int r, int c, int create) {      /* block (r, c), r >= c; NULL if absent and !create */
    const long long k = (long long)r*B->nb + c; unsigned long long h = (unsigned long long)k*0x9E3779B97F4A7C15ull; int q = (int)(h >> 40) & (B->cap - 1);
    while (B->key[q] >= 0) { if (B->key[q] == k) return B->val + 36*(size_t)B->slot[q]; q = (q + 1) & (B->cap - 1); }
    if (!create) return NULL;
    if (B->nblk == B->cblk) { B->cblk *= 2; B->br = (int *)realloc(B->br, sizeof(int)*(size_t)B->cblk); B->bc = (int *)realloc(B->bc, sizeof(int)*(size_t)B->cblk); B->val = (double *)realloc(B->val, sizeof(double)*36*(size_t)B->cblk); }
    const int id = B->nblk++; B->br[id] = r; B->bc[id] = c; memset(B->val + 36*(size_t)id, 0, sizeof(double)*36);
    B->key[q] = k; B->slot[q] = id;
    if (2*(size_t)B->nblk > (size_t)B->cap) bsp_rehash(B);
    return B->val + 36*(size_t)id;
}
static void bsp_copy(bsp_t *D, const bsp_t *S) {                     /* same blocks, same order */
    bsp_init(D, S->nb);
    for (int i = 0; i < S->nblk; i++) memcpy(bsp_block(D, S->br[i], S->bc[i], 1), S->val + 36*(size_t)i, sizeof(double)*36);
}

typedef struct {
    int nf, nlm;
    int bw;                            /* -1: Hpp dense (6nf)^2; >= 0: lower band, Hpp[a*(bw+1) + (a-c)] = H(a, c) for a-bw <= c <= a
                                          (maps of thousands of keyframes: the dense matrix would be 7 GB at 5000 keyframes);
                                          -2: block-sparse (hs): any co-visibility graph, the linear solve is handed to g_sparse_solver */
    bsp_t hs;
    double *Hpp, *bp;                  /* (6nf)^2 or 6nf x (bw+1), 6nf */
    double *V, *bl; int *dim;          /* per landmark: 3x3, 3, d */
    lm_entry *ent; int *ent_off, *ent_cnt;
    double cost;
} neq_t;

static lm_entry *lm_find(neq_t *N, int li, int col) {
    lm_entry *e = N->ent + N->ent_off[li];
    for (int i = 0; i < N->ent_cnt[li]; i++) if (e[i].col == col) return &e[i];
    e += N->ent_cnt[li]++; e->col = col; memset(e->W, 0, sizeof(e->W)); return e;
}

/* keyframe count from which run_pass keeps H_pp / S as a band (tsba_oracle_set_band_threshold: tests force either mode) */
static int g_band_min_nf = 400;
void tsba_oracle_set_band_threshold(int nf) { g_band_min_nf = nf; }

static inline void hpp_add(neq_t *N, int n6, int r, int c, double v) {
    if (N->bw == -2) { if (r/6 >= c/6) bsp_block(&N->hs, r/6, c/6, 1)[6*(r % 6) + (c % 6)] += v; }     /* block-sparse: the lower block triangle (diagonal blocks in full) */
    else if (N->bw < 0) N->Hpp[(size_t)r*n6 + c] += v;
    else if (r >= c) N->Hpp[(size_t)r*(N->bw + 1) + (r - c)] += v;        /* band: the lower triangle only */
}
static inline double hpp_diag(const neq_t *N, int n6, int a) {
    if (N->bw == -2) { const double *b = bsp_block((bsp_t *)&N->hs, a/6, a/6, 0); return b ? b[7*(a % 6)] : 0.0; }
    return N->bw < 0 ? N->Hpp[(size_t)a*n6 + a] : N->Hpp[(size_t)a*(N->bw + 1)]; }

static void neq_alloc(neq_t *N, const pass_t *P, int band) {
    memset(N, 0, sizeof(*N));
    N->nf = P->nf; N->nlm = P->nlm; int n6 = 6*N->nf;
    N->bw = -1;
    if (band == 2) { N->bw = -2; bsp_init(&N->hs, N->nf); }
    else if (band) {      /* half bandwidth: a landmark couples every pair of its free poses (Cholesky without pivoting keeps the band) */
        int *lo = (int *)malloc(sizeof(int)*((size_t)N->nlm + 1)), *hi = (int *)malloc(sizeof(int)*((size_t)N->nlm + 1)), bwb = 0;
        for (int i = 0; i < N->nlm; i++) { lo[i] = N->nf; hi[i] = -1; }
        for (int i = 0; i < P->nblk; i++) { const blk_t *b = &P->blk[i];
            int ct = P->free_idx[b->kf], ch = b->host >= 0 ? P->free_idx[b->host] : -1;
            int li = b->type == BLK_SCENE_BA ? P->pt_lm[b->lm] : (b->type == BLK_TEXT_BA ? P->tx_lm[b->lm] : -1);
            if (ct >= 0 && ch >= 0 && abs(ct - ch) > bwb) bwb = abs(ct - ch);
            if (li >= 0) { if (ct >= 0) { if (ct < lo[li]) lo[li] = ct; if (ct > hi[li]) hi[li] = ct; }
                           if (ch >= 0) { if (ch < lo[li]) lo[li] = ch; if (ch > hi[li]) hi[li] = ch; } } }
        for (int i = 0; i < N->nlm; i++) if (hi[i] >= 0 && hi[i] - lo[i] > bwb) bwb = hi[i] - lo[i];
        free(lo); free(hi);
        N->bw = 6*bwb + 5;
    }
    N->Hpp = (double *)calloc((N->bw == -2 ? 0 : N->bw < 0 ? (size_t)n6*n6 : (size_t)n6*(N->bw + 1)) + 1, sizeof(double)); N->bp = (double *)calloc((size_t)n6 + 1, sizeof(double));
    N->V = (double *)calloc((size_t)9*N->nlm + 1, sizeof(double)); N->bl = (double *)calloc((size_t)3*N->nlm + 1, sizeof(double));
    N->dim = (int *)calloc((size_t)N->nlm + 1, sizeof(int));
    N->ent_off = (int *)calloc((size_t)N->nlm + 1, sizeof(int)); N->ent_cnt = (int *)calloc((size_t)N->nlm + 1, sizeof(int));
    int *cnt = (int *)calloc((size_t)N->nlm + 1, sizeof(int));
    for (int i = 0; i < P->nblk; i++) { const blk_t *b = &P->blk[i];
        if (b->type == BLK_SCENE_BA) cnt[P->pt_lm[b->lm]] += 2; else if (b->type == BLK_TEXT_BA) cnt[P->tx_lm[b->lm]] += 2; }
    int tot = 0; for (int i = 0; i < N->nlm; i++) { N->ent_off[i] = tot; tot += cnt[i]; }
    N->ent = (lm_entry *)calloc((size_t)tot + 1, sizeof(lm_entry));
    free(cnt);
}
static void neq_free(neq_t *N) { if (N->bw == -2) bsp_free(&N->hs); free(N->Hpp); free(N->bp); free(N->V); free(N->bl); free(N->dim); free(N->ent); free(N->ent_off); free(N->ent_cnt); }

/* shard filter for the multi-GPU restatement: a block belongs to the rank that owns its landmark */
static int blk_in_shard(const pass_t *P, const blk_t *b) {
    const tsba_options *o = P->o;
    if (o->lm_nshard <= 1) return 1;
    /* the rank whose keyframe range holds the landmark's host; the observations of a frozen landmark go with their target */
    int k = b->host >= 0 ? b->host : b->kf;
    return (int)(((long long)k*o->lm_nshard)/P->p->n_kf) == o->lm_shard;
}

/* linearise at (pose,rho,theta): loss-corrected J^T J, J^T r.  Optionally keep corrected (r,J) per block for the model-cost test. */
static void linearize(const pass_t *P, const double *pose, const double *rho, const double *theta, neq_t *N, double *keep) {
    int n6 = 6*N->nf;
    if (N->bw == -2) memset(N->hs.val, 0, sizeof(double)*36*(size_t)N->hs.nblk);        /* (the blocks of an earlier linearisation stay, as zeros) */
    else memset(N->Hpp, 0, sizeof(double)*(N->bw < 0 ? (size_t)n6*n6 : (size_t)n6*(N->bw + 1)));
    memset(N->bp, 0, sizeof(double)*(size_t)n6);
    memset(N->V, 0, sizeof(double)*9*(size_t)N->nlm); memset(N->bl, 0, sizeof(double)*3*(size_t)N->nlm);
    memset(N->ent_cnt, 0, sizeof(int)*(size_t)N->nlm);
    N->cost = 0;
#ifdef _OPENMP
    /* all-core CPU baseline build only (bench.py: gcc -fopenmp): the residual / Jacobian evaluation of the blocks -- the expensive
     * part, numeric differentiation of the text blocks above all -- in parallel into `keep`; the accumulation below stays serial
     * and in block order, so the sums are those of the single-thread build */
    double *bcost = NULL;
    if (keep) {
        bcost = (double *)malloc(sizeof(double)*((size_t)P->nblk + 1));
        #pragma omp parallel for schedule(dynamic, 64)
        for (int i = 0; i < P->nblk; i++) {
            const blk_t *b = &P->blk[i];
            double *kp = keep + (size_t)i*128; bcost[i] = 0;
            if (b->fixed || !blk_in_shard(P, b)) { memset(kp, 0, sizeof(double)*128); continue; }
            double r[8], Jt[48], Jh[48], Jl[24];
            blk_eval(P, b, pose, rho, theta, r, Jt, Jh, Jl);
            double s = 0; for (int k = 0; k < b->nres; k++) s += r[k]*r[k];
            double scale, delta = b->nres == 2 ? P->o->huber_scene : P->o->huber_text;
            bcost[i] = 0.5*huber(s, delta, &scale);
            for (int k = 0; k < b->nres; k++) { r[k] *= scale; for (int c = 0; c < 6; c++) { Jt[k*6+c] *= scale; Jh[k*6+c] *= scale; } for (int c = 0; c < 3; c++) Jl[k*3+c] *= scale; }
            memcpy(kp, r, 64); memcpy(kp + 8, Jt, 384); memcpy(kp + 56, Jh, 384); memcpy(kp + 104, Jl, 192);
        }
    }
#endif
    for (int i = 0; i < P->nblk; i++) {
        const blk_t *b = &P->blk[i];
        double *kp = keep ? keep + (size_t)i*(8 + 48 + 48 + 24) : NULL;
        double r[8], Jt[48], Jh[48], Jl[24];
#ifdef _OPENMP
        if (bcost) {
            if (b->fixed || !blk_in_shard(P, b)) continue;
            N->cost += bcost[i];
            memcpy(r, kp, 64); memcpy(Jt, kp + 8, 384); memcpy(Jh, kp + 56, 384); memcpy(Jl, kp + 104, 192);
        } else
#endif
        {
        if (b->fixed || !blk_in_shard(P, b)) { if (kp) memset(kp, 0, sizeof(double)*128); continue; }
        blk_eval(P, b, pose, rho, theta, r, Jt, Jh, Jl);
        double s = 0; for (int k = 0; k < b->nres; k++) s += r[k]*r[k];
        double scale, delta = b->nres == 2 ? P->o->huber_scene : P->o->huber_text;
        N->cost += 0.5*huber(s, delta, &scale);
        for (int k = 0; k < b->nres; k++) { r[k] *= scale; for (int c = 0; c < 6; c++) { Jt[k*6+c] *= scale; Jh[k*6+c] *= scale; } for (int c = 0; c < 3; c++) Jl[k*3+c] *= scale; }
        if (kp) { memcpy(kp, r, 64); memcpy(kp + 8, Jt, 384); memcpy(kp + 56, Jh, 384); memcpy(kp + 104, Jl, 192); }
        }
        int ct = P->free_idx[b->kf], ch = b->host >= 0 ? P->free_idx[b->host] : -1;
        int has_lm = (b->type == BLK_SCENE_BA || b->type == BLK_TEXT_BA);
        int d = b->nres == 2 ? 1 : 3;
        int li = -1; if (has_lm) { li = b->nres == 2 ? P->pt_lm[b->lm] : P->tx_lm[b->lm]; N->dim[li] = d; }
        for (int k = 0; k < b->nres; k++) {
            const double *jt = Jt + 6*k, *jh = Jh + 6*k, *jl = Jl + 3*k;
            if (ct >= 0) {
                for (int a = 0; a < 6; a++) { for (int c = 0; c < 6; c++) hpp_add(N, n6, 6*ct + a, 6*ct + c, jt[a]*jt[c]); N->bp[6*ct + a] += jt[a]*r[k]; }
                if (ch >= 0) for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) {
                    hpp_add(N, n6, 6*ct + a, 6*ch + c, jt[a]*jh[c]); hpp_add(N, n6, 6*ch + c, 6*ct + a, jt[a]*jh[c]); }
            }
            if (ch >= 0) for (int a = 0; a < 6; a++) { for (int c = 0; c < 6; c++) hpp_add(N, n6, 6*ch + a, 6*ch + c, jh[a]*jh[c]); N->bp[6*ch + a] += jh[a]*r[k]; }
            if (has_lm) {
                for (int a = 0; a < d; a++) { for (int c = 0; c < d; c++) N->V[9*li + 3*a + c] += jl[a]*jl[c]; N->bl[3*li + a] += jl[a]*r[k]; }
                if (ct >= 0) { lm_entry *e = lm_find(N, li, ct); for (int a = 0; a < 6; a++) for (int c = 0; c < d; c++) e->W[3*a + c] += jt[a]*jl[c]; }
                if (ch >= 0) { lm_entry *e = lm_find(N, li, ch); for (int a = 0; a < 6; a++) for (int c = 0; c < d; c++) e->W[3*a + c] += jh[a]*jl[c]; }
            }
        }
    }
#ifdef _OPENMP
    free(bcost);
#endif
}

/* cost only (EvaluateCost at a candidate) */
static double eval_cost(const pass_t *P, const double *pose, const double *rho, const double *theta) {
    double cost = 0;
#ifdef _OPENMP
    #pragma omp parallel for schedule(static) reduction(+:cost)     /* (baseline build only: the summation order differs from the serial build) */
#endif
    for (int i = 0; i < P->nblk; i++) {
        const blk_t *b = &P->blk[i]; if (b->fixed || !blk_in_shard(P, b)) continue;
        double r[8]; blk_residual(P, b, pose, rho, theta, r);
        double s = 0; for (int k = 0; k < b->nres; k++) s += r[k]*r[k];
        double scale; cost += 0.5*huber(s, b->nres == 2 ? P->o->huber_scene : P->o->huber_text, &scale);
    }
    return cost;
}

static int chol_inplace(double *A, int n) {          /* lower Cholesky, row-major, in place */
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j*n + j];
        for (int k = 0; k < j; k++) d -= A[(size_t)j*n + k]*A[(size_t)j*n + k];
        if (!(d > 0)) return -1;
        d = sqrt(d); A[(size_t)j*n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = A[(size_t)i*n + j];
            const double *ai = A + (size_t)i*n, *aj = A + (size_t)j*n;
            for (int k = 0; k < j; k++) s -= ai[k]*aj[k];
            A[(size_t)i*n + j] = s/d;
        }
    }
    return 0;
}
static void chol_solve(const double *L, int n, double *b) {
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[(size_t)i*n + k]*b[k]; b[i] = s/L[(size_t)i*n + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = b[i]; for (int k = i + 1; k < n; k++) s -= L[(size_t)k*n + i]*b[k]; b[i] = s/L[(size_t)i*n + i]; }
}
static int inv_sym(const double *V, int d, double *Vi) {    /* inverse of d x d (d=1 or 3) stored 3x3 */
    if (d == 1) { if (!(V[0] > 0)) return -1; Vi[0] = 1.0/V[0]; return 0; }
    double a = V[0], b = V[1], c = V[2], e = V[4], f = V[5], i = V[8];
    double A = e*i - f*f, B = -(b*i - c*f), C = b*f - c*e;
    double det = a*A + b*B + c*C;
    if (!(det > 0) || !(a > 0) || !(a*e - b*b > 0)) return -1;
    double id = 1.0/det;
    Vi[0] = A*id; Vi[1] = B*id; Vi[2] = C*id;
    Vi[3] = B*id; Vi[4] = (a*i - c*c)*id; Vi[5] = -(a*f - b*c)*id;
    Vi[6] = C*id; Vi[7] = Vi[5]; Vi[8] = (a*e - b*b)*id;
    return 0;
}

/* Schur complement in Jacobi-scaled coordinates with LM damping D2 = diag/radius:
 *   (Hs + D2) y = -gs.  sp/sl = column scales, dgp/dgl = clamped scaled diagonals.
 *   Outputs y (scaled step) for poses (6nf) and landmarks (3 per lm).  Optionally exports S, g. */
static int schur_solve(const neq_t *N, const double *sp, const double *sl, const double *dgp, const double *dgl, double radius,
                       double *yp, double *yl, double *S_out, double *g_out) {
    int n6 = 6*N->nf;
    double *S = (double *)malloc(sizeof(double)*((size_t)n6*n6 + 1)), *g = (double *)malloc(sizeof(double)*((size_t)n6 + 1));
    for (int a = 0; a < n6; a++) { for (int c = 0; c < n6; c++) S[(size_t)a*n6 + c] = sp[a]*sp[c]*N->Hpp[(size_t)a*n6 + c]; S[(size_t)a*n6 + a] += dgp[a]/radius; g[a] = sp[a]*N->bp[a]; }
    double *Vinv = (double *)malloc(sizeof(double)*(9*(size_t)N->nlm + 1));
    int rc = 0;
    for (int li = 0; li < N->nlm && !rc; li++) {
        int d = N->dim[li]; if (d == 0) continue;
        double Vs[9] = {0}, bs[3];
        for (int a = 0; a < d; a++) { for (int c = 0; c < d; c++) Vs[3*a + c] = sl[3*li + a]*sl[3*li + c]*N->V[9*li + 3*a + c]; Vs[3*a + a] += dgl[3*li + a]/radius; bs[a] = sl[3*li + a]*N->bl[3*li + a]; }
        if (inv_sym(Vs, d, Vinv + 9*li)) { rc = -1; break; }
        const double *Vi = Vinv + 9*li;
        const lm_entry *e = N->ent + N->ent_off[li]; int ne = N->ent_cnt[li];
        for (int i = 0; i < ne; i++) {
            double WV[18];   /* (scaled W_i) * Vinv : 6 x d */
            for (int a = 0; a < 6; a++) for (int c = 0; c < d; c++) { double s = 0; for (int k = 0; k < d; k++) s += sp[6*e[i].col + a]*e[i].W[3*a + k]*sl[3*li + k]*Vi[3*k + c]; WV[3*a + c] = s; }
            for (int a = 0; a < 6; a++) { double s = 0; for (int c = 0; c < d; c++) s += WV[3*a + c]*bs[c]; g[6*e[i].col + a] -= s; }
            for (int j = 0; j < ne; j++)
                for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) { double s = 0; for (int k = 0; k < d; k++) s += WV[3*a + k]*sp[6*e[j].col + c]*e[j].W[3*c + k]*sl[3*li + k]; S[(size_t)(6*e[i].col + a)*n6 + 6*e[j].col + c] -= s; }
        }
    }
    if (S_out) memcpy(S_out, S, sizeof(double)*(size_t)n6*n6);
    if (g_out) memcpy(g_out, g, sizeof(double)*(size_t)n6);
    if (!rc && yp) {
        if (n6 > 0 && chol_inplace(S, n6)) rc = -1;
        if (!rc) {
            for (int a = 0; a < n6; a++) yp[a] = -g[a];
            if (n6 > 0) chol_solve(S, n6, yp);
            for (int li = 0; li < N->nlm; li++) {
                int d = N->dim[li]; if (d == 0) { yl[3*li] = yl[3*li+1] = yl[3*li+2] = 0; continue; }
                double rhs[3];
                for (int a = 0; a < d; a++) rhs[a] = -sl[3*li + a]*N->bl[3*li + a];
                const lm_entry *e = N->ent + N->ent_off[li];
                for (int i = 0; i < N->ent_cnt[li]; i++) for (int k = 0; k < d; k++) { double s = 0; for (int a = 0; a < 6; a++) s += sp[6*e[i].col + a]*e[i].W[3*a + k]*yp[6*e[i].col + a]; rhs[k] -= sl[3*li + k]*s; }
                for (int a = 0; a < d; a++) { double s = 0; for (int c = 0; c < d; c++) s += Vinv[9*li + 3*a + c]*rhs[c]; yl[3*li + a] = s; }
                for (int a = d; a < 3; a++) yl[3*li + a] = 0;
            }
        }
    }
    free(S); free(g); free(Vinv);
    return rc;
}

/* The same for a band H_pp (N->bw >= 0): S as a lower band, band Cholesky (no fill outside the band), the sums over exactly the
 * structurally non-zero terms of the dense variant in the same order -- the two variants agree to the last bit up to signed zeros. */
static int schur_solve_band(const neq_t *N, const double *sp, const double *sl, const double *dgp, const double *dgl, double radius,
                            double *yp, double *yl) {
    const int n6 = 6*N->nf, bw = N->bw, ld = bw + 1;
    double *S = (double *)calloc((size_t)n6*ld + 1, sizeof(double)), *g = (double *)malloc(sizeof(double)*((size_t)n6 + 1));
    #define SB(a, c) S[(size_t)(a)*ld + ((a) - (c))]
    for (int a = 0; a < n6; a++) { for (int c = a - bw < 0 ? 0 : a - bw; c <= a; c++) SB(a, c) = sp[a]*sp[c]*N->Hpp[(size_t)a*ld + (a - c)]; SB(a, a) += dgp[a]/radius; g[a] = sp[a]*N->bp[a]; }
    double *Vinv = (double *)malloc(sizeof(double)*(9*(size_t)N->nlm + 1));
    int rc = 0;
    for (int li = 0; li < N->nlm && !rc; li++) {
        int d = N->dim[li]; if (d == 0) continue;
        double Vs[9] = {0}, bs[3];
        for (int a = 0; a < d; a++) { for (int c = 0; c < d; c++) Vs[3*a + c] = sl[3*li + a]*sl[3*li + c]*N->V[9*li + 3*a + c]; Vs[3*a + a] += dgl[3*li + a]/radius; bs[a] = sl[3*li + a]*N->bl[3*li + a]; }
        if (inv_sym(Vs, d, Vinv + 9*li)) { rc = -1; break; }
        const double *Vi = Vinv + 9*li;
        const lm_entry *e = N->ent + N->ent_off[li]; int ne = N->ent_cnt[li];
        for (int i = 0; i < ne; i++) {
            double WV[18];
            for (int a = 0; a < 6; a++) for (int c = 0; c < d; c++) { double s = 0; for (int k = 0; k < d; k++) s += sp[6*e[i].col + a]*e[i].W[3*a + k]*sl[3*li + k]*Vi[3*k + c]; WV[3*a + c] = s; }
            for (int a = 0; a < 6; a++) { double s = 0; for (int c = 0; c < d; c++) s += WV[3*a + c]*bs[c]; g[6*e[i].col + a] -= s; }
            for (int j = 0; j < ne; j++)
                for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) { const int rr = 6*e[i].col + a, cc = 6*e[j].col + c; if (rr < cc) continue;
                    double s = 0; for (int k = 0; k < d; k++) s += WV[3*a + k]*sp[cc]*e[j].W[3*c + k]*sl[3*li + k]; SB(rr, cc) -= s; }
        }
    }
    if (!rc) {
        for (int j = 0; j < n6 && !rc; j++) {                      /* band Cholesky, lower, in place */
            double dj = SB(j, j);
            for (int k = j - bw < 0 ? 0 : j - bw; k < j; k++) dj -= SB(j, k)*SB(j, k);
            if (!(dj > 0)) { rc = -1; break; }
            dj = sqrt(dj); SB(j, j) = dj;
            const int iend = j + bw < n6 - 1 ? j + bw : n6 - 1;
            for (int i = j + 1; i <= iend; i++) { double s = SB(i, j);
                for (int k = i - bw < 0 ? 0 : i - bw; k < j; k++) s -= SB(i, k)*SB(j, k);
                SB(i, j) = s/dj; }
        }
    }
    if (!rc) {
        for (int a = 0; a < n6; a++) yp[a] = -g[a];
        for (int i = 0; i < n6; i++) { double s = yp[i]; for (int k = i - bw < 0 ? 0 : i - bw; k < i; k++) s -= SB(i, k)*yp[k]; yp[i] = s/SB(i, i); }
        for (int i = n6 - 1; i >= 0; i--) { double s = yp[i]; const int kend = i + bw < n6 - 1 ? i + bw : n6 - 1; for (int k = i + 1; k <= kend; k++) s -= SB(k, i)*yp[k]; yp[i] = s/SB(i, i); }
        for (int li = 0; li < N->nlm; li++) {
            int d = N->dim[li]; if (d == 0) { yl[3*li] = yl[3*li+1] = yl[3*li+2] = 0; continue; }
            double rhs[3];
            for (int a = 0; a < d; a++) rhs[a] = -sl[3*li + a]*N->bl[3*li + a];
            const lm_entry *e = N->ent + N->ent_off[li];
            for (int i = 0; i < N->ent_cnt[li]; i++) for (int k = 0; k < d; k++) { double s = 0; for (int a = 0; a < 6; a++) s += sp[6*e[i].col + a]*e[i].W[3*a + k]*yp[6*e[i].col + a]; rhs[k] -= sl[3*li + k]*s; }
            for (int a = 0; a < d; a++) { double s = 0; for (int c = 0; c < d; c++) s += Vinv[9*li + 3*a + c]*rhs[c]; yl[3*li + a] = s; }
            for (int a = d; a < 3; a++) yl[3*li + a] = 0;
        }
    }
    #undef SB
    free(S); free(g); free(Vinv);
    return rc;
}


/* The same for a block-sparse H_pp (N->bw == -2): S as hashed 6x6 blocks of the lower block triangle, the sums over exactly the terms of the
 * dense variant in the same order (the blocks agree with the dense S to the last bit), and the linear solve S y = -g handed to
 * g_sparse_solver -- the reference's SPARSE_NORMAL_CHOLESKY is an exact sparse factorisation (optimizer.cc:1833-1840); the tests plug
 * scipy's sparse direct solver in here (a converged iterative solve where the factor would fill).  With yp == NULL the blocks are
 * only exported: S_out takes the hashed store (caller frees with bsp_free), g_out the reduced gradient. */
typedef int (*tsba_oracle_sparse_solver)(int n, int nblk, const int *br, const int *bc, const double *val, const double *rhs, double *y);
static tsba_oracle_sparse_solver g_sparse_solver = NULL;
void tsba_oracle_set_sparse_solver(tsba_oracle_sparse_solver f) { g_sparse_solver = f; }

static int schur_solve_sparse(const neq_t *N, const double *sp, const double *sl, const double *dgp, const double *dgl, double radius,
                              double *yp, double *yl, bsp_t *S_out, double *g_out) {
    const int n6 = 6*N->nf;
    bsp_t S; bsp_copy(&S, &N->hs);
    double *g = (double *)malloc(sizeof(double)*((size_t)n6 + 1));
    for (int i = 0; i < S.nblk; i++) { double *v = S.val + 36*(size_t)i; const int r0 = 6*S.br[i], c0 = 6*S.bc[i];
        for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) v[6*a + c] *= sp[r0 + a]*sp[c0 + c]; }
    for (int a = 0; a < n6; a++) { bsp_block(&S, a/6, a/6, 1)[7*(a % 6)] += dgp[a]/radius; g[a] = sp[a]*N->bp[a]; }
    double *Vinv = (double *)malloc(sizeof(double)*(9*(size_t)N->nlm + 1));
    int rc = 0;
    for (int li = 0; li < N->nlm && !rc; li++) {
        int d = N->dim[li]; if (d == 0) continue;
        double Vs[9] = {0}, bs[3];
        for (int a = 0; a < d; a++) { for (int c = 0; c < d; c++) Vs[3*a + c] = sl[3*li + a]*sl[3*li + c]*N->V[9*li + 3*a + c]; Vs[3*a + a] += dgl[3*li + a]/radius; bs[a] = sl[3*li + a]*N->bl[3*li + a]; }
        if (inv_sym(Vs, d, Vinv + 9*li)) { rc = -1; break; }
        const double *Vi = Vinv + 9*li;
        const lm_entry *e = N->ent + N->ent_off[li]; int ne = N->ent_cnt[li];
        for (int i = 0; i < ne; i++) {
            double WV[18];
            for (int a = 0; a < 6; a++) for (int c = 0; c < d; c++) { double s = 0; for (int k = 0; k < d; k++) s += sp[6*e[i].col + a]*e[i].W[3*a + k]*sl[3*li + k]*Vi[3*k + c]; WV[3*a + c] = s; }
            for (int a = 0; a < 6; a++) { double s = 0; for (int c = 0; c < d; c++) s += WV[3*a + c]*bs[c]; g[6*e[i].col + a] -= s; }
            for (int j = 0; j < ne; j++) { if (e[i].col < e[j].col) continue;
                double *blk = bsp_block(&S, e[i].col, e[j].col, 1);
                for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) {
                    double s = 0; for (int k = 0; k < d; k++) s += WV[3*a + k]*sp[6*e[j].col + c]*e[j].W[3*c + k]*sl[3*li + k]; blk[6*a + c] -= s; } }
        }
    }
    if (g_out) memcpy(g_out, g, sizeof(double)*(size_t)n6);
    if (!rc && yp) {
        if (!g_sparse_solver) rc = -1;
        else {
            double *rhs = (double *)malloc(sizeof(double)*((size_t)n6 + 1));
            for (int a = 0; a < n6; a++) rhs[a] = -g[a];
            if (n6 > 0 && g_sparse_solver(n6, S.nblk, S.br, S.bc, S.val, rhs, yp)) rc = -1;
            free(rhs);
        }
        if (!rc) for (int li = 0; li < N->nlm; li++) {
            int d = N->dim[li]; if (d == 0) { yl[3*li] = yl[3*li+1] = yl[3*li+2] = 0; continue; }
            double rhs[3];
            for (int a = 0; a < d; a++) rhs[a] = -sl[3*li + a]*N->bl[3*li + a];
            const lm_entry *e = N->ent + N->ent_off[li];
            for (int i = 0; i < N->ent_cnt[li]; i++) for (int k = 0; k < d; k++) { double s = 0; for (int a = 0; a < 6; a++) s += sp[6*e[i].col + a]*e[i].W[3*a + k]*yp[6*e[i].col + a]; rhs[k] -= sl[3*li + k]*s; }
            for (int a = 0; a < d; a++) { double s = 0; for (int c = 0; c < d; c++) s += Vinv[9*li + 3*a + c]*rhs[c]; yl[3*li + a] = s; }
            for (int a = d; a < 3; a++) yl[3*li + a] = 0;
        }
    }
    if (S_out) *S_out = S; else bsp_free(&S);
    free(g); free(Vinv);
    return rc;
}

/* Per-trial record of the LM loop for the parity tests of long runs (where do two correct implementations part?):
 * trace[4*k] = candidate cost (NaN: invalid step), [4*k+1] = model cost change, [4*k+2] = radius after the decision,
 * [4*k+3] = 1 accepted / 0 rejected / -1 invalid step / 2 terminated by a tolerance on this trial.  cap trials per pass, pass-major. */
static double *g_trace = NULL; static int g_trace_cap = 0;
void tsba_oracle_set_trace(double *buf, int cap_per_pass) { g_trace = buf; g_trace_cap = cap_per_pass; }
#define TRACE(pass, it, c, m, r, f) do { if (g_trace && (it) >= 1 && (it) <= g_trace_cap) { double *t_ = g_trace + 4*((size_t)(pass)*g_trace_cap + (it) - 1); t_[0] = (c); t_[1] = (m); t_[2] = (r); t_[3] = (f); } } while (0)

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* x (+) delta over the reduced program */
static void apply_step(const pass_t *P, const double *pose, const double *rho, const double *theta,
                       const double *dp, const double *dl, double *pose2, double *rho2, double *theta2) {
    const tsba_problem *p = P->p;
    memcpy(pose2, pose, sizeof(double)*7*(size_t)p->n_kf); memcpy(rho2, rho, sizeof(double)*(size_t)p->n_pt); memcpy(theta2, theta, sizeof(double)*3*(size_t)p->n_text);
    for (int k = 0; k < p->n_kf; k++) { int c = P->free_idx[k]; if (c < 0) continue;
        quat_plus(pose + 7*k, dp + 6*c, pose2 + 7*k);
        for (int a = 0; a < 3; a++) pose2[7*k + 4 + a] = pose[7*k + 4 + a] + dp[6*c + 3 + a]; }
    for (int j = 0; j < p->n_pt; j++) { int li = P->pt_lm[j]; if (li >= 0) rho2[j] = rho[j] + dl[3*li]; }
    for (int j = 0; j < p->n_text; j++) { int li = P->tx_lm[j]; if (li >= 0) for (int a = 0; a < 3; a++) theta2[3*j + a] = theta[3*j + a] + dl[3*li + a]; }
}
static double reduced_norm(const pass_t *P, const double *pose, const double *rho, const double *theta,
                           const double *pose_b, const double *rho_b, const double *theta_b) {
    const tsba_problem *p = P->p; double s = 0;
    for (int k = 0; k < p->n_kf; k++) if (P->free_idx[k] >= 0) for (int a = 0; a < 7; a++) { double d = pose[7*k + a] - (pose_b ? pose_b[7*k + a] : 0); s += d*d; }
    for (int j = 0; j < p->n_pt; j++) if (P->pt_lm[j] >= 0) { double d = rho[j] - (rho_b ? rho_b[j] : 0); s += d*d; }
    for (int j = 0; j < p->n_text; j++) if (P->tx_lm[j] >= 0) for (int a = 0; a < 3; a++) { double d = theta[3*j + a] - (theta_b ? theta_b[3*j + a] : 0); s += d*d; }
    return sqrt(s);
}

static void jacobi_and_diag(const neq_t *N, double *sp, double *sl, int compute_scale, double *dgp, double *dgl, const tsba_options *o) {
    int n6 = 6*N->nf;
    for (int a = 0; a < n6; a++) {
        double h = hpp_diag(N, n6, a);
        if (compute_scale) sp[a] = 1.0/(1.0 + sqrt(h));
        dgp[a] = clampd(sp[a]*sp[a]*h, o->min_diagonal, o->max_diagonal);
    }
    for (int li = 0; li < N->nlm; li++) for (int a = 0; a < 3; a++) {
        double h = a < N->dim[li] ? N->V[9*li + 3*a + a] : 0;
        if (compute_scale) sl[3*li + a] = 1.0/(1.0 + sqrt(h));
        dgl[3*li + a] = clampd(sl[3*li + a]*sl[3*li + a]*h, o->min_diagonal, o->max_diagonal);
    }
}

int tsba_oracle_reduced_system(const tsba_problem *p, const tsba_options *o, int level, double radius,
                               int32_t *free_idx, double *S, double *g, double *Hpp, double *bp, double *cost) {
    if (!p || !o || level < 0 || level >= p->n_levels) return TSBA_ERR_ARG;
    pass_t P; pass_build(&P, p, o, level);
    neq_t N; neq_alloc(&N, &P, 0);
    linearize(&P, p->pose, p->rho, p->theta, &N, NULL);
    int n6 = 6*N.nf;
    double *sp = (double *)malloc(sizeof(double)*(n6 + 1)), *dgp = (double *)malloc(sizeof(double)*(n6 + 1));
    double *sl = (double *)malloc(sizeof(double)*(3*(size_t)N.nlm + 1)), *dgl = (double *)malloc(sizeof(double)*(3*(size_t)N.nlm + 1));
    jacobi_and_diag(&N, sp, sl, 1, dgp, dgl, o);
    /* export in UNSCALED coordinates: S_unscaled = Sigma^-1 S_scaled Sigma^-1, g likewise */
    double *Ss = (double *)malloc(sizeof(double)*((size_t)n6*n6 + 1)), *gs = (double *)malloc(sizeof(double)*(n6 + 1));
    int rc = schur_solve(&N, sp, sl, dgp, dgl, radius, NULL, NULL, Ss, gs);
    for (int a = 0; a < n6; a++) { for (int c = 0; c < n6; c++) if (S) S[(size_t)a*n6 + c] = Ss[(size_t)a*n6 + c]/(sp[a]*sp[c]); if (g) g[a] = gs[a]/sp[a]; }
    if (Hpp) memcpy(Hpp, N.Hpp, sizeof(double)*(size_t)n6*n6);
    if (bp) memcpy(bp, N.bp, sizeof(double)*(size_t)n6);
    if (cost) *cost = N.cost;
    if (free_idx) for (int k = 0; k < p->n_kf; k++) free_idx[k] = P.free_idx[k];
    int nf = N.nf;
    free(sp); free(dgp); free(sl); free(dgl); free(Ss); free(gs);
    neq_free(&N); pass_free(&P);
    return rc ? TSBA_ERR_NUMERIC : nf;
}


/* The reduced camera system of the first linearisation as 6x6 blocks, for maps where the dense (6 nf)^2 copy is not an option: block q
 * couples free poses br[q] >= bc[q] (column-block indices as in free_idx), val[36 q ..] row-major with rows = pose br[q]; UNSCALED
 * coordinates, pose damping for `radius` included, as tsba_oracle_reduced_system.  Call with br == NULL for the number of blocks
 * (return value); then with buffers of that size.  g [6 nf], cost, free_idx [n_kf] as there. */
int tsba_oracle_reduced_blocks(const tsba_problem *p, const tsba_options *o, int level, double radius,
                               int32_t *free_idx, int32_t *br, int32_t *bc, double *val, double *g, double *cost) {
    if (!p || !o || level < 0 || level >= p->n_levels) return TSBA_ERR_ARG;
    pass_t P; pass_build(&P, p, o, level);
    neq_t N; neq_alloc(&N, &P, 2);
    linearize(&P, p->pose, p->rho, p->theta, &N, NULL);
    int n6 = 6*N.nf;
    double *sp = (double *)malloc(sizeof(double)*(n6 + 1)), *dgp = (double *)malloc(sizeof(double)*(n6 + 1));
    double *sl = (double *)malloc(sizeof(double)*(3*(size_t)N.nlm + 1)), *dgl = (double *)malloc(sizeof(double)*(3*(size_t)N.nlm + 1));
    jacobi_and_diag(&N, sp, sl, 1, dgp, dgl, o);
    bsp_t S; double *gs = (double *)malloc(sizeof(double)*(n6 + 1));
    int rc = schur_solve_sparse(&N, sp, sl, dgp, dgl, radius, NULL, NULL, &S, gs);
    int nblk = S.nblk;
    if (br && bc && val) for (int i = 0; i < S.nblk; i++) { br[i] = S.br[i]; bc[i] = S.bc[i];
        for (int a = 0; a < 6; a++) for (int c = 0; c < 6; c++) val[36*(size_t)i + 6*a + c] = S.val[36*(size_t)i + 6*a + c]/(sp[6*S.br[i] + a]*sp[6*S.bc[i] + c]); }
    if (g) for (int a = 0; a < n6; a++) g[a] = gs[a]/sp[a];
    if (cost) *cost = N.cost;
    if (free_idx) for (int k = 0; k < p->n_kf; k++) free_idx[k] = P.free_idx[k];
    bsp_free(&S);
    free(sp); free(dgp); free(sl); free(dgl); free(gs);
    neq_free(&N); pass_free(&P);
    return rc ? TSBA_ERR_NUMERIC : nblk;
}

/* Multi-GPU restatement: what ONE rank (options.lm_shard of lm_nshard) contributes to the reduced normal equations before the
 * all-reduce: S_part = H_pp,part - sum_{own landmarks} W (V + Lambda_l)^-1 W^T (no pose damping, unscaled pose coordinates),
 * g_part, diag(H_pp,part) and the partial cost.  Summing the parts over the ranks and adding the pose damping
 * Lambda_p = clamp(s^2 Hd)/(radius s^2), s = 1/(1+sqrt(Hd)), reproduces tsba_oracle_reduced_system of the unsharded problem. */
int tsba_oracle_partial_system(const tsba_problem *p, const tsba_options *o, int level, double radius,
                               int32_t *free_idx, double *S, double *g, double *Hd, double *cost) {
    if (!p || !o || level < 0 || level >= p->n_levels) return TSBA_ERR_ARG;
    pass_t P; pass_build(&P, p, o, level);
    neq_t N; neq_alloc(&N, &P, 0);
    linearize(&P, p->pose, p->rho, p->theta, &N, NULL);
    int n6 = 6*N.nf;
    double *sp = (double *)malloc(sizeof(double)*(n6 + 1)), *dgp = (double *)calloc(n6 + 1, sizeof(double));
    double *sl = (double *)malloc(sizeof(double)*(3*(size_t)N.nlm + 1)), *dgl = (double *)malloc(sizeof(double)*(3*(size_t)N.nlm + 1));
    jacobi_and_diag(&N, sp, sl, 1, dgp, dgl, o);
    for (int a = 0; a < n6; a++) { sp[a] = 1.0; dgp[a] = 0.0; }          /* poses: unscaled, undamped */
    int rc = schur_solve(&N, sp, sl, dgp, dgl, radius, NULL, NULL, S, g);
    if (Hd) for (int a = 0; a < n6; a++) Hd[a] = N.Hpp[(size_t)a*n6 + a];
    if (cost) *cost = N.cost;
    if (free_idx) for (int k = 0; k < p->n_kf; k++) free_idx[k] = P.free_idx[k];
    int nf = N.nf;
    free(sp); free(dgp); free(sl); free(dgl);
    neq_free(&N); pass_free(&P);
    return rc ? TSBA_ERR_NUMERIC : nf;
}

/* ceres::Covariance of theta[text] with every other parameter constant (optimizer.cc:2219-2238): inverse of the loss-corrected
 * J^T J of its residual blocks at the current parameters.  Returns 0, or TSBA_ERR_NUMERIC if singular. */
int tsba_oracle_theta_cov(const tsba_problem *p, const tsba_options *o, int level, int text, double cov[9]) {
    if (!p || !o || level < 0 || level >= p->n_levels || text < 0 || text >= p->n_text) return TSBA_ERR_ARG;
    pass_t P; pass_build(&P, p, o, level);
    double V[9] = {0};
    for (int i = 0; i < P.nblk; i++) {
        const blk_t *b = &P.blk[i];
        if (b->type != BLK_TEXT_BA || b->lm != text) continue;
        double r[8], Jt[48], Jh[48], Jl[24];
        blk_eval(&P, b, p->pose, p->rho, p->theta, r, Jt, Jh, Jl);
        double s = 0; for (int k = 0; k < 8; k++) s += r[k]*r[k];
        double scale; huber(s, o->huber_text, &scale);
        for (int k = 0; k < 8; k++) for (int a = 0; a < 3; a++) for (int c2 = 0; c2 < 3; c2++) V[3*a + c2] += scale*scale*Jl[3*k + a]*Jl[3*k + c2];
    }
    pass_free(&P);
    return inv_sym(V, 3, cov) ? TSBA_ERR_NUMERIC : TSBA_OK;
}

/* ------------------------------------------------------------------ one pyramid pass: LM + outlier pass */
static int run_pass(tsba_problem *p, const tsba_options *o, int pass, tsba_report *rep, int cov_text, double *cov_out, int *cov_rc) {
    int level = o->levels[pass], max_it = o->its[pass];
    pass_t P; pass_build(&P, p, o, level);
    neq_t N; neq_alloc(&N, &P, g_sparse_solver ? 2 : P.nf >= g_band_min_nf);
    int n6 = 6*N.nf; size_t nl3 = 3*(size_t)N.nlm;
    size_t npose = 7*(size_t)p->n_kf, nrho = (size_t)p->n_pt, nth = 3*(size_t)p->n_text;
    double *x_pose = (double *)malloc(sizeof(double)*(npose + 1)), *x_rho = (double *)malloc(sizeof(double)*(nrho + 1)), *x_th = (double *)malloc(sizeof(double)*(nth + 1));
    double *c_pose = (double *)malloc(sizeof(double)*(npose + 1)), *c_rho = (double *)malloc(sizeof(double)*(nrho + 1)), *c_th = (double *)malloc(sizeof(double)*(nth + 1));
    memcpy(x_pose, p->pose, sizeof(double)*npose);
    if (nrho) memcpy(x_rho, p->rho, sizeof(double)*nrho);                 /* (a problem without points / planes passes NULL: memcpy(…, NULL, 0) is undefined) */
    if (nth) memcpy(x_th, p->theta, sizeof(double)*nth);
    double *sp = (double *)calloc(n6 + 1, sizeof(double)), *dgp = (double *)calloc(n6 + 1, sizeof(double)), *yp = (double *)calloc(n6 + 1, sizeof(double)), *dp = (double *)calloc(n6 + 1, sizeof(double));
    double *sl = (double *)calloc(nl3 + 1, sizeof(double)), *dgl = (double *)calloc(nl3 + 1, sizeof(double)), *yl = (double *)calloc(nl3 + 1, sizeof(double)), *dl = (double *)calloc(nl3 + 1, sizeof(double));
    double *keep = (double *)malloc(sizeof(double)*128*((size_t)P.nblk + 1));
    int64_t nres_blk = 0; for (int i = 0; i < P.nblk; i++) if (!P.blk[i].fixed) nres_blk += P.blk[i].nres;

    /* iteration 0 */
    linearize(&P, x_pose, x_rho, x_th, &N, keep);
    rep->n_resid_evals += nres_blk;
    double x_cost = N.cost; rep->cost0[pass] = x_cost;
    jacobi_and_diag(&N, sp, sl, 1, dgp, dgl, o);
    double x_norm = reduced_norm(&P, x_pose, x_rho, x_th, NULL, NULL, NULL);
    double radius = o->initial_radius, decrease_factor = 2.0; int reuse_diag = 0, invalid = 0;
    int term = 0, it = 0, accepted = 0;
    #define GRAD_MAX(gm) do { gm = 0; for (int a = 0; a < n6; a++) if (fabs(N.bp[a]) > gm) gm = fabs(N.bp[a]); \
        for (int li = 0; li < N.nlm; li++) for (int a = 0; a < N.dim[li]; a++) if (fabs(N.bl[3*li + a]) > gm) gm = fabs(N.bl[3*li + a]); } while (0)
    double gmax; GRAD_MAX(gmax);
    if (n6 == 0 && N.nlm == 0) { term = 5; goto done; }
    if (gmax <= o->gradient_tolerance) { term = 3; goto done; }
    while (1) {
        if (it >= max_it) { term = 0; break; }
        if (radius < o->min_radius) { term = 4; break; }
        it++;
        if (!reuse_diag) jacobi_and_diag(&N, sp, sl, 0, dgp, dgl, o);
        int rc = N.bw == -2 ? schur_solve_sparse(&N, sp, sl, dgp, dgl, radius, yp, yl, NULL, NULL) : N.bw < 0 ? schur_solve(&N, sp, sl, dgp, dgl, radius, yp, yl, NULL, NULL) : schur_solve_band(&N, sp, sl, dgp, dgl, radius, yp, yl);
        double model_change = -1;
        if (!rc) {
            for (int a = 0; a < n6; a++) dp[a] = sp[a]*yp[a];
            for (size_t a = 0; a < nl3; a++) dl[a] = sl[a]*yl[a];
            /* model_cost_change = -(J d)^T (r + J d / 2), trust_region_minimizer.cc */
            model_change = 0;
            for (int i = 0; i < P.nblk; i++) { const blk_t *b = &P.blk[i]; if (b->fixed || !blk_in_shard(&P, b)) continue;
                const double *kp = keep + (size_t)i*128; int ct = P.free_idx[b->kf], ch = b->host >= 0 ? P.free_idx[b->host] : -1;
                int li = b->type == BLK_SCENE_BA ? P.pt_lm[b->lm] : (b->type == BLK_TEXT_BA ? P.tx_lm[b->lm] : -1);
                for (int k = 0; k < b->nres; k++) { double jd = 0;
                    if (ct >= 0) for (int a = 0; a < 6; a++) jd += kp[8 + 6*k + a]*dp[6*ct + a];
                    if (ch >= 0) for (int a = 0; a < 6; a++) jd += kp[56 + 6*k + a]*dp[6*ch + a];
                    if (li >= 0) for (int a = 0; a < 3; a++) jd += kp[104 + 3*k + a]*dl[3*li + a];
                    model_change -= jd*(kp[k] + jd/2); } }
        }
        if (rc || !(model_change > 0)) {           /* invalid step: StepIsInvalid() */
            if (++invalid >= 5) { term = 5; TRACE(pass, it, NAN, model_change, radius, -1.0); break; }
            radius *= 0.5; reuse_diag = 1; TRACE(pass, it, NAN, model_change, radius, -1.0); continue;
        }
        invalid = 0;
        apply_step(&P, x_pose, x_rho, x_th, dp, dl, c_pose, c_rho, c_th);
        double c_cost = eval_cost(&P, c_pose, c_rho, c_th);
        rep->n_resid_evals += nres_blk;
        if (!(c_cost == c_cost)) c_cost = DBL_MAX;
        double step_norm = reduced_norm(&P, x_pose, x_rho, x_th, c_pose, c_rho, c_th);
        if (step_norm <= o->parameter_tolerance*(x_norm + o->parameter_tolerance)) { term = 2; TRACE(pass, it, c_cost, model_change, radius, 2.0); break; }
        double cost_change = x_cost - c_cost;
        if (fabs(cost_change) <= o->function_tolerance*x_cost) { term = 1; TRACE(pass, it, c_cost, model_change, radius, 2.0); break; }
        double rel = cost_change/model_change;
        if (rel > o->min_relative_decrease) {
            memcpy(x_pose, c_pose, sizeof(double)*npose); memcpy(x_rho, c_rho, sizeof(double)*nrho); memcpy(x_th, c_th, sizeof(double)*nth);
            x_norm = reduced_norm(&P, x_pose, x_rho, x_th, NULL, NULL, NULL);
            linearize(&P, x_pose, x_rho, x_th, &N, keep);
            rep->n_resid_evals += nres_blk;
            x_cost = N.cost; accepted++;
            double t = 2.0*rel - 1.0, f = 1.0 - t*t*t; if (f < 1.0/3.0) f = 1.0/3.0;
            radius = radius/f; if (radius > o->max_radius) radius = o->max_radius;
            decrease_factor = 2.0; reuse_diag = 0;
            TRACE(pass, it, c_cost, model_change, radius, 1.0);
            GRAD_MAX(gmax);
            if (gmax <= o->gradient_tolerance) { term = 3; break; }
        } else {
            radius = radius/decrease_factor; decrease_factor *= 2.0; reuse_diag = 1;
            TRACE(pass, it, c_cost, model_change, radius, 0.0);
        }
    }
done:
    rep->iters[pass] = it; rep->accepted[pass] = accepted; rep->termination[pass] = term; rep->cost1[pass] = x_cost;
    rep->n_sblock[pass] = P.ns; rep->n_tblock[pass] = P.nt;
    memcpy(p->pose, x_pose, sizeof(double)*npose);
    if (nrho) memcpy(p->rho, x_rho, sizeof(double)*nrho);
    if (nth) memcpy(p->theta, x_th, sizeof(double)*nth);

    if (cov_out && cov_text >= 0 && cov_text < p->n_text) {      /* ceres::Covariance on the same problem (mu / sigma of this pass) */
        int li = P.tx_lm[cov_text]; double ctmp[9];                /* the reference keeps the LAST pass whose Covariance::Compute succeeds, optimizer.cc:2224-2241 */
        if (li >= 0 && !inv_sym(N.V + 9*li, 3, ctmp)) { memcpy(cov_out, ctmp, sizeof ctmp); *cov_rc = TSBA_OK; }
    }
    /* outlier pass on loss-corrected residuals, optimizer.cc:1609-1686 / :1228-1305 */
    if (o->outlier_scene || o->outlier_text) {
        double chi2m = o->chi2_mono[pass]; if (P.nt < 50) chi2m += 4;
        double chi2t = o->chi2_text[pass];
        int *bad_in_tobs = (int *)calloc((size_t)p->n_tobs + 1, sizeof(int));
        for (int i = 0; i < P.nblk; i++) {
            const blk_t *b = &P.blk[i]; double r[8];
            blk_residual(&P, b, x_pose, x_rho, x_th, r);
            double s = 0; for (int k = 0; k < b->nres; k++) s += r[k]*r[k];
            double scale; huber(s, b->nres == 2 ? o->huber_scene : o->huber_text, &scale);
            if (b->nres == 2) {
                if (!o->outlier_scene) continue;
                double cx = (r[0]*scale/o->w_sx)*(r[0]*scale/o->w_sx), cy = (r[1]*scale/o->w_sy)*(r[1]*scale/o->w_sy);
                if (cx > chi2m || cy > chi2m) { p->sgood[p->sobs_flag[level][b->src]] = 0; rep->n_bad_scene[pass]++; }
            } else {
                if (!o->outlier_text) continue;
                int bad = 0; for (int k = 0; k < 8; k++) if (fabs(r[k]*scale/o->w_t) > chi2t) bad = 1;
                if (bad) { p->tfgood[p->tobs_fgood_off[b->tobs] + p->tfeat_raw[level][b->src]] = 0; bad_in_tobs[b->tobs]++; rep->n_bad_tfeat[pass]++; }
            }
        }
        if (o->outlier_text) for (int t = 0; t < p->n_tobs; t++) if (P.tobs_size[t] > 0) {
            double ratio = (double)bad_in_tobs[t]/(double)P.tobs_size[t];
            if (ratio > o->text_bad_ratio) { p->tobs_good[t] = 0; rep->n_bad_text[pass]++; }
        }
        free(bad_in_tobs);
    }
    free(x_pose); free(x_rho); free(x_th); free(c_pose); free(c_rho); free(c_th);
    free(sp); free(dgp); free(yp); free(dp); free(sl); free(dgl); free(yl); free(dl); free(keep);
    neq_free(&N); pass_free(&P);
    return term == 5 ? TSBA_ERR_NUMERIC : TSBA_OK;
}

int tsba_oracle_solve(tsba_problem *p, const tsba_options *o, tsba_report *r) {
    if (!p || !o || !r) return TSBA_ERR_ARG;
    memset(r, 0, sizeof(*r));
    r->n_passes = o->n_passes;
    for (int pass = 0; pass < o->n_passes; pass++) {
        if (o->levels[pass] < 0 || o->levels[pass] >= p->n_levels) return TSBA_ERR_ARG;
        run_pass(p, o, pass, r, -1, NULL, NULL);
    }
    return TSBA_OK;
}

/* optimizer::ThetaOptimMultiFs: the solve plus the covariance of theta[text] from the last pass whose information matrix is invertible (optimizer.cc:2219-2241) */
int tsba_oracle_theta_optim(tsba_problem *p, const tsba_options *o, tsba_report *r, int text, double cov[9]) {
    if (!p || !o || !r || !cov) return TSBA_ERR_ARG;
    memset(r, 0, sizeof(*r));
    r->n_passes = o->n_passes;
    int crc = TSBA_ERR_NUMERIC;
    for (int pass = 0; pass < o->n_passes; pass++) {
        if (o->levels[pass] < 0 || o->levels[pass] >= p->n_levels) return TSBA_ERR_ARG;
        run_pass(p, o, pass, r, text, cov, &crc);
    }
    return crc;
}

/* ------------------------------------------------------------------ defaults (duplicated on purpose: the oracle links nothing from the product) */
static void lm_defaults(tsba_options *o) {
    o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32; o->min_relative_decrease = 1e-3;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->min_diagonal = 1e-6; o->max_diagonal = 1e32; o->lm_shard = 0; o->lm_nshard = 1;
}
void tsba_oracle_default_options(tsba_options *o, int kind) {       /* 0 local, 1 pose, 2 global */
    memset(o, 0, sizeof(*o)); lm_defaults(o);
    o->huber_scene = sqrt(5.991); o->huber_text = 3.0; o->text_bad_ratio = 0.99;
    if (kind == 2) {
        o->w_sx = o->w_sy = 1.0; o->w_t = 1.0; o->n_passes = 1; o->levels[0] = 0; o->its[0] = 20; o->chi2_mono[0] = 18;
        o->state = TSBA_STATE_GLOBAL; o->use_text = 0; o->filter_good = 0;
    } else {
        o->w_sx = o->w_sy = 1.0/1.2; o->w_t = 1.0/0.2; o->n_passes = 3;
        for (int i = 0; i < 3; i++) { o->levels[i] = 2 - i; o->its[i] = 10; o->chi2_mono[i] = 12.25; o->chi2_text[i] = i == 2 ? 0.95 : 0.5; }
        o->state = kind == 0 ? TSBA_STATE_LOCAL : TSBA_STATE_NOTREACHWIN; o->use_text = 1; o->filter_good = 1;
        o->outlier_scene = 1; o->outlier_text = 1;
    }
}
