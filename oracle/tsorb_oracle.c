/*
 * TEST INFRASTRUCTURE -- NOT PRODUCT CODE.
 *
 * CPU oracle for TextSLAM's ORBextractor (src/ORBextractor.cc): plain C, exact integer / fp32 arithmetic.
 * PARITY UNPINNED: the reference delegates to OpenCV (cv::resize, copyMakeBorder, cv::FAST, GaussianBlur, fastAtan2, cvRound),
 * which does not exist in the build container; the OpenCV behaviours below are restated from the OpenCV 3.x sources as
 * recalled (README pins 3.3.1), each marked "OpenCV:".  The quadtree tie-break of the reference compares list-node ADDRESSES
 * (ORBextractor.cc:685); here "address order" is defined as creation order, which is what a bump allocator gives.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <float.h>
#include "../include/orb_pattern.h"

#define PATCH_SIZE 31
#define HALF_PATCH 15
#define EDGE 19                     /* EDGE_THRESHOLD, ORBextractor.cc:74 */
#define MAXL 8

static int cv_round_f(float v) { return (int)lrintf(v); }          /* OpenCV: cvRound = round half to even */
static int cv_round_d(double v) { return (int)lrint(v); }

typedef struct {
    int nfeatures, nlevels, ini_th, min_th;
    float scale;
    float sf[MAXL], isf[MAXL];
    int nfl[MAXL], umax[HALF_PATCH + 2];
    int w[MAXL], h[MAXL];
    uint8_t *pyr[MAXL];             /* bordered: (h+38) x (w+38) */
    uint8_t *blur[MAXL];            /* h x w */
} orb_t;

/* ORBextractor::ORBextractor, ORBextractor.cc:410-471 */
static void orb_init(orb_t *o, int nfeatures, float scale, int nlevels, int ini_th, int min_th) {
    memset(o, 0, sizeof(*o));
    o->nfeatures = nfeatures; o->scale = scale; o->nlevels = nlevels; o->ini_th = ini_th; o->min_th = min_th;
    o->sf[0] = 1.0f;
    for (int i = 1; i < nlevels; i++) o->sf[i] = o->sf[i-1]*scale;
    for (int i = 0; i < nlevels; i++) o->isf[i] = 1.0f/o->sf[i];
    float factor = 1.0f/scale;
    float nd = nfeatures*(1 - factor)/(1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) { o->nfl[l] = cv_round_f(nd); sum += o->nfl[l]; nd *= factor; }
    o->nfl[nlevels-1] = nfeatures - sum > 0 ? nfeatures - sum : 0;
    int v, v0, vmax = (int)floor(HALF_PATCH*sqrtf(2.f)/2 + 1), vmin = (int)ceil(HALF_PATCH*sqrtf(2.f)/2);
    const double hp2 = HALF_PATCH*HALF_PATCH;
    for (v = 0; v <= vmax; ++v) o->umax[v] = cv_round_d(sqrt(hp2 - v*v));
    for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) { while (o->umax[v0] == o->umax[v0+1]) ++v0; o->umax[v] = v0; ++v0; }
}

static int reflect101(int x, int n) { if (x < 0) x = -x; if (x >= n) x = 2*(n - 1) - x; return x; }

/* OpenCV: cv::resize(8UC1, INTER_LINEAR): 11-bit fixed-point coefficients, VResizeLinear<uchar,int,short> rounding */
static void resize_linear_u8(const uint8_t *src, int sw, int sh, int sstride, uint8_t *dst, int dw, int dh, int dstride) {
    double inv_x = (double)dw/sw, inv_y = (double)dh/sh;
    double scale_x = 1./inv_x, scale_y = 1./inv_y;
    int *xofs = (int *)malloc(sizeof(int)*dw), *yofs = (int *)malloc(sizeof(int)*dh);
    short *ia = (short *)malloc(sizeof(short)*2*dw), *ib = (short *)malloc(sizeof(short)*2*dh);
    for (int dx = 0; dx < dw; dx++) {
        float fx = (float)((dx + 0.5)*scale_x - 0.5);
        int sx = (int)floorf(fx); fx -= sx;
        if (sx < 0) { fx = 0; sx = 0; }
        if (sx >= sw - 1) { fx = 0; sx = sw - 1; }
        xofs[dx] = sx;
        ia[2*dx] = (short)cv_round_f((1.f - fx)*2048); ia[2*dx+1] = (short)cv_round_f(fx*2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        float fy = (float)((dy + 0.5)*scale_y - 0.5);
        int sy = (int)floorf(fy); fy -= sy;
        yofs[dy] = sy;
        ib[2*dy] = (short)cv_round_f((1.f - fy)*2048); ib[2*dy+1] = (short)cv_round_f(fy*2048);
    }
    for (int dy = 0; dy < dh; dy++) {
        int sy0 = yofs[dy], sy1 = sy0 + 1;
        if (sy0 < 0) sy0 = 0; if (sy0 > sh - 1) sy0 = sh - 1;
        if (sy1 < 0) sy1 = 0; if (sy1 > sh - 1) sy1 = sh - 1;
        const uint8_t *r0 = src + (size_t)sy0*sstride, *r1 = src + (size_t)sy1*sstride;
        int b0 = ib[2*dy], b1 = ib[2*dy+1];
        for (int dx = 0; dx < dw; dx++) {
            int sx = xofs[dx], sx1 = sx + 1 < sw ? sx + 1 : sx, a0 = ia[2*dx], a1 = ia[2*dx+1];
            int S0 = r0[sx]*a0 + r0[sx1]*a1, S1 = r1[sx]*a0 + r1[sx1]*a1;
            dst[(size_t)dy*dstride + dx] = (uint8_t)((((b0*(S0 >> 4)) >> 16) + ((b1*(S1 >> 4)) >> 16) + 2) >> 2);
        }
    }
    free(xofs); free(yofs); free(ia); free(ib);
}

/* ORBextractor::ComputePyramid, ORBextractor.cc:1118-1143 */
static void compute_pyramid(orb_t *o, const uint8_t *img, int w, int h, int stride) {
    for (int l = 0; l < o->nlevels; l++) {
        float sc = o->isf[l];
        int lw = cv_round_f((float)w*sc), lh = cv_round_f((float)h*sc);
        o->w[l] = lw; o->h[l] = lh;
        int bw = lw + 2*EDGE, bh = lh + 2*EDGE;
        o->pyr[l] = (uint8_t *)malloc((size_t)bw*bh);
        uint8_t *in = o->pyr[l] + (size_t)EDGE*bw + EDGE;
        if (l == 0) for (int y = 0; y < lh; y++) memcpy(in + (size_t)y*bw, img + (size_t)y*stride, lw);
        else resize_linear_u8(o->pyr[l-1] + (size_t)EDGE*(o->w[l-1] + 2*EDGE) + EDGE, o->w[l-1], o->h[l-1], o->w[l-1] + 2*EDGE, in, lw, lh, bw);
        for (int y = 0; y < bh; y++) { int sy = reflect101(y - EDGE, lh);
            for (int x = 0; x < bw; x++) { int sx = reflect101(x - EDGE, lw); o->pyr[l][(size_t)y*bw + x] = in[(size_t)sy*bw + sx]; } }
    }
}

/* OpenCV: cv::FAST(img, kps, threshold, nonmaxSuppression = true), FAST-9/16 */
static const int CIRC_X[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
static const int CIRC_Y[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };
static int fast_corner_score(const uint8_t *p, int stride, int threshold, int *is_corner) {
    int v = p[0]; short d[25];
    for (int k = 0; k < 25; k++) d[k] = (short)(v - p[CIRC_Y[k & 15]*stride + CIRC_X[k & 15]]);
    /* corner test: 9 contiguous pixels all darker (d > t) or all brighter (d < -t) */
    int corner = 0;
    for (int s = 0; s < 16 && !corner; s++) {
        int allp = 1, alln = 1;
        for (int k = 0; k < 9; k++) { int dd = d[(s + k) & 15]; if (!(dd > threshold)) allp = 0; if (!(dd < -threshold)) alln = 0; }
        corner = allp | alln;
    }
    *is_corner = corner;
    if (!corner) return 0;
    /* OpenCV: cornerScore<16> */
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k+1] < d[k+2] ? d[k+1] : d[k+2]; if (d[k+3] < a) a = d[k+3];
        if (a <= a0) continue;
        for (int q = 4; q <= 8; q++) if (d[k+q] < a) a = d[k+q];
        int m = a < d[k] ? a : d[k]; if (m > a0) a0 = m;
        m = a < d[k+9] ? a : d[k+9]; if (m > a0) a0 = m;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k+1] > d[k+2] ? d[k+1] : d[k+2]; if (d[k+3] > b) b = d[k+3]; if (d[k+4] > b) b = d[k+4]; if (d[k+5] > b) b = d[k+5];
        if (b >= b0) continue;
        if (d[k+6] > b) b = d[k+6]; if (d[k+7] > b) b = d[k+7]; if (d[k+8] > b) b = d[k+8];
        int m = b > d[k] ? b : d[k]; if (m < b0) b0 = m;
        m = b > d[k+9] ? b : d[k+9]; if (m < b0) b0 = m;
    }
    return -b0 - 1;
}
typedef struct { float x, y, response; } kp_t;
/* FAST on the ROI (rx, ry, rw, rh) of a bordered level image; keypoints relative to the ROI, row-major; returns count */
static int fast_roi(const uint8_t *img, int stride, int rw, int rh, int threshold, kp_t *out, int cap) {
    if (rw < 7 || rh < 7) return 0;
    int *score = (int *)calloc((size_t)rw*rh, sizeof(int));
    for (int y = 3; y < rh - 3; y++) for (int x = 3; x < rw - 3; x++) {
        int c; int s = fast_corner_score(img + (size_t)y*stride + x, stride, threshold, &c);
        score[y*rw + x] = c ? s : 0;
    }
    int n = 0;
    for (int y = 3; y < rh - 3; y++) for (int x = 3; x < rw - 3; x++) {
        int s = score[y*rw + x]; if (!s) continue;
        if (s > score[(y-1)*rw + x-1] && s > score[(y-1)*rw + x] && s > score[(y-1)*rw + x+1] && s > score[y*rw + x-1] &&
            s > score[y*rw + x+1] && s > score[(y+1)*rw + x-1] && s > score[(y+1)*rw + x] && s > score[(y+1)*rw + x+1]) {
            if (n < cap) { out[n].x = (float)x; out[n].y = (float)y; out[n].response = (float)s; }
            n++;
        }
    }
    free(score);
    return n < cap ? n : cap;
}

/* ---------------------------------------------------------------- quadtree, ORBextractor.cc:482-764 */
typedef struct node { int ulx, uly, urx, ury, blx, bly, brx, bry; int *keys; int nk; int nomore; int prev, next; int id; } node_t;
typedef struct { node_t *n; int cnt, cap, head, tail, size; } nlist_t;
static int nl_new(nlist_t *L) { if (L->cnt == L->cap) { L->cap = L->cap ? 2*L->cap : 256; L->n = (node_t *)realloc(L->n, sizeof(node_t)*L->cap); }
    memset(&L->n[L->cnt], 0, sizeof(node_t)); L->n[L->cnt].id = L->cnt; L->n[L->cnt].prev = L->n[L->cnt].next = -1; return L->cnt++; }
static void nl_push_back(nlist_t *L, int i) { L->n[i].prev = L->tail; L->n[i].next = -1; if (L->tail >= 0) L->n[L->tail].next = i; else L->head = i; L->tail = i; L->size++; }
static void nl_push_front(nlist_t *L, int i) { L->n[i].next = L->head; L->n[i].prev = -1; if (L->head >= 0) L->n[L->head].prev = i; else L->tail = i; L->head = i; L->size++; }
static int nl_erase(nlist_t *L, int i) { int p = L->n[i].prev, nx = L->n[i].next; if (p >= 0) L->n[p].next = nx; else L->head = nx; if (nx >= 0) L->n[nx].prev = p; else L->tail = p; L->size--; return nx; }

static void divide_node(nlist_t *L, const kp_t *kps, int src, int c[4]) {          /* ExtractorNode::DivideNode */
    for (int k = 0; k < 4; k++) c[k] = nl_new(L);
    node_t *s = &L->n[src];
    int halfX = (int)ceilf((float)(s->urx - s->ulx)/2), halfY = (int)ceilf((float)(s->bry - s->uly)/2);
    node_t *n1 = &L->n[c[0]], *n2 = &L->n[c[1]], *n3 = &L->n[c[2]], *n4 = &L->n[c[3]];
    n1->ulx = s->ulx; n1->uly = s->uly; n1->urx = s->ulx + halfX; n1->ury = s->uly; n1->blx = s->ulx; n1->bly = s->uly + halfY; n1->brx = s->ulx + halfX; n1->bry = s->uly + halfY;
    n2->ulx = n1->urx; n2->uly = n1->ury; n2->urx = s->urx; n2->ury = s->ury; n2->blx = n1->brx; n2->bly = n1->bry; n2->brx = s->urx; n2->bry = s->uly + halfY;
    n3->ulx = n1->blx; n3->uly = n1->bly; n3->urx = n1->brx; n3->ury = n1->bry; n3->blx = s->blx; n3->bly = s->bly; n3->brx = n1->brx; n3->bry = s->bly;
    n4->ulx = n3->urx; n4->uly = n3->ury; n4->urx = n2->brx; n4->ury = n2->bry; n4->blx = n3->brx; n4->bly = n3->bry; n4->brx = s->brx; n4->bry = s->bry;
    for (int k = 0; k < 4; k++) { L->n[c[k]].keys = (int *)malloc(sizeof(int)*(s->nk + 1)); L->n[c[k]].nk = 0; }
    for (int i = 0; i < s->nk; i++) {
        const kp_t *kp = &kps[s->keys[i]]; node_t *d;
        if (kp->x < n1->urx) d = (kp->y < n1->bry) ? n1 : n3;
        else d = (kp->y < n1->bry) ? n2 : n4;
        d->keys[d->nk++] = s->keys[i];
    }
    for (int k = 0; k < 4; k++) if (L->n[c[k]].nk == 1) L->n[c[k]].nomore = 1;
}
typedef struct { int size, node; } sn_t;
static nlist_t *g_sortL;
static int cmp_sn(const void *a, const void *b) {           /* std::sort of pair<int, ExtractorNode*>: (size, address := creation order) */
    const sn_t *x = (const sn_t *)a, *y = (const sn_t *)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    int ix = g_sortL->n[x->node].id, iy = g_sortL->n[y->node].id;
    return ix < iy ? -1 : (ix > iy ? 1 : 0);
}
/* returns selected keypoint indices (into kps) in list order */
static int distribute_octtree(const kp_t *kps, int nk, int minX, int maxX, int minY, int maxY, int N, int *sel) {
    nlist_t L; memset(&L, 0, sizeof(L)); L.head = L.tail = -1;
    const int nIni = (int)roundf((float)(maxX - minX)/(maxY - minY));
    const float hX = (float)(maxX - minX)/nIni;
    int *ini = (int *)malloc(sizeof(int)*(nIni > 0 ? nIni : 1));
    for (int i = 0; i < nIni; i++) {
        int q = nl_new(&L); node_t *n = &L.n[q];
        n->ulx = (int)(hX*(float)i); n->uly = 0; n->urx = (int)(hX*(float)(i + 1)); n->ury = 0;
        n->blx = n->ulx; n->bly = maxY - minY; n->brx = n->urx; n->bry = maxY - minY;
        n->keys = (int *)malloc(sizeof(int)*(nk + 1)); n->nk = 0;
        nl_push_back(&L, q); ini[i] = q;
    }
    for (int i = 0; i < nk; i++) { int q = ini[(int)(kps[i].x/hX)]; L.n[q].keys[L.n[q].nk++] = i; }
    for (int it = L.head; it >= 0; ) { node_t *n = &L.n[it]; if (n->nk == 1) { n->nomore = 1; it = n->next; } else if (n->nk == 0) it = nl_erase(&L, it); else it = n->next; }
    int finish = 0;
    sn_t *vs = (sn_t *)malloc(sizeof(sn_t)*(4*(size_t)nk + 16)); int nvs = 0;
    sn_t *vprev = (sn_t *)malloc(sizeof(sn_t)*(4*(size_t)nk + 16));
    while (!finish) {
        int prevSize = L.size, nToExpand = 0; nvs = 0;
        for (int it = L.head; it >= 0; ) {
            if (L.n[it].nomore) { it = L.n[it].next; continue; }
            int c[4]; divide_node(&L, kps, it, c);
            for (int k = 0; k < 4; k++) if (L.n[c[k]].nk > 0) {
                nl_push_front(&L, c[k]);
                if (L.n[c[k]].nk > 1) { nToExpand++; vs[nvs].size = L.n[c[k]].nk; vs[nvs].node = c[k]; nvs++; }
            }
            it = nl_erase(&L, it);
        }
        if (L.size >= N || L.size == prevSize) finish = 1;
        else if (L.size + nToExpand*3 > N) {
            while (!finish) {
                prevSize = L.size;
                int np = nvs; memcpy(vprev, vs, sizeof(sn_t)*np); nvs = 0;
                g_sortL = &L; qsort(vprev, np, sizeof(sn_t), cmp_sn);
                for (int j = np - 1; j >= 0; j--) {
                    int c[4]; divide_node(&L, kps, vprev[j].node, c);
                    for (int k = 0; k < 4; k++) if (L.n[c[k]].nk > 0) {
                        nl_push_front(&L, c[k]);
                        if (L.n[c[k]].nk > 1) { vs[nvs].size = L.n[c[k]].nk; vs[nvs].node = c[k]; nvs++; }
                    }
                    nl_erase(&L, vprev[j].node);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prevSize) finish = 1;
            }
        }
    }
    int ns = 0;
    for (int it = L.head; it >= 0; it = L.n[it].next) {
        node_t *n = &L.n[it]; int best = n->keys[0]; float mr = kps[best].response;
        for (int k = 1; k < n->nk; k++) if (kps[n->keys[k]].response > mr) { best = n->keys[k]; mr = kps[best].response; }
        sel[ns++] = best;
    }
    for (int i = 0; i < L.cnt; i++) free(L.n[i].keys);
    free(L.n); free(ini); free(vs); free(vprev);
    return ns;
}

/* OpenCV: cv::fastAtan2 (degrees) */
static float fast_atan2f(float y, float x) {
    const float p1 = 0.9997878412794807f*(float)(180/3.14159265358979323846), p3 = -0.3258083974640975f*(float)(180/3.14159265358979323846);
    const float p5 = 0.1555786518463281f*(float)(180/3.14159265358979323846), p7 = -0.04432655554792128f*(float)(180/3.14159265358979323846);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) { c = ay/(ax + (float)DBL_EPSILON); c2 = c*c; a = (((p7*c2 + p5)*c2 + p3)*c2 + p1)*c; }
    else { c = ax/(ay + (float)DBL_EPSILON); c2 = c*c; a = 90.f - (((p7*c2 + p5)*c2 + p3)*c2 + p1)*c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}
/* IC_Angle, ORBextractor.cc:77-104 */
static float ic_angle(const uint8_t *img, int stride, float px, float py, const int *umax) {
    int m01 = 0, m10 = 0;
    const uint8_t *c = img + (size_t)cv_round_f(py)*stride + cv_round_f(px);
    for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m10 += u*c[u];
    for (int v = 1; v <= HALF_PATCH; ++v) {
        int vs = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) { int vp = c[u + v*stride], vm = c[u - v*stride]; vs += (vp - vm); m10 += u*(vp + vm); }
        m01 += v*vs;
    }
    return fast_atan2f((float)m01, (float)m10);
}

/* OpenCV 3.3: GaussianBlur 7x7 sigma 2 on CV_8U = separable filter with the float kernel converted to 8-bit fixed point */
static void gauss_kernel_q8(int k[7]) {
    float cf[7]; double sum = 0, s2 = -0.5/(2.0*2.0);
    for (int i = 0; i < 7; i++) { double x = i - 3.0; cf[i] = (float)exp(s2*x*x); sum += cf[i]; }
    sum = 1./sum;
    for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i]*sum); k[i] = cv_round_f(cf[i]*256.f); }
}
static void gaussian_blur7(const uint8_t *src, int w, int h, int sstride, uint8_t *dst) {
    int k[7]; gauss_kernel_q8(k);
    int *tmp = (int *)malloc(sizeof(int)*(size_t)w*h);
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { int s = 0;
        for (int i = 0; i < 7; i++) s += k[i]*src[(size_t)y*sstride + reflect101(x + i - 3, w)];
        tmp[(size_t)y*w + x] = s; }
    for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) { int s = 0;
        for (int i = 0; i < 7; i++) s += k[i]*tmp[(size_t)reflect101(y + i - 3, h)*w + x];
        int v = (s + (1 << 15)) >> 16; dst[(size_t)y*w + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }
    free(tmp);
}

/* computeOrbDescriptor, ORBextractor.cc:108-147.  cos / sin: the reference calls std::cos(float); restated as the double-precision
 * function rounded to float so that CPU and GPU agree (libm vs device cosf differ in the last ulp). */
static void orb_descriptor(const uint8_t *img, int stride, float px, float py, float angle_deg, uint8_t *desc) {
    const float factorPI = (float)(3.14159265358979323846/180.f);
    float angle = angle_deg*factorPI;
    float a = (float)cos((double)angle), b = (float)sin((double)angle);
    const uint8_t *c = img + (size_t)cv_round_f(py)*stride + cv_round_f(px);
    const int8_t *pat = ORB_BIT_PATTERN_31;
    for (int i = 0; i < 32; i++, pat += 32) {
        int val = 0;
        for (int t = 0; t < 8; t++) {
            float x0 = pat[4*t], y0 = pat[4*t+1], x1 = pat[4*t+2], y1 = pat[4*t+3];
            int t0 = c[cv_round_f(x0*b + y0*a)*stride + cv_round_f(x0*a - y0*b)];
            int t1 = c[cv_round_f(x1*b + y1*a)*stride + cv_round_f(x1*a - y1*b)];
            val |= (t0 < t1) << t;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ---------------------------------------------------------------- public entry points */
/* ORBextractor::operator(), ORBextractor.cc:1054-1116 (+ ComputeKeyPointsOctTree :766-854).
 * kp[cap][6] = x, y, size, angle, response, octave.  Returns the number of keypoints (<= cap) or a negative error. */
int tsorb_oracle_extract(const uint8_t *img, int w, int h, int stride, int nfeatures, float scale, int nlevels, int ini_th, int min_th,
                         float *kp, uint8_t *desc, int cap) {
    if (!img || nlevels < 1 || nlevels > MAXL || w < 64 || h < 64) return -1;
    orb_t o; orb_init(&o, nfeatures, scale, nlevels, ini_th, min_th);
    compute_pyramid(&o, img, w, h, stride);
    int total = 0;
    const float Wc = 30;
    for (int l = 0; l < nlevels; l++) {
        const int bw = o.w[l] + 2*EDGE;
        const uint8_t *lev = o.pyr[l] + (size_t)EDGE*bw + EDGE;          /* the ROI view mvImagePyramid[level] */
        const int minBX = EDGE - 3, minBY = minBX, maxBX = o.w[l] - EDGE + 3, maxBY = o.h[l] - EDGE + 3;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
        const int nCols = (int)(width/Wc), nRows = (int)(height/Wc);
        const int wCell = (int)ceilf(width/nCols), hCell = (int)ceilf(height/nRows);
        int capc = (maxBX - minBX)*(maxBY - minBY)/4 + 64;        /* strict 3x3 NMS: at most one corner per 2x2 block */
        kp_t *cand = (kp_t *)malloc(sizeof(kp_t)*capc); int nc = 0;
        kp_t *cell = (kp_t *)malloc(sizeof(kp_t)*4096);
        for (int i = 0; i < nRows; i++) {
            const float iniY = (float)(minBY + i*hCell); float maxY = iniY + hCell + 6;
            if (iniY >= maxBY - 3) continue;
            if (maxY > maxBY) maxY = (float)maxBY;
            for (int j = 0; j < nCols; j++) {
                const float iniX = (float)(minBX + j*wCell); float maxX = iniX + wCell + 6;
                if (iniX >= maxBX - 6) continue;
                if (maxX > maxBX) maxX = (float)maxBX;
                const int rx = (int)iniX, ry = (int)iniY, rw = (int)maxX - rx, rh = (int)maxY - ry;
                int n = fast_roi(lev + (size_t)ry*bw + rx, bw, rw, rh, ini_th, cell, 4096);
                if (n == 0) n = fast_roi(lev + (size_t)ry*bw + rx, bw, rw, rh, min_th, cell, 4096);
                for (int q = 0; q < n && nc < capc; q++) { cand[nc] = cell[q]; cand[nc].x += j*wCell; cand[nc].y += i*hCell; nc++; }
            }
        }
        int *sel = (int *)malloc(sizeof(int)*(nc + 16));
        int ns = nc > 0 ? distribute_octtree(cand, nc, minBX, maxBX, minBY, maxBY, o.nfl[l], sel) : 0;
        const int scaledPatch = (int)(PATCH_SIZE*o.sf[l]);
        if (ns > 0) { o.blur[l] = (uint8_t *)malloc((size_t)o.w[l]*o.h[l]); gaussian_blur7(lev, o.w[l], o.h[l], bw, o.blur[l]); }
        for (int q = 0; q < ns && total < cap; q++) {
            float x = cand[sel[q]].x + minBX, y = cand[sel[q]].y + minBY;
            float ang = ic_angle(lev, bw, x, y, o.umax);
            if (desc) orb_descriptor(o.blur[l], o.w[l], x, y, ang, desc + 32*(size_t)total);
            float sc = o.sf[l];
            float *k = kp + 6*(size_t)total;
            k[0] = l ? x*sc : x; k[1] = l ? y*sc : y; k[2] = (float)scaledPatch; k[3] = ang; k[4] = cand[sel[q]].response; k[5] = (float)l;
            total++;
        }
        free(cand); free(cell); free(sel);
    }
    for (int l = 0; l < nlevels; l++) { free(o.pyr[l]); free(o.blur[l]); }
    return total;
}

/* stage hooks for the tests */
int tsorb_oracle_level(const uint8_t *img, int w, int h, int stride, float scale, int nlevels, int level, int blurred, uint8_t *out, int *lw, int *lh) {
    orb_t o; orb_init(&o, 1000, scale, nlevels, 20, 7);
    compute_pyramid(&o, img, w, h, stride);
    int bw = o.w[level] + 2*EDGE, bh = o.h[level] + 2*EDGE;
    *lw = o.w[level]; *lh = o.h[level];
    if (blurred) gaussian_blur7(o.pyr[level] + (size_t)EDGE*bw + EDGE, o.w[level], o.h[level], bw, out);
    else memcpy(out, o.pyr[level], (size_t)bw*bh);
    for (int l = 0; l < nlevels; l++) free(o.pyr[l]);
    return 0;
}
int tsorb_oracle_fast(const uint8_t *img, int w, int h, int stride, int threshold, float *kp3, int cap) {
    kp_t *k = (kp_t *)malloc(sizeof(kp_t)*cap);
    int n = fast_roi(img, stride, w, h, threshold, k, cap);
    for (int i = 0; i < n; i++) { kp3[3*i] = k[i].x; kp3[3*i+1] = k[i].y; kp3[3*i+2] = k[i].response; }
    free(k); return n;
}
void tsorb_oracle_params(int nfeatures, float scale, int nlevels, float *sf, int *nfl, int *umax16, int *gk7) {
    orb_t o; orb_init(&o, nfeatures, scale, nlevels, 20, 7);
    for (int i = 0; i < nlevels; i++) { sf[i] = o.sf[i]; nfl[i] = o.nfl[i]; }
    for (int i = 0; i < 16; i++) umax16[i] = o.umax[i];
    gauss_kernel_q8(gk7);
}
float tsorb_oracle_atan2(float y, float x) { return fast_atan2f(y, x); }

/* ------------------------------------------------------------------ window / projection search
 * frame::AssignFeaturesToGrid + PosInGrid (src/frame.cc:372-407), frame::GetFeaturesInArea (src/frame.cc:415-468),
 * tracking::DescriptorDistance (src/tracking.cc:2762-2778) and the best / second-best scan shared by tracking::SearchFrom3D,
 * SearchFrom3DAdd, SearchFrom3DLocalTrack and (without its running vMatchDist filter) SearchForInitializ (tracking.cc:1045-1400).
 * kp6: [n][6] = x, y, size, angle, response, octave (the extractor's output).  Per query: candidates in the reference's order
 * (cell column ix outer, cell row iy inner, features of a cell in index order), their Hamming distances, the first minimum
 * (strict <) and the second-best distance.  PARITY UNPINNED like the rest of this file (no OpenCV / reference build here). */
#define GRID_COLS 64
#define GRID_ROWS 48
static int hamming256(const uint8_t *a, const uint8_t *b) {
    const int32_t *pa = (const int32_t *)a, *pb = (const int32_t *)b;
    int dist = 0;
    for (int i = 0; i < 8; i++) {
        unsigned int v = (unsigned int)(pa[i] ^ pb[i]);
        v = v - ((v >> 1) & 0x55555555);
        v = (v & 0x33333333) + ((v >> 2) & 0x33333333);
        dist += (((v + (v >> 4)) & 0xF0F0F0F) * 0x1010101) >> 24;
    }
    return dist;
}
int tsorb_oracle_match(const float *kp6, const uint8_t *desc, int n, double min_x, double max_x, double min_y, double max_y,
                       int nq, const float *qxy, const float *qr, const int32_t *qlev, const uint8_t *qdesc, int max_cand,
                       int32_t *cand_idx, int32_t *cand_dist, int32_t *cand_cnt, int32_t *best_idx, int32_t *best_dist, int32_t *best_dist2) {
    const double iw = (double)GRID_COLS/(max_x - min_x), ih = (double)GRID_ROWS/(max_y - min_y);
    /* mGrid[ix][iy]: feature indices in insertion (= index) order */
    int *cnt = (int *)calloc(GRID_COLS*GRID_ROWS + 1, sizeof(int)), *cell = (int *)malloc(sizeof(int)*(size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; i++) {
        int px = (int)round(((double)kp6[6*i] - min_x)*iw), py = (int)round(((double)kp6[6*i+1] - min_y)*ih);
        cell[i] = (px < 0 || px >= GRID_COLS || py < 0 || py >= GRID_ROWS) ? -1 : px*GRID_ROWS + py;
        if (cell[i] >= 0) cnt[cell[i] + 1]++;
    }
    for (int c = 0; c < GRID_COLS*GRID_ROWS; c++) cnt[c+1] += cnt[c];
    int *list = (int *)malloc(sizeof(int)*(size_t)(n > 0 ? n : 1)), *cur = (int *)malloc(sizeof(int)*GRID_COLS*GRID_ROWS);
    memcpy(cur, cnt, sizeof(int)*GRID_COLS*GRID_ROWS);
    for (int i = 0; i < n; i++) if (cell[i] >= 0) list[cur[cell[i]]++] = i;
    for (int q = 0; q < nq; q++) {
        const float x = qxy[2*q], y = qxy[2*q+1], r = qr[q];
        const int minLevel = qlev ? qlev[2*q] : -1, maxLevel = qlev ? qlev[2*q+1] : -1;
        int nc = 0, bi = -1, bd = 2147483647, bd2 = 2147483647;
        const int c0x = (int)floor(((double)x - min_x - (double)r)*iw) > 0 ? (int)floor(((double)x - min_x - (double)r)*iw) : 0;
        const int c1x = (int)ceil(((double)x - min_x + (double)r)*iw) < GRID_COLS - 1 ? (int)ceil(((double)x - min_x + (double)r)*iw) : GRID_COLS - 1;
        const int c0y = (int)floor(((double)y - min_y - (double)r)*ih) > 0 ? (int)floor(((double)y - min_y - (double)r)*ih) : 0;
        const int c1y = (int)ceil(((double)y - min_y + (double)r)*ih) < GRID_ROWS - 1 ? (int)ceil(((double)y - min_y + (double)r)*ih) : GRID_ROWS - 1;
        if (!(c0x >= GRID_COLS || c1x < 0 || c0y >= GRID_ROWS || c1y < 0)) {
            const int check = (minLevel > 0) || (maxLevel >= 0);
            for (int ix = c0x; ix <= c1x; ix++) for (int iy = c0y; iy <= c1y; iy++) {
                const int c = ix*GRID_ROWS + iy;
                for (int k = cnt[c]; k < cnt[c+1]; k++) {
                    const int i = list[k]; const int oct = (int)kp6[6*i+5];
                    if (check) { if (oct < minLevel) continue; if (maxLevel >= 0 && oct > maxLevel) continue; }
                    const float dx = kp6[6*i] - x, dy = kp6[6*i+1] - y;
                    if (!(fabsf(dx) < r && fabsf(dy) < r)) continue;
                    const int d = hamming256(qdesc + 32*(size_t)q, desc + 32*(size_t)i);
                    if (nc < max_cand) { cand_idx[(size_t)q*max_cand + nc] = i; cand_dist[(size_t)q*max_cand + nc] = d; }
                    nc++;
                    if (d < bd) { bd2 = bd; bd = d; bi = i; } else if (d < bd2) bd2 = d;
                }
            }
        }
        cand_cnt[q] = nc; best_idx[q] = bi; best_dist[q] = bd; best_dist2[q] = bd2;
    }
    free(cnt); free(cell); free(list); free(cur);
    return 0;
}
