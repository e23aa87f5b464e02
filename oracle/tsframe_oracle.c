/* TEST INFRASTRUCTURE -- NOT PRODUCT CODE.  PARITY UNPINNED (no OpenCV in this image to run the reference's calls against).
 *
 * Plain-C restatement of the reference's BA-pyramid and reference-feature construction:
 *   frame::GetPyrMat                 src/frame.cc:178-204   cv::pyrDown chain, cv::Sobel (ddepth = CV_8U) x / y, cv::addWeighted(.5, .5)
 *   tool::GetPyramidPts (text)       src/tool.cc:564-710    per level: grid over the detection box, one feature per cell
 *   tool::GetPyramidPts (scene)      src/tool.cc:862-980    per level: grid over the image, one feature per cell
 *   tool::GetIntenBilinterPtr        src/tool.cc:1150-1176  bilinear sample of a uint8 image in double
 *   tool::CalNormvec / GetNeighbour  src/tool.cc:1342-1364,1540-1566  INTERVAL8 neighbour intensities, raw and (I - mu) / sigma
 *
 * OpenCV behaviour restated (3.x, 8-bit paths):
 *   pyrDown   5x5 Gaussian [1 4 6 4 1] x [1 4 6 4 1], BORDER_REFLECT_101, dst = ((w + 1) / 2, (h + 1) / 2), (sum + 128) >> 8
 *   Sobel     ksize 3, scale 1, delta 0, BORDER_REFLECT_101, result saturate_cast<uchar> (negative gradients clip to 0)
 *   addWeighted on 8U: float t = a * 0.5f + b * 0.5f; saturate_cast<uchar>(t) = cvRound(t) = round half to even
 *
 * Reference quirk kept on purpose: the per-cell arg-max never updates MAX (tool.cc:678-685, :951-957), so the selected feature of a
 * cell is the LAST one (in input order) whose gradient is > 0 (text) / >= 0 (scene: always, i.e. simply the last one). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2*n - 2 - p; }
    return p;
}

/* cv::pyrDown, 8U single channel.  dst is (w + 1) / 2 x (h + 1) / 2 */
void tsframe_oracle_pyrdown(const uint8_t *src, int w, int h, uint8_t *dst) {
    const int dw = (w + 1)/2, dh = (h + 1)/2;
    int *row = (int *)malloc(sizeof(int)*(size_t)dw*5);
    for (int y = 0; y < dh; y++) {
        for (int k = 0; k < 5; k++) {
            const uint8_t *s = src + (size_t)reflect101(2*y - 2 + k, h)*w;
            for (int x = 0; x < dw; x++)
                row[k*dw + x] = s[reflect101(2*x - 2, w)] + 4*s[reflect101(2*x - 1, w)] + 6*s[reflect101(2*x, w)] + 4*s[reflect101(2*x + 1, w)] + s[reflect101(2*x + 2, w)];
        }
        for (int x = 0; x < dw; x++)
            dst[(size_t)y*dw + x] = (uint8_t)((row[x] + 4*row[dw + x] + 6*row[2*dw + x] + 4*row[3*dw + x] + row[4*dw + x] + 128) >> 8);
    }
    free(row);
}

static uint8_t sat_u8(int v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* cv::Sobel(src, gx, CV_8U, 1, 0, 3), cv::Sobel(src, gy, CV_8U, 0, 1, 3), cv::addWeighted(gx, .5, gy, .5, 0, grad) */
void tsframe_oracle_gradients(const uint8_t *src, int w, int h, uint8_t *gx, uint8_t *gy, uint8_t *grad) {
    for (int y = 0; y < h; y++) {
        const uint8_t *r0 = src + (size_t)reflect101(y - 1, h)*w, *r1 = src + (size_t)y*w, *r2 = src + (size_t)reflect101(y + 1, h)*w;
        for (int x = 0; x < w; x++) {
            const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
            const int sx = (r0[xp] - r0[xm]) + 2*(r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
            const int sy = (r2[xm] - r0[xm]) + 2*(r2[x] - r0[x]) + (r2[xp] - r0[xp]);
            const uint8_t a = sat_u8(sx), b = sat_u8(sy);
            gx[(size_t)y*w + x] = a; gy[(size_t)y*w + x] = b;
            const int s = a + b;                                   /* a/2 + b/2 exactly in float; .5 ties to even */
            grad[(size_t)y*w + x] = (uint8_t)((s & 1) ? ((s >> 1) + ((s >> 1) & 1)) : (s >> 1));
        }
    }
}

/* tool::GetIntenBilinterPtr */
int tsframe_oracle_bilinear(const uint8_t *img, int w, int h, double u, double v, double *out) {
    const int x0 = (int)floor(u), y0 = (int)floor(v), x1 = (int)ceil(u), y1 = (int)ceil(v);
    if (x0 < 0 || y0 < 0 || x1 >= w || y1 >= h) { *out = 0.0; return 0; }
    const double a = u - x0, b = v - y0;
    const double wtl = (1.0 - a)*(1.0 - b), wtr = a*(1.0 - b), wbl = (1.0 - a)*b, wbr = a*b;
    /* the reference reads ptr[1] / ptr[stride] / ptr[stride + 1] even when u or v is integral (weight 0, possibly one past the row or
     * the image): clamped here, which cannot change the value */
    const int xr = x0 + 1 < w ? x0 + 1 : w - 1, yb = y0 + 1 < h ? y0 + 1 : h - 1;
    *out = wtl*img[(size_t)y0*w + x0] + wtr*img[(size_t)y0*w + xr] + wbl*img[(size_t)yb*w + x0] + wbr*img[(size_t)yb*w + xr];
    return 1;
}

/* tool::GetPyramidPts.  mode 0 = text (grid over the box pmin..pmax, gradient > 0), 1 = scene (grid over the image, >= 0).
 * xy: n raw features (float, level-0 pixels).  imgs / grads: per level (w[l] x h[l]).  inv_scale[l].
 * Outputs, level-major with level_off[n_levels + 1]: u, v (level coordinates), idx (IdxToRaw), inten (bilinear on imgs[l]), in (flag).
 * Returns the total count (capacity needed: n * n_levels). */
int tsframe_oracle_pyramid_pts(int mode, const float *xy, int n, const double *box, int n_levels, const uint8_t *const *imgs,
                               const uint8_t *const *grads, const int *w, const int *h, const double *inv_scale,
                               int *level_off, double *u, double *v, int *idx, double *inten, uint8_t *in) {
    int cnt = 0;
    level_off[0] = 0;
    for (int j = 0; j < n; j++) {                                   /* level 0: every raw feature */
        u[cnt] = xy[2*j]; v[cnt] = xy[2*j + 1]; idx[cnt] = j;
        in[cnt] = (uint8_t)tsframe_oracle_bilinear(imgs[0], w[0], h[0], u[cnt], v[cnt], &inten[cnt]);
        cnt++;
    }
    level_off[1] = cnt;
    for (int l = 1; l < n_levels; l++) {
        const double s = inv_scale[l];
        const size_t ncell = (size_t)((double)n*s*s + (mode == 0 ? 100 : 500));
        double x0 = 0.0, y0 = 0.0, WH, fx, fy; int cw, ch;
        if (mode == 0) {
            const double pminx = box[0]*s, pminy = box[1]*s, pmaxx = box[2]*s, pmaxy = box[3]*s;
            WH = (pmaxx - pminx)/(pmaxy - pminy);
            ch = (int)sqrt((double)ncell/WH); cw = (int)sqrt((double)ncell*WH);
            fx = (pmaxx - pminx)/(double)cw; fy = (pmaxy - pminy)/(double)ch;
            x0 = pminx; y0 = pminy;
        } else {
            WH = (double)w[l]/(double)h[l];
            ch = (int)sqrt((double)ncell/WH); cw = (int)sqrt((double)ncell*WH);
            fx = (double)w[l]/(double)cw; fy = (double)h[l]/(double)ch;
        }
        int *sel = (int *)malloc(sizeof(int)*(size_t)cw*ch);
        for (int k = 0; k < cw*ch; k++) sel[k] = -1;
        for (int j = 0; j < n; j++) {
            const double pu = (double)xy[2*j]*s, pv = (double)xy[2*j + 1]*s;
            double g; tsframe_oracle_bilinear(grads[l], w[l], h[l], pu, pv, &g);
            int m = (int)round(mode == 0 ? (pu - x0)/fx : pu/fx), q = (int)round(mode == 0 ? (pv - y0)/fy : pv/fy);
            if (m == cw) m = cw - 1;
            if (q == ch) q = ch - 1;
            if (m < 0 || q < 0 || m >= cw || q >= ch) continue;       /* (outside the grid: undefined in the reference, dropped here) */
            if (mode == 0 ? (g > 0.0) : (g >= 0.0)) sel[q*cw + m] = j;   /* MAX is never updated: the last qualifying feature wins */
        }
        for (int i3 = 0; i3 < cw; i3++)
            for (int i4 = 0; i4 < ch; i4++) {
                const int j = sel[i4*cw + i3];
                if (j < 0) continue;
                u[cnt] = (double)xy[2*j]*s; v[cnt] = (double)xy[2*j + 1]*s; idx[cnt] = j;
                in[cnt] = (uint8_t)tsframe_oracle_bilinear(imgs[l], w[l], h[l], u[cnt], v[cnt], &inten[cnt]);
                cnt++;
            }
        free(sel);
        level_off[l + 1] = cnt;
    }
    return cnt;
}

/* tool::CalNormvec -> GetNeighbour(INTERVAL8): out_inten / out_ninten [n][8], out_in [n] = IN flag of the LAST tap (as the reference) */
static const double NB_DX[8] = { 0, 2, 1, 0, -1, -2, -1, 0 }, NB_DY[8] = { 0, 0, -1, -2, -1, 0, 1, 2 };
void tsframe_oracle_neighbours(const uint8_t *img, int w, int h, const double *uv, int n, double mu, double sigma,
                               double *out_inten, double *out_ninten, uint8_t *out_in) {
    for (int j = 0; j < n; j++)
        for (int k = 0; k < 8; k++) {
            double I;
            out_in[j] = (uint8_t)tsframe_oracle_bilinear(img, w, h, uv[2*j] + NB_DX[k], uv[2*j + 1] + NB_DY[k], &I);
            out_inten[8*j + k] = I; out_ninten[8*j + k] = (I - mu)/sigma;
        }
}
