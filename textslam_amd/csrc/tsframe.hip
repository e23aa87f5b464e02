// libtsframe.so -- BA pyramid, gradient planes, per-level feature selection and INTERVAL8 reference intensities on gfx950
// (include/tsframe.h).  Integer image arithmetic is exact; the fp64 sampling is compiled without FMA
// contraction so that it rounds like the CPU restatement (oracle/tsframe_oracle.c).  Everything here is HBM-bound byte work:
// one thread per output pixel with coalesced rows, no LDS tiling needed at 640x480 (the 5x5 / 3x3 footprints live in L2).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>
#include "../../include/tsframe.h"
#include "tsraster.h"

struct FCtx {
    int device = 0; hipStream_t stream = nullptr; std::string err;
    int n_levels = 0, w[TSFRAME_MAX_LEVELS] = {0}, h[TSFRAME_MAX_LEVELS] = {0};
    uint8_t *plane[4][TSFRAME_MAX_LEVELS] = {{nullptr}};     // device planes: img, grad, gx, gy
    size_t plane_cap = 0; uint8_t *plane_base = nullptr;
    uint8_t *h_stage = nullptr; size_t h_cap = 0;             // pinned staging (image in, results out)
    uint8_t *d_work = nullptr; size_t d_cap = 0;              // device scratch for the feature calls
};
#define CKF(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { c->err = std::string(#x) + ": " + hipGetErrorString(e_); return TSFRAME_ERR_DEVICE; } } while (0)

__device__ __forceinline__ int reflect101(int p, int n) {
    if (n == 1) return 0;
    while (p < 0 || p >= n) { if (p < 0) p = -p; else p = 2*n - 2 - p; }
    return p;
}

// cv::pyrDown (8U): [1 4 6 4 1] x [1 4 6 4 1], BORDER_REFLECT_101, (sum + 128) >> 8
__global__ __launch_bounds__(256) void k_pyrdown(const uint8_t *__restrict__ src, int w, int h, uint8_t *__restrict__ dst, int dw, int dh) {
    const int x = blockIdx.x*64 + (threadIdx.x & 63), y = blockIdx.y*4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    int xs[5];
#pragma unroll
    for (int k = 0; k < 5; k++) xs[k] = reflect101(2*x - 2 + k, w);
    const int wk[5] = { 1, 4, 6, 4, 1 };
    int sum = 0;
#pragma unroll
    for (int r = 0; r < 5; r++) {
        const uint8_t *s = src + (size_t)reflect101(2*y - 2 + r, h)*w;
        sum += wk[r]*(s[xs[0]] + 4*s[xs[1]] + 6*s[xs[2]] + 4*s[xs[3]] + s[xs[4]]);
    }
    dst[(size_t)y*dw + x] = (uint8_t)((sum + 128) >> 8);
}

// cv::Sobel x / y (CV_8U: saturating) and addWeighted(.5, .5) (float, round half to even)
__global__ __launch_bounds__(256) void k_gradients(const uint8_t *__restrict__ src, int w, int h, uint8_t *__restrict__ gx, uint8_t *__restrict__ gy, uint8_t *__restrict__ grad) {
    const int x = blockIdx.x*64 + (threadIdx.x & 63), y = blockIdx.y*4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const uint8_t *r0 = src + (size_t)reflect101(y - 1, h)*w, *r1 = src + (size_t)y*w, *r2 = src + (size_t)reflect101(y + 1, h)*w;
    const int xm = reflect101(x - 1, w), xp = reflect101(x + 1, w);
    const int sx = (r0[xp] - r0[xm]) + 2*(r1[xp] - r1[xm]) + (r2[xp] - r2[xm]);
    const int sy = (r2[xm] - r0[xm]) + 2*(r2[x] - r0[x]) + (r2[xp] - r0[xp]);
    const int a = min(max(sx, 0), 255), b = min(max(sy, 0), 255), s = a + b;
    gx[(size_t)y*w + x] = (uint8_t)a; gy[(size_t)y*w + x] = (uint8_t)b;
    grad[(size_t)y*w + x] = (uint8_t)((s & 1) ? ((s >> 1) + ((s >> 1) & 1)) : (s >> 1));
}

// tool::GetIntenBilinterPtr (reads of zero-weight neighbours past the image are clamped: they do not change the value)
__device__ __forceinline__ bool bilinear(const uint8_t *__restrict__ img, int w, int h, double u, double v, double &out) {
    const int x0 = (int)floor(u), y0 = (int)floor(v), x1 = (int)ceil(u), y1 = (int)ceil(v);
    if (x0 < 0 || y0 < 0 || x1 >= w || y1 >= h) { out = 0.0; return false; }
    const double a = u - x0, b = v - y0;
    const double wtl = (1.0 - a)*(1.0 - b), wtr = a*(1.0 - b), wbl = (1.0 - a)*b, wbr = a*b;
    const int xr = min(x0 + 1, w - 1), yb = min(y0 + 1, h - 1);
    const double p00 = img[(size_t)y0*w + x0], p01 = img[(size_t)y0*w + xr], p10 = img[(size_t)yb*w + x0], p11 = img[(size_t)yb*w + xr];
    out = wtl*p00 + wtr*p01 + wbl*p10 + wbr*p11;
    return true;
}

struct GridDev { int mode, cw, ch; double s, x0, y0, fx, fy; };

// per raw feature: gradient sample, cell, "last qualifying feature of the cell" = max index (tool.cc:678-685: MAX is never updated)
__global__ __launch_bounds__(256) void k_pts_cells(const float *__restrict__ xy, int n, const uint8_t *__restrict__ grad, int w, int h, GridDev G, int *sel) {
    const int j = blockIdx.x*256 + threadIdx.x;
    if (j >= n) return;
    const double pu = (double)xy[2*j]*G.s, pv = (double)xy[2*j + 1]*G.s;
    double g; bilinear(grad, w, h, pu, pv, g);
    int m = (int)round(G.mode == 0 ? (pu - G.x0)/G.fx : pu/G.fx), q = (int)round(G.mode == 0 ? (pv - G.y0)/G.fy : pv/G.fy);
    if (m == G.cw) m = G.cw - 1;
    if (q == G.ch) q = G.ch - 1;
    if (m < 0 || q < 0 || m >= G.cw || q >= G.ch) return;
    if (G.mode == 0 ? (g > 0.0) : (g >= 0.0)) atomicMax(&sel[q*G.cw + m], j);
}

// level 0: every raw feature; level l: the cells in the reference's visiting order (x outer, y inner), ordered compaction by one
// workgroup (a few thousand cells), appended after the previous levels (running offset in cnt[0])
__global__ __launch_bounds__(1024) void k_pts_emit(const float *__restrict__ xy, int n, int level, const uint8_t *__restrict__ img, int w, int h, GridDev G,
                                                   const int *__restrict__ sel, int *cnt, int *level_off,
                                                   double *u, double *v, int *idx, double *inten, uint8_t *in) {
    __shared__ int s_scan[1024]; __shared__ int s_base;
    const int tid = threadIdx.x;
    const int base0 = level == 0 ? 0 : cnt[0];
    if (level == 0) {
        for (int j = tid; j < n; j += 1024) {
            const double pu = xy[2*j], pv = xy[2*j + 1];
            u[j] = pu; v[j] = pv; idx[j] = j;
            double I; in[j] = bilinear(img, w, h, pu, pv, I) ? 1 : 0; inten[j] = I;
        }
        if (tid == 0) { level_off[0] = 0; level_off[1] = n; cnt[0] = n; }
        return;
    }
    if (tid == 0) s_base = 0;
    __syncthreads();
    const int ncell = G.cw*G.ch;
    for (int c0 = 0; c0 < ncell; c0 += 1024) {
        const int o = c0 + tid;                                // visiting order: o = i3 * ch + i4
        int j = -1;
        if (o < ncell) { const int i3 = o / G.ch, i4 = o - i3*G.ch; j = sel[i4*G.cw + i3]; }
        s_scan[tid] = j >= 0 ? 1 : 0;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {                   // inclusive scan
            const int t = tid >= d ? s_scan[tid - d] : 0;
            __syncthreads();
            s_scan[tid] += t;
            __syncthreads();
        }
        const int rank = s_base + s_scan[tid] - 1, total = s_scan[1023];
        if (j >= 0) {
            const int o2 = base0 + rank;
            const double pu = (double)xy[2*j]*G.s, pv = (double)xy[2*j + 1]*G.s;
            u[o2] = pu; v[o2] = pv; idx[o2] = j;
            double I; in[o2] = bilinear(img, w, h, pu, pv, I) ? 1 : 0; inten[o2] = I;
        }
        __syncthreads();
        if (tid == 0) s_base += total;
        __syncthreads();
    }
    if (tid == 0) { cnt[0] = base0 + s_base; level_off[level + 1] = base0 + s_base; }
}

__device__ __constant__ double NB_DX[8] = { 0, 2, 1, 0, -1, -2, -1, 0 };
__device__ __constant__ double NB_DY[8] = { 0, 0, -1, -2, -1, 0, 1, 2 };
// tool::GetNeighbour(INTERVAL8): thread = (feature, tap)
__global__ __launch_bounds__(256) void k_neighbours(const uint8_t *__restrict__ img, int w, int h, const double *__restrict__ uv, int n, double mu, double sigma,
                                                    double *inten8, double *ninten8, uint8_t *in) {
    const int e = blockIdx.x*256 + threadIdx.x, j = e >> 3, k = e & 7;
    if (j >= n) return;
    double I; const bool ok = bilinear(img, w, h, uv[2*j] + NB_DX[k], uv[2*j + 1] + NB_DY[k], I);
    inten8[e] = I; ninten8[e] = (I - mu)/sigma;
    if (k == 7) in[j] = ok ? 1 : 0;                            // feat->IN keeps the flag of the last tap
}

// ------------------------------------------------------------------------------------------------ host side
// ---- tool::GetBoxAllPixs (tool.cc:1264-1337): every pixel of the level image inside the filled detection quad, in row-major order
// of the clamped bounding box.  One workgroup: scan conversion of the quad into a bit mask (cv::fillPoly semantics, tsraster.h), then an
// ordered compaction of the box in tiles of 1024 pixels (ballot + wave counts).
struct BoxDev { int xy[8]; int x0, x1, y0, y1; };
__global__ __launch_bounds__(1024) void k_box_pixels(const uint8_t *__restrict__ img, int w, int h, BoxDev B, double mu, double sigma, unsigned *mask,
                                                     int *cnt, int *__restrict__ u, int *__restrict__ v, double *__restrict__ inten, double *__restrict__ ninten) {
    __shared__ int s_xy[8], s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid < 8) s_xy[tid] = B.xy[tid];
    for (int k = tid; k < (w*h + 31) >> 5; k += 1024) mask[k] = 0;
    __syncthreads();
    raster_quad(mask, s_xy, w, h, tid, 1024);
    __syncthreads();
    const int bw = B.x1 - B.x0 + 1, bh = B.y1 - B.y0 + 1, npx = bw*bh;
    int base = 0;
    for (int k0 = 0; k0 < npx; k0 += 1024) {
        const int k = k0 + tid;
        int x = 0, y = 0; bool in = false;
        if (k < npx) { x = B.x0 + k % bw; y = B.y0 + k / bw; const int bit = y*w + x; in = (mask[bit >> 5] >> (bit & 31)) & 1u; }
        const unsigned long long bal = __ballot(in);
        if (lane == 0) s_w[wave] = __popcll(bal);
        __syncthreads();
        int off = base, tot = 0;
        for (int q = 0; q < 16; q++) { const int c = s_w[q]; if (q < wave) off += c; tot += c; }
        if (in) {
            const int at = off + __popcll(bal & ((1ull << lane) - 1ull));
            const double I = (double)img[y*w + x];
            u[at] = x; v[at] = y; inten[at] = I; ninten[at] = (I - mu)/sigma;
        }
        base += tot;
        __syncthreads();
    }
    if (tid == 0) *cnt = base;
}

static int ensure_host(FCtx *c, size_t bytes) {
    if (bytes <= c->h_cap) return 0;
    if (c->h_stage) hipHostFree(c->h_stage);
    c->h_stage = nullptr; c->h_cap = 0;
    CKF(hipHostMalloc((void **)&c->h_stage, bytes, hipHostMallocDefault));
    c->h_cap = bytes; return 0;
}
static int ensure_work(FCtx *c, size_t bytes) {
    if (bytes <= c->d_cap) return 0;
    if (c->d_work) hipFree(c->d_work);
    c->d_work = nullptr; c->d_cap = 0;
    CKF(hipMalloc((void **)&c->d_work, bytes));
    c->d_cap = bytes; return 0;
}

extern "C" {

int tsframe_create(int device, void **ctx) {
    if (!ctx) return TSFRAME_ERR_ARG;
    int nd = 0;
    if (hipGetDeviceCount(&nd) != hipSuccess || device < 0 || device >= nd) return TSFRAME_ERR_DEVICE;      // no GPU: fail loudly, no CPU path
    FCtx *c = new FCtx(); c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&c->stream) != hipSuccess) { delete c; return TSFRAME_ERR_DEVICE; }
    *ctx = c; return TSFRAME_OK;
}
void tsframe_destroy(void *ctx) {
    FCtx *c = (FCtx *)ctx; if (!c) return;
    hipSetDevice(c->device);
    if (c->plane_base) hipFree(c->plane_base);
    if (c->d_work) hipFree(c->d_work);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}
const char *tsframe_last_error(void *ctx) { FCtx *c = (FCtx *)ctx; return c ? c->err.c_str() : "null context"; }

int tsframe_set_image(void *ctx, const uint8_t *img, int w, int h, int n_levels) {
    FCtx *c = (FCtx *)ctx; if (!c || !img || w < 2 || h < 2 || n_levels < 1 || n_levels > TSFRAME_MAX_LEVELS) return TSFRAME_ERR_ARG;
    hipSetDevice(c->device);
    size_t tot = 0; int lw = w, lh = h;
    for (int l = 0; l < n_levels; l++) { c->w[l] = lw; c->h[l] = lh; tot += ((size_t)lw*lh + 255) & ~(size_t)255; lw = (lw + 1)/2; lh = (lh + 1)/2; }
    if (4*tot > c->plane_cap) {
        if (c->plane_base) hipFree(c->plane_base);
        c->plane_base = nullptr; c->plane_cap = 0;
        CKF(hipMalloc((void **)&c->plane_base, 4*tot));
        c->plane_cap = 4*tot;
    }
    uint8_t *p = c->plane_base;
    for (int k = 0; k < 4; k++) for (int l = 0; l < n_levels; l++) { c->plane[k][l] = p; p += ((size_t)c->w[l]*c->h[l] + 255) & ~(size_t)255; }
    c->n_levels = n_levels;
    int rc = ensure_host(c, (size_t)w*h); if (rc) return rc;
    memcpy(c->h_stage, img, (size_t)w*h);
    CKF(hipMemcpyAsync(c->plane[TSFRAME_IMG][0], c->h_stage, (size_t)w*h, hipMemcpyHostToDevice, c->stream));
    for (int l = 0; l < n_levels; l++) {
        if (l > 0) hipLaunchKernelGGL(k_pyrdown, dim3((c->w[l] + 63)/64, (c->h[l] + 3)/4), dim3(256), 0, c->stream,
                                      (const uint8_t *)c->plane[TSFRAME_IMG][l - 1], c->w[l - 1], c->h[l - 1], c->plane[TSFRAME_IMG][l], c->w[l], c->h[l]);
        hipLaunchKernelGGL(k_gradients, dim3((c->w[l] + 63)/64, (c->h[l] + 3)/4), dim3(256), 0, c->stream,
                           (const uint8_t *)c->plane[TSFRAME_IMG][l], c->w[l], c->h[l], c->plane[TSFRAME_GRADX][l], c->plane[TSFRAME_GRADY][l], c->plane[TSFRAME_GRAD][l]);
    }
    CKF(hipStreamSynchronize(c->stream)); CKF(hipGetLastError());
    return TSFRAME_OK;
}
int tsframe_level_size(void *ctx, int level, int *w, int *h) {
    FCtx *c = (FCtx *)ctx; if (!c || level < 0 || level >= c->n_levels) return TSFRAME_ERR_ARG;
    if (w) *w = c->w[level];
    if (h) *h = c->h[level];
    return TSFRAME_OK;
}
int tsframe_level_ptr(void *ctx, int level, int which, const uint8_t **dev) {
    FCtx *c = (FCtx *)ctx; if (!c || !dev || level < 0 || level >= c->n_levels || which < 0 || which > 3) return TSFRAME_ERR_ARG;
    *dev = c->plane[which][level]; return TSFRAME_OK;
}
int tsframe_get_level(void *ctx, int level, int which, uint8_t *out) {
    FCtx *c = (FCtx *)ctx; if (!c || !out || level < 0 || level >= c->n_levels || which < 0 || which > 3) return TSFRAME_ERR_ARG;
    hipSetDevice(c->device);
    CKF(hipMemcpy(out, c->plane[which][level], (size_t)c->w[level]*c->h[level], hipMemcpyDeviceToHost));
    return TSFRAME_OK;
}

int tsframe_pyramid_pts(void *ctx, int mode, const float *xy, int n, const double *box, const double *inv_scale,
                        int32_t *level_off, double *u, double *v, int32_t *idx, double *inten, uint8_t *in) {
    FCtx *c = (FCtx *)ctx;
    if (!c || n < 0 || (n > 0 && !xy) || !inv_scale || !level_off || !u || !v || !idx || !inten || !in || (mode != 0 && mode != 1) || (mode == 0 && !box)) return TSFRAME_ERR_ARG;
    if (c->n_levels == 0) { c->err = "no image set"; return TSFRAME_ERR_STATE; }
    hipSetDevice(c->device);
    const int L = c->n_levels; const size_t cap = (size_t)n*L;
    // grids (host doubles, the reference's expressions: tool.cc:599-616 / :898-907)
    std::vector<GridDev> G(L); size_t max_cell = 1;
    for (int l = 1; l < L; l++) {
        const double s = inv_scale[l];
        const size_t ncell = (size_t)((double)n*s*s + (mode == 0 ? 100 : 500));
        GridDev g; g.mode = mode; g.s = s; g.x0 = 0.0; g.y0 = 0.0;
        if (mode == 0) {
            const double pminx = box[0]*s, pminy = box[1]*s, pmaxx = box[2]*s, pmaxy = box[3]*s;
            const double WH = (pmaxx - pminx)/(pmaxy - pminy);
            g.ch = (int)sqrt((double)ncell/WH); g.cw = (int)sqrt((double)ncell*WH);
            g.fx = (pmaxx - pminx)/(double)g.cw; g.fy = (pmaxy - pminy)/(double)g.ch;
            g.x0 = pminx; g.y0 = pminy;
        } else {
            const double WH = (double)c->w[l]/(double)c->h[l];
            g.ch = (int)sqrt((double)ncell/WH); g.cw = (int)sqrt((double)ncell*WH);
            g.fx = (double)c->w[l]/(double)g.cw; g.fy = (double)c->h[l]/(double)g.ch;
        }
        if (g.cw < 1 || g.ch < 1) { c->err = "degenerate feature grid (empty box?)"; return TSFRAME_ERR_ARG; }
        G[l] = g; max_cell = std::max(max_cell, (size_t)g.cw*g.ch);
    }
    // device scratch: xy | sel | cnt, level_off | u | v | inten | idx | in
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_xy = 0, o_sel = o_xy + al(sizeof(float)*2*(size_t)std::max(n, 1)), o_cnt = o_sel + al(sizeof(int)*max_cell), o_u = o_cnt + al(sizeof(int)*(L + 4)),
                 o_v = o_u + al(8*cap), o_I = o_v + al(8*cap), o_idx = o_I + al(8*cap), o_in = o_idx + al(4*cap), tot = o_in + al(cap);
    int rc = ensure_work(c, tot + 256); if (rc) return rc;
    rc = ensure_host(c, std::max(tot, (size_t)c->w[0]*c->h[0])); if (rc) return rc;
    uint8_t *d = c->d_work;
    if (n > 0) { memcpy(c->h_stage, xy, sizeof(float)*2*(size_t)n); CKF(hipMemcpyAsync(d + o_xy, c->h_stage, sizeof(float)*2*(size_t)n, hipMemcpyHostToDevice, c->stream)); }
    int *cnt = (int *)(d + o_cnt), *loff = cnt + 2;
    for (int l = 0; l < L; l++) {
        if (l > 0) {
            CKF(hipMemsetAsync(d + o_sel, 0xff, sizeof(int)*(size_t)G[l].cw*G[l].ch, c->stream));
            if (n > 0) hipLaunchKernelGGL(k_pts_cells, dim3((n + 255)/256), dim3(256), 0, c->stream, (const float *)(d + o_xy), n,
                                          (const uint8_t *)c->plane[TSFRAME_GRAD][l], c->w[l], c->h[l], G[l], (int *)(d + o_sel));
        }
        hipLaunchKernelGGL(k_pts_emit, dim3(1), dim3(1024), 0, c->stream, (const float *)(d + o_xy), n, l, (const uint8_t *)c->plane[TSFRAME_IMG][l], c->w[l], c->h[l],
                           l > 0 ? G[l] : GridDev{mode, 1, 1, 1.0, 0, 0, 1, 1}, (const int *)(d + o_sel), cnt, loff,
                           (double *)(d + o_u), (double *)(d + o_v), (int *)(d + o_idx), (double *)(d + o_I), d + o_in);
    }
    CKF(hipMemcpyAsync(c->h_stage + o_cnt, d + o_cnt, tot - o_cnt, hipMemcpyDeviceToHost, c->stream));      // one copy for all outputs
    CKF(hipStreamSynchronize(c->stream)); CKF(hipGetLastError());
    const int *hl = (const int *)(c->h_stage + o_cnt) + 2;
    for (int l = 0; l <= L; l++) level_off[l] = hl[l];
    const size_t m = (size_t)level_off[L];
    memcpy(u, c->h_stage + o_u, 8*m); memcpy(v, c->h_stage + o_v, 8*m); memcpy(inten, c->h_stage + o_I, 8*m);
    memcpy(idx, c->h_stage + o_idx, 4*m); memcpy(in, c->h_stage + o_in, m);
    return TSFRAME_OK;
}

int tsframe_neighbours(void *ctx, int level, const double *uv, int n, double mu, double sigma, double *inten8, double *ninten8, uint8_t *in) {
    FCtx *c = (FCtx *)ctx;
    if (!c || n < 0 || (n > 0 && (!uv || !inten8 || !ninten8 || !in)) || level < 0) return TSFRAME_ERR_ARG;
    if (level >= c->n_levels) { c->err = "level not built"; return TSFRAME_ERR_STATE; }
    if (sigma == 0.0) { c->err = "sigma == 0 (tool::CalNormvec returns false)"; return TSFRAME_ERR_ARG; }
    if (n == 0) return TSFRAME_OK;
    hipSetDevice(c->device);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_uv = 0, o_I = al(16*(size_t)n), o_N = o_I + al(64*(size_t)n), o_in = o_N + al(64*(size_t)n), tot = o_in + al(n);
    int rc = ensure_work(c, tot); if (rc) return rc;
    rc = ensure_host(c, std::max(tot, (size_t)c->w[0]*c->h[0])); if (rc) return rc;
    uint8_t *d = c->d_work;
    memcpy(c->h_stage, uv, 16*(size_t)n);
    CKF(hipMemcpyAsync(d + o_uv, c->h_stage, 16*(size_t)n, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(k_neighbours, dim3((8*n + 255)/256), dim3(256), 0, c->stream, (const uint8_t *)c->plane[TSFRAME_IMG][level], c->w[level], c->h[level],
                       (const double *)(d + o_uv), n, mu, sigma, (double *)(d + o_I), (double *)(d + o_N), d + o_in);
    CKF(hipMemcpyAsync(c->h_stage + o_I, d + o_I, tot - o_I, hipMemcpyDeviceToHost, c->stream));
    CKF(hipStreamSynchronize(c->stream)); CKF(hipGetLastError());
    memcpy(inten8, c->h_stage + o_I, 64*(size_t)n); memcpy(ninten8, c->h_stage + o_N, 64*(size_t)n); memcpy(in, c->h_stage + o_in, n);
    return TSFRAME_OK;
}

int tsframe_box_pixels(void *ctx, int level, const double *quad, double mu, double sigma, int cap, int32_t *n_out,
                       int32_t *u, int32_t *v, double *inten, double *ninten) {
    FCtx *c = (FCtx *)ctx;
    if (!c || !quad || !n_out || cap < 0 || (cap > 0 && (!u || !v || !inten || !ninten)) || level < 0) return TSFRAME_ERR_ARG;
    if (level >= c->n_levels) { c->err = "level not built"; return TSFRAME_ERR_STATE; }
    for (int i = 0; i < 8; i++) if (!(fabs(quad[i]) < 1e9)) { c->err = "quad corner not finite"; return TSFRAME_ERR_ARG; }
    hipSetDevice(c->device);
    const int w = c->w[level], h = c->h[level];
    // corners (cv::Point(double, double): truncation) and the clamped bounding box, statement by statement as tool.cc:1269-1298
    BoxDev B;
    int xMin = w + 1, xMax = -1, yMin = h + 1, yMax = -1;
    for (int i = 0; i < 4; i++) {
        const double x = quad[2*i], y = quad[2*i + 1];
        B.xy[2*i] = (int)x; B.xy[2*i + 1] = (int)y;
        if (x > xMax) xMax = (int)ceil(x);
        if (x < xMin) xMin = (int)floor(x);
        if (y > yMax) yMax = (int)ceil(y);
        if (y < yMin) yMin = (int)floor(y);
    }
    if (xMin < 0) xMin = 0;
    if (xMin >= w) xMin = w - 1;
    if (yMin < 0) yMin = 0;
    if (yMin >= h) yMin = h - 1;
    if (xMax >= w) xMax = w - 1;
    if (xMax < 0) xMax = 0;
    if (yMax >= h) yMax = h - 1;
    if (yMax < 0) yMax = 0;
    B.x0 = xMin; B.x1 = xMax; B.y0 = yMin; B.y1 = yMax;
    *n_out = 0;
    if (xMax < xMin || yMax < yMin) return TSFRAME_OK;
    const size_t npx = (size_t)(xMax - xMin + 1)*(size_t)(yMax - yMin + 1);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_cnt = 0, o_mask = 256, o_u = o_mask + al(4*(((size_t)w*h + 31) >> 5)), o_v = o_u + al(4*npx), o_I = o_v + al(4*npx), o_N = o_I + al(8*npx), tot = o_N + al(8*npx);
    int rc = ensure_work(c, tot); if (rc) return rc;
    rc = ensure_host(c, std::max(tot, (size_t)c->w[0]*c->h[0])); if (rc) return rc;
    uint8_t *d = c->d_work;
    hipLaunchKernelGGL(k_box_pixels, dim3(1), dim3(1024), 0, c->stream, (const uint8_t *)c->plane[TSFRAME_IMG][level], w, h, B, mu, sigma, (unsigned *)(d + o_mask),
                       (int *)(d + o_cnt), (int *)(d + o_u), (int *)(d + o_v), (double *)(d + o_I), (double *)(d + o_N));
    CKF(hipMemcpyAsync(c->h_stage, d + o_cnt, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    CKF(hipStreamSynchronize(c->stream)); CKF(hipGetLastError());
    const int n = *(const int *)c->h_stage;
    *n_out = n;
    if (cap == 0 || n == 0) return TSFRAME_OK;                      // cap == 0: count only
    if (n > cap) { c->err = "tsframe_box_pixels: capacity too small (n_out holds the count)"; return TSFRAME_ERR_ARG; }
    CKF(hipMemcpyAsync(c->h_stage + o_u, d + o_u, 4*(size_t)n, hipMemcpyDeviceToHost, c->stream));
    CKF(hipMemcpyAsync(c->h_stage + o_v, d + o_v, 4*(size_t)n, hipMemcpyDeviceToHost, c->stream));
    CKF(hipMemcpyAsync(c->h_stage + o_I, d + o_I, 8*(size_t)n, hipMemcpyDeviceToHost, c->stream));
    CKF(hipMemcpyAsync(c->h_stage + o_N, d + o_N, 8*(size_t)n, hipMemcpyDeviceToHost, c->stream));
    CKF(hipStreamSynchronize(c->stream));
    memcpy(u, c->h_stage + o_u, 4*(size_t)n); memcpy(v, c->h_stage + o_v, 4*(size_t)n);
    memcpy(inten, c->h_stage + o_I, 8*(size_t)n); memcpy(ninten, c->h_stage + o_N, 8*(size_t)n);
    return TSFRAME_OK;
}

}  // extern "C"
