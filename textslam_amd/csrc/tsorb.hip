// libtsorb.so -- MI355X (gfx950) ORB front-end for TextSLAM's ORBextractor hot path.  C ABI in include/tsorb.h.
//
// Batch of n frames, every stage one launch over (frame, level, ...):
//   k_resize              pyramid: 8 levels x1.2, cv::resize INTER_LINEAR fixed point (11-bit weights), 19-px REFLECT_101 frame
//                         fused in (a border pixel is the resize result at the reflected coordinate); level 1 from the input image,
//                         level 0's copy inside its launch                                                 ORBextractor.cc:1118-1143
//   k_fast                one workgroup per 30-px cell: ROI tile in LDS, FAST-9/16 + cornerScore + 3x3 NMS at threshold 20,
//                         per-cell fallback to 7, row-major ordered compaction                               :766-830
//   k_octree              one workgroup per (frame, level): DistributeOctTree -- the full passes and the first pass of phase 2 in
//                         closed form from the keys' coordinates, whatever is left one generation of splits at a time      :540-764
//   k_orient              16 lanes per keypoint: intensity-centroid moments + fastAtan2                         :77-104
//   k_blur                7x7 sigma-2 Gaussian, Q8 fixed point separable, LDS tiled                            :1096-1097
//   k_describe            32 lanes per keypoint: 256 steered BRIEF tests, one byte per lane, written straight into the
//                         level-major output (coordinates scaled back to level 0)                               :108-147, :1106-1112
//   (k_orient and k_blur share a launch: k_orient_blur.)
// Up to P1_MAX_N frames -- the per-frame call of frame.cc:328-331 -- take a launch plan of their own, five launches: k_pyramid_one x 2 (a level's tile formed
// inside one workgroup from the input image / from level 3), k_fast_blur (the blur's tiles behind the detector's workgroups), k_octree, k_orient_describe
// (orientation inside the descriptor's launch, results also stored in the pinned block the download hands out).  Levels the LDS quadtree cannot hold are
// flagged in pinned memory and redone by k_octree_serial behind the run's synchronisation.
// Window search (tracking::SearchFrom3D*): k_mg_cell / k_mg_scan / k_mg_place build a frame's 64 x 48 grid, k_match searches it, a wave per query.
// Integer / fp32 arithmetic is written so that every rounding matches the CPU oracle: no FMA contraction where the
// reference has separate multiplies and adds (__fmul_rn / __fadd_rn / __dmul_rn ...), rintf = cvRound (half to even).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <cfloat>
#include <string>
#include <vector>
#include <type_traits>
#include "../../include/tsorb.h"
#include "../../include/orb_pattern.h"

#define EDGE 19
#define HALF_PATCH 15
#define PATCH_SIZE 31
#define MAXL TSORB_MAX_LEVELS
#define CELL_CAP 1024           // >= keypoints a cell (< 60 x 60 inner pixels) can hold after strict 3x3 NMS (one per 2x2)
#define TILE_MAX 72             // cell ROI is at most (wCell + 6) < 66 pixels wide

struct LevelGeo {
    int w, h, bw, bh;           // level size, bordered size
    int nCols, nRows, wCell, hCell, minB, maxBX, maxBY;
    int cell0;                  // first cell index of this level in the per-frame cell table
    int nfeat, capL;            // quadtree target, slot capacity
    int kp0;                    // first slot of this level in the per-frame selected-keypoint table
    float sf;
    size_t pyr_off, blur_off;   // byte offsets of this level inside one frame's pyramid / blur slab
    int bt0;                    // first blur tile of this level in the per-frame tile table
    int rx_off, ry_off;         // this level's resize tables inside OrbDev::rtab (ints): per bordered column int2, per bordered row int4
};
struct OrbDev {
    int n, nlevels, ini_th, min_th, w, h, stride;
    LevelGeo L[MAXL];
    int cells_per_frame, slots_per_frame, btiles_per_frame, cand_cap, node_cap, pool_cap, cap;
    int fast_nl[2], fast_cells[2], fast_c0[2][MAXL]; signed char fast_lv[2][MAXL];     // k_fast's two launches: [0] the levels whose cell ROIs fit a 40 x 40 tile, [1] the others; c0: a level's first cell inside its launch
    size_t pyr_frame, blur_frame;          // bytes per frame
    const uint8_t *img; uint8_t *pyr, *blur;
    int *rtab;                             // cv::resize coordinate / weight tables of levels 1.. (k_resize_tab, once per geometry)
    double rsx[MAXL], rsy[MAXL];           // cv::resize's inverse scales of level l against level l-1 (1 / (w_l / w_{l-1}): IEEE divisions, formed once on the host)
    uint32_t *cellkp; int *cellcnt;        // [n][cells][CELL_CAP] packed (x | y<<8 | score<<16), [n][cells]
    float *cand;                           // [n][nlevels][cand_cap][3]  x, y, response (relative to minBorder)
    int *nodes, *pool, *snbuf;             // quadtree scratch per (frame, level)
    float *sel;                            // [n][slots][4]  x, y (level coords incl. border offset), response, angle
    int *selcnt;                           // [n][nlevels]
    int *qfallback;                        // [n][nlevels] 1 = the LDS quadtree could not hold this level (serial kernel takes over)
    int *h_fallback;                       // (pinned host memory) set when any level of the run was flagged: the host launches the serial pass only then
    float *selab;                          // [n][slots][2]  cos, sin of the keypoint angle (k_orient; read by k_describe)
    float *out_kp; uint8_t *out_desc; int *out_cnt;
    float *h_kp; uint8_t *h_desc; int *h_cnt;     // the same three blocks in the pinned host buffer tsorb_download hands out: a few frames' results are stored there as well (no copy back)
    int umax[16]; int gk[7];
};
__device__ __constant__ int8_t d_pattern[1024];

__device__ __forceinline__ int reflect101(int x, int n) { if (x < 0) x = -x; if (x >= n) x = 2*(n - 1) - x; return x; }
__device__ __forceinline__ uint8_t *lev_ptr(const OrbDev &D, uint8_t *base, int f, int l) { return base + (size_t)f*D.pyr_frame + D.L[l].pyr_off; }

// ---------------------------------------------------------------- pyramid
// level 0 = the input inside its REFLECT_101 frame.  grid (x chunks of 4 x 128 pixels, groups of L0_ROWS bordered rows, frames); a thread
// moves four neighbouring pixels as one (unaligned) dword where they do not touch the reflected columns.
#ifndef L0_ROWS
#define L0_ROWS 8
#endif
typedef uint32_t __attribute__((aligned(1))) u32_unaligned;
// Workgroup b of a launch runs on XCD b mod 8, each XCD behind an L2 of its own: neighbours in a launch's work order (cells of one row, tiles of one
// level, keypoints of one frame) share cache lines, and in launch order every one of them would sit behind a different L2 -- each line fetched up to eight
// times (measured: k_fast 2.2 x its tiles' bytes, k_blur 2.1 x).  xcd_order gives XCD x the x-th contiguous eighth of the work instead.
#ifndef TSORB_XCD_ORDER
#define TSORB_XCD_ORDER 1
#endif
__device__ __forceinline__ int xcd_order(int b, int n) {
#if TSORB_XCD_ORDER
    const int q = n >> 3, rem = n & 7, x = b & 7;
    return x*q + min(x, rem) + (b >> 3);
#else
    return b;
#endif
}
// four neighbouring pixels (bordered column x ..) of the L0_ROWS bordered rows from y0 of frame f
__device__ __forceinline__ void level0_item(const OrbDev &D, int x, int y0, int f) {
    const LevelGeo &G = D.L[0];
    if (x >= G.bw) return;
    const uint8_t *src = D.img + (size_t)f*D.h*D.stride;
    uint8_t *dst = D.pyr + (size_t)f*D.pyr_frame + G.pyr_off + x;
    const int ny = min(L0_ROWS, G.bh - y0);
    if (x >= EDGE && x + 3 < EDGE + G.w) {
        uint32_t v[L0_ROWS];
#pragma unroll
        for (int r = 0; r < L0_ROWS; r++) v[r] = *(const u32_unaligned *)(src + (size_t)reflect101(y0 + min(r, ny - 1) - EDGE, G.h)*D.stride + (x - EDGE));
#pragma unroll
        for (int r = 0; r < L0_ROWS; r++) { if (r >= ny) break; *(u32_unaligned *)(dst + (size_t)(y0 + r)*G.bw) = v[r]; }
    } else {                                                    // reflected columns: bytes, but all loads of the row group before the first store (source and
        int cx[4]; uint32_t v[L0_ROWS];                         // destination may alias for the compiler: loads after a store wait for it -- eight round trips)
#pragma unroll
        for (int i = 0; i < 4; i++) cx[i] = reflect101(min(x + i, G.bw - 1) - EDGE, G.w);
#pragma unroll
        for (int r = 0; r < L0_ROWS; r++) { const uint8_t *row = src + (size_t)reflect101(y0 + min(r, ny - 1) - EDGE, G.h)*D.stride;
            v[r] = (uint32_t)row[cx[0]] | ((uint32_t)row[cx[1]] << 8) | ((uint32_t)row[cx[2]] << 16) | ((uint32_t)row[cx[3]] << 24); }
#pragma unroll
        for (int r = 0; r < L0_ROWS; r++) { if (r >= ny) break;
            uint8_t *d = dst + (size_t)(y0 + r)*G.bw;
            if (x + 3 < G.bw) *(u32_unaligned *)d = v[r]; else for (int i = 0; x + i < G.bw; i++) d[i] = (uint8_t)(v[r] >> (8*i)); }
    }
}
__global__ __launch_bounds__(128) void k_level0(OrbDev D) { level0_item(D, 4*(blockIdx.x*128 + threadIdx.x), blockIdx.y*L0_ROWS, blockIdx.z); }
// cv::resize(8UC1, INTER_LINEAR) from level l-1 to level l, evaluated at the reflected coordinate for border pixels.
// The source coordinates and the 11-bit weights depend on the geometry only: k_resize_tab fills them once per upload (the per-pixel
// version spent ~150 instructions per pixel on fp64 coordinate arithmetic, conversions and 64-bit index products).
//   column x: { sx | sx1 << 16, a0 | a1 << 16 }     row y: { sy0, sy1, b0, b1 }
// one column / row of cv::resize's tables: dx, dy = destination coordinate inside the level (after the frame's reflection), Sw, Sh = source size
__device__ __forceinline__ int2 resize_xent(int dx, double scale_x, int Sw) {
    float fx = (float)__dsub_rn(__dmul_rn((double)dx + 0.5, scale_x), 0.5);
    int sx = (int)floorf(fx); fx = __fsub_rn(fx, (float)sx);
    if (sx < 0) { fx = 0.f; sx = 0; }
    if (sx >= Sw - 1) { fx = 0.f; sx = Sw - 1; }
    const int a0 = (int)rintf(__fmul_rn(__fsub_rn(1.f, fx), 2048.f)), a1 = (int)rintf(__fmul_rn(fx, 2048.f));
    const int sx1 = sx + 1 < Sw ? sx + 1 : sx;
    return make_int2(sx | (sx1 << 16), a0 | (a1 << 16));
}
__device__ __forceinline__ int4 resize_yent(int dy, double scale_y, int Sh) {
    float fy = (float)__dsub_rn(__dmul_rn((double)dy + 0.5, scale_y), 0.5);
    int sy = (int)floorf(fy); fy = __fsub_rn(fy, (float)sy);
    const int b0 = (int)rintf(__fmul_rn(__fsub_rn(1.f, fy), 2048.f)), b1 = (int)rintf(__fmul_rn(fy, 2048.f));
    return make_int4(min(max(sy, 0), Sh - 1), min(max(sy + 1, 0), Sh - 1), b0, b1);
}
__global__ __launch_bounds__(256) void k_resize_tab(OrbDev D) {
    const int l = blockIdx.x + 1;
    const LevelGeo &G = D.L[l], &S = D.L[l-1];
    const double scale_x = D.rsx[l], scale_y = D.rsy[l];
    int2 *xt = (int2 *)(D.rtab + G.rx_off); int4 *yt = (int4 *)(D.rtab + G.ry_off);
    for (int x = threadIdx.x; x < G.bw; x += 256) xt[x] = resize_xent(reflect101(x - EDGE, G.w), scale_x, S.w);
    for (int y = threadIdx.x; y < G.bh; y += 256) yt[y] = resize_yent(reflect101(y - EDGE, G.h), scale_y, S.h);
}
// grid (x chunks of 4 x RS_T pixels, groups of RS_ROWS bordered rows, frames).  A thread produces FOUR neighbouring pixels of RS_ROWS rows: away from
// the reflected columns their source columns are monotone and span at most 8 bytes, so a source row is one (unaligned) 8-byte load instead of
// eight byte gathers, and the result one dword store instead of four byte stores (the byte version moved 1 byte per lane and instruction: 19 us
// per level for 64 frames).  The column terms stay in registers for the row group; the row terms are uniform.
#ifndef RS_ROWS
#define RS_ROWS 8
#endif
#ifndef RS_T
#define RS_T 64
#endif
typedef uint64_t __attribute__((aligned(1))) u64_unaligned;
// nyb0 > 0 (level 1 only): the launch also carries level 0's copy -- level 1 is formed from the INPUT image (the bytes of level 0's interior), so the copy is no longer a launch
// in front of the chain but the row groups blockIdx.y >= gridDim.y - nyb0 of this one (two bandwidth-bound launches of a batch side by side instead of one after the other)
__global__ __launch_bounds__(RS_T) void k_resize(OrbDev D, int l, int nyb0) {
    const LevelGeo &G = D.L[l], &S = D.L[l-1];
    if (nyb0 > 0 && (int)blockIdx.y >= (int)gridDim.y - nyb0) { level0_item(D, 4*(blockIdx.x*RS_T + threadIdx.x), ((int)blockIdx.y - ((int)gridDim.y - nyb0))*L0_ROWS, blockIdx.z); return; }
    const int x = 4*(blockIdx.x*RS_T + threadIdx.x), y0 = blockIdx.y*RS_ROWS, f = blockIdx.z;
    if (x >= G.bw) return;
    const int2 *xtab = (const int2 *)(D.rtab + G.rx_off);
    const int4 *ytab = (const int4 *)(D.rtab + G.ry_off);
    const uint8_t *src = nyb0 > 0 ? D.img + (size_t)f*D.h*D.stride : D.pyr + (size_t)f*D.pyr_frame + S.pyr_off + (size_t)EDGE*S.bw + EDGE;
    const int spitch = nyb0 > 0 ? D.stride : S.bw;
    uint8_t *dst = D.pyr + (size_t)f*D.pyr_frame + G.pyr_off + x;
    const int ny = min(RS_ROWS, G.bh - y0);
    int sx[4], sx1[4], a0[4], a1[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { const int2 t = xtab[min(x + k, G.bw - 1)]; sx[k] = t.x & 0xffff; sx1[k] = t.x >> 16; a0[k] = t.y & 0xffff; a1[k] = t.y >> 16; }
    const int base = min(min(sx[0], sx[1]), min(sx[2], sx[3])), top = max(max(sx1[0], sx1[1]), max(sx1[2], sx1[3]));
    if (top - base <= 7) {                                      // (also across the reflected columns: the four source columns are neighbours in any order)
        // byte o of the 8 loaded ones: word o >> 2, bits 8 (o & 3) .. -- 32-bit selects and field extracts (variable 64-bit shifts and 32-bit
        // multiplies run at a quarter of the rate: the operands here are below 2^24, the products exact in 24-bit multiplies)
        // (round 6: a column's two source bytes as ONE byte permute of the 8 loaded ones -> (p0 | p1 << 16), its horizontal term as one 16-bit dot product with the
        // table's packed weights (a0 | a1 << 16): 2 + 2 instructions per pixel where selects, shifts, masks and two multiply-adds were 16)
        typedef unsigned short __attribute__((ext_vector_type(2))) us2;
        uint32_t psel[4]; us2 wts[4];
#pragma unroll
        for (int k = 0; k < 4; k++) { const int o0 = sx[k] - base, o1 = sx1[k] - base; psel[k] = (uint32_t)o0 | (0x0cu << 8) | ((uint32_t)o1 << 16) | (0x0cu << 24);
            wts[k] = __builtin_bit_cast(us2, (uint32_t)a0[k] | ((uint32_t)a1[k] << 16)); }
        uint2 q0[RS_ROWS], q1[RS_ROWS]; int4 yt[RS_ROWS];
#pragma unroll
        for (int r = 0; r < RS_ROWS; r++) {                      // all loads of the row group in flight together
            yt[r] = ytab[y0 + min(r, ny - 1)];
            const uint64_t u0 = *(const u64_unaligned *)(src + (size_t)yt[r].x*spitch + base), u1 = *(const u64_unaligned *)(src + (size_t)yt[r].y*spitch + base);
            q0[r] = make_uint2((uint32_t)u0, (uint32_t)(u0 >> 32)); q1[r] = make_uint2((uint32_t)u1, (uint32_t)(u1 >> 32));
        }
#pragma unroll
        for (int r = 0; r < RS_ROWS; r++) {
            if (r >= ny) break;
            uint32_t o = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t S0 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(q0[r].y, q0[r].x, psel[k])), wts[k], 0u, false);
                const uint32_t S1 = __builtin_amdgcn_udot2(__builtin_bit_cast(us2, __builtin_amdgcn_perm(q1[r].y, q1[r].x, psel[k])), wts[k], 0u, false);
                o |= ((((__umul24(yt[r].z, S0 >> 4)) >> 16) + ((__umul24(yt[r].w, S1 >> 4)) >> 16) + 2) >> 2) << (8*k);
            }
            uint8_t *d = dst + (size_t)(y0 + r)*G.bw;
            if (x + 3 < G.bw) *(u32_unaligned *)d = o;
            else for (int k = 0; x + k < G.bw; k++) d[k] = (uint8_t)(o >> (8*k));     // (the last columns of a row)
        }
    } else {
        for (int r = 0; r < ny; r++) {
            const int4 yt = ytab[y0 + r];
            const uint8_t *r0 = src + (size_t)yt.x*spitch, *r1 = src + (size_t)yt.y*spitch;
            for (int k = 0; k < 4 && x + k < G.bw; k++) {
                const int S0 = r0[sx[k]]*a0[k] + r0[sx1[k]]*a1[k], S1 = r1[sx[k]]*a0[k] + r1[sx1[k]]*a1[k];
                dst[(size_t)(y0 + r)*G.bw + k] = (uint8_t)((((yt.z*(S0 >> 4)) >> 16) + ((yt.w*(S1 >> 4)) >> 16) + 2) >> 2);
            }
        }
    }
}

// ---------------------------------------------------------------- the pyramid of a FEW frames in one or two launches
// k_level0 + seven k_resize are eight dependent launches: 52 of the 132 us the per-frame call of the SLAM front-end takes on the device (frame.cc:328-331 extracts
// one frame at a time), each a table round trip, a pixel round trip and a launch for a few hundred KB.  Here a workgroup owns one tile of ONE level l and forms it
// from a BASE level several levels up (the input image, or a level an earlier launch wrote): the base region under the tile -> LDS, then the levels between, of
// exactly the pixels the next level asks for, LDS to LDS, then the tile.  Every intermediate pixel is cv::resize's fixed-point result from the same four source
// pixels with the same weights as k_resize's (the same table entries: resize_xent / resize_yent), so the levels are the same bytes; what is paid is arithmetic --
// a 16 x 16 tile seven levels below its base forms 13.7 k intermediate pixels, four levels below 2.7 k -- on a device that one frame leaves idle.  tsorb_run
// takes this path for up to P1_MAX_N frames (a batch loses: 64 frames would do several times the work of a pipeline that takes 106 us), as two launches:
// levels 0 .. P1_SPLIT from the input image, the rest from level P1_SPLIT (tsorb_debug_pyramid: 0 = always the chain, 1 = always this, 2 = one launch for all levels).
// Needed pixels: the tile's bordered columns reflect to an interior range of level l; an interior range [a, b] of level k reads the columns sx(a) .. sx1(b) of
// level k-1 (the tables are monotone) -- lanes 0..3 walk the four range ends up the levels.  Level 0 (the input inside its reflected frame) is a copy by
// workgroups of its own in the first launch.  The deepest level's tiles come first in a launch: they are the longest.
#define P1_L0_ROWS 16
#ifndef P1_MAX_N
#define P1_MAX_N 5                    // (ms per call, few-frames plan / a launch per stage and level, final tree: 1: 0.063 / 0.087, 2: 0.070 / 0.096, 4: 0.097 / 0.109, 5: 0.109 / 0.112, 6: 0.118 / 0.119, 8: 0.139 / 0.126)
#endif
#ifndef P1_SPLIT
#define P1_SPLIT 3
#endif
struct PyrOne { int base, top, nl, t0[MAXL + 1], ncol[MAXL], tw[MAXL], th[MAXL], per_frame; };     // levels base+1 .. top from level base (+ level 0's copy when base = 0); t0[i]: first workgroup (inside a frame) of level top - i
__device__ __forceinline__ void reflect_range(int lo, int hi, int n, int &rlo, int &rhi) {       // the interior coordinates that lo .. hi (bordered coordinate minus EDGE) reflect to
    const int a = reflect101(lo, n), b = reflect101(hi, n);
    rlo = min(a, b); rhi = max(a, b);
    if (lo <= 0 && hi >= 0) rlo = 0;
    if (lo <= n - 1 && hi >= n - 1) rhi = n - 1;
}
// four neighbouring destination pixels from the LDS region src (pitch, origin ox, oy): xt = the destination columns' entries, ye = the row's
__device__ __forceinline__ uint32_t p1_quad(const uint8_t *src, int pitch, int ox, int oy, const int2 *xt, int x, int n, const int4 ye) {
    const int base0 = (ye.x - oy)*pitch - ox, base1 = (ye.y - oy)*pitch - ox;
    uint32_t o = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int2 xe = xt[min(x + j, n - 1)];
        const int sx = xe.x & 0xffff, sx1 = xe.x >> 16, a0 = xe.y & 0xffff, a1 = xe.y >> 16;
        const int S0 = (int)src[base0 + sx]*a0 + (int)src[base0 + sx1]*a1, S1 = (int)src[base1 + sx]*a0 + (int)src[base1 + sx1]*a1;
        o |= (uint32_t)(((((ye.z*(S0 >> 4)) >> 16) + ((ye.w*(S1 >> 4)) >> 16) + 2) >> 2) & 255) << (8*j);
    }
    return o;
}
// BUF: bytes per LDS region buffer (two: a stage reads one and writes the other), ENT: table entries (columns, rows) of all stages of a tile, T: threads.
// Two instances: <4096, 192, 256> for tiles up to four levels below their base, <16384, 512, 512> for any depth (the host sizes the tiles and picks)
template <int BUF, int ENT, int T>
__global__ __launch_bounds__(T) void k_pyramid_one(OrbDev D, PyrOne Q) {
    __shared__ __attribute__((aligned(16))) uint8_t buf[2][BUF];
    __shared__ int2 s_xt[ENT];
    __shared__ int4 s_yt[ENT];
    __shared__ int s_rng[MAXL][4];                          // level k < l: the interior columns lo, hi and rows lo, hi the tile needs of it
    const int tid = threadIdx.x;
    const int f = blockIdx.x / Q.per_frame; int r = blockIdx.x - f*Q.per_frame, i = 0;
#pragma unroll
    for (int k = 1; k < MAXL; k++) if (k < Q.nl && r >= Q.t0[k]) i = k;
    const int l = Q.top - i, B = Q.base; r -= Q.t0[i];
    if (l == 0) {                                           // level 0: P1_L0_ROWS bordered rows per workgroup
        const int nx4 = (D.L[0].bw + 3) >> 2;
        for (int it = tid; it < nx4*(P1_L0_ROWS/L0_ROWS); it += T) { const int g = it/nx4, y0 = r*P1_L0_ROWS + g*L0_ROWS; if (y0 < D.L[0].bh) level0_item(D, 4*(it - g*nx4), y0, f); }
        return;
    }
    const LevelGeo &G = D.L[l];
    const int ty = r/Q.ncol[l], tx = r - ty*Q.ncol[l];
    const int X0 = tx*Q.tw[l], Y0 = ty*Q.th[l], TW = min(Q.tw[l], G.bw - X0), TH = min(Q.th[l], G.bh - Y0);
    if (tid < 4) {                                          // lane: axis (x, y) and end (lo, hi) of the needed range, level by level
        const int ax = tid >> 1, hi_end = tid & 1;
        int rlo, rhi;
        if (ax == 0) reflect_range(X0 - EDGE, X0 + TW - 1 - EDGE, G.w, rlo, rhi); else reflect_range(Y0 - EDGE, Y0 + TH - 1 - EDGE, G.h, rlo, rhi);
        int v = hi_end ? rhi : rlo;
        for (int k = l; k > B; k--) {
            if (ax == 0) { const int2 e = resize_xent(v, D.rsx[k], D.L[k-1].w); v = hi_end ? (e.x >> 16) : (e.x & 0xffff); }
            else { const int4 e = resize_yent(v, D.rsy[k], D.L[k-1].h); v = hi_end ? e.y : e.x; }
            s_rng[k-1][tid] = v;
        }
    }
    __syncthreads();
    // ---- the base level's region under the tile: requested now, stored behind the tables' arithmetic
    const int ax0 = s_rng[B][0], ay0 = s_rng[B][2], dpr0 = (s_rng[B][1] - ax0 + 4) >> 2, nd0 = dpr0*(s_rng[B][3] - ay0 + 1);
    if (nd0*4 > BUF) return;                                // (never: the host sized the tiles against the buffers)
    uint32_t v[BUF/4/T];
    {   const uint8_t *img = B == 0 ? D.img + (size_t)f*D.h*D.stride : D.pyr + (size_t)f*D.pyr_frame + D.L[B].pyr_off + (size_t)EDGE*D.L[B].bw + EDGE;
        const int stride = B == 0 ? D.stride : D.L[B].bw, w0 = D.L[B].w;
        const float inv = 1.0f/(float)dpr0;
#pragma unroll
        for (int u = 0; u < BUF/4/T; u++) {
            const int d = min(tid + u*T, nd0 - 1), y = (int)(((float)d + 0.5f)*inv), x = ax0 + 4*(d - y*dpr0);
            const uint8_t *row = img + (size_t)(ay0 + y)*stride;
            if (x + 3 < w0) v[u] = *(const u32_unaligned *)(row + x);
            else v[u] = (uint32_t)row[min(x, w0 - 1)] | ((uint32_t)row[min(x + 1, w0 - 1)] << 8) | ((uint32_t)row[min(x + 2, w0 - 1)] << 16) | ((uint32_t)row[min(x + 3, w0 - 1)] << 24);
        }
    }
    // ---- the table entries of every stage: a thread finds the stage of its entry (levels B+1 .. l-1: the needed interior columns / rows; level l: the tile's bordered ones)
    for (int e0 = tid; e0 < 2*ENT; e0 += T) {
        int e = e0, xo = 0, yo = 0;
        for (int k = B + 1; k <= l; k++) {
            const int lo = k < l ? s_rng[k][0] : X0, n = k < l ? s_rng[k][1] - lo + 1 : TW, loy = k < l ? s_rng[k][2] : Y0, ny = k < l ? s_rng[k][3] - loy + 1 : TH;
            if (e < n) { if (xo + e < ENT) s_xt[xo + e] = resize_xent(k < l ? lo + e : reflect101(lo + e - EDGE, G.w), D.rsx[k], D.L[k-1].w); break; }
            e -= n;
            if (e < ny) { if (yo + e < ENT) s_yt[yo + e] = resize_yent(k < l ? loy + e : reflect101(loy + e - EDGE, G.h), D.rsy[k], D.L[k-1].h); break; }
            e -= ny; xo += n; yo += ny;
        }
    }
#pragma unroll
    for (int u = 0; u < BUF/4/T; u++) if (tid + u*T < nd0) ((uint32_t *)buf[0])[tid + u*T] = v[u];
    __syncthreads();
    // ---- the levels between, of the needed pixels, LDS to LDS, four neighbouring pixels per thread
    int cur = 0, ox = ax0, oy = ay0, pitch = 4*dpr0, xo = 0, yo = 0;
    for (int k = B + 1; k < l; k++) {
        const int lo = s_rng[k][0], n = s_rng[k][1] - lo + 1, loy = s_rng[k][2], ny = s_rng[k][3] - loy + 1, qpr = (n + 3) >> 2, nq = qpr*ny;
        if (4*nq > BUF || xo + n > ENT || yo + ny > ENT) return;        // (never)
        const uint8_t *src = buf[cur]; uint32_t *dst = (uint32_t *)buf[cur ^ 1];
        const float inv = 1.0f/(float)qpr;
        for (int q = tid; q < nq; q += T) { const int y = (int)(((float)q + 0.5f)*inv), x = 4*(q - y*qpr); dst[q] = p1_quad(src, pitch, ox, oy, s_xt + xo, x, n, s_yt[yo + y]); }
        __syncthreads();
        cur ^= 1; ox = lo; oy = loy; pitch = 4*qpr; xo += n; yo += ny;
    }
    // ---- the tile itself
    {   const int qpr = (TW + 3) >> 2, nq = qpr*TH;
        if (xo + TW > ENT || yo + TH > ENT) return;                      // (never)
        const uint8_t *src = buf[cur];
        uint8_t *out = D.pyr + (size_t)f*D.pyr_frame + G.pyr_off + (size_t)Y0*G.bw + X0;
        const float inv = 1.0f/(float)qpr;
        for (int q = tid; q < nq; q += T) { const int y = (int)(((float)q + 0.5f)*inv), x = 4*(q - y*qpr);
            const uint32_t o = p1_quad(src, pitch, ox, oy, s_xt + xo, x, TW, s_yt[yo + y]);
            uint8_t *d = out + (size_t)y*G.bw + x;
            if (x + 3 < TW) *(u32_unaligned *)d = o; else for (int j = 0; x + j < TW; j++) d[j] = (uint8_t)(o >> (8*j)); }
    }
}

// ---------------------------------------------------------------- FAST per cell
// quick reject: 9 contiguous ring pixels always contain at least two of the four compass points
__device__ __forceinline__ bool fast_maybe(const uint8_t *p, int stride, int threshold) {
    // "at least two of the four darker than v - t" is "the second smallest of the four is", likewise the second largest against v + t: eight min / max
    // on the pixel values instead of eight compares and their sums on the differences
    const int v = p[0], a = p[3*stride], b = p[-3*stride], c = p[3], d = p[-3];
    const int mn1 = min(a, b), mx1 = max(a, b), mn2 = min(c, d), mx2 = max(c, d);
    const int second_smallest = min(max(mn1, mn2), min(mx1, mx2)), second_largest = max(min(mx1, mx2), max(mn1, mn2));
    return second_smallest < v - threshold || second_largest > v + threshold;
}
__device__ __forceinline__ int fast_score(const uint8_t *p, int stride, int threshold) {     // 0 = not a corner
    const int cx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
    const int cy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };
    const int v = p[0];
    int q[18];
#pragma unroll
    for (int k = 0; k < 16; k++) q[k] = p[cy[k]*stride + cx[k]];
    q[16] = q[0]; q[17] = q[1];
    // cornerScore (OpenCV fast_score.cpp, called at ORBextractor.cc:808-812 through cv::FAST): with d_k = v - q_k, a0 = max(threshold, the largest over
    // the 16 arcs of 9 of the arc's smallest d), b0 = min(-a0, the smallest over the arcs of the arc's largest d), score = -b0 - 1.  No separate arc
    // test: a pixel is a corner at `threshold` (an arc with every |d| above it) exactly when that score reaches the threshold; otherwise it is
    // threshold - 1.  (The bit-mask run-length test this replaces cost a third of a survivor's instructions and spared the rest only to a wave without
    // a single corner.)  The arcs' extrema as a sliding window over the ring -- threes, then three threes -- on the pixel values themselves
    // (min d = v - max q): the numbers of OpenCV's ladders, whose early-outs only skip arcs that cannot change the result, in under half the
    // instructions and without a branch.
    int lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) { lo3[k] = min(min(q[k], q[k+1]), q[k+2]); hi3[k] = max(max(q[k], q[k+1]), q[k+2]); }
    int arc_hi = 255, arc_lo = 0;               // the smallest of the arcs' maxima, the largest of their minima
#pragma unroll
    for (int k = 0; k < 16; k++) {
        arc_hi = min(arc_hi, max(max(hi3[k], hi3[(k+3) & 15]), hi3[(k+6) & 15]));
        arc_lo = max(arc_lo, min(min(lo3[k], lo3[(k+3) & 15]), lo3[(k+6) & 15]));
    }
    const int a0 = max(threshold, v - arc_hi), b0 = min(-a0, v - arc_lo);
    const int r = -b0 - 1;
    return r >= threshold ? r : 0;
}
#ifndef FAST_TU
#define FAST_TU 2                       // tile dwords per thread in flight (measured on a batch of 64, four waves per cell: 1 -> 0.494, 2 -> 0.490, 3 -> 0.494, 4 -> 0.497, 6 -> 0.505 ms: more requests per thread cost occupancy)
#endif
// S: row stride of the LDS tile (>= the widest cell ROI of the geometry, a multiple of 4), NC: quick-reject survivors listed per cell (a typical cell has
// ~1000 inner pixels, 10-30 % survive; the rest are scored in place), NK: corners a cell can keep (one per 2 x 2 inner pixels after the strict 3 x 3
// suppression), T: threads.  Two instances: <40, 1024, 320, 128> for geometries whose cells are 30 - 34 px (every level of a 752 x 480 frame; 6.4 KB of
// LDS: sixteen workgroups = all 32 wave slots of a compute unit), <72, 2048, 1024, 256> for anything up to the 59-px cell the grid rule allows.
// Measured on a batch of 64 (rocprofv3, phases by elimination, round 5): four waves per cell on the 72 x 72 tile 133 us; two waves per cell on the 40 x 40
// tile 101 us + 9 us for the levels with larger cells = an empty launch of the 51 k workgroups 26 (42 with four waves each), tiles 18, quick reject 22,
// scores 25, suppression + ordering + output 10.  One wave per cell, two cells per workgroup with both tiles requested together, the blur on a second
// stream beside the detector or the quadtree (events, or hipExtAnyOrderLaunch, which gfx9 ignores): all measured, none faster (docs/ledger_r05.md).
template <int S, int NC, int NK, int T, int GRP>
__device__ __forceinline__ void fast_body(const OrbDev &D, const int block, const int nblocks) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[S*S];
    __shared__ __attribute__((aligned(16))) uint8_t score[S*S];      // cornerScore <= 255
    __shared__ int s_ncand, s_nkeep;
    __shared__ unsigned short s_cand[NC];
    __shared__ unsigned int s_keep[NK];
    const int bid = xcd_order(block, nblocks);
    const int f = bid / D.fast_cells[GRP], tid = threadIdx.x;
    int cc = bid % D.fast_cells[GRP], c0 = 0, lv = D.fast_lv[GRP][0];      // the cell inside this launch's levels (every offset into the kernel arguments static: one batch of scalar loads)
#pragma unroll
    for (int k = 1; k < MAXL; k++) { const bool in = k < D.fast_nl[GRP] && cc >= D.fast_c0[GRP][k]; c0 = in ? D.fast_c0[GRP][k] : c0; lv = in ? D.fast_lv[GRP][k] : lv; }
    cc -= c0;
    const LevelGeo &G = D.L[lv];
    const int cidx = G.cell0 + cc, i = cc / G.nCols, j = cc % G.nCols;
    int *cnt_out = D.cellcnt + (size_t)f*D.cells_per_frame + cidx;
    // cell ROI, ORBextractor.cc:790-806 (float arithmetic on small integers is exact)
    const int iniY = G.minB + i*G.hCell, iniX = G.minB + j*G.wCell;
    int maxY = iniY + G.hCell + 6, maxX = iniX + G.wCell + 6;
    if (iniY >= G.maxBY - 3 || iniX >= G.maxBX - 6) { if (tid == 0) *cnt_out = 0; return; }
    if (maxY > G.maxBY) maxY = G.maxBY;
    if (maxX > G.maxBX) maxX = G.maxBX;
    const int rw = maxX - iniX, rh = maxY - iniY;
    if (rw < 7 || rh < 7) { if (tid == 0) *cnt_out = 0; return; }
    const uint8_t *src = D.pyr + (size_t)f*D.pyr_frame + G.pyr_off + (size_t)(EDGE + iniY)*G.bw + EDGE + iniX;
    // the tile as (unaligned) dwords, two per thread in flight -- a byte per thread and iteration was five dependent round trips --, the score
    // map cleared on the way (a row may be read up to 3 bytes past the ROI: still inside the level's frame)
    const int iw = rw - 6, ih = rh - 6;             // inner pixels (3-px margin)
    {
        const int rw4 = (rw + 3) >> 2, n4 = rw4*rh; const float inv_rw4 = 1.0f/(float)rw4;
        for (int k0 = tid; k0 < n4; k0 += FAST_TU*T) {       // FAST_TU dwords per thread in flight
            uint32_t v[FAST_TU]; int at[FAST_TU];
#pragma unroll
            for (int u = 0; u < FAST_TU; u++) { const int k = min(k0 + T*u, n4 - 1), y = (int)(((float)k + 0.5f)*inv_rw4), x = 4*(k - y*rw4);
                at[u] = y*S + x; v[u] = *(const u32_unaligned *)(src + (size_t)y*G.bw + x); }
#pragma unroll
            for (int u = 0; u < FAST_TU; u++) if (k0 + T*u < n4) { *(uint32_t *)&tile[at[u]] = v[u]; *(uint32_t *)&score[at[u]] = 0u; }
        }
        if (tid == 0) { s_ncand = 0; s_nkeep = 0; }
    }
    __syncthreads();
    uint32_t *out = D.cellkp + ((size_t)f*D.cells_per_frame + cidx)*CELL_CAP;
    // ONE scoring pass at the fallback threshold: cornerScore is the largest threshold the pixel still passes at (minus one), so
    // "corner at iniTh" is score >= iniTh, and a corner at iniTh beats every neighbour that is not one (their score is below
    // iniTh) -- the 3x3 maximum test on the full score map selects exactly what cv::FAST(iniTh) + NMS selects.
    const float inv_iw = 1.0f/(float)iw;
    const int npx = iw*ih;
    // the cornerScore is ~100 instructions and a wave runs them for all 64 lanes if one needs them: first a 4-pixel quick reject over all
    // inner pixels, the survivors' tile positions compacted into a list, then the score on the list with every lane busy (the order of the
    // list is irrelevant: scores go to the score map by position)
    for (int k = tid; k < npx; k += T) { const int yy = (int)(((float)k + 0.5f)*inv_iw), pos = (3 + yy)*S + 3 + k - yy*iw;
        if (fast_maybe(tile + pos, S, D.min_th)) { const int i = atomicAdd(&s_ncand, 1);
            if (i < NC) s_cand[i] = (unsigned short)pos; else score[pos] = (uint8_t)fast_score(tile + pos, S, D.min_th); } }   // (list full: scored in place; one atomic per wave through a ballot measured slower: 197 vs 185 us)
    __syncthreads();
    // (the survivors in two steps -- arc test on all, the score on the compacted corners only -- was slower: two more barriers, and the 16 ring
    // reads are what a survivor costs)
    // (a thread scores at most NC / T = 8 survivors: which of them came out as corners stays in a bit mask, the suppression pass below looks at
    // those positions only instead of at every pixel of the cell again -- 22 us of the 187 were that second sweep)
    const int ncand = s_ncand;
    unsigned mine = 0;
    for (int k = tid, m = 0; k < min(ncand, NC); k += T, m++) { const int pos = s_cand[k], sc = fast_score(tile + pos, S, D.min_th); score[pos] = (uint8_t)sc; mine |= (sc > 0 ? 1u : 0u) << m; }
    __syncthreads();
    // 3x3 non-maximum suppression.  The survivors are few (a handful per cell): appended in any order, then put into the reference's
    // row-major order by a rank sort on the pixel index -- two barriers per pass instead of a ballot scan over every 256-pixel chunk
    int n = 0;
    for (int pass = 0; pass < 2; pass++) {
        const int th = pass == 0 ? D.ini_th : 1;
        if (ncand <= NC) {                   // uniform: every scored pixel is on the list
            for (unsigned left = mine; left; left &= left - 1) {
                const int pos = s_cand[tid + T*(__ffs(left) - 1)], y = pos/S, x = pos - y*S;
                const uint8_t *q = score + pos; const int sc = q[0];
                if (sc >= th && sc > q[-S-1] && sc > q[-S] && sc > q[-S+1] && sc > q[-1] && sc > q[1] &&
                    sc > q[S-1] && sc > q[S] && sc > q[S+1]) { const int i = atomicAdd(&s_nkeep, 1); if (i < NK) s_keep[i] = (unsigned int)((y - 3)*iw + x - 3) | ((unsigned int)sc << 16); }
            }
        } else
        for (int k = tid; k < npx; k += T) {      // (list full: some pixels were scored in place)
            const int yy = (int)(((float)k + 0.5f)*inv_iw), y = 3 + yy, x = 3 + k - yy*iw;
            const uint8_t *q = score + y*S + x; const int sc = q[0];
            if (sc >= th && sc > q[-S-1] && sc > q[-S] && sc > q[-S+1] && sc > q[-1] && sc > q[1] &&
                sc > q[S-1] && sc > q[S] && sc > q[S+1]) { const int i = atomicAdd(&s_nkeep, 1); if (i < NK) s_keep[i] = (unsigned int)k | ((unsigned int)sc << 16); }
        }
        __syncthreads();
        n = min(s_nkeep, NK);
        if (n > 0) break;                           // uniform (nothing at iniTh: the cell is searched again at the fallback threshold)
    }
    for (int a = tid; a < n; a += T) {
        const unsigned int e = s_keep[a]; const int ka = (int)(e & 0xffffu); int rank = 0;
        for (int b2 = 0; b2 < n; b2++) rank += (int)(s_keep[b2] & 0xffffu) < ka;
        const int yy = (int)(((float)ka + 0.5f)*inv_iw), y = 3 + yy, x = 3 + ka - yy*iw;
        out[rank] = (uint32_t)x | ((uint32_t)y << 8) | ((e >> 16) << 16);
    }
    if (tid == 0) *cnt_out = n;
}
template <int S, int NC, int NK, int T, int GRP>
__global__ __launch_bounds__(T) void k_fast(OrbDev D) { fast_body<S, NC, NK, T, GRP>(D, blockIdx.x, gridDim.x); }

// ---------------------------------------------------------------- quadtree (DistributeOctTree), one workgroup per (frame, level)
struct QNode { int ulx, uly, urx, ury, blx, bly, brx, bry, key0, nk, nomore, prev, next; };
struct QList { QNode *n; int cnt, head, tail, size; int *pool; int pool_top; const float *kp; };
__device__ int q_new(QList &L, int cap) { int i = L.cnt < cap ? L.cnt++ : cap - 1; QNode &q = L.n[i]; q.nomore = 0; q.prev = q.next = -1; q.nk = 0; q.key0 = 0; return i; }
__device__ void q_push_back(QList &L, int i) { L.n[i].prev = L.tail; L.n[i].next = -1; if (L.tail >= 0) L.n[L.tail].next = i; else L.head = i; L.tail = i; L.size++; }
__device__ void q_push_front(QList &L, int i) { L.n[i].next = L.head; L.n[i].prev = -1; if (L.head >= 0) L.n[L.head].prev = i; else L.tail = i; L.head = i; L.size++; }
__device__ int q_erase(QList &L, int i) { int p = L.n[i].prev, nx = L.n[i].next; if (p >= 0) L.n[p].next = nx; else L.head = nx; if (nx >= 0) L.n[nx].prev = p; else L.tail = p; L.size--; return nx; }
__device__ void q_divide(QList &L, int src, int c[4], int node_cap, int pool_cap) {
    for (int k = 0; k < 4; k++) c[k] = q_new(L, node_cap);
    const QNode s = L.n[src];
    const int halfX = (int)ceilf((float)(s.urx - s.ulx)/2), halfY = (int)ceilf((float)(s.bry - s.uly)/2);
    QNode &n1 = L.n[c[0]], &n2 = L.n[c[1]], &n3 = L.n[c[2]], &n4 = L.n[c[3]];
    n1.ulx = s.ulx; n1.uly = s.uly; n1.urx = s.ulx + halfX; n1.ury = s.uly; n1.blx = s.ulx; n1.bly = s.uly + halfY; n1.brx = s.ulx + halfX; n1.bry = s.uly + halfY;
    n2.ulx = n1.urx; n2.uly = n1.ury; n2.urx = s.urx; n2.ury = s.ury; n2.blx = n1.brx; n2.bly = n1.bry; n2.brx = s.urx; n2.bry = s.uly + halfY;
    n3.ulx = n1.blx; n3.uly = n1.bly; n3.urx = n1.brx; n3.ury = n1.bry; n3.blx = s.blx; n3.bly = s.bly; n3.brx = n1.brx; n3.bry = s.bly;
    n4.ulx = n3.urx; n4.uly = n3.ury; n4.urx = n2.brx; n4.ury = n2.bry; n4.blx = n3.brx; n4.bly = n3.bry; n4.brx = s.brx; n4.bry = s.bry;
    // count, carve four order-preserving sub-arrays out of the pool, fill
    int cn[4] = {0, 0, 0, 0};
    const float ux = (float)n1.urx, by = (float)n1.bry;
    for (int i = 0; i < s.nk; i++) { const float *kp = L.kp + 3*L.pool[s.key0 + i];
        int q = (kp[0] < ux) ? ((kp[1] < by) ? 0 : 2) : ((kp[1] < by) ? 1 : 3); cn[q]++; }
    int base = L.pool_top; if (base + s.nk > pool_cap) base = pool_cap - s.nk;     // (capacity guard; sized so that it never triggers)
    L.pool_top = base + s.nk;
    int st[4]; st[0] = base; st[1] = st[0] + cn[0]; st[2] = st[1] + cn[1]; st[3] = st[2] + cn[2];
    for (int k = 0; k < 4; k++) { L.n[c[k]].key0 = st[k]; L.n[c[k]].nk = cn[k]; if (cn[k] == 1) L.n[c[k]].nomore = 1; }
    int w[4] = { st[0], st[1], st[2], st[3] };
    for (int i = 0; i < s.nk; i++) { int key = L.pool[s.key0 + i]; const float *kp = L.kp + 3*key;
        int q = (kp[0] < ux) ? ((kp[1] < by) ? 0 : 2) : ((kp[1] < by) ? 1 : 3); L.pool[w[q]++] = key; }
}
// (one thread: the (frame, level) problems the LDS version cannot hold -- none on camera images)
__device__ __noinline__ void octree_serial(const OrbDev &D, int f, int l) {
    const LevelGeo &G = D.L[l];
    float *cand = D.cand + ((size_t)f*D.nlevels + l)*D.cand_cap*3;
    int s_nc;
    // 1. gather the cells of this level in the reference's order (rows of cells, then columns; row-major inside a cell)
    {
        int nc = 0;
        const int *cnt = D.cellcnt + (size_t)f*D.cells_per_frame + G.cell0;
        const uint32_t *ck = D.cellkp + ((size_t)f*D.cells_per_frame + G.cell0)*CELL_CAP;
        for (int c = 0; c < G.nCols*G.nRows; c++) {
            const int i = c / G.nCols, j = c % G.nCols;
            for (int q = 0; q < cnt[c] && nc < D.cand_cap; q++) {
                uint32_t p = ck[(size_t)c*CELL_CAP + q];
                cand[3*nc] = (float)((int)(p & 255u) + j*G.wCell); cand[3*nc+1] = (float)((int)((p >> 8) & 255u) + i*G.hCell); cand[3*nc+2] = (float)(p >> 16);
                nc++;
            }
        }
        s_nc = nc;
    }
    const int nk = s_nc;
    int *selcnt = D.selcnt + (size_t)f*D.nlevels + l;
    float *sel = D.sel + ((size_t)f*D.slots_per_frame + G.kp0)*4;
    if (nk == 0) { *selcnt = 0; return; }
    QList L;
    L.n = (QNode *)(D.nodes + ((size_t)f*D.nlevels + l)*(size_t)D.node_cap*(sizeof(QNode)/sizeof(int)));
    L.pool = D.pool + ((size_t)f*D.nlevels + l)*(size_t)D.pool_cap;
    int *snbuf = D.snbuf + ((size_t)f*D.nlevels + l)*(size_t)(4*D.node_cap);
    L.cnt = 0; L.head = L.tail = -1; L.size = 0; L.pool_top = 0; L.kp = cand;
    const int minX = G.minB, maxX = G.maxBX, minY = G.minB, maxY = G.maxBY, N = G.nfeat;
    const int nIni = (int)roundf((float)(maxX - minX)/(float)(maxY - minY));
    const float hX = __fdiv_rn((float)(maxX - minX), (float)nIni);
    // initial nodes: count, carve, fill (order preserving)
    for (int i = 0; i < nIni; i++) {
        int q = q_new(L, D.node_cap); QNode &n = L.n[q];
        n.ulx = (int)__fmul_rn(hX, (float)i); n.uly = 0; n.urx = (int)__fmul_rn(hX, (float)(i + 1)); n.ury = 0;
        n.blx = n.ulx; n.bly = maxY - minY; n.brx = n.urx; n.bry = maxY - minY;
        q_push_back(L, q);
    }
    for (int i = 0; i < nk; i++) { int q = (int)__fdiv_rn(cand[3*i], hX); if (q >= nIni) q = nIni - 1; L.n[q].nk++; }
    { int top = 0; for (int i = 0; i < nIni; i++) { L.n[i].key0 = top; top += L.n[i].nk; L.n[i].nk = 0; } L.pool_top = top; }
    for (int i = 0; i < nk; i++) { int q = (int)__fdiv_rn(cand[3*i], hX); if (q >= nIni) q = nIni - 1; L.pool[L.n[q].key0 + L.n[q].nk++] = i; }
    for (int it = L.head; it >= 0; ) { QNode &n = L.n[it]; if (n.nk == 1) { n.nomore = 1; it = n.next; } else if (n.nk == 0) it = q_erase(L, it); else it = n.next; }
    bool finish = false;
    int *vs = snbuf, *vprev = snbuf + 2*D.node_cap; int nvs = 0;       // pairs (size, node)
    while (!finish) {
        int prevSize = L.size, nToExpand = 0; nvs = 0;
        for (int it = L.head; it >= 0; ) {
            if (L.n[it].nomore) { it = L.n[it].next; continue; }
            int c[4]; q_divide(L, it, c, D.node_cap, D.pool_cap);
            for (int k = 0; k < 4; k++) if (L.n[c[k]].nk > 0) {
                q_push_front(L, c[k]);
                if (L.n[c[k]].nk > 1) { nToExpand++; if (nvs < D.node_cap) { vs[2*nvs] = L.n[c[k]].nk; vs[2*nvs+1] = c[k]; nvs++; } }
            }
            it = q_erase(L, it);
        }
        if (L.size >= N || L.size == prevSize) finish = true;
        else if (L.size + nToExpand*3 > N) {
            while (!finish) {
                prevSize = L.size;
                const int np = nvs;
                for (int k = 0; k < 2*np; k++) vprev[k] = vs[k];
                nvs = 0;
                // ascending sort by (size, creation order) -- insertion sort, the list is short
                for (int a = 1; a < np; a++) { int s0 = vprev[2*a], n0 = vprev[2*a+1]; int b = a - 1;
                    while (b >= 0 && (vprev[2*b] > s0 || (vprev[2*b] == s0 && vprev[2*b+1] > n0))) { vprev[2*b+2] = vprev[2*b]; vprev[2*b+3] = vprev[2*b+1]; b--; }
                    vprev[2*b+2] = s0; vprev[2*b+3] = n0; }
                for (int jq = np - 1; jq >= 0; jq--) {
                    int c[4]; q_divide(L, vprev[2*jq+1], c, D.node_cap, D.pool_cap);
                    for (int k = 0; k < 4; k++) if (L.n[c[k]].nk > 0) {
                        q_push_front(L, c[k]);
                        if (L.n[c[k]].nk > 1 && nvs < D.node_cap) { vs[2*nvs] = L.n[c[k]].nk; vs[2*nvs+1] = c[k]; nvs++; }
                    }
                    q_erase(L, vprev[2*jq+1]);
                    if (L.size >= N) break;
                }
                if (L.size >= N || L.size == prevSize) finish = true;
            }
        }
    }
    int ns = 0;
    for (int it = L.head; it >= 0 && ns < G.capL; it = L.n[it].next) {
        const QNode &n = L.n[it]; int best = L.pool[n.key0]; float mr = cand[3*best+2];
        for (int k = 1; k < n.nk; k++) { int key = L.pool[n.key0 + k]; if (cand[3*key+2] > mr) { best = key; mr = cand[3*key+2]; } }
        sel[4*ns] = cand[3*best] + (float)minX; sel[4*ns+1] = cand[3*best+1] + (float)minY; sel[4*ns+2] = mr; sel[4*ns+3] = 0.f;
        ns++;
    }
    *selcnt = ns;
}
__global__ __launch_bounds__(64) void k_octree_serial(OrbDev D) { if (threadIdx.x == 0 && D.qfallback[blockIdx.x]) octree_serial(D, blockIdx.x / D.nlevels, blockIdx.x % D.nlevels); }


// ---------------------------------------------------------------- quadtree, wave-cooperative, everything in LDS.
// One wave per (frame, level).  Candidates (x, y, response), the per-node key lists and the node pool live in LDS; the list
// surgery is executed redundantly by all 64 lanes (uniform control flow, lane 0 writes), the 4-way stable partition of a
// node's keys and the final arg-max are spread over the lanes.  Levels that do not fit fall back to k_octree_serial.
#ifndef QL_CAND
#define QL_CAND 4096
#endif
#define QL_NODES 1024
struct QN { short x0, y0, x1, y1; unsigned short key0, nk, id, pad; };     // 16 bytes: box, key range, creation number
#ifndef QT
#define QT 512                  // (one frame: 256 threads 0.0938 ms, 512 0.0903, 1024 0.0901; 64 frames: 0.424, 0.424, 0.458)
#endif
#define QL_CELLS (4*QT)
// exclusive block scan of 4 values per thread (thread t owns elements 4t .. 4t+3): ex[j] = sum of everything before element 4t+j
__device__ __forceinline__ void qscan4(const int v[4], int *s_w, int tid, int ex[4], int &total) {
    const int lane = tid & 63, wv = tid >> 6, t = v[0] + v[1] + v[2] + v[3];
    int incl = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
    if (lane == 63) s_w[wv] = incl;
    __syncthreads();
    int base = 0; total = 0;
#pragma unroll
    for (int q = 0; q < QT/64; q++) { const int c = s_w[q]; if (q < wv) base += c; total += c; }
    __syncthreads();
    ex[0] = base + incl - t; ex[1] = ex[0] + v[0]; ex[2] = ex[1] + v[1]; ex[3] = ex[2] + v[2];
}
// exclusive block scan of two values per thread (one entry each)
__device__ __forceinline__ void qscan1x2(const int a, const int b, int *s_w2, int tid, int &ea, int &eb) {
    const int lane = tid & 63, wv = tid >> 6;
    int ia = a, ib = b;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int ua = __shfl_up(ia, o, 64), ub = __shfl_up(ib, o, 64); if (lane >= o) { ia += ua; ib += ub; } }
    if (lane == 63) { s_w2[2*wv] = ia; s_w2[2*wv + 1] = ib; }
    __syncthreads();
    int ba = 0, bb = 0;
#pragma unroll
    for (int q = 0; q < QT/64; q++) if (q < wv) { ba += s_w2[2*q]; bb += s_w2[2*q + 1]; }
    __syncthreads();
    ea = ba + ia - a; eb = bb + ib - b;
}
#ifndef Q_FF
#define Q_FF (QT >= 512)        // the generations of full passes in one step (below); needs a thread per cell of the four levels (340)
#endif
#define FF_D 4                  // generations it can take at once (4^4 = 256 cells at the deepest)
__device__ __forceinline__ int ff_base(int g) { return ((1 << (2*g)) - 1)/3; }      // cells of the depths before g, the root included: 0, 1, 5, 21, 85, 341
// position digits <-> path digits of a generation-g node: digit i (1 = the root's child) runs backwards where g - i is even (an involution)
__device__ __forceinline__ int ff_flip(int code, int g) { int out = 0;
#pragma unroll
    for (int i = 1; i <= FF_D; i++) if (i <= g) { const int d = (code >> (2*(g - i))) & 3; out |= (((g - i) & 1) ? d : 3 - d) << (2*(g - i)); }
    return out; }
// DistributeOctTree (ORBextractor.cc:537-753) for one (frame, level).  The loop below takes one generation of splits at a time; since the last session of round 6 the
// generations of full passes and the first pass of phase 2 are taken in ONE step before it (further down: "the generations of FULL passes in one step"), after which
// the loop has nothing left to do on camera images (it still runs for two initial nodes, for more than four full generations, for a second pass of phase 2).  The reference walks a
// std::list and splits node after node; what a pass does to the list is nevertheless a function of the pass's processing order only:
//   * phase 1 (a full pass): every expandable node (more than one key) of the list, in list order, is split; children are pushed to the
//     FRONT (so they are not visited in the same pass) and the parent is erased;
//   * phase 2 (once size + 3 nToExpand > N): the expandable nodes sorted by (size, creation order), largest first, are split until the list
//     holds N nodes -- the cut is a prefix sum over "non-empty children - 1", found before any key moves (ties in the final arg-max go to the
//     node's first key in the reference = its smallest candidate number here, whatever order the node's keys are in).
// New list = reverse(non-empty children in processing order) ++ (old list without the split nodes); creation numbers = 4 per split in
// processing order (they break the ties of phase 2's sort, as the node addresses do in the reference).  So a pass is: processing order
// (compaction / rank sort) -> child counts (one wave per node) -> scan + cut -> stable 4-way partition of the keys (one wave per node)
// -> child records + new list (scans).  Nodes live in a pool indexed by the list: the first non-empty child takes its parent's slot, so
// the pool never holds more than the list.  Levels that do not fit (candidates, nodes) fall back to k_octree_serial.
// A problem that does not fit is flagged for k_octree_serial and, in pinned host memory, for the host: tsorb_run looks at that word after its synchronisation and only then
// launches the serial pass and the orientation / descriptor kernels a second time (never on camera images; until round 6 k_octree_serial was launched every time: 4.8 us
// to find nothing).  Solved on the spot by thread 0 instead, the serial code's registers and scratch cost this kernel more than that launch -- 64 frames: 0.449 against
// 0.424 ms; one frame: 54.4 us against 44.8 + 4.8.
#define Q_FALLBACK() do { if (tid == 0) { D.qfallback[blockIdx.x] = 1; *selcnt = 0; *D.h_fallback = 1; } return; } while (0)     // (the level counts as empty until the serial pass has run)
#ifdef Q_STAMPS       // (tools/diag/octree_stamps.sh) thread 0's cycles by phase into D.snbuf: gather, first nodes, per pass: order / counts / cut / partition / lists, arg-max; 8 .. 13: the phases of the full generations' step
#define QS(i) do { if (tid == 0) { const long long t1_ = clock64(); q_acc[i] += (int)(t1_ - q_t0); q_t0 = t1_; } } while (0)
#else
#define QS(i) do { } while (0)
#endif
__global__ __launch_bounds__(QT) void k_octree(OrbDev D) {
#ifdef Q_STAMPS
    int q_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, q_pass = 0; long long q_t0 = clock64();
#endif
    const int f = blockIdx.x / D.nlevels, l = blockIdx.x % D.nlevels, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const LevelGeo &G = D.L[l];
    // 16-bit candidates / keys (pixel coordinates and FAST scores are small integers): 78 KB of LDS, two workgroups per CU
    __shared__ unsigned short cx[QL_CAND], cy[QL_CAND], cr[QL_CAND];
    __shared__ __attribute__((aligned(4))) unsigned short keys[QL_CAND], tmpk[QL_CAND];
    __shared__ QN nd[QL_NODES];
    __shared__ unsigned short lst[2][QL_NODES], proc[QL_NODES];
    __shared__ __attribute__((aligned(4))) unsigned short cnt4[4*QL_NODES];
    __shared__ unsigned char fsplit[QL_NODES];
    __shared__ unsigned short coff[QL_CELLS], mexs[QL_NODES];
    __shared__ int s_w[QT/64], s_cut, s_front, s_nexp, s_seg[QT/64][4];
    int *selcnt = D.selcnt + (size_t)f*D.nlevels + l;
    float *sel = D.sel + ((size_t)f*D.slots_per_frame + G.kp0)*4;
    // ---- gather the cells (reference order) into LDS: cell offsets by a block scan, then one thread per ENTRY (the cell by binary search)
    // so that all loads of a round are in flight together (one thread per cell walked its ~20 entries one load after the other)
    const int ncell = G.nCols*G.nRows;
    const int *cnt = D.cellcnt + (size_t)f*D.cells_per_frame + G.cell0;
    const uint32_t *ck = D.cellkp + ((size_t)f*D.cells_per_frame + G.cell0)*CELL_CAP;
    if (ncell > QL_CELLS) Q_FALLBACK();
    int nk = 0;
    {   // thread t owns cells 4t .. 4t+3 (QL_CELLS = 4 QT): one scan, the four counts in flight together
        int v[4], ex[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = 4*tid + j < ncell ? cnt[4*tid + j] : 0;
        qscan4(v, s_w, tid, ex, nk);
#pragma unroll
        for (int j = 0; j < 4; j++) if (4*tid + j < ncell) coff[4*tid + j] = (unsigned short)min(ex[j], 65535);
    }
    if (tid == 0) D.qfallback[blockIdx.x] = 0;
    if (nk == 0) { if (tid == 0) *selcnt = 0; return; }
    if (nk > QL_CAND) Q_FALLBACK();
    __syncthreads();
    const float inv_ncols = 1.0f/(float)G.nCols;
    // U entries per thread and round: their loads are in flight together (eight for a level that needs them; a camera frame's level has two per thread: the eight searches of a
    // round were mostly clamped duplicates)
    auto gather = [&](auto utag) {
        constexpr int U = decltype(utag)::value;
        for (int k0 = tid; k0 < nk; k0 += U*QT) {
            int cell[U]; uint32_t p[U];
#pragma unroll
            for (int u = 0; u < U; u++) cell[u] = 0;
            // the last cell whose offset is <= k (an empty cell shares its successor's offset): a fixed number of steps, the searches' LDS reads of a step in
            // flight together (a while loop per entry was eight times ten dependent reads: 8 k of the 18 k cycles this gather took)
            for (int step = QL_CELLS/2; step > 0; step >>= 1) {
                if (step >= ncell) continue;
#pragma unroll
                for (int u = 0; u < U; u++) { const int k = min(k0 + u*QT, nk - 1), c = cell[u] + step; if (c < ncell && coff[min(c, QL_CELLS - 1)] <= k) cell[u] = c; }
            }
#pragma unroll
            for (int u = 0; u < U; u++) { const int k = min(k0 + u*QT, nk - 1); p[u] = ck[(size_t)cell[u]*CELL_CAP + (k - coff[cell[u]])]; }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int k = k0 + u*QT; if (k >= nk) break;
                const int i = (int)(((float)cell[u] + 0.5f)*inv_ncols), j = cell[u] - i*G.nCols;          // (cell / nCols: exact for these sizes, a tenth of the division's instructions)
                cx[k] = (unsigned short)((int)(p[u] & 255u) + j*G.wCell); cy[k] = (unsigned short)((int)((p[u] >> 8) & 255u) + i*G.hCell); cr[k] = (unsigned short)(p[u] >> 16);
            }
        }
    };
    if (nk <= 2*QT) gather(std::integral_constant<int, 2>()); else gather(std::integral_constant<int, 8>());
    QS(0);
    const int minX = G.minB, maxX = G.maxBX, minY = G.minB, maxY = G.maxBY, N = G.nfeat;
    const int nIni = (int)roundf((float)(maxX - minX)/(float)(maxY - minY));
    const float hX = (float)(maxX - minX)/(float)nIni;
    if (nIni < 1 || nIni > 16) Q_FALLBACK();
    // ---- initial nodes (ORBextractor.cc:544-573): empty ones are dropped, the list keeps the rest in order
    for (int k = tid; k < nk; k += QT) keys[k] = (unsigned short)k;
    __syncthreads();
    if (tid == 0) {
        int top = 0, sz = 0;
        for (int i = 0; i < nIni; i++) {
            int c2 = nk;
            if (nIni > 1) {          // (never for 4:3 images) stable bucket by x / hX, serial
                c2 = 0;
                for (int k = 0; k < nk; k++) { int q = (int)((float)cx[k]/hX); if (q >= nIni) q = nIni - 1; if (q == i) tmpk[top + c2++] = (unsigned short)k; }
            }
            if (c2 > 0) { QN q; q.x0 = (short)(int)(hX*(float)i); q.y0 = 0; q.x1 = (short)(int)(hX*(float)(i + 1)); q.y1 = (short)(maxY - minY);
                q.key0 = (unsigned short)top; q.nk = (unsigned short)c2; q.id = (unsigned short)i; q.pad = 0; nd[sz] = q; lst[0][sz] = (unsigned short)sz; sz++; }
            top += c2;
        }
        s_front = sz;
    }
    __syncthreads();
    if (nIni > 1) { for (int k = tid; k < nk; k += QT) keys[k] = tmpk[k]; __syncthreads(); }
    int size = s_front, next_id = nIni, cur = 0;
    bool phase2 = false, overflow = false, ff_done = false;
    __syncthreads();
    QS(1);
    // ---- the generations of FULL passes in one step.  While every expandable node is split (until size + 3 nToExpand > N) the tree after g passes is a function of the
    // keys' coordinates alone: a key's path is four box halvings in registers, a cell's key count a histogram, and the list after pass g is
    //   [generation g] ++ [the one-key nodes of generation g-1] ++ .. ++ [.. of generation 1],
    // a generation ordered by its path digits with digit i running backwards where g - i is even (children go to the FRONT in order 0..3, parents are visited front to
    // back: every pass reverses the order of the one before and appends a backwards digit).  Creation numbers (phase 2's tie-break): 4 per split node in the order of
    // the pass = the parent's rank among the expandable nodes of its generation.  Keys: grouped by path = a counting sort by the deepest cell.  One step instead of three
    // passes of five barrier-separated phases each (~20 k cycles a pass, whatever the number of keys); the loop below goes on from there (phase 2, or further full passes).
    if (Q_FF && nIni == 1 && nk > 1) {
        int *hist = (int *)cnt4, *pref = hist + 340;          // keys per cell, depth g = 1..4 at ff_base(g) - 1; exclusive prefix over the deepest cells (then the scatter's cursors)
        __shared__ int s_ff[FF_D + 1][2], s_w2[2*(QT/64)];
        unsigned short *rk = mexs;                            // per scan entry: expandable nodes before it
        for (int i = tid; i < 340 + 256; i += QT) hist[i] = 0;
        if (tid < 2*(FF_D + 1)) s_ff[tid >> 1][tid & 1] = 0;
        const QN root = nd[0];
        __syncthreads();
        for (int k = tid; k < nk; k += QT) {
            int x0 = root.x0, y0 = root.y0, x1 = root.x1, y1 = root.y1, code = 0; const int X = cx[k], Y = cy[k];
#pragma unroll
            for (int g = 1; g <= FF_D; g++) {
                const int hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1, zx = X < x0 + hx ? 0 : 1, zy = Y < y0 + hy ? 0 : 1;
                if (zx) x0 += hx; else x1 = x0 + hx;
                if (zy) y0 += hy; else y1 = y0 + hy;
                code = 4*code + zx + 2*zy;
                atomicAdd(&hist[ff_base(g) - 1 + code], 1);
            }
        }
        __syncthreads();
        QS(8);
        // a cell is a node of its generation when it holds keys and every cell above it holds more than one
        auto cnt_of = [&](int g, int c) { return g == 0 ? nk : hist[ff_base(g) - 1 + c]; };
        auto exists = [&](int g, int c) { if (cnt_of(g, c) == 0) return false; for (int a = g - 1; a >= 1; a--) if (hist[ff_base(a) - 1 + (c >> (2*(g - a)))] <= 1) return false; return true; };
        if (tid >= 1 && tid < 341) { int g = 1; while (tid >= ff_base(g + 1)) g++; const int c = tid - ff_base(g);
            if (exists(g, c)) { atomicAdd(&s_ff[g][0], 1); if (hist[ff_base(g) - 1 + c] > 1) atomicAdd(&s_ff[g][1], 1); } }
        __syncthreads();
        QS(9);
        int G = FF_D, sizeG = 1, nid = nIni; bool stop = false;
        {   int sz = 1, nexp_prev = 1;                        // generation 0: the root, expandable
            for (int g = 1; g <= FF_D; g++) {
                const int szg = sz - nexp_prev + s_ff[g][0], nexp = s_ff[g][1];
                nid += 4*nexp_prev;
                G = g; sizeG = szg;
                if (szg >= N || szg == sz) { stop = true; break; }
                if (szg + 3*nexp > N) { phase2 = true; break; }
                sz = szg; nexp_prev = nexp;
            }
        }
        // scan entries: [generation G by position] [generation G-1] .. [generation 1]; a thread per entry
        int g = 0, pos = 0, c = 0; bool fin = false, expd = false;
        {   int e = tid;
            for (int gg = G; gg >= 1; gg--) { const int n = 1 << (2*gg); if (e < n) { g = gg; pos = e; break; } e -= n; }
            if (g) { c = ff_flip(pos, g); if (exists(g, c)) { const int cn = hist[ff_base(g) - 1 + c]; fin = g == G || cn == 1; expd = cn > 1; } }
        }
        // the first pass of phase 2 as well where the cells one level further down are counted (G < FF_D): the expandable nodes of generation G, largest first
        // (ties: the later creation number), are split until the list holds N nodes -- a rank sort over at most 64 keys, the non-empty children of each from the
        // histogram, the cut from their running sum; the list is [children of the split nodes, last split first, quadrant 3 first] ++ [the others in order]
        const bool p2 = phase2 && !stop && G < FF_D;
        const int GK = p2 ? G + 1 : G;                        // depth of the cells the keys are grouped by
        int eA, eP;
        qscan1x2((fin ? 1 : 0) | (expd ? 1 << 16 : 0), tid < (1 << (2*GK)) ? hist[ff_base(GK) - 1 + tid] : 0, s_w2, tid, eA, eP);
        if (tid < (1 << (2*GK))) pref[tid] = eP;
        if (tid < 340) rk[tid] = (unsigned short)(eA >> 16);
        if (tid == 0) s_cut = s_ff[G][1];
        __syncthreads();
        QS(10);
        int P = eA & 0xffff, myid = 0, x0 = root.x0, y0 = root.y0, x1 = root.x1, y1 = root.y1;
        if (fin) {                                            // box and creation number of the node
            for (int i = 1; i <= g; i++) { const int z = (c >> (2*(g - i))) & 3, hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1;
                if (z & 1) x0 += hx; else x1 = x0 + hx;
                if (z & 2) y0 += hy; else y1 = y0 + hy; }
            // creation number: 4 x the parent's rank among the expandable nodes of ITS generation (in that generation's order) + the quadrant, behind the numbers of the passes before
            int id0 = nIni, blk = 0, prank = 0;
            for (int gg = 1; gg < g; gg++) id0 += 4*(gg == 1 ? 1 : s_ff[gg - 1][1]);
            if (g > 1) { for (int gg = G; gg > g - 1; gg--) blk += 1 << (2*gg); prank = rk[blk + ff_flip(c >> 2, g - 1)] - rk[blk]; }
            myid = id0 + 4*prank + (c & 3);
        }
        bool split = false; int r2 = 0, mex2 = 0, m2 = 0, front = 0, newSize = sizeG;
        if (p2) {
            unsigned int *ck = (unsigned int *)tmpk;          // (at most 64 keys)
            const int np2 = s_ff[G][1]; const bool cand = g == G && expd;
            const unsigned int mykey = cand ? ((unsigned int)hist[ff_base(G) - 1 + c] << 16) | (unsigned int)myid : 0u;
            if (cand) { ck[eA >> 16] = mykey;                 // (generation G is the first block of the scan: the count of expandable entries before this one is its index)
#pragma unroll
                for (int z = 0; z < 4; z++) m2 += hist[ff_base(G + 1) - 1 + 4*c + z] > 0; }
            __syncthreads();
            if (cand) {
#pragma unroll 8
                for (int b2 = 0; b2 < np2; b2++) r2 += ck[b2] > mykey;
                proc[r2] = (unsigned short)m2; }
            __syncthreads();
            if (cand) {
#pragma unroll 8
                for (int b2 = 0; b2 < r2; b2++) mex2 += proc[b2];
                if (sizeG + mex2 + m2 - (r2 + 1) >= N && !(r2 > 0 && sizeG + mex2 - r2 >= N)) s_cut = r2 + 1; }      // (the running size never falls: ONE rank is the first to reach N -- an atomicMin by every rank behind it was up to 60 atomics on one word)
            __syncthreads();
            const int S = s_cut;
            split = cand && r2 < S;
            if (cand && r2 == S - 1) s_front = mex2 + m2;
            int eS, e0;
            qscan1x2(split ? 1 : 0, 0, s_w2, tid, eS, e0);    // (its barriers publish s_front)
            front = s_front; newSize = front + sizeG - S; nid += 4*S;
            P = front + P - eS;                               // a kept node: behind the new front, in the old order
            if (split) {                                      // the children's records: quadrant z of the r-th split node at front - 1 - (children before it)
                const int hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1; int jj = 0;
#pragma unroll
                for (int z = 0; z < 4; z++) { const int cn = hist[ff_base(G + 1) - 1 + 4*c + z];
                    if (cn > 0) { QN q; q.x0 = (short)((z & 1) ? x0 + hx : x0); q.x1 = (short)((z & 1) ? x1 : x0 + hx); q.y0 = (short)((z & 2) ? y0 + hy : y0); q.y1 = (short)((z & 2) ? y1 : y0 + hy);
                        q.key0 = (unsigned short)pref[4*c + z]; q.nk = (unsigned short)cn; q.id = (unsigned short)(nid - 4*S + 4*r2 + z); q.pad = 0;
                        const int at = front - 1 - (mex2 + jj); nd[at] = q; lst[0][at] = (unsigned short)at; jj++; } }
            }
        }
        if (fin && !split) {                                  // the node's record at its list position (pool slot = list position)
            QN q; q.x0 = (short)x0; q.y0 = (short)y0; q.x1 = (short)x1; q.y1 = (short)y1;
            q.key0 = (unsigned short)pref[c << (2*(GK - g))]; q.nk = (unsigned short)hist[ff_base(g) - 1 + c]; q.id = (unsigned short)myid; q.pad = 0;
            nd[P] = q; lst[0][P] = (unsigned short)P;
        }
        __syncthreads();
        QS(11);
        // keys: into their deepest cell's range, in any order (cursor = the prefix): the order of a node's keys matters to nothing but the arg-max's ties, and those go by
        // candidate number (the stable order cost 6 k cycles: a rank inside the cell per key)
        auto code_of = [&](int k) { int x0 = root.x0, y0 = root.y0, x1 = root.x1, y1 = root.y1, code = 0; const int X = cx[k], Y = cy[k];
            for (int gg = 1; gg <= GK; gg++) { const int hx = (x1 - x0 + 1) >> 1, hy = (y1 - y0 + 1) >> 1, zx = X < x0 + hx ? 0 : 1, zy = Y < y0 + hy ? 0 : 1;
                if (zx) x0 += hx; else x1 = x0 + hx;
                if (zy) y0 += hy; else y1 = y0 + hy;
                code = 4*code + zx + 2*zy; }
            return code; };
        for (int k = tid; k < nk; k += QT) keys[atomicAdd(&pref[code_of(k)], 1)] = (unsigned short)k;
        __syncthreads();
        QS(12);
        if (p2) { if (newSize >= N || newSize == sizeG) stop = true; sizeG = newSize; }
        size = sizeG; next_id = nid; cur = 0; ff_done = stop;
    }
    QS(2);
    if (!ff_done)
    for (;;) {
        const int prevSize = size;
        unsigned short *L = lst[cur], *Ln = lst[cur ^ 1];
        // ---- 1. processing order: list positions of the expandable nodes
        int np;
        {
            int v[4], ex[4];
#pragma unroll
            for (int j = 0; j < 4; j++) { const int p = 4*tid + j; v[j] = (p < size && nd[L[p]].nk > 1) ? 1 : 0; }
            qscan4(v, s_w, tid, ex, np);
            if (!phase2) {
#pragma unroll
                for (int j = 0; j < 4; j++) if (v[j]) proc[ex[j]] = (unsigned short)(4*tid + j);
            } else {         // candidates (position | (size, creation number) key), then rank sort, largest first
                unsigned int *ckey = (unsigned int *)tmpk;                   // np <= QL_NODES u32 keys fit the 8 KB of tmpk
#pragma unroll
                for (int j = 0; j < 4; j++) if (v[j]) { const int p = 4*tid + j; const QN q = nd[L[p]]; cnt4[ex[j]] = (unsigned short)p; ckey[ex[j]] = ((unsigned int)q.nk << 16) | q.id; }
            }
            __syncthreads();
            if (phase2) {
                const unsigned int *ckey = (const unsigned int *)tmpk;
                for (int a0 = tid; a0 < np; a0 += QT) { const unsigned int ka = ckey[a0]; int rank = 0;
                    for (int b2 = 0; b2 < np; b2++) rank += ckey[b2] > ka;
                    proc[rank] = cnt4[a0]; }
                __syncthreads();
            }
        }
        QS(2);
        if (np == 0) break;                                                   // nothing left to split: size == prevSize
        // ---- 2. child counts, one wave per node
        // (a pass with ONE node -- the first generation: all candidates of the level -- is shared by the waves: wave w takes the w-th
        // contiguous segment of the keys; counts per segment, so that the partition below stays stable)
        // small nodes (later passes: tens of nodes with a handful of keys) share a wave: lpn lanes per node, 64 / lpn nodes per wave at a time
        const bool one = np == 1;
        int lpn = 64; { const int avg = nk/np; while (lpn > 8 && lpn >= 2*avg) lpn >>= 1; }
        const int gpw = 64/lpn, grp = lane/lpn, sub = lane - grp*lpn;
        const unsigned long long gmask = (lpn == 64 ? ~0ull : ((1ull << lpn) - 1ull)) << (grp*lpn);
        for (int r0 = one ? 0 : wv*gpw; r0 < np; r0 += (QT/64)*gpw) {
            const int r = r0 + grp; const bool act = r < np;
            QN q = nd[L[proc[act ? r : r0]]]; if (!act) q.nk = 0;
            const int ux = q.x0 + ((q.x1 - q.x0 + 1) >> 1), by = q.y0 + ((q.y1 - q.y0 + 1) >> 1);     // ceil(half extent), ORBextractor.cc:496-497
            const int seg = one ? (((q.nk + QT/64 - 1)/(QT/64) + 63) & ~63) : q.nk, k_lo = one ? min(wv*seg, (int)q.nk) : 0, k_hi = one ? min(k_lo + seg, (int)q.nk) : q.nk;
            int c0 = 0, c1 = 0, c2 = 0;
            for (int b2 = k_lo; __any(b2 < k_hi); b2 += lpn) {            // (four rounds' reads in flight together: measured slower -- most nodes take one round, and a lone wave pays for the other three's ballots)
                const int k = b2 + sub; int z = -1;
                if (k < k_hi) { const int key = keys[q.key0 + k]; z = (cx[key] < ux) ? ((cy[key] < by) ? 0 : 2) : ((cy[key] < by) ? 1 : 3); }
                c0 += __popcll(__ballot(z == 0) & gmask); c1 += __popcll(__ballot(z == 1) & gmask); c2 += __popcll(__ballot(z == 2) & gmask);
            }
            if (one) { if (lane == 0) { s_seg[wv][0] = c0; s_seg[wv][1] = c1; s_seg[wv][2] = c2; s_seg[wv][3] = (k_hi - k_lo) - c0 - c1 - c2; } }
            else if (sub == 0 && act) { cnt4[4*r] = (unsigned short)c0; cnt4[4*r+1] = (unsigned short)c1; cnt4[4*r+2] = (unsigned short)c2; cnt4[4*r+3] = (unsigned short)(q.nk - c0 - c1 - c2); }
        }
        if (one) { __syncthreads(); if (tid < 4) { int t = 0; for (int w = 0; w < QT/64; w++) t += s_seg[w][tid]; cnt4[tid] = (unsigned short)t; } }
        if (tid == 0) { s_cut = np; s_nexp = 0; }
        for (int p = tid; p < size; p += QT) fsplit[p] = 0;
        __syncthreads();
        QS(3);
        // ---- 3. cut (phase 2) and offsets: thread t owns processing ranks 4t .. 4t+3
        int m[4], mex[4], mtot;
#pragma unroll
        for (int j = 0; j < 4; j++) { const int r = 4*tid + j; m[j] = 0;
            if (r < np) m[j] = (cnt4[4*r] > 0) + (cnt4[4*r+1] > 0) + (cnt4[4*r+2] > 0) + (cnt4[4*r+3] > 0); }
        qscan4(m, s_w, tid, mex, mtot);
        if (phase2) {
#pragma unroll
            for (int j = 0; j < 4; j++) { const int r = 4*tid + j; if (r < np && size + mex[j] + m[j] - (r + 1) >= N) atomicMin(&s_cut, r + 1); }
            __syncthreads();
        }
        const int S = s_cut;
#pragma unroll
        for (int j = 0; j < 4; j++) { const int r = 4*tid + j; if (r < S) { fsplit[proc[r]] = 1; mexs[r] = (unsigned short)mex[j]; } if (r == S - 1) s_front = mex[j] + m[j]; }
        __syncthreads();
        const int front = s_front, newSize = front + size - S;
        if (newSize > QL_NODES || next_id + 4*S > 65000) { overflow = true; break; }
        QS(4);
        // ---- 4. stable 4-way partition of the split nodes' keys, one wave per node (tmpk at the node's own key positions)
        for (int r0 = one ? 0 : wv*gpw; r0 < S; r0 += (QT/64)*gpw) {
            const int r = r0 + grp; const bool act = r < S;
            QN q = nd[L[proc[act ? r : r0]]]; if (!act) q.nk = 0;
            const int ux = q.x0 + ((q.x1 - q.x0 + 1) >> 1), by = q.y0 + ((q.y1 - q.y0 + 1) >> 1);
            const int seg = one ? (((q.nk + QT/64 - 1)/(QT/64) + 63) & ~63) : q.nk, k_lo = one ? min(wv*seg, (int)q.nk) : 0, k_hi = one ? min(k_lo + seg, (int)q.nk) : q.nk;
            for (int b2 = k_lo + sub; b2 < k_hi; b2 += lpn) tmpk[q.key0 + b2] = keys[q.key0 + b2];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // one wave: LDS accesses are ordered
            const int rc = act ? r : r0;
            int run0 = q.key0, run1 = run0 + cnt4[4*rc], run2 = run1 + cnt4[4*rc+1], run3 = run2 + cnt4[4*rc+2];
            if (one) for (int w = 0; w < wv; w++) { run0 += s_seg[w][0]; run1 += s_seg[w][1]; run2 += s_seg[w][2]; run3 += s_seg[w][3]; }
            if (one) __syncthreads();                    // every wave has copied its segment before any wave scatters into it
            for (int b2 = k_lo; __any(b2 < k_hi); b2 += lpn) {
                const int k = b2 + sub; int z = -1, key = 0;
                if (k < k_hi) { key = tmpk[q.key0 + k]; z = (cx[key] < ux) ? ((cy[key] < by) ? 0 : 2) : ((cy[key] < by) ? 1 : 3); }
                const unsigned long long lt = (1ull << lane) - 1ull;
                const unsigned long long m0 = __ballot(z == 0) & gmask, m1 = __ballot(z == 1) & gmask, m2 = __ballot(z == 2) & gmask, m3 = __ballot(z == 3) & gmask;
                if (z == 0) keys[run0 + __popcll(m0 & lt)] = (unsigned short)key;
                if (z == 1) keys[run1 + __popcll(m1 & lt)] = (unsigned short)key;
                if (z == 2) keys[run2 + __popcll(m2 & lt)] = (unsigned short)key;
                if (z == 3) keys[run3 + __popcll(m3 & lt)] = (unsigned short)key;
                run0 += __popcll(m0); run1 += __popcll(m1); run2 += __popcll(m2); run3 += __popcll(m3);
            }
        }
        QS(5);
        // ---- 5. the kept part of the list (order preserved) behind the new front
        {
            int v[4], ex[4], tot;
#pragma unroll
            for (int j = 0; j < 4; j++) { const int p = 4*tid + j; v[j] = (p < size && !fsplit[p]) ? 1 : 0; }
            qscan4(v, s_w, tid, ex, tot);                   // (its barriers also order step 4's reads of nd before step 6's writes)
#pragma unroll
            for (int j = 0; j < 4; j++) if (v[j]) Ln[front + ex[j]] = L[4*tid + j];
        }
        // ---- 6. child records: the first non-empty child takes the parent's pool slot, the others fresh ones (the pool holds `size` nodes)
        {
            int nexp = 0;
            for (int r = tid; r < S; r += QT) {              // (one split node per thread: S rarely exceeds QT)
                const int slot = L[proc[r]], mexr = mexs[r];
                const QN q = nd[slot];
                const int hx = (q.x1 - q.x0 + 1) >> 1, hy = (q.y1 - q.y0 + 1) >> 1;
                int st = q.key0, jj = 0;
                const int fresh = size + (mexr - r);        // slots taken by the extra children of the nodes before r
#pragma unroll
                for (int z = 0; z < 4; z++) {
                    const int cn = cnt4[4*r + z];
                    if (cn > 0) {
                        QN c;
                        c.x0 = (short)((z & 1) ? q.x0 + hx : q.x0); c.x1 = (short)((z & 1) ? q.x1 : q.x0 + hx);
                        c.y0 = (short)((z & 2) ? q.y0 + hy : q.y0); c.y1 = (short)((z & 2) ? q.y1 : q.y0 + hy);
                        c.key0 = (unsigned short)st; c.nk = (unsigned short)cn; c.id = (unsigned short)(next_id + 4*r + z); c.pad = 0;
                        const int sl = jj == 0 ? slot : fresh + jj - 1;
                        nd[sl] = c;
                        Ln[front - 1 - (mexr + jj)] = (unsigned short)sl;
                        nexp += cn > 1; jj++;
                    }
                    st += cn;
                }
            }
            if (nexp) atomicAdd(&s_nexp, nexp);
        }
        __syncthreads();
        size = newSize; next_id += 4*S; cur ^= 1;
        const int nToExpand = s_nexp;
        __syncthreads();                                     // (s_nexp / s_cut are reset by the next pass)
        QS(6);
#ifdef Q_STAMPS
        q_pass++;
#endif
        if (size >= N || size == prevSize) break;
        if (!phase2 && size + nToExpand*3 > N) phase2 = true;
    }
    if (overflow) Q_FALLBACK();
    // ---- best response per node, in list order
    const unsigned short *L = lst[cur];
    const int ns = min(size, G.capL);
    for (int q = tid; q < ns; q += QT) {
        const QN n = nd[L[q]];
        int best = keys[n.key0]; int mr = cr[best];
        for (int k = 1; k < n.nk; k++) { const int key = keys[n.key0 + k]; const int r = cr[key]; if (r > mr || (r == mr && key < best)) { best = key; mr = r; } }      // (ties: the reference keeps the FIRST key of the node = the smallest candidate number, whatever order the node's keys are in here)
        sel[4*q] = (float)cx[best] + (float)minX; sel[4*q+1] = (float)cy[best] + (float)minY; sel[4*q+2] = (float)mr; sel[4*q+3] = 0.f;
    }
    if (tid == 0) *selcnt = ns;
    QS(7);
#ifdef Q_STAMPS
    if (tid == 0) { int *o = D.snbuf + 32*blockIdx.x; for (int i = 0; i < 8; i++) o[i] = q_acc[i]; o[8] = q_pass; o[9] = nk; o[10] = size; for (int i = 8; i < 16; i++) o[16 + i - 8] = q_acc[i]; }
#endif
}

// ---------------------------------------------------------------- orientation: 16 lanes per keypoint
__device__ __forceinline__ float fast_atan2f_dev(float y, float x) {     // cv::fastAtan2, degrees
    const float p1 = 0.9997878412794807f*(float)(180/3.14159265358979323846), p3 = -0.3258083974640975f*(float)(180/3.14159265358979323846);
    const float p5 = 0.1555786518463281f*(float)(180/3.14159265358979323846), p7 = -0.04432655554792128f*(float)(180/3.14159265358979323846);
    float ax = fabsf(x), ay = fabsf(y), a, c, c2;
    if (ax >= ay) { c = __fdiv_rn(ay, __fadd_rn(ax, (float)DBL_EPSILON)); c2 = __fmul_rn(c, c);
        a = __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c); }
    else { c = __fdiv_rn(ax, __fadd_rn(ay, (float)DBL_EPSILON)); c2 = __fmul_rn(c, c);
        a = __fsub_rn(90.f, __fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(__fadd_rn(__fmul_rn(p7, c2), p5), c2), p3), c2), p1), c)); }
    if (x < 0) a = __fsub_rn(180.f, a);
    if (y < 0) a = __fsub_rn(360.f, a);
    return a;
}
// rows +v / -v of a keypoint's circular patch (lane v of its sixteen): the lane's share of the moments m10, m01 (ORBextractor.cc:77-104)
__device__ __forceinline__ void orient_rows(const OrbDev &D, const LevelGeo &G, int f, float kx, float ky, int v, int &m10, int &m01) {
    // as 8 + 8 unaligned dwords (columns -16 .. 15; a keypoint is >= 16 px inside the level, the level sits in a 19-px frame), the columns beyond
    // umax[v] masked out: 16 loads in flight instead of up to 62 dependent byte loads
    const uint8_t *c = D.pyr + (size_t)f*D.pyr_frame + G.pyr_off + (size_t)(EDGE + (int)rintf(ky))*G.bw + EDGE + (int)rintf(kx);
    const int d = v == 0 ? HALF_PATCH : D.umax[v];
    const uint8_t *rp = c + (ptrdiff_t)v*G.bw - 16, *rm = c - (ptrdiff_t)v*G.bw - 16;
    uint32_t wp[8], wm[8];
#pragma unroll
    for (int w = 0; w < 8; w++) { wp[w] = *(const u32_unaligned *)(rp + 4*w); wm[w] = *(const u32_unaligned *)(rm + 4*w); }
    int vs = 0;
#pragma unroll
    for (int w = 0; w < 8; w++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int u = 4*w + b - 16;
            if (u >= -d && u <= d) { const int vp = (wp[w] >> (8*b)) & 255, vm = (wm[w] >> (8*b)) & 255; vs += vp - vm; m10 += u*(v == 0 ? vp : vp + vm); }
        }
    m01 = v*vs;
}
__device__ __forceinline__ void orient_body(const OrbDev &D, const int block, const int nblocks) {
    const int bid = xcd_order(block, nblocks);
    const int g = (bid*256 + threadIdx.x) >> 4, v = threadIdx.x & 15;
    const int per = D.slots_per_frame, f = g / per, slot = g % per;
    if (f >= D.n) return;
    int l = 0; while (l + 1 < D.nlevels && slot >= D.L[l+1].kp0) l++;
    const LevelGeo &G = D.L[l];
    const bool valid = (slot - G.kp0) < D.selcnt[(size_t)f*D.nlevels + l];
    float *s = D.sel + ((size_t)f*per + slot)*4;
    int m10 = 0, m01 = 0;
    if (valid) orient_rows(D, G, f, s[0], s[1], v, m10, m01);
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { m10 += __shfl_xor(m10, o, 16); m01 += __shfl_xor(m01, o, 16); }
    // the steering terms of rBRIEF, (float)cos((double)angle), (float)sin((double)angle): fp64 library code of a few hundred instructions that the
    // descriptor kernel ran on every lane of every keypoint (32 lanes each: 21 of its 45 us) -- here once per keypoint, sixteen keypoints per wave pass
    __shared__ float ang[16];
    const int grp = threadIdx.x >> 4;
    float angle = 0.f;
    if (valid && v == 0) { angle = fast_atan2f_dev((float)m01, (float)m10); s[3] = angle; }
    if (v == 0) ang[grp] = valid ? angle : -1.f;
    __syncthreads();
    if (threadIdx.x < 16) {
        const float an = ang[threadIdx.x];
        if (an >= 0.f) {
            const int g2 = (bid*256 >> 4) + threadIdx.x;
            const float factorPI = (float)(3.14159265358979323846/180.f);
            const float rad = __fmul_rn(an, factorPI);
            D.selab[2*(size_t)g2] = (float)cos((double)rad); D.selab[2*(size_t)g2 + 1] = (float)sin((double)rad);
        }
    }
}

// ---------------------------------------------------------------- Gaussian blur 7x7, Q8 separable, reflect101 at the image edge
#define BT_W 64
#define BT_H 64
#define BT_R 4                  // output rows per thread of the vertical pass
__device__ __forceinline__ void blur_body(const OrbDev &D, const int block, const int nblocks) {
    // one launch for all levels (the small levels do not fill the chip on their own): block -> (frame, level, tile)
    const int bid = xcd_order(block, nblocks);
    const int f = bid / D.btiles_per_frame, bt = bid % D.btiles_per_frame;
    int l = 0;
    while (l + 1 < D.nlevels && bt >= D.L[l+1].bt0) l++;
    const LevelGeo &G = D.L[l];
    // row sums of the horizontal pass as 16-bit values (256 x 255 at most).  The first version kept them as 32-bit words and had every thread of
    // the vertical pass read its seven rows for ONE output row: 135 KB of LDS traffic per 1024 pixels -- the kernel ran at the LDS's rate (108 us),
    // not at memory's.  16-bit sums and four output rows per thread (ten rows read for four written): 4.4 x less.
    __shared__ __attribute__((aligned(16))) unsigned short rowf[(BT_H + 6)*BT_W];
    const int t = bt - G.bt0, ntx = (G.w + BT_W - 1)/BT_W, tx = t % ntx, ty = t / ntx;
    const uint8_t *src = D.pyr + (size_t)f*D.pyr_frame + G.pyr_off + (size_t)EDGE*G.bw + EDGE;
    const int x0 = tx*BT_W, y0 = ty*BT_H, tid = threadIdx.x;
    const int nrow = min(BT_H, G.h - y0) + 6;                   // rows of horizontal sums this tile needs
    int gk[7];
#pragma unroll
    for (int i = 0; i < 7; i++) gk[i] = D.gk[i];
    // horizontal pass, four neighbouring outputs per thread from three (unaligned) dword loads, two rows per thread in flight.  The level sits in a
    // 19-px REFLECT_101 frame (ComputePyramid's copyMakeBorder) -- the blur's own border rule: the 3-px apron is read straight from the frame
    for (int k0 = tid; k0 < nrow*(BT_W/4); k0 += 512) {
        uint32_t w[2][3]; int yy[2], xx[2]; bool on[2];
#pragma unroll
        for (int u = 0; u < 2; u++) { const int k = k0 + 256*u; yy[u] = k / (BT_W/4); xx[u] = 4*(k % (BT_W/4)); on[u] = yy[u] < nrow && x0 + xx[u] < G.w;
            const int y = min(y0 + yy[u] - 3, G.h + 2);
            const uint8_t *r = src + (ptrdiff_t)y*G.bw + x0 + xx[u] - 3;         // bytes x-3 .. x+8 (x+6 is the last one used): inside the frame
            w[u][0] = on[u] ? *(const u32_unaligned *)r : 0u; w[u][1] = on[u] ? *(const u32_unaligned *)(r + 4) : 0u; w[u][2] = on[u] ? *(const u32_unaligned *)(r + 8) : 0u; }
#pragma unroll
        for (int u = 0; u < 2; u++) {
            if (yy[u] >= nrow) continue;
            int p[12];
#pragma unroll
            for (int i = 0; i < 4; i++) { p[i] = (w[u][0] >> (8*i)) & 255; p[4 + i] = (w[u][1] >> (8*i)) & 255; p[8 + i] = (w[u][2] >> (8*i)) & 255; }
            int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
            for (int i = 0; i < 7; i++) { s0 += __mul24(gk[i], p[i]); s1 += __mul24(gk[i], p[i + 1]); s2 += __mul24(gk[i], p[i + 2]); s3 += __mul24(gk[i], p[i + 3]); }     // (Q8 weights x bytes; v_mul_lo_u32 runs at a quarter of the rate)
            *(uint2 *)&rowf[yy[u]*BT_W + xx[u]] = make_uint2((uint32_t)s0 | ((uint32_t)s1 << 16), (uint32_t)s2 | ((uint32_t)s3 << 16));
        }
    }
    __syncthreads();
    uint8_t *dst = D.blur + (size_t)f*D.blur_frame + G.blur_off;
    {   // vertical pass: thread = (strip of BT_R rows, four neighbouring columns), one dword store per row where the four columns exist
        const int ys = BT_R*(tid / (BT_W/4)), xx = 4*(tid % (BT_W/4)), x = x0 + xx;
        if (x < G.w && y0 + ys < G.h) {
            uint2 q[BT_R + 6];
#pragma unroll
            for (int i = 0; i < BT_R + 6; i++) q[i] = *(const uint2 *)&rowf[min(ys + i, nrow - 1)*BT_W + xx];
#pragma unroll
            for (int r = 0; r < BT_R; r++) {
                const int y = y0 + ys + r;
                if (y >= G.h) break;
                int s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
                for (int i = 0; i < 7; i++) { const uint2 v = q[r + i];
                    s0 += __mul24(gk[i], (int)(v.x & 0xffffu)); s1 += __mul24(gk[i], (int)(v.x >> 16)); s2 += __mul24(gk[i], (int)(v.y & 0xffffu)); s3 += __mul24(gk[i], (int)(v.y >> 16)); }
                const int v0 = min(max((s0 + (1 << 15)) >> 16, 0), 255), v1 = min(max((s1 + (1 << 15)) >> 16, 0), 255);
                const int v2 = min(max((s2 + (1 << 15)) >> 16, 0), 255), v3 = min(max((s3 + (1 << 15)) >> 16, 0), 255);
                uint8_t *o = dst + (size_t)y*G.w + x;
                if (x + 3 < G.w) *(u32_unaligned *)o = (uint32_t)v0 | ((uint32_t)v1 << 8) | ((uint32_t)v2 << 16) | ((uint32_t)v3 << 24);
                else { o[0] = (uint8_t)v0; if (x + 1 < G.w) o[1] = (uint8_t)v1; if (x + 2 < G.w) o[2] = (uint8_t)v2; }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_blur(OrbDev D) { blur_body(D, blockIdx.x, gridDim.x); }
__global__ __launch_bounds__(256) void k_orient(OrbDev D) { orient_body(D, blockIdx.x, gridDim.x); }
// A batch: orientation (waits for the pyramid's lines around 1000 keypoints per frame: latency) and blur (every pixel once: bandwidth) need different things of the device
// and nothing of each other: one launch, the orientation's workgroups first.
__global__ __launch_bounds__(256) void k_orient_blur(OrbDev D, int norient) {
    if ((int)blockIdx.x < norient) orient_body(D, blockIdx.x, norient);
    else blur_body(D, blockIdx.x - norient, gridDim.x - norient);
}
// A few frames: the blur's tiles ride in the detector's launch (both read the pyramid only; the blur as a launch of its own is 6.7 us of the ~100 of a per-frame
// call, 1.7 of them work) -- the detector's workgroups first.  On a batch the two stay apart (k_fast's small-tile instance has half the threads and a third of the LDS).
__global__ __launch_bounds__(256) void k_fast_blur(OrbDev D, int nfast) {
    if ((int)blockIdx.x < nfast) fast_body<TILE_MAX, 2048, CELL_CAP, 256, 1>(D, blockIdx.x, nfast);
    else blur_body(D, blockIdx.x - nfast, gridDim.x - nfast);
}

// ---------------------------------------------------------------- descriptors: 32 lanes per keypoint, lane i -> byte i; straight into the level-major output
// (ORBextractor.cc:1106-1112: coordinates scaled back to level 0).  A keypoint's place is its level's first place -- the counts of the levels before it -- plus
// its place in the level: the separate packing launch (a frame's keypoints and descriptors read back and written again, 4.7 us of the 107 of a per-frame call)
// is gone since round 6.
// byte `lane` of the descriptor of the keypoint sv = (x, y, response, angle) with a, b = cos, sin of its angle, and the keypoint's six output values, at place o of frame f
#define DW_R 18                 // the descriptor's window: rows / columns -18 .. 18 around the keypoint
#define DW_H (2*DW_R + 1)
#define DW_P 40                 // bytes per window row in LDS (ten dwords)
template <bool HOST>
__device__ __forceinline__ void describe_out(const OrbDev &D, const LevelGeo &G, int f, int l, int o, int lane, const float4 sv, const float a, const float b) {
    // The 512 taps of a keypoint lie within 18 px of it (the pattern's coordinates are at most 13 on either axis: 13 sqrt 2 = 18.4, rounded 18).  Straight from the blurred level they
    // were 16 byte loads per lane, each a wave instruction that touches up to 64 different cache lines: the kernel ran at the rate its compute unit's vector cache looks lines up
    // (45 us for 64 frames, the lines of 64 k windows several times over).  The window once, as rows of ten dwords, into LDS (37 rows x 40 bytes; a keypoint is at least 19 px inside
    // its level, the three bytes past the window's right edge at worst the start of the next row), the taps from there.
    __shared__ __attribute__((aligned(4))) uint8_t s_win[8][DW_H*DW_P];
    uint8_t *win = s_win[threadIdx.x >> 5];
    {   const uint8_t *c = D.blur + (size_t)f*D.blur_frame + G.blur_off + (size_t)((int)rintf(sv.y) - DW_R)*G.w + (int)rintf(sv.x) - DW_R;
        uint32_t v[12];
#pragma unroll
        for (int k = 0; k < 12; k++) { const int idx = min(lane + 32*k, DW_H*(DW_P/4) - 1), row = idx/(DW_P/4), col = idx - row*(DW_P/4); v[k] = *(const u32_unaligned *)(c + (size_t)row*G.w + 4*col); }
#pragma unroll
        for (int k = 0; k < 12; k++) if (lane + 32*k < DW_H*(DW_P/4)) ((uint32_t *)win)[lane + 32*k] = v[k];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (a keypoint's 32 lanes are one half of a wave: its LDS accesses are in order)
    const uint8_t *c = win + DW_R*DW_P + DW_R;
    const int8_t *pat = d_pattern + 32*lane;
    int val = 0;
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const float x0 = pat[4*t], y0 = pat[4*t+1], x1 = pat[4*t+2], y1 = pat[4*t+3];
        const int t0 = c[(int)rintf(__fadd_rn(__fmul_rn(x0, b), __fmul_rn(y0, a)))*DW_P + (int)rintf(__fsub_rn(__fmul_rn(x0, a), __fmul_rn(y0, b)))];
        const int t1 = c[(int)rintf(__fadd_rn(__fmul_rn(x1, b), __fmul_rn(y1, a)))*DW_P + (int)rintf(__fsub_rn(__fmul_rn(x1, a), __fmul_rn(y1, b)))];
        val |= (t0 < t1) << t;
    }
    D.out_desc[((size_t)f*D.cap + o)*32 + lane] = (uint8_t)val;
    if (HOST) D.h_desc[((size_t)f*D.cap + o)*32 + lane] = (uint8_t)val;
    if (lane < 6) {                                             // x, y, size, angle, response, octave
        const float kx = l ? __fmul_rn(sv.x, G.sf) : sv.x, ky = l ? __fmul_rn(sv.y, G.sf) : sv.y, ks = (float)(int)__fmul_rn((float)PATCH_SIZE, G.sf);
        const float kv = lane == 0 ? kx : lane == 1 ? ky : lane == 2 ? ks : lane == 3 ? sv.w : lane == 4 ? sv.z : (float)l;
        D.out_kp[((size_t)f*D.cap + o)*6 + lane] = kv;
        if (HOST) D.h_kp[((size_t)f*D.cap + o)*6 + lane] = kv;
    }
}
__global__ __launch_bounds__(256) void k_describe(OrbDev D) {
    const int g = (xcd_order(blockIdx.x, gridDim.x)*256 + threadIdx.x) >> 5, lane = threadIdx.x & 31;      // 32 lanes per keypoint: two keypoints per wave
    const int per = D.slots_per_frame, f = g / per, slot = g % per;
    if (f >= D.n) return;
    int l = 0; while (l + 1 < D.nlevels && slot >= D.L[l+1].kp0) l++;
    const LevelGeo &G = D.L[l];
    const int *sc = D.selcnt + (size_t)f*D.nlevels;
    int before = 0, total = 0, mine = 0;
#pragma unroll
    for (int k = 0; k < MAXL; k++) { const int cnt = k < D.nlevels ? sc[k] : 0; before += k < l ? cnt : 0; mine = k == l ? cnt : mine; total += cnt; }
    if (slot == 0 && lane == 0) D.out_cnt[f] = min(total, D.cap);
    const int o = before + slot - G.kp0;
    if ((slot - G.kp0) >= mine || o >= D.cap) return;
    const float4 sv = *(const float4 *)(D.sel + ((size_t)f*per + slot)*4);
    const float a = D.selab[2*((size_t)f*per + slot)], b = D.selab[2*((size_t)f*per + slot) + 1];      // cos, sin of the angle (k_orient)
    describe_out<false>(D, G, f, l, o, lane, sv, a, b);
}
// A few frames: orientation and descriptor in one launch (each launch of the per-frame chain costs ~5 us before its first instruction).  The keypoint's 32 lanes: sixteen
// take the patch's rows, lane 0 the angle and its cosine / sine (fp64 library code: on a batch that is why k_orient does them sixteen keypoints per wave pass, not two).
__global__ __launch_bounds__(256) void k_orient_describe(OrbDev D) {
    const int g = (xcd_order(blockIdx.x, gridDim.x)*256 + threadIdx.x) >> 5, lane = threadIdx.x & 31;
    const int per = D.slots_per_frame, f = g / per, slot = g % per;
    if (f >= D.n) return;
    int l = 0; while (l + 1 < D.nlevels && slot >= D.L[l+1].kp0) l++;
    const LevelGeo &G = D.L[l];
    const int *sc = D.selcnt + (size_t)f*D.nlevels;
    int before = 0, total = 0, mine = 0;
#pragma unroll
    for (int k = 0; k < MAXL; k++) { const int cnt = k < D.nlevels ? sc[k] : 0; before += k < l ? cnt : 0; mine = k == l ? cnt : mine; total += cnt; }
    if (slot == 0 && lane == 0) { D.out_cnt[f] = min(total, D.cap); D.h_cnt[f] = min(total, D.cap); }
    const int o = before + slot - G.kp0;
    if ((slot - G.kp0) >= mine || o >= D.cap) return;
    float4 sv = *(const float4 *)(D.sel + ((size_t)f*per + slot)*4);
    int m10 = 0, m01 = 0;
    if (lane < 16) orient_rows(D, G, f, sv.x, sv.y, lane, m10, m01);
#pragma unroll
    for (int w = 8; w > 0; w >>= 1) { m10 += __shfl_xor(m10, w, 16); m01 += __shfl_xor(m01, w, 16); }
    float angle = 0.f, a = 0.f, b = 0.f;
    if (lane == 0) {
        angle = fast_atan2f_dev((float)m01, (float)m10);
        const float factorPI = (float)(3.14159265358979323846/180.f);
        const float rad = __fmul_rn(angle, factorPI);
        a = (float)cos((double)rad); b = (float)sin((double)rad);
    }
    sv.w = __shfl(angle, 0, 32); a = __shfl(a, 0, 32); b = __shfl(b, 0, 32);
    describe_out<true>(D, G, f, l, o, lane, sv, a, b);
}

// ------------------------------------------------------------------------------------------------ host side

// ================================================================== window / projection search
// frame::AssignFeaturesToGrid / PosInGrid (frame.cc:372-407), frame::GetFeaturesInArea (frame.cc:415-468),
// tracking::DescriptorDistance (tracking.cc:2762-2778): the feature grid of the searched frame is built on the device (stable
// counting sort: a cell lists its features in index order, as the reference's push_back does), one thread per query walks the
// cells in the reference's order, filters by octave and window, and scores the candidates with a 256-bit Hamming distance.
#define MG_COLS 64
#define MG_ROWS 48
#define MG_CELLS (MG_COLS*MG_ROWS)
struct MatchDev {
    const float *kp; const uint8_t *desc; int n;      // [n][6], [n][32]
    double min_x, min_y, iw, ih;
    int *cell, *off, *list;                           // [n], [MG_CELLS + 1], [n]
};
__global__ __launch_bounds__(256) void k_mg_cell(MatchDev M) {
    const int i = blockIdx.x*256 + threadIdx.x;
    if (i >= M.n) return;
    const int px = (int)round(((double)M.kp[6*i] - M.min_x)*M.iw), py = (int)round(((double)M.kp[6*i+1] - M.min_y)*M.ih);   // PosInGrid
    const int c = (px < 0 || px >= MG_COLS || py < 0 || py >= MG_ROWS) ? -1 : px*MG_ROWS + py;
    M.cell[i] = c;
    if (c >= 0) atomicAdd(&M.off[c + 1], 1);
}
__global__ __launch_bounds__(1024) void k_mg_scan(MatchDev M) {          // inclusive scan of the 3072 cell counts, one workgroup
    __shared__ int part[1024];
    const int t = threadIdx.x;
    int v[3], s = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) { v[k] = M.off[1 + 3*t + k]; s += v[k]; }
    part[t] = s; __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { int a = t >= d ? part[t - d] : 0; __syncthreads(); part[t] += a; __syncthreads(); }
    int run = part[t] - s;
#pragma unroll
    for (int k = 0; k < 3; k++) { run += v[k]; M.off[1 + 3*t + k] = run; }
}
__global__ __launch_bounds__(256) void k_mg_place(MatchDev M) {           // rank inside the cell = features of the cell with a smaller index
    // (64 features per workgroup, four threads each: the features before the workgroup's last one through LDS, 1024 at a time, a quarter of a tile per thread -- every thread
    // walking all of them in global memory was 40 us for 1000 features)
    __shared__ int s_cell[1024];
    const int i = blockIdx.x*64 + (threadIdx.x >> 2), part = threadIdx.x & 3, iend = min((int)(blockIdx.x + 1)*64, M.n);
    const int c = i < M.n ? M.cell[i] : -1;
    int rank = 0;
    for (int t0 = 0; t0 < iend; t0 += 1024) {
#pragma unroll
        for (int u = 0; u < 4; u++) { const int j = t0 + threadIdx.x + 256*u; s_cell[threadIdx.x + 256*u] = j < iend ? M.cell[j] : -2; }
        __syncthreads();
        const int n = min(1024, i - t0), j0 = 256*part, j1 = min(n, j0 + 256);         // the features of this tile before feature i; this thread's quarter of the tile
        if (c >= 0) for (int j = j0; j < j1; j += 8) {
#pragma unroll
            for (int u = 0; u < 8; u++) rank += (j + u < j1 && s_cell[j + u] == c) ? 1 : 0; }
        __syncthreads();
    }
    rank += __shfl_xor(rank, 1, 4); rank += __shfl_xor(rank, 2, 4);
    if (c >= 0 && part == 0) M.list[M.off[c] + rank] = i;
}
// One WAVE per query (until round 6: one thread per query walking its window's cells one after the other -- two dependent loads per cell, two more per feature, 81 cells for a
// 40-px radius: 289 us for 1000 queries, 0.37 ms per call of tracking::SearchFrom3D's search).  The window's cells on the lanes in the reference's order (column by column,
// frame.cc:415-468), their feature counts as a wave scan; then the window's features on the lanes, again in that order (a lane finds its feature's cell in the scan), the
// reference's filters and the Hamming distance per lane; the survivors keep their order through a ballot, best / second best as the two smallest of the multiset with the
// first minimum's index (what the reference's sequential scan ends with).
__global__ __launch_bounds__(256) void k_match(MatchDev M, int nq, const float *qxy, const float *qr, const int *qlev, const uint8_t *qdesc, int max_cand,
                                               int *cand_idx, int *cand_dist, int *cand_cnt, int *best_idx, int *best_dist, int *best_dist2) {
    __shared__ int s_inc[4][64], s_o0[4][64];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, q = blockIdx.x*4 + wv;
    if (q >= nq) return;                                     // (a whole wave)
    int *inc = s_inc[wv], *o0s = s_o0[wv];
    const float x = qxy[2*q], y = qxy[2*q+1], r = qr[q];
    const int minLevel = qlev ? qlev[2*q] : -1, maxLevel = qlev ? qlev[2*q+1] : -1;
    uint32_t qd[8];
#pragma unroll
    for (int k = 0; k < 8; k++) qd[k] = ((const uint32_t *)qdesc)[8*(size_t)q + k];
    int nc = 0, bi = -1, bd = 2147483647, bd2 = 2147483647;
    const int c0x = max(0, (int)floor(((double)x - M.min_x - (double)r)*M.iw)), c1x = min(MG_COLS - 1, (int)ceil(((double)x - M.min_x + (double)r)*M.iw));
    const int c0y = max(0, (int)floor(((double)y - M.min_y - (double)r)*M.ih)), c1y = min(MG_ROWS - 1, (int)ceil(((double)y - M.min_y + (double)r)*M.ih));
    if (!(c0x >= MG_COLS || c1x < 0 || c0y >= MG_ROWS || c1y < 0)) {
        const bool check = (minLevel > 0) || (maxLevel >= 0);
        const int ny = c1y - c0y + 1, ncell = (c1x - c0x + 1)*ny;
        const unsigned long long lt = (1ull << lane) - 1ull;
        for (int e0 = 0; e0 < ncell; e0 += 64) {             // 64 cells of the window at a time
            const int e = e0 + lane; int o0 = 0, cnt = 0;
            if (e < ncell) { const int ix = c0x + e/ny, iy = c0y + e - (e/ny)*ny, c = ix*MG_ROWS + iy; o0 = M.off[c]; cnt = M.off[c+1] - o0; }
            int incl = cnt;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
            const int tot = __shfl(incl, 63, 64);
            inc[lane] = incl; o0s[lane] = o0 - (incl - cnt);                                 // (list position of the cell's first feature minus the features before the cell)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            for (int j0 = 0; j0 < tot; j0 += 64) {            // 64 features of those cells at a time, in order
                const int j = j0 + lane; bool pass = false; int i = 0, d = 0;
                if (j < tot) {
                    int lo = 0;                                // the first cell whose inclusive count exceeds j
#pragma unroll
                    for (int step = 32; step > 0; step >>= 1) if (inc[lo + step - 1] <= j) lo += step;
                    i = M.list[o0s[lo] + j];
                    const float fx = M.kp[6*i], fy = M.kp[6*i+1]; const int oct = (int)M.kp[6*i+5];
                    const uint32_t *fd = (const uint32_t *)(M.desc + 32*(size_t)i);
                    uint32_t f8[8];
#pragma unroll
                    for (int w = 0; w < 8; w++) f8[w] = fd[w];
                    pass = true;
                    if (check) { if (oct < minLevel) pass = false; if (maxLevel >= 0 && oct > maxLevel) pass = false; }
                    const float dx = fx - x, dy = fy - y;
                    if (!(fabsf(dx) < r && fabsf(dy) < r)) pass = false;
#pragma unroll
                    for (int w = 0; w < 8; w++) d += __popc(qd[w] ^ f8[w]);
                }
                const unsigned long long pm = __ballot(pass);
                if (pm) {
                    const int rank = nc + __popcll(pm & lt);
                    if (pass && rank < max_cand) { cand_idx[(size_t)q*max_cand + rank] = i; cand_dist[(size_t)q*max_cand + rank] = d; }
                    nc += __popcll(pm);
                    // the chunk's two smallest distances and the first lane of the smallest
                    int m1 = pass ? d : 2147483647;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) m1 = min(m1, __shfl_xor(m1, o, 64));
                    const unsigned long long at1 = __ballot(pass && d == m1);
                    const int first = __ffsll((long long)at1) - 1;
                    int m2 = (pass && lane != first) ? d : 2147483647;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) m2 = min(m2, __shfl_xor(m2, o, 64));
                    const int i1 = __shfl(i, first, 64);
                    // merged behind what came before (a later equal minimum does not take the index: strict < in the reference)
                    if (m1 < bd) { bd2 = min(bd, m2); bd = m1; bi = i1; } else bd2 = min(bd2, m1);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (the next 64 cells overwrite the scan)
        }
    }
    for (int k = nc + lane; k < max_cand; k += 64) { cand_idx[(size_t)q*max_cand + k] = -1; cand_dist[(size_t)q*max_cand + k] = -1; }     // unused slots
    if (lane == 0) { cand_cnt[q] = nc; best_idx[q] = bi; best_dist[q] = bd; best_dist2[q] = bd2; }
}

struct OCtx {
    int device = 0; hipStream_t stream = nullptr; std::string err;
    int nfeatures = 1000, nlevels = 8, ini_th = 20, min_th = 7; float scale = 1.2f;
    float sf[MAXL], isf[MAXL]; int nfl[MAXL], umax[16], gk[7];
    std::vector<void *> allocs; bool uploaded = false; int fast_shape = -1;       // (tsorb_debug_fast_shape)
    bool out_on_host = false;                                                   // the last run stored its results in h_out itself (a few frames)
    int merge_ob = 1;                                                           // (tsorb_debug_pyramid 200 / 201) a batch's orientation and blur in one launch
    int *h_fb = nullptr; int fallbacks = 0;                                     // the run's fallback word (pinned), runs that took the serial pass
    int pyr_shape = -1, pyr_split = P1_SPLIT; PyrOne Q[3 + MAXL/2 + 1]; int q_inst[3 + MAXL/2 + 1], n_pairs = 0;           // (tsorb_debug_pyramid) k_pyramid_one's launches: [0] levels 0 .. split from the image, [1] the rest from level split, [2] every level from the image; instance 0 = small buffers, 1 = large, -1 = does not fit
    OrbDev D;
    // the SLAM front-end calls once per frame with the same geometry: buffers and pinned staging are kept between calls
    MatchDev M; bool m_set = false; void *m_buf = nullptr; size_t m_cap = 0; void *m_feat = nullptr; size_t m_feat_cap = 0;   // search grid of the current frame
    void *mq_dev = nullptr, *mq_host = nullptr; size_t mq_cap = 0;
    int key[5] = {0, 0, 0, 0, 0}; uint8_t *h_img = nullptr; void *h_out = nullptr; size_t h_img_sz = 0, h_out_sz = 0;
};
static int cv_round_f(float v) { return (int)lrintf(v); }
#define OCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { c->err = std::string(#x) + ": " + hipGetErrorString(e_); return TSORB_ERR_DEVICE; } } while (0)
template <typename T> static int oalloc(OCtx *c, T **p, size_t n) { void *q = nullptr; if (hipMalloc(&q, std::max<size_t>(n, 1)*sizeof(T)) != hipSuccess) { c->err = "hipMalloc failed"; return TSORB_ERR_DEVICE; }
    c->allocs.push_back(q); *p = (T *)q; return 0; }
static void ofree(OCtx *c) { hipStreamSynchronize(c->stream); for (void *p : c->allocs) hipFree(p); c->allocs.clear(); c->uploaded = false; c->key[0] = 0; c->m_set = false;
    if (c->h_img) hipHostFree(c->h_img); if (c->h_out) hipHostFree(c->h_out); c->h_img = nullptr; c->h_out = nullptr; c->h_img_sz = c->h_out_sz = 0; }

extern "C" {

int tsorb_create(void **ctx, int nfeatures, float scale, int nlevels, int ini_th, int min_th, int device) {
    if (!ctx || nlevels < 1 || nlevels > MAXL || nfeatures < 1 || !(scale > 1.f)) return TSORB_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return TSORB_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return TSORB_ERR_DEVICE;
    OCtx *c = new OCtx(); c->device = device; c->nfeatures = nfeatures; c->scale = scale; c->nlevels = nlevels; c->ini_th = ini_th; c->min_th = min_th;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return TSORB_ERR_DEVICE; }
    if (hipHostMalloc((void **)&c->h_fb, 64, hipHostMallocDefault) != hipSuccess) { hipStreamDestroy(c->stream); delete c; return TSORB_ERR_DEVICE; }
    *c->h_fb = 0;
    // ORBextractor::ORBextractor, ORBextractor.cc:410-471 (fp32 arithmetic as there)
    c->sf[0] = 1.0f; for (int i = 1; i < nlevels; i++) c->sf[i] = c->sf[i-1]*scale;
    for (int i = 0; i < nlevels; i++) c->isf[i] = 1.0f/c->sf[i];
    float factor = 1.0f/scale;
    float nd = nfeatures*(1 - factor)/(1 - (float)pow((double)factor, (double)nlevels));
    int sum = 0;
    for (int l = 0; l < nlevels - 1; l++) { c->nfl[l] = cv_round_f(nd); sum += c->nfl[l]; nd *= factor; }
    c->nfl[nlevels-1] = std::max(nfeatures - sum, 0);
    int v, v0, vmax = (int)floor(HALF_PATCH*sqrtf(2.f)/2 + 1), vmin = (int)ceil(HALF_PATCH*sqrtf(2.f)/2);
    const double hp2 = HALF_PATCH*HALF_PATCH;
    memset(c->umax, 0, sizeof(c->umax));
    for (v = 0; v <= vmax; ++v) c->umax[v] = (int)lrint(sqrt(hp2 - v*v));
    for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) { while (c->umax[v0] == c->umax[v0+1]) ++v0; c->umax[v] = v0; ++v0; }
    // cv::getGaussianKernel(7, 2, CV_32F) -> Q8 (OpenCV 3.3 8-bit separable path)
    { float cf[7]; double s = 0, s2 = -0.5/(2.0*2.0); for (int i = 0; i < 7; i++) { double x = i - 3.0; cf[i] = (float)exp(s2*x*x); s += cf[i]; }
      s = 1./s; for (int i = 0; i < 7; i++) { cf[i] = (float)(cf[i]*s); c->gk[i] = cv_round_f(cf[i]*256.f); } }
    if (hipMemcpyToSymbol(HIP_SYMBOL(d_pattern), ORB_BIT_PATTERN_31, 1024) != hipSuccess) { delete c; return TSORB_ERR_DEVICE; }
    *ctx = c; return TSORB_OK;
}
int tsorb_destroy(void *ctx) { OCtx *c = (OCtx *)ctx; if (!c) return TSORB_ERR_ARG; hipSetDevice(c->device); ofree(c);
    if (c->h_fb) hipHostFree(c->h_fb);
    if (c->m_buf) hipFree(c->m_buf); if (c->m_feat) hipFree(c->m_feat); if (c->mq_dev) hipFree(c->mq_dev); if (c->mq_host) hipHostFree(c->mq_host);
    hipStreamDestroy(c->stream); delete c; return TSORB_OK; }
const char *tsorb_last_error(void *ctx) { return ctx ? ((OCtx *)ctx)->err.c_str() : "null ctx"; }
int tsorb_get_levels(void *ctx) { return ctx ? ((OCtx *)ctx)->nlevels : TSORB_ERR_ARG; }
int tsorb_get_scale_factors(void *ctx, float *sf, float *isf) { OCtx *c = (OCtx *)ctx; if (!c) return TSORB_ERR_ARG;
    for (int i = 0; i < c->nlevels; i++) { if (sf) sf[i] = c->sf[i]; if (isf) isf[i] = c->isf[i]; } return TSORB_OK; }
int tsorb_get_features_per_level(void *ctx, int32_t *n) { OCtx *c = (OCtx *)ctx; if (!c || !n) return TSORB_ERR_ARG; for (int i = 0; i < c->nlevels; i++) n[i] = c->nfl[i]; return TSORB_OK; }

int tsorb_upload(void *ctx, const uint8_t *imgs, int n, int w, int h, int stride, int cap) {
    OCtx *c = (OCtx *)ctx; if (!c || !imgs || n < 1 || w < 64 || h < 64 || stride < w || cap < 1) return TSORB_ERR_ARG;
    hipSetDevice(c->device);
    c->out_on_host = false;                                  // (until the next run)
    if (c->uploaded && c->key[0] == n && c->key[1] == w && c->key[2] == h && c->key[3] == stride && c->key[4] == cap) {
        // same geometry as the previous call: only the pixels travel (pinned staging, one asynchronous copy)
        memcpy(c->h_img, imgs, (size_t)n*h*stride);
        OCK(hipMemcpyAsync((void *)c->D.img, c->h_img, (size_t)n*h*stride, hipMemcpyHostToDevice, c->stream));
        return TSORB_OK;
    }
    ofree(c);
    OrbDev &D = c->D; memset(&D, 0, sizeof(D));
    { void *dp = nullptr; if (hipHostGetDevicePointer(&dp, c->h_fb, 0) != hipSuccess) { c->err = "hipHostGetDevicePointer failed"; return TSORB_ERR_DEVICE; } D.h_fallback = (int *)dp; }
    D.n = n; D.nlevels = c->nlevels; D.ini_th = c->ini_th; D.min_th = c->min_th; D.w = w; D.h = h; D.stride = stride; D.cap = cap;
    memcpy(D.umax, c->umax, sizeof(D.umax)); memcpy(D.gk, c->gk, sizeof(D.gk));
    size_t po = 0, bo = 0; int cell0 = 0, kp0 = 0;
    for (int l = 0; l < c->nlevels; l++) {
        LevelGeo &G = D.L[l];
        G.w = cv_round_f((float)w*c->isf[l]); G.h = cv_round_f((float)h*c->isf[l]); G.bw = G.w + 2*EDGE; G.bh = G.h + 2*EDGE;
        G.minB = EDGE - 3; G.maxBX = G.w - EDGE + 3; G.maxBY = G.h - EDGE + 3;
        const float width = (float)(G.maxBX - G.minB), height = (float)(G.maxBY - G.minB);
        G.nCols = (int)(width/30.f); G.nRows = (int)(height/30.f);
        if (G.nCols < 1 || G.nRows < 1) { c->err = "image too small for the requested pyramid"; return TSORB_ERR_ARG; }
        G.wCell = (int)ceilf(width/G.nCols); G.hCell = (int)ceilf(height/G.nRows);
        if (G.wCell + 6 > TILE_MAX || G.hCell + 6 > TILE_MAX) { c->err = "cell larger than the LDS tile"; return TSORB_ERR_ARG; }
        { const int shape = c->fast_shape;                   // (tsorb_debug_fast_shape: 0 every level through the general instance, 1 - 3 the split at any batch size)
          // below ~24 frames the device is not full and a cell's own time counts: four waves per cell, one launch (measured: 1 frame 0.144 against 0.157 ms
          // split, 16 frames equal, 32 frames 0.299 against 0.292, 64 frames 0.446 against 0.426)
          const bool one_group = shape == 0 || (shape < 0 && n < 24);
          const int g = (G.wCell + 6 > 40 || G.hCell + 6 > 40 || one_group) ? 1 : 0;          // (rows are staged as dwords: up to 3 bytes past the ROI, still inside a stride of 40)
          if (l == 0) { D.fast_nl[0] = D.fast_nl[1] = D.fast_cells[0] = D.fast_cells[1] = 0; }
          D.fast_c0[g][D.fast_nl[g]] = D.fast_cells[g]; D.fast_lv[g][D.fast_nl[g]++] = (signed char)l; D.fast_cells[g] += G.nCols*G.nRows; }
        G.cell0 = cell0; cell0 += G.nCols*G.nRows;
        G.nfeat = c->nfl[l]; G.capL = c->nfl[l] + 8; G.kp0 = kp0; kp0 += G.capL; G.sf = c->sf[l];
        G.pyr_off = po; po += (size_t)G.bw*G.bh; G.blur_off = bo; bo += (size_t)G.w*G.h;
    }
    D.cells_per_frame = cell0; D.slots_per_frame = kp0; D.pyr_frame = po; D.blur_frame = bo;
    { int bt = 0; for (int l = 0; l < c->nlevels; l++) { LevelGeo &G = D.L[l]; G.bt0 = bt; bt += ((G.w + BT_W - 1)/BT_W)*((G.h + BT_H - 1)/BT_H); } D.btiles_per_frame = bt; }
    int rt = 0;
    for (int l = 1; l < c->nlevels; l++) { LevelGeo &G = D.L[l]; G.rx_off = rt; rt += 2*((G.bw + 1) & ~1); G.ry_off = rt; rt += 4*G.bh; }
    for (int l = 1; l < c->nlevels; l++) { D.rsx[l] = 1.0/((double)D.L[l].w/(double)D.L[l-1].w); D.rsy[l] = 1.0/((double)D.L[l].h/(double)D.L[l-1].h); }
    {   // k_pyramid_one's tilings: 64 x 16 one or two levels below the base, 32 x 16 three, 16 x 16 further down (a tile's cost grows with its depth), halved until
        // every stage's region (bounded from above: a range of n pixels reads at most ceil(n scale) + 3 of the level before; + 4 taken) fits the LDS buffers and tables
        const int NL = c->nlevels, sp = std::min(std::max(c->pyr_split, 0), NL - 1);
        auto tiling = [&](int base, int top, PyrOne &Q, int &qi) {
            memset(&Q, 0, sizeof(Q)); Q.base = base; Q.top = top; qi = 0;
            for (int l = Q.base + 1; l <= Q.top; l++) {
                const int dep = l - Q.base; int tw = dep <= 2 ? 64 : dep == 3 ? 32 : 16, th = 16;
                for (;;) {
                    int wk = tw, hk = th, sw = tw, sh = th, need = 0;
                    for (int k = l; k > Q.base; k--) { wk = std::min((int)ceil(wk*D.rsx[k]) + 4, D.L[k-1].w); hk = std::min((int)ceil(hk*D.rsy[k]) + 4, D.L[k-1].h);
                        need = std::max(need, 4*((wk + 3)/4)*hk); if (k > Q.base + 1) { sw += wk; sh += hk; } }
                    const int inst = (need <= 4096 && sw <= 192 && sh <= 192) ? 0 : (need <= 16384 && sw <= 512 && sh <= 512) ? 1 : -1;
                    if (inst >= 0) { qi = qi < 0 ? -1 : std::max(qi, inst); break; }
                    if (tw > 8) tw /= 2; else if (th > 4) th /= 2; else { qi = -1; break; }
                }
                Q.tw[l] = tw; Q.th[l] = th; Q.ncol[l] = (D.L[l].bw + tw - 1)/tw;
            }
            int t = 0; Q.nl = Q.top - Q.base + (Q.base == 0 ? 1 : 0);
            for (int i = 0; i < Q.nl; i++) { const int l = Q.top - i; Q.t0[i] = t;
                t += l == 0 ? (D.L[0].bh + P1_L0_ROWS - 1)/P1_L0_ROWS : Q.ncol[l]*((D.L[l].bh + Q.th[l] - 1)/Q.th[l]); }
            Q.t0[Q.nl] = t; Q.per_frame = t;
        };
        tiling(0, sp, c->Q[0], c->q_inst[0]); tiling(sp, NL - 1, c->Q[1], c->q_inst[1]); tiling(0, NL - 1, c->Q[2], c->q_inst[2]);
        c->n_pairs = 0;                                       // (tsorb_debug_pyramid 3) two levels per launch: 0 - 1 from the image, 2 - 3 from level 1, ..
        for (int b0 = 0; b0 < NL; b0 += 2) { const int q = 3 + c->n_pairs++; tiling(b0 == 0 ? 0 : b0 - 1, std::min(b0 + 1, NL - 1), c->Q[q], c->q_inst[q]); }
    }
    // strict 3x3 NMS leaves at most one corner per 2x2 block: the level-0 search area bounds every level's candidate count
    D.cand_cap = ((D.L[0].maxBX - D.L[0].minB)*(D.L[0].maxBY - D.L[0].minB))/4 + 64; D.node_cap = 64*(c->nfl[0] + 64); D.pool_cap = 16*D.cand_cap;
    int rc;
    uint8_t *img; if ((rc = oalloc(c, &img, (size_t)n*h*stride + 8))) return rc; D.img = img;        // (+ 8: level 1's 8-byte source loads at the end of the last row)
    c->h_img_sz = (size_t)n*h*stride; OCK(hipHostMalloc((void **)&c->h_img, c->h_img_sz, hipHostMallocDefault));
    memcpy(c->h_img, imgs, c->h_img_sz);
    OCK(hipMemcpyAsync(img, c->h_img, c->h_img_sz, hipMemcpyHostToDevice, c->stream));
    if ((rc = oalloc(c, &D.pyr, (size_t)n*po)) || (rc = oalloc(c, &D.blur, (size_t)n*bo))) return rc;
    if ((rc = oalloc(c, &D.rtab, (size_t)std::max(rt, 4)))) return rc;
    if ((rc = oalloc(c, &D.cellkp, (size_t)n*cell0*CELL_CAP)) || (rc = oalloc(c, &D.cellcnt, (size_t)n*cell0))) return rc;
    if ((rc = oalloc(c, &D.cand, (size_t)n*c->nlevels*D.cand_cap*3))) return rc;
    if ((rc = oalloc(c, &D.nodes, (size_t)n*c->nlevels*D.node_cap*(sizeof(QNode)/sizeof(int)))) || (rc = oalloc(c, &D.pool, (size_t)n*c->nlevels*D.pool_cap)) ||
        (rc = oalloc(c, &D.snbuf, (size_t)n*c->nlevels*4*D.node_cap))) return rc;
    if ((rc = oalloc(c, &D.sel, (size_t)n*kp0*4)) || (rc = oalloc(c, &D.selcnt, (size_t)n*c->nlevels)) || (rc = oalloc(c, &D.qfallback, (size_t)n*c->nlevels)) || (rc = oalloc(c, &D.selab, (size_t)n*kp0*2))) return rc;
    {   // the three outputs in one allocation (kp | count | desc): one device-to-host copy per call
        const size_t bkp = sizeof(float)*(size_t)n*cap*6, bcnt = ((sizeof(int)*(size_t)n + 15)/16)*16, bdesc = (size_t)n*cap*32;
        uint8_t *ob; if ((rc = oalloc(c, &ob, bkp + bcnt + bdesc))) return rc;
        D.out_kp = (float *)ob; D.out_cnt = (int *)(ob + bkp); D.out_desc = ob + bkp + bcnt;
        c->h_out_sz = bkp + bcnt + bdesc; OCK(hipHostMalloc(&c->h_out, c->h_out_sz, hipHostMallocDefault));
        void *hd = nullptr; OCK(hipHostGetDevicePointer(&hd, c->h_out, 0));
        D.h_kp = (float *)hd; D.h_cnt = (int *)((uint8_t *)hd + bkp); D.h_desc = (uint8_t *)hd + bkp + bcnt;
    }
    c->key[0] = n; c->key[1] = w; c->key[2] = h; c->key[3] = stride; c->key[4] = cap;
    if (c->nlevels > 1) hipLaunchKernelGGL(k_resize_tab, dim3(c->nlevels - 1), dim3(256), 0, c->stream, D);
    c->uploaded = true; return TSORB_OK;
}
int tsorb_run(void *ctx) {
    OCtx *c = (OCtx *)ctx; if (!c || !c->uploaded) return TSORB_ERR_ARG;
    hipSetDevice(c->device);
    OrbDev &D = c->D;
    const bool few = c->pyr_shape == 1 || c->pyr_shape == 2 || (c->pyr_shape < 0 && D.n <= P1_MAX_N);
    auto pyr_launch = [&](const PyrOne &Q, int inst) { if (Q.per_frame <= 0) return;
        if (inst == 0) hipLaunchKernelGGL((k_pyramid_one<4096, 192, 256>), dim3(D.n*Q.per_frame), dim3(256), 0, c->stream, D, Q);
        else hipLaunchKernelGGL((k_pyramid_one<16384, 512, 512>), dim3(D.n*Q.per_frame), dim3(512), 0, c->stream, D, Q); };
    if (few && c->pyr_shape != 2 && c->q_inst[0] >= 0 && c->q_inst[1] >= 0) { pyr_launch(c->Q[0], c->q_inst[0]); if (c->Q[1].top > c->Q[1].base) pyr_launch(c->Q[1], c->q_inst[1]); }     // a few frames: two launches
    else if (few && c->q_inst[2] >= 0) pyr_launch(c->Q[2], c->q_inst[2]);                                                                                    // (or one)
    else if (c->pyr_shape == 3 && [&] { for (int q = 0; q < c->n_pairs; q++) if (c->q_inst[3 + q] < 0) return false; return true; }())
        for (int q = 0; q < c->n_pairs; q++) pyr_launch(c->Q[3 + q], c->q_inst[3 + q]);                                                                    // (experiment: two levels per launch, any batch)
    else {
        if (D.nlevels == 1) hipLaunchKernelGGL(k_level0, dim3((D.L[0].bw + 511)/512, (D.L[0].bh + L0_ROWS - 1)/L0_ROWS, D.n), dim3(128), 0, c->stream, D);
        for (int l = 1; l < D.nlevels; l++) { const int nyb0 = l == 1 ? (D.L[0].bh + L0_ROWS - 1)/L0_ROWS : 0;        // (level 0's copy rides in level 1's launch)
            hipLaunchKernelGGL(k_resize, dim3((std::max(D.L[l].bw, nyb0 ? D.L[0].bw : 0) + 4*RS_T - 1)/(4*RS_T), (D.L[l].bh + RS_ROWS - 1)/RS_ROWS + nyb0, D.n), dim3(RS_T), 0, c->stream, D, l, nyb0); }
    }
    const int fast_shape = c->fast_shape < 0 ? 2 : c->fast_shape;      // (diagnostics: 1 / 3 the small tile with 256 / 64 threads)
    if (D.fast_cells[0] > 0) {
        if (fast_shape == 1) hipLaunchKernelGGL((k_fast<40, 1024, 320, 256, 0>), dim3(D.n*D.fast_cells[0]), dim3(256), 0, c->stream, D);
        else if (fast_shape == 3) hipLaunchKernelGGL((k_fast<40, 1024, 320, 64, 0>), dim3(D.n*D.fast_cells[0]), dim3(64), 0, c->stream, D);
        else hipLaunchKernelGGL((k_fast<40, 1024, 320, 128, 0>), dim3(D.n*D.fast_cells[0]), dim3(128), 0, c->stream, D);
    }
    const bool blur_rides = few && D.fast_cells[1] > 0;                         // (on a batch the blur beside the general detector instance's few cells instead of beside the orientation: 0.372 against 0.369 ms, not kept)
    if (blur_rides) hipLaunchKernelGGL(k_fast_blur, dim3(D.n*D.fast_cells[1] + D.n*D.btiles_per_frame), dim3(256), 0, c->stream, D, D.n*D.fast_cells[1]);
    else if (D.fast_cells[1] > 0) hipLaunchKernelGGL((k_fast<TILE_MAX, 2048, CELL_CAP, 256, 1>), dim3(D.n*D.fast_cells[1]), dim3(256), 0, c->stream, D);
    *c->h_fb = 0;                                                                             // (the previous run has been waited for)
    hipLaunchKernelGGL(k_octree, dim3(D.n*D.nlevels), dim3(QT), 0, c->stream, D);
    if (!few && !blur_rides && c->merge_ob) { const int no = (D.n*D.slots_per_frame*16 + 255)/256;
        hipLaunchKernelGGL(k_orient_blur, dim3(no + D.n*D.btiles_per_frame), dim3(256), 0, c->stream, D, no); }
    else {
    if (!few) hipLaunchKernelGGL(k_orient, dim3((D.n*D.slots_per_frame*16 + 255)/256), dim3(256), 0, c->stream, D);
    if (!blur_rides) hipLaunchKernelGGL(k_blur, dim3(D.n*D.btiles_per_frame), dim3(256), 0, c->stream, D);
    }
    if (few) hipLaunchKernelGGL(k_orient_describe, dim3((D.n*D.slots_per_frame*32 + 255)/256), dim3(256), 0, c->stream, D);      // a few frames: orientation inside the descriptor launch
    else hipLaunchKernelGGL(k_describe, dim3((D.n*D.slots_per_frame*32 + 255)/256), dim3(256), 0, c->stream, D);
    OCK(hipStreamSynchronize(c->stream)); OCK(hipGetLastError());
    c->out_on_host = few;
    if (*(volatile int *)c->h_fb) {             // a level the LDS quadtree could not hold (counted as empty so far): the serial pass, then orientation and descriptors again with its keypoints in place
        c->fallbacks++;
        hipLaunchKernelGGL(k_octree_serial, dim3(D.n*D.nlevels), dim3(64), 0, c->stream, D);
        if (few) hipLaunchKernelGGL(k_orient_describe, dim3((D.n*D.slots_per_frame*32 + 255)/256), dim3(256), 0, c->stream, D);
        else { hipLaunchKernelGGL(k_orient, dim3((D.n*D.slots_per_frame*16 + 255)/256), dim3(256), 0, c->stream, D);
               hipLaunchKernelGGL(k_describe, dim3((D.n*D.slots_per_frame*32 + 255)/256), dim3(256), 0, c->stream, D); }
        OCK(hipStreamSynchronize(c->stream)); OCK(hipGetLastError());
    }
    return TSORB_OK;
}
int tsorb_download(void *ctx, float *kp, uint8_t *desc, int32_t *count) {
    OCtx *c = (OCtx *)ctx; if (!c || !c->uploaded) return TSORB_ERR_ARG;
    hipSetDevice(c->device); OrbDev &D = c->D;
    const size_t bkp = sizeof(float)*(size_t)D.n*D.cap*6, bcnt = ((sizeof(int)*(size_t)D.n + 15)/16)*16, bdesc = (size_t)D.n*D.cap*32;
    const size_t bytes = desc ? bkp + bcnt + bdesc : bkp + bcnt;
    if (!c->out_on_host) {                  // (a few frames: k_orient_describe stored them in h_out as well, and tsorb_run has waited for it)
        OCK(hipMemcpyAsync(c->h_out, D.out_kp, bytes, hipMemcpyDeviceToHost, c->stream));
        OCK(hipStreamSynchronize(c->stream));
    }
    const uint8_t *hb = (const uint8_t *)c->h_out;
    if (kp) memcpy(kp, hb, bkp);
    if (count) memcpy(count, hb + bkp, sizeof(int)*(size_t)D.n);
    if (desc) memcpy(desc, hb + bkp + bcnt, bdesc);
    return TSORB_OK;
}
int tsorb_extract_batch(void *ctx, const uint8_t *imgs, int n, int w, int h, int stride, float *kp, uint8_t *desc, int32_t *count, int cap) {
    int rc = tsorb_upload(ctx, imgs, n, w, h, stride, cap); if (rc) return rc;
    rc = tsorb_run(ctx); if (rc) return rc;
    return tsorb_download(ctx, kp, desc, count);
}
int tsorb_debug_fast_shape(void *ctx, int shape) { OCtx *c = (OCtx *)ctx; if (!c || shape < -1 || shape > 3) return TSORB_ERR_ARG; c->fast_shape = shape; c->key[0] = 0; return TSORB_OK; }      // (key: the next upload sets the geometry up again)
int tsorb_debug_fallbacks(void *ctx) { OCtx *c = (OCtx *)ctx; return c ? c->fallbacks : TSORB_ERR_ARG; }
int tsorb_debug_pyramid(void *ctx, int shape) { OCtx *c = (OCtx *)ctx; if (!c || shape < -1 || (shape > 3 && shape < 100) || (shape >= 100 + MAXL && shape != 200 && shape != 201)) return TSORB_ERR_ARG;
    if (shape == 200 || shape == 201) { c->merge_ob = shape - 200; return TSORB_OK; }
    if (shape >= 100) { c->pyr_split = shape - 100; c->key[0] = 0; } else c->pyr_shape = shape; return TSORB_OK; }      // (100 + s: the split level of the two launches, at the next upload)
#ifdef Q_STAMPS
int tsorb_debug_stamps(void *ctx, int32_t *out, int n) { OCtx *c = (OCtx *)ctx; if (!c || !c->uploaded) return TSORB_ERR_ARG; hipSetDevice(c->device);
    return hipMemcpy(out, c->D.snbuf, sizeof(int32_t)*(size_t)n, hipMemcpyDeviceToHost) == hipSuccess ? TSORB_OK : TSORB_ERR_DEVICE; }
#endif
int tsorb_debug_level(void *ctx, int frame, int level, int blurred, uint8_t *out, int32_t *w_out, int32_t *h_out) {
    OCtx *c = (OCtx *)ctx; if (!c || !c->uploaded || !out) return TSORB_ERR_ARG;
    OrbDev &D = c->D; if (frame < 0 || frame >= D.n || level < 0 || level >= D.nlevels) return TSORB_ERR_ARG;
    hipSetDevice(c->device);
    const LevelGeo &G = D.L[level];
    if (w_out) *w_out = G.w; if (h_out) *h_out = G.h;
    if (blurred) OCK(hipMemcpy(out, D.blur + (size_t)frame*D.blur_frame + G.blur_off, (size_t)G.w*G.h, hipMemcpyDeviceToHost));
    else OCK(hipMemcpy(out, D.pyr + (size_t)frame*D.pyr_frame + G.pyr_off, (size_t)G.bw*G.bh, hipMemcpyDeviceToHost));
    return TSORB_OK;
}

// ---- window / projection search
static int match_build(OCtx *c, const float *kp_dev, const uint8_t *desc_dev, int n, double min_x, double max_x, double min_y, double max_y) {
    if (!(max_x > min_x) || !(max_y > min_y) || n < 0) return TSORB_ERR_ARG;
    const size_t need = sizeof(int)*((size_t)2*std::max(n, 1) + MG_CELLS + 1);
    if (c->m_cap < need) { if (c->m_buf) hipFree(c->m_buf); c->m_cap = 0; OCK(hipMalloc(&c->m_buf, need)); c->m_cap = need; }
    MatchDev &M = c->M;
    M.kp = kp_dev; M.desc = desc_dev; M.n = n; M.min_x = min_x; M.min_y = min_y;
    M.iw = (double)MG_COLS/(max_x - min_x); M.ih = (double)MG_ROWS/(max_y - min_y);          // frame.cc:124-125
    M.off = (int *)c->m_buf; M.cell = M.off + MG_CELLS + 1; M.list = M.cell + std::max(n, 1);
    OCK(hipMemsetAsync(M.off, 0, sizeof(int)*(MG_CELLS + 1), c->stream));
    if (n > 0) hipLaunchKernelGGL(k_mg_cell, dim3((n + 255)/256), dim3(256), 0, c->stream, M);
    hipLaunchKernelGGL(k_mg_scan, dim3(1), dim3(1024), 0, c->stream, M);
    if (n > 0) hipLaunchKernelGGL(k_mg_place, dim3((n + 63)/64), dim3(256), 0, c->stream, M);
    c->m_set = true; return TSORB_OK;
}
int tsorb_match_set_frame(void *ctx, int frame, double min_x, double max_x, double min_y, double max_y) {
    OCtx *c = (OCtx *)ctx; if (!c || !c->uploaded) return TSORB_ERR_ARG;
    OrbDev &D = c->D; if (frame < 0 || frame >= D.n) return TSORB_ERR_ARG;
    hipSetDevice(c->device);
    int cnt = 0;
    if (c->out_on_host) cnt = ((const int *)((const uint8_t *)c->h_out + sizeof(float)*(size_t)D.n*D.cap*6))[frame];       // (a few frames: the run stored the counts in the pinned block and waited for it -- no copy, no synchronisation: 0.035 -> ~0.01 ms)
    else { OCK(hipMemcpyAsync(&cnt, D.out_cnt + frame, sizeof(int), hipMemcpyDeviceToHost, c->stream)); OCK(hipStreamSynchronize(c->stream)); }
    return match_build(c, D.out_kp + (size_t)frame*D.cap*6, D.out_desc + (size_t)frame*D.cap*32, cnt, min_x, max_x, min_y, max_y);
}
int tsorb_match_set_features(void *ctx, const float *kp6, const uint8_t *desc, int n, double min_x, double max_x, double min_y, double max_y) {
    OCtx *c = (OCtx *)ctx; if (!c || n < 0 || (n > 0 && (!kp6 || !desc))) return TSORB_ERR_ARG;
    hipSetDevice(c->device);
    const size_t bk = sizeof(float)*6*(size_t)std::max(n, 1), need = bk + 32*(size_t)std::max(n, 1);
    if (c->m_feat_cap < need) { if (c->m_feat) hipFree(c->m_feat); c->m_feat_cap = 0; OCK(hipMalloc(&c->m_feat, need)); c->m_feat_cap = need; }
    if (n > 0) { OCK(hipMemcpyAsync(c->m_feat, kp6, sizeof(float)*6*(size_t)n, hipMemcpyHostToDevice, c->stream));
                 OCK(hipMemcpyAsync((char *)c->m_feat + bk, desc, 32*(size_t)n, hipMemcpyHostToDevice, c->stream)); OCK(hipStreamSynchronize(c->stream)); }
    return match_build(c, (const float *)c->m_feat, (const uint8_t *)c->m_feat + bk, n, min_x, max_x, min_y, max_y);
}
int tsorb_match_search(void *ctx, int nq, const float *qxy, const float *qr, const int32_t *qlev, const uint8_t *qdesc, int max_cand,
                       int32_t *cand_idx, int32_t *cand_dist, int32_t *cand_cnt, int32_t *best_idx, int32_t *best_dist, int32_t *best_dist2) {
    OCtx *c = (OCtx *)ctx; if (!c || !c->m_set || nq < 0 || max_cand < 0 || (nq > 0 && (!qxy || !qr || !qdesc))) return TSORB_ERR_ARG;
    if (nq == 0) return TSORB_OK;
    hipSetDevice(c->device);
    // one pinned staging block in, one out: [qxy | qr | qlev | qdesc]  ->  [cand_idx | cand_dist | cnt | best | dist | dist2]
    const size_t b_xy = 8*(size_t)nq, b_r = 4*(size_t)nq, b_lev = 8*(size_t)nq, b_d = 32*(size_t)nq, in_sz = b_xy + b_r + b_lev + b_d;
    const size_t b_c = 4*(size_t)nq*max_cand, out_sz = 2*b_c + 16*(size_t)nq, tot = in_sz + out_sz;
    if (c->mq_cap < tot) { if (c->mq_host) hipHostFree(c->mq_host); c->mq_cap = 0;
        OCK(hipHostMalloc(&c->mq_host, tot, hipHostMallocDefault)); c->mq_cap = tot; }
    char *h = (char *)c->mq_host, *d = nullptr;
    memcpy(h, qxy, b_xy); memcpy(h + b_xy, qr, b_r);
    if (qlev) memcpy(h + b_xy + b_r, qlev, b_lev); else for (size_t k = 0; k < 2*(size_t)nq; k++) ((int32_t *)(h + b_xy + b_r))[k] = -1;
    memcpy(h + b_xy + b_r + b_lev, qdesc, b_d);
    // (round 6: the kernel reads the queries from, and stores the results into, the pinned block itself -- a wave reads its query's 52 bytes once, the results are a few KB (or
    // the candidate lists' 8 bytes per entry): two copies on the stream cost the call more than the bus does; d = the block's device address)
    { void *hd = nullptr; OCK(hipHostGetDevicePointer(&hd, c->mq_host, 0)); d = (char *)hd; }
    int *o_ci = (int *)(d + in_sz), *o_cd = o_ci + (size_t)nq*max_cand, *o_cnt = o_cd + (size_t)nq*max_cand, *o_bi = o_cnt + nq, *o_bd = o_bi + nq, *o_bd2 = o_bd + nq;
    hipLaunchKernelGGL(k_match, dim3((nq + 3)/4), dim3(256), 0, c->stream, c->M, nq, (const float *)d, (const float *)(d + b_xy), (const int *)(d + b_xy + b_r),
                       (const uint8_t *)(d + b_xy + b_r + b_lev), max_cand, o_ci, o_cd, o_cnt, o_bi, o_bd, o_bd2);
    OCK(hipStreamSynchronize(c->stream)); OCK(hipGetLastError());
    const char *ho = h + in_sz;
    if (cand_idx) memcpy(cand_idx, ho, b_c);
    if (cand_dist) memcpy(cand_dist, ho + b_c, b_c);
    const int32_t *tail = (const int32_t *)(ho + 2*b_c);
    if (cand_cnt) memcpy(cand_cnt, tail, 4*(size_t)nq);
    if (best_idx) memcpy(best_idx, tail + nq, 4*(size_t)nq);
    if (best_dist) memcpy(best_dist, tail + 2*(size_t)nq, 4*(size_t)nq);
    if (best_dist2) memcpy(best_dist2, tail + 3*(size_t)nq, 4*(size_t)nq);
    return TSORB_OK;
}

} // extern "C"
