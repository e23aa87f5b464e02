// Separator system of the partitioned band solver (tsba_bandp.h) by block cyclic reduction.
//
// The separator system is block tridiagonal: m = P - 1 diagonal blocks D_i of s = 6 B rows and the couplings C_i = S(i+1, i) the interiors
// leave between neighbouring separators.  k_band_solve walks it block after block on ONE workgroup (845 us per LM iteration at 5000
// keyframes, as long as all the interiors together); cyclic reduction eliminates every second separator of a level at once:
//   level h = 1, 2, 4, ...: pivots i = (2k + 1) h, neighbours a = i - h, c = i + h (if < m)
//     k_cr_pivot   D_i = L L^T (in LDS), W_a = L^-1 S(i, a), W_c = L^-1 S(c, i)^T, y_i = L^-1 g_i -- all in place (L over D_i, W_a over S(i, a),
//                  W_c^T over S(c, i), y over g)
//     k_cr_update  every remaining block e: D_e -= W^T W of the pivots next to it, g_e -= W^T y; every pivot: S(c, a) = -W_c^T W_a (the new
//                  coupling of the next level: a fresh block of the compact pool, cr_blk in tsba_bandp.h) -- 16x16 MFMA tiles
//   root: block 0 alone; then back substitution level by level: x_i = L^-T (y_i - W_a x_a - W_c x_c)          (k_cr_back)
// log2(m) levels of small dense kernels on m / 2h workgroups instead of m B sequential pose-block steps.  The number of separators is only
// known on the device (bandp_part): the host launches the worst case, workgroups without a pivot return.
#pragma once

#define CR_T 256
#define CR_SMAX 78                           // separator rows (13 pose blocks): L + all right-hand sides (pivot), three blocks (back) fit the LDS

static size_t cr_pivot_lds_doubles(int s) { return (size_t)s*(s + 1) + (size_t)s*(2*s + 1) + s; }
static size_t cr_back_lds_doubles(int s) { return 3*(size_t)s*(s + 1) + 5*(size_t)s; }

// separator labels of the system: lo .. m - 1, root r0 (ring maps: merged with label m - 1, the ghost of the loop's first separator)
struct CrRange { int lo, m, r0; };
__device__ __forceinline__ CrRange cr_range(const Work &W, int bw, int Pmax) {
    CrRange r = {0, 0, 0};
    const int nf = *W.nfree; if (nf <= 0) return r;
    const BandpPart P0 = bandp_part_w(W, bw/6, Pmax, 0);
    if (W.ring) { const int off = P0.Pt > 0 ? RING_OFF : 0; r.r0 = off; r.lo = P0.Pt > 0 ? off - P0.Pt + 1 : off; r.m = off + P0.G + 1; }
    else r.m = P0.P - 1;
    return r;
}
__device__ __forceinline__ int cr_nsep(const Work &W, int bw, int Pmax) { return cr_range(W, bw, Pmax).m; }      // (chain: the number of separators)

// n elements through f(index) -> value and st(index, value), 16 per thread in flight (a plain copy loop waits for every load before it
// issues the next: ~0.6 us each)
template <int U = 16, class F, class G>
__device__ __forceinline__ void cr_batched(int n, int tid, F f, G st) {
    for (int e0 = tid; e0 < n; e0 += U*CR_T) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int e = e0 + u*CR_T; v[u] = e < n ? f(e) : 0.0; }
#pragma unroll
        for (int u = 0; u < U; u++) { const int e = e0 + u*CR_T; if (e < n) st(e, v[u]); }
    }
}

// root = 1: the last remaining block (index 0), no neighbours
__global__ __launch_bounds__(CR_T) void k_cr_pivot(Work W, Work Ws, int bw, int Pmax, int h, int root) {
    LmState *st = W.st;
    if (st->done || st->step_fail) return;
    const int m = cr_nsep(W, bw, Pmax), s = bw;
    int i, a, c;
    if (root) { if (blockIdx.x > 0 || m <= 0) return; i = 0; a = -1; c = -1; }
    else { i = (2*(int)blockIdx.x + 1)*h; if (i >= m) return; a = i - h; c = i + h < m ? i + h : -1; }
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int ncol = 2*s + 1, tid = threadIdx.x, mmax = Pmax - 1;
    double *Ls = smem, *R = smem + (size_t)s*(s + 1), *invd = R + (size_t)s*ncol;   // invd: 1 / L(r, r) (a division is ~35 instructions: once per row, in parallel)
    // R[r][t]: columns [0, s) S(i, a)(:, t), [s, 2s) S(c, i)(t - s, :)^T, 2s: g_i
    double *g = Ws.g;
    double *Bii = cr_blk(Ws.S, s, mmax, i, i), *Bia = a >= 0 ? cr_blk(Ws.S, s, mmax, i, a) : nullptr, *Bci = c >= 0 ? cr_blk(Ws.S, s, mmax, c, i) : nullptr;
    // (e -> (row, column) by a float reciprocal: an integer division by the run-time s costs ~30 instructions, twice per element)
    const float inv_s = 1.0f/(float)s;
    auto rowof = [&](int e) { return (int)(((float)e + 0.5f)*inv_s); };
    cr_batched(s*s, tid, [&](int e) { const int r = rowof(e), q = e - r*s; return q <= r ? Bii[e] : 0.0; },
               [&](int e, double v) { const int r = rowof(e), q = e - r*s; Ls[r*(s + 1) + q] = v; });
    if (a >= 0) cr_batched(s*s, tid, [&](int e) { return Bia[e]; },
                           [&](int e, double v) { const int r = rowof(e), t = e - r*s; R[r*ncol + t] = v; });
    if (c >= 0) cr_batched(s*s, tid, [&](int e) { return Bci[e]; },
                           [&](int e, double v) { const int j = rowof(e), r = e - j*s; R[r*ncol + s + j] = v; });
    for (int r = tid; r < s; r += CR_T) R[r*ncol + 2*s] = g[i*s + r];
    // blocked (6 columns = one pose) right-looking Cholesky in LDS: every thread factors the 6x6 diagonal block in registers (no
    // broadcast), the threads of the rows below solve their panel row, then the rank-6 trailing update: two barriers per block
    for (int k0 = 0; k0 < s; k0 += 6) {
        __syncthreads();
        double d[21], id[6];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int q = 0; q <= r; q++) d[tri(r) + q] = Ls[(k0 + r)*(s + 1) + k0 + q];
        bool bad = false;
#pragma unroll
        for (int q = 0; q < 6; q++) {                             // d -> its Cholesky factor (row-packed lower)
            double dq = d[tri(q) + q];
#pragma unroll
            for (int k = 0; k < q; k++) dq -= d[tri(q) + k]*d[tri(q) + k];
            if (!(dq > 0.0)) { bad = true; dq = 1.0; }
            double inv = __builtin_amdgcn_rsq(dq);                 // 1 / sqrt: hardware estimate + two Newton steps (a division is ~35 instructions)
            inv = inv*fma(-0.5*dq*inv, inv, 1.5); inv = inv*fma(-0.5*dq*inv, inv, 1.5);
            const double lq = dq*inv;
            d[tri(q) + q] = lq; id[q] = inv;
#pragma unroll
            for (int r = q + 1; r < 6; r++) { double v = d[tri(r) + q];
#pragma unroll
                for (int k = 0; k < q; k++) v -= d[tri(r) + k]*d[tri(q) + k];
                d[tri(r) + q] = v*inv; }
        }
        if (bad) { if (tid == 0) st->step_fail = 1; return; }     // (uniform) not positive definite: an invalid LM step
        __syncthreads();                                          // everybody has read the diagonal block
        for (int r = k0 + tid; r < s; r += CR_T) {
            double *row = Ls + r*(s + 1) + k0;
            if (r < k0 + 6) {
#pragma unroll
                for (int rr = 0; rr < 6; rr++) if (r == k0 + rr) {       // (constant indices: d stays in registers)
#pragma unroll
                    for (int q = 0; q <= rr; q++) row[q] = d[tri(rr) + q]; }
            }
            else {
                double x[6];
#pragma unroll
                for (int q = 0; q < 6; q++) { double v = row[q];
#pragma unroll
                    for (int k = 0; k < q; k++) v -= x[k]*d[tri(q) + k];
                    x[q] = v*id[q]; }
#pragma unroll
                for (int q = 0; q < 6; q++) row[q] = x[q];
            }
        }
        __syncthreads();
        const int nt = s - k0 - 6;
        for (int e0 = tid; e0 < tri(nt); e0 += 3*CR_T) {           // three elements per thread in flight (the reads of one are ~13 LDS round trips)
            double xv[3][6], yv[3][6], cv[3]; int ci[3];
#pragma unroll
            for (int u = 0; u < 3; u++) { const int e = min(e0 + u*CR_T, tri(nt) - 1), rr = tri_row(e), cc = e - tri(rr);
                const double *x = Ls + (k0 + 6 + rr)*(s + 1) + k0, *y = Ls + (k0 + 6 + cc)*(s + 1) + k0;
                ci[u] = (k0 + 6 + rr)*(s + 1) + k0 + 6 + cc; cv[u] = Ls[ci[u]];
#pragma unroll
                for (int q = 0; q < 6; q++) { xv[u][q] = x[q]; yv[u][q] = y[q]; } }
#pragma unroll
            for (int u = 0; u < 3; u++) if (e0 + u*CR_T < tri(nt))
                Ls[ci[u]] = cv[u] - ((xv[u][0]*yv[u][0] + xv[u][1]*yv[u][1] + xv[u][2]*yv[u][2]) + (xv[u][3]*yv[u][3] + xv[u][4]*yv[u][4] + xv[u][5]*yv[u][5]));
        }
    }
    __syncthreads();
    for (int r = tid; r < s; r += CR_T) invd[r] = 1.0/Ls[r*(s + 1) + r];
    cr_batched(s*s, tid, [&](int e) { const int r = rowof(e), q = e - r*s; return Ls[r*(s + 1) + q]; },
               [&](int e, double v) { const int r = rowof(e), q = e - r*s; if (q <= r) Bii[e] = v; });
    // forward substitution L^-1 [S(i, a) | S(c, i)^T | g], blocked by 6 rows: the columns' 6x6 triangular solves (one column per thread),
    // then the rank-6 update of the rows below on all threads (wave -> column half and row parity: the six solved values of the column
    // stay in registers, the L row is a broadcast).  (One column per thread over all 60 rows was a 1770-step chain of dependent LDS
    // reads on two waves: 35 us.)
    {
        // (up to 128 columns: two waves per column half, the rows below split by parity; more -- separators of 11+ pose blocks --: one column per thread)
        const int lane = tid & 63, wave = tid >> 6; const bool wide = ncol > 128;
        const int col = wide ? tid : (wave & 1)*64 + lane, rg = wide ? 0 : wave >> 1, rstep = wide ? 1 : 2;
        const bool con = col < ncol && (col == 2*s || (col < s ? a >= 0 : c >= 0));
        for (int kb = 0; kb < s; kb += 6) {
            __syncthreads();
            double x[6] = {0, 0, 0, 0, 0, 0};
            if (con) {
                double l6[21], rv[6];                              // operands first, then the dependent chain
#pragma unroll
                for (int q = 0; q < 6; q++) { rv[q] = R[(kb + q)*ncol + col];
#pragma unroll
                    for (int k = 0; k <= q; k++) l6[tri(q) + k] = Ls[(kb + q)*(s + 1) + kb + k]; }
#pragma unroll
                for (int q = 0; q < 6; q++) { double v = rv[q];
#pragma unroll
                    for (int k = 0; k < q; k++) v -= l6[tri(q) + k]*x[k];
                    x[q] = v*invd[kb + q]; }
            }
            __syncthreads();                                       // both row groups have read the block's rows (solved redundantly); one stores them
            if (con) {
                if (rg == 0) {
#pragma unroll
                    for (int q = 0; q < 6; q++) R[(kb + q)*ncol + col] = x[q];
                }
                for (int r0 = kb + 6 + rg; r0 < s; r0 += 4*rstep) {    // four rows per round in flight
                    double lv[4][6], cv[4];
#pragma unroll
                    for (int u = 0; u < 4; u++) { const int r = min(r0 + rstep*u, s - 1); const double *lr = Ls + r*(s + 1) + kb; cv[u] = R[r*ncol + col];
#pragma unroll
                        for (int q = 0; q < 6; q++) lv[u][q] = lr[q]; }
#pragma unroll
                    for (int u = 0; u < 4; u++) if (r0 + rstep*u < s)
                        R[(r0 + rstep*u)*ncol + col] = cv[u] - ((lv[u][0]*x[0] + lv[u][1]*x[1] + lv[u][2]*x[2]) + (lv[u][3]*x[3] + lv[u][4]*x[4] + lv[u][5]*x[5]));
                }
            }
        }
    }
    __syncthreads();
    if (a >= 0) cr_batched(s*s, tid, [&](int e) { const int r = rowof(e), t = e - r*s; return R[r*ncol + t]; },
                           [&](int e, double v) { Bia[e] = v; });
    if (c >= 0) cr_batched(s*s, tid, [&](int e) { const int j = rowof(e), r = e - j*s; return R[r*ncol + s + j]; },
                           [&](int e, double v) { Bci[e] = v; });
    for (int r = tid; r < s; r += CR_T) g[i*s + r] = R[r*ncol + 2*s];
}

// grid: [0, npiv) the new couplings S(c, a) of the pivots, [npiv, npiv + nev) the remaining blocks e = 2 k h.
// All three products have the form C -= P Q^T (K = s) once W_a is held transposed: 16x16 tiles on the matrix cores (v_mfma_f64_16x16x4),
// operands [row][k] in LDS with an odd row stride (conflict-free), rows and K padded with zeros to multiples of 16 / 4.
__device__ __forceinline__ void cr_tile_acc(const double *Pm, const double *Qm, int stride, int kn, int ti, int tj, int lr, int lk, v4d &c) {
    const double *pa = Pm + (size_t)(16*ti + lr)*stride + lk, *pb = Qm + (size_t)(16*tj + lr)*stride + lk;
    for (int k0 = 0; k0 < kn; k0 += 4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k0], pb[k0], c, 0, 0, 0);
}
static size_t cr_update_lds_doubles(int s) { const int rows = (s + 15) & ~15, stride = ((s + 3) & ~3) + 1; return 2*(size_t)rows*stride; }
__global__ __launch_bounds__(CR_T) void k_cr_update(Work W, Work Ws, int bw, int Pmax, int h, int npiv) {
    const LmState *st = W.st;
    if (st->done || st->step_fail) return;
    const int m = cr_nsep(W, bw, Pmax), s = bw, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, mmax = Pmax - 1;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int rows = (s + 15) & ~15, kn = (s + 3) & ~3, stride = kn + 1, nt = rows >> 4;
    double *X = smem, *Yt = smem + (size_t)rows*stride;
    const float inv_s = 1.0f/(float)s;
    double *S = Ws.S, *g = Ws.g;
    for (int e = tid; e < 2*rows*stride; e += CR_T) smem[e] = 0.0;
    __syncthreads();
    auto load_block = [&](double *dst, int br, int bc, bool transpose) {
        const double *Bk = cr_blk(S, s, mmax, br, bc);
        cr_batched(s*s, tid, [&](int e) { return Bk[e]; },
                   [&](int e, double v) { const int r = (int)(((float)e + 0.5f)*inv_s), q = e - r*s; if (transpose) dst[q*stride + r] = v; else dst[r*stride + q] = v; }); };
    const int lr = lane & 15, lk = lane >> 4;
    if ((int)blockIdx.x < npiv) {
        const int i = (2*(int)blockIdx.x + 1)*h, a = i - h, c = i + h;
        if (i >= m || c >= m) return;
        load_block(X, c, i, false); load_block(Yt, i, a, true);   // X = W_c^T (rows of c), Yt = W_a^T:  S(c, a) = -X Yt^T
        double *Bca = cr_blk(S, s, mmax, c, a);
        __syncthreads();
        for (int t = wave; t < nt*nt; t += CR_T/64) {
            const int ti = t/nt, tj = t - ti*nt;
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            cr_tile_acc(X, Yt, stride, kn, ti, tj, lr, lk, acc);
#pragma unroll
            for (int r = 0; r < 4; r++) { const int j = 16*ti + lk + 4*r, q = 16*tj + lr; if (j < s && q < s) Bca[j*s + q] = -acc[r]; }        // (a fresh block: nothing to add to)
        }
        return;
    }
    const int e0 = 2*((int)blockIdx.x - npiv)*h;                  // remaining block
    if (e0 >= m) return;
    const int il = e0 - h, ir = e0 + h < m ? e0 + h : -1;
    const bool hl = il >= 0, hr = ir >= 0;
    if (!hl && !hr) return;
    double *Bee = cr_blk(S, s, mmax, e0, e0);
    if (hl) load_block(X, e0, il, false);                         // W_c^T of the left pivot:  D -= X X^T,    g -= X y_l
    if (hr) load_block(Yt, ir, e0, true);                         // W_a^T of the right pivot: D -= Yt Yt^T,  g -= Yt y_r
    __syncthreads();
    for (int t = wave; t < nt*nt; t += CR_T/64) {
        const int ti = t/nt, tj = t - ti*nt;
        if (tj > ti) continue;                                    // lower triangle
        v4d acc = {0.0, 0.0, 0.0, 0.0};
        if (hl) cr_tile_acc(X, X, stride, kn, ti, tj, lr, lk, acc);
        if (hr) cr_tile_acc(Yt, Yt, stride, kn, ti, tj, lr, lk, acc);
#pragma unroll
        for (int r = 0; r < 4; r++) { const int j = 16*ti + lk + 4*r, q = 16*tj + lr; if (j < s && q <= j) Bee[j*s + q] -= acc[r]; }
    }
    for (int j = tid; j < s; j += CR_T) { double acc = 0.0;
        if (hl) for (int r = 0; r < s; r++) acc += X[j*stride + r]*g[il*s + r];
        if (hr) for (int r = 0; r < s; r++) acc += Yt[j*stride + r]*g[ir*s + r];
        g[e0*s + j] -= acc; }
}

// x_i = L^-T (y_i - W_a x_a - W_c x_c) -> Ws.Sy
__global__ __launch_bounds__(CR_T) void k_cr_back(Work W, Work Ws, int bw, int Pmax, int h, int root) {
    const LmState *st = W.st;
    if (st->done || st->step_fail) return;
    const int m = cr_nsep(W, bw, Pmax), s = bw, tid = threadIdx.x, lane = tid & 63, mmax = Pmax - 1;
    int i, a, c;
    if (root) { if (blockIdx.x > 0 || m <= 0) return; i = 0; a = -1; c = -1; }
    else { i = (2*(int)blockIdx.x + 1)*h; if (i >= m) return; a = i - h; c = i + h < m ? i + h : -1; }
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *Ls = smem, *X = Ls + (size_t)s*(s + 1), *Y = X + (size_t)s*(s + 1), *t = Y + (size_t)s*(s + 1), *xn = t + s, *xa = xn + s, *xc = xa + s, *invd = xc + s;
    const double *S = Ws.S, *g = Ws.g; double *x = Ws.Sy;
    const float inv_s = 1.0f/(float)s;
    auto load_block = [&](double *dst, int br, int bc) {
        const double *Bk = cr_blk(S, s, mmax, br, bc);
        cr_batched(s*s, tid, [&](int e) { return Bk[e]; },
                   [&](int e, double v) { const int r = (int)(((float)e + 0.5f)*inv_s), q = e - r*s; dst[r*(s + 1) + q] = v; }); };
    load_block(Ls, i, i);
    if (a >= 0) load_block(X, i, a);                              // W_a
    if (c >= 0) load_block(Y, c, i);                              // W_c^T
    for (int r = tid; r < s; r += CR_T) { t[r] = g[i*s + r]; xa[r] = a >= 0 ? x[a*s + r] : 0.0; xc[r] = c >= 0 ? x[c*s + r] : 0.0; }
    __syncthreads();
    for (int r = tid; r < s; r += CR_T) { double acc = t[r];
        if (a >= 0) for (int j = 0; j < s; j++) acc -= X[r*(s + 1) + j]*xa[j];
        if (c >= 0) for (int j = 0; j < s; j++) acc -= Y[j*(s + 1) + r]*xc[j];
        t[r] = acc; invd[r] = 1.0/Ls[r*(s + 1) + r]; }
    __syncthreads();
    if (tid >= 64) return;
    // one wave, right-looking: lane owns rows lane and lane + 64
    double t0 = lane < s ? t[lane] : 0.0, t1 = lane + 64 < s ? t[lane + 64] : 0.0;
    for (int r = s - 1; r >= 0; r--) {
        const double tr = r < 64 ? readlane_f64(t0, r) : readlane_f64(t1, r - 64);
        const double xr = tr*invd[r];
        if (lane == 0) xn[r] = xr;
        if (lane < r) t0 -= Ls[r*(s + 1) + lane]*xr;
        if (lane + 64 < r) t1 -= Ls[r*(s + 1) + lane + 64]*xr;
    }
    wave_lds_fence();
    if (lane < s) x[i*s + lane] = xn[lane];
    if (lane + 64 < s) x[i*s + lane + 64] = xn[lane + 64];
}
