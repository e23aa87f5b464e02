// Dense blocked Cholesky of a large reduced camera system (global BA: 6 n_free up to tens of thousands), multi-workgroup.
//
//   S (n x n, lower triangle used, leading dimension ld) and the right-hand side g held as row n of the same array.
//   Right-looking, block size NB = 96 (16 keyframes):
//     k_chol_diag    1 workgroup: LDS Cholesky of the NB x NB diagonal block
//     k_chol_panel   rows below (and the rhs row): X L^T = A  (one thread per row, L broadcast from LDS)
//     k_chol_update  trailing matrix -= panel panel^T on the matrix cores (v_mfma_f64_16x16x4_f64), 64x64 tiles per workgroup
//   then k_chol_backsub (1 workgroup) solves L^T x = y and scatters dp = -x to the pose order.
// The forward substitution rides along as the extra row, exactly as in the LDS solver (tsba_solve.h).
#pragma once

#define CH_NB 96
#define CH_T 256

// ---- diagonal block: in-LDS Cholesky (LL^T), NB x NB, 256 threads
__global__ __launch_bounds__(CH_T) void k_chol_diag(Work W, int j0) {
    LmState *st = W.st;
    if (st->done || st->step_fail) return;
    const int n = 6 * *W.nfree;
    if (j0 >= n) return;
    const int nb = min(CH_NB, n - j0);
    __shared__ double L[CH_NB*(CH_NB + 1)];
    __shared__ int bad;
    const int tid = threadIdx.x, ld = W.N, lds = CH_NB + 1;
    double *A = W.S;
    if (tid == 0) bad = 0;
    for (int k = tid; k < nb*nb; k += CH_T) { int r = k / nb, c = k - r*nb; L[r*lds + c] = (c <= r) ? A[(size_t)(j0 + r)*ld + j0 + c] : 0.0; }
    __syncthreads();
    for (int j = 0; j < nb; j++) {
        double d = L[j*lds + j];
        if (!(d > 0.0)) { if (tid == 0) bad = 1; d = 1.0; }
        const double s = sqrt(d), is = 1.0/s;
        __syncthreads();
        for (int r = j + tid; r < nb; r += CH_T) L[r*lds + j] = (r == j) ? s : L[r*lds + j]*is;
        __syncthreads();
        // rank-1 update of the trailing lower triangle
        const int m = nb - j - 1;
        for (int k = tid; k < m*m; k += CH_T) {
            int r = k / m, c = k - r*m;
            if (c <= r) L[(j + 1 + r)*lds + j + 1 + c] -= L[(j + 1 + r)*lds + j]*L[(j + 1 + c)*lds + j];
        }
        __syncthreads();
    }
    for (int k = tid; k < nb*nb; k += CH_T) { int r = k / nb, c = k - r*nb; if (c <= r) A[(size_t)(j0 + r)*ld + j0 + c] = L[r*lds + c]; }
    if (tid == 0 && bad) st->step_fail = 1;
}

// ---- panel: rows i in (j0+nb .. n] (row n = rhs): x L^T = a.  One thread per row; the row is staged in LDS (stride 97 keeps
// the lanes on distinct banks), L is held packed-lower in LDS and broadcast.
#define CH_PT 128
__global__ __launch_bounds__(CH_PT) void k_chol_panel(Work W, int j0) {
    LmState *st = W.st;
    if (st->done || st->step_fail) return;
    const int n = 6 * *W.nfree;
    if (j0 >= n) return;
    const int nb = min(CH_NB, n - j0);
    const int i0 = j0 + nb + blockIdx.x*CH_PT;
    if (i0 > n) return;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *L = sm;                                 // packed lower, nb(nb+1)/2
    double *X = sm + CH_NB*(CH_NB + 1)/2;           // [CH_PT][CH_NB + 1]
    const int tid = threadIdx.x, ld = W.N, xs = CH_NB + 1;
    double *A = W.S;
    for (int k = tid; k < nb*nb; k += CH_PT) { int r = k / nb, c = k - r*nb; if (c <= r) L[r*(r + 1)/2 + c] = A[(size_t)(j0 + r)*ld + j0 + c]; }
    const int nrow = min(CH_PT, n + 1 - i0);
    for (int k = tid; k < nrow*nb; k += CH_PT) { int r = k / nb, c = k - r*nb; X[r*xs + c] = A[(size_t)(i0 + r)*ld + j0 + c]; }
    __syncthreads();
    if (tid < nrow) {
        double *x = X + tid*xs;
        for (int c = 0; c < nb; c++) {
            const double *lc = L + c*(c + 1)/2;
            double v = x[c];
            for (int k = 0; k < c; k++) v -= x[k]*lc[k];
            x[c] = v/lc[c];
        }
    }
    __syncthreads();
    for (int k = tid; k < nrow*nb; k += CH_PT) { int r = k / nb, c = k - r*nb; A[(size_t)(i0 + r)*ld + j0 + c] = X[r*xs + c]; }
}

// ---- trailing update: C[i][k] -= sum_c P[i][c] P[k][c]  for i,k >= c0 (lower triangle, 64x64 tiles) and the rhs row.
// Workgroup = 4 waves, each a 32x32 quadrant = 2x2 MFMA 16x16 tiles; the two 64 x nb panels are staged through LDS.
typedef double v4d_c __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(CH_T) void k_chol_update(Work W, int j0) {
    LmState *st = W.st;
    if (st->done || st->step_fail) return;
    const int n = 6 * *W.nfree;
    if (j0 >= n) return;
    const int nb = min(CH_NB, n - j0);
    const int c0 = j0 + nb;
    if (c0 >= n + 1) return;
    // tile (ti, tj), tj <= ti over the (n + 1 - c0) rows x (n - c0) columns
    const int ntile = (n + 1 - c0 + 63)/64;
    int t = blockIdx.x, ti = (int)((sqrtf(8.0f*(float)t + 1.0f) - 1.0f)*0.5f);
    while (ti*(ti + 1)/2 > t) ti--;
    while ((ti + 1)*(ti + 2)/2 <= t) ti++;
    const int tj = t - ti*(ti + 1)/2;
    if (ti >= ntile) return;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Pa = sm, *Pb = sm + 64*(CH_NB + 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ld = W.N, lds = CH_NB + 1;
    double *A = W.S;
    const int r0 = c0 + 64*ti, q0 = c0 + 64*tj;
    const int nbp = (nb + 3) & ~3;                              // K padded to the MFMA depth with exact zeros
    for (int k = tid; k < 64*nbp; k += CH_T) {
        int r = k / nbp, c = k - r*nbp;
        Pa[r*lds + c] = (c < nb && r0 + r <= n) ? A[(size_t)(r0 + r)*ld + j0 + c] : 0.0;
        Pb[r*lds + c] = (c < nb && q0 + r < n) ? A[(size_t)(q0 + r)*ld + j0 + c] : 0.0;
    }
    __syncthreads();
    const int wr = (wave >> 1)*32, wc = (wave & 1)*32;          // quadrant of this wave
    const int lr = lane & 15, lk = lane >> 4;
    v4d_c acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (v4d_c){0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < nb; k += 4) {
        double av[2], bv[2];
#pragma unroll
        for (int a = 0; a < 2; a++) av[a] = Pa[(wr + 16*a + lr)*lds + k + lk];
#pragma unroll
        for (int b = 0; b < 2; b++) bv[b] = Pb[(wc + 16*b + lr)*lds + k + lk];
#pragma unroll
        for (int a = 0; a < 2; a++)
#pragma unroll
            for (int b = 0; b < 2; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = r0 + wr + 16*a + lk + 4*r, col = q0 + wc + 16*b + lr;
                if (row <= n && col < n && (col <= row)) A[(size_t)row*ld + col] -= acc[a][b][r];
            }
}

// ---- back substitution L^T x = y (y = row n), one workgroup; dp[6a + k] = -x[6 fidx[a] + k]
__global__ __launch_bounds__(1024) void k_chol_backsub(Work W) {
    LmState *st = W.st;
    if (st->done) return;
    const int n = 6 * *W.nfree;
    const int tid = threadIdx.x, ld = W.N;
    double *A = W.S, *y = W.S + (size_t)n*ld;
    if (st->step_fail) { for (int k = tid; k < W.N; k += 1024) W.dp[k] = 0.0; return; }
    __shared__ double xs[CH_NB];
    for (int j1 = n; j1 > 0; j1 -= CH_NB) {
        const int j0 = max(0, j1 - CH_NB), nb = j1 - j0;
        // triangular solve of the diagonal block by wave 0: lane c owns y[j0 + c] (two per lane when nb > 64)
        if (tid < 64) {
            double y0 = tid < nb ? y[j0 + tid] : 0.0, y1 = tid + 64 < nb ? y[j0 + tid + 64] : 0.0;
            for (int c = nb - 1; c >= 0; c--) {
                const double lcc = A[(size_t)(j0 + c)*ld + j0 + c];
                double yc = c < 64 ? readlane_f64(y0, c) : readlane_f64(y1, c - 64);
                const double xc = yc/lcc;
                if (tid == (c & 63)) { if (c < 64) y0 = xc; else y1 = xc; }
                if (tid < c) y0 -= A[(size_t)(j0 + c)*ld + j0 + tid]*xc;
                if (tid + 64 < c) y1 -= A[(size_t)(j0 + c)*ld + j0 + tid + 64]*xc;
            }
            if (tid < nb) { xs[tid] = y0; y[j0 + tid] = y0; }
            if (tid + 64 < nb) { xs[tid + 64] = y1; y[j0 + tid + 64] = y1; }
        }
        __syncthreads();
        for (int k = tid; k < j0; k += 1024) {
            double v = 0.0;
            for (int c = 0; c < nb; c++) v += A[(size_t)(j0 + c)*ld + k]*xs[c];
            y[k] -= v;
        }
        __syncthreads();
    }
    for (int a = tid; a < W.n_kf; a += 1024) {
        int ia = W.fidx[a];
        for (int k = 0; k < 6; k++) W.dp[6*a + k] = ia >= 0 ? -y[6*ia + k] : 0.0;
    }
}

// copies g into row n of S (the extra rhs row) -- the LDS solver does this itself
__global__ void k_chol_rhs(Work W) {
    if (W.st->done) return;
    const int n = 6 * *W.nfree;
    int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k < n) W.S[(size_t)n*W.N + k] = W.g[k];
}
