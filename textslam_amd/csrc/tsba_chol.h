// Blocked Cholesky of a large reduced camera system (global BA: 6 n_free up to tens of thousands), multi-workgroup,
// profile-aware: the reduced camera matrix of a SLAM map is (block-)banded -- a keyframe is coupled to the keyframes it shares
// landmarks with -- and Cholesky without pivoting keeps the band.  The host knows an upper bound `bw` of the number of rows
// below a pose block that can be non-zero (tsba_plan.h: envelope of the S-block list, fill included); every step touches only
// those rows plus the right-hand-side row, so a 1000-keyframe map with a 24-keyframe band costs ~0.1 GFLOP instead of 72.
//
//   S (n x n) in band storage -- S(i,j) = S[i*ldS + j] over a skewed view whose rows overlap outside the band -- and the
//   right-hand side g as an extra row W.Sy (forward substitution rides along).
//   Right-looking, block size NB = 96 (16 keyframes), per block column:
//     k_solve_t<true>  1 workgroup: LDL^T of the NB x NB diagonal block in LDS (tsba_solve.h), written back as L D^1/2 (lower),
//                      and its inverse transposed (strict upper triangle + W.LDbuf for the diagonal)
//     k_chol_panel     band rows below + the rhs row:  X = A W^T  on the matrix cores (the inverse turns the triangular solve into
//                      a GEMM), 64 rows per workgroup
//     k_chol_update    band window -= panel panel^T (v_mfma_f64_16x16x4_f64), 64x64 tiles per workgroup; local row `wr` of the
//                      window stands for the rhs row
//   then k_chol_backsub (1 workgroup): per block x = W^T y (matvec with the stored inverse), band update of y; dp = -x.
#pragma once

#define CH_NB 96
#define CH_T 256

// ---- panel: X = A W^T for 64 rows per workgroup.  W (NB x NB lower, inverse of the diagonal factor) is rebuilt in LDS from
// the strict upper triangle of the diagonal block + LDbuf; A rows are staged in LDS; 4 waves x (16 rows x 96 columns).
typedef double v4d_c __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(CH_T) void k_chol_panel(Work W, int j0, int bw) {
    LmState *st = W.st;
    if (st->done || st->step_fail) return;
    const int n = 6 * *W.nfree;
    if (j0 >= n) return;
    const int nb = min(CH_NB, n - j0), c0 = j0 + nb;
    const int wr = min(bw, n - c0);                             // band rows below the block (all < n); local row wr = rhs row n
    const int R0 = blockIdx.x*64;
    if (R0 > wr) return;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const int lds = CH_NB + 1;
    double *Wl = sm, *X = sm + CH_NB*lds;                       // W [96][97] (zeros above the diagonal), A rows [64][97]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ld = W.ldS;
    double *A = W.S;
    const int nbp = (nb + 3) & ~3;
    for (int k = tid; k < CH_NB*CH_NB; k += CH_T) {
        const int c = k / CH_NB, r = k - c*CH_NB;               // W[r][c], read along r (contiguous in the upper triangle's row c)
        double v = 0.0;
        if (r < nb && c < nb) v = r > c ? A[(size_t)(j0 + c)*ld + j0 + r] : (r == c ? W.LDbuf[j0 + r] : 0.0);
        Wl[r*lds + c] = v;
    }
    for (int k = tid; k < 64*nbp; k += CH_T) {
        const int r = k / nbp, c = k - r*nbp, rl = R0 + r;
        const int gi = rl < wr ? c0 + rl : n;
        X[r*lds + c] = (c < nb && rl <= wr) ? (gi == n ? W.Sy : A + (size_t)gi*ld)[j0 + c] : 0.0;
    }
    __syncthreads();
    // wave w: rows 16w..16w+15, all 6 column tiles; X[i][c] = sum_{k <= c} A[i][k] W[c][k]
    const int lr = lane & 15, lk = lane >> 4;
    v4d_c acc[6];
#pragma unroll
    for (int t = 0; t < 6; t++) acc[t] = (v4d_c){0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < nbp; k += 4) {
        const double av = X[(16*wave + lr)*lds + k + lk];
#pragma unroll
        for (int t = 0; t < 6; t++) {
            if (16*t + 15 < k) continue;                        // W[c][k] = 0 for c < k: tile entirely above the diagonal
            const double bv = Wl[(16*t + lr)*lds + k + lk];
            acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < 6; t++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int rl = R0 + 16*wave + lk + 4*r, c = 16*t + lr;
            if (rl <= wr && c < nb) { const int gi = rl < wr ? c0 + rl : n; (gi == n ? W.Sy : A + (size_t)gi*ld)[j0 + c] = acc[t][r]; }
        }
}

// ---- trailing update of the band window: C[i][k] -= sum_c P[i][c] P[k][c], window rows c0 .. c0+wr-1 and (local row wr) the rhs row.
// Workgroup = 4 waves, each a 32x32 quadrant = 2x2 MFMA 16x16 tiles; the two 64 x nb panels are staged through LDS.
__global__ __launch_bounds__(CH_T) void k_chol_update(Work W, int j0, int bw) {
    LmState *st = W.st;
    if (st->done || st->step_fail) return;
    const int n = 6 * *W.nfree;
    if (j0 >= n) return;
    const int nb = min(CH_NB, n - j0);
    const int c0 = j0 + nb;
    if (c0 >= n + 1) return;
    const int wr = min(bw, n - c0);
    // tile (ti, tj), tj <= ti over local rows 0..wr x local columns 0..wr-1
    const int ntile = (wr + 1 + 63)/64;
    int t = blockIdx.x, ti = (int)((sqrtf(8.0f*(float)t + 1.0f) - 1.0f)*0.5f);
    while (ti*(ti + 1)/2 > t) ti--;
    while ((ti + 1)*(ti + 2)/2 <= t) ti++;
    const int tj = t - ti*(ti + 1)/2;
    if (ti >= ntile) return;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Pa = sm, *Pb = sm + 64*(CH_NB + 1);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ld = W.ldS, lds = CH_NB + 1;
    double *A = W.S;
    const int r0 = 64*ti, q0 = 64*tj;                           // local
    const int nbp = (nb + 3) & ~3;                              // K padded to the MFMA depth with exact zeros
    for (int k = tid; k < 64*nbp; k += CH_T) {
        int r = k / nbp, c = k - r*nbp;
        const int rl = r0 + r, ql = q0 + r;
        const int gr = rl < wr ? c0 + rl : n;
        Pa[r*lds + c] = (c < nb && rl <= wr) ? (gr == n ? W.Sy : A + (size_t)gr*ld)[j0 + c] : 0.0;
        Pb[r*lds + c] = (c < nb && ql < wr) ? A[(size_t)(c0 + ql)*ld + j0 + c] : 0.0;
    }
    __syncthreads();
    const int wrw = (wave >> 1)*32, wc = (wave & 1)*32;         // quadrant of this wave
    const int lr = lane & 15, lk = lane >> 4;
    v4d_c acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) acc[a][b] = (v4d_c){0.0, 0.0, 0.0, 0.0};
    for (int k = 0; k < nb; k += 4) {
        double av[2], bv[2];
#pragma unroll
        for (int a = 0; a < 2; a++) av[a] = Pa[(wrw + 16*a + lr)*lds + k + lk];
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
                const int rl = r0 + wrw + 16*a + lk + 4*r, ql = q0 + wc + 16*b + lr;
                if (rl <= wr && ql < wr && (rl == wr || ql <= rl)) { const int gr = rl < wr ? c0 + rl : n; (gr == n ? W.Sy : A + (size_t)gr*ld)[c0 + ql] -= acc[a][b][r]; }
            }
}

// ---- back substitution L^T x = y (y = row n), one workgroup of 1024 threads; dp[6a + k] = -x[6 fidx[a] + k].
// Per block from the last: x = W^T y_block with the stored inverse (staged in LDS), then the band update of the rows above.
__global__ __launch_bounds__(1024) void k_chol_backsub(Work W, int bw) {
    LmState *st = W.st;
    if (st->done) return;
    const int n = 6 * *W.nfree;
    const int tid = threadIdx.x, ld = W.ldS;
    double *A = W.S, *y = W.Sy;
    if (st->step_fail) { for (int k = tid; k < W.N; k += 1024) W.dp[k] = 0.0; return; }
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Wt = sm, *ys = sm + CH_NB*(CH_NB + 1), *xs = ys + CH_NB, *part = xs + CH_NB;      // Wt[c][r] = W[r][c]; part [8][96]
    const int nblk = (n + CH_NB - 1)/CH_NB;
    for (int jb = nblk - 1; jb >= 0; jb--) {
        const int j0 = jb*CH_NB, nb = min(CH_NB, n - j0);
        for (int k = tid; k < nb*nb; k += 1024) {
            const int c = k / nb, r = k - c*nb;
            Wt[c*(CH_NB + 1) + r] = r > c ? A[(size_t)(j0 + c)*ld + j0 + r] : (r == c ? W.LDbuf[j0 + r] : 0.0);
        }
        if (tid < nb) ys[tid] = y[j0 + tid];
        __syncthreads();
        if (tid < 8*CH_NB) {                                    // x[c] = sum_{r >= c} W[r][c] y[r], 8 partial sums per entry
            const int c = tid >> 3, p = tid & 7;
            double v = 0.0;
            if (c < nb) for (int r = c + p; r < nb; r += 8) v = fma(Wt[c*(CH_NB + 1) + r], ys[r], v);
            part[p*CH_NB + c] = v;
        }
        __syncthreads();
        if (tid < nb) {
            double v = 0.0;
#pragma unroll
            for (int p = 0; p < 8; p++) v += part[p*CH_NB + tid];
            xs[tid] = v; y[j0 + tid] = v;
        }
        __syncthreads();
        // rows j0..j0+nb-1 of L reach back at most bw + 2 NB columns (column block J holds rows up to J + NB - 1 + bw)
        // L[i][k] is stored (and can be non-zero) only for i - k <= Wb = bw + NB - 1: with band storage anything further left
        // aliases another row, so both the column range and, per column, the row range are cut there
        const int Wb = bw + CH_NB - 1, k0 = max(0, j0 - Wb);
        for (int k = k0 + tid; k < j0; k += 1024) {
            const int cmax = min(nb, k + Wb - j0 + 1);
            double v0 = 0.0, v1 = 0.0;
            int c = 0;
            for (; c + 1 < cmax; c += 2) { v0 = fma(A[(size_t)(j0 + c)*ld + k], xs[c], v0); v1 = fma(A[(size_t)(j0 + c + 1)*ld + k], xs[c + 1], v1); }
            if (c < cmax) v0 = fma(A[(size_t)(j0 + c)*ld + k], xs[c], v0);
            y[k] -= v0 + v1;
        }
        __syncthreads();
    }
    for (int a = tid; a < W.n_kf; a += 1024) {
        int ia = W.fidx[a];
        for (int k = 0; k < 6; k++) W.dp[6*a + k] = ia >= 0 ? -y[6*ia + k] : 0.0;
    }
}

// Forward substitution alone, for another right-hand side on the factor the kernels above left in S (the factorisation carries ITS right-hand side
// along as an extra row; the low-rank correction of tsba_wb.h applies one k x k factor to several vectors per LM trial): y = L^-1 y in W.Sy, block
// by block  y_J = W_J y_J (the stored inverse of the diagonal factor),  y_below -= L(below, J) y_J.  One workgroup, the mirror of k_chol_backsub
// (which follows it unchanged).
__global__ __launch_bounds__(1024) void k_chol_fwd(Work W, int bw) {
    LmState *st = W.st;
    if (st->done || st->step_fail) return;
    const int n = 6 * *W.nfree;
    const int tid = threadIdx.x, ld = W.ldS;
    const double *A = W.S; double *y = W.Sy;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double *Wt = sm, *ys = sm + CH_NB*(CH_NB + 1), *xs = ys + CH_NB, *part = xs + CH_NB;      // Wt[c][r] = W[r][c]; part [8][96]
    const int nblk = (n + CH_NB - 1)/CH_NB, Wb = bw + CH_NB - 1;
    for (int jb = 0; jb < nblk; jb++) {
        const int j0 = jb*CH_NB, nb = min(CH_NB, n - j0);
        for (int k = tid; k < nb*nb; k += 1024) {
            const int c = k / nb, r = k - c*nb;
            Wt[c*(CH_NB + 1) + r] = r > c ? A[(size_t)(j0 + c)*ld + j0 + r] : (r == c ? W.LDbuf[j0 + r] : 0.0);
        }
        if (tid < nb) ys[tid] = y[j0 + tid];
        __syncthreads();
        if (tid < 8*CH_NB) {                                    // x[r] = sum_{c <= r} W[r][c] y[c], 8 partial sums per entry
            const int r = tid >> 3, p = tid & 7;
            double v = 0.0;
            if (r < nb) for (int c = p; c <= r; c += 8) v = fma(Wt[c*(CH_NB + 1) + r], ys[c], v);
            part[p*CH_NB + r] = v;
        }
        __syncthreads();
        if (tid < nb) {
            double v = 0.0;
#pragma unroll
            for (int p = 0; p < 8; p++) v += part[p*CH_NB + tid];
            xs[tid] = v; y[j0 + tid] = v;
        }
        __syncthreads();
        // rows below the block: L[i][k] is stored for i - k <= Wb
        for (int i = j0 + nb + tid; i < min(n, j0 + nb + Wb); i += 1024) {
            const int c0 = max(0, i - Wb - j0);
            const double *row = A + (size_t)i*ld + j0;
            double v0 = 0.0, v1 = 0.0;
            int c = c0;
            for (; c + 1 < nb; c += 2) { v0 = fma(row[c], xs[c], v0); v1 = fma(row[c + 1], xs[c + 1], v1); }
            if (c < nb) v0 = fma(row[c], xs[c], v0);
            y[i] -= v0 + v1;
        }
        __syncthreads();
    }
}

// copies g into row n of S (the extra rhs row) -- the LDS solver does this itself
__global__ void k_chol_rhs(Work W) {
    if (W.st->done) return;
    const int n = 6 * *W.nfree;
    int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k < n) W.Sy[k] = W.g[k];
}
