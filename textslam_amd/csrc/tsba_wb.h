// Loop closures as a LOW-RANK correction of the band solve (included by tsba.hip after tsba_pcg.h).
//
// A map with a few loop closures has S = M + E with E a few hundred 6x6 blocks between a few dozen keyframes U (HostPlan::wb_kf: the
// keyframes a block of E touches, at most WB_MAXKF of them): E = I_U E^ I_U^T with a k x k matrix E^, k = 6 |U|.  Then (Woodbury)
//     S^-1 r = y - Z E^ x_U,      y = M^-1 r,   Z = M^-1 I_U  (k columns: ONE run of the many-right-hand-side solve phase, tsba_bandms.h),
//     x_U = G w,   (G + G E^ G) w = y_U,   G = Z_U  (k x k, symmetric positive definite: a principal block of M^-1)   [from (I + G E^) x_U = y_U]
// -- a direct solve: the band factorisation, one solve phase with k columns, a dense k x k Cholesky (the multi-workgroup Cholesky of
// tsba_chol.h on a second Work) per LM trial.  The k x k system squares the condition of G; instead of trusting it blindly the result is used
// as the PRECONDITIONER of the conjugate gradients of tsba_pcg.h (M_W^-1 r as above for any r): where it is accurate the first iteration
// ends the solve, where the trial is ill-conditioned a few more make up for it.  The reference hands such maps to Ceres' sparse Cholesky
// (optimizer.cc:1727-1765, loopClosing.cc:587-591: GlobalBA after every closure).
#pragma once

#define WB_MAXKF 128                        // keyframes touched by E for this path (k <= 768)
static_assert(WB_MAXKF == WB_MAXKF_PLAN, "the plan builder (tsba_plan.h) and the kernels must agree on the size of the low-rank part");

struct WbBuf {
    double *Gm, *T1, *xu, *vu, *z;          // [k][k], [k][k], x_U [k], v = E^ x_U [k], z = M_W^-1 r [6 n_kf] (compressed rows)
    const int *wb_kf, *wb_idx;              // U (keyframes, ascending); index of a keyframe in U or -1
    int n_u, k;
};

__global__ void k_wb_init(int *fidx, int *nfree, int n_u) {          // the k x k system as a "map" of n_u free poses in their own order
    for (int i = threadIdx.x; i < n_u; i += blockDim.x) fidx[i] = i;
    if (threadIdx.x == 0) { nfree[0] = n_u; nfree[1] = 0; }
}
// unit columns: R[row][j] = 1 where row is row j % 6 of the free pose U[j / 6]
__global__ __launch_bounds__(256) void k_wb_units(Work W, MsBuf M, WbBuf B) {
    const LmState *st = W.st; if (st->done || st->step_fail) return;
    const int nrow = 6*W.nfree[0], k = B.k;
    for (long long e = (long long)blockIdx.x*256 + threadIdx.x; e < (long long)nrow*k; e += (long long)gridDim.x*256) {
        const int row = (int)(e/k), j = (int)(e - (long long)row*k);
        const int ia = W.fidx[B.wb_kf[j/6]];
        M.R[e] = (ia >= 0 && row == 6*ia + j % 6) ? 1.0 : 0.0;
    }
}
// G = rows U of Z (identity rows / columns for keyframes of U that are not free)
__global__ __launch_bounds__(256) void k_wb_gather(Work W, MsBuf M, WbBuf B) {
    const LmState *st = W.st; if (st->done || st->step_fail) return;
    const int k = B.k;
    for (int e = blockIdx.x*256 + threadIdx.x; e < k*k; e += gridDim.x*256) {
        const int i = e/k, j = e - i*k;
        const int ia = W.fidx[B.wb_kf[i/6]], ja = W.fidx[B.wb_kf[j/6]];
        B.Gm[e] = (ia >= 0 && ja >= 0) ? M.X[(size_t)(6*ia + i % 6)*M.T + j] : (i == j ? 1.0 : 0.0);
    }
}
// out rows of pose u = sum over the blocks of E at u:  E_q in[rows of the other pose]  (in, out: [k][ncol]); one workgroup per pose of U
__device__ __forceinline__ void wb_apply_E(const Work &W, const LevelDev &L, const WbBuf &B, int u, const double *in, double *out, int ncol) {
    const int a = B.wb_kf[u], tid = threadIdx.x;
    const bool live = W.fidx[a] >= 0;
    for (int j = tid; j < ncol; j += blockDim.x) {
        double acc[6] = {0, 0, 0, 0, 0, 0};
        if (live) for (int e = L.far_off[a]; e < L.far_off[a + 1]; e++) {
            const int ent = L.far_ent[e], fid = ent >> 1, side = ent & 1, o = side ? L.far_a[fid] : L.far_b[fid];
            if (W.fidx[o] < 0) continue;
            const int uo = B.wb_idx[o]; const double *sb = W.Sfar + (size_t)fid*36;
#pragma unroll
            for (int c = 0; c < 6; c++) { const double x = in[(size_t)(6*uo + c)*ncol + j];
#pragma unroll
                for (int r = 0; r < 6; r++) acc[r] = fma(side ? sb[6*c + r] : sb[6*r + c], x, acc[r]); }
        }
#pragma unroll
        for (int r = 0; r < 6; r++) out[(size_t)(6*u + r)*ncol + j] = acc[r];
    }
}
__global__ __launch_bounds__(256) void k_wb_EG(Work W, LevelDev L, WbBuf B) {           // T1 = E^ G
    const LmState *st = W.st; if (st->done || st->step_fail) return;
    wb_apply_E(W, L, B, blockIdx.x, B.Gm, B.T1, B.k);
}
// K2 = G + G^T T1 (symmetrised) into the dense matrix of the second Work (row stride k)
__global__ __launch_bounds__(256) void k_wb_K2(Work W, WbBuf B, double *K2) {
    const LmState *st = W.st; if (st->done || st->step_fail) return;
    const int k = B.k;
    for (int e = blockIdx.x*256 + threadIdx.x; e < k*k; e += gridDim.x*256) {
        const int i = e/k, j = e - i*k;
        if (j > i) continue;
        double s0 = 0.0, s1 = 0.0;
        for (int l = 0; l < k; l++) { s0 = fma(B.Gm[(size_t)l*k + i], B.T1[(size_t)l*k + j], s0); s1 = fma(B.Gm[(size_t)l*k + j], B.T1[(size_t)l*k + i], s1); }
        const double v = 0.5*(B.Gm[(size_t)i*k + j] + B.Gm[(size_t)j*k + i]) + 0.5*(s0 + s1);
        K2[(size_t)i*k + j] = v; K2[(size_t)j*k + i] = v;
    }
}
// right-hand side of the k x k system (G + G E^ G) w = y_U: g_k = -y_U, y = ys * yp[] (the dense solver returns dp = -K2^-1 g_k = w); one workgroup
__global__ __launch_bounds__(512) void k_wb_rhs(Work W, WbBuf B, const double *yp, double ys, double *gk) {
    const LmState *st = W.st; if (st->done || st->step_fail || st->lin_done) return;
    for (int j = threadIdx.x; j < B.k; j += 512) { const int ja = W.fidx[B.wb_kf[j/6]]; gk[j] = ja >= 0 ? -ys*yp[6*ja + j % 6] : 0.0; }
}
// x_U = G w; one workgroup
__global__ __launch_bounds__(512) void k_wb_Gw(Work W, WbBuf B, const double *w) {
    __shared__ double ws[6*WB_MAXKF];
    const LmState *st = W.st; if (st->done || st->step_fail || st->lin_done) return;
    const int k = B.k, tid = threadIdx.x;
    for (int j = tid; j < k; j += 512) ws[j] = w[j];
    __syncthreads();
    for (int i = tid; i < k; i += 512) { double s = 0.0; const bool live = W.fidx[B.wb_kf[i/6]] >= 0;
        if (live) for (int j = 0; j < k; j++) s = fma(B.Gm[(size_t)i*k + j], ws[j], s);
        B.xu[i] = live ? s : 0.0; }
}
// v = E^ x_U (one workgroup per pose of U), x_U = the dense solver's dp
__global__ __launch_bounds__(64) void k_wb_Ex(Work W, LevelDev L, WbBuf B, const double *xu) {
    const LmState *st = W.st; if (st->done || st->step_fail || st->lin_done) return;
    if (threadIdx.x == 0) wb_apply_E(W, L, B, blockIdx.x, xu, B.vu, 1);
}
// z = y - Z v for every free row (a wave per row, lanes over the k columns)
__global__ __launch_bounds__(256) void k_wb_apply(Work W, MsBuf M, WbBuf B, const double *yp, double ys) {
    const LmState *st = W.st; if (st->done || st->step_fail || st->lin_done) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, k = B.k, nrow = 6*W.nfree[0];
    for (int row = blockIdx.x*4 + wave; row < nrow; row += gridDim.x*4) {
        double s = 0.0;
        for (int j = lane; j < k; j += 64) s = fma(M.X[(size_t)row*M.T + j], B.vu[j], s);
        s = wave_sum1(s);
        if (lane == 0) B.z[row] = ys*yp[row] - s;
    }
}
