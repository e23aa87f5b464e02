// Separator kernels of the many-column solve phase (tsba_bandms.h) in product form, for separators of at most MX_SMAX rows: with the inverse
// unit-lower factors k_sv_linv builds once per factorisation (tsba_bandsv.h) a cyclic-reduction level is three dense s x s products per pivot
// and 64 columns -- no substitution on one wave of eight, no coupling rows fetched one round trip at a time:
//   * the three matrices (inverse factor, X_a, X_c) are requested at once by all threads and parked in LDS, together with the flags and the
//     right-hand side rows (one wait; the pending updates of the higher levels in at most three more);
//   * a lane owns a column and keeps it in REGISTERS while its wave's rows are formed (a matrix entry is one broadcast LDS read for two FMAs; with
//     the vector in LDS too every FMA cost two reads);
//   * rows of the results are dealt to the eight waves.
// k_ms_cre_fwd / root / back stay for wider separators (the LDS holds three matrices and a vector block only up to 66 rows).
#pragma once

#define MX_T 512
#define MX_SMAX 66
#define MX_RW ((MX_SMAX + 7)/8)              // rows of a vector per wave
#define MX_ST ((3*MX_SMAX*MX_SMAX/2 + MX_T - 1)/MX_T)     // 16-byte pieces of the three matrices per thread
static size_t mx_lds_doubles(int s) { return 3*(size_t)s*(s + 2) + (size_t)s*64; }

struct MxStage { v2d v[MX_ST]; };
__device__ __forceinline__ void mx_stage_load(const double *m0, const double *m1, const double *m2, int s, int tid, MxStage &St) {
    const int ss = s*s, n2 = 3*ss/2;
#pragma unroll
    for (int u = 0; u < MX_ST; u++) { const int e = tid + MX_T*u; St.v[u] = v2d{0.0, 0.0};
        if (e < n2) { const int e2 = 2*e, m = e2 >= 2*ss ? 2 : (e2 >= ss ? 1 : 0); const double *src = m == 0 ? m0 : (m == 1 ? m1 : m2);
            St.v[u] = *(const v2d *)(src + (e2 - m*ss)); } }
}
__device__ __forceinline__ void mx_stage_pin(MxStage &St) {
#pragma unroll
    for (int u = 0; u < MX_ST; u++) sv_pin(St.v[u]);
}
__device__ __forceinline__ void mx_stage_store(const MxStage &St, int s, int tid, double *lds) {
    const int n2 = 3*s*s/2;
#pragma unroll
    for (int u = 0; u < MX_ST; u++) { const int e = tid + MX_T*u; if (e < n2) *(v2d *)(lds + 2*e) = St.v[u]; }
}
// the same, every matrix TRANSPOSED on the way into LDS (row stride s + 2): the backward level multiplies by L^-T, X_a^T, X_c^T, and a column of a
// row-major matrix is one 8-byte broadcast read per FMA where a row is one 16-byte read per two
__device__ __forceinline__ void mx_stage_store_t(const MxStage &St, int s, int tid, double *lds) {
    const int ss = s*s, n2 = 3*ss/2, ld = s + 2;
#pragma unroll
    for (int u = 0; u < MX_ST; u++) { const int e = tid + MX_T*u;
        if (e < n2) { const int e2 = 2*e, m = e2 >= 2*ss ? 2 : (e2 >= ss ? 1 : 0), off = e2 - m*ss, t = off/s, r = off - t*s; double *d = lds + (size_t)m*s*ld + (size_t)r*ld + t;
            d[0] = St.v[u].x; d[ld] = St.v[u].y; } }
}
// row . column, the column in registers.  S (rows of a separator) is a template parameter of the kernels: every loop has compile-time bounds and
// LDS offsets, no guards (with run-time s and guarded, unrolled loops a row cost 33 uniform branches: the backward kernel ran 72 us).  The zeros
// above the diagonal of an inverse factor are multiplied along.
template <int S>
__device__ __forceinline__ double mx_dot(const double *row, const double (&c)[S]) {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int k = 0; k < S; k += 2) { const v2d m = *(const v2d *)(row + k); a0 = fma(m.x, c[k], a0); a1 = fma(m.y, c[k + 1], a1); }
    return a0 + a1;
}
// column r of a row-major S x S matrix . column in registers: sum_t M[t][r] c[t]
template <int S>
__device__ __forceinline__ double mx_dot_t(const double *col, const double (&c)[S]) {
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int t = 0; t < S; t += 2) { a0 = fma(col[t*S], c[t], a0); a1 = fma(col[(t + 1)*S], c[t + 1], a1); }
    return a0 + a1;
}
template <int S>
__device__ __forceinline__ void mx_column(const double *vb, int lane, double (&c)[S]) {
#pragma unroll
    for (int k = 0; k < S; k++) c[k] = vb[k*64 + lane];
}

// ---- level h, forward.  grid (pivots, column groups), MX_T threads.
template <int S>
__global__ __launch_bounds__(MX_T) void k_mx_cre_fwd(Work W, Work Ws, int bw, int Pmax, int h, int kb, MsBuf M, const double *__restrict__ Li, const double *__restrict__ Lid) {
    extern __shared__ __attribute__((aligned(16))) double ms_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = ms_uni(tid >> 6), T = M.T, col = 64*blockIdx.y + lane; const bool on = col < T; const int cc_ = on ? col : 0;
    constexpr int s = S, B = S/6; const int mmax = cr_mmax(W.ring, Pmax, W.ring_g);
    const int i = (2*(kb + (int)blockIdx.x) + 1)*h, ia = i - h, ic = i + h;
    const LmState *st_ = W.st; const int flags = st_->done | st_->lin_done | st_->step_fail, nf = *W.nfree;
    double *mat = ms_smem, *vb = mat + 3*(size_t)s*s;          // [Li | X_a | X_c], vector block [s][64]
    MxStage St; mx_stage_load(Li + (size_t)i*s*s, cr_blk(Ws.S, s, mmax, i, ia), cr_blk(Ws.S, s, mmax, ic, i), s, tid, St);
    double gv[MX_RW], idv[MX_RW];
#pragma unroll
    for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j; gv[j] = (on && r < s) ? M.G[((size_t)i*s + r)*T + cc_] : 0.0; idv[j] = r < s ? Lid[(size_t)i*s + r] : 0.0; sv_pin(gv[j]); }
    mx_stage_pin(St);
    if (flags) return;
    const int m = ms_uni(sv_nsep(nf, B, Pmax)), lo = 0, r0 = 0;
    if (i < lo || i >= m) return;
    const bool has_a = ia >= lo, has_c = ic < m;
    mx_stage_store(St, s, tid, mat);
    // pending updates: pivots i -+ 2^l of the levels 2^l < h (always pivots of their level), two levels per round trip
    for (int l0 = 0; (1 << l0) < h && (1 << l0) < m - lo; l0 += 2) {
        double pv[2][2][MX_RW];
#pragma unroll
        for (int q = 0; q < 2; q++) { const int hp = 1 << (l0 + q); const bool lev = hp < h && hp < m - lo; const int pl = i - hp, pr = i + hp;
            const bool okl = lev && pl >= lo && pl != r0, okr = lev && pr < m && pr != r0;
#pragma unroll
            for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j;
                pv[q][0][j] = (on && okl && r < s) ? M.Cg[(((size_t)pl*2 + 1)*s + r)*T + cc_] : 0.0;
                pv[q][1][j] = (on && okr && r < s) ? M.Cg[(((size_t)pr*2 + 0)*s + r)*T + cc_] : 0.0; } }
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int j = 0; j < MX_RW; j++) { gv[j] -= pv[q][0][j]; gv[j] -= pv[q][1][j]; }
    }
#pragma unroll
    for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j; if (r < s) vb[r*64 + lane] = gv[j]; }
    __syncthreads();
    double c[S];
    mx_column<S>(vb, lane, c);
    __syncthreads();
    {   // w = L^-1 v (lower triangle), z = D^-1 w
        double wr[MX_RW];
#pragma unroll
        for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j; wr[j] = r < s ? mx_dot<S>(mat + r*s, c) : 0.0; }
#pragma unroll
        for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j; if (r < s) { vb[r*64 + lane] = wr[j]; if (on) M.Z[((size_t)i*s + r)*T + cc_] = wr[j]*idv[j]; } }
    }
    __syncthreads();
    mx_column<S>(vb, lane, c);
    // the neighbours' updates X_a w, X_c w
    const double *Xa = mat + (size_t)s*s, *Xc = Xa + (size_t)s*s;
    for (int q = wave; q < 2*s; q += 8) {
        const bool first = q < s; const int r = first ? q : q - s;
        const double acc = mx_dot<S>((first ? Xa : Xc) + r*s, c);
        if (on) M.Cg[(((size_t)i*2 + (first ? 0 : 1))*s + r)*T + cc_] = (first ? has_a : has_c) ? acc : 0.0;
    }
}

// ---- the last block: forward and backward.  grid (1, column groups).
template <int S>
__global__ __launch_bounds__(MX_T) void k_mx_cre_root(Work W, int bw, int Pmax, MsBuf M, const double *__restrict__ Li, const double *__restrict__ Lid) {
    extern __shared__ __attribute__((aligned(16))) double ms_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = ms_uni(tid >> 6), T = M.T, col = 64*blockIdx.y + lane; const bool on = col < T; const int cc_ = on ? col : 0;
    constexpr int s = S, B = S/6; const int mmax = cr_mmax(W.ring, Pmax, W.ring_g), i = 0;           // (chains: the root is label 0)
    const LmState *st_ = W.st; const int flags = st_->done | st_->lin_done | st_->step_fail, nf = *W.nfree;
    double *mat = ms_smem, *vb = mat + 3*(size_t)s*s;
    MxStage St; mx_stage_load(Li, Li, Li, s, tid, St);            // (only the first third is used)
    double gv[MX_RW], idv[MX_RW];
#pragma unroll
    for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j; gv[j] = (on && r < s) ? M.G[((size_t)i*s + r)*T + cc_] : 0.0; idv[j] = r < s ? Lid[(size_t)i*s + r] : 0.0; sv_pin(gv[j]); }
    mx_stage_pin(St);
    if (flags) return;
    const int m = ms_uni(sv_nsep(nf, B, Pmax));
    if (m <= 0) return;
    mx_stage_store(St, s, tid, mat);
    for (int l0 = 0; (1 << l0) < m; l0 += 2) {                  // pending: the pivots 2^l of every level
        double pv[2][MX_RW];
#pragma unroll
        for (int q = 0; q < 2; q++) { const int hp = 1 << (l0 + q); const bool okr = hp < m && hp < mmax;
#pragma unroll
            for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j; pv[q][j] = (on && okr && r < s) ? M.Cg[(((size_t)hp*2 + 0)*s + r)*T + cc_] : 0.0; } }
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
            for (int j = 0; j < MX_RW; j++) gv[j] -= pv[q][j];
    }
#pragma unroll
    for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j; if (r < s) vb[r*64 + lane] = gv[j]; }
    __syncthreads();
    double c[S];
    mx_column<S>(vb, lane, c);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j; if (r < s) vb[r*64 + lane] = mx_dot<S>(mat + r*s, c)*idv[j]; }      // z = D^-1 L^-1 v
    __syncthreads();
    mx_column<S>(vb, lane, c);
#pragma unroll
    for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j; if (r < s && on) M.Xs[((size_t)i*s + r)*T + cc_] = mx_dot_t<S>(mat + r, c); }          // x = L^-T z
}

// ---- level h, backward.  grid (pivots, column groups):  x_i = L^-T (z_i - X_a^T x_a - X_c^T x_c)
template <int S>
__global__ __launch_bounds__(MX_T) void k_mx_cre_back(Work W, Work Ws, int bw, int Pmax, int h, int kb, MsBuf M, const double *__restrict__ Li) {
    extern __shared__ __attribute__((aligned(16))) double ms_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = ms_uni(tid >> 6), T = M.T, col = 64*blockIdx.y + lane; const bool on = col < T; const int cc_ = on ? col : 0;
    constexpr int s = S, B = S/6; const int mmax = cr_mmax(W.ring, Pmax, W.ring_g);
    const int i = (2*(kb + (int)blockIdx.x) + 1)*h, ia = i - h, ic = i + h; const bool cin = ic < mmax;
    const LmState *st_ = W.st; const int flags = st_->done | st_->lin_done | st_->step_fail, nf = *W.nfree;
    constexpr int LD = S + 2; double *mat = ms_smem, *vb = mat + 3*(size_t)s*LD;          // [L^-T | X_a^T | X_c^T] (row stride LD), vector block [s][64]
    MxStage St; mx_stage_load(Li + (size_t)i*s*s, cr_blk(Ws.S, s, mmax, i, ia), cr_blk(Ws.S, s, mmax, ic, i), s, tid, St);
    double zv[MX_RW], xcr[MX_RW], c[S];
#pragma unroll
    for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j; zv[j] = (on && r < s) ? M.Z[((size_t)i*s + r)*T + cc_] : 0.0; xcr[j] = (on && cin && r < s) ? M.Xs[((size_t)ic*s + r)*T + cc_] : 0.0; sv_pin(zv[j]); sv_pin(xcr[j]); }
#pragma unroll
    for (int k = 0; k < S; k++) { c[k] = on ? M.Xs[((size_t)ia*s + k)*T + cc_] : 0.0; sv_pin(c[k]); }      // x_a: the whole column
    mx_stage_pin(St);
    if (flags) return;
    const int m = ms_uni(sv_nsep(nf, B, Pmax)), lo = 0;
    if (i < lo || i >= m) return;
    const bool has_a = ia >= lo, has_c = ic < m;
    mx_stage_store_t(St, s, tid, mat);
#pragma unroll
    for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j; if (r < s) vb[r*64 + lane] = has_c ? xcr[j] : 0.0; }
    __syncthreads();
    const double *Xa = mat + (size_t)s*LD, *Xc = Xa + (size_t)s*LD;
    double u[MX_RW];
#pragma unroll
    for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j; u[j] = (r < s && has_a) ? zv[j] - mx_dot<S>(Xa + r*LD, c) : zv[j]; }
    mx_column<S>(vb, lane, c);                                  // x_c
    __syncthreads();
#pragma unroll
    for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j; if (r < s) { if (has_c) u[j] -= mx_dot<S>(Xc + r*LD, c); vb[r*64 + lane] = u[j]; } }
    __syncthreads();
    mx_column<S>(vb, lane, c);                                  // u
#pragma unroll
    for (int j = 0; j < MX_RW; j++) { const int r = wave + 8*j; if (r < s && on) M.Xs[((size_t)i*s + r)*T + cc_] = mx_dot<S>(mat + r*LD, c); }
}
