// Solve phase of the partitioned band solver for MANY right-hand sides: M X = R with the factor k_bandp_factor / k_bandp_sepf / k_cre_elim
// left behind (tsba_bandp.h, tsba_bandcre.h) -- no factorisation, no S.  The factorisation carries ONE right-hand side (it rides along as
// the last row); the iterative and low-rank solvers on top of it (tsba_pcg.h) need M^-1 applied to other vectors: a residual per
// conjugate-gradient iteration, a few hundred unit-like columns for the loop-closure correction.
//
// Layout: a right-hand side block is [rows][T] (T columns, row-major): a LANE owns a COLUMN.  Everything a step needs from the factor
// (a 6x6 block, a row of the packed triangle, a diagonal table) is the same for all lanes -- scalar / broadcast loads -- and every lane
// runs the same recurrence on its own column: 64 columns cost what one costs.  The chain through an interior is one wave per (interior,
// 64 columns); the dense parts of a separator pivot are dealt to 8 waves by rows.
//
//   k_ms_fwd_int     interiors forward:   w_q = l_q^-1 (r_q - sum_j L(q, j) w_j),  v = D^-1 w
//   k_ms_sep_rhs     separator right-hand sides:  g_s = r_s - [rows of s below the interior on its left] w - [border rows of the interior on its right] w
//   k_ms_cre_fwd     cyclic-reduction level h, forward:  z_i = D^-1 L_i^-1 (g_i - pending),  updates X_a w, X_c w for the neighbours' g
//   k_ms_cre_root    the last block: forward and backward
//   k_ms_cre_back    level h, backward:   x_i = L_i^-T (z_i - X_a^T x_a - X_c^T x_c)
//   k_ms_back_border interiors, the part of the back substitution that does not depend on the running solution:  v_q -= Lb_q^T x_left
//   k_ms_back_int    interiors backward:  x_q = l_q^-T (v_q - sum_R L(R, q)^T x_R)
// Chain partitions only (no ring / ghost rows), separator system by cyclic reduction (tsba_bandcre.h).
#pragma once

#define MS_BMAX 13                          // widest separator in pose blocks (CR_SMAX / 6)
#define MS_CT 512                           // workgroup of the separator kernels: 8 waves

struct MsBuf {                              // device buffers of one multi-right-hand-side solve (sized for T_cap columns)
    double *R, *Wm, *V, *X;                 // [6 nfree][T]: right-hand sides, w = L^-1 r (interior rows), v = D^-1 w, solution
    double *G, *Z, *Xs, *Cg;                // [labels][s][T]: separator right-hand sides, z, solution; [labels][2][s][T] pending updates (for a | for c)
    double *G2, *Li, *Lid;                  // single-vector solve phase (tsba_bandsv.h): [labels][s] border part of a separator's right-hand side; [labels][s][s] inverse unit-lower factors, [labels][s] 1/d
    double *Pp;                             // [labels][2][s][s]: the couplings of a pivot to its two neighbours times its inverse factor (k_sv_linv)
    int T;                                  // columns (row stride)
};

__device__ __forceinline__ int ms_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- interiors, forward.  grid (Pmax, column groups), one wave.  The factor data of a step -- the row block's record (36 B doubles) and its
// diagonal table -- is the same for every lane: it is fetched by the wave as ONE coalesced load a step ahead (registers), parked in LDS and
// read back as broadcasts.  (As uniform scalar loads the compiler emits one s_load_dwordx16 and waits for it, 40 times per step: 8 us per
// step, 240 us per interior; staged: ~1 us per step.)
#define MS_RECMAX (36*MS_BMAX)
__global__ __launch_bounds__(64) void k_ms_fwd_int(Work W, int bw, int Pmax, const double *__restrict__ Lrow, MsBuf M) {
    __shared__ __attribute__((aligned(16))) double hist[MS_BMAX*6*64];
    __shared__ __attribute__((aligned(16))) double recb[2][MS_RECMAX + 32];
    const int lane = threadIdx.x, col = 64*blockIdx.y + lane, T = M.T; const bool on = col < T; const int cc_ = on ? col : 0;
    { const LmState *st_ = W.st; if (st_->done | st_->lin_done | st_->step_fail) return; }     // (a converged iterative solve: the launches the host still had in flight)
    const int B = bw/6, nf = ms_uni(*W.nfree);
    if (nf <= 0) return;
    const BandpPart PT = bandp_part(nf, B, Pmax, blockIdx.x);
    const int a = ms_uni(PT.a), b = ms_uni(PT.b);
    if ((int)blockIdx.x >= ms_uni(PT.P)) return;
    const int REC = bw*6;
    constexpr int NL = (MS_RECMAX + 63)/64;
    double pre[NL], preld, tn[6];
    auto fetch = [&](int q) {                                   // operands of step q: issued one step ahead
        const double *rec = Lrow + (size_t)q*REC; const int nv = 36*min(B, q - a);
#pragma unroll
        for (int k = 0; k < NL; k++) { const int e = lane + 64*k; pre[k] = e < nv ? rec[e] : 0.0; }
        preld = lane < 22 ? W.LDbuf[32*(size_t)q + lane] : 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) tn[k] = on ? M.R[(size_t)(6*q + k)*T + cc_] : 0.0;
    };
    fetch(a);
    for (int q = a; q < b; q++) {
        double *rb = recb[q & 1];
        double t[6];
#pragma unroll
        for (int k = 0; k < NL; k++) { const int e = lane + 64*k; if (e < MS_RECMAX) rb[e] = pre[k]; }
        if (lane < 22) rb[MS_RECMAX + lane] = preld;
#pragma unroll
        for (int k = 0; k < 6; k++) t[k] = tn[k];
        if (q + 1 < b) fetch(q + 1);
        wave_lds_fence();
        const int nbk = min(B, q - a);
        for (int bb = 0; bb < nbk; bb++) {
            const int j = q - 1 - bb; const double *Lb_ = rb + bb*36, *hj = hist + (j % B)*6*64 + lane;
            double wj[6];
#pragma unroll
            for (int c = 0; c < 6; c++) wj[c] = hj[c*64];
#pragma unroll
            for (int c = 0; c < 6; c++)
#pragma unroll
                for (int r = 0; r < 6; r++) t[r] = fma(-Lb_[c*6 + r], wj[c], t[r]);
        }
        const double *ld = rb + MS_RECMAX;
#pragma unroll
        for (int r = 1; r < 6; r++)
#pragma unroll
            for (int c = 0; c < r; c++) t[r] = fma(-ld[tri(r - 1) + c], t[c], t[r]);
        double *hq = hist + (q % B)*6*64 + lane;
#pragma unroll
        for (int k = 0; k < 6; k++) { hq[k*64] = t[k];
            if (on) { M.Wm[(size_t)(6*q + k)*T + cc_] = t[k]; M.V[(size_t)(6*q + k)*T + cc_] = t[k]*ld[16 + k]; } }
        wave_lds_fence();
    }
}

// ---- separator right-hand sides.  grid (P - 1, column groups), a wave per pose block of separator s (B waves).  (One-wave workgroups per pose block
// had every block of a separator fetch the w rows of the whole interior on the right from L2 / HBM: 456 MB per launch at 5000 keyframes and 210 columns;
// as waves of one workgroup the other B - 1 reads are hits in the CU's cache.)
__global__ __launch_bounds__(64*MS_BMAX) void k_ms_sep_rhs(Work W, int bw, int Pmax, const double *__restrict__ Lrow, const double *__restrict__ Lb, MsBuf M) {
    const int lane = threadIdx.x & 63, jb = ms_uni(threadIdx.x >> 6), col = 64*blockIdx.y + lane, T = M.T; const bool on = col < T; const int cc_ = on ? col : 0;
    { const LmState *st_ = W.st; if (st_->done | st_->lin_done | st_->step_fail) return; }     // (a converged iterative solve: the launches the host still had in flight)
    const int B = bw/6, nf = ms_uni(*W.nfree);
    if (nf <= 0 || jb >= B) return;
    const int s = blockIdx.x;
    const BandpPart Pl = bandp_part(nf, B, Pmax, s);
    if (s >= ms_uni(Pl.P) - 1) return;
    const BandpPart Pr = bandp_part(nf, B, Pmax, s + 1);
    const int la = ms_uni(Pl.a), lb = ms_uni(Pl.b), ra = ms_uni(Pr.a), rb = ms_uni(Pr.b), REC = bw*6;
    const int gR = lb + jb;
    double acc[6];
#pragma unroll
    for (int k = 0; k < 6; k++) acc[k] = on ? M.R[(size_t)(6*gR + k)*T + cc_] : 0.0;
    for (int bb = 0; bb < B; bb++) {                           // the separator's rows below the interior on its left (its last B column blocks)
        const int j = gR - 1 - bb;
        if (j >= lb) continue;                                 // a column of the separator itself
        if (j < la) break;
        const double *Lk = Lrow + (size_t)gR*REC + bb*36;
        double wj[6];
#pragma unroll
        for (int c = 0; c < 6; c++) wj[c] = M.Wm[(size_t)(6*j + c)*T + cc_];
#pragma unroll
        for (int c = 0; c < 6; c++)
#pragma unroll
            for (int r = 0; r < 6; r++) acc[r] = fma(-Lk[c*6 + r], wj[c], acc[r]);
    }
    for (int q = ra; q < rb; q++) {                            // its rows as the border of the interior on its right
        const double *Lq = Lb + (size_t)q*REC + 6*(6*jb);
        double wq[6];
#pragma unroll
        for (int c = 0; c < 6; c++) wq[c] = M.Wm[(size_t)(6*q + c)*T + cc_];
#pragma unroll
        for (int r = 0; r < 6; r++)
#pragma unroll
            for (int c = 0; c < 6; c++) acc[r] = fma(-Lq[6*r + c], wq[c], acc[r]);
    }
    if (on) {
#pragma unroll
        for (int k = 0; k < 6; k++) M.G[((size_t)s*bw + 6*jb + k)*T + cc_] = acc[k];
    }
}

// forward / backward substitution with the packed factor of a separator block (rec: packed rows | LD table) on the columns of this wave; the
// vector lives in LDS as v[row*64 + lane]
__device__ __forceinline__ void ms_block_fwd(const double *__restrict__ rec, int s, int B, double *v, int lane) {
    const double *LDt = rec + rowoff(s);
    for (int jb = 0; jb < B; jb++) {
        double t[6];
#pragma unroll
        for (int k = 0; k < 6; k++) t[k] = v[(6*jb + k)*64 + lane];
        for (int e = 0; e < jb; e++) {
            double we[6];
#pragma unroll
            for (int c = 0; c < 6; c++) we[c] = v[(6*e + c)*64 + lane];
#pragma unroll
            for (int r = 0; r < 6; r++) { const double *row = rec + rowoff(6*jb + r) + 6*e;
#pragma unroll
                for (int c = 0; c < 6; c++) t[r] = fma(-row[c], we[c], t[r]); }
        }
        const double *ld = LDt + SOLVE_LD*jb;
#pragma unroll
        for (int r = 1; r < 6; r++)
#pragma unroll
            for (int c = 0; c < r; c++) t[r] = fma(-ld[tri(r - 1) + c], t[c], t[r]);
#pragma unroll
        for (int k = 0; k < 6; k++) v[(6*jb + k)*64 + lane] = t[k];
    }
}
__device__ __forceinline__ void ms_block_back(const double *__restrict__ rec, int s, int B, double *v, int lane) {      // L^T x = v in place
    const double *LDt = rec + rowoff(s);
    for (int jb = B - 1; jb >= 0; jb--) {
        double t[6];
#pragma unroll
        for (int k = 0; k < 6; k++) t[k] = v[(6*jb + k)*64 + lane];
        for (int e = jb + 1; e < B; e++) {
            double xe[6];
#pragma unroll
            for (int r = 0; r < 6; r++) xe[r] = v[(6*e + r)*64 + lane];
#pragma unroll
            for (int r = 0; r < 6; r++) { const double *row = rec + rowoff(6*e + r) + 6*jb;
#pragma unroll
                for (int c = 0; c < 6; c++) t[c] = fma(-row[c], xe[r], t[c]); }
        }
        const double *ld = LDt + SOLVE_LD*jb;
#pragma unroll
        for (int q = 4; q >= 0; q--)
#pragma unroll
            for (int k = q + 1; k < 6; k++) t[q] = fma(-ld[tri(k - 1) + q], t[k], t[q]);
#pragma unroll
        for (int k = 0; k < 6; k++) v[(6*jb + k)*64 + lane] = t[k];
    }
}
// g of block `blk` minus its pending updates (the producers of k_cre_elim's pending_of: pivots blk -+ h', h' < H, of level h')
__device__ __forceinline__ double ms_pending(const MsBuf &M, int s, int blk, int H, int lo, int m, int r0, int row, int col, double v) {
    const int T = M.T;
    for (int l = 0; l < 8; l++) {
        const int hp = 1 << l; if (!(hp < H && hp < m - lo)) break;
        const int pl = blk - hp, pr = blk + hp;
        if (pl >= lo && pl != r0 && (pl & (2*hp - 1)) == hp) v -= M.Cg[(((size_t)pl*2 + 1)*s + row)*T + col];     // blk is pl's right neighbour
        if (pr < m && pr != r0 && (pr & (2*hp - 1)) == hp) v -= M.Cg[(((size_t)pr*2 + 0)*s + row)*T + col];       // blk is pr's left neighbour
    }
    return v;
}

// the factor record of a separator (packed rows | LD table | z) into LDS, all threads, coalesced
template <int NT>
__device__ __forceinline__ void ms_stage_rec(const double *__restrict__ rec, int s, double *dst, int tid) {
    const int n = (int)cre_rec_doubles(s);
    for (int e0 = tid; e0 < n; e0 += 8*NT) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) { const int e = e0 + u*NT; v[u] = e < n ? rec[e] : 0.0; }
#pragma unroll
        for (int u = 0; u < 8; u++) { const int e = e0 + u*NT; if (e < n) dst[e] = v[u]; }
    }
}
static size_t ms_cre_lds_doubles(int s, int nvec) { return (cre_rec_doubles(s) + 8) + (size_t)nvec*s*64 + 8*(size_t)(s + 2) + 16; }

// ---- cyclic reduction, level h, forward.  grid (pivots, column groups), 8 waves.  LDS: the pivot's factor record | v [s][64] | a row buffer per wave.
__global__ __launch_bounds__(MS_CT) void k_ms_cre_fwd(Work W, Work Ws, int bw, int Pmax, int h, int kb, const double *__restrict__ fac, MsBuf M) {
    extern __shared__ __attribute__((aligned(16))) double ms_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = ms_uni(tid >> 6), col = 64*blockIdx.y + lane, T = M.T; const bool on = col < T; const int cc_ = on ? col : 0;
    { const LmState *st_ = W.st; if (st_->done | st_->lin_done | st_->step_fail) return; }     // (a converged iterative solve: the launches the host still had in flight)
    const CrRange rg = cr_range(W, bw, Pmax);
    const int m = ms_uni(rg.m), lo = ms_uni(rg.lo), r0 = ms_uni(rg.r0), s = bw, B = s/6, mmax = cr_mmax(W.ring, Pmax, W.ring_g);
    const int i = (2*(kb + (int)blockIdx.x) + 1)*h;
    if (i < lo || i >= m) return;
    const int a = i - h >= lo ? i - h : -1, c = i + h < m ? i + h : -1;
    double *recl = ms_smem, *v = recl + ((cre_rec_doubles(s) + 8) & ~(size_t)1), *rowb = v + (size_t)s*64 + wave*(s + 2);
    const double *rec = fac + (size_t)i*cre_rec_doubles(s);
    ms_stage_rec<MS_CT>(rec, s, recl, tid);
    for (int r = wave; r < s; r += MS_CT/64) v[r*64 + lane] = on ? ms_pending(M, s, i, h, lo, m, r0, r, cc_, M.G[((size_t)i*s + r)*T + cc_]) : 0.0;
    __syncthreads();
    if (wave == 0) {
        ms_block_fwd(recl, s, B, v, lane);
        const double *LDt = recl + rowoff(s);
        if (on) for (int r = 0; r < s; r++) M.Z[((size_t)i*s + r)*T + cc_] = v[r*64 + lane]*LDt[SOLVE_LD*(r/6) + LD_ID + r % 6];
    }
    __syncthreads();
    // the neighbours' updates X_a w, X_c w: rows dealt to the waves; a row of X parked in the wave's row buffer (broadcast reads), the next one in flight
    const double *Xa = a >= 0 ? cr_blk(Ws.S, s, mmax, i, a) : nullptr, *Xc = c >= 0 ? cr_blk(Ws.S, s, mmax, c, i) : nullptr;
    auto rowp = [&](int r) -> const double * { const double *X = r < s ? Xa : Xc; return X ? X + (size_t)(r < s ? r : r - s)*s : nullptr; };
    double n0 = 0.0, n1 = 0.0;
    { const double *rp = wave < 2*s ? rowp(wave) : nullptr; n0 = (rp && lane < s) ? rp[lane] : 0.0; n1 = (rp && lane + 64 < s) ? rp[lane + 64] : 0.0; }
    for (int r = wave; r < 2*s; r += MS_CT/64) {
        if (lane < s + 2) rowb[lane] = n0;                      // (s may be below 64: the buffers of the other waves follow this one)
        if (lane + 64 < s + 2) rowb[lane + 64] = n1;
        { const int rn = r + MS_CT/64; const double *rp = rn < 2*s ? rowp(rn) : nullptr; n0 = (rp && lane < s) ? rp[lane] : 0.0; n1 = (rp && lane + 64 < s) ? rp[lane + 64] : 0.0; }
        wave_lds_fence();
        double acc0 = 0.0, acc1 = 0.0;
        for (int k = 0; k + 1 < s; k += 2) { acc0 = fma(rowb[k], v[k*64 + lane], acc0); acc1 = fma(rowb[k + 1], v[(k + 1)*64 + lane], acc1); }
        if (on) M.Cg[(((size_t)i*2 + (r < s ? 0 : 1))*s + (r < s ? r : r - s))*T + cc_] = rowp(r) ? acc0 + acc1 : 0.0;
        wave_lds_fence();
    }
}

// ---- the last block: forward and backward.  grid (1, column groups), 4 waves stage the factor, wave 0 solves.
__global__ __launch_bounds__(256) void k_ms_cre_root(Work W, Work Ws, int bw, int Pmax, const double *__restrict__ fac, MsBuf M) {
    extern __shared__ __attribute__((aligned(16))) double ms_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = ms_uni(tid >> 6), col = 64*blockIdx.y + lane, T = M.T; const bool on = col < T; const int cc_ = on ? col : 0;
    { const LmState *st_ = W.st; if (st_->done | st_->lin_done | st_->step_fail) return; }     // (a converged iterative solve: the launches the host still had in flight)
    const CrRange rg = cr_range(W, bw, Pmax);
    const int m = ms_uni(rg.m), lo = ms_uni(rg.lo), r0 = ms_uni(rg.r0), s = bw, B = s/6;
    if (m <= 0) return;
    double *recl = ms_smem, *v = recl + ((cre_rec_doubles(s) + 8) & ~(size_t)1);
    ms_stage_rec<256>(fac + (size_t)r0*cre_rec_doubles(s), s, recl, tid);
    for (int r = wave; r < s; r += 4) v[r*64 + lane] = on ? ms_pending(M, s, r0, 1 << 30, lo, m, -1, r, cc_, M.G[((size_t)r0*s + r)*T + cc_]) : 0.0;
    __syncthreads();
    if (wave > 0) return;
    const double *LDt = recl + rowoff(s);
    ms_block_fwd(recl, s, B, v, lane);
    for (int r = 0; r < s; r++) v[r*64 + lane] *= LDt[SOLVE_LD*(r/6) + LD_ID + r % 6];
    ms_block_back(recl, s, B, v, lane);
    if (on) for (int r = 0; r < s; r++) M.Xs[((size_t)r0*s + r)*T + cc_] = v[r*64 + lane];
    (void)Ws;
}

// ---- cyclic reduction, level h, backward.  grid (pivots, column groups), 8 waves.  LDS: factor record | v | x_a | x_c | a column buffer per wave.
__global__ __launch_bounds__(MS_CT) void k_ms_cre_back(Work W, Work Ws, int bw, int Pmax, int h, int kb, const double *__restrict__ fac, MsBuf M) {
    extern __shared__ __attribute__((aligned(16))) double ms_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = ms_uni(tid >> 6), col = 64*blockIdx.y + lane, T = M.T; const bool on = col < T; const int cc_ = on ? col : 0;
    { const LmState *st_ = W.st; if (st_->done | st_->lin_done | st_->step_fail) return; }     // (a converged iterative solve: the launches the host still had in flight)
    const CrRange rg = cr_range(W, bw, Pmax);
    const int m = ms_uni(rg.m), lo = ms_uni(rg.lo), s = bw, B = s/6, mmax = cr_mmax(W.ring, Pmax, W.ring_g);
    const int i = (2*(kb + (int)blockIdx.x) + 1)*h;
    if (i < lo || i >= m) return;
    const int a = i - h >= lo ? i - h : -1, c = i + h < m ? i + h : -1;
    double *recl = ms_smem, *v = recl + ((cre_rec_doubles(s) + 8) & ~(size_t)1), *xa = v + (size_t)s*64, *xc = xa + (size_t)s*64, *colb = xc + (size_t)s*64 + wave*2*(s + 2);
    ms_stage_rec<MS_CT>(fac + (size_t)i*cre_rec_doubles(s), s, recl, tid);
    for (int r = wave; r < s; r += MS_CT/64) {
        xa[r*64 + lane] = (on && a >= 0) ? M.Xs[((size_t)a*s + r)*T + cc_] : 0.0;
        xc[r*64 + lane] = (on && c >= 0) ? M.Xs[((size_t)c*s + r)*T + cc_] : 0.0; }
    __syncthreads();
    const double *Xa = a >= 0 ? cr_blk(Ws.S, s, mmax, i, a) : nullptr, *Xc = c >= 0 ? cr_blk(Ws.S, s, mmax, c, i) : nullptr;
    // (X_a^T x_a + X_c^T x_c)[r] = sum_t X_a[t][r] x_a[t] + X_c[t][r] x_c[t]: column r of both blocks parked in the wave's buffer, the next column in flight
    double na0 = 0.0, na1 = 0.0, nc0 = 0.0, nc1 = 0.0;
    auto colfetch = [&](int r) { const bool okr = r < s;
        na0 = (Xa && okr && lane < s) ? Xa[(size_t)lane*s + r] : 0.0; na1 = (Xa && okr && lane + 64 < s) ? Xa[(size_t)(lane + 64)*s + r] : 0.0;
        nc0 = (Xc && okr && lane < s) ? Xc[(size_t)lane*s + r] : 0.0; nc1 = (Xc && okr && lane + 64 < s) ? Xc[(size_t)(lane + 64)*s + r] : 0.0; };
    colfetch(wave);
    for (int r = wave; r < s; r += MS_CT/64) {
        if (lane < s + 2) { colb[lane] = na0; colb[s + 2 + lane] = nc0; }      // (s may be below 64: the buffers of the other waves follow this one)
        if (lane + 64 < s + 2) { colb[lane + 64] = na1; colb[s + 2 + lane + 64] = nc1; }
        const double z = on ? M.Z[((size_t)i*s + r)*T + cc_] : 0.0;
        colfetch(r + MS_CT/64);
        wave_lds_fence();
        double acc0 = z, acc1 = 0.0;
        for (int t = 0; t < s; t++) { acc0 = fma(-colb[t], xa[t*64 + lane], acc0); acc1 = fma(-colb[s + 2 + t], xc[t*64 + lane], acc1); }
        v[r*64 + lane] = acc0 + acc1;
        wave_lds_fence();
    }
    __syncthreads();
    if (wave > 0) return;
    ms_block_back(recl, s, B, v, lane);
    if (on) for (int r = 0; r < s; r++) M.Xs[((size_t)i*s + r)*T + cc_] = v[r*64 + lane];
}

// ---- interiors: v_q -= Lb_q^T x_left for every column block (no chain).  grid (Pmax, column groups), four waves: a lane keeps the solution of the
// separator on the left (bw values of its column) in registers, wave w corrects the blocks a + w, a + w + 4, ... of the interior, their coefficient
// records through LDS a block ahead.  (A workgroup per pose block re-read that solution for every block: 614 MB per launch at 5000 keyframes.)
#define BB_T 256
__global__ __launch_bounds__(BB_T) void k_ms_back_border(Work W, int bw, int Pmax, const double *__restrict__ Lb, MsBuf M) {
    __shared__ __attribute__((aligned(16))) double cfs[BB_T/64][6*CR_SMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = ms_uni(tid >> 6), col = 64*blockIdx.y + lane, T = M.T; const bool on = col < T; const int cc_ = on ? col : 0;
    { const LmState *st_ = W.st; if (st_->done | st_->lin_done | st_->step_fail) return; }     // (a converged iterative solve: the launches the host still had in flight)
    const int B = bw/6, nf = ms_uni(*W.nfree), p = blockIdx.x;
    if (nf <= 0 || p == 0) return;                             // (the first interior has no separator on its left)
    const BandpPart PT = bandp_part(nf, B, Pmax, p);
    const int a = ms_uni(PT.a), b = ms_uni(PT.b), P = ms_uni(PT.P), REC = bw*6;
    if (p >= P) return;
    double xl[CR_SMAX];
#pragma unroll
    for (int br = 0; br < CR_SMAX; br++) xl[br] = (on && br < bw) ? M.Xs[((size_t)(p - 1)*bw + br)*T + cc_] : 0.0;
    constexpr int NC = (6*CR_SMAX + 63)/64;
    double cl[NC], v6[6];
    auto fetch = [&](int q) {
#pragma unroll
        for (int u = 0; u < NC; u++) { const int e = lane + 64*u; cl[u] = (q < b && e < REC) ? Lb[(size_t)q*REC + e] : 0.0; }
#pragma unroll
        for (int k = 0; k < 6; k++) v6[k] = (on && q < b) ? M.V[(size_t)(6*q + k)*T + cc_] : 0.0;
    };
    fetch(a + wave);
    for (int q = a + wave; q < b; q += BB_T/64) {
        double *cf = cfs[wave];
        wave_lds_fence();                                       // (the previous block's reads)
#pragma unroll
        for (int u = 0; u < NC; u++) { const int e = lane + 64*u; if (e < REC) cf[e] = cl[u]; }
        double acc[6];
#pragma unroll
        for (int k = 0; k < 6; k++) acc[k] = v6[k];
        fetch(q + BB_T/64);
        wave_lds_fence();
#pragma unroll
        for (int br = 0; br < CR_SMAX; br++) {
            if (br < bw) { const v2d c0 = *(const v2d *)(cf + 6*br), c1 = *(const v2d *)(cf + 6*br + 2), c2 = *(const v2d *)(cf + 6*br + 4);
                acc[0] = fma(-c0.x, xl[br], acc[0]); acc[1] = fma(-c0.y, xl[br], acc[1]); acc[2] = fma(-c1.x, xl[br], acc[2]);
                acc[3] = fma(-c1.y, xl[br], acc[3]); acc[4] = fma(-c2.x, xl[br], acc[4]); acc[5] = fma(-c2.y, xl[br], acc[5]); } }
        if (on) {
#pragma unroll
            for (int k = 0; k < 6; k++) M.V[(size_t)(6*q + k)*T + cc_] = acc[k]; }
    }
}

// ---- interiors, backward.  grid (Pmax, column groups), one wave.  Right-looking: once x_R is known, the B blocks before it take L(R, q)^T x_R --
// the coefficients of a step are ONE row record (B x 36 contiguous doubles, staged a step ahead as in k_ms_fwd_int), the running right-hand sides of the
// window live in LDS.  (Left-looking -- x_q = v_q - sum_R L(R, q)^T x_R -- gathers its B blocks from B different records: 288 us per launch at 5000
// keyframes against 77 us of the forward kernel for the same arithmetic.)  The separator on the right comes first: its blocks are pivots whose
// solution is known.
__global__ __launch_bounds__(64) void k_ms_back_int(Work W, int bw, int Pmax, const double *__restrict__ Lrow, MsBuf M) {
    __shared__ __attribute__((aligned(16))) double vwin[MS_BMAX*6*64];
    __shared__ __attribute__((aligned(16))) double recb[2][MS_RECMAX + 32];
    const int lane = threadIdx.x, col = 64*blockIdx.y + lane, T = M.T; const bool on = col < T; const int cc_ = on ? col : 0;
    { const LmState *st_ = W.st; if (st_->done | st_->lin_done | st_->step_fail) return; }     // (a converged iterative solve: the launches the host still had in flight)
    const int B = bw/6, nf = ms_uni(*W.nfree);
    if (nf <= 0) return;
    const BandpPart PT = bandp_part(nf, B, Pmax, blockIdx.x);
    const int a = ms_uni(PT.a), b = ms_uni(PT.b), P = ms_uni(PT.P), p = blockIdx.x;
    if (p >= P) return;
    const int REC = bw*6, rtop = p < P - 1 ? b + B : b;
    constexpr int NL = (MS_RECMAX + 63)/64;
    double pre[NL], preld, tn[6];
    auto fetch = [&](int R) {                                   // operands of pivot R: its row record, its diagonal table, the block R - B entering the window (or x of a separator block)
        const int nv = 36*min(B, R - a);
        const double *rec = Lrow + (size_t)R*REC;
#pragma unroll
        for (int k = 0; k < NL; k++) { const int e = lane + 64*k; pre[k] = e < nv ? rec[e] : 0.0; }
        preld = (lane < 22 && R < b) ? W.LDbuf[32*(size_t)R + lane] : 0.0;
        const int qn = R - B;
#pragma unroll
        for (int k = 0; k < 6; k++) tn[k] = (on && qn >= a) ? M.V[(size_t)(6*qn + k)*T + cc_] : 0.0;
    };
    // the window before the first pivot: blocks rtop - 1 .. rtop - B (interior blocks: v; separator blocks: x)
    for (int d = 0; d < B; d++) { const int q = rtop - 1 - d; double v6[6];
#pragma unroll
        for (int k = 0; k < 6; k++) v6[k] = (on && q >= a) ? (q >= b ? M.Xs[((size_t)p*bw + 6*(q - b) + k)*T + cc_] : M.V[(size_t)(6*q + k)*T + cc_]) : 0.0;
#pragma unroll
        for (int k = 0; k < 6; k++) vwin[((q % B + B) % B*6 + k)*64 + lane] = v6[k]; }
    fetch(rtop - 1);
    wave_lds_fence();
    for (int R = rtop - 1; R >= a; R--) {
        double *rb = recb[R & 1];
#pragma unroll
        for (int k = 0; k < NL; k++) { const int e = lane + 64*k; if (e < MS_RECMAX) rb[e] = pre[k]; }
        if (lane < 22) rb[MS_RECMAX + lane] = preld;
        double ent[6];
#pragma unroll
        for (int k = 0; k < 6; k++) ent[k] = tn[k];
        if (R > a) fetch(R - 1);
        wave_lds_fence();
        // x_R: the window's values of block R with the diagonal block solved (separator blocks are solutions already: their table is zero)
        double *wR = vwin + (R % B)*6*64 + lane;
        double x[6];
#pragma unroll
        for (int k = 0; k < 6; k++) x[k] = wR[k*64];
        const double *ld = rb + MS_RECMAX;
#pragma unroll
        for (int c = 4; c >= 0; c--)
#pragma unroll
            for (int k = c + 1; k < 6; k++) x[c] = fma(-ld[tri(k - 1) + c], x[k], x[c]);
        if (on) {
#pragma unroll
            for (int k = 0; k < 6; k++) M.X[(size_t)(6*R + k)*T + cc_] = x[k]; }
        // its slot goes to the entering block R - B
#pragma unroll
        for (int k = 0; k < 6; k++) wR[k*64] = ent[k];
        // the blocks before it (interior blocks only) take L(R, q)^T x_R
        const int qlo = max(a, R - B);
        for (int q = min(R - 1, b - 1); q >= qlo; q--) {
            const double *Lk = rb + (R - 1 - q)*36;            // L(6 R + r, 6 q + c) at [c*6 + r]
            double *wq = vwin + (q % B)*6*64 + lane;
            double t[6];
#pragma unroll
            for (int c = 0; c < 6; c++) t[c] = wq[c*64];
#pragma unroll
            for (int c = 0; c < 6; c++)
#pragma unroll
                for (int r = 0; r < 6; r++) t[c] = fma(-Lk[c*6 + r], x[r], t[c]);
#pragma unroll
            for (int c = 0; c < 6; c++) wq[c*64] = t[c];
        }
        wave_lds_fence();
    }
}
