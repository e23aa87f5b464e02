// Dense solve of the reduced camera system of a small window, second schedule: the factorisation of the NEXT diagonal block runs ahead of the
// panel.  (part of the single translation unit tsba.hip: included after tsba_solve.h, whose layout, LD table and back-substitution it shares)
//
// k_solve_t (tsba_solve.h) walks the 17 pose blocks of a 20-keyframe window with two panel waves that do, per block and one after the other,
// look-ahead -> 6x6 LDL^T -> panel solve (2900 cycles per block by cycle stamps, of which the in-register LDL^T 1300), while ten waves update the
// trailing matrix.  Here the chain is cut to what really is sequential:
//   wave 0  "D"  carries the dependent chain only.  At step jb it owns the six rows of block jb + 1: applies panel jb - 1 to them (columns of
//                blocks jb and jb + 1), solves them against the factor of block jb (in its registers since the last step) -- that is
//                L(jb+1, jb) --, applies that to the diagonal block (jb+1, jb+1) and factors it: the factor of block jb + 1 is in the LD table
//                when the step ends.  Every lane holds the whole 6x6 factor (the rows meet through 36 doubles of scratch: no cross-wave hand-off).
//   wave 1  "P"  the panel proper, one step behind the factor: every other row below block jb + 1 (and the right-hand-side row), look-ahead of
//                panel jb - 1 on its own row, solve against the factor of block jb from the LD table.  No factorisation, no hand-off.
//   waves 2..11 "T"  trailing update with panel jb - 1 on the matrix cores as before, except the diagonal block (jb+1, jb+1), which D owns in
//                this step; tile indices on the scalar unit, row offsets from a table (the packed triangle's offsets cost a 32-bit multiply each:
//                quarter rate), and the waves that share a SIMD with D / P take tiles last.
// One barrier per step, as before.  Same arithmetic per entry as k_solve_t up to the order in which the two panel updates reach the block
// column jb + 1 (panel jb - 1 first, then jb: the same order), so the factor is bit-identical to k_solve_t's.
#pragma once

static size_t solve_la_lds_doubles(int N) { return solve_lds_doubles(N) + (size_t)(N + 6)/2 + 2; }

__global__ __launch_bounds__(SOLVE_THREADS) void k_solve_la(Work W, int tile_order) {
    LmState *st = W.st;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = SOLVE_THREADS/64, NT = NW - 2;
    const int Nmax = W.N;
    double *A = smem;
    const int neMax = tri(Nmax);
    const size_t ldS = (size_t)W.ldS;
    const double *Sb = W.S;
    const int rmax = Nmax - 1;
    double v[12]; int er[12], ec[12];
#pragma unroll
    for (int u = 0; u < 12; u++) {                               // first round of loads in flight together with the solver state
        const int e = min(u*SOLVE_THREADS + tid, neMax - 1);
        er[u] = tri_row(e); ec[u] = e - tri(er[u]);
        v[u] = Sb[(size_t)min(er[u], rmax)*ldS + min(ec[u], rmax)];
    }
    const int done = st->done, nfree = *W.nfree, sfail = st->step_fail;
    if (done) return;
    const int n = 6*nfree, ne = tri(n);
    double *LD = A + rowoff(n + 1) + 16;
    double *scr = LD + SOLVE_LD*nfree;                           // [0..35] rows of L(jb+1, jb) (D's hand-over to itself and to the next step), [36..71] rows of the diagonal block
    int *rofs = (int *)(scr + 36*SOLVE_PW + 8);                  // row offsets of the padded packed triangle
#pragma unroll
    for (int u = 0; u < 12; u++) if (u*SOLVE_THREADS + tid < ne) A[rowoff(er[u]) + ec[u]] = v[u];
    for (int base = 12*SOLVE_THREADS; base < ne; base += 12*SOLVE_THREADS) {
#pragma unroll
        for (int u = 0; u < 12; u++) {
            const int e = min(base + u*SOLVE_THREADS + tid, ne - 1);
            er[u] = tri_row(e); ec[u] = e - tri(er[u]);
            v[u] = Sb[(size_t)er[u]*ldS + ec[u]];
        }
#pragma unroll
        for (int u = 0; u < 12; u++) if (base + u*SOLVE_THREADS + tid < ne) A[rowoff(er[u]) + ec[u]] = v[u];
    }
    for (int k = tid; k < n; k += SOLVE_THREADS) A[rowoff(n) + k] = W.g[k];
    for (int k = tid; k <= n + 1; k += SOLVE_THREADS) rofs[k] = rowoff(k);
    if (tid == 0) fail = sfail;
    __syncthreads();
    if (nfree == 0 || fail) { for (int k = tid; k < Nmax; k += SOLVE_THREADS) W.dp[k] = 0.0; return; }
    // ---- prologue: the factor of block 0 (wave D; it stays in D's registers)
    double fl[15], fid[6];                                       // D: unit factor and inverse pivots of the block the panel is being solved against (d goes through the LD table)
    if (wave == 0) {
        double s[21], fd[6]; bool bad = false;
        {
            double t[36];
#pragma unroll
            for (int r = 0; r < 6; r++) ld6(A + rowoff(r), t + 6*r);      // (row r holds r + 1 entries; what follows them is not used)
#pragma unroll
            for (int r = 0; r < 6; r++)
#pragma unroll
                for (int c = 0; c <= r; c++) s[tri(r) + c] = t[6*r + c];
        }
        ldl6(s, fl, fd, fid, bad);
        if (lane == 0) {
            double *o = LD;
#pragma unroll
            for (int k = 0; k < 15; k++) o[k] = fl[k];
            st6(o + LD_D, fd); st6(o + LD_ID, fid);
            if (bad) { fail = 1; st->step_fail = 1; }
        }
    }
    __syncthreads();
    // the update waves that share a SIMD with D (waves 4, 8) or P (5, 9) take tiles last: position of wave w in the hand-out order
    const int tpos = !tile_order ? wave - 2 : (wave == 2 ? 0 : wave == 3 ? 1 : wave == 6 ? 2 : wave == 7 ? 3 : wave == 10 ? 4 : wave == 11 ? 5 : wave == 5 ? 6 : wave == 9 ? 7 : wave == 4 ? 8 : 9);
    for (int jb = 0; jb < nfree && !fail; jb++) {
        const int j0 = 6*jb, R0 = j0 + 6, p0 = j0 - 6;
        const bool next = jb + 1 < nfree;                         // there is a block jb + 1 for D to factor
#ifdef TSBA_SOLVE_STAMPS
        const long long tx0_ = clock64(); long long sa_ = tx0_, sb_ = tx0_;
#endif
        if (wave == 0) {
            if (next) {
                const int rr = min(lane, 5);
                const double *row = A + rofs[R0 + rr];
                double x[6], sd[6];
                ld6(row + j0, x); ld6(row + R0, sd);
                if (jb > 0) {                                     // panel jb - 1 on both block columns of these rows
                    double y[6], dprev[6];
                    ld6(row + p0, y); ld6(LD + SOLVE_LD*(jb - 1) + LD_D, dprev);
#pragma unroll
                    for (int k = 0; k < 6; k++) y[k] *= dprev[k];
#pragma unroll
                    for (int c = 0; c < 6; c++) {                 // block column jb: against L(jb, jb-1), D's rows of the last step (still in scratch)
                        double Lc[6]; ld6(scr + 6*c, Lc);
                        double v0 = y[0]*Lc[0], v1 = y[1]*Lc[1];
                        v0 = fma(y[2], Lc[2], v0); v1 = fma(y[3], Lc[3], v1);
                        v0 = fma(y[4], Lc[4], v0); v1 = fma(y[5], Lc[5], v1);
                        x[c] -= v0 + v1;
                    }
#pragma unroll
                    for (int c = 0; c < 6; c++) {                 // block column jb + 1 (the diagonal block): against the rows of block jb + 1 themselves
                        double Lc[6]; ld6(A + rofs[R0 + c] + p0, Lc);
                        double v0 = y[0]*Lc[0], v1 = y[1]*Lc[1];
                        v0 = fma(y[2], Lc[2], v0); v1 = fma(y[3], Lc[3], v1);
                        v0 = fma(y[4], Lc[4], v0); v1 = fma(y[5], Lc[5], v1);
                        sd[c] -= v0 + v1;
                    }
                }
                // x L^T = a against the factor of block jb: x D = the solved row before scaling (what the trailing product multiplies by), xs = L(jb+1, jb)
#pragma unroll
                for (int c = 0; c < 5; c++)
#pragma unroll
                    for (int q = c + 1; q < 6; q++) x[q] = fma(-x[c], fl[tri(q - 1) + c], x[q]);
                double xs[6];
#pragma unroll
                for (int c = 0; c < 6; c++) xs[c] = x[c]*fid[c];
                if (lane < 6) { st6(A + rofs[R0 + lane] + j0, xs); st6(scr + 6*lane, xs); }
                wave_lds_fence();
#ifdef TSBA_SOLVE_STAMPS
                sa_ = clock64();
#endif
#pragma unroll
                for (int c = 0; c < 6; c++) {                     // panel jb on the diagonal block:  sd[c] -= (row r of L D) . (row c of L)
                    double Lc[6]; ld6(scr + 6*c, Lc);
                    double v0 = x[0]*Lc[0], v1 = x[1]*Lc[1];
                    v0 = fma(x[2], Lc[2], v0); v1 = fma(x[3], Lc[3], v1);
                    v0 = fma(x[4], Lc[4], v0); v1 = fma(x[5], Lc[5], v1);
                    sd[c] -= v0 + v1;
                }
                if (lane < 6) st6(scr + 36 + 6*lane, sd);
                wave_lds_fence();
                double s[21], fd[6]; bool bad = false;
                {
                    double t[36];
#pragma unroll
                    for (int r = 0; r < 6; r++) ld6(scr + 36 + 6*r, t + 6*r);
#pragma unroll
                    for (int r = 0; r < 6; r++)
#pragma unroll
                        for (int c = 0; c <= r; c++) s[tri(r) + c] = t[6*r + c];
                }
#ifdef TSBA_SOLVE_STAMPS
                sb_ = clock64();
#endif
                ldl6(s, fl, fd, fid, bad);
                if (lane == 0) {
                    double *o = LD + SOLVE_LD*(jb + 1);
#pragma unroll
                    for (int k = 0; k < 15; k++) o[k] = fl[k];
                    st6(o + LD_D, fd); st6(o + LD_ID, fid);
                    if (bad) { fail = 1; st->step_fail = 1; }
                }
            }
        } else if (wave == 1) {
            // the panel: rows below block jb + 1 (all rows below block jb when there is no block jb + 1: the right-hand-side row)
            const double *ldj = LD + SOLVE_LD*jb;
            double l[15], id[6], dp_[6];
            {
                double t[16];
#pragma unroll
                for (int k = 0; k < 8; k++) { const v2d q = ((const v2d *)ldj)[k]; t[2*k] = q.x; t[2*k + 1] = q.y; }
#pragma unroll
                for (int k = 0; k < 15; k++) l[k] = t[k];
                ld6(ldj + LD_ID, id);
            }
            if (jb > 0) ld6(LD + SOLVE_LD*(jb - 1) + LD_D, dp_);
            const double *Lkp = A + rofs[j0] + p0;               // L(jb, jb-1): row c at Lkp + (rofs[j0 + c] - rofs[j0]); read per round (at most two rounds per step), not held:
            for (int i = (next ? R0 + 6 : R0) + lane; i <= n; i += 64) {      // the kernel's register budget is set by D's factor, which lives across the steps
                double *row = A + rofs[i];
                double a[6];
                ld6(row + j0, a);
                if (jb > 0) {
                    double y[6];
                    ld6(row + p0, y);
#pragma unroll
                    for (int k = 0; k < 6; k++) y[k] *= dp_[k];
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        double Lc[6]; ld6(Lkp + (rofs[j0 + c] - rofs[j0]), Lc);
                        double v0 = y[0]*Lc[0], v1 = y[1]*Lc[1];
                        v0 = fma(y[2], Lc[2], v0); v1 = fma(y[3], Lc[3], v1);
                        v0 = fma(y[4], Lc[4], v0); v1 = fma(y[5], Lc[5], v1);
                        a[c] -= v0 + v1;
                    }
                }
#pragma unroll
                for (int c = 0; c < 5; c++)
#pragma unroll
                    for (int q = c + 1; q < 6; q++) a[q] = fma(-a[c], l[tri(q - 1) + c], a[q]);
#pragma unroll
                for (int c = 0; c < 6; c++) a[c] *= id[c];
                st6(row + j0, a);
            }
        } else {
            // trailing update with panel jb - 1: rows >= R0 (incl. the rhs row n), columns R0 .. n - 1, minus the diagonal block D owns
            const int mr = n - R0 + 1, mc = n - R0;
            if (jb > 0 && mc > 0) {
                const double *ldp = LD + SOLVE_LD*(jb - 1);
                const int ntr = (mr + 15) >> 4, ntc = (mc + 15) >> 4, ntile = tri(ntr);
                const int lr = lane & 15, lk = lane >> 4;
                const int k1 = min(4 + lk, 5);
                const double dk0 = ldp[LD_D + lk], dk1 = lk < 2 ? ldp[LD_D + 4 + lk] : 0.0;
                int ti = 0, tb = 0;                               // tile t = tri(ti) + tj, tb = tri(ti): advanced on the scalar unit
                for (int t = tpos; t < ntile; t += NT) {
                    while (tb + ti + 1 <= t) { tb += ti + 1; ti++; }
                    const int tj = t - tb;
                    if (tj >= ntc) continue;
                    const int arow = rofs[min(R0 + 16*ti + lr, n)] + p0, brow = rofs[min(R0 + 16*tj + lr, n - 1)] + p0;
                    double a0 = -A[arow + lk], a1 = -A[arow + k1];
                    double b0 = A[brow + lk]*dk0, b1 = A[brow + k1]*dk1;
                    if (lk >= 2) { a1 = 0.0; b1 = 0.0; }
                    const int ccol = R0 + 16*tj + lr;
                    v4d c; int ci[4]; bool ok[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int crow = R0 + 16*ti + lk + 4*r;
                        ok[r] = crow <= n && ccol <= crow && ccol < n && crow >= R0 + 6;      // (rows of block jb + 1: only their diagonal block lies left of the diagonal, and D owns it)
                        ci[r] = rofs[min(crow, n)] + min(ccol, min(crow, n - 1));
                        c[r] = A[ci[r]];
                    }
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; r++) if (ok[r]) A[ci[r]] = c[r];
                }
            }
            if (wave == NW - 1) {                                // inverse of the unit-lower factor of block jb (for the back-substitution)
                const double *ldj = LD + SOLVE_LD*jb;
                double l[15], m[15];
#pragma unroll
                for (int k = 0; k < 15; k++) l[k] = ldj[k];
                inv_unit_lower6(l, m);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 15; k++) LD[SOLVE_LD*jb + LD_M + k] = m[k];
                }
            }
        }
#ifdef TSBA_SOLVE_STAMPS
        if (lane == 0 && jb < 32) { const long long se_ = clock64();
            if (wave == 0) { ts_step_stamps[jb] = sa_ - tx0_; ts_step_stamps[32 + jb] = sb_ - sa_; ts_step_stamps[64 + jb] = se_ - sb_; }
            if (wave == 2) ts_step_stamps[96 + jb] = se_ - tx0_;
            if (wave == 1) W.dbg[44 + (jb & 3)] = se_ - tx0_; }
#endif
        __syncthreads();
    }
    if (fail) { for (int k = tid; k < Nmax; k += SOLVE_THREADS) W.dp[k] = 0.0; return; }
    double *rhs = A + rowoff(n);
    if (wave == 0) solve_backsub_wave(A, LD, n, nfree, lane);
    __syncthreads();
    for (int a = tid; a < W.n_kf; a += SOLVE_THREADS) {
        int ia = W.fidx[a];
#pragma unroll
        for (int k = 0; k < 6; k++) W.dp[6*a + k] = ia >= 0 ? -rhs[6*ia + k] : 0.0;
    }
}
