// Dense solve of the reduced camera system S dp = -g (included by tsba.hip after the device structs).
#pragma once

// ---- dense solve of S dp = -g for the free poses: blocked (6x6) LDL^T in LDS, one workgroup of 16 waves.
// A = [S; g^T] is held as (n+1) rows; the right-hand side rides along as an extra panel row, so the forward
// substitution is part of the factorisation.  Per 6x6 block column: every thread factors the diagonal block redundantly
// in registers (no division chain: one reciprocal per pivot), one thread per row solves the panel, then 6 threads per
// 6x6 block apply the rank-6 trailing update.
#define SOLVE_THREADS 1024
__device__ __forceinline__ void ldl6(const double *A, int ld, double l[15], double d[6], double id[6], bool &bad) {
    // lower 6x6 at A (row stride ld) -> unit-lower l (packed rows: (1,0) (2,0) (2,1) (3,0) ...), d, 1/d
    double a[21];
#pragma unroll
    for (int r = 0; r < 6; r++)
#pragma unroll
        for (int c = 0; c <= r; c++) a[r*(r+1)/2 + c] = A[(size_t)r*ld + c];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        double dc = a[c*(c+1)/2 + c];
#pragma unroll
        for (int k = 0; k < c; k++) dc -= l[c*(c-1)/2 + k]*l[c*(c-1)/2 + k]*d[k];
        if (!(dc > 0.0)) { bad = true; dc = 1.0; }
        d[c] = dc; id[c] = 1.0/dc;
#pragma unroll
        for (int r = c + 1; r < 6; r++) {
            double v = a[r*(r+1)/2 + c];
#pragma unroll
            for (int k = 0; k < c; k++) v -= l[r*(r-1)/2 + k]*l[c*(c-1)/2 + k]*d[k];
            l[r*(r-1)/2 + c] = v*id[c];
        }
    }
}
typedef double v4d __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double rcp_nr(double d) {        // v_rcp_f64 + two Newton steps (d > 0, normal range)
    double x = __builtin_amdgcn_rcp(d);
    double e = fma(-d, x, 1.0); x = fma(x, e, x);
    e = fma(-d, x, 1.0); x = fma(x, e, x);
    return x;
}
__device__ __forceinline__ double readlane_f64(double v, int src) {   // src must be wave-uniform
    int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
// Schedule per block column jb (two barriers):
//   wave 0 ("D"):      applies panel jb-1 to the 6x6 diagonal block jb, factors it (LDL^T in registers)       | concurrently
//   waves 1,2 ("P"):   apply panel jb-1 to the rest of block column jb (one row per lane)                     | with
//   waves 3..15 ("T"): trailing update of the columns >= jb+1 with panel jb-1 on the matrix cores             | each other
//   -- barrier --      P: panel jb (x L^T = a, l_row = x D^-1), rows below + rhs row
//   -- barrier --
template <bool use_lds>
__global__ __launch_bounds__(SOLVE_THREADS) void k_solve(Work W) {
    LmState *st = W.st;
    if (st->done) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long T0 = clock64();
    const int nfree = *W.nfree, n = 6*nfree, Nmax = W.N;
    const int ld = use_lds ? (n | 1) : Nmax;
    // separate instantiations keep LDS accesses as ds_* instructions (a runtime-selected pointer would go through FLAT)
    auto sel = [&](auto lds_ptr, double *glob) { if constexpr (use_lds) return lds_ptr; else return glob; };
    auto A = sel(smem, W.S);                                            // rows 0..n, row n = right-hand side g
    auto LD = sel(smem + (n + 1)*ld, W.LDbuf);                  // per block: 15 l, 6 d, 6 1/d (stride 32)
    if (use_lds) {
        for (int r = tid >> 5; r < n; r += SOLVE_THREADS/32)
            for (int cidx = tid & 31; cidx <= r; cidx += 32) A[r*ld + cidx] = W.S[r*Nmax + cidx];
    }
    for (int k = tid; k < n; k += SOLVE_THREADS) A[n*ld + k] = W.g[k];
    __shared__ int fail;
    if (tid == 0) fail = st->step_fail;
    __syncthreads();
    long long T1 = clock64();
    long long tw = 0, tb1 = 0, tp = 0, tb2 = 0, tx;
    for (int jb = 0; jb < nfree; jb++) {
        const int j0 = 6*jb, R0 = j0 + 6, p0 = j0 - 6;
        tx = clock64();
        if (wave == 0) {
            if (jb > 0 && lane < 36) {       // diagonal block jb -= Lp D Lp^T of panel jb-1 (one entry per lane)
                const int r = lane/6, c = lane - 6*r;        // full 6x6 (the upper half is never read)
                double v = 0.0;
#pragma unroll
                for (int k = 0; k < 6; k++) v += A[(j0 + r)*ld + p0 + k]*LD[32*(jb - 1) + 15 + k]*A[(j0 + c)*ld + p0 + k];
                A[(j0 + r)*ld + j0 + c] -= v;
            }
            if (!fail) {
                double a[21], l[15], d[6], id[6]; bool bad = false;
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = 0; c <= r; c++) a[r*(r+1)/2 + c] = A[(j0 + r)*ld + j0 + c];
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double dc = a[c*(c+1)/2 + c];
#pragma unroll
                    for (int k = 0; k < c; k++) dc -= l[c*(c-1)/2 + k]*l[c*(c-1)/2 + k]*d[k];
                    if (!(dc > 0.0)) { bad = true; dc = 1.0; }
                    d[c] = dc; id[c] = rcp_nr(dc);
#pragma unroll
                    for (int r = c + 1; r < 6; r++) {
                        double v = a[r*(r+1)/2 + c];
#pragma unroll
                        for (int k = 0; k < c; k++) v -= l[r*(r-1)/2 + k]*l[c*(c-1)/2 + k]*d[k];
                        l[r*(r-1)/2 + c] = v*id[c];
                    }
                }
                if (lane == 0) {
                    if (bad) { fail = 1; st->step_fail = 1; }
                    auto o = LD + 32*jb;
#pragma unroll
                    for (int k = 0; k < 15; k++) o[k] = l[k];
#pragma unroll
                    for (int k = 0; k < 6; k++) { o[15 + k] = d[k]; o[21 + k] = id[k]; }
                }
            }
        } else if (wave <= 2) {
            if (jb > 0) {                    // rest of block column jb (rows j0+6..n) -= panel jb-1 contribution
                double dprev[6], Lk[36];
#pragma unroll
                for (int k = 0; k < 6; k++) dprev[k] = LD[32*(jb - 1) + 15 + k];
#pragma unroll
                for (int c = 0; c < 6; c++)
#pragma unroll
                    for (int k = 0; k < 6; k++) Lk[c*6 + k] = A[(j0 + c)*ld + p0 + k];
                for (int i = R0 + (wave - 1)*64 + lane; i <= n; i += 128) {
                    auto row = A + i*ld;
                    double y[6];
#pragma unroll
                    for (int k = 0; k < 6; k++) y[k] = row[p0 + k]*dprev[k];
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        double v = 0.0;
#pragma unroll
                        for (int k = 0; k < 6; k++) v += y[k]*Lk[c*6 + k];
                        row[j0 + c] -= v;
                    }
                }
            }
        } else if (jb > 0) {
            // trailing update with panel jb-1 on columns >= j0+6, rows >= j0+6 and the rhs row (block column jb is D/P work)
            const int C0 = j0 + 6;
            const int mr = n - C0 + 1, mc = n - C0;
            if (mc > 0) {
                const int ntr = (mr + 15) >> 4, ntc = (mc + 15) >> 4;
                const int lr = lane & 15, lk = lane >> 4;
                const double dk0 = LD[32*(jb - 1) + 15 + lk], dk1 = (4 + lk < 6) ? LD[32*(jb - 1) + 15 + 4 + lk] : 0.0;
                int t = wave - 3;
                for (int ti = 0; ti < ntr; ti++) for (int tj = 0; tj <= ti && tj < ntc; tj++) {
                    if (t-- != 0) continue;
                    t = SOLVE_THREADS/64 - 4;               // this wave's next tile: 13 tiles further
                    // unconditional, index-clamped loads (rows / columns past the edge only feed outputs that are never stored);
                    // only the K padding (k = 6, 7) must be exact zeros
                    const int arow = min(C0 + 16*ti + lr, n), bcol = min(C0 + 16*tj + lr, n - 1);
                    const int k1 = min(4 + lk, 5);
                    double a0 = -A[arow*ld + p0 + lk], a1 = -A[arow*ld + p0 + k1];
                    double b0 = A[bcol*ld + p0 + lk]*dk0, b1 = A[bcol*ld + p0 + k1]*dk1;
                    if (lk >= 2) { a1 = 0.0; b1 = 0.0; }
                    v4d c;
                    const int ccol = C0 + 16*tj + lr, ccol_c = min(ccol, n - 1);
                    bool ok[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int crow = C0 + 16*ti + lk + 4*r;
                        ok[r] = crow <= n && ccol < n && (ccol <= crow);
                        c[r] = A[min(crow, n)*ld + ccol_c];
                    }
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int crow = C0 + 16*ti + lk + 4*r;
                        if (ok[r]) A[crow*ld + ccol] = c[r];
                    }
                }
            }
        }
        tw += clock64() - tx; tx = clock64();
        __syncthreads();                       // block column jb up to date, diagonal block jb factored
        tb1 += clock64() - tx; tx = clock64();
        if (fail) break;
        if (wave == 1 || wave == 2) {          // panel jb: rows below the diagonal block and the rhs row
            double l[15], id[6];
#pragma unroll
            for (int k = 0; k < 15; k++) l[k] = LD[32*jb + k];
#pragma unroll
            for (int k = 0; k < 6; k++) id[k] = LD[32*jb + 21 + k];
            for (int i = R0 + (wave - 1)*64 + lane; i <= n; i += 128) {
                auto row = A + i*ld + j0;
                double x[6];
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double v = row[c];
#pragma unroll
                    for (int k = 0; k < c; k++) v -= x[k]*l[c*(c-1)/2 + k];
                    x[c] = v;
                }
#pragma unroll
                for (int c = 0; c < 6; c++) row[c] = x[c]*id[c];
            }
        }
        tp += clock64() - tx; tx = clock64();
        __syncthreads();                       // panel jb complete
        tb2 += clock64() - tx;
    }
    __syncthreads();
    long long T2 = clock64();
    if (fail) { for (int k = tid; k < Nmax; k += SOLVE_THREADS) W.dp[k] = 0.0; return; }
    // back substitution L^T x = z (z = D^-1 L^-1 g sits in row n): wave 0, solution kept in registers (rows lane, lane+64, ...)
    auto rhs = A + n*ld;
    if (wave == 0) {
        if (n <= 128) {
            double z0 = lane < n ? rhs[lane] : 0.0, z1 = lane + 64 < n ? rhs[lane + 64] : 0.0;
            for (int jb = nfree - 1; jb >= 0; jb--) {
                const int j0 = 6*jb;
                auto l = LD + 32*jb;
                double x[6];
#pragma unroll
                for (int c = 0; c < 6; c++) { int r = j0 + c; x[c] = r < 64 ? readlane_f64(z0, r) : readlane_f64(z1, r - 64); }
#pragma unroll
                for (int c = 4; c >= 0; c--) {
#pragma unroll
                    for (int k = c + 1; k < 6; k++) x[c] -= l[k*(k-1)/2 + c]*x[k];
                }
#pragma unroll
                for (int c = 0; c < 6; c++) { if (lane == ((j0 + c) & 63)) { if (j0 + c < 64) z0 = x[c]; else z1 = x[c]; } }
                if (lane < j0) {
                    double v = 0.0;
#pragma unroll
                    for (int c = 0; c < 6; c++) v += A[(j0 + c)*ld + lane]*x[c];
                    z0 -= v;
                }
                if (lane + 64 < j0) {
                    double v = 0.0;
#pragma unroll
                    for (int c = 0; c < 6; c++) v += A[(j0 + c)*ld + lane + 64]*x[c];
                    z1 -= v;
                }
            }
            if (lane < n) rhs[lane] = z0;
            if (lane + 64 < n) rhs[lane + 64] = z1;
        } else {
            for (int jb = nfree - 1; jb >= 0; jb--) {
                const int j0 = 6*jb;
                auto l = LD + 32*jb;
                double x[6];
#pragma unroll
                for (int c = 5; c >= 0; c--) {
                    double v = rhs[j0 + c];
#pragma unroll
                    for (int k = c + 1; k < 6; k++) v -= l[k*(k-1)/2 + c]*x[k];
                    x[c] = v;
                }
                for (int k = lane; k < j0; k += 64) {
                    double v = 0.0;
#pragma unroll
                    for (int c = 0; c < 6; c++) v += A[(j0 + c)*ld + k]*x[c];
                    rhs[k] -= v;
                }
                if (lane == 0) {
#pragma unroll
                    for (int c = 0; c < 6; c++) rhs[j0 + c] = x[c];
                }
                __threadfence_block();
            }
        }
    }
    __syncthreads();
    if (tid == 0) { long long T3 = clock64(); W.dbg[0] = T1 - T0; W.dbg[1] = T2 - T1; W.dbg[2] = T3 - T2; W.dbg[3] = 0; W.dbg[4] = 0; W.dbg[5] = 0; W.dbg[6] = nfree; }
    if (lane == 0 && (wave == 0 || wave == 1 || wave == 3)) { int o = 8 + 4*(wave == 0 ? 0 : wave == 1 ? 1 : 2); W.dbg[o] = tw; W.dbg[o+1] = tb1; W.dbg[o+2] = tp; W.dbg[o+3] = tb2; }
    for (int a = tid; a < W.n_kf; a += SOLVE_THREADS) {
        int ia = W.fidx[a];
#pragma unroll
        for (int k = 0; k < 6; k++) W.dp[6*a + k] = ia >= 0 ? -rhs[6*ia + k] : 0.0;
    }
}

