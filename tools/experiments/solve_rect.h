// Experiment (round 2, not part of the product build): the small-window LDS solver with a square (same stride for every row) layout.
// C4 window (17 free poses): 34.95 us per launch with ten update waves, 37.0 us with the update tiles on SIMD 2 and 3 only; k_solve_t: 33.5 us.

// ---- k_solve_r: the same solver with every row of the matrix at the same stride (the upper triangle is dead space), for windows whose
// square fits the LDS (N <= 138: the reference's local-BA window of 20 keyframes is 120 rows).  What round 2 learnt on the cyclic-reduction
// kernels (tsba_bandcre.h) and why it matters here: (1) with the packed triangle two thirds of an update tile's instructions are index
// arithmetic (rowoff, clamps, masks, a square root to find the tile) -- here an address is one multiply-add, whole tiles take an
// unmasked path and the tile indices come from the scalar unit; (2) fp64 MFMA runs on the vector unit's double-precision pipe, so an
// update wave on the SIMD of a panel wave delays the panel chain directly -- the update tiles go to the waves of SIMD 2 and 3 only
// (wave w runs on SIMD w mod 4; the panel waves are 0 and 1): with the leaner tiles six waves keep up.  The load maps threads to
// (row, column chunk) instead of inverting the triangular index per element.
__host__ __device__ __forceinline__ int solve_r_stride(int N) { return (N & 3) == 2 ? N : N + 2; }     // doubles; = 2 mod 4: b128 rows of 16 lanes hit 64 different banks
static size_t solve_r_lds_doubles(int N) { return (size_t)(N + 1)*solve_r_stride(N) + 16 + (size_t)SOLVE_LD*(N/6) + 36*SOLVE_PW + 8; }
#define SOLVE_R_LOADS 20                    // (row, 64-column chunk) pairs per wave: 176 pairs at N = 120, 222 at N = 138, twelve waves
// pair p of the lower triangle's rows in 64-column chunks: rows < 64 have one chunk, rows 64 .. 127 two, rows >= 128 three (scalar unit)
__host__ __device__ __forceinline__ void solve_r_pair(int p, int &r, int &ch) {
    if (p < 64) { r = p; ch = 0; }
    else if (p < 192) { const int q = p - 64; r = 64 + (q >> 1); ch = q & 1; }
    else { const int q = p - 192, t = (q*21846) >> 16; r = 128 + t; ch = q - 3*t; }
}
static int solve_r_pairs(int N) { return std::min(N, 64) + 2*std::max(0, std::min(N, 128) - 64) + 3*std::max(0, N - 128); }

__global__ __launch_bounds__(SOLVE_THREADS) void k_solve_r(Work W) {
    LmState *st = W.st;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = SOLVE_THREADS/64;
    const int Nmax = W.N, sst = solve_r_stride(Nmax);
    double *A = smem;
    const size_t ldS = (size_t)W.ldS;
    // every address is known from the launch arguments: the loads of S go out together with the loads of the solver state
    double v[SOLVE_R_LOADS];
#pragma unroll
    for (int u = 0; u < SOLVE_R_LOADS; u++) {
        int r, ch; solve_r_pair(wave + NW*u, r, ch);
        const int c = 64*ch + lane;
        v[u] = (r < Nmax && c <= r) ? W.S[(size_t)r*ldS + c] : 0.0;
    }
    const double gv = tid < Nmax ? W.g[tid] : 0.0;
    const int done = st->done, nfree = *W.nfree, sfail = st->step_fail;
    if (done) return;
    const int n = 6*nfree;
    double *LD = A + (size_t)(Nmax + 1)*sst + 16;
    double *scr = LD + SOLVE_LD*(Nmax/6);
#pragma unroll
    for (int u = 0; u < SOLVE_R_LOADS; u++) {
        int r, ch; solve_r_pair(wave + NW*u, r, ch);
        const int c = 64*ch + lane;
        if (r < n && c <= r) A[r*sst + c] = v[u];
    }
    if (tid < n) A[n*sst + tid] = gv;
    if (tid == 0) fail = sfail;
    __syncthreads();
    for (int jb = 0; jb < nfree && !fail; jb++) {
        const int j0 = 6*jb, R0 = j0 + 6, p0 = j0 - 6;
        if (wave < SOLVE_PW) {
            double Lk[36], dprev[6];
            if (jb > 0) {
                ld6(LD + SOLVE_LD*(jb - 1) + LD_D, dprev);
#pragma unroll
                for (int c = 0; c < 6; c++) ld6(A + (j0 + c)*sst + p0, Lk + 6*c);
            }
            auto load_row = [&](int i, double a[6]) {           // row i of block column jb with panel jb-1 applied
                const double *row = A + i*sst;
                ld6(row + j0, a);
                if (jb > 0) {
                    double y[6];
                    ld6(row + p0, y);
#pragma unroll
                    for (int k = 0; k < 6; k++) y[k] *= dprev[k];
#pragma unroll
                    for (int c = 0; c < 6; c++) {
                        double v0 = y[0]*Lk[c*6], v1 = y[1]*Lk[c*6 + 1];
                        v0 = fma(y[2], Lk[c*6 + 2], v0); v1 = fma(y[3], Lk[c*6 + 3], v1);
                        v0 = fma(y[4], Lk[c*6 + 4], v0); v1 = fma(y[5], Lk[c*6 + 5], v1);
                        a[c] -= v0 + v1;
                    }
                }
            };
            const int i0 = lane < 6 ? j0 + lane : R0 + wave*SOLVE_PROWS + lane - 6;
            double a[6];
            load_row(min(i0, n), a);
            if (lane < 6) st6(scr + wave*36 + lane*6, a);
            wave_lds_fence();
            double s[21], l[15], d[6], id[6]; bool bad = false;
            {
                double t[36];
#pragma unroll
                for (int r = 0; r < 6; r++) ld6(scr + wave*36 + r*6, t + 6*r);
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int c = 0; c <= r; c++) s[tri(r) + c] = t[6*r + c];
            }
            ldl6(s, l, d, id, bad);
            if (wave == 0 && lane == 0) {
                double *o = LD + SOLVE_LD*jb;
#pragma unroll
                for (int k = 0; k < 15; k++) o[k] = l[k];
                st6(o + LD_D, d); st6(o + LD_ID, id);
                if (bad) { fail = 1; st->step_fail = 1; }
            }
            auto solve_row = [&](int i, double a[6]) {           // x L^T = a (right-looking: 5-deep chain), stored row = x D^-1
#pragma unroll
                for (int c = 0; c < 5; c++)
#pragma unroll
                    for (int q = c + 1; q < 6; q++) a[q] = fma(-a[c], l[tri(q - 1) + c], a[q]);
#pragma unroll
                for (int c = 0; c < 6; c++) a[c] *= id[c];
                st6(A + i*sst + j0, a);
            };
            if (lane >= 6) {
                if (i0 <= n) solve_row(i0, a);
                for (int i = i0 + SOLVE_PW*SOLVE_PROWS; i <= n; i += SOLVE_PW*SOLVE_PROWS) { load_row(i, a); solve_row(i, a); }
            }
            if (wave == 1 && jb == nfree - 1) {                  // inverse factor of the last block (the others: wave 4)
                double m[15];
                inv_unit_lower6(l, m);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 15; k++) LD[SOLVE_LD*jb + LD_M + k] = m[k];
                }
            }
        } else if (jb > 0) {
            const double *ldp = LD + SOLVE_LD*(jb - 1);
#ifndef SOLVE_R_ALLT
#define SOLVE_R_ALLT 0
#endif
            if (SOLVE_R_ALLT ? wave >= SOLVE_PW : (wave & 3) >= 2) {
                // trailing update with panel jb-1: rows >= R0 (incl. the rhs row n), columns R0..n-1; the six waves of SIMD 2 and 3
                const int uw = SOLVE_R_ALLT ? wave - SOLVE_PW : (wave >> 2)*2 + (wave & 1);      // 0 .. 5
                constexpr int NTW = SOLVE_R_ALLT ? NW - SOLVE_PW : 6;
                const int mr = n - R0 + 1, mc = n - R0;
                if (mc > 0) {
                    const int ntr = (mr + 15) >> 4, ntile = tri(ntr);
                    const int lr = lane & 15, lk = lane >> 4;
                    const int k1 = min(4 + lk, 5);
                    const double dk0 = ldp[LD_D + lk], dk1 = lk < 2 ? ldp[LD_D + 4 + lk] : 0.0;
                    for (int t = uw; t < ntile; t += NTW) {
                        const int ti = (t >= 1) + (t >= 3) + (t >= 6) + (t >= 10) + (t >= 15) + (t >= 21) + (t >= 28) + (t >= 36), tj = t - tri(ti);
                        const int rb = R0 + 16*ti, cb0 = R0 + 16*tj;
                        if (cb0 >= n) continue;
                        const bool full = rb + 15 <= n && cb0 + 15 < n && cb0 + 15 <= rb;          // (wave-uniform)
                        const double *pa = A + min(rb + lr, n)*sst + p0, *pb = A + min(cb0 + lr, n - 1)*sst + p0;
                        double a0 = -pa[lk], a1 = -pa[k1];
                        double b0 = pb[lk]*dk0, b1 = pb[k1]*dk1;
                        if (lk >= 2) { a1 = 0.0; b1 = 0.0; }
                        if (full) {
                            double *pc = A + (rb + lk)*sst + cb0 + lr;
                            v4d cv;
#pragma unroll
                            for (int r = 0; r < 4; r++) cv[r] = pc[4*r*sst];
                            cv = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, cv, 0, 0, 0);
                            cv = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, cv, 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 4; r++) pc[4*r*sst] = cv[r];
                        } else {
                            const int ccol = cb0 + lr;
                            v4d cv; int ci[4]; bool ok[4];
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const int crow = rb + lk + 4*r;
                                ok[r] = crow <= n && ccol <= crow && ccol < n;
                                ci[r] = min(crow, n)*sst + min(ccol, min(crow, n - 1));
                                cv[r] = A[ci[r]];
                            }
                            cv = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, cv, 0, 0, 0);
                            cv = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, cv, 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 4; r++) if (ok[r]) A[ci[r]] = cv[r];
                        }
                    }
                }
            }
            if (wave == (SOLVE_R_ALLT ? NW - 1 : 4)) {           // inverse of the unit-lower factor of block jb-1 (twenty instructions next to panel wave 0)
                double l[15], m[15];
#pragma unroll
                for (int k = 0; k < 15; k++) l[k] = ldp[k];
                inv_unit_lower6(l, m);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 15; k++) LD[SOLVE_LD*(jb - 1) + LD_M + k] = m[k];
                }
            }
        }
        __syncthreads();                       // panel jb complete, trailing update with panel jb-1 complete
    }
    if (fail || nfree == 0) { for (int k = tid; k < Nmax; k += SOLVE_THREADS) W.dp[k] = 0.0; return; }
    double *rhs = A + n*sst;
    if (wave == 0) solve_backsub_wave_ro(A, LD, n, nfree, lane, [sst](int i) { return i*sst; });
    __syncthreads();
    for (int a = tid; a < W.n_kf; a += SOLVE_THREADS) {
        int ia = W.fidx[a];
#pragma unroll
        for (int k = 0; k < 6; k++) W.dp[6*a + k] = ia >= 0 ? -rhs[6*ia + k] : 0.0;
    }
}
