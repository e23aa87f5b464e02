// Dense solve of the reduced camera system S dp = -g (included by tsba.hip after the device structs).
#pragma once

// ---- one workgroup of 12 waves, the whole system resident in LDS as a packed lower triangle (rows padded to an even length so
// that the 6-wide pose blocks move as b128), the right-hand side g riding along as row n so that the forward substitution is
// part of the factorisation.  Blocked (6x6 = one pose) LDL^T with ONE workgroup barrier per block column:
//   waves 0,1 ("P")   own the latency chain.  After barrier jb every lane owns one row below the diagonal block (lanes 0..5 of
//                     BOTH waves own the 6 rows of the diagonal block itself, redundantly): it applies panel jb-1 to its row of
//                     block column jb (look-ahead), the 6 diagonal rows go through a wave-private scratch to all lanes, every
//                     lane factors the 6x6 block in registers (no cross-wave hand-off) and solves its own panel row.
//   waves 2..11 ("T") apply panel jb-1 to everything right of block column jb on the matrix cores (v_mfma_f64_16x16x4, K = 6
//                     padded to 8), 16x16 tiles of the packed triangle; the last wave also inverts the unit-lower factor of
//                     block jb-1 for the back-substitution.
// Measured on MI355X (tools/lat_bench, lds_bench, branch_bench): one wave issues an instruction every ~4.2 cycles (fp64 VALU
// ~5.5, ds_read_b128 ~12, v_mfma_f64_16x16x4 64, taken scalar branch ~10), so each phase is sized in INSTRUCTIONS, not in
// dependent-latency terms.  Variants that keep the trailing matrix in register tiles (LDS only for the panels) or use a
// two-barrier MFMA look-ahead were measured within 10 % of this one; this is the simplest of the three.
// The back-substitution L^T x = z runs in wave 0 with z in registers (60 rows per register: the register a block lives in is a
// compile-time constant), x_block = L_jj^-T z_block from the stored inverse factors, next block's operands prefetched.
#define SOLVE_THREADS 768
#define SOLVE_PW 2                          // panel waves
#define SOLVE_PROWS (64 - 6)                // panel rows per wave and chunk
#ifdef TSBA_SOLVE_STAMPS                    // make stamps: cycle stamps into W.dbg (tsba_debug_stamps), perturbs the timing
__device__ long long ts_step_stamps[4*32];  // per factorisation step: panel wave 0 (load+apply | ldl | solve), update wave 2 (tiles)
#define STAMP(v) do { long long t_ = clock64(); v += t_ - tx; tx = t_; } while (0)
#else
#define STAMP(v) do { } while (0)
#endif
#define SOLVE_LD 48                         // per block: l[15] pad d[6] 1/d[6] inv(L)[15] pad  (16-byte aligned groups)
#define LD_D 16
#define LD_ID 22
#define LD_M 28
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

__host__ __device__ __forceinline__ int tri(int i) { return (i*(i + 1)) >> 1; }
// packed lower triangle with every row padded to an even length: rows start 16-byte aligned, so the 6-wide pose blocks
// (even column offsets) move as ds_read_b128 / ds_write_b128
// (Rows padded to a length = 2 mod 4 -- neighbouring rows an odd number of 16-byte slots apart: fewer LDS bank conflicts, which the counters
// show at 1.7 cycles per LDS instruction in k_bandp_factor -- were measured in round 2: C4 3.03 vs 2.84 ms, C6 16.35 vs 15.8 ms.  The longer
// address arithmetic costs more than the conflicts: these kernels are bound by instruction issue.)
__host__ __device__ __forceinline__ int rowoff(int i) { return ((i + 1) >> 1)*((i >> 1) + 1)*2; }
__device__ __forceinline__ int tri_row(int e) {          // largest i with tri(i) <= e   (e < 2^22)
    int i = (int)((__fsqrt_rn(8.f*(float)e + 1.f) - 1.f)*0.5f);
    if (tri(i) > e) i--; else if (tri(i + 1) <= e) i++;
    return i;
}
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
__device__ __forceinline__ void wave_lds_fence() {          // order this wave's LDS writes before its following LDS reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void ld6(const double *p, double v[6]) {      // p 16-byte aligned
    const v2d a = ((const v2d *)p)[0], b = ((const v2d *)p)[1], c = ((const v2d *)p)[2];
    v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y; v[4] = c.x; v[5] = c.y;
}
__device__ __forceinline__ void st6(double *p, const double v[6]) {
    ((v2d *)p)[0] = v2d{v[0], v[1]}; ((v2d *)p)[1] = v2d{v[2], v[3]}; ((v2d *)p)[2] = v2d{v[4], v[5]};
}
// lower 6x6 s (packed rows) -> unit-lower l (packed strictly-lower rows (1,0) (2,0) (2,1) ...), d, 1/d.  Right-looking; a
// single wave issues one instruction per ~4.5 cycles, so the instruction count is what matters here, not the chain
__device__ __forceinline__ void ldl6(double s[21], double l[15], double d[6], double id[6], bool &bad) {
#pragma unroll
    for (int c = 0; c < 6; c++) {
        double dc = s[tri(c) + c];
        if (!(dc > 0.0)) { bad = true; dc = 1.0; }
        d[c] = dc; id[c] = rcp_nr(dc);
#pragma unroll
        for (int r = c + 1; r < 6; r++) {
            const double v = s[tri(r) + c], lr = v*id[c];
            l[tri(r - 1) + c] = lr;
#pragma unroll
            for (int q = c + 1; q <= r; q++) s[tri(r) + q] = fma(-lr, s[tri(q) + c], s[tri(r) + q]);
        }
    }
}
// The same factorisation of a 6x6 block whose row r (entries 0..r) sits in lane r of the calling wave (a[], lanes 0..5; what the other lanes hold is not
// used): the pivot and the column below it are broadcast by v_readlane, every lane gets l, d, 1/d as uniform values and updates its own row.  Operation
// for operation what ldl6 does on a private copy of the block -- l(r,c) = s(r,c) * 1/d(c), s(r,q) -= l(r,c) s(q,c) in the same order -- so the
// results are the same bits; what it saves is the hand-over of the six rows through LDS to all 64 lanes.
__device__ __forceinline__ void ldl6_lanes(const double a[6], double l[15], double d[6], double id[6], bool &bad) {
    double sd[6];
#pragma unroll
    for (int c = 0; c < 6; c++) sd[c] = a[c];
#pragma unroll
    for (int c = 0; c < 6; c++) {
        double dc = readlane_f64(sd[c], c);
        if (!(dc > 0.0)) { bad = true; dc = 1.0; }
        d[c] = dc; id[c] = rcp_nr(dc);
        double col[6];
#pragma unroll
        for (int r = c + 1; r < 6; r++) col[r] = readlane_f64(sd[c], r);       // column c below the diagonal (not yet scaled)
#pragma unroll
        for (int r = c + 1; r < 6; r++) l[tri(r - 1) + c] = col[r]*id[c];
        const double lr = sd[c]*id[c];                            // this lane's own row: l(lane, c)
#pragma unroll
        for (int q = c + 1; q < 6; q++) sd[q] = fma(-lr, col[q], sd[q]);       // (entries right of the lane's diagonal are never read)
    }
}
// M = L^-1 for unit-lower L (both packed strictly-lower)
__device__ __forceinline__ void inv_unit_lower6(const double l[15], double m[15]) {
#pragma unroll
    for (int c = 0; c < 5; c++)
#pragma unroll
        for (int r = c + 1; r < 6; r++) {
            double v = -l[tri(r - 1) + c];
#pragma unroll
            for (int k = c + 1; k < r; k++) v = fma(-l[tri(r - 1) + k], m[tri(k - 1) + c], v);
            m[tri(r - 1) + c] = v;
        }
}

template <int w> struct IC { static constexpr int value = w; };
// Back substitution L^T x = z for the factor in LDS (packed rows of the unit-lower L, row n = z = D^-1 L^-1 g, LD table with the
// inverse 6x6 diagonal factors at LD_M), run by ONE wave with z in registers (60 rows per register: the register a block lives in
// is a compile-time constant), x_block = L_jj^-T z_block, next block's operands prefetched.  The solution replaces the rhs row.
template <class RO>
__device__ __forceinline__ void solve_backsub_wave_ro(double *A, double *LD, int n, int nfree, int lane, RO ro) {
    double *rhs = A + ro(n);
        // rows of blocks that were already consumed (and lanes 60..63) keep receiving updates: they are never read again, the
        // solution goes to the rhs row through lane 0
        double z[4];
#pragma unroll
        for (int w = 0; w < 4; w++) z[w] = rhs[min(60*w + min(lane, 59), n - 1)];
        auto back = [&](auto WC) {
            constexpr int w = decltype(WC)::value;
            auto fetch = [&](int q, double col[w + 1][6], double m[15]) {      // operands independent of the running solution
                const int j0 = 6*(10*w + q);
#pragma unroll
                for (int u = 0; u <= w; u++) {
                    const int k = min(60*u + lane, j0);
#pragma unroll
                    for (int c = 0; c < 6; c++) col[u][c] = A[ro(j0 + c) + k];
                }
                const double *o = LD + SOLVE_LD*(10*w + q) + LD_M;
                double t[16];
#pragma unroll
                for (int k = 0; k < 8; k++) { const v2d x = ((const v2d *)o)[k]; t[2*k] = x.x; t[2*k + 1] = x.y; }
#pragma unroll
                for (int k = 0; k < 15; k++) m[k] = t[k];
            };
            auto step = [&](int q, const double col[w + 1][6], const double m[15]) {
                double zb[6], x[6];
#pragma unroll
                for (int c = 0; c < 6; c++) zb[c] = readlane_f64(z[w], 6*q + c);
#pragma unroll
                for (int c = 0; c < 6; c++) {
                    double v0 = zb[c], v1 = 0.0;
#pragma unroll
                    for (int k = c + 1; k < 6; k++) { if ((k - c) & 1) v0 = fma(m[tri(k - 1) + c], zb[k], v0); else v1 = fma(m[tri(k - 1) + c], zb[k], v1); }
                    x[c] = v0 + v1;
                }
                if (lane == 0) st6(rhs + 6*(10*w + q), x);
#pragma unroll
                for (int u = 0; u <= w; u++)
                    z[u] -= fma(col[u][0], x[0], fma(col[u][1], x[1], col[u][2]*x[2])) + fma(col[u][3], x[3], fma(col[u][4], x[4], col[u][5]*x[5]));
            };
            int q = min(9, nfree - 1 - 10*w);
            if constexpr (w < 2) {                               // operands of the next block prefetched (two register sets)
                double colA[w + 1][6], mA[15], colB[w + 1][6], mB[15];
                fetch(q, colA, mA);
                for (; q >= 1; q -= 2) {
                    fetch(q - 1, colB, mB);
                    step(q, colA, mA);
                    if (q >= 2) fetch(q - 2, colA, mA);
                    step(q - 1, colB, mB);
                }
                if (q == 0) step(0, colA, mA);
            } else {
                double col[w + 1][6], m[15];
                for (; q >= 0; q--) { fetch(q, col, m); step(q, col, m); }
            }
        };
        if (nfree > 30) back(IC<3>{});
        if (nfree > 20) back(IC<2>{});
        if (nfree > 10) back(IC<1>{});
        back(IC<0>{});
}
__device__ __forceinline__ void solve_backsub_wave(double *A, double *LD, int n, int nfree, int lane) {
    solve_backsub_wave_ro(A, LD, n, nfree, lane, [](int i) { return rowoff(i); });
}

#define SOLVE_DIAG_NB 96                     // diagonal block of the large-system Cholesky (tsba_chol.h)
static size_t solve_lds_doubles(int N) { return (size_t)rowoff(N + 1) + 16 + (size_t)SOLVE_LD*(N/6) + 36*SOLVE_PW + 8; }
static size_t solve_diag_lds_doubles() { return solve_lds_doubles(SOLVE_DIAG_NB) + rowoff(SOLVE_DIAG_NB + 1) + SOLVE_DIAG_NB + 36*(SOLVE_DIAG_NB/6) + 16; }


// DIAG = false: the reduced system of a small window (S, g) -> pose step dp.
// DIAG = true : the 96x96 (or shorter, last) diagonal block at (B0, B0) of a LARGE system (tsba_chol.h): same factorisation, then
//               the inverse of the factor; written back as Cholesky factor L D^1/2 (lower triangle), its inverse transposed
//               (strict upper triangle) and the inverse's diagonal (W.LDbuf).
// PUB: the result is published for workgroups of the SAME launch that poll it (k_solve_back, tsba_kernels_step.h): dp and, behind it, the failure flag
// go out as device-coherent stores of values that are never NaN (the slots hold NaN until then)
__device__ __forceinline__ double co_load(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void co_store(double *p, double v) { __hip_atomic_store(p, v == v ? v : __builtin_inf(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// RL: the 6x6 diagonal block is factored where it already lives -- row r in lane r of the panel wave -- with the pivot and the column below it
// broadcast by v_readlane (uniform operands), instead of going through a wave-private LDS scratch to every lane and being factored there from a
// private copy: same operations on the same operands in the same order (bit-identical: RL = false is kept for that comparison,
// tsba_debug_options.solve_variant = 4), ~120 instructions on the chain of a block step instead of ~9 LDS reads + ~110 + the scratch hand-over
template <bool DIAG, bool PUB, bool RL = true>
__device__ __forceinline__ void solve_body(const Work &W, int B0, double *smem) {
    LmState *st = W.st;
    __shared__ int fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = SOLVE_THREADS/64, NT = NW - SOLVE_PW;
#ifdef TSBA_SOLVE_STAMPS
    const long long Tl = clock64();
#endif
    const int Nmax = W.N;
    double *A = smem;                                           // padded packed lower triangle, rows 0..n (row n = g)
    // the element addresses do not depend on the number of free poses: the first round of loads is issued together with the
    // loads of the solver state (one global round trip instead of two)
    const int neMax = DIAG ? tri(SOLVE_DIAG_NB) : tri(Nmax);
    const size_t ldS = (size_t)W.ldS;
    const double *Sb = DIAG ? W.S + (size_t)B0*ldS + B0 : W.S;
    const int rmax = DIAG ? Nmax - 1 - B0 : Nmax - 1;           // (the last block of a large system may be short)
    double v[12]; int er[12], ec[12];
#pragma unroll
    for (int u = 0; u < 12; u++) {
        const int e = min(u*SOLVE_THREADS + tid, neMax - 1);
        er[u] = tri_row(e); ec[u] = e - tri(er[u]);
        v[u] = Sb[(size_t)min(er[u], rmax)*ldS + min(ec[u], rmax)];
    }
    const int done = st->done, nfree_all = *W.nfree, sfail = st->step_fail;
    if (done) return;
    if (DIAG && (sfail || 6*nfree_all <= B0)) return;
    const int nfree = DIAG ? min(SOLVE_DIAG_NB, 6*nfree_all - B0)/6 : nfree_all;
    const int n = 6*nfree, ne = tri(n);
    double *LD = A + rowoff(n + 1) + 16;
    double *scr = LD + SOLVE_LD*nfree;
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
    for (int k = tid; k < n; k += SOLVE_THREADS) A[rowoff(n) + k] = DIAG ? 0.0 : W.g[k];
    if (tid == 0) fail = DIAG ? 0 : sfail;
    __syncthreads();
#ifdef TSBA_SOLVE_STAMPS
    if (tid == 0) W.dbg[0] = clock64() - Tl;
    long long tx = clock64(), T0 = tx, ta = 0, tb = 0, tc = 0, td = 0;
#endif
    for (int jb = 0; jb < nfree && !fail; jb++) {
        const int j0 = 6*jb, R0 = j0 + 6, p0 = j0 - 6;
#ifdef TSBA_SOLVE_STAMPS
        const long long tx0_ = clock64();
#endif
        if (wave < SOLVE_PW) {
            double Lk[36], dprev[6];
            if (jb > 0) {
                ld6(LD + SOLVE_LD*(jb - 1) + LD_D, dprev);
#pragma unroll
                for (int c = 0; c < 6; c++) ld6(A + rowoff(j0 + c) + p0, Lk + 6*c);
            }
            auto load_row = [&](int i, double a[6]) {           // row i of block column jb with panel jb-1 applied
                const double *row = A + rowoff(i);
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
            double l[15], d[6], id[6]; bool bad = false;
#ifdef TSBA_SOLVE_STAMPS
            long long sa_ = 0;
#endif
            if constexpr (RL) {
                STAMP(ta);
#ifdef TSBA_SOLVE_STAMPS
                sa_ = clock64();
#endif
                ldl6_lanes(a, l, d, id, bad);
            } else {
            if (lane < 6) st6(scr + wave*36 + lane*6, a);
            wave_lds_fence();
            STAMP(ta);
#ifdef TSBA_SOLVE_STAMPS
            sa_ = clock64();
#endif
            double s[21];
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
            }
            if (wave == 0 && lane == 0) {
                double *o = LD + SOLVE_LD*jb;
#pragma unroll
                for (int k = 0; k < 15; k++) o[k] = l[k];
                st6(o + LD_D, d); st6(o + LD_ID, id);
                if (bad) { fail = 1; st->step_fail = 1; }
            }
            STAMP(tb);
#ifdef TSBA_SOLVE_STAMPS
            const long long sb_ = clock64();
#endif
            auto solve_row = [&](int i, double a[6]) {           // x L^T = a (right-looking: 5-deep chain), stored row = x D^-1
#pragma unroll
                for (int c = 0; c < 5; c++)
#pragma unroll
                    for (int q = c + 1; q < 6; q++) a[q] = fma(-a[c], l[tri(q - 1) + c], a[q]);
#pragma unroll
                for (int c = 0; c < 6; c++) a[c] *= id[c];
                st6(A + rowoff(i) + j0, a);
            };
            if (lane >= 6) {
                if (i0 <= n) solve_row(i0, a);
                for (int i = i0 + SOLVE_PW*SOLVE_PROWS; i <= n; i += SOLVE_PW*SOLVE_PROWS) { load_row(i, a); solve_row(i, a); }
            }
#ifdef TSBA_SOLVE_STAMPS
            if (!DIAG && wave == 0 && lane == 0 && jb < 32) { const long long se_ = clock64(); ts_step_stamps[jb] = sa_ - tx0_; ts_step_stamps[32 + jb] = sb_ - sa_; ts_step_stamps[64 + jb] = se_ - sb_; }
#endif
            if (wave == 1 && jb == nfree - 1) {                  // inverse factor of the last block (the others: last T wave)
                double m[15];
                inv_unit_lower6(l, m);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 15; k++) LD[SOLVE_LD*jb + LD_M + k] = m[k];
                }
            }
        } else if (jb > 0) {
            // trailing update with panel jb-1: rows >= R0 (incl. the rhs row n), columns R0..n-1
            const int mr = n - R0 + 1, mc = n - R0;
            const double *ldp = LD + SOLVE_LD*(jb - 1);
            if (mc > 0) {
                const int ntr = (mr + 15) >> 4, ntc = (mc + 15) >> 4, ntile = tri(ntr);
                const int lr = lane & 15, lk = lane >> 4;
                const int k1 = min(4 + lk, 5);
                const double dk0 = ldp[LD_D + lk], dk1 = lk < 2 ? ldp[LD_D + 4 + lk] : 0.0;
                for (int t = wave - SOLVE_PW; t < ntile; t += NT) {
                    const int ti = tri_row(t), tj = t - tri(ti);
                    if (tj >= ntc) continue;
                    // unconditional, index-clamped loads (rows / columns past the edge only feed outputs that are never stored);
                    // only the K padding (k = 6, 7) must be exact zeros
                    const int arow = rowoff(min(R0 + 16*ti + lr, n)) + p0, brow = rowoff(min(R0 + 16*tj + lr, n - 1)) + p0;
                    double a0 = -A[arow + lk], a1 = -A[arow + k1];
                    double b0 = A[brow + lk]*dk0, b1 = A[brow + k1]*dk1;
                    if (lk >= 2) { a1 = 0.0; b1 = 0.0; }
                    const int ccol = R0 + 16*tj + lr;
                    v4d c; int ci[4]; bool ok[4];
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int crow = R0 + 16*ti + lk + 4*r;
                        ok[r] = crow <= n && ccol <= crow && ccol < n;
                        ci[r] = rowoff(min(crow, n)) + min(ccol, min(crow, n - 1));
                        c[r] = A[ci[r]];
                    }
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c, 0, 0, 0);
#pragma unroll
                    for (int r = 0; r < 4; r++) if (ok[r]) A[ci[r]] = c[r];
                }
            }
            if (wave == NW - 1) {                                // inverse of the unit-lower factor of block jb-1
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
        STAMP(tc);
#ifdef TSBA_SOLVE_STAMPS
        if (!DIAG && lane == 0 && jb < 32) { const long long se_ = clock64();
            if (wave == 2) ts_step_stamps[3*32 + jb] = se_ - tx0_; }
#endif
        __syncthreads();                       // panel jb complete, trailing update with panel jb-1 complete
        STAMP(td);
    }
#ifdef TSBA_SOLVE_STAMPS
    if (lane == 0 && (wave == 0 || wave == 2 || wave == NW - 1)) {
        long long *o = W.dbg + 8 + 4*(wave == 0 ? 0 : wave == 2 ? 1 : 2); o[0] = ta; o[1] = tb; o[2] = tc; o[3] = td;
        if (wave == 0) { W.dbg[1] = clock64() - T0; W.dbg[6] = nfree; }
    }
    long long T1 = clock64();
#endif
    if (DIAG) {
        if (fail) return;                                     // (st->step_fail is set: the back-substitution kernel zeroes dp)
        // ---- inverse of the unit-lower factor, block row by block row:  M_bc = -M_bb sum_{k=c}^{b-1} L_bk M_kc   (c < b)
        double *Mi = scr + 36*SOLVE_PW;                       // second padded packed triangle
        double *sq = Mi + rowoff(SOLVE_DIAG_NB + 1), *tmp = sq + SOLVE_DIAG_NB;   // sqrt(d), 6 x (6 b) scratch
        for (int t = tid; t < 21*nfree; t += SOLVE_THREADS) {                     // diagonal blocks of the inverse
            const int b = t/21, e = t - 21*b, r = tri_row(e), q = e - tri(r);
            Mi[rowoff(6*b + r) + 6*b + q] = r == q ? 1.0 : LD[SOLVE_LD*b + LD_M + tri(r - 1) + q];
        }
        for (int t = tid; t < n; t += SOLVE_THREADS) sq[t] = sqrt(LD[SOLVE_LD*(t/6) + LD_D + t % 6]);
        __syncthreads();
        for (int b = 1; b < nfree; b++) {
            const int nout = 36*b;                            // (c, r, q): 6x6 block (b, c), c < b
            for (int t = tid; t < nout; t += SOLVE_THREADS) {
                const int c = t/36, r = (t - 36*c)/6, q = t % 6;
                double acc = 0.0;
                const double *lrow = A + rowoff(6*b + r);
                for (int k = 6*c + q; k < 6*b; k++) acc = fma(lrow[k], Mi[rowoff(k) + 6*c + q], acc);   // column 6c+q of M is zero above its diagonal
                tmp[t] = acc;
            }
            __syncthreads();
            for (int t = tid; t < nout; t += SOLVE_THREADS) {
                const int c = t/36, r = (t - 36*c)/6, q = t % 6;
                double acc = tmp[36*c + 6*r + q];             // M_bb is unit lower: row r = e_r + strictly-lower part
                for (int m = 0; m < r; m++) acc = fma(LD[SOLVE_LD*b + LD_M + tri(r - 1) + m], tmp[36*c + 6*m + q], acc);
                Mi[rowoff(6*b + r) + 6*c + q] = -acc;
            }
            __syncthreads();
        }
        // ---- write back: lower triangle L D^1/2, strict upper triangle (L D^1/2)^-T, diagonal of the inverse to LDbuf
        double *Sw = W.S + (size_t)B0*ldS + B0;
        for (int t = tid; t < n*n; t += SOLVE_THREADS) {
            const int r = t/n, c = t - r*n;
            if (c < r) Sw[(size_t)r*ldS + c] = A[rowoff(r) + c]*sq[c];
            else if (c == r) { Sw[(size_t)r*ldS + c] = sq[c]; W.LDbuf[B0 + r] = 1.0/sq[r]; }
            else Sw[(size_t)r*ldS + c] = Mi[rowoff(c) + r]/sq[c];          // W^T[r][c] = W[c][r] = M[c][r] / sqrt(d_c)
        }
        return;
    }
    if (fail || nfree == 0) {
        for (int k = tid; k < Nmax; k += SOLVE_THREADS) { if (PUB) co_store(&W.dp[k], 0.0); else W.dp[k] = 0.0; }
        if (PUB && tid == 0) co_store(&W.dp[Nmax], fail ? 1.0 : 0.0);
        return; }
    double *rhs = A + rowoff(n);
    if (wave == 0) solve_backsub_wave(A, LD, n, nfree, lane);
    __syncthreads();
#ifdef TSBA_SOLVE_STAMPS
    if (tid == 0) W.dbg[2] = clock64() - T1;
#endif
    for (int a = tid; a < W.n_kf; a += SOLVE_THREADS) {
        int ia = W.fidx[a];
#pragma unroll
        for (int k = 0; k < 6; k++) { const double d = ia >= 0 ? -rhs[6*ia + k] : 0.0; if (PUB) co_store(&W.dp[6*a + k], d); else W.dp[6*a + k] = d; }
    }
    if (PUB && tid == 0) co_store(&W.dp[Nmax], 0.0);
}
template <bool DIAG, bool RL = true>
__global__ __launch_bounds__(SOLVE_THREADS) void k_solve_t(Work W, int B0) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    solve_body<DIAG, false, RL>(W, B0, smem);
}

// (k_solve_r -- the same solver with every row at the same stride, update tiles with an unmasked path and scalar tile indices, optionally
// only on the SIMDs without a panel wave -- was built and measured in round 2 on the C4 window: 34.95 us with ten update waves, 37.0 us
// with the six of SIMD 2 and 3, against 33.5 us for this kernel.  What paid on the cyclic-reduction levels (tsba_bandcre.h), where the update
// waves set the pace, does not here, where the panel chain does.  In the history of this repository: tools/experiments/solve_rect.h, removed in round 3.)
