// EXPERIMENT, NOT PART OF THE PRODUCT BUILD (kept for the record; see DESIGN.md section 8, "what was tried on the reduced-system solve").
//
// Round 2 measured this kernel against the blocked 6x6 LDL^T of csrc/tsba_solve.h on MI355X (gpurun_out/r2_solvers.log, copied to
// profiles/r02_small_solver_experiment.txt): correct on every size (it passed tests/test_gpu_parity.py::test_small_system_solvers_agree
// for 2 .. 29 free poses), but SLOWER -- C4 (102 rows) 66.5 us against 33.3 us, 31 keyframes 150 us against 80 us, 10 keyframes 23 us
// against 14 us.  A column step costs ~1000 cycles instead of the ~250 of its dependent chain: per step every thread issues up to
// 36 fp64 FMAs + 16 ds_read_b64 + 8 multiplies + the register copies the compiler inserts around the uniform row-block skip, and one
// CU issues an fp64 instruction every ~5 cycles per wave -- the column variant is ISSUE-bound where the blocked kernel is chain-bound.
// To build it again: append this file to tsba_solve.h and launch k_solve_col<4|8|12> with solvec_lds_doubles<NB>(N) doubles of LDS.
// ---- Column LDL^T: the same small reduced system, factored one COLUMN per step instead of one pose block per step.
// The blocked kernel above spends ~3600 cycles per 6 columns on a chain that a single wave has to issue alone (6x6 LDL^T in
// registers, panel solves, hand-offs through LDS); what bounds a 100-row factorisation is the length of the dependent chain, not
// FLOPs.  Here the matrix lives in REGISTERS, 2-D block-cyclic over a 16 x 16 thread grid -- thread (tr, tc) owns the elements
// (16 bi + tr, 16 bj + tc) of the lower triangle, the right-hand side rides along as row n -- and a step is
//     read column k from LDS (d_k, 1/d_k, a_ik for my rows, a_jk / d_k for my columns)  ->  rank-1 update of my elements, the
//     elements of column k + 1 FIRST, whose owners (16 lanes) store them to LDS at once, the owner of the next pivot with its
//     reciprocal  ->  the rest of the update while those stores are in flight  ->  ONE barrier.
// The finished columns stay in the packed triangle in LDS (unscaled: a_ik = l_ik d_k); after the last step every thread stores
// its elements scaled from registers, one thread per pose block inverts the 6x6 unit-lower diagonal factor, and the
// back-substitution of the blocked kernel runs unchanged.  Work per step shrinks with the trailing matrix (the loops over the
// 16-row blocks start at the pivot's block: compile-time unrolled per block, so that the register indices stay static).
#define SOLVEC_T 256
template <int NB> static size_t solvec_lds_doubles(int N) { return solve_lds_doubles(N) + (size_t)16*NB + 16; }

template <int NB>
__global__ __launch_bounds__(SOLVEC_T) void k_solve_col(Work W) {
    LmState *st = W.st;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tr = tid & 15, tc = tid >> 4;
    const int Nmax = W.N;
    const size_t ldS = (size_t)W.ldS;
    // every load whose address does not depend on the number of free poses goes out together with the solver state: the lower
    // triangle of S for this thread's positions (clamped: dead positions are discarded below) and the gradient for its columns
    double a[NB][NB], gv[NB];
#pragma unroll
    for (int bi = 0; bi < NB; bi++)
#pragma unroll
        for (int bj = 0; bj <= bi; bj++)
            a[bi][bj] = W.S[(size_t)min(16*bi + tr, Nmax - 1)*ldS + min(16*bj + tc, Nmax - 1)];
#pragma unroll
    for (int bj = 0; bj < NB; bj++) gv[bj] = W.g[min(16*bj + tc, Nmax - 1)];
    const int done = st->done, nfree = *W.nfree, sfail = st->step_fail;
    if (done) return;
    const int n = 6*nfree;                                       // rows 0 .. n-1: S, row n: g
    if (sfail || nfree == 0) { for (int k = tid; k < Nmax; k += SOLVEC_T) W.dp[k] = 0.0; return; }
    double *A = smem;
    double *LD = A + rowoff(n + 1) + 16;
    double *invd = LD + SOLVE_LD*nfree + 36*SOLVE_PW + 8;        // 1 / d_k
    __shared__ int fail;
    if (tid == 0) fail = 0;
    const int nbl = (n >> 4) + 1;                                // row blocks that hold live rows (0 .. n)
    int rb[NB], cb[NB];                                          // LDS offsets of my rows / of my columns taken as rows (clamped to row n)
#pragma unroll
    for (int b = 0; b < NB; b++) { rb[b] = rowoff(min(16*b + tr, n)); cb[b] = rowoff(min(16*b + tc, n)); }
#pragma unroll
    for (int bi = 0; bi < NB; bi++)
#pragma unroll
        for (int bj = 0; bj <= bi; bj++) {
            const int i = 16*bi + tr, j = 16*bj + tc;
            double v = a[bi][bj];
            if (i == n) v = gv[bj];
            a[bi][bj] = (i <= n && j <= i && j < n) ? v : 0.0;   // dead positions: exact zeros, so that nothing non-finite can enter
        }
    // column c (block column CB) from the registers of its owners (tc == c & 15) to the packed triangle, with 1 / d_c
    auto put_col = [&](auto CBC, int c) {
        constexpr int CB = decltype(CBC)::value;
        if (tc != (c & 15)) return;
        if (tr == tc) {                                          // the owner of the pivot a[CB][CB]
            double d = a[CB][CB];
            if (!(d > 0.0)) { d = 1.0; a[CB][CB] = 1.0; fail = 1; st->step_fail = 1; }     // not positive definite: an invalid LM step
            invd[c] = rcp_nr(d);
        }
#pragma unroll
        for (int bi = CB; bi < NB; bi++) { const int i = 16*bi + tr; if (i >= c && i <= n) A[rb[bi] + c] = a[bi][CB]; }
    };
    put_col(IC<0>{}, 0);
    __syncthreads();
    // one step: pivot column k = 16 KB + kk; NC = block column of column k + 1 (KB, or KB + 1 after the last column of a block)
    auto step = [&](auto KBC, auto NCC, int kk) {
        constexpr int KB = decltype(KBC)::value, NC = decltype(NCC)::value;
        const int k = 16*KB + kk;
        const double inv = invd[k];
        double ci[NB], cj[NB];
#pragma unroll
        for (int b = KB; b < NB; b++) { ci[b] = A[rb[b] + k]; cj[b] = A[cb[b] + k]; }
#pragma unroll
        for (int b = KB; b < NB; b++) cj[b] *= inv;
        if (tr <= kk) ci[KB] = 0.0;                              // rows up to the pivot are finished: they take no update
        if (tc <= kk) cj[KB] = 0.0;                              // ... and so are the columns
        if (NC < NB && k + 1 < n && tc == ((kk + 1) & 15)) {     // look-ahead: the next column first, straight to LDS
#pragma unroll
            for (int bi = NC; bi < NB; bi++) a[bi][NC] = fma(-ci[bi], cj[NC], a[bi][NC]);
            cj[NC] = 0.0;                                        // (done: the general update below leaves it alone)
            put_col(NCC, k + 1);
        }
#pragma unroll
        for (int bi = KB; bi < NB; bi++) {
            if (bi >= nbl) break;                                // (uniform) row blocks beyond the right-hand side
#pragma unroll
            for (int bj = KB; bj <= bi; bj++) a[bi][bj] = fma(-ci[bi], cj[bj], a[bi][bj]);
        }
        __syncthreads();
    };
    auto block = [&](auto KBC) {
        constexpr int KB = decltype(KBC)::value;
        const int kend = min(16, n - 16*KB);                     // columns of this block (k < n)
        for (int kk = 0; kk < kend; kk++) {
            if (kk < 15) step(KBC, KBC, kk);
            else step(KBC, IC<(KB + 1 < NB ? KB + 1 : KB)>{}, kk);
        }
    };
    if (n > 0) block(IC<0>{});
    if constexpr (NB > 1) if (n > 16) block(IC<1>{});
    if constexpr (NB > 2) if (n > 32) block(IC<2>{});
    if constexpr (NB > 3) if (n > 48) block(IC<3>{});
    if constexpr (NB > 4) if (n > 64) block(IC<4>{});
    if constexpr (NB > 5) if (n > 80) block(IC<5>{});
    if constexpr (NB > 6) if (n > 96) block(IC<6>{});
    if constexpr (NB > 7) if (n > 112) block(IC<7>{});
    if constexpr (NB > 8) if (n > 128) block(IC<8>{});
    if constexpr (NB > 9) if (n > 144) block(IC<9>{});
    if constexpr (NB > 10) if (n > 160) block(IC<10>{});
    if constexpr (NB > 11) if (n > 176) block(IC<11>{});
    if (fail) { for (int k = tid; k < Nmax; k += SOLVEC_T) W.dp[k] = 0.0; return; }
    // unit-lower L into the packed triangle: every thread scales its own elements (they have not changed since their column's step)
#pragma unroll
    for (int bj = 0; bj < NB; bj++) {
        const int j = 16*bj + tc;
        const double idj = invd[min(j, n - 1)];
#pragma unroll
        for (int bi = bj; bi < NB; bi++) { const int i = 16*bi + tr; if (i > j && i <= n && j < n) A[rb[bi] + j] = a[bi][bj]*idj; }
    }
    __syncthreads();
    if (tid < nfree) {                                           // inverse of the unit-lower 6x6 diagonal factor of pose block tid
        double l[15], m[15];
#pragma unroll
        for (int r = 1; r < 6; r++)
#pragma unroll
            for (int c = 0; c < r; c++) l[tri(r - 1) + c] = A[rowoff(6*tid + r) + 6*tid + c];
        inv_unit_lower6(l, m);
#pragma unroll
        for (int k = 0; k < 15; k++) LD[SOLVE_LD*tid + LD_M + k] = m[k];
    }
    __syncthreads();
    double *rhs = A + rowoff(n);
    if (wave == 0) solve_backsub_wave(A, LD, n, nfree, lane);
    __syncthreads();
    for (int q = tid; q < W.n_kf; q += SOLVEC_T) {
        const int ia = W.fidx[q];
#pragma unroll
        for (int k = 0; k < 6; k++) W.dp[6*q + k] = ia >= 0 ? -rhs[6*ia + k] : 0.0;
    }
}
