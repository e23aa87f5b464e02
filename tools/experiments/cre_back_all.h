// Experiment (round 2, not part of the product build): every back-substitution level of the cyclic reduction in one launch.
// 5000 keyframes, 126 separators: 91.4 us per LM trial against 78 us for seven k_cre_back launches (profiles: DESIGN.md section 6).

// ---- all back-substitution levels in ONE launch.  Every separator (except the root, which k_cre_elim(root) has already solved) gets a
// workgroup; a workgroup loads what does not depend on the solution (its factor, X_a, X_c, z), then waits for the flags of its two
// neighbours (they belong to higher levels), solves, publishes x_i and raises its own flag.  A level was a launch of its own (11 us:
// two dependent global round trips, the matrix-vector product, ten block steps of back substitution, and a kernel boundary);
// here the loads of all levels overlap and a hop of the dependency tree costs the flag hand-over instead of a launch.
// Workgroups are ordered top level first and all of them are resident at once (at most BANDP_MAXP - 1 workgroups on 256 compute
// units), so a waiting workgroup never keeps its producer from running.  Flags carry the epoch of the launch (no reset between
// launches); release / acquire at agent scope (the L2 of another XCD does not snoop this one).
__global__ __launch_bounds__(CRE_BT) void k_cre_back_all(Work W, Work Ws, int bw, int Pmax, const double *fac, int *flag, int epoch) {
    const LmState *st = W.st;
    if (st->done || st->step_fail) return;
    const int m = cr_nsep(W, bw, Pmax), s = bw, B = s/6, mmax = Pmax - 1, tid = threadIdx.x, lane = tid & 63;
    // workgroup -> pivot: levels from the top (largest h) down, (2 k + 1) h within a level; the worst case mmax decides the order
    int i = -1, h = 0;
    {
        int b = blockIdx.x, ht = 1;
        while (2*ht < mmax) ht <<= 1;
        for (int hh = ht; hh >= 1; hh >>= 1) {
            const int cnt = hh < mmax ? (mmax - hh - 1)/(2*hh) + 1 : 0;
            if (b < cnt) { i = (2*b + 1)*hh; h = hh; break; }
            b -= cnt;
        }
    }
    if (i < 0 || i >= m) return;
    const int a = i - h, c = i + h < m ? i + h : -1;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int xbase = rowoff(s);
    double *A = smem, *LD = A + rowoff(s + 1) + 16, *part = LD + SOLVE_LD*B, *xs = part + 4*128;
    const double *S = Ws.S; double *x = Ws.Sy;
    const double *rec = fac + (size_t)i*cre_rec_doubles(s);
    const double *Xa = cr_blk(S, s, mmax, i, a), *Xc = c >= 0 ? cr_blk(S, s, mmax, c, i) : nullptr;
    const int nx = c >= 0 ? 2*s : s;                             // rows of [X_a ; X_c]; a thread: column `col`, rows grp, grp + 4, ...
    const int col = tid & 127, grp = tid >> 7;
    const double zc = (grp == 0 && col < s) ? rec[xbase + SOLVE_LD*B + col] : 0.0;
    auto xrow = [&](int r) { return r < s ? Xa + (size_t)r*s : Xc + (size_t)(r - s)*s; };
    constexpr int UB = 10;
    double xv[UB];
#pragma unroll
    for (int u = 0; u < UB; u++) { const int r = grp + 4*u; xv[u] = (col < s && r < nx) ? xrow(r)[col] : 0.0; }
    cre_batched<CRE_BT, 8>(xbase, tid, [&](int e) { return rec[e]; }, [&](int e, double v) { A[e] = v; });
    for (int k = tid; k < SOLVE_LD*B; k += CRE_BT) LD[k] = rec[xbase + k];
    // the neighbours' solutions (block 0 was solved by the launch before this one)
    if (tid == 0) {
        if (a > 0) while (__hip_atomic_load(flag + a, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(4);
        if (c > 0) while (__hip_atomic_load(flag + c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != epoch) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    for (int k = tid; k < nx; k += CRE_BT) xs[k] = __hip_atomic_load(k < s ? x + (size_t)a*s + k : x + (size_t)c*s + k - s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < UB; u++) { const int r = grp + 4*u; if (r < nx) acc = fma(xv[u], xs[r], acc); }
    for (int r0 = grp + 4*UB; r0 < nx; r0 += 4*UB) {
#pragma unroll
        for (int u = 0; u < UB; u++) { const int r = r0 + 4*u; xv[u] = (col < s && r < nx) ? xrow(r)[col] : 0.0; }
#pragma unroll
        for (int u = 0; u < UB; u++) { const int r = r0 + 4*u; if (r < nx) acc = fma(xv[u], xs[r], acc); }
    }
    part[grp*128 + col] = acc;
    __syncthreads();
    if (grp == 0 && col < s) A[xbase + col] = zc - ((part[col] + part[128 + col]) + (part[256 + col] + part[384 + col]));
    __syncthreads();
    if (tid >= 64) return;
    solve_backsub_wave(A, LD, s, B, lane);
    wave_lds_fence();
    for (int k = lane; k < s; k += 64) __hip_atomic_store(x + (size_t)i*s + k, A[xbase + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // this wave's stores of x_i before the flag
    if (lane == 0) __hip_atomic_store(flag + i, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
