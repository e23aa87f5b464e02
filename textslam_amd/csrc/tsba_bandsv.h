// Solve phase of the partitioned band solver for ONE right-hand side: M x = r with the factor k_bandp_factor / k_bandp_sepf / k_cre_elim left
// behind (tsba_bandp.h, tsba_bandcre.h).  This is what a conjugate-gradient iteration applies as preconditioner (tsba_pcg.h); re-running the
// factorisation with the residual as right-hand side costs 0.5 ms at 5000 keyframes, the many-column solve phase (tsba_bandms.h: a lane owns a
// COLUMN) as much for one column as for 64.  Fifteen to twenty applications per LM trial: everything here is about latency.
//
//   interiors (a chain of pose blocks, one wave): a lane owns a ROW of the running right-hand side, the window of the B blocks after (before) the
//     pivot.  A step: the pivot block's six values are broadcast (v_readlane), its 6 x 6 unit triangle is solved redundantly by every lane
//     (15 FMAs on uniform values), every lane subtracts its six coefficients times the result -- no reduction across lanes anywhere, ~100
//     instructions per pose block.  The chain wave never waits for memory: the other three waves of the workgroup stage what the next SV_K pivots
//     need in LDS a chunk ahead and do the dense border work (the rows of the separator on the left) on the side.
//   separators (cyclic reduction, a workgroup per pivot of a level): NO substitution at all -- k_sv_linv inverts every separator's unit-lower
//     factor once per factorisation (block recursion, six threads per column), after which a level is three small dense products (L^-1 v, X_a w, X_c w) whose
//     operands are all requested up front, speculatively, and looked at after a single wait (a round trip to memory costs ~1.5 us; a kernel
//     that asks for one thing after the other -- flags, number of free poses, record, neighbours, coupling blocks -- pays it six times).
// Vectors: MsBuf with T = 1 (the right-hand side comes as rs * r[]).  Chain partitions only, cyclic-reduction separator system.
#pragma once

#define SV_T 512                            // interiors: chain wave + seven staging waves
#define SV_CT 512                           // separators: 8 waves
#define SV_K 16                             // pivot blocks per staged chunk (a chunk's chain time ~ one round trip to memory)

__device__ __forceinline__ double sv_bcast(double v, int lane) {           // lane: uniform
    const int l = __builtin_amdgcn_readfirstlane(lane);
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
template <int NREG>
__device__ __forceinline__ double sv_pivot(const double (&t)[NREG], int rho) {     // the value of window row rho (uniform)
    if (NREG == 1) return sv_bcast(t[0], rho);
    const double lo = sv_bcast(t[0], rho & 63), hi = sv_bcast(t[NREG - 1], rho & 63);
    return rho >= 64 ? hi : lo;
}
__device__ __forceinline__ void sv_pin(double &x) { asm volatile("" : "+v"(x)); }     // the request for x is issued before this point (the compiler would sink it below
__device__ __forceinline__ void sv_pin(v2d &x) { asm volatile("" : "+v"(x)); }        // the early-out branches, into the block that uses it: one more round trip)
__device__ __forceinline__ int sv_nsep(int nf, int B, int Pmax) { return nf > 0 ? bandp_part(nf, B, Pmax, 0).P - 1 : 0; }      // chain: separators 0 .. m - 1, root 0 (cr_range)

// staged pivot record (CS = 36 B + 32 doubles): [0, 36 B) coefficients | 36 B + [0, 15) unit-lower l, [16, 22) 1/d | 36 B + 22 + [0, 6) entering values
__host__ __device__ __forceinline__ int sv_cs(int B) { return 36*B + 32; }
static size_t sv_fwd_lds_doubles(int B) { return 2*(size_t)SV_K*sv_cs(B) + 2*SV_K*6 + 4*96; }
static size_t sv_back_lds_doubles(int B, int lmax) { return 2*(size_t)SV_K*sv_cs(B) + 6*(size_t)lmax + 80; }

// producers (threads 64 .. 255): pivot records of a chunk, 6 pivots x 32 threads per round, every request of the chunk in flight at once
#define SV_PT ((SV_T - 64)/SV_K)               // producer threads per pivot of a chunk
#define SV_PU ((36*MS_BMAX + 28)/2/SV_PT + 1)  // 16-byte pieces per producer thread
struct SvStage { v2d v[SV_PU]; };
#define SV_BG 4                             // border: groups of 96 producer threads, SV_BQ pivots of a chunk each
#define SV_BQ (SV_K/SV_BG)
template <int NREG> struct SvStep { double cf[NREG][6], ent[NREG], l[15], idl; };

// ---- interiors, forward.  grid Pmax, SV_T threads.
//   M.V = D^-1 w (L w = r), M.G [label][s] = rows of the separator on the right below this interior, M.G2 [label][s] = border rows of the separator on the left
// upd (conjugate gradients, tsba_pcg.h): the step of the iteration that leads to this application -- alpha = r.z / p.q from the partial sums,
// x += alpha p, r -= alpha q -- is taken HERE on this interior's rows before anything else (k_pcg_update as a launch of its own: 5.3 us + a gap per iteration)
struct SvUpd { int on, it, npq, pq_off; };
template <int NREG>
__global__ __launch_bounds__(SV_T) void k_sv_fwd_int(Work W, int bw, int Pmax, const double *__restrict__ Lrow, const double *__restrict__ Lb, const double *r, double rs, MsBuf M, int tree, SvUpd upd) {
    extern __shared__ __attribute__((aligned(16))) double ms_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = ms_uni(tid >> 6);
#ifdef SV_STAMPS
    long long stamps[8]; int nstamp = 0;
#define SVS() do { if (nstamp < 8) stamps[nstamp++] = __builtin_readcyclecounter(); } while (0)
#else
#define SVS() do {} while (0)
#endif
    SVS();
    LmState *st_ = W.st; const int flags = st_->done | st_->lin_done | st_->step_fail, nf = ms_uni(*W.nfree);
    double pqs = 0.0, rz_u = 0.0;
    if (upd.on) { for (int k = lane; k < upd.npq; k += 64) pqs += W.pc_part[upd.pq_off + k]; rz_u = W.pcs[upd.it & 1].rz; }     // (pcg_sum_parts: requested with the state)
    if (flags || nf <= 0) return;
    SVS();
    const int B = bw/6, p = blockIdx.x, CS = sv_cs(B), TB = 36*B;
    double *stg = ms_smem, *wch = stg + 2*(size_t)SV_K*CS, *red = wch + 2*SV_K*6;       // stg [2][SV_K][CS], wch [2][SV_K][6], red [2][96]
    const BandpPart PT = bandp_part(nf, B, Pmax, p);
    const int a = ms_uni(PT.a), b = ms_uni(PT.b), P = ms_uni(PT.P);
    if (p >= P) return;
    // (k_sv_cre_tree polls these slots: this application's values have not arrived while they hold a NaN)
    if (tree && tid >= SV_T - 3*bw) { const int e = tid - (SV_T - 3*bw), row = e % bw, which = e/bw; const double qn = __builtin_nan("");
        if (which < 2) M.Cg[((size_t)p*2 + which)*bw + row] = qn; else M.Xs[(size_t)p*bw + row] = qn; }
    const int REC = bw*6, rend = p < P - 1 ? b + B : b, nch = (b - a + SV_K - 1)/SV_K;
    if (upd.on) {                                               // this interior's rows 6 a .. 6 rend - 1 (the interiors' ranges tile the system)
        const double pq = wave_sum1(pqs);
        if (!(pq > 0.0)) { if (p == 0 && tid == 0) st_->step_fail = 1; return; }       // S is positive definite (damped): a breakdown is a failed step
        const double alpha = rz_u/pq;
        const double *pp = W.pc_p[upd.it & 1];
        for (int i = 6*a + tid; i < 6*rend; i += SV_T) {
            W.pc_x[i] = fma(alpha, pp[i], W.pc_x[i]);
            W.pc_r[i] = fma(-alpha, W.pc_q[i], W.pc_r[i]); }
        __syncthreads();                                        // (r below is W.pc_r: this workgroup reads only rows it has just written)
    }
    const int ptid = tid - 64, pk = ptid & (SV_K - 1), pj = ptid/SV_K, perp = (TB + 28)/2;       // producers: (pivot of the chunk, SV_PT threads across its record, 16 bytes each)
    auto stage = [&](int c) {                                   // producers: chunk c
        const int q = a + c*SV_K + pk; const bool on = q < b;
        const double *Lq = Lrow + (size_t)(q + 1)*REC, *ldq = W.LDbuf + 32*(size_t)q - TB, *rq = r + 6*(size_t)(q + B) - (TB + 22);
        v2d vv[SV_PU];
#pragma unroll
        for (int u = 0; u < SV_PU; u++) { const int x = 2*(pj + SV_PT*u); vv[u] = v2d{0.0, 0.0};
            if (on && x < TB) { const int d1 = x/36; if (q + 1 + d1 < rend) vv[u] = *(const v2d *)(Lq + (size_t)d1*REC + x); }       // L(R, q), R = q + 1 .. q + B, at [(R - q - 1) 36 + 6 col + row]
            else if (on && x < TB + 22) vv[u] = *(const v2d *)(ldq + x);
            else if (on && x < TB + 28) { if (q + B < rend) { const v2d e = *(const v2d *)(rq + x); vv[u] = v2d{rs*e.x, rs*e.y}; } } }     // the block entering the window after pivot q
        double *dst = stg + ((size_t)(c & 1)*SV_K + pk)*CS;
#pragma unroll
        for (int u = 0; u < SV_PU; u++) { const int xp = pj + SV_PT*u; if (on && xp < perp) *(v2d *)(dst + 2*xp) = vv[u]; }
    };
    const int bbr = ptid % 96, bhalf = ptid/96;                 // border: row of the separator on the left, a quarter of a chunk's pivots (SV_BQ of them)
    double bacc = 0.0;
    auto border = [&](int c) {                                  // - Lb_q w_q over the pivots of chunk c
        if (bbr >= bw || bhalf >= SV_BG) return;
        v2d lv[SV_BQ][3];
#pragma unroll
        for (int k = 0; k < SV_BQ; k++) { const int q = a + c*SV_K + bhalf*SV_BQ + k; const v2d *Lq = (const v2d *)(Lb + (size_t)q*REC + 6*bbr);
#pragma unroll
            for (int cc = 0; cc < 3; cc++) lv[k][cc] = q < b ? Lq[cc] : v2d{0.0, 0.0}; }
#pragma unroll
        for (int k = 0; k < SV_BQ; k++) {
            if (a + c*SV_K + bhalf*SV_BQ + k >= b) break;        // (past the interior: nothing was written to wch)
            const double *wk = wch + ((size_t)(c & 1)*SV_K + bhalf*SV_BQ + k)*6;
#pragma unroll
            for (int cc = 0; cc < 3; cc++) bacc = fma(-lv[k][cc].x, wk[2*cc], fma(-lv[k][cc].y, wk[2*cc + 1], bacc)); }
    };
    // window: the B blocks q .. q + B - 1 before a step, q + 1 .. q + B after it; block R in slot R mod B, row (slot, k) on lane / register (6 slot + k) mod 64, / 64
    int slot[NREG], ri[NREG], dd[NREG]; bool rowok[NREG]; double t[NREG];
#pragma unroll
    for (int g = 0; g < NREG; g++) { const int rho = lane + 64*g; rowok[g] = rho < 6*B; slot[g] = rho/6; ri[g] = rho - 6*slot[g]; t[g] = 0.0; dd[g] = 1; }
    int sq = a % B;
    if (wave > 0) stage(0);
    else {
#pragma unroll
        for (int g = 0; g < NREG; g++) { const int d = (slot[g] - sq + B) % B, R = a + d; t[g] = (rowok[g] && R < rend) ? rs*r[6*(size_t)R + ri[g]] : 0.0;
            dd[g] = d == 0 ? B : d; }                           // distance to the pivot AFTER the pivot's slot has been handed to the entering block
    }
    __syncthreads();
    SVS();
    for (int c = 0; c < nch; c++) {
        if (wave > 0) { if (c + 1 < nch) stage(c + 1); if (p > 0 && c > 0) border(c - 1); }
        else {
            const int q0 = a + c*SV_K, nk = min(SV_K, b - q0);
            const double *sk0 = stg + (size_t)(c & 1)*SV_K*CS;
            // the requests of a step do not depend on the running solution: those of step k + 1 are issued before the arithmetic of step k (two register sets)
            auto load = [&](SvStep<NREG> &S, int k, const int (&dn)[NREG]) {
                const double *sk = sk0 + (size_t)k*CS;
#pragma unroll
                for (int g = 0; g < NREG; g++) { const double *cp = sk + (rowok[g] ? (dn[g] - 1)*36 + ri[g] : 0);
#pragma unroll
                    for (int cc = 0; cc < 6; cc++) S.cf[g][cc] = cp[6*cc];
                    S.ent[g] = sk[TB + 22 + ri[g]]; }
#pragma unroll
                for (int e = 0; e < 15; e++) S.l[e] = sk[TB + e];
                S.idl = sk[TB + 16 + (lane < 6 ? lane : 0)];
            };
            auto exec = [&](const SvStep<NREG> &S, int k) {
                double w[6];                                    // the pivot block: broadcast, unit-lower solve on uniform values
#pragma unroll
                for (int cc = 0; cc < 6; cc++) w[cc] = sv_pivot<NREG>(t, 6*sq + cc);
#pragma unroll
                for (int i = 1; i < 6; i++)
#pragma unroll
                    for (int j = 0; j < i; j++) w[i] = fma(-S.l[tri(i - 1) + j], w[j], w[i]);
                if (lane < 6) { double wl = w[0];
#pragma unroll
                    for (int cc = 1; cc < 6; cc++) wl = lane == cc ? w[cc] : wl;
                    wch[((size_t)(c & 1)*SV_K + k)*6 + lane] = wl; M.V[6*(size_t)(q0 + k) + lane] = wl*S.idl; }
#pragma unroll
                for (int g = 0; g < NREG; g++) {                // its slot goes to the entering block; every row of the window takes the pivot's contribution
                    double tv = (rowok[g] && slot[g] == sq) ? S.ent[g] : t[g];
#pragma unroll
                    for (int cc = 0; cc < 6; cc++) tv = fma(-S.cf[g][cc], w[cc], tv);
                    t[g] = rowok[g] ? tv : 0.0;
                    dd[g] = dd[g] == 1 ? B : dd[g] - 1;
                }
                sq = sq + 1 == B ? 0 : sq + 1;
            };
            auto nextd = [&](int (&dn)[NREG]) {
#pragma unroll
                for (int g = 0; g < NREG; g++) dn[g] = dd[g] == 1 ? B : dd[g] - 1;
            };
            SvStep<NREG> S0, S1; int dn[NREG];
            load(S0, 0, dd);
            for (int k = 0; k < nk; k += 2) {
                if (k + 1 < nk) { nextd(dn); load(S1, k + 1, dn); }
                exec(S0, k);
                if (k + 1 < nk) {
                    if (k + 2 < nk) { nextd(dn); load(S0, k + 2, dn); }
                    exec(S1, k + 1);
                }
            }
            SVS();
        }
        __syncthreads();
        SVS();
    }
#ifdef SV_STAMPS
    if (tid == 0) for (int k = 0; k < 8; k++) M.Wm[8*(size_t)p + k] = k < nstamp ? (double)(stamps[k] - stamps[0]) : -1.0;
#endif
    if (wave == 0 && p < P - 1) {                               // what the window holds now: the separator's rows, minus this interior's part
#pragma unroll
        for (int g = 0; g < NREG; g++) { const int d = (slot[g] - b % B + B) % B;
            if (rowok[g]) M.G[(size_t)p*bw + 6*d + ri[g]] = t[g]; }
    }
    if (p == 0) return;
    if (wave > 0) { border(nch - 1); if (bhalf < SV_BG) red[bhalf*96 + bbr] = bacc; }
    __syncthreads();
    if (tid < bw) M.G2[(size_t)(p - 1)*bw + tid] = (red[tid] + red[96 + tid]) + (red[192 + tid] + red[288 + tid]);
}
#undef SVS

// ---- interiors, backward.  grid Pmax, SV_T threads: the border part  v_q -= Lb_q^T x_left  by all threads into LDS (vc), then the chain on wave 0 with the
// other waves staging.  M.X = the solution (the rows of the separator on the right written along).  lmax: bound of an interior's length (host).
// POLL (k_sv_tree_back): the interior runs in the launch of the separator tree -- the solutions of its two separators (M.Xs) are polled where the
// kernel of its own reads them, after everything that does not depend on them has been requested
__device__ __forceinline__ double sv_poll1(const double *p, bool on) {
    double v = 0.0;
    if (on) { v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spins = 0; v != v && spins < (1 << 15); spins++) { __builtin_amdgcn_s_sleep(1); v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (v != v) { v = __builtin_inf(); atomicAdd(&ts_poll_giveups, 1u); } }
    return v;
}
template <int NREG, bool POLL>
__device__ __forceinline__ void sv_back_int_body(const Work &W, int bw, int Pmax, int lmax, const double *__restrict__ Lrow, const double *__restrict__ Lb, const MsBuf &M, const double *__restrict__ rdot, double *__restrict__ rz_part, double *ms_smem, int p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = ms_uni(tid >> 6);
    const LmState *st_ = W.st; const int flags = st_->done | st_->lin_done | st_->step_fail, nf = ms_uni(*W.nfree);
    double xlv = (!POLL && p > 0 && tid < bw) ? M.Xs[(size_t)(p - 1)*bw + tid] : 0.0;      // (requested along with the flags: the address does not depend on them)
    sv_pin(xlv);
    if (flags || nf <= 0) return;
    const int B = bw/6, CS = sv_cs(B), TB = 36*B;
    double *stg = ms_smem, *vc = stg + 2*(size_t)SV_K*CS, *xl = vc + 6*(size_t)lmax;     // stg [2][SV_K][CS], vc [6 lmax]: v of the interior's rows with the border part taken off, xl [80]
    const BandpPart PT = bandp_part(nf, B, Pmax, p);
    const int a = ms_uni(PT.a), b = ms_uni(PT.b), P = ms_uni(PT.P);
    if (p >= P) { if (rdot && tid == 0) rz_part[p] = 0.0; return; }
    if (b - a > lmax) { if (tid == 0) W.st->step_fail = 1; return; }       // (cannot happen: sv_lmax bounds what bandp_part produces; a failed step, not a wrong one)
    const int REC = bw*6, rtop = p < P - 1 ? b + B : b;         // pivots rtop - 1 .. a (the separator on the right first: its solution is known)
    const int ptid = tid - 64, pk = ptid & (SV_K - 1), pj = ptid/SV_K, perp = (TB + (rdot ? 28 : 22))/2;
    const int nst = rtop - a, nch = (nst + SV_K - 1)/SV_K;
    SvStage S;
    auto stage_load = [&](int c) {                              // producers: chunk c = the pivots rtop - 1 - c SV_K - pk
        const int R = rtop - 1 - c*SV_K - pk; const bool on = R >= a;
        const double *LR = Lrow + (size_t)R*REC, *ldR = W.LDbuf + 32*(size_t)R - TB;
#pragma unroll
        for (int u = 0; u < SV_PU; u++) { const int x = 2*(pj + SV_PT*u); S.v[u] = v2d{0.0, 0.0};
            if (on && x < TB) S.v[u] = *(const v2d *)(LR + x);  // L(R, q), q = R - 1 .. R - B, at [(R - q - 1) 36 + 6 col + row]
            else if (on && x < TB + 22) { if (R < b) S.v[u] = *(const v2d *)(ldR + x); }
            else if (on && rdot && x < TB + 28) S.v[u] = *(const v2d *)(rdot + 6*(size_t)R + (x - TB - 22)); }     // r of the pivot's rows (for r.z)
    };
    auto stage_store = [&](int c) {
        const int R = rtop - 1 - c*SV_K - pk;
        double *dst = stg + ((size_t)(c & 1)*SV_K + pk)*CS;
#pragma unroll
        for (int u = 0; u < SV_PU; u++) { const int xp = pj + SV_PT*u; if (R >= a && xp < perp) *(v2d *)(dst + 2*xp) = S.v[u]; }
    };
    // window: the blocks R - B + 1 .. R before a step, R - B .. R - 1 after it; block q in slot q mod B
    int slot[NREG], ri[NREG], dd[NREG]; bool rowok[NREG]; double t[NREG], tsep[NREG];
#pragma unroll
    for (int g = 0; g < NREG; g++) { const int rho = lane + 64*g; rowok[g] = rho < 6*B; slot[g] = rho/6; ri[g] = rho - 6*slot[g]; t[g] = 0.0; dd[g] = 1; tsep[g] = 0.0; }
    int sR = (rtop - 1) % B;
    if (wave > 0) stage_load(0);
    else {
#pragma unroll
        for (int g = 0; g < NREG; g++) { const int d = (sR - slot[g] + B) % B, q = rtop - 1 - d;
            tsep[g] = (!POLL && rowok[g] && q >= b) ? M.Xs[(size_t)p*bw + 6*(q - b) + ri[g]] : 0.0; }
    }
    {   // vc = v - Lb^T x_left: four lanes per output row, two passes of 128 outputs in flight
        if (!POLL && tid < 80) xl[tid] = xlv;
        const int part = tid & 3, eo = tid >> 2, nout = 6*(b - a);
        bool first = true;
        for (int e0 = 0; e0 < nout; e0 += 2*(SV_T/4)) {
            double lv[2][20], v0[2];
#pragma unroll
            for (int ps = 0; ps < 2; ps++) { const int e = e0 + ps*(SV_T/4) + eo; const bool ok = e < nout; const int q = a + e/6, cc = e % 6;
                const double *Lq = Lb + (size_t)q*REC + cc;
#pragma unroll
                for (int j = 0; j < 20; j++) { const int br = part + 4*j; lv[ps][j] = (ok && p > 0 && br < bw) ? Lq[6*br] : 0.0; }
                v0[ps] = (ok && part == 0) ? M.V[6*(size_t)q + cc] : 0.0; }
            if (first) { if (POLL && tid < 80) xl[tid] = sv_poll1(&M.Xs[(size_t)max(p - 1, 0)*bw + tid], p > 0 && tid < bw);
                __syncthreads(); first = false; }      // (xl)
#pragma unroll
            for (int ps = 0; ps < 2; ps++) { const int e = e0 + ps*(SV_T/4) + eo; const bool ok = e < nout;
                double acc = 0.0;
#pragma unroll
                for (int j = 0; j < 20; j++) { const int br = part + 4*j; if (br < bw) acc = fma(-lv[ps][j], xl[br], acc); }
                acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);
                if (ok && part == 0) vc[e] = v0[ps] + acc; }
        }
        if (first) __syncthreads();
    }
    if (wave > 0) stage_store(0);
    if (POLL && wave == 0) {
#pragma unroll
        for (int g = 0; g < NREG; g++) { const int d = (sR - slot[g] + B) % B, q = rtop - 1 - d; const bool on = rowok[g] && q >= b;
            tsep[g] = sv_poll1(&M.Xs[(size_t)p*bw + (on ? 6*(q - b) + ri[g] : 0)], on); }
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int g = 0; g < NREG; g++) { const int d = (sR - slot[g] + B) % B, q = rtop - 1 - d;
            t[g] = (rowok[g] && q >= a) ? (q >= b ? tsep[g] : vc[6*(q - a) + ri[g]]) : 0.0;
            dd[g] = d == 0 ? B : d; }
    }
    double racc = 0.0;
    for (int c = 0; c < nch; c++) {
        if (wave > 0) { if (c + 1 < nch) { stage_load(c + 1); stage_store(c + 1); } }
        else {
            const int R0 = rtop - 1 - c*SV_K, nk = min(SV_K, R0 - a + 1);
            const double *sk0 = stg + (size_t)(c & 1)*SV_K*CS;
            auto load = [&](SvStep<NREG> &T, int k, const int (&dn)[NREG]) {
                const double *sk = sk0 + (size_t)k*CS; const int R = R0 - k;
#pragma unroll
                for (int g = 0; g < NREG; g++) {
                    const int q = R - dn[g]; const bool on = rowok[g] && q >= a && q < b;          // (rows of the separator itself take nothing)
                    const double *cp = sk + (on ? (dn[g] - 1)*36 + 6*ri[g] : 0);
#pragma unroll
                    for (int k2 = 0; k2 < 6; k2++) { const double cv = cp[k2]; T.cf[g][k2] = on ? cv : 0.0; }
                    const int qn = R - B; T.ent[g] = qn >= a ? vc[6*(qn - a) + ri[g]] : 0.0; }       // the block entering the window
#pragma unroll
                for (int e = 0; e < 15; e++) T.l[e] = sk[TB + e];
                T.idl = rdot ? sk[TB + 22 + (lane < 6 ? lane : 0)] : 0.0;      // (r of the pivot's row `lane`)
            };
            auto exec = [&](const SvStep<NREG> &T, int k) {
                const int R = R0 - k;
                double x[6];
#pragma unroll
                for (int k2 = 0; k2 < 6; k2++) x[k2] = sv_pivot<NREG>(t, 6*sR + k2);
#pragma unroll
                for (int i = 4; i >= 0; i--)
#pragma unroll
                    for (int j = i + 1; j < 6; j++) x[i] = fma(-T.l[tri(j - 1) + i], x[j], x[i]);
                if (lane < 6) { double xo = x[0];
#pragma unroll
                    for (int k2 = 1; k2 < 6; k2++) xo = lane == k2 ? x[k2] : xo;
                    M.X[6*(size_t)R + lane] = xo; racc = fma(T.idl, xo, racc); }
#pragma unroll
                for (int g = 0; g < NREG; g++) {
                    double tv = (rowok[g] && slot[g] == sR) ? T.ent[g] : t[g];
#pragma unroll
                    for (int k2 = 0; k2 < 6; k2++) tv = fma(-T.cf[g][k2], x[k2], tv);
                    t[g] = rowok[g] ? tv : 0.0;
                    dd[g] = dd[g] == 1 ? B : dd[g] - 1;
                }
                sR = sR == 0 ? B - 1 : sR - 1;
            };
            auto nextd = [&](int (&dn)[NREG]) {
#pragma unroll
                for (int g = 0; g < NREG; g++) dn[g] = dd[g] == 1 ? B : dd[g] - 1;
            };
            SvStep<NREG> S0, S1; int dn[NREG];
            load(S0, 0, dd);
            for (int k = 0; k < nk; k += 2) {
                if (k + 1 < nk) { nextd(dn); load(S1, k + 1, dn); }
                exec(S0, k);
                if (k + 1 < nk) {
                    if (k + 2 < nk) { nextd(dn); load(S0, k + 2, dn); }
                    exec(S1, k + 1);
                }
            }
        }
        __syncthreads();
    }
    if (rdot && wave == 0) { const double sr = wave_sum1(lane < 6 ? racc : 0.0); if (lane == 0) rz_part[p] = sr; }
}
template <int NREG>
__global__ __launch_bounds__(SV_T) void k_sv_back_int(Work W, int bw, int Pmax, int lmax, const double *__restrict__ Lrow, const double *__restrict__ Lb, MsBuf M, const double *__restrict__ rdot, double *__restrict__ rz_part) {
    extern __shared__ __attribute__((aligned(16))) double ms_smem[];
    sv_back_int_body<NREG, false>(W, bw, Pmax, lmax, Lrow, Lb, M, rdot, rz_part, ms_smem, (int)blockIdx.x);
}

// ---- the inverse of every separator's unit-lower factor, once per factorisation: M.Li [label][s][s] row-major (zeros above the diagonal), M.Lid [label][s] = 1/d.
// grid labels, SV_LT threads = six per column (one per row of a pose block).  Block recursion on the packed record (in LDS) with the 6 x 6 inverses
// of the diagonal blocks that the factorisation already left in the record's table:
//   Linv(I, I) = inv(l_I),   Linv(I, J) = - inv(l_I) sum_{K = J .. I-1} L(I, K) Linv(K, J)   for the block rows I = 1, 2, ... in turn
// -- 300 FMAs per thread and two barriers per block row (a thread per column solving L y = e_j row by row: 1800 dependent FMAs, 64 us per launch).
#define SV_LT 512
static size_t sv_linv_lds_doubles(int s) { return ((cre_rec_doubles(s) + 8) & ~(size_t)1) + (size_t)s*(s + 1) + std::max(6*(size_t)(s + 2), (size_t)16*((s + 15)/16)*(s + 3)); }
__global__ __launch_bounds__(SV_LT) void k_sv_linv(Work W, int bw, int Pmax, const double *__restrict__ fac, const double *__restrict__ pool, MsBuf M, double *xreset) {
    extern __shared__ __attribute__((aligned(16))) double ms_smem[];
    const int tid = threadIdx.x, s = bw, B = s/6, i = blockIdx.x;
    const LmState *st_ = W.st; const int flags = st_->done | st_->step_fail, nf = ms_uni(*W.nfree);
    if (flags) return;
    if (i >= sv_nsep(nf, B, Pmax)) return;
    if (xreset && i > 0 && tid < s) xreset[(size_t)i*s + tid] = __builtin_nan("");       // (k_cre_back_tree polls the separators' solution: not there yet)
    const size_t nrec = cre_rec_doubles(s);
    double *recl = ms_smem, *Y = recl + ((nrec + 8) & ~(size_t)1), *T = Y + (size_t)s*(s + 1);      // packed record | Linv [s][s + 1] | T [6][s + 2]
    const double *rec = fac + (size_t)i*nrec;
    for (size_t e = tid; e < nrec; e += SV_LT) recl[e] = rec[e];
    __syncthreads();
    const double *LDt = recl + rowoff(s);
    if (tid < s) M.Lid[(size_t)i*s + tid] = LDt[SOLVE_LD*(tid/6) + LD_ID + tid % 6];
    const int j = tid/6, r = tid - 6*j, J = j/6, q = j - 6*J; const bool on = j < s;
    // diagonal blocks (and zeros everywhere else of this thread's column)
    if (on) for (int I = 0; I < B; I++) Y[(size_t)(6*I + r)*(s + 1) + j] = I != J ? 0.0 : (r == q ? 1.0 : (r > q ? LDt[SOLVE_LD*I + LD_M + tri(r - 1) + q] : 0.0));
    __syncthreads();
    for (int I = 1; I < B; I++) {
        const bool act = on && J < I;
        if (act) {
            const double *Lr = recl + rowoff(6*I + r);          // row 6 I + r of the packed factor: entries of the blocks K < I
            double a0 = 0.0, a1 = 0.0;
            for (int K = J; K < I; K++) {
                const double *yk = Y + (size_t)(6*K)*(s + 1) + j;
#pragma unroll
                for (int c = 0; c < 6; c += 2) { a0 = fma(Lr[6*K + c], yk[(size_t)c*(s + 1)], a0); a1 = fma(Lr[6*K + c + 1], yk[(size_t)(c + 1)*(s + 1)], a1); }
            }
            T[r*(s + 2) + j] = a0 + a1;
        }
        __syncthreads();
        if (act) {
            const double *mi = LDt + SOLVE_LD*I + LD_M;
            double y = T[r*(s + 2) + j];
            for (int q2 = 0; q2 < r; q2++) y = fma(mi[tri(r - 1) + q2], T[q2*(s + 2) + j], y);
            Y[(size_t)(6*I + r)*(s + 1) + j] = -y;
        }
        __syncthreads();
    }
    double *out = M.Li + (size_t)i*s*s;
    for (int e = tid; e < s*s; e += SV_LT) { const int rr = e/s, cc = e - rr*s; out[e] = cc <= rr ? Y[(size_t)rr*(s + 1) + cc] : 0.0; }
    // the products the separator steps apply (M.Pp [label][2][s][s]): P_a = X_a L^-1, P_c = X_c L^-1 with the couplings X of this pivot to its neighbours
    // a = i - h, c = i + h as the elimination left them in the pool.  One block at a time: X through LDS (T's place, [rows padded to 16][s + 3] with the
    // pad columns zero), 16 x 16 tiles of X Y on the matrix cores (v_mfma_f64_16x16x4: lane (lr, lk) feeds X[16 ti + lr][t + lk] and Y[t + lk][16 tj + lr]
    // and holds P[16 ti + lk + 4 r][16 tj + lr], r = 0 .. 3).  (A thread per four columns of a row with the sums in registers: 72 of the kernel's 86 us.)
    if (i == 0) return;                                         // (the root has no neighbours)
    const int h = i & -i, ia = i - h, ic = i + h, mmax = cr_mmax(W.ring, Pmax, W.ring_g), m = sv_nsep(nf, B, Pmax);
    double *XS = T;
    const int SX = s + 3, nt = (s + 15)/16, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
    for (int which = 0; which < 2; which++) {
        const bool have = which == 0 || ic < m;                 // (uniform)
        const double *X = cr_blk(pool, s, mmax, which && have ? ic : i, which && have ? i : ia);
        __syncthreads();
        for (int e0 = 0; e0 < 16*nt*SX; e0 += 6*SV_LT) {        // rows >= s and the pad columns: zero
            double xv[6];
#pragma unroll
            for (int u = 0; u < 6; u++) { const int e = e0 + u*SV_LT + tid, rr = e/SX, cc = e - rr*SX; xv[u] = (have && rr < s && cc < s) ? X[(size_t)rr*s + cc] : 0.0; }
#pragma unroll
            for (int u = 0; u < 6; u++) { const int e = e0 + u*SV_LT + tid; if (e < 16*nt*SX) XS[e] = xv[u]; }
        }
        __syncthreads();
        double *o = M.Pp + ((size_t)i*2 + which)*s*s;
        for (int tile = wave; tile < nt*nt; tile += SV_LT/64) {
            const int ti = tile/nt, tj = tile - ti*nt;
            v4d c = {0.0, 0.0, 0.0, 0.0};
            const double *pa = XS + (size_t)(16*ti + lr)*SX + lk;
            const int col = 16*tj + lr;
            for (int t0 = 0; t0 < s; t0 += 4) { const int t = t0 + lk;
                const double av = pa[t0], bv = (t < s && col < s) ? Y[(size_t)t*(s + 1) + col] : 0.0;
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0); }
#pragma unroll
            for (int r = 0; r < 4; r++) { const int R = 16*ti + lk + 4*r; if (R < s && col < s) o[(size_t)R*s + col] = c[r]; }
        }
    }
}

// ---- the separator kernels: every operand requested at once, speculatively (every address is inside its allocation whatever the number of
// separators turns out to be), one wait.
// pending updates of block blk's right-hand side row `row` (the producers of ms_pending): pivots blk -+ 2^l of level 2^l, 2^l < H
struct SvPend { double lo_[8], hi_[8]; };
__device__ __forceinline__ void sv_pending_load(const MsBuf &M, int s, int blk, int mmax, int row, bool on, SvPend &P) {
#pragma unroll
    for (int l = 0; l < 8; l++) {
        const int hp = 1 << l, pl = blk - hp, pr = blk + hp;
        P.lo_[l] = (on && pl >= 0) ? M.Cg[((size_t)pl*2 + 1)*s + row] : 0.0;
        P.hi_[l] = (on && pr < mmax) ? M.Cg[((size_t)pr*2 + 0)*s + row] : 0.0;
    }
}
__device__ __forceinline__ void sv_pin(SvPend &P) {
#pragma unroll
    for (int l = 0; l < 8; l++) { sv_pin(P.lo_[l]); sv_pin(P.hi_[l]); }
}
__device__ __forceinline__ double sv_pending_sum(const SvPend &P, int blk, int H, int lo, int m, int r0, double v) {      // the order of ms_pending
#pragma unroll
    for (int l = 0; l < 8; l++) {
        const int hp = 1 << l; const bool lev = hp < H && hp < m - lo;
        const int pl = blk - hp, pr = blk + hp;
        if (lev && pl >= lo && pl != r0 && (pl & (2*hp - 1)) == hp) v -= P.lo_[l];
        if (lev && pr < m && pr != r0 && (pr & (2*hp - 1)) == hp) v -= P.hi_[l];
    }
    return v;
}
// rows of a dense [.. ][s] block against a vector in LDS: four lanes per row (16-byte pieces 2 part + 8 j)
struct SvRow { v2d x[10]; };
__device__ __forceinline__ void sv_row_load(const double *row, int s, int part, bool on, SvRow &R) {
#pragma unroll
    for (int j = 0; j < 10; j++) { const int k = 2*part + 8*j; R.x[j] = (on && k < s) ? *(const v2d *)(row + k) : v2d{0.0, 0.0}; }
}
__device__ __forceinline__ void sv_pin(SvRow &R) {
#pragma unroll
    for (int j = 0; j < 10; j++) sv_pin(R.x[j]);
}
__device__ __forceinline__ double sv_row_dot(const SvRow &R, const double *v, int s, int part) {     // every lane of the wave takes part (shuffles)
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 10; j++) { const int k = 2*part + 8*j; if (k < s) acc = fma(R.x[j].x, v[k], fma(R.x[j].y, v[k + 1], acc)); }
    acc += __shfl_xor(acc, 1, 64); acc += __shfl_xor(acc, 2, 64);
    return acc;
}

// columns of a dense [s][s] block against a vector in LDS (the transposed product): six groups of rows t = g + 6 j, a thread per column
struct SvCol { double x[13]; };
__device__ __forceinline__ void sv_col_load(const double *X, int s, int g, int r, bool on, SvCol &C) {
#pragma unroll
    for (int j = 0; j < 13; j++) { const int t = g + 6*j; C.x[j] = (on && t < s) ? X[(size_t)t*s + r] : 0.0; }
}
__device__ __forceinline__ void sv_pin(SvCol &C) {
#pragma unroll
    for (int j = 0; j < 13; j++) sv_pin(C.x[j]);
}
__device__ __forceinline__ double sv_col_dot(const SvCol &C, const double *v, int s, int g) {
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < 13; j++) { const int t = g + 6*j; if (t < s) acc = fma(C.x[j], v[t], acc); }
    return acc;
}


#define SV_SPIN_MAX (1 << 15)
__device__ __forceinline__ double sv_ld_co(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sv_st_co(double *p, double v) { __hip_atomic_store(p, v == v ? v : __builtin_inf(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void sv_pending_poll(const MsBuf &M, int s, int blk, int H, int m, int row, bool on, SvPend &P) {
    unsigned need = 0;                                          // bit l: from blk - 2^l, bit 8 + l: from blk + 2^l (the terms sv_pending_sum takes, lo = r0 = 0)
#pragma unroll
    for (int l = 0; l < 8; l++) { const int hp = 1 << l, pl = blk - hp, pr = blk + hp; const bool lev = on && hp < H && hp < m;
        if (lev && pl >= 1 && (pl & (2*hp - 1)) == hp) need |= 1u << l;
        if (lev && pr < m && (pr & (2*hp - 1)) == hp) need |= 1u << (8 + l);
        P.lo_[l] = 0.0; P.hi_[l] = 0.0; }
    for (int spins = 0; need && spins < SV_SPIN_MAX; spins++) {  // every outstanding request of a round in flight at once; a slot is final once it is not a NaN
#pragma unroll
        for (int l = 0; l < 8; l++) { const int hp = 1 << l;
            if (need >> l & 1) P.lo_[l] = sv_ld_co(&M.Cg[((size_t)(blk - hp)*2 + 1)*s + row]);
            if (need >> (8 + l) & 1) P.hi_[l] = sv_ld_co(&M.Cg[((size_t)(blk + hp)*2 + 0)*s + row]); }
#pragma unroll
        for (int l = 0; l < 8; l++) {
            if ((need >> l & 1) && P.lo_[l] == P.lo_[l]) need &= ~(1u << l);
            if ((need >> (8 + l) & 1) && P.hi_[l] == P.hi_[l]) need &= ~(1u << (8 + l)); }
        if (need) __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int l = 0; l < 8; l++) { if (need >> l & 1) P.lo_[l] = __builtin_inf(); if (need >> (8 + l) & 1) P.hi_[l] = __builtin_inf(); }     // (gave up: the result is not finite)
    if (need) atomicAdd(&ts_poll_giveups, 1u);
}
__device__ __forceinline__ void sv_poll2(const double *pa, bool on_a, const double *pc, bool on_c, double &va, double &vc) {      // both requests in one round trip
    va = 0.0; vc = 0.0;
    bool na = on_a, nc = on_c;
    for (int spins = 0; (na || nc) && spins < SV_SPIN_MAX; spins++) {
        const double ta = na ? sv_ld_co(pa) : 0.0, tc = nc ? sv_ld_co(pc) : 0.0;
        if (na && ta == ta) { va = ta; na = false; }
        if (nc && tc == tc) { vc = tc; nc = false; }
        if (na || nc) __builtin_amdgcn_s_sleep(1);
    }
    if (na) va = __builtin_inf();
    if (nc) vc = __builtin_inf();
    if (na || nc) atomicAdd(&ts_poll_giveups, 1u);
}

// ---- the separator steps.  With the products P_a = X_a L^-1, P_c = X_c L^-1 that k_sv_linv leaves next to L^-1 (once per factorisation), a pivot's
// forward step has ONE product on the path to its neighbours -- [P_a; P_c] v, v = g - pending -- and z = D^-1 L^-1 v beside it; its back substitution is
//   x_i = L^-T z - P_a^T x_a - P_c^T x_c:   L^-T z before the neighbours' solutions arrive, one product after.
// (X_a (L^-1 v) and L^-T (z - X_a^T x_a - X_c^T x_c): two products and four barriers in sequence on each way -- 2.3 us per level by wall-clock stamps.)
struct SvFwd { double g0, g1, idq[2]; SvRow row[2]; };          // rows of [P_a; P_c; L^-1] (3 s <= 234 of them), 128 per pass, four lanes each
__device__ __forceinline__ void sv_fwd_load(const MsBuf &M, int s, int i, int tid, SvFwd &F) {
    const bool vrow = tid < s; const int part = tid & 3, rq = tid >> 2;
    F.g0 = vrow ? M.G[(size_t)i*s + tid] : 0.0; F.g1 = vrow ? M.G2[(size_t)i*s + tid] : 0.0;
#pragma unroll
    for (int ps = 0; ps < 2; ps++) { const int R = ps*(SV_CT/4) + rq; const bool isp = R < 2*s, isl = !isp && R < 3*s;
        const double *row = isp ? M.Pp + ((size_t)i*2*s + R)*s : M.Li + ((size_t)i*s + (isl ? R - 2*s : 0))*s;      // rows 0 .. 2 s - 1 of [label][2][s][s]: P_a, then P_c
        F.idq[ps] = isl ? M.Lid[(size_t)i*s + R - 2*s] : 0.0;
        sv_row_load(row, s, part, isp || isl, F.row[ps]); }
    sv_pin(F.g0); sv_pin(F.g1); sv_pin(F.idq[0]); sv_pin(F.idq[1]); sv_pin(F.row[0]); sv_pin(F.row[1]);
}
// the products of a forward step with v (LDS): f(R, value) for this thread's rows R < 2 s of [P_a; P_c] v, z[r] = (L^-1 v)[r] / d[r] into zz (LDS)
template <class F2>
__device__ __forceinline__ void sv_fwd_products(const SvFwd &F, const double *v, int s, int tid, double *zz, F2 f) {
    const int part = tid & 3, rq = tid >> 2;
#pragma unroll
    for (int ps = 0; ps < 2; ps++) { const int R = ps*(SV_CT/4) + rq;
        const double d = sv_row_dot(F.row[ps], v, s, part);
        if (part == 0) { if (R < 2*s) f(R, d); else if (R < 3*s) zz[R - 2*s] = d*F.idq[ps]; } }
}
struct SvBack { SvCol ca, cc, lc; };
__device__ __forceinline__ void sv_back_load(const MsBuf &M, int s, int i, int g, int r, bool con, bool has_c, SvBack &K) {
    sv_col_load(M.Pp + ((size_t)i*2 + 0)*s*s, s, g, con ? r : 0, con, K.ca); sv_col_load(M.Pp + ((size_t)i*2 + 1)*s*s, s, g, con ? r : 0, con && has_c, K.cc);
    sv_col_load(M.Li + (size_t)i*s*s, s, g, con ? r : 0, con, K.lc);
    sv_pin(K.ca); sv_pin(K.cc); sv_pin(K.lc);
}
#define SV_SUM6(red, k) (((red[k] + red[80 + (k)]) + (red[160 + (k)] + red[240 + (k)])) + (red[320 + (k)] + red[400 + (k)]))

// ---- cyclic reduction, level h, forward.  grid pivots, SV_CT threads:  updates [P_a; P_c] v for the neighbours, z = D^-1 L^-1 v,  v = g - pending
__global__ __launch_bounds__(SV_CT) void k_sv_cre_fwd(Work W, Work Ws, int bw, int Pmax, int h, int kb, MsBuf M) {
    __shared__ __attribute__((aligned(16))) double v[80];
    const int tid = threadIdx.x;
    const int s = bw, B = s/6, mmax = cr_mmax(W.ring, Pmax, W.ring_g);
    const int i = (2*(kb + (int)blockIdx.x) + 1)*h, ia = i - h, ic = i + h;             // (i < mmax by the launch; ia >= 0)
    const LmState *st_ = W.st; const int flags = st_->done | st_->lin_done | st_->step_fail, nf = *W.nfree;
    const bool vrow = tid < s;
    SvFwd F; sv_fwd_load(M, s, i, tid, F);
    SvPend pd; sv_pending_load(M, s, i, mmax, tid, vrow, pd); sv_pin(pd);
    if (flags) return;
    const int m = ms_uni(sv_nsep(nf, B, Pmax)), lo = 0, r0 = 0;
    if (i < lo || i >= m) return;
    const bool has_a = ia >= lo, has_c = ic < m;
    if (vrow) v[tid] = sv_pending_sum(pd, i, h, lo, m, r0, F.g0 + F.g1);
    __syncthreads();
    sv_fwd_products(F, v, s, tid, M.Z + (size_t)i*s, [&](int R, double acc) { M.Cg[((size_t)i*2 + (R < s ? 0 : 1))*s + (R < s ? R : R - s)] = (R < s ? has_a : has_c) ? acc : 0.0; });
}

// ---- the last block: forward and backward.  One workgroup.
__global__ __launch_bounds__(SV_CT) void k_sv_cre_root(Work W, int bw, int Pmax, MsBuf M) {
    __shared__ __attribute__((aligned(16))) double v[80], w[80], red[6*80];
    const int tid = threadIdx.x;
    const int s = bw, B = s/6, mmax = cr_mmax(W.ring, Pmax, W.ring_g), r0 = 0;            // (chains: the root is label 0)
    const LmState *st_ = W.st; const int flags = st_->done | st_->lin_done | st_->step_fail, nf = *W.nfree;
    const bool vrow = tid < s;
    double g0 = vrow ? M.G[(size_t)r0*s + tid] : 0.0, g1 = vrow ? M.G2[(size_t)r0*s + tid] : 0.0, idv = vrow ? M.Lid[(size_t)r0*s + tid] : 0.0;
    SvPend pd; sv_pending_load(M, s, r0, mmax, tid, vrow, pd);
    const int part = tid & 3, rq = tid >> 2, g = tid/80, r = tid - 80*g; const bool con = g < 6 && r < s;
    SvRow li; SvCol lc;
    sv_row_load(M.Li + ((size_t)r0*s + (rq < s ? rq : 0))*s, s, part, rq < s, li);
    sv_col_load(M.Li + (size_t)r0*s*s, s, g, con ? r : 0, con, lc);
    sv_pin(g0); sv_pin(g1); sv_pin(idv); sv_pin(pd); sv_pin(li); sv_pin(lc);
    if (flags) return;
    const int m = ms_uni(sv_nsep(nf, B, Pmax)), lo = 0;
    if (m <= 0) return;
    if (vrow) v[tid] = sv_pending_sum(pd, r0, 1 << 30, lo, m, -1, g0 + g1);
    __syncthreads();
    { const double wv = sv_row_dot(li, v, s, part);
      if (rq < s && part == 0) w[rq] = wv; }
    __syncthreads();
    if (vrow) v[tid] = w[tid]*idv;                              // z
    __syncthreads();
    if (g < 6) red[g*80 + r] = con ? sv_col_dot(lc, v, s, g) : 0.0;       // x = L^-T z
    __syncthreads();
    if (vrow) M.Xs[(size_t)r0*s + tid] = ((red[tid] + red[80 + tid]) + (red[160 + tid] + red[240 + tid])) + (red[320 + tid] + red[400 + tid]);
}

// ---- the top of the tree in one launch: the single pivot of the highest level (i = h: its left neighbour is the root, it has no right one), the root,
// and the pivot's back substitution -- three dependent launches of ~5 us as one workgroup's work, every operand of the three steps requested up front.
// (With no pivot at that level, m <= h, this is the root kernel.)  TREE: as a workgroup of k_sv_cre_tree -- pending updates polled, results published.
template <bool TREE>
__device__ __forceinline__ void sv_top_body(const Work &W, int s, int B, int Pmax, int mmax, int h, const MsBuf &M, double *v, double *w, double *cga, double *x0, double *zz, double *red) {
    const int tid = threadIdx.x, i = h, r0 = 0, lo = 0;
    const LmState *st_ = W.st; const int flags = st_->done | st_->lin_done | st_->step_fail, nf = *W.nfree;
    const bool vrow = tid < s;
    const int part = tid & 3, rq = tid >> 2, g = tid/80, r = tid - 80*g; const bool con = g < 6 && r < s;
    SvFwd F; sv_fwd_load(M, s, i, tid, F);
    double g0r = vrow ? M.G[(size_t)r0*s + tid] : 0.0, g1r = vrow ? M.G2[(size_t)r0*s + tid] : 0.0, idr = vrow ? M.Lid[(size_t)r0*s + tid] : 0.0;
    SvRow lir; SvCol lcr;
    sv_row_load(M.Li + ((size_t)r0*s + (rq < s ? rq : 0))*s, s, part, rq < s, lir);
    sv_col_load(M.Li + (size_t)r0*s*s, s, g, con ? r : 0, con, lcr);
    sv_pin(g0r); sv_pin(g1r); sv_pin(idr); sv_pin(lir); sv_pin(lcr);
    SvPend pd; double pr_[8];                                  // pr_: the root's pending updates -- the pivots 2^l of the levels below h (from the right only)
    if (!TREE) { sv_pending_load(M, s, i, mmax, tid, vrow, pd); sv_pin(pd);
#pragma unroll
        for (int l = 0; l < 8; l++) { const int hp = 1 << l; pr_[l] = (vrow && hp < h && hp < mmax) ? M.Cg[((size_t)hp*2 + 0)*s + tid] : 0.0; sv_pin(pr_[l]); } }
    if (flags) return;
    const int m = ms_uni(sv_nsep(nf, B, Pmax));
    if (m <= 0) return;
    const bool piv = i < m;                                      // (uniform)
    if (TREE) { sv_pending_poll(M, s, i, piv ? h : 0, m, tid, vrow, pd);
        unsigned need = 0;
#pragma unroll
        for (int l = 0; l < 8; l++) { const int hp = 1 << l; pr_[l] = 0.0; if (vrow && hp < h && hp < m) need |= 1u << l; }
        for (int spins = 0; need && spins < SV_SPIN_MAX; spins++) {
#pragma unroll
            for (int l = 0; l < 8; l++) if (need >> l & 1) pr_[l] = sv_ld_co(&M.Cg[((size_t)(1 << l)*2 + 0)*s + tid]);
#pragma unroll
            for (int l = 0; l < 8; l++) if ((need >> l & 1) && pr_[l] == pr_[l]) need &= ~(1u << l);
            if (need) __builtin_amdgcn_s_sleep(1); }
#pragma unroll
        for (int l = 0; l < 8; l++) if (need >> l & 1) pr_[l] = __builtin_inf();
        if (need) atomicAdd(&ts_poll_giveups, 1u); }
    if (piv) {                                                  // forward step of the pivot: the root's update P_a v, z = D^-1 L^-1 v
        if (vrow) v[tid] = sv_pending_sum(pd, i, h, lo, m, r0, F.g0 + F.g1);
        __syncthreads();
        sv_fwd_products(F, v, s, tid, zz, [&](int R, double acc) { if (R < s) cga[R] = acc; });
        __syncthreads();
    }
    // (the operands of the pivot's back substitution are requested here: the registers of its forward step are free, the root's work hides the wait)
    SvCol ca, lc;
    sv_col_load(M.Pp + ((size_t)i*2 + 0)*s*s, s, g, con ? r : 0, con && piv, ca);
    sv_col_load(M.Li + (size_t)i*s*s, s, g, con ? r : 0, con && piv, lc);
    // the root: its pending updates of the levels below h as the root kernel takes them, the pivot's last
    if (vrow) { double t = g0r + g1r;
#pragma unroll
        for (int l = 0; l < 8; l++) { const int hp = 1 << l; if (hp < h && hp < m) t -= pr_[l]; }
        if (piv) t -= cga[tid];
        v[tid] = t; }
    __syncthreads();
    { const double wv = sv_row_dot(lir, v, s, part); if (rq < s && part == 0) w[rq] = wv; }
    __syncthreads();
    if (vrow) v[tid] = w[tid]*idr;                               // z of the root
    __syncthreads();
    if (g < 6) red[g*80 + r] = con ? sv_col_dot(lcr, v, s, g) : 0.0;
    __syncthreads();
    if (vrow) { const double xv = SV_SUM6(red, tid); x0[tid] = xv; if (TREE) sv_st_co(&M.Xs[(size_t)r0*s + tid], xv); else M.Xs[(size_t)r0*s + tid] = xv; }
    if (!piv) return;
    __syncthreads();
    if (g < 6) red[g*80 + r] = con ? sv_col_dot(lc, zz, s, g) : 0.0;     // L^-T z
    __syncthreads();
    const double u0 = vrow ? SV_SUM6(red, tid) : 0.0;
    __syncthreads();
    if (g < 6) red[g*80 + r] = con ? sv_col_dot(ca, x0, s, g) : 0.0;     // P_a^T x_0
    __syncthreads();
    if (vrow) { const double xv = u0 - SV_SUM6(red, tid); if (TREE) sv_st_co(&M.Xs[(size_t)i*s + tid], xv); else M.Xs[(size_t)i*s + tid] = xv; }
}
__global__ __launch_bounds__(SV_CT) void k_sv_cre_top(Work W, Work Ws, int bw, int Pmax, int h, MsBuf M) {
    __shared__ __attribute__((aligned(16))) double v[80], w[80], cga[80], x0[80], zz[80], red[6*80];
    sv_top_body<false>(W, bw, bw/6, Pmax, cr_mmax(W.ring, Pmax, W.ring_g), h, M, v, w, cga, x0, zz, red);
    (void)Ws;
}

// ---- cyclic reduction, level h, backward.  grid pivots, SV_CT threads:  x_i = L^-T z_i - P_a^T x_a - P_c^T x_c
__global__ __launch_bounds__(SV_CT) void k_sv_cre_back(Work W, Work Ws, int bw, int Pmax, int h, int kb, MsBuf M) {
    __shared__ __attribute__((aligned(16))) double zz[80], xa[80], xc[80], red[6*80];
    const int tid = threadIdx.x;
    const int s = bw, B = s/6, mmax = cr_mmax(W.ring, Pmax, W.ring_g);
    const int i = (2*(kb + (int)blockIdx.x) + 1)*h, ia = i - h, ic = i + h;
    const LmState *st_ = W.st; const int flags = st_->done | st_->lin_done | st_->step_fail, nf = *W.nfree;
    const bool vrow = tid < s, cin = ic < mmax;
    double zv = vrow ? M.Z[(size_t)i*s + tid] : 0.0, xav = vrow ? M.Xs[(size_t)ia*s + tid] : 0.0, xcv = (vrow && cin) ? M.Xs[(size_t)ic*s + tid] : 0.0;
    const int g = tid/80, r = tid - 80*g; const bool con = g < 6 && r < s;
    SvBack K; sv_back_load(M, s, i, g, r, con, true, K);
    sv_pin(zv); sv_pin(xav); sv_pin(xcv);
    if (flags) return;
    const int m = ms_uni(sv_nsep(nf, B, Pmax)), lo = 0;
    if (i < lo || i >= m) return;
    const bool has_a = ia >= lo, has_c = ic < m;
    if (vrow) { zz[tid] = zv; xa[tid] = has_a ? xav : 0.0; xc[tid] = has_c ? xcv : 0.0; }
    __syncthreads();
    if (g < 6) red[g*80 + r] = con ? sv_col_dot(K.lc, zz, s, g) : 0.0;
    __syncthreads();
    const double u0 = vrow ? SV_SUM6(red, tid) : 0.0;
    __syncthreads();
    if (g < 6) red[g*80 + r] = con ? (has_a ? sv_col_dot(K.ca, xa, s, g) : 0.0) + (has_c ? sv_col_dot(K.cc, xc, s, g) : 0.0) : 0.0;
    __syncthreads();
    if (vrow) M.Xs[(size_t)i*s + tid] = u0 - SV_SUM6(red, tid);
    (void)Ws;
}

// ---- the whole separator tree in ONE launch: forward steps of every level, the root, back substitution of every level.
// A level of the tree is ~1 us of work behind a launch of ~6 us, twelve levels and the top per application, fifteen to twenty applications per LM
// trial.  Here every pivot has its workgroup for the whole application (grid = labels 1 .. mmax - 1; the workgroup of the top pivot also does the
// root): it requests its matrix operands, then POLLS the vector entries it needs -- the pending updates of the lower levels going up, the
// solution of its neighbours coming down -- until they are no longer the NaN that k_sv_fwd_int left in those slots for this application.  A value
// travels from its producer to a polling consumer on another XCD in ~0.65 us (tools/handover_bench.hip: flag + data 1.2 - 2 us, a dependent
// launch 2.8 us at best), and nothing else is handed over: every other operand was written by an earlier launch.  Producers never publish a
// NaN (a NaN result goes out as +inf: the iteration above sees it in r.z) and polling is bounded, so a broken factor cannot park the device.
// All workgroups are resident at once (at most 127 of them on 256 CUs), and a waiting workgroup holds nothing its producers need.
// Same arithmetic in the same order as k_sv_cre_fwd / _top / _back: the result is bit-identical to the launch-per-level path.
// a pivot below the top: its forward step, then -- once both neighbours are solved -- its back substitution
__device__ __forceinline__ void sv_tree_node_body(const Work &W, int s, int B, int Pmax, int i, int htop, const MsBuf &M, double *v, double *w, double *x0, double *xc, double *red) {
    const int tid = threadIdx.x, lo = 0, r0 = 0;
#ifdef TSBA_SOLVE_STAMPS
    // wall-clock stamps (10 ns ticks, one clock for the whole device) of pivot 1 and of the pivot below the top: W.dbg[0 / 8 ..]
    const int sslot = i == 1 ? 0 : i == htop/2 ? 8 : -1; int sn = 0;
#define TREE_STAMP() do { if (sslot >= 0 && tid == 0 && sn < 8) W.dbg[sslot + sn++] = wall_clock64(); } while (0)
#else
#define TREE_STAMP() do { } while (0)
#endif
    TREE_STAMP();
    const int h = i & -i, ia = i - h, ic = i + h;
    const LmState *st_ = W.st; const int flags = st_->done | st_->lin_done | st_->step_fail, nf = *W.nfree;
    const bool vrow = tid < s;
    const int g = tid/80, r = tid - 80*g; const bool con = g < 6 && r < s;
    SvFwd F; sv_fwd_load(M, s, i, tid, F);
    if (flags) return;
    const int m = ms_uni(sv_nsep(nf, B, Pmax));
    if (i >= m) return;
    const bool has_a = ia >= lo, has_c = ic < m;
    TREE_STAMP();
    // ---- forward step (k_sv_cre_fwd) ...
    SvPend pd; sv_pending_poll(M, s, i, h, m, tid, vrow, pd);
    TREE_STAMP();
    if (vrow) v[tid] = sv_pending_sum(pd, i, h, lo, m, r0, F.g0 + F.g1);
    __syncthreads();
    sv_fwd_products(F, v, s, tid, w, [&](int R, double acc) { sv_st_co(&M.Cg[((size_t)i*2 + (R < s ? 0 : 1))*s + (R < s ? R : R - s)], (R < s ? has_a : has_c) ? acc : 0.0); });     // (w: z)
    TREE_STAMP();
    // ---- ... and its back substitution (k_sv_cre_back): the operands and L^-T z while the levels above work, one product once both neighbours are solved
    SvBack K; sv_back_load(M, s, i, g, r, con, has_c, K);
    __syncthreads();
    if (g < 6) red[g*80 + r] = con ? sv_col_dot(K.lc, w, s, g) : 0.0;
    __syncthreads();
    const double u0 = vrow ? SV_SUM6(red, tid) : 0.0;
    double xav, xcv; sv_poll2(&M.Xs[(size_t)ia*s + tid], vrow && has_a, &M.Xs[(size_t)ic*s + tid], vrow && has_c, xav, xcv);
    TREE_STAMP();
    if (vrow) { x0[tid] = xav; xc[tid] = xcv; }
    __syncthreads();
    if (g < 6) red[g*80 + r] = con ? (has_a ? sv_col_dot(K.ca, x0, s, g) : 0.0) + (has_c ? sv_col_dot(K.cc, xc, s, g) : 0.0) : 0.0;
    __syncthreads();
    if (vrow) sv_st_co(&M.Xs[(size_t)i*s + tid], u0 - SV_SUM6(red, tid));
    TREE_STAMP();
}
__global__ __launch_bounds__(SV_CT) void k_sv_cre_tree(Work W, Work Ws, int bw, int Pmax, int htop, MsBuf M) {
    __shared__ __attribute__((aligned(16))) double v[80], w[80], cga[80], x0[80], xc[80], red[6*80];
    const int s = bw, B = s/6, mmax = cr_mmax(W.ring, Pmax, W.ring_g), i = (int)blockIdx.x + 1;
    if (i == htop) sv_top_body<true>(W, s, B, Pmax, mmax, htop, M, v, w, cga, x0, xc, red);
    else sv_tree_node_body(W, s, B, Pmax, i, htop, M, v, w, x0, xc, red);
    (void)Ws;
}
// ---- the tree and the interiors' back substitution in one launch: workgroup p is the pivot of label p (the top pivot's also the root; workgroups 0 and P - 1
// have no pivot) and then interior p, which polls the solutions of its two separators where k_sv_back_int reads them -- with the records of its first
// chunk and the border rows already requested.  (k_sv_back_int behind its own launch: 20.9 us per application.)
template <int NREG>
__global__ __launch_bounds__(SV_T) void k_sv_tree_back(Work W, int bw, int Pmax, int htop, int lmax, const double *__restrict__ Lrow, const double *__restrict__ Lb, MsBuf M, const double *__restrict__ rdot, double *__restrict__ rz_part) {
    extern __shared__ __attribute__((aligned(16))) double ms_smem[];
    __shared__ __attribute__((aligned(16))) double v[80], w[80], cga[80], x0[80], xc[80], red[6*80];
    const int s = bw, B = s/6, mmax = cr_mmax(W.ring, Pmax, W.ring_g), i = (int)blockIdx.x;
    if (i >= 1 && i < mmax) {
        if (i == htop) sv_top_body<true>(W, s, B, Pmax, mmax, htop, M, v, w, cga, x0, xc, red);
        else sv_tree_node_body(W, s, B, Pmax, i, htop, M, v, w, x0, xc, red);
    }
    __syncthreads();
    sv_back_int_body<NREG, true>(W, bw, Pmax, lmax, Lrow, Lb, M, rdot, rz_part, ms_smem, i);
}

// ---- the back substitution of the factorisation's OWN right-hand side through the same products, in one launch: x_i = L^-T z_i - P_a^T x_a - P_c^T x_c with z_i
// from the pivot's record (k_cre_elim carries the right-hand side along), the root's x from the root launch, every other separator polled as in
// k_sv_cre_tree (k_sv_linv, the launch before, leaves the NaNs in Ws.Sy).  Replaces the six k_cre_back launches of a 5000-keyframe chain (10.5 us each:
// a substitution on one wave behind its launch) by k_sv_linv (which the iterative path runs anyway) + ~15 us.
__global__ __launch_bounds__(SV_CT) void k_cre_back_tree(Work W, Work Ws, int bw, int Pmax, const double *__restrict__ fac, MsBuf M) {
    __shared__ __attribute__((aligned(16))) double zz[80], xa[80], xc[80], red[6*80];
    const int tid = threadIdx.x;
    const int s = bw, B = s/6, i = (int)blockIdx.x + 1, h = i & -i, ia = i - h, ic = i + h;
    const LmState *st_ = W.st; const int flags = st_->done | st_->lin_done | st_->step_fail, nf = *W.nfree;
    const bool vrow = tid < s;
    double zv = vrow ? fac[(size_t)i*cre_rec_doubles(s) + rowoff(s) + SOLVE_LD*B + tid] : 0.0;
    const int g = tid/80, r = tid - 80*g; const bool con = g < 6 && r < s;
    SvBack K; sv_back_load(M, s, i, g, r, con, true, K);
    sv_pin(zv);
    if (flags || nf <= 0) return;
    const int m = ms_uni(sv_nsep(nf, B, Pmax));
    if (i >= m) return;
    const bool has_c = ic < m;
    double *x = Ws.Sy;
    if (vrow) zz[tid] = zv;
    __syncthreads();
    if (g < 6) red[g*80 + r] = con ? sv_col_dot(K.lc, zz, s, g) : 0.0;
    __syncthreads();
    const double u0 = vrow ? SV_SUM6(red, tid) : 0.0;
    double xav, xcv; sv_poll2(&x[(size_t)ia*s + tid], vrow, &x[(size_t)ic*s + tid], vrow && has_c, xav, xcv);
    if (vrow) { xa[tid] = xav; xc[tid] = xcv; }
    __syncthreads();
    if (g < 6) red[g*80 + r] = con ? sv_col_dot(K.ca, xa, s, g) + (has_c ? sv_col_dot(K.cc, xc, s, g) : 0.0) : 0.0;
    __syncthreads();
    if (vrow) sv_st_co(&x[(size_t)i*s + tid], u0 - SV_SUM6(red, tid));
}
