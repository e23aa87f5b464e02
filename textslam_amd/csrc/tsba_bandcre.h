// Cyclic reduction of the separator system (tsba_bandcr.h), one launch per level.
//
// A pivot's elimination is a PARTIAL factorisation of the arrow matrix
//        [ D_i                      ]      rows of i (s)
//        [ S(a, i)   .              ]      rows of a (s)      "extra rows": they ride below the s x s triangle exactly like the
//        [ S(c, i)   .      .       ]      rows of c (s)      right-hand-side row of k_solve_t, s columns wide
//        [ g_i^T                    ]      rhs row
// over its first s columns -- the blocked LDL^T of k_solve_t (panel waves with look-ahead and an in-register 6x6 factorisation,
// MFMA trailing update, one barrier per pose block) with a row-address function instead of rowoff(): afterwards the extra rows
// hold X = S(., i) L^-T D^-1 and z = D^-1 L^-1 g_i, and the Schur complement onto the neighbours is ONE symmetric product
// E D E^T over the extra rows (16x16 MFMA tiles, K = s):  (a, a) and (c, c) are the updates of D_a, D_c, (c, a) is minus the next
// level's coupling, (g, .) the updates of g_a, g_c.  A remaining block receives two updates per level (from the pivots on its left
// and right) -- the pivots do not add them in place (two writers) but leave them in their own slot (contrib, indexed by the
// PRODUCING pivot: every separator is a pivot exactly once), and a pivot of level H collects its 2 log2(H) pending updates when
// it loads D_i, g_i, in a fixed order (levels ascending, left before right): deterministic, and no update kernel.
// The pivot + update pair of tsba_bandcr.h took 42 + 22 us per level at s = 60.  Back substitution: the factor goes out as the LDS
// image the small-window back substitution reads (packed unit-lower L, table of d, 1/d, inverse 6x6 factors),
// x_i = L^-T (z - X_a^T x_a - X_c^T x_c).
#pragma once

#define CRE_T 768
#ifndef CRE_PW
#define CRE_PW 3                            // panel waves (58 rows each per round; up to s - 6 + 2 s + 1 = 229 rows below a diagonal block at s = 78). Measured at s = 60 (175 rows in the first step), 5000 keyframes, ms per solve: 2 waves 17.04 - 17.3, 3 waves 17.01, 4 waves (always one round) 17.26
#endif
#ifndef TSBA_CRE_KMAX
#define TSBA_CRE_KMAX 4                      // most workgroups per pivot (k_cre_elim: they share the product and the stores)
#endif
#define CRE_BT 512
__host__ __device__ __forceinline__ int cre_stride(int s) { return (s & 3) == 2 ? s : s + 2; }      // doubles; = 2 mod 4: b128 rows of 16 lanes hit 64 different banks
__host__ __device__ __forceinline__ size_t cre_rec_doubles(int s) { return ((size_t)rowoff(s) + (size_t)SOLVE_LD*(s/6) + s + 1) & ~(size_t)1; }     // packed factor | LD table | z
__host__ __device__ __forceinline__ size_t cre_contrib_doubles(int s) { return 2*(size_t)s*s + 2*(size_t)s; }                     // aa | cc | ga | gc
static size_t cre_elim_lds_doubles(int s) { return (size_t)(3*s + 1)*cre_stride(s) + 16 + (size_t)SOLVE_LD*(s/6) + 36*CRE_PW + s + 8; }      // RECTANGULAR rows 0 .. 3 s (see k_cre_elim)
static size_t cre_back_lds_doubles(int s) { return (size_t)rowoff(s + 1) + 16 + (size_t)SOLVE_LD*(s/6) + 4*128 + 2*(size_t)s + 8; }

template <int T, int U, class F, class G>
__device__ __forceinline__ void cre_batched(int n, int tid, F f, G st) {
    for (int e0 = tid; e0 < n; e0 += U*T) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; u++) { const int e = e0 + u*T; v[u] = e < n ? f(e) : 0.0; }
#pragma unroll
        for (int u = 0; u < U; u++) { const int e = e0 + u*T; if (e < n) st(e, v[u]); }
    }
}

// root = 1: the last remaining block (index 0), no neighbours: factor, then x_0 = L^-T z right here
// K workgroups per pivot: all of them run the same factorisation (bit-identical: same instruction sequence on the same data), then
// share the product and the stores -- E D E^T is 36 tiles x 15 fp64 MFMA of 64 cycles at s = 60, 8.6 k cycles of ONE compute unit's four
// matrix pipes, and three quarters of the chip idle next to the 32 pivots of the first level.
// The K workgroups of a pivot overwrite the couplings S(i, a), S(c, i) IN PLACE with X_a, X_c (what the back substitution reads) -- the very blocks every one
// of them loads first.  Started together they are 25 us past their loads when the first of them stores; with another context on the device (TextSLAM extracts
// ORB features while a bundle adjustment runs) one of the K can be dispatched that much later and load a mix of S and X: found in round 5 by running the
// 5000-keyframe solves beside a busy context (1 run in 15 ended with other numbers, some with a failed step).  `gate` [label][K]: workgroup `part` counts its
// completed load phases there (its own word: a plain counter, one writer); before the first store every workgroup makes sure its K - 1 siblings' counters
// have reached its own -- checked by an update wave during the last factorisation step, where it costs nothing; a sibling that is late is waited for
// (not for ever, and not once the step has failed anyway).  The counter is the factorisation's ordinal `epoch` (host), so a label that sits a pass out, or
// a workgroup that left early, leaves nothing behind.
__global__ __launch_bounds__(CRE_T) void k_cre_elim(Work W, Work Ws, int bw, int Pmax, int h, int root, int K, int kb, double *contrib, double *fac, int *gate, int epoch) {
    LmState *st = W.st;
    if (st->done || st->step_fail || st->lin_done) return;
    const CrRange rg = cr_range(W, bw, Pmax);
    const int m = rg.m, lo = rg.lo, r0 = rg.r0, s = bw, B = s/6, mmax = cr_mmax(W.ring, Pmax, W.ring_g);
    int i, a, c;
    if (root) { if (blockIdx.x > 0 || m <= 0) return; i = r0; a = -1; c = -1; }
    else { i = (2*(kb + (int)blockIdx.x/K) + 1)*h;               // (kb: the first pivot the host launches -- labels of a ring with a tail do not start at 0)
        if (i < lo || i >= m || (W.ring && i == m - 1)) return;  // (ring: label m - 1 is the ghost of the root, never a pivot)
        a = i - h >= lo ? i - h : -1; c = i + h < m ? i + h : -1; }
    const int part = root ? 0 : (int)blockIdx.x % K;
    int *gate_i = gate + (size_t)i*TSBA_CRE_KMAX;
    const int my_gen = epoch;
    __shared__ int gate_ok;
    const int H = root ? (1 << 30) : h;                          // pending updates come from the pivots i -+ h', h' < H
    const int na = a >= 0 ? s : 0, nc = c >= 0 ? s : 0, ne = na + nc + 1, n = s + ne - 1;     // rows 0 .. n, row n = g_i
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = CRE_T/64, NT = NW - CRE_PW;
    // every row of the arrow matrix has the same stride (the upper triangle of D_i is dead space): an element address is one
    // multiply-add -- the update waves are bound by instruction issue, and with the packed triangle of k_solve_t two thirds of a
    // tile's instructions were index arithmetic (rowoff, clamps, masks); whole tiles below the diagonal now take an unmasked path
    const int sst = cre_stride(s), xbase = s*sst;
    double *A = smem, *LD = A + (size_t)(3*s + 1)*sst + 16, *scr = LD + SOLVE_LD*B, *dflat = scr + 36*CRE_PW;
    double *S = Ws.S, *g = Ws.g;
    const size_t csz = cre_contrib_doubles(s), ss = (size_t)s*s;
    const float inv_s = 1.0f/(float)s;
    auto rowof = [&](int e) { return (int)(((float)e + 0.5f)*inv_s); };
#ifdef TSBA_SOLVE_STAMPS
    long long q0_ = clock64(), q1_ = 0, q2_ = 0, q3_ = 0, qp_ = 0, qt_ = 0;
#define CRE_STAMP(v) v = clock64()
#else
#define CRE_STAMP(v) do { } while (0)
#endif
    // ---- load: D_i and g_i minus their pending updates; the couplings as rows of a / rows of c
    {
        const double *Bii = cr_blk(S, s, mmax, i, i);
        // pending updates of block `blk`: from the pivots blk - h' (their cc / gc slots, offL) and blk + h' (aa / ga, offR), h' < H
        auto pending_of = [&](int blk, bool left, bool right, size_t offL, size_t offR, size_t idx, double v) {
            double p[16];
#pragma unroll
            for (int l = 0; l < 8; l++) {
                const int hp = 1 << l; const bool on = hp < H && hp < m - lo;
                // (a producer is a pivot of level hp -- the lowest set bit of its label --: not the root, not the ghost; the ghost of a ring with a long
                // tail is hp = 2 G away from a tail pivot of level G that is no neighbour of it)
                const int pl = blk - hp, pr = blk + hp;
                p[2*l] = (on && left && pl >= lo && pl != r0 && (pl & (2*hp - 1)) == hp) ? contrib[(size_t)pl*csz + offL + idx] : 0.0;
                p[2*l + 1] = (on && right && pr < m - (W.ring ? 1 : 0) && pr != r0 && (pr & (2*hp - 1)) == hp) ? contrib[(size_t)pr*csz + offR + idx] : 0.0;
            }
#pragma unroll
            for (int l = 0; l < 16; l++) v -= p[l];
            return v;
        };
        auto pending = [&](size_t offL, size_t offR, size_t idx, double v) { return pending_of(i, true, true, offL, offR, idx, v); };
        if (root == 2) {
            // ring: blocks 0 and m - 1 are the same unknowns (m - 1 is the ghost of the first poses behind the last one): the root block is
            // D_0 + D_{m-1} + S(m-1, 0) + S(m-1, 0)^T with both blocks' pending updates, the gradient the sum of both
            const double *BiiP = cr_blk(S, s, mmax, m - 1, m - 1), *Bp0 = cr_blk(S, s, mmax, m - 1, r0);
            for (int e = tid; e < tri(s); e += CRE_T) {
                const int r = tri_row(e), q = e - tri(r); const size_t idx = (size_t)r*s + q;
                const double v0 = pending_of(r0, true, true, ss, 0, idx, Bii[idx]), v1 = pending_of(m - 1, true, false, ss, 0, idx, BiiP[idx]);
                A[r*sst + q] = (v0 + v1) + (Bp0[idx] + Bp0[(size_t)q*s + r]);
            }
            for (int q = tid; q < s; q += CRE_T)
                A[n*sst + q] = pending_of(r0, true, true, 2*ss + s, 2*ss, (size_t)q, g[(size_t)r0*s + q]) + pending_of(m - 1, true, false, 2*ss + s, 2*ss, (size_t)q, g[(size_t)(m - 1)*s + q]);
        } else {
        for (int e = tid; e < tri(s); e += CRE_T) {
            const int r = tri_row(e), q = e - tri(r); const size_t idx = (size_t)r*s + q;
            A[r*sst + q] = pending(ss, 0, idx, Bii[idx]);
        }
        for (int q = tid; q < s; q += CRE_T) A[n*sst + q] = pending(2*ss + s, 2*ss, (size_t)q, g[(size_t)i*s + q]);
        }
        if (a >= 0) { const double *Bia = cr_blk(S, s, mmax, i, a);          // S(i, a)(r, t) -> row t of a, column r
            cre_batched<CRE_T, 8>(s*s, tid, [&](int e) { return Bia[e]; }, [&](int e, double v) { const int r = rowof(e), t = e - r*s; A[xbase + t*sst + r] = v; }); }
        if (c >= 0) { const double *Bci = cr_blk(S, s, mmax, c, i);          // S(c, i)(j, r) -> row j of c, column r
            cre_batched<CRE_T, 8>(s*s, tid, [&](int e) { return Bci[e]; }, [&](int e, double v) { const int j = rowof(e), r = e - j*s; A[xbase + (na + j)*sst + r] = v; }); }
    }
    if (tid == 0) { fail = 0; gate_ok = K > 1 ? 0 : 1; }
    __syncthreads();
    if (K > 1 && tid == CRE_T - 64) __hip_atomic_store(gate_i + part, my_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // this workgroup has loaded (the barrier: every thread's loads have returned)
    CRE_STAMP(q1_);
    for (int jb = 0; jb < B && !fail; jb++) {
        const int j0 = 6*jb, R0 = j0 + 6, p0 = j0 - 6;
#ifdef TSBA_SOLVE_STAMPS
        const long long qs_ = clock64();
#endif
        if (wave < CRE_PW) {
            double Lk[36], dprev[6];
            if (jb > 0) {
                ld6(LD + SOLVE_LD*(jb - 1) + LD_D, dprev);
#pragma unroll
                for (int cc = 0; cc < 6; cc++) ld6(A + (j0 + cc)*sst + p0, Lk + 6*cc);
            }
            auto load_row = [&](int ir, double av[6]) {          // row ir of block column jb with panel jb-1 applied
                const double *row = A + ir*sst;
                ld6(row + j0, av);
                if (jb > 0) {
                    double y[6];
                    ld6(row + p0, y);
#pragma unroll
                    for (int k = 0; k < 6; k++) y[k] *= dprev[k];
#pragma unroll
                    for (int cc = 0; cc < 6; cc++) {
                        double v0 = y[0]*Lk[cc*6], v1 = y[1]*Lk[cc*6 + 1];
                        v0 = fma(y[2], Lk[cc*6 + 2], v0); v1 = fma(y[3], Lk[cc*6 + 3], v1);
                        v0 = fma(y[4], Lk[cc*6 + 4], v0); v1 = fma(y[5], Lk[cc*6 + 5], v1);
                        av[cc] -= v0 + v1;
                    }
                }
            };
            const int i0 = lane < 6 ? j0 + lane : R0 + wave*SOLVE_PROWS + lane - 6;
            double av[6];
            load_row(min(i0, n), av);
            if (lane < 6) st6(scr + wave*36 + lane*6, av);
            wave_lds_fence();
            double sd[21], l[15], d[6], id[6]; bool bad = false;
            {
                double t[36];
#pragma unroll
                for (int r = 0; r < 6; r++) ld6(scr + wave*36 + r*6, t + 6*r);
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int q = 0; q <= r; q++) sd[tri(r) + q] = t[6*r + q];
            }
            ldl6(sd, l, d, id, bad);
            if (wave == 0 && lane == 0) {
                double *o = LD + SOLVE_LD*jb;
#pragma unroll
                for (int k = 0; k < 15; k++) o[k] = l[k];
                st6(o + LD_D, d); st6(o + LD_ID, id);
                if (bad) { fail = 1; st->step_fail = 1; }
            }
            auto solve_row = [&](int ir, double av[6]) {         // x L^T = a, stored row = x D^-1
#pragma unroll
                for (int cc = 0; cc < 5; cc++)
#pragma unroll
                    for (int q = cc + 1; q < 6; q++) av[q] = fma(-av[cc], l[tri(q - 1) + cc], av[q]);
#pragma unroll
                for (int cc = 0; cc < 6; cc++) av[cc] *= id[cc];
                st6(A + ir*sst + j0, av);
            };
            if (lane >= 6) {
                if (i0 <= n) solve_row(i0, av);                  // (further rounds when more rows lie below the block than CRE_PW x 58 lanes hold)
                for (int ir = i0 + CRE_PW*SOLVE_PROWS; ir <= n; ir += CRE_PW*SOLVE_PROWS) { load_row(ir, av); solve_row(ir, av); }
            }
            if (wave == 1 && jb == B - 1) {                      // inverse factor of the last block (the others: last update wave)
                double mi[15];
                inv_unit_lower6(l, mi);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 15; k++) LD[SOLVE_LD*jb + LD_M + k] = mi[k];
                }
            }
        } else if (jb > 0) {
            // trailing update with panel jb-1: rows R0 .. n, columns R0 .. s-1
            const int mr = n - R0 + 1, mc = s - R0;
            const double *ldp = LD + SOLVE_LD*(jb - 1);
            if (mc > 0) {
                const int ntr = (mr + 15) >> 4, ntc = (mc + 15) >> 4, ntt = tri(ntc), ntile = ntt + (ntr - ntc)*ntc;
                const int lr = lane & 15, lk = lane >> 4;
                const int k1 = min(4 + lk, 5);
                const double dk0 = ldp[LD_D + lk], dk1 = lk < 2 ? ldp[LD_D + 4 + lk] : 0.0;
                const int q16 = 65536/ntc + 1;                    // u / ntc for the few tile indices of a step, scalar unit only
                for (int t = wave - CRE_PW; t < ntile; t += NT) {
                    int ti, tj;
                    if (t < ntt) { ti = (t >= 1) + (t >= 3) + (t >= 6) + (t >= 10) + (t >= 15); tj = t - tri(ti); }
                    else { const int u = t - ntt, q = (u*q16) >> 16; ti = ntc + q; tj = u - q*ntc; }
                    const int rb = R0 + 16*ti, cb0 = R0 + 16*tj;
                    const bool full = rb + 15 <= n && cb0 + 15 < s && cb0 + 15 <= rb;          // (wave-uniform)
                    const double *pa = A + min(rb + lr, n)*sst + p0, *pb = A + min(cb0 + lr, s - 1)*sst + p0;
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
                            ok[r] = crow <= n && ccol <= crow && ccol < s;
                            ci[r] = min(crow, n)*sst + min(ccol, s - 1);
                            cv[r] = A[ci[r]];
                        }
                        cv = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, cv, 0, 0, 0);
                        cv = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, cv, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; r++) if (ok[r]) A[ci[r]] = cv[r];
                    }
                }
            }
            if (K > 1 && jb == B - 1 && tid == CRE_T - 64) {      // have the siblings loaded?  (normally long ago: one round trip beside this step's tiles)
                bool ok = true;
                for (int p2 = 0; p2 < K; p2++) if (p2 != part) ok = ok && __hip_atomic_load(gate_i + p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - my_gen >= 0;
                gate_ok = ok ? 1 : 0;
            }
            if (wave == NW - 1) {                                // inverse of the unit-lower factor of block jb-1
                double l[15], mi[15];
#pragma unroll
                for (int k = 0; k < 15; k++) l[k] = ldp[k];
                inv_unit_lower6(l, mi);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 15; k++) LD[SOLVE_LD*(jb - 1) + LD_M + k] = mi[k];
                }
            }
        }
#ifdef TSBA_SOLVE_STAMPS
        if (wave == 0) qp_ += clock64() - qs_; else if (wave == CRE_PW) qt_ += clock64() - qs_;
#endif
        __syncthreads();
    }
    CRE_STAMP(q2_);
    if (fail) return;
    if (root) {                                                  // packed rows (what solve_backsub_wave reads) behind the rhs row; x_0 right here
        double *Pk = A + (size_t)(s + 1)*sst;
        for (int e = tid; e < tri(s); e += CRE_T) { const int r = tri_row(e), q = e - tri(r); Pk[rowoff(r) + q] = A[r*sst + q]; }
        for (int k = tid; k < s; k += CRE_T) Pk[rowoff(s) + k] = A[n*sst + k];
        {   // the root's factor too (packed rows | LD table): the multi-right-hand-side solve phase (tsba_bandms.h) applies it to other vectors
            double *recr = fac + (size_t)r0*cre_rec_doubles(s);
            for (int e = tid; e < tri(s); e += CRE_T) { const int r = tri_row(e), q = e - tri(r); recr[rowoff(r) + q] = A[r*sst + q]; }
            for (int k = tid; k < SOLVE_LD*B; k += CRE_T) recr[rowoff(s) + k] = LD[k]; }
        __syncthreads();
        if (wave == 0) {
            solve_backsub_wave(Pk, LD, s, B, lane);
            wave_lds_fence();
            for (int k = lane; k < s; k += 64) { Ws.Sy[(size_t)r0*s + k] = Pk[rowoff(s) + k]; if (root == 2) Ws.Sy[(size_t)(m - 1)*s + k] = Pk[rowoff(s) + k]; }
        }
        return;
    }
    if (!gate_ok) {                                              // (uniform) a sibling had not loaded yet: wait for it -- bounded, counted, and the step fails if it never comes
        if (tid == CRE_T - 64) {
            bool ok = false;
            for (int spins = 0; !ok && spins < (1 << 16); spins++) { ok = true;
                for (int p2 = 0; p2 < K; p2++) if (p2 != part) ok = ok && __hip_atomic_load(gate_i + p2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - my_gen >= 0;
                if (!ok && __hip_atomic_load(&st->step_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;      // (the step has failed elsewhere: a sibling may have left at once)
                if (!ok) __builtin_amdgcn_s_sleep(8); }
            if (!ok && !__hip_atomic_load(&st->step_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { atomicAdd(&ts_poll_giveups, 1u); st->step_fail = 1; }
        }
        __syncthreads();
    }
    // ---- the factor for the back substitution (packed rows), X_a / X_c over the couplings, z over g
    double *rec = fac + (size_t)i*cre_rec_doubles(s);
    const int pk = rowoff(s);
    for (int e = tid + part*CRE_T; e < tri(s); e += K*CRE_T) { const int r = tri_row(e), q = e - tri(r); rec[rowoff(r) + q] = A[r*sst + q]; }
    if (part == 0) for (int k = tid; k < SOLVE_LD*B; k += CRE_T) rec[pk + k] = LD[k];
    for (int k = tid; k < s; k += CRE_T) { if (part == 0) rec[pk + SOLVE_LD*B + k] = A[n*sst + k]; dflat[k] = LD[SOLVE_LD*(k/6) + LD_D + k % 6]; }
    if (tid == 0) dflat[s] = 0.0;
    if (a >= 0) { double *Bia = cr_blk(S, s, mmax, i, a);
        for (int e = tid + part*CRE_T; e < s*s; e += K*CRE_T) { const int t = rowof(e), r = e - t*s; Bia[e] = A[xbase + t*sst + r]; } }
    if (c >= 0) { double *Bci = cr_blk(S, s, mmax, c, i);
        for (int e = tid + part*CRE_T; e < s*s; e += K*CRE_T) { const int j = rowof(e), r = e - j*s; Bci[e] = A[xbase + (na + j)*sst + r]; } }
    __syncthreads();
    CRE_STAMP(q3_);
    // ---- E D E^T over the extra rows (lower triangle of 16x16 tiles)
    {
        double *cb = contrib + (size_t)i*csz, *Bca = (a >= 0 && c >= 0) ? cr_blk(S, s, mmax, c, a) : nullptr;
        const int ntr = (ne + 15) >> 4, ntile = tri(ntr), lr = lane & 15, lk = lane >> 4;
        for (int t = part*NW + wave; t < ntile; t += K*NW) {
            const int ti = tri_row(t), tj = t - tri(ti);
            const double *pa = A + xbase + (size_t)min(16*ti + lr, ne - 1)*sst, *pb = A + xbase + (size_t)min(16*tj + lr, ne - 1)*sst;
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            for (int k0 = 0; k0 < s; k0 += 20) {                  // operands of five K steps in flight
                double av[5], bv[5];
#pragma unroll
                for (int u = 0; u < 5; u++) { const int k = k0 + 4*u + lk, kk = min(k, s - 1); av[u] = pa[kk]; bv[u] = pb[kk]*dflat[min(k, s)]; }     // dflat[s] = 0: the K padding
#pragma unroll
                for (int u = 0; u < 5; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv[u], acc, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int R = 16*ti + lk + 4*r, Q = 16*tj + lr;
                if (R >= ne || Q > R || Q >= ne - 1) continue;
                if (R == ne - 1) { if (Q < na) cb[2*ss + Q] = acc[r]; else cb[2*ss + s + (Q - na)] = acc[r]; }
                else if (R < na) cb[(size_t)R*s + Q] = acc[r];
                else if (Q < na) Bca[(size_t)(R - na)*s + Q] = -acc[r];
                else cb[ss + (size_t)(R - na)*s + (Q - na)] = acc[r];
            }
        }
    }
#ifdef TSBA_SOLVE_STAMPS
    if (h == 1 && blockIdx.x == (unsigned)K && lane == 0 && (wave == 0 || wave == CRE_PW)) {      // load, factor loop, stores, product; panel / update waves before the barrier
        if (wave == 0) { W.dbg[48] = q1_ - q0_; W.dbg[49] = q2_ - q1_; W.dbg[50] = q3_ - q2_; W.dbg[51] = clock64() - q3_; W.dbg[52] = qp_; }
        else W.dbg[53] = qt_;
    }
#endif
}

// x_i = L^-T (z_i - X_a^T x_a - X_c^T x_c) -> Ws.Sy
// Every address is known from the launch arguments (the pool is laid out for the worst case mmax): all loads are issued before the solver
// state and the number of separators are looked at -- one global round trip instead of two (-1.4 us per level).
__global__ __launch_bounds__(CRE_BT) void k_cre_back(Work W, Work Ws, int bw, int Pmax, int h, int kb, const double *fac) {
    const LmState *st = W.st;
    const int s = bw, B = s/6, mmax = cr_mmax(W.ring, Pmax, W.ring_g), tid = threadIdx.x, lane = tid & 63;
    const int i = (2*(kb + (int)blockIdx.x) + 1)*h;
    if (i >= mmax) return;
    const int a = i - h, cmx = i + h < mmax ? i + h : -1;        // (c exists if cmx < m: decided below)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int xbase = rowoff(s);
    double *A = smem, *LD = A + rowoff(s + 1) + 16, *part = LD + SOLVE_LD*B, *xs = part + 4*128;
    const double *S = Ws.S; double *x = Ws.Sy;
    const double *rec = fac + (size_t)i*cre_rec_doubles(s);
    const double *Xa = cr_blk(S, s, mmax, i, a), *Xc = cmx >= 0 ? cr_blk(S, s, mmax, cmx, i) : Xa;
    const int col = tid & 127, grp = tid >> 7;                   // a thread: column `col` of [X_a ; X_c], rows grp, grp + 4, ...
    const int done = st->done | st->lin_done, sfail = st->step_fail, nb = *W.nfree;
    const double xin = tid < s ? x[(size_t)a*s + tid] : (tid < 2*s && cmx >= 0 ? x[(size_t)cmx*s + tid - s] : 0.0);      // (2 s <= 156 < CRE_BT)
    const double zc = (grp == 0 && col < s) ? rec[xbase + SOLVE_LD*B + col] : 0.0;
    auto xrow = [&](int r) { return r < s ? Xa + (size_t)r*s : Xc + (size_t)(r - s)*s; };
    constexpr int UB = 10;
    const int nxm = cmx >= 0 ? 2*s : s;
    double xv[UB];
#pragma unroll
    for (int u = 0; u < UB; u++) { const int r = grp + 4*u; xv[u] = (col < s && r < nxm) ? xrow(r)[col] : 0.0; }
    cre_batched<CRE_BT, 8>(xbase, tid, [&](int e) { return rec[e]; }, [&](int e, double v) { A[e] = v; });
    for (int k = tid; k < SOLVE_LD*B; k += CRE_BT) LD[k] = rec[xbase + k];
    if (done || sfail) return;
    if (nb <= 0) return;
    const CrRange rg = cr_range(W, bw, Pmax);
    const int m = rg.m;
    if (i < rg.lo || i >= m || (W.ring && i == m - 1)) return;   // (ring: label m - 1 is the ghost of the root, solved with it)
    const bool has_a = a >= rg.lo;                               // (the first separator of a tail has no left neighbour)
    const int nx = (cmx >= 0 && cmx < m) ? 2*s : s;              // rows of [X_a ; X_c]
    if (tid < 2*s) xs[tid] = (tid < s && !has_a) ? 0.0 : xin;
#pragma unroll
    for (int u = 0; u < UB; u++) if (grp + 4*u < s && !has_a) xv[u] = 0.0;
    __syncthreads();
    double acc = 0.0;
#pragma unroll
    for (int u = 0; u < UB; u++) { const int r = grp + 4*u; if (r < nx) acc = fma(xv[u], xs[r], acc); }
    for (int r0 = grp + 4*UB; r0 < nx; r0 += 4*UB) {
#pragma unroll
        for (int u = 0; u < UB; u++) { const int r = r0 + 4*u; xv[u] = (col < s && r < nx && (r >= s || has_a)) ? xrow(r)[col] : 0.0; }
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
    for (int k = lane; k < s; k += 64) x[(size_t)i*s + k] = A[xbase + k];
}

// (All back-substitution levels in ONE launch -- a workgroup per separator, waiting on its neighbours' flags with agent-scope release /
// acquire -- was built and measured in round 2: 91 us against 7 x 11.2 us for the launches per level.  A hop of the dependency tree
// through flags costs more than a kernel boundary: the release writes back the XCD's L2, the acquire invalidates it, and the poll
// adds its own round trips.  The experiment is in the history of this repository: tools/experiments/cre_back_all.h, removed in round 3.)
