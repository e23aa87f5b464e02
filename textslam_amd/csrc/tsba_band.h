// Streaming band solver for the reduced camera system of a large map (global BA): S dp = -g with S block-banded (tsba_plan.h
// bounds the rows below a pose block that can be non-zero, fill included: bw).  Included by tsba.hip after tsba_solve.h.
//
// The multi-kernel blocked Cholesky of tsba_chol.h spends three dependent launches (diagonal factor / panel / update) per 96
// columns -- ~130 us of launch gaps and load chains per block column, 41 ms per factorisation at 5000 keyframes.  A band of 13
// pose blocks is a 78-row problem: it fits ONE workgroup's LDS with room to spare, so here the whole factorisation is one
// launch of one workgroup that slides a dense window down the band:
//
//   window   = rows [base, base + Wn) of S as a padded packed lower triangle in LDS, Wn = 6 CB + bw, + the right-hand side as the
//              last row (forward substitution rides along), exactly the layout of the small-window solver (tsba_solve.h)
//   chunk    = the small solver's blocked LDL^T step (2 panel waves with look-ahead, 10 MFMA trailing-update waves, one barrier
//              per pose block) for CB pose blocks; rows further than bw below a column are zeros and stay zeros
//   write out the finished columns: L by row block (contiguous for the back substitution), unit-lower diagonal factor
//              and 1/d (LDbuf), forward-substituted right-hand side (Sy)
//   slide    = the trailing (Wn - 6 (CB - 1))^2 / 2 part moves to the top-left corner (the LAST factored block stays as local
//              block 0: its trailing update is applied by the next chunk's first step, as inside a chunk), new rows come
//              straight from S in HBM -- no earlier column reaches them
//   back substitution (k_band_backsub): wave 0 walks the row blocks backwards, right-looking, operands staged through LDS one
//              chunk ahead by the other eleven waves; dp[6a + k] = -x[6 fidx[a] + k].
#pragma once

#define BAND_CK 8                           /* block columns per back-substitution chunk */
#define BAND_BW_MAX 156                     /* widest band the streaming solver takes (26 keyframes): Wn = 6 CB + bw must fit LDS */

// pose blocks per chunk for a band of bw rows (0: the window does not fit)
static int band_chunk_blocks(int bw) {
    for (int cb = 16; cb >= 3; cb--)
        if ((solve_lds_doubles(6*cb + bw) + 64)*sizeof(double) <= 150*1024) return cb;
    return 0;
}
static size_t band_lds_doubles(int bw, int cb) {
    const size_t fac = solve_lds_doubles(6*cb + bw) + 64;
    const size_t bs = 2*(size_t)BAND_CK*((size_t)bw*6 + 32) + 6*64 + 64;
    return fac > bs ? fac : bs;
}

__device__ __forceinline__ double quad_sum(double v) {       // sum over the 4 lanes of a quad (DPP quad_perm), all 4 get it
    int lo = __double2loint(v), hi = __double2hiint(v);
    int lo1 = __builtin_amdgcn_mov_dpp(lo, 0xB1, 0xF, 0xF, true), hi1 = __builtin_amdgcn_mov_dpp(hi, 0xB1, 0xF, 0xF, true);   // [1,0,3,2]
    v += __hiloint2double(hi1, lo1);
    lo = __double2loint(v); hi = __double2hiint(v);
    lo1 = __builtin_amdgcn_mov_dpp(lo, 0x4E, 0xF, 0xF, true); hi1 = __builtin_amdgcn_mov_dpp(hi, 0x4E, 0xF, 0xF, true);       // [2,3,0,1]
    return v + __hiloint2double(hi1, lo1);
}

__global__ __launch_bounds__(SOLVE_THREADS) void k_band_solve(Work W, int bw, int CB, double *Lcol) {
    LmState *st = W.st;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = SOLVE_THREADS/64, NT = NW - SOLVE_PW;
    if (st->done | st->lin_done) return;
    const int nfree_all = *W.nfree, ntot = 6*nfree_all;
    if (st->step_fail || ntot == 0) { for (int k = tid; k < W.N; k += SOLVE_THREADS) W.dp[k] = 0.0; return; }
    const int Wn = 6*CB + bw;
    const size_t ld = (size_t)W.ldS;
    const double *S = W.S;
    double *A = smem;
    double *LD = A + rowoff(Wn + 1) + 16;
    double *scr = LD + SOLVE_LD*(Wn/6);
    const int REC = bw*6;
    if (tid == 0) fail = 0;

    int base = 0, n = min(Wn, ntot);
    // rows [r0, n) of the window from HBM: zeros left of the band, then the band entries; right-hand side columns [r0, n)
    auto load_rows = [&](int r0) {
        for (int r = r0 + wave; r < n; r += NW) { double *row = A + rowoff(r); for (int c = lane; c <= r; c += 64) row[c] = 0.0; }
        __syncthreads();
        const int nr = n - r0, per = bw + 6;                   // the band is block-aligned: row r reaches back to column 6 (r/6) - bw
        for (int e = tid; e < nr*per; e += SOLVE_THREADS) {
            const int rr = e/per, k = e - rr*per, r = r0 + rr, c = r - k;
            if (c >= 0 && c >= 6*(r/6) - bw) A[rowoff(r) + c] = S[(size_t)(base + r)*ld + (base + c)];
        }
        for (int c = r0 + tid; c < n; c += SOLVE_THREADS) A[rowoff(n) + c] = W.g[base + c];
        __syncthreads();
    };
    long long tF = 0, tW = 0, tS = 0, tL = 0, tx = clock64();
    load_rows(0);
    { const long long t_ = clock64(); tL += t_ - tx; tx = t_; }
    bool first = true;
    if (tid == 0 && W.dbg) { W.dbg[16] = bw; W.dbg[17] = CB; W.dbg[18] = ntot; W.dbg[19] = 0; W.dbg[20] = -1; }
    for (;;) {
        const bool last = base + n == ntot;
        const int jend = last ? n/6 : CB, jstart = first ? 0 : 1;
        // ---------------- blocked LDL^T of block columns jstart .. jend-1 of the window (tsba_solve.h, one barrier per block)
        for (int jb = jstart; jb < jend && !fail; jb++) {
            const int j0 = 6*jb, R0 = j0 + 6, p0 = j0 - 6;
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
                // rows below the band of this block column hold zeros and stay zeros: only rows < re (and the rhs row n, which
                // follows them as virtual row re) are touched
                const int re = min(n, R0 + bw);
                auto vrow = [&](int iv) { return iv < re ? iv : n; };
                const int i0 = lane < 6 ? j0 + lane : R0 + wave*SOLVE_PROWS + lane - 6;
                double a[6];
                load_row(vrow(min(i0, re)), a);
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
                    if (bad) { fail = 1; st->step_fail = 1; if (W.dbg) { W.dbg[20] = base; W.dbg[21] = jb; W.dbg[22] = n; W.dbg[23] = __double_as_longlong(d[0]); W.dbg[24] = __double_as_longlong(d[5]); } }
                }
                auto solve_row = [&](int i, double a[6]) {           // x L^T = a (right-looking), stored row = x D^-1
#pragma unroll
                    for (int c = 0; c < 5; c++)
#pragma unroll
                        for (int q = c + 1; q < 6; q++) a[q] = fma(-a[c], l[tri(q - 1) + c], a[q]);
#pragma unroll
                    for (int c = 0; c < 6; c++) a[c] *= id[c];
                    st6(A + rowoff(i) + j0, a);
                };
                if (lane >= 6) {
                    if (i0 <= re) solve_row(vrow(i0), a);
                    for (int i = i0 + SOLVE_PW*SOLVE_PROWS; i <= re; i += SOLVE_PW*SOLVE_PROWS) { load_row(vrow(i), a); solve_row(vrow(i), a); }
                }
            } else if (jb > 0) {
                // trailing update with panel jb-1, whose band ends at row re - 1: rows R0 .. re-1 and the rhs row n (virtual row
                // re), columns R0 .. re-1
                const int re = min(n, j0 + bw);
                const int mr = re - R0 + 1, mc = re - R0;
                const double *ldp = LD + SOLVE_LD*(jb - 1);
                if (mc > 0) {
                    const int ntr = (mr + 15) >> 4, ntc = (mc + 15) >> 4, ntile = tri(ntr);
                    const int lr = lane & 15, lk = lane >> 4;
                    const int k1 = min(4 + lk, 5);
                    const double dk0 = ldp[LD_D + lk], dk1 = lk < 2 ? ldp[LD_D + 4 + lk] : 0.0;
                    // (the tiles of a wave: decode the first, then step -- no square root per tile)
                    int t = wave - SOLVE_PW, ti = 0, tj = 0;
                    if (t < ntile) { ti = tri_row(t); tj = t - tri(ti); }
                    for (; t < ntile; t += NT) {
                        if (tj < ntc) {
                            const int r0v = R0 + 16*ti, c0v = R0 + 16*tj;
                            double a0, a1, b0, b1;
                            if (ti > tj && r0v + 15 < re) {
                                // a tile strictly below the diagonal and entirely inside the band rows: nothing to mask, nothing to
                                // clamp, row offsets by recurrence (rowoff(i + 4) = rowoff(i) + 4 i + 12) -- half the instructions
                                const int arow = rowoff(r0v + lr) + p0, brow = rowoff(c0v + lr) + p0;
                                a0 = -A[arow + lk]; a1 = -A[arow + k1]; b0 = A[brow + lk]*dk0; b1 = A[brow + k1]*dk1;
                                if (lk >= 2) { a1 = 0.0; b1 = 0.0; }
                                const int cv0 = r0v + lk, ccol = c0v + lr;
                                int ci0 = rowoff(cv0) + ccol; const int ci1 = ci0 + 4*cv0 + 12, ci2 = ci1 + 4*cv0 + 28, ci3 = ci2 + 4*cv0 + 44;
                                v4d c = { A[ci0], A[ci1], A[ci2], A[ci3] };
                                c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c, 0, 0, 0);
                                A[ci0] = c[0]; A[ci1] = c[1]; A[ci2] = c[2]; A[ci3] = c[3];
                            } else {
                                // unconditional, index-clamped loads (rows / columns past the edge only feed outputs that are never stored);
                                // only the K padding (k = 6, 7) must be exact zeros
                                const int av = min(r0v + lr, re);
                                const int arow = rowoff(av < re ? av : n) + p0, brow = rowoff(min(c0v + lr, re - 1)) + p0;
                                a0 = -A[arow + lk]; a1 = -A[arow + k1]; b0 = A[brow + lk]*dk0; b1 = A[brow + k1]*dk1;
                                if (lk >= 2) { a1 = 0.0; b1 = 0.0; }
                                const int ccol = c0v + lr;
                                v4d c; int ci[4]; bool ok[4];
#pragma unroll
                                for (int r = 0; r < 4; r++) {
                                    const int cv = r0v + lk + 4*r;                                // virtual row
                                    ok[r] = cv <= re && ccol <= cv && ccol < re;
                                    const int cvc = min(cv, re), crow = cvc < re ? cvc : n;
                                    ci[r] = rowoff(crow) + min(ccol, min(cvc, re - 1));
                                    c[r] = A[ci[r]];
                                }
                                c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c, 0, 0, 0);
                                c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c, 0, 0, 0);
#pragma unroll
                                for (int r = 0; r < 4; r++) if (ok[r]) A[ci[r]] = c[r];
                            }
                        }
                        tj += NT; while (tj > ti) { tj -= ti + 1; ti++; }                         // next tile of this wave
                    }
                }
            }
            __syncthreads();                       // panel jb complete, trailing update with panel jb-1 complete
        }
        { const long long t_ = clock64(); tF += t_ - tx; tx = t_; }
        if (fail && !W.dbg) break;
        // ---------------- finished columns -> HBM: L by block column, unit-lower diagonal factor + 1/d, v = D^-1 L^-1 g
        {
            const int nq = jend - jstart, per = REC + 32;
            for (int e = tid; e < nq*per; e += SOLVE_THREADS) {
                const int qq = e/per, k = e - qq*per, q = jstart + qq, gq = base/6 + q;
                if (k < REC) {                                   // L(r, 6 q + cc) -> row block gr, block b = gr - gq - 1, [b][cc][ri]
                    const int dr = k/6, cc = k - 6*dr, r = 6*q + 6 + dr;
                    if (r < n) { const int b = dr/6, ri = dr - 6*b;
                        Lcol[(size_t)(gq + 1 + b)*REC + b*36 + cc*6 + ri] = A[rowoff(r) + 6*q + cc]; } }
                else { const int u = k - REC;
                    if (u < 15) W.LDbuf[32*(size_t)gq + u] = LD[SOLVE_LD*q + u];
                    else if (u >= 16 && u < 22) W.LDbuf[32*(size_t)gq + u] = LD[SOLVE_LD*q + LD_ID + (u - 16)];
                    else if (u >= 24 && u < 30) W.Sy[6*gq + (u - 24)] = A[rowoff(n) + 6*q + (u - 24)]; }
            }
        }
        { const long long t_ = clock64(); tW += t_ - tx; tx = t_; }
        if (tid == 0 && W.dbg) { W.dbg[19] += 1; W.dbg[32] = tF; W.dbg[33] = tW; W.dbg[34] = tS; W.dbg[35] = tL; }
        if (last || fail) break;
        // ---------------- slide by s rows: (r, c) -> (r - s, c - s) for r, c >= s, the rhs row to the new last row.  Ascending
        // packed order, a batch of 4 per thread through registers: a destination lies below every source not yet read
        {
            const int s = 6*(jend - 1), m = n - s, n_new = min(Wn, ntot - (base + s)), ne = tri(m) + m;
            __syncthreads();
            for (int e0 = 0; e0 < ne; e0 += 4*SOLVE_THREADS) {
                double v[4]; int dst[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int e = e0 + u*SOLVE_THREADS + tid;
                    dst[u] = -1;
                    if (e < ne) { const int r = tri_row(e), c = e - tri(r);
                        v[u] = A[rowoff(r < m ? r + s : n) + c + s]; dst[u] = rowoff(r < m ? r : n_new) + c; }
                }
                __syncthreads();
#pragma unroll
                for (int u = 0; u < 4; u++) if (dst[u] >= 0) A[dst[u]] = v[u];
                __syncthreads();
            }
            if (tid < SOLVE_LD) LD[tid] = LD[SOLVE_LD*(jend - 1) + tid];
            base += s; n = n_new;
            { const long long t_ = clock64(); tS += t_ - tx; tx = t_; }
            // (the rhs row moved first: its new columns and the new rows are loaded below; load_rows starts with a barrier pair)
            load_rows(m);
            { const long long t_ = clock64(); tL += t_ - tx; tx = t_; }
            first = false;
        }
    }
    // (a failed pivot set st->step_fail: k_band_backsub zeroes dp)
}

// back substitution L^T x = v (second launch: the factor kernel's registers are sized for its panel waves), right-looking: as
// soon as x_r is known, every column block j in [r - B, r) receives its term L_rj^T x_r -- only the first of them is needed by
// the next step, so the dependent chain per step is one 6x6 back-solve and one 6-term update, not the whole band.
//   Lrow record of row block r: [b][c][ri] = L(6 r + ri, 6 (r - 1 - b) + c), then l_r (15), 1/d (6, unused here), v_r (6)
//   wave 0: lane = task (b, c) keeps nothing across steps; the running sums u_j live in an LDS ring (6 x 64 row blocks)
//   waves 1..11 stage the next chunk of records from HBM into the other LDS buffer.
#define BAND_RINGB 64
#define BAND_BS_T 512
template <int NU>                           // NU = tasks per lane = ceil(6 B / 64)
__global__ __launch_bounds__(BAND_BS_T) void k_band_backsub(Work W, int bw, const double *Lrow) {
    LmState *st = W.st;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (st->done | st->lin_done) return;
    const int nfree_all = *W.nfree;
    if (st->step_fail || nfree_all == 0) { for (int k = tid; k < W.N; k += BAND_BS_T) W.dp[k] = 0.0; return; }
    const int REC = bw*6, B = bw/6, NTASK = 6*B;
    const int nblk = nfree_all, RECB = REC + 32;
    double *buf0 = smem, *buf1 = smem + (size_t)BAND_CK*RECB, *ring = buf1 + (size_t)BAND_CK*RECB;
    for (int k = tid; k < 6*BAND_RINGB; k += BAND_BS_T) ring[k] = 0.0;
    // row blocks [jlo, jhi) of a chunk -> buf (record q = j - jlo).  A stager thread issues ALL its loads before the first
    // store: one HBM latency per chunk, not one per element
    auto stage = [&](int chunk, double *buf, int t0, int nt) {
        const int jhi = nblk - chunk*BAND_CK, jlo = max(0, jhi - BAND_CK), nq = jhi - jlo;
        const int h = REC >> 1, n2 = nq*h;                                   // the L parts are one contiguous run of double2
        const v2d *src = (const v2d *)(Lrow + (size_t)jlo*REC);
        const int tx = t0;                                                   // extras: 32 per record, one per thread
        double xv = 0.0; int xd = -1;
        if (tx < nq*32) { const int q = tx >> 5, u = tx & 31, j = jlo + q; xd = q*RECB + REC + u;
            xv = u < 24 ? W.LDbuf[32*(size_t)j + u] : (u < 30 ? W.Sy[6*j + (u - 24)] : 0.0); }
        for (int e0 = t0; e0 < n2; e0 += 4*nt) {
            v2d v[4]; int dst[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = e0 + u*nt; dst[u] = -1;
                if (e < n2) { const int q = e/h, k2 = e - q*h; dst[u] = q*RECB + 2*k2;
                    v[u] = (jlo + q - 1 - (2*k2)/36 >= 0) ? src[e] : v2d{0.0, 0.0}; }    // (blocks left of column 0 were never written)
            }
#pragma unroll
            for (int u = 0; u < 4; u++) if (dst[u] >= 0) *(v2d *)(buf + dst[u]) = v[u];
        }
        if (xd >= 0) buf[xd] = xv;
    };
    const int nchunk = (nblk + BAND_CK - 1)/BAND_CK;
    stage(0, buf0, tid, BAND_BS_T);
    __syncthreads();
    // this lane's tasks (b, c): target row block r - 1 - b, component c
    int tb[NU], tc[NU];
#pragma unroll
    for (int u = 0; u < NU; u++) { const int t = lane + 64*u; tb[u] = t < NTASK ? t/6 : -1; tc[u] = t - 6*(t/6); }
    struct Ops { double Lr[NU][6], l[16], vv[6]; };
    auto fetch = [&](const double *rec, Ops &o) {                  // the record data of one step: independent of the running solution
#pragma unroll
        for (int u = 0; u < NU; u++) if (tb[u] >= 0) ld6(rec + 6*(lane + 64*u), o.Lr[u]);
#pragma unroll
        for (int k = 0; k < 8; k++) { const v2d x2 = ((const v2d *)(rec + REC))[k]; o.l[2*k] = x2.x; o.l[2*k + 1] = x2.y; }
        ld6(rec + REC + 24, o.vv);
    };
    auto step = [&](int r, const Ops &o, const double *nrec, Ops &on) {
        // the reads that depend on the previous step first, then the next step's record
        double ur[6], uo[NU]; int slot[NU];
        ld6(ring + 6*(r & (BAND_RINGB - 1)), ur);
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int j = r - 1 - tb[u];
            slot[u] = (tb[u] >= 0 && j >= 0) ? 6*(j & (BAND_RINGB - 1)) + tc[u] : -1;
            uo[u] = slot[u] >= 0 ? ring[slot[u]] : 0.0;
        }
        if (nrec) fetch(nrec, on);
        // x = l^-T (v - u): unit lower l packed (1,0) (2,0) (2,1) ...
        double x[6];
#pragma unroll
        for (int q = 5; q >= 0; q--) { double v = o.vv[q] - ur[q];
#pragma unroll
            for (int k = q + 1; k < 6; k++) v = fma(-o.l[tri(k - 1) + q], x[k], v);
            x[q] = v; }
#pragma unroll
        for (int u = 0; u < NU; u++)
            if (slot[u] >= 0) {
                const double s0 = fma(o.Lr[u][0], x[0], fma(o.Lr[u][1], x[1], o.Lr[u][2]*x[2])), s1 = fma(o.Lr[u][3], x[3], fma(o.Lr[u][4], x[4], o.Lr[u][5]*x[5]));
                ring[slot[u]] = uo[u] + (s0 + s1);
            }
        if (lane < 6) {
            double xv = x[0];
#pragma unroll
            for (int q = 1; q < 6; q++) if (lane == q) xv = x[q];
            ring[6*(r & (BAND_RINGB - 1)) + lane] = 0.0;                   // the slot is reused 64 row blocks further up
            W.Sy[6*r + lane] = xv;
        }
        wave_lds_fence();
    };
    for (int ch = 0; ch < nchunk; ch++) {
        double *buf = (ch & 1) ? buf1 : buf0, *nxt = (ch & 1) ? buf0 : buf1;
        if (wave > 0) { if (ch + 1 < nchunk) stage(ch + 1, nxt, tid - 64, BAND_BS_T - 64); }
        else {
            const int jhi = nblk - ch*BAND_CK, jlo = max(0, jhi - BAND_CK);
            Ops oa, ob;
            int r = jhi - 1;
            fetch(buf + (size_t)(r - jlo)*RECB, oa);
            for (; r - 1 >= jlo; r -= 2) {                                   // two steps per trip: the register sets alternate
                step(r, oa, buf + (size_t)(r - 1 - jlo)*RECB, ob);
                step(r - 1, ob, r - 2 >= jlo ? buf + (size_t)(r - 2 - jlo)*RECB : nullptr, oa);
            }
            if (r >= jlo) step(r, oa, nullptr, ob);
        }
        __syncthreads();
    }
    __threadfence();
    __syncthreads();
    for (int a = tid; a < W.n_kf; a += BAND_BS_T) {
        const int ia = W.fidx[a];
#pragma unroll
        for (int k = 0; k < 6; k++) W.dp[6*a + k] = ia >= 0 ? -W.Sy[6*ia + k] : 0.0;
    }
}
