// Partitioned streaming band solver (substructuring) for the reduced camera system of a large map.  Included by tsba.hip after
// tsba_band.h.
//
// The single-workgroup streaming solver (tsba_band.h) walks the band's pose blocks one after the other: ~2 us per block, 10 ms at
// 5000 keyframes -- 75 % of an LM iteration, and inherently sequential.  A band matrix falls apart once B = bw / 6 consecutive pose
// blocks are set aside as a SEPARATOR: the interiors on either side do not couple.  With P - 1 separators:
//
//   k_bandp_factor   P workgroups, one per interior [a_p, b_p): the streaming factorisation of tsba_band.h on its own rows.  The right
//                    separator's rows are simply the rows that follow the interior in the band (they stay behind as trailing rows);
//                    the LEFT separator's rows ride along below the window like the right-hand-side row does ("border" rows: zero
//                    at first except for the coupling to the first B blocks, then filling in along the whole interior).  What is
//                    left in the window at the end -- [right separator; left separator] x the same, plus the reduced rhs -- is this
//                    interior's Schur complement contribution T_p.
//   k_bandp_sep      assembles the separator system (block tridiagonal, blocks of 6 B: D_s = T_s.RR + T_{s+1}.LL, off-diagonal from
//                    T_{s+1}.LR) -- a band matrix again, solved by k_band_solve / k_band_backsub on a second Work.
//   k_bandp_backsub  P workgroups: right-looking back substitution of each interior with both separator solutions known
//                    (the right one enters as ordinary steps with given x, the left one through the stored border panel).
//   k_bandp_dp       dp[6a + k] = -x[6 fidx[a] + k].
// Every workgroup derives the same partition table from the number of free poses on the device (the host does not know it).
#pragma once

#define BANDP_MAXP 256
#ifndef BANDP_T
#define BANDP_T SOLVE_THREADS               // threads of k_bandp_factor (round 6: 512 = two waves per SIMD = 256 registers was tried against the 248 bytes per lane this kernel spills at 768 / 168 registers)
#endif
#ifndef BANDP_PW
#define BANDP_PW 2                          // panel waves of k_bandp_factor
#endif

struct BandpPart { int P, a, b, has_left, has_right, G, Pt, lblL; };      // ring maps: G interiors in the loop, Pt in the tail, label of the left separator (right = + 1)
#define RING_OFF 128                        // ring with a tail: the loop's first separator has label RING_OFF, the tail's separators count down from it
// interiors of q = (nb - (P - 1) B) / P blocks -- the first `rem` of them one more: the launch lasts as long as its longest interior, and a
// last interior that took the whole remainder was 85 blocks against 68 at 5000 keyframes / P = 64 --, separators of B blocks between
// them; P shrinks until an interior holds at least 2 B + 2 blocks
__device__ __host__ __forceinline__ BandpPart bandp_part(int nb, int B, int Pmax, int p) {
    BandpPart r;
    int P = Pmax;
    while (P > 1 && (nb - (P - 1)*B)/P < 2*B + 2) P--;
    r.P = P; r.G = 0; r.Pt = 0; r.lblL = p - 1;
    const int tot = nb - (P - 1)*B, q = tot/P, rem = tot - q*P;
    r.a = p*(q + B) + (p < rem ? p : rem); r.b = r.a + q + (p < rem ? 1 : 0);
    if (p == P - 1) r.b = nb;
    r.has_left = p > 0; r.has_right = p < P - 1;
    return r;
}
// Ring maps (one loop closure, tsba_plan.h).  nf free poses in keyframe order, the loop starts at free row row0 (0: the loop is the whole
// trajectory), B ghost blocks follow the last pose:   [tail: interior, sep, ..., interior][S = first B rows of the loop][interior, sep, ...,
// interior][ghost of S].   G interiors in the loop -- a power of two: the separator tree must end with S and its ghost, which are the
// same unknowns --, Pt < RING_OFF in the tail (its separators are eliminated at the levels of their labels like any other; the ghost is
// never a pivot, S never an odd multiple of a level's stride below RING_OFF).
// Separator labels: S = off, the loop's separators off + 1 .. off + G - 1, the ghost off + G, the tail's off - 1, off - 2, ... towards the
// start (off = RING_OFF with a tail, 0 without): interior p has the separators off - Pt + p and off - Pt + p + 1 on its sides.
__device__ __host__ __forceinline__ BandpPart bandp_part_ring(int nf, int row0, int B, int Pmax, int Gmax, int p) {
    BandpPart r;
    const int na = nf - row0;                                    // S + the loop
    int G = Gmax; while (G > 2 && (na - G*B)/G < 2*B + 2) G >>= 1;
    int Pt = 0;
    if (row0 > 0) { Pt = Pmax - Gmax < RING_OFF - 1 ? Pmax - Gmax : RING_OFF - 1; if (Pt < 1) Pt = 1; while (Pt > 1 && (row0 - (Pt - 1)*B)/Pt < 2*B + 2) Pt--; }
    r.G = G; r.Pt = Pt; r.P = G + Pt; r.lblL = (Pt > 0 ? RING_OFF : 0) - Pt + p;
    if (p < Pt) {
        const int tot = row0 - (Pt - 1)*B, q = tot/Pt, rem = tot - q*Pt;
        r.a = p*(q + B) + (p < rem ? p : rem); r.b = r.a + q + (p < rem ? 1 : 0);
        if (p == Pt - 1) r.b = row0;
        r.has_left = p > 0; r.has_right = 1;
    } else {
        const int pp = p - Pt, tot = na - G*B, q = tot/G, rem = tot - q*G;
        r.a = row0 + B + pp*(q + B) + (pp < rem ? pp : rem); r.b = r.a + q + (pp < rem ? 1 : 0);
        if (pp == G - 1) r.b = nf;
        r.has_left = 1; r.has_right = 1;
    }
    return r;
}
// number of pose blocks the band solvers walk (ring: the ghost separator behind the last free pose), separators of the system, pool size
__device__ __forceinline__ int bandp_nb(const Work &W, int B) { const int nf = *W.nfree; return nf > 0 && W.ring ? nf + B : nf; }
__device__ __forceinline__ BandpPart bandp_part_w(const Work &W, int B, int Pmax, int p) {       // the partition every kernel of a launch derives
    const int nf = *W.nfree;
    return W.ring ? bandp_part_ring(nf, W.nfree[1], B, Pmax, W.ring_g, p) : bandp_part(nf, B, Pmax, p);
}
// the same out of line, for k_bandp_factor: the kernel sits at its register cap, and the partition arithmetic inlined into it moved the
// allocation of the step loop (+5 us per launch).  Returns (a, b, has_left | has_right << 1, P).
__device__ __noinline__ int4 bandp_part_call(const int *nfree, int ring, int ring_g, int B, int Pmax, int p) {
    const int nf = nfree[0];
    const BandpPart r = ring ? bandp_part_ring(nf, nfree[1], B, Pmax, ring_g, p) : bandp_part(nf, B, Pmax, p);
    return make_int4(r.a, r.b, r.has_left | (r.has_right << 1), r.P);
}
// separator labels run below this bound (pool, slots and solution are indexed by label)
__device__ __host__ __forceinline__ int cr_mmax(int ring, int Pmax, int Gmax) { return ring ? (Pmax > Gmax ? RING_OFF : 0) + Gmax + 1 : Pmax - 1; }
static size_t bandp_lds_doubles(int bw, int cb) {               // window + border rows + rhs row, LD table, scratch
    const int rows = 6*cb + 2*bw;
    return (size_t)rowoff(rows + 2) + 16 + (size_t)SOLVE_LD*((6*cb + bw)/6) + 36*BANDP_PW + 8 + 64;
}
static int bandp_chunk_blocks(int bw) {
    for (int cb = 16; cb >= 4; cb--) if (bandp_lds_doubles(bw, cb)*sizeof(double) <= 152*1024) return cb;
    return 0;
}

// ---- the cold phases of k_bandp_factor as functions of their own (NOT inlined): the kernel sits at its 168-register cap, and whatever these
// phases keep live competes with the step loop -- inlined, the compiler spilled loop invariants and reloaded them from scratch inside
// the copy loops (a scratch reload is a memory round trip).
typedef __attribute__((address_space(3))) double lds_f64;
// workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every global store in flight (vmcnt(0)) -- the write-out
// of a chunk (130 KB per workgroup) would drain before the slide could start
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// rows [r0, n) of the window from HBM (band storage); for the border / rhs rows the columns [r0, n).  first: also the border-border
// block and the couplings to the left separator.  Row-wise: a wave takes whole rows (row addresses on the scalar unit, no division
// per element) and keeps the loads of LOAD_U rows in flight -- the element-wise loop waited for every load before it issued the next
// (5 - 6 dependent HBM round trips per chunk).
__device__ __noinline__ void bandp_load_rows(double *Ag, const double *S, size_t ld, const double *g, int r0_, int first_, int n_, int nbr_, int base_, int gl0_, int bw_, int glim_) {
    lds_f64 *A = (lds_f64 *)Ag;
    // (arguments of a function arrive in vector registers: back to the scalar unit)
    const int r0 = __builtin_amdgcn_readfirstlane(r0_), first = __builtin_amdgcn_readfirstlane(first_), n = __builtin_amdgcn_readfirstlane(n_), nbr = __builtin_amdgcn_readfirstlane(nbr_),
              base = __builtin_amdgcn_readfirstlane(base_), gl0 = __builtin_amdgcn_readfirstlane(gl0_), bw = __builtin_amdgcn_readfirstlane(bw_), glim = __builtin_amdgcn_readfirstlane(glim_);
    constexpr int NW = BANDP_T/64, LOAD_U = 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int k = wave; k < nbr; k += NW) { lds_f64 *row = A + rowoff(n + k);
        for (int c = r0 + lane; c < n; c += 64) row[c] = 0.0;
        if (first) for (int c = lane; c <= k; c += 64) row[n + c] = 0.0; }
    if (first) __syncthreads();                              // (the band rows of the first chunk put the couplings into the border rows)
    const int per = bw + 6;                                  // the band is block-aligned: row r reaches back to column 6 (r/6) - bw
    lds_f64 *rhs = A + rowoff(n + nbr);
    double gv[2];
#pragma unroll
    for (int u = 0; u < 2; u++) { const int c = r0 + tid + u*BANDP_T; gv[u] = (c < n && base + c < glim) ? g[base + c] : 0.0; }
    for (int rb = r0 + wave; rb < n; rb += NW*LOAD_U) {
        double v[LOAD_U][2];
#pragma unroll
        for (int u = 0; u < LOAD_U; u++) {
            const int r = rb + NW*u, gr = base + r, cmin = 6*(r/6) - bw;
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                const int k = 64*kk + lane, c = r - k;
                v[u][kk] = 0.0;
                if (r < n && k < per && c >= cmin && (c >= 0 || (first && nbr > 0 && base + c >= gl0))) v[u][kk] = S[(size_t)gr*ld + (base + c)];
            }
        }
#pragma unroll
        for (int u = 0; u < LOAD_U; u++) {
            const int r = rb + NW*u, cmin = 6*(r/6) - bw;
            if (r >= n) continue;
            lds_f64 *row = A + rowoff(r);
            for (int c = lane; c < cmin; c += 64) row[c] = 0.0;            // left of the band
#pragma unroll
            for (int kk = 0; kk < 2; kk++) {
                const int k = 64*kk + lane, c = r - k;
                if (k < per && c >= cmin) {
                    if (c >= 0) row[c] = v[u][kk];
                    else if (first && nbr > 0 && base + c >= gl0) A[rowoff(n + (base + c - gl0)) + r] = v[u][kk];   // S(gr, gl) = border(gl, gr)
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; u++) { const int c = r0 + tid + u*BANDP_T; if (c < n) rhs[c] = gv[u]; }
    if (first) for (int k = tid; k < nbr; k += BANDP_T) rhs[n + k] = 0.0;
    __syncthreads();
}
// slide by s rows.  Virtual index v: band rows 0 .. m-1, border rows m .. mv-1, rhs row mv.  Row-wise through registers: a wave takes the
// virtual rows wave, wave + NW, ... (row addresses on the scalar unit, a lane a column of each 64-column chunk); ascending batches of SL_B
// rows per wave, read - barrier - write - barrier: a destination lies at or below its own source and below every source of a later
// batch.  (The element-wise version spent ~60 instructions of index arithmetic per element -- tri_row, two rowoff, four selects.)
__device__ __noinline__ void bandp_slide(double *Ag, int n_, int n_new_, int m_, int s_, int mv_) {
    lds_f64 *A = (lds_f64 *)Ag;
    typedef __attribute__((address_space(3))) v2d lds_v2d;
    // (arguments of a function arrive in vector registers: back to the scalar unit)
    const int n = __builtin_amdgcn_readfirstlane(n_), n_new = __builtin_amdgcn_readfirstlane(n_new_), m = __builtin_amdgcn_readfirstlane(m_),
              s = __builtin_amdgcn_readfirstlane(s_), mv = __builtin_amdgcn_readfirstlane(mv_);
    constexpr int NW = BANDP_T/64, SL_B = (2*(BAND_BW_MAX/2) + 6 + NW)/NW, SL_C = (2*(BAND_BW_MAX/2) + 6 + 127)/128;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // two doubles per lane: rows start 16-byte aligned, s, m, n, n_new are even, so a pair never straddles the band / border boundary
    // (the second slot of a row's last pair is the row's padding)
    const int nch = (mv + 127) >> 7;
    v2d sv[SL_B][SL_C];
#pragma unroll
    for (int u = 0; u < SL_B; u++) {
        const int r = min(wave + NW*u, mv);
        const lds_f64 *src = A + rowoff(r < m ? r + s : n + (r - m));
        const int ncol = r < mv ? r + 1 : mv;
#pragma unroll
        for (int k = 0; k < SL_C; k++) if (k < nch) { const int c = min(128*k + 2*lane, (ncol - 1) & ~1); sv[u][k] = *(const lds_v2d *)(src + (c < m ? c + s : n + (c - m))); }
    }
    lds_barrier();
#pragma unroll
    for (int u = 0; u < SL_B; u++) {
        const int r = wave + NW*u;
        if (r > mv) continue;
        lds_f64 *dst = A + rowoff(r < m ? r : n_new + (r - m));
        const int ncol = r < mv ? r + 1 : mv;
#pragma unroll
        for (int k = 0; k < SL_C; k++) { const int c = 128*k + 2*lane; if (c < ncol) *(lds_v2d *)(dst + (c < m ? c : n_new + (c - m))) = sv[u][k]; }
    }
    lds_barrier();
}

// finished column blocks jstart .. jend-1 of the window -> HBM.  A wave takes a column block, a lane one (row block b, column cc) of its L
// panel -- six rows, i.e. 48 contiguous bytes of the [b][cc][ri] record -- and one border row (six contiguous columns): 16-byte stores,
// no division per element (the element-wise loop: two run-time divisions and ~80 instructions per 8-byte store).
__device__ __noinline__ void bandp_write_out(double *Ag, double *LDg, double *Lrow, double *Lb, double *LDbuf, double *Sy, int jstart_, int jend_, int base_, int n_, int nbr_, int bw_) {
    lds_f64 *A = (lds_f64 *)Ag, *LD = (lds_f64 *)LDg;
    typedef __attribute__((address_space(3))) v2d lds_v2d;
    const int jstart = __builtin_amdgcn_readfirstlane(jstart_), jend = __builtin_amdgcn_readfirstlane(jend_), base = __builtin_amdgcn_readfirstlane(base_),
              n = __builtin_amdgcn_readfirstlane(n_), nbr = __builtin_amdgcn_readfirstlane(nbr_), bw = __builtin_amdgcn_readfirstlane(bw_);
    constexpr int NW = BANDP_T/64;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int REC = bw*6, NR = n + nbr;
    for (int q = jstart + wave; q < jend; q += NW) {
        const int gq = base/6 + q, c0 = 6*q;
        for (int t = lane; t < REC/6; t += 64) {                 // (b, cc): rows 6 q + 6 + 6 b + ri
            const int b = t/6, cc = t - 6*b, r0 = c0 + 6 + 6*b;
            double v[6];
#pragma unroll
            for (int ri = 0; ri < 6; ri++) v[ri] = A[rowoff(min(r0 + ri, n - 1)) + c0 + cc];
            double *o = Lrow + (size_t)(gq + 1 + b)*REC + b*36 + cc*6;
            if (r0 + 5 < n) { ((v2d *)o)[0] = v2d{v[0], v[1]}; ((v2d *)o)[1] = v2d{v[2], v[3]}; ((v2d *)o)[2] = v2d{v[4], v[5]}; }
            else {
#pragma unroll
                for (int ri = 0; ri < 6; ri++) if (r0 + ri < n) o[ri] = v[ri];
            }
        }
        for (int br = lane; br < bw; br += 64) {                 // border panel of the column block: [border row][cc]
            v2d x0 = {0.0, 0.0}, x1 = x0, x2 = x0;
            if (br < nbr) { const lds_v2d *src = (const lds_v2d *)(A + rowoff(n + br) + c0); x0 = src[0]; x1 = src[1]; x2 = src[2]; }
            v2d *o = (v2d *)(Lb + (size_t)gq*REC + 6*br);
            o[0] = x0; o[1] = x1; o[2] = x2;
        }
        if (lane < 15) LDbuf[32*(size_t)gq + lane] = LD[SOLVE_LD*q + lane];
        else if (lane >= 16 && lane < 22) LDbuf[32*(size_t)gq + lane] = LD[SOLVE_LD*q + LD_ID + (lane - 16)];
        else if (lane >= 24 && lane < 30) Sy[6*gq + (lane - 24)] = A[rowoff(NR) + c0 + (lane - 24)];
    }
}

// T_p layout: nT = nR + nL rows ([right separator rows; left separator rows]), dense nTmax x nTmax row-major (lower used) + gT
// BANDP_PW = 2 panel waves.  A step's panel has bw band rows + bw border rows + the rhs row -- 121 rows at a band of 60, which two panel
// waves (58 rows each) take in two rounds.  Three panel waves (ONE round, nine update waves), measured twice in round 2 on the
// 5000-keyframe map (cycle stamps of the factor loops of one interior): 484 k cycles against 448 k with two -- the step is bound by
// the update waves (a third panel wave takes a SIMD's issue slots from them), not by the second panel round.
__global__ __launch_bounds__(BANDP_T) void k_bandp_factor(Work W, int bw, int CB, int Pmax, double *Lrow, double *Lb, double *Tbuf) {
    LmState *st = W.st;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int fail;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = BANDP_T/64, NT = NW - BANDP_PW;
    if (st->done || st->step_fail || st->lin_done) return;
    const int B = bw/6, nb = bandp_nb(W, B);
    if (nb == 0) return;
    BandpPart PT;
    { const int4 q = bandp_part_call(W.nfree, W.ring, W.ring_g, B, Pmax, blockIdx.x); PT.a = q.x; PT.b = q.y; PT.has_left = q.z & 1; PT.has_right = q.z >> 1; PT.P = q.w; }
    if ((int)blockIdx.x >= PT.P) return;
    const int nbr = PT.has_left ? bw : 0;                       // border rows: the left separator
    const int row_lim = 6*(PT.has_right ? PT.b + B : PT.b);     // rows of the band this workgroup ever holds
    const int col_end = 6*PT.b;                                 // interior columns end here
    const int Wn = 6*CB + bw;
    const size_t ld = (size_t)W.ldS;
    const double *S = W.S;
    double *A = smem;
    double *LD = A + rowoff(Wn + bw + 2) + 16;
    double *scr = LD + SOLVE_LD*(Wn/6);
    if (tid == 0) fail = 0;

    int base = 6*PT.a, n = min(Wn, row_lim - base);
    const int gl0 = 6*(PT.a - B);                               // first row of the left separator
    // rows [r0, n) of the window from HBM; for the border / rhs rows the columns [r0, n).  first: also the border-border block
    const int glim = 6*(*W.nfree);                             // (ring: the ghost rows behind the last free pose have no gradient of their own)
    auto load_rows = [&](int r0, bool first) { bandp_load_rows(A, S, ld, W.g, r0, first ? 1 : 0, n, nbr, base, gl0, bw, glim); };
    long long tF = 0, tW = 0, tS = 0, tL = 0, tx = clock64(); int nchunks = 0;       // phase stamps of interior 1 (-> W.dbg[32..36])
    load_rows(0, true);
    { const long long t_ = clock64(); tL += t_ - tx; tx = t_; }
    bool first = true;
    for (;;) {
        const bool last = base + n == row_lim;
        const int jend = last ? (col_end - base)/6 : CB, jstart = first ? 0 : 1;
        const bool flush = last && (n + nbr > 6*jend);          // rows stay behind: the last panel must still be applied to them
        const int NR = n + nbr;                                 // the rhs row
        for (int jb = jstart; jb < jend + (flush ? 1 : 0) && !fail; jb++) {
            const int j0 = 6*jb, R0 = j0 + 6, p0 = j0 - 6;
            const bool fl = jb == jend;                         // flush step: no factorisation, panel jb-1 onto everything right of it
            if (wave < BANDP_PW) {
                double Lk[36], dprev[6];
                if (jb > 0) {
                    ld6(LD + SOLVE_LD*(jb - 1) + LD_D, dprev);
#pragma unroll
                    for (int c = 0; c < 6; c++) ld6(A + rowoff(min(j0 + c, n - 1)) + p0, Lk + 6*c);
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
                // rows of this block column that can be non-zero: the band rows < re, then (virtual rows re ..) the border rows and the rhs row
                const int re = min(n, R0 + bw), nx = nbr + 1;
                auto vrow = [&](int iv) { return iv < re ? iv : n + (iv - re); };
                if (!fl) {
                    const int i0 = lane < 6 ? j0 + lane : R0 + wave*SOLVE_PROWS + lane - 6;
                    double a[6];
                    load_row(vrow(min(i0, re + nx - 1)), a);
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
                    auto solve_row = [&](int i, double a[6]) {       // x L^T = a (right-looking), stored row = x D^-1
#pragma unroll
                        for (int c = 0; c < 5; c++)
#pragma unroll
                            for (int q = c + 1; q < 6; q++) a[q] = fma(-a[c], l[tri(q - 1) + c], a[q]);
#pragma unroll
                        for (int c = 0; c < 6; c++) a[c] *= id[c];
                        st6(A + rowoff(i) + j0, a);
                    };
                    if (lane >= 6) {
                        if (i0 < re + nx) solve_row(vrow(i0), a);
                        for (int i = i0 + BANDP_PW*SOLVE_PROWS; i < re + nx; i += BANDP_PW*SOLVE_PROWS) { load_row(vrow(i), a); solve_row(vrow(i), a); }
                    }
                }
            } else if (jb > 0) {
                // trailing update with panel jb-1 (its band ends at row re - 1): virtual rows R0 .. re-1 (band), re .. re+nbr-1 (border),
                // re+nbr (rhs); virtual columns R0 .. re+nbr-1
                // (flush step: the last panel onto EVERYTHING that stays behind, block column jb included -- no panel wave work)
                const int Rs = fl ? j0 : R0;
                const int re = max(Rs, min(n, j0 + bw)), vend = re + nbr;    // vend = virtual index of the rhs row
                // columns: the band part only.  The border x border block and the border part of the rhs never feed back into the
                // factorisation -- they are sums over ALL interior columns, formed afterwards from the stored border panel on many
                // workgroups (k_bandp_border) instead of costing this workgroup 10 of its 28 MFMA tiles per step
                const int mr = vend - Rs + 1, mcb = re - Rs;
                const double *ldp = LD + SOLVE_LD*(jb - 1);
                if (mcb > 0) {
                    const int ntr = (mr + 15) >> 4, ntcb = (mcb + 15) >> 4, ntri = tri(ntcb), ntile = ntri + (ntr - ntcb)*ntcb;
                    const int lr = lane & 15, lk = lane >> 4;
                    const int k1 = min(4 + lk, 5);
                    const double dk0 = ldp[LD_D + lk], dk1 = lk < 2 ? ldp[LD_D + 4 + lk] : 0.0;
                    auto real = [&](int v) { return v < re ? v : n + (v - re); };
                    for (int t = wave - BANDP_PW; t < ntile; t += NT) {
                        int ti, tj;
                        if (t < ntri) { ti = tri_row(t); tj = t - tri(ti); } else { const int u = t - ntri; ti = ntcb + u/ntcb; tj = u - (ti - ntcb)*ntcb; }
                        const int r0v = Rs + 16*ti, c0v = Rs + 16*tj;
                        double a0, a1, b0, b1;
                        const bool rows_band = r0v + 15 < re, rows_border = r0v >= re && r0v + 15 < vend;
                        if (ti > tj && c0v + 15 < re && (rows_band || rows_border)) {
                            // nothing to mask or clamp: the 16 rows are all band rows or all border rows (consecutive real rows either
                            // way), the columns all band columns left of them; row offsets by recurrence (rowoff(i + 4) = rowoff(i) + 4 i + 12)
                            const int rr0 = rows_band ? r0v : n + (r0v - re);                    // real row of the tile's first row
                            const int arow = rowoff(rr0 + lr) + p0, brow = rowoff(c0v + lr) + p0;
                            a0 = -A[arow + lk]; a1 = -A[arow + k1]; b0 = A[brow + lk]*dk0; b1 = A[brow + k1]*dk1;
                            if (lk >= 2) { a1 = 0.0; b1 = 0.0; }
                            const int cv0 = rr0 + lk, ccol = c0v + lr;
                            const int ci0 = rowoff(cv0) + ccol, ci1 = ci0 + 4*cv0 + 12, ci2 = ci1 + 4*cv0 + 28, ci3 = ci2 + 4*cv0 + 44;
                            v4d c = { A[ci0], A[ci1], A[ci2], A[ci3] };
                            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c, 0, 0, 0);
                            A[ci0] = c[0]; A[ci1] = c[1]; A[ci2] = c[2]; A[ci3] = c[3];
                            continue;
                        }
                        const int arow = rowoff(real(min(r0v + lr, vend))) + p0, brow = rowoff(min(c0v + lr, re - 1)) + p0;
                        a0 = -A[arow + lk]; a1 = -A[arow + k1]; b0 = A[brow + lk]*dk0; b1 = A[brow + k1]*dk1;
                        if (lk >= 2) { a1 = 0.0; b1 = 0.0; }
                        const int ccol = c0v + lr;                                        // band column
                        v4d c; int ci[4]; bool ok[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            const int cv = r0v + lk + 4*r;                                // virtual row
                            ok[r] = cv <= vend && ccol <= cv && ccol < re;
                            ci[r] = rowoff(real(min(cv, vend))) + ccol;
                            c[r] = ok[r] ? A[ci[r]] : 0.0;
                        }
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c, 0, 0, 0);
#pragma unroll
                        for (int r = 0; r < 4; r++) if (ok[r]) A[ci[r]] = c[r];
                    }
                }
            }
            __syncthreads();                       // panel jb complete, trailing update with panel jb-1 complete
        }
        { const long long t_ = clock64(); tF += t_ - tx; tx = t_; }
        if (fail) break;
        // ---------------- finished columns -> HBM: L by row block, the border panel, unit-lower diagonal factor + 1/d, v = D^-1 L^-1 g
        bandp_write_out(A, LD, Lrow, Lb, W.LDbuf, W.Sy, jstart, jend, base, n, nbr, bw);
        { const long long t_ = clock64(); tW += t_ - tx; tx = t_; nchunks++; }
        if (tid == 0 && W.dbg && blockIdx.x == 1) { W.dbg[32] = tF; W.dbg[33] = tW; W.dbg[34] = tS; W.dbg[35] = tL; W.dbg[19] = nchunks; }
        if (last) {
            // ---------------- what stays behind: T_p = [right separator rows; border rows] x the same (lower), reduced rhs
            const int nR = n - 6*jend, nT = nR + nbr, nTm = 2*bw;
            double *T = Tbuf + (size_t)blockIdx.x*((size_t)nTm*nTm + nTm), *gT = T + (size_t)nTm*nTm;
            __syncthreads();
            auto realT = [&](int v) { return v < nR ? 6*jend + v : n + (v - nR); };
            for (int e = tid; e < nT*nT; e += BANDP_T) { const int i = e/nT, j = e - i*nT; if (j <= i) T[(size_t)i*nTm + j] = A[rowoff(realT(i)) + realT(j)]; }
            for (int i = tid; i < nT; i += BANDP_T) gT[i] = A[rowoff(NR) + realT(i)];
            break;
        }
        // ---------------- slide by s rows.  Virtual index v: band rows 0 .. m-1, border rows m .. m+nbr-1, rhs row m+nbr; ascending
        // packed order through registers: a destination lies at or below its own source and below every source not yet read
        {
            const int s = 6*(jend - 1), m = n - s, n_new = min(Wn, row_lim - (base + s)), mv = m + nbr, ne = tri(mv) + mv;
            lds_barrier();
            (void)ne;
            bandp_slide(A, n, n_new, m, s, mv);
            if (tid < SOLVE_LD) LD[tid] = LD[SOLVE_LD*(jend - 1) + tid];
            base += s; n = n_new;
            { const long long t_ = clock64(); tS += t_ - tx; tx = t_; }
            load_rows(m, false);
            { const long long t_ = clock64(); tL += t_ - tx; tx = t_; }
            first = false;
        }
    }
}

// ---- the deferred part of T_p: LL = -sum_j Lb_j D_j Lb_j^T and gL = -sum_j Lb_j D_j v_j over the interior's column blocks, from the border
// panel in HBM.  grid (P, BANDP_NS): every workgroup takes a slice of the interior's blocks and writes its partial (summed in a
// fixed order by k_bandp_sep: deterministic).  part layout per (p, slice): [nbr x nbr] (lower used) | [nbr]
#ifndef BANDP_NS
#define BANDP_NS 4                          // slices of an interior in k_bandp_border (5000 keyframes, 127 interiors of 29 blocks: 8 slices 17.05 ms per solve, 4: 16.80, 2: 16.85)
#endif
#define BANDP_JC 8
__global__ __launch_bounds__(256) void k_bandp_border(Work W, int bw, int Pmax, const double *Lb, double *part) {
    const LmState *st = W.st;
    if (st->done || st->step_fail || st->lin_done) return;
    const int B = bw/6, nb = bandp_nb(W, B);
    if (nb == 0) return;
    const BandpPart PT = bandp_part_w(W, B, Pmax, blockIdx.x);
    if ((int)blockIdx.x >= PT.P || !PT.has_left) return;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int nbr = bw, REC = bw*6, tid = threadIdx.x;
    double *sL = smem, *sLd = smem + (size_t)BANDP_JC*REC, *sv = sLd + (size_t)BANDP_JC*REC;       // Lb, Lb o d, v   per staged block
    const int q = PT.b - PT.a, per = (q + BANDP_NS - 1)/BANDP_NS, j_lo = PT.a + blockIdx.y*per, j_hi = min(PT.b, j_lo + per);
    const int ntask = tri(nbr) + nbr;                           // (k >= k') pairs, then the nbr entries of gL
    constexpr int NU = (BAND_BW_MAX/2*(BAND_BW_MAX/2 + 1)/2 + BAND_BW_MAX/2 + 255)/256;       // tasks per thread at the widest border (78 rows: 3159 tasks)
    int tk[NU], tq[NU]; double acc[NU];
#pragma unroll
    for (int u = 0; u < NU; u++) { const int t = tid + 256*u; acc[u] = 0.0; tk[u] = -1; tq[u] = 0;
        if (t < tri(nbr)) { tk[u] = tri_row(t); tq[u] = t - tri(tk[u]); } else if (t < ntask) { tk[u] = t - tri(nbr); tq[u] = -1; } }
    for (int j0 = j_lo; j0 < j_hi; j0 += BANDP_JC) {
        const int nj = min(BANDP_JC, j_hi - j0);
        __syncthreads();
        for (int e = tid; e < nj*REC; e += 256) { const int jj = e/REC, k = e - jj*REC, a6 = k % 6;
            const double v = Lb[(size_t)(j0 + jj)*REC + k], d = 1.0/W.LDbuf[32*(size_t)(j0 + jj) + 16 + a6];
            sL[e] = v; sLd[e] = v*d; }
        for (int e = tid; e < nj*6; e += 256) sv[e] = W.Sy[6*(size_t)j0 + e];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < NU; u++) {
            if (tk[u] < 0) continue;
            double a = 0.0;
            if (tq[u] >= 0) for (int jj = 0; jj < nj; jj++) { const double *x = sL + jj*REC + 6*tk[u], *y = sLd + jj*REC + 6*tq[u];
                a += x[0]*y[0] + x[1]*y[1] + x[2]*y[2] + x[3]*y[3] + x[4]*y[4] + x[5]*y[5]; }
            else for (int jj = 0; jj < nj; jj++) { const double *x = sLd + jj*REC + 6*tk[u], *y = sv + 6*jj;
                a += x[0]*y[0] + x[1]*y[1] + x[2]*y[2] + x[3]*y[3] + x[4]*y[4] + x[5]*y[5]; }
            acc[u] -= a;
        }
    }
    double *o = part + ((size_t)blockIdx.x*BANDP_NS + blockIdx.y)*((size_t)nbr*nbr + nbr);
#pragma unroll
    for (int u = 0; u < NU; u++) { if (tk[u] < 0) continue; if (tq[u] >= 0) o[(size_t)tk[u]*nbr + tq[u]] = acc[u]; else o[(size_t)nbr*nbr + tk[u]] = acc[u]; }
}

// Compact block pool of the separator system for the cyclic-reduction solver (tsba_bandcr.h): only the blocks it ever touches --
// [D_0 .. D_{mmax-1}] [stride-1 couplings (k + 1, k)] [stride-2 couplings (2 (k + 1), 2 k)] ... each s x s row-major.  (The dense
// (P - 1) s square was 114 MB at P = 64 with rows 30 KB apart: every block row in another page.)
__device__ __host__ __forceinline__ size_t cr_blk_index(int mmax, int br, int bc) {
    if (br == bc) return (size_t)br;
    const int h = br - bc; size_t base = (size_t)mmax;
    for (int hh = 1; hh < h; hh <<= 1) base += (size_t)(mmax + hh - 1)/hh;
    return base + (size_t)(bc/h);
}
__device__ __forceinline__ double *cr_blk(double *pool, int s, int mmax, int br, int bc) { return pool + cr_blk_index(mmax, br, bc)*(size_t)s*s; }
__device__ __forceinline__ const double *cr_blk(const double *pool, int s, int mmax, int br, int bc) { return pool + cr_blk_index(mmax, br, bc)*(size_t)s*s; }
static size_t cr_pool_blocks(int mmax) { size_t n = (size_t)mmax; for (int hh = 1; hh < 2*mmax; hh <<= 1) n += (size_t)(mmax + hh - 1)/hh; return n + 2; }

// ---- separator system: dense row-major (ld = nsep_ld) or, blocked = 1, the block pool above; rows of separator s at [6 B s, 6 B (s + 1)); number of separator pose
// blocks -> *nfree_sep (what k_band_solve reads)
__global__ __launch_bounds__(256) void k_bandp_sep(Work W, int bw, int Pmax, const double *Tbuf, const double *part, double *Ssep, int nsep_ld, double *gsep, int *nfree_sep, int blocked) {
    const LmState *st = W.st;
    if (st->done || st->step_fail || st->lin_done) return;
    const int B = bw/6;
    const BandpPart P0 = bandp_part_w(W, B, Pmax, 0);
    const int P = P0.P, nS = bw, nTm = 2*bw;
    if (blockIdx.x == 0 && threadIdx.x == 0) *nfree_sep = (P - 1)*B;
    const int s = blockIdx.x;                                    // separator s sits between interiors s and s + 1
    if (s >= P - 1) return;
    const double *Ta = Tbuf + (size_t)s*((size_t)nTm*nTm + nTm), *ga = Ta + (size_t)nTm*nTm;            // interior s: this separator is its RIGHT one (rows 0 .. nS-1 of T)
    const double *Tb = Tbuf + (size_t)(s + 1)*((size_t)nTm*nTm + nTm), *gb = Tb + (size_t)nTm*nTm;      // interior s + 1: its LEFT one (rows nRb .. of T)
    const int nRb = (s + 1 < P - 1) ? nS : 0;                    // interior s + 1 has a right separator unless it is the last
    for (int e = threadIdx.x; e < nS*nS; e += 256) {
        const int i = e/nS, j = e - i*nS;
        if (j <= i) { double v = Ta[(size_t)i*nTm + j];          // RR of interior s (from its window) + LL of interior s + 1 (k_bandp_border partials)
            for (int sl = 0; sl < BANDP_NS; sl++) v += part[((size_t)(s + 1)*BANDP_NS + sl)*((size_t)nS*nS + nS) + (size_t)i*nS + j];
            if (blocked) cr_blk(Ssep, nS, Pmax - 1, s, s)[(size_t)i*nS + j] = v; else Ssep[(size_t)(nS*s + i)*nsep_ld + nS*s + j] = v; }
        // coupling to the NEXT separator through interior s + 1: T_{s+1}(border row i, right-separator column j) = S(sep s row i, sep s+1 col j)
        if (nRb > 0) { const double v = Tb[(size_t)(nRb + i)*nTm + j];
            if (blocked) cr_blk(Ssep, nS, Pmax - 1, s + 1, s)[(size_t)j*nS + i] = v; else Ssep[(size_t)(nS*(s + 1) + j)*nsep_ld + nS*s + i] = v; }
    }
    for (int i = threadIdx.x; i < nS; i += 256) { double v = ga[i];
        for (int sl = 0; sl < BANDP_NS; sl++) v += part[((size_t)(s + 1)*BANDP_NS + sl)*((size_t)nS*nS + nS) + (size_t)nS*nS + i];
        gsep[nS*s + i] = v; }
    (void)gb;
}

// ---- separator system, block pool (cyclic reduction), in ONE launch: a workgroup per separator forms the deferred part of its right-hand
// interior's T_p itself -- LL = -sum_j Lb_j D_j Lb_j^T, gL = -sum_j Lb_j D_j v_j from the border panel in HBM, as ONE product on the matrix
// cores: rows = border rows (+ one row holding v), K = (column block, column) of the interior -- and adds the window part of its
// left-hand interior.  (k_bandp_border on P x 4 workgroups + a clear of the partials + k_bandp_sep: 20 + 7 + 20 us per LM trial at
// 5000 keyframes; the sequential separator solve keeps those kernels.)
#define BSF_T 512
#define BSF_JC 32                           // interior column blocks per staged chunk: K = 192
#define BSF_LD (6*BSF_JC + 2)               // row stride of the staged panel (doubles; = 2 mod 4)
#define BSF_XR 80                           // staged rows: up to 78 border rows + the v row
static size_t bandp_sepf_lds_doubles() { return (size_t)BSF_XR*BSF_LD + 6*BSF_JC + 8; }
__global__ __launch_bounds__(BSF_T) void k_bandp_sepf(Work W, int bw, int Pmax, const double *Tbuf, const double *Lb, double *Ssep, double *gsep, int *nfree_sep) {
    const LmState *st = W.st;
    if (st->done || st->step_fail || st->lin_done) return;
    const int B = bw/6, nb = bandp_nb(W, B);
    if (nb == 0) return;
    const BandpPart P0 = bandp_part_w(W, B, Pmax, 0);
    const int P = P0.P, nS = bw, nTm = 2*bw, REC = bw*6;
    const size_t tsz = (size_t)nTm*nTm + nTm;
    // separator labels (pool, right-hand side and solution are indexed by label): chain s = 0 .. P - 2 between interiors s and s + 1; ring maps
    // lo .. off + G (bandp_part_ring): ir = the interior whose RIGHT separator this is (its window part RR of T; none for a loop without a tail's
    // first separator), il = the one whose LEFT separator it is (border products, coupling; none for the ghost)
    const int mm = cr_mmax(W.ring, Pmax, W.ring_g);
    const int off = W.ring && P0.Pt > 0 ? RING_OFF : 0, lo = W.ring ? (P0.Pt > 0 ? off - P0.Pt + 1 : off) : 0;
    const int nsep = W.ring ? off + P0.G - lo + 1 : P - 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) *nfree_sep = nsep*B;
    if ((int)blockIdx.x >= nsep) return;
    const int s = lo + (int)blockIdx.x;
    const int ir = W.ring ? s - off + P0.Pt - 1 : s, il = ir + 1 < P ? ir + 1 : -1;
    const double *Ta = ir >= 0 ? Tbuf + (size_t)ir*tsz : nullptr, *Tb = il >= 0 ? Tbuf + (size_t)il*tsz : nullptr;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *X = smem, *dK = smem + (size_t)BSF_XR*BSF_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lk = lane >> 4;
    // tiles of [border rows x d] [border rows | v]^T: the lower triangle of the 4 x 4 row tiles + column tile 3 (it holds v in column 60 ... nS)
    // -- for nS <= 48 the v row sits in an earlier tile: every (ti, tj) with tj <= ti or tj == tv is computed
    const int nrt = (nS + 15) >> 4, tv = nS >> 4;                 // row tiles of the border; column tile that holds the v row (row index nS)
    const int nct = tv + 1;
    v4d acc[3] = { {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0} };
    int tti[3], ttj[3], ntl = 0;
    {   int cnt = 0;                                             // tile list in a fixed order, dealt to the 8 waves round-robin (<= 3 each: 19 tiles at nS = 78)
        for (int ti = 0; ti < nrt; ti++) for (int tj = 0; tj < nct; tj++) { if (!(tj <= ti || tj == tv)) continue;
            if ((cnt & 7) == wave && ntl < 3) { tti[ntl] = ti; ttj[ntl] = tj; ntl++; }
            cnt++; }
    }
    if (il >= 0) {
        const BandpPart PT = bandp_part_w(W, B, Pmax, il);
        const int xrows = min(BSF_XR, 16*(tv + 1));              // rows any tile reads
        for (int j0 = PT.a; j0 < PT.b; j0 += BSF_JC) {
            const int nj = min(BSF_JC, PT.b - j0);
            __syncthreads();
            // stage: X[k][6 jj + cc] = Lb_j[k][cc] (k < nS), X[nS][.] = v_j, the other rows and the columns past 6 nj zero; dK = d_j[cc].
            // Pairs of doubles, eight per thread in flight.
            const int npair = xrows*(3*BSF_JC);
            for (int e0 = tid; e0 < npair; e0 += 8*BSF_T) {
                v2d v[8];
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int e = e0 + u*BSF_T, k = e/(3*BSF_JC), q = e - k*(3*BSF_JC), jj = q/3, c2 = 2*(q - 3*jj);
                    v[u] = v2d{0.0, 0.0};
                    if (e < npair && jj < nj) { if (k < nS) v[u] = *(const v2d *)(Lb + (size_t)(j0 + jj)*REC + k*6 + c2);
                                                else if (k == nS) v[u] = *(const v2d *)(W.Sy + 6*(size_t)(j0 + jj) + c2); }
                }
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const int e = e0 + u*BSF_T, k = e/(3*BSF_JC), q = e - k*(3*BSF_JC), jj = q/3, c2 = 2*(q - 3*jj);
                    if (e < npair) *(v2d *)(X + (size_t)k*BSF_LD + 6*jj + c2) = v[u];
                }
            }
            for (int e = tid; e < 6*BSF_JC; e += BSF_T) { const int jj = e/6, cc = e - 6*jj; dK[e] = jj < nj ? 1.0/W.LDbuf[32*(size_t)(j0 + jj) + 16 + cc] : 0.0; }
            __syncthreads();
            for (int u = 0; u < ntl; u++) {
                const double *pa = X + (size_t)(16*tti[u] + lr)*BSF_LD, *pb = X + (size_t)(16*ttj[u] + lr)*BSF_LD;
                for (int k0 = 0; k0 < 6*nj; k0 += 16) {
                    double av[4], bv[4];
#pragma unroll
                    for (int w = 0; w < 4; w++) { const int k = k0 + 4*w + lk; av[w] = pa[k]*dK[k]; bv[w] = pb[k]; }
#pragma unroll
                    for (int w = 0; w < 4; w++) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[w], bv[w], acc[u], 0, 0, 0);
                }
            }
        }
    }
    // ---- D_s (lower), g_s: window part of interior ir (or, ring separator 0, the diagonal block and gradient of S, g themselves) minus the products
    const size_t ldS = (size_t)W.ldS;
    double *Dss = cr_blk(Ssep, nS, mm, s, s);
    if (il >= 0) for (int u = 0; u < ntl; u++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int R = 16*tti[u] + lk + 4*r, Q = 16*ttj[u] + lr;
            if (R >= nS) continue;
            if (Q <= R) Dss[(size_t)R*nS + Q] = (Ta ? Ta[(size_t)R*nTm + Q] : W.S[(size_t)R*ldS + Q]) - acc[u][r];
            else if (Q == nS) gsep[nS*s + R] = (Ta ? (Ta + (size_t)nTm*nTm)[R] : W.g[R]) - acc[u][r];
        }
    }
    if (il < 0) {                                                // (ring: the ghost separator has no interior on its right)
        for (int e = tid; e < nS*nS; e += BSF_T) { const int i = e/nS, j = e - i*nS; if (j <= i) Dss[e] = Ta[(size_t)i*nTm + j]; }
        for (int i = tid; i < nS; i += BSF_T) gsep[nS*s + i] = (Ta + (size_t)nTm*nTm)[i];
        return;
    }
    // coupling to the next separator through interior il: T_il(border row i, right-separator column j) = S(sep s row i, sep s+1 col j)
    const bool has_r = W.ring || il < P - 1;                   // (every interior of a ring map has a separator on its right)
    if (has_r) { double *Cn = cr_blk(Ssep, nS, mm, s + 1, s);
        for (int e = tid; e < nS*nS; e += BSF_T) { const int i = e/nS, j = e - i*nS; Cn[(size_t)j*nS + i] = Tb[(size_t)(nS + i)*nTm + j]; } }
}

// ---- back substitution of the interiors (right-looking, as k_band_backsub), both separator solutions known
template <int NU>
__global__ __launch_bounds__(BAND_BS_T) void k_bandp_backsub(Work W, int bw, int Pmax, const double *Lrow, const double *Lb, const double *xsep) {
    LmState *st = W.st;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (st->done || st->step_fail || st->lin_done) return;
    const int B = bw/6, nb = bandp_nb(W, B);
    if (nb == 0) return;
    const BandpPart PT = bandp_part_w(W, B, Pmax, blockIdx.x);
    if ((int)blockIdx.x >= PT.P) return;
    const int REC = bw*6, NTASK = 6*B, RECB = 2*REC + 32;
    const int r_lo = PT.a, r_hi = PT.has_right ? PT.b + B : PT.b;          // row blocks walked: [r_lo, r_hi), the top B of them given
    double *buf0 = smem, *buf1 = smem + (size_t)BAND_CK*RECB, *ring = buf1 + (size_t)BAND_CK*RECB, *xL = ring + 6*BAND_RINGB, *xR = xL + bw;
    for (int k = tid; k < 6*BAND_RINGB; k += BAND_BS_T) ring[k] = 0.0;
    for (int k = tid; k < bw; k += BAND_BS_T) {
        xL[k] = PT.has_left ? xsep[(size_t)bw*PT.lblL + k] : 0.0;
        xR[k] = PT.has_right ? xsep[(size_t)bw*(PT.lblL + 1) + k] : 0.0;
        if (PT.has_right) W.Sy[6*PT.b + k] = xR[k];              // the separator's solution goes to its rows of x
        if (W.ring && PT.Pt == 0 && blockIdx.x == 0) W.Sy[k] = xL[k];     // (a loop without a tail: nobody has its first separator on its right but the ghost rows)
    }
    const int nrows = r_hi - r_lo, nchunk = (nrows + BAND_CK - 1)/BAND_CK;
    auto stage = [&](int chunk, double *buf, int t0, int nt) {
        const int jhi = r_hi - chunk*BAND_CK, jlo = max(r_lo, jhi - BAND_CK), nq = jhi - jlo;
        const int h = REC >> 1, n2 = nq*2*h;                                 // L parts and border parts, as double2
        const int tx = t0;
        double xv = 0.0; int xd = -1;
        if (tx < nq*32) { const int q = tx >> 5, u = tx & 31, j = jlo + q; xd = q*RECB + 2*REC + u;
            xv = j < PT.b ? (u < 24 ? W.LDbuf[32*(size_t)j + u] : (u < 30 ? W.Sy[6*j + (u - 24)] : 0.0)) : 0.0; }
        for (int e0 = t0; e0 < n2; e0 += 4*nt) {
            v2d v[4]; int dst[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int e = e0 + u*nt; dst[u] = -1;
                if (e < n2) { const int q = e/(2*h), k2 = e - q*2*h, j = jlo + q; dst[u] = q*RECB + 2*k2;
                    if (k2 < h) { const int col = j - 1 - (2*k2)/36;       // L(row block j, column block col): written only for interior columns
                        v[u] = (col >= PT.a && col < PT.b) ? ((const v2d *)(Lrow + (size_t)j*REC))[k2] : v2d{0.0, 0.0}; }
                    else v[u] = (j < PT.b && PT.has_left) ? ((const v2d *)(Lb + (size_t)j*REC))[k2 - h] : v2d{0.0, 0.0}; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) if (dst[u] >= 0) *(v2d *)(buf + dst[u]) = v[u];
        }
        if (xd >= 0) buf[xd] = xv;
    };
    stage(0, buf0, tid, BAND_BS_T);
    __syncthreads();
    int tb[NU], tc[NU];
#pragma unroll
    for (int u = 0; u < NU; u++) { const int t = lane + 64*u; tb[u] = t < NTASK ? t/6 : -1; tc[u] = t - 6*(t/6); }
    struct Ops { double Lr[NU][6], l[16], vv[6]; };
    // record data of one step; the left separator's term  sum_k Lb[k][c] xL[k]  is folded into v here (it does not depend on the
    // running solution): four partial sums per c on the lanes of a quad
    auto fetch = [&](const double *rec, Ops &o, bool interior) {
#pragma unroll
        for (int u = 0; u < NU; u++) if (tb[u] >= 0) ld6(rec + 6*(lane + 64*u), o.Lr[u]);
#pragma unroll
        for (int k = 0; k < 8; k++) { const v2d x2 = ((const v2d *)(rec + 2*REC))[k]; o.l[2*k] = x2.x; o.l[2*k + 1] = x2.y; }
        ld6(rec + 2*REC + 24, o.vv);
        if (interior && PT.has_left) {
            const int c = min(lane >> 2, 5), p = lane & 3;
            double acc = 0.0;
            for (int k = p; k < bw; k += 4) acc = fma(rec[REC + 6*k + c], xL[k], acc);
            acc = quad_sum(acc);
#pragma unroll
            for (int q = 0; q < 6; q++) o.vv[q] -= readlane_f64(acc, 4*q);
        }
    };
    auto step = [&](int r, const Ops &o, const double *nrec, Ops &on, bool ninterior) {
        double ur[6], uo[NU]; int slot[NU];
        ld6(ring + 6*(r & (BAND_RINGB - 1)), ur);
#pragma unroll
        for (int u = 0; u < NU; u++) {
            const int j = r - 1 - tb[u];
            slot[u] = (tb[u] >= 0 && j >= PT.a && j < PT.b) ? 6*(j & (BAND_RINGB - 1)) + tc[u] : -1;
            uo[u] = slot[u] >= 0 ? ring[slot[u]] : 0.0;
        }
        if (nrec) fetch(nrec, on, ninterior);
        double x[6];
        if (r >= PT.b) {                                          // a row block of the right separator: its solution is given
#pragma unroll
            for (int q = 0; q < 6; q++) x[q] = xR[6*(r - PT.b) + q];
        } else {
#pragma unroll
            for (int q = 5; q >= 0; q--) { double v = o.vv[q] - ur[q];
#pragma unroll
                for (int k = q + 1; k < 6; k++) v = fma(-o.l[tri(k - 1) + q], x[k], v);
                x[q] = v; }
        }
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
            ring[6*(r & (BAND_RINGB - 1)) + lane] = 0.0;
            if (r < PT.b) W.Sy[6*r + lane] = xv;
        }
        wave_lds_fence();
    };
    for (int ch = 0; ch < nchunk; ch++) {
        double *buf = (ch & 1) ? buf1 : buf0, *nxt = (ch & 1) ? buf0 : buf1;
        if (wave > 0) { if (ch + 1 < nchunk) stage(ch + 1, nxt, tid - 64, BAND_BS_T - 64); }
        else {
            const int jhi = r_hi - ch*BAND_CK, jlo = max(r_lo, jhi - BAND_CK);
            Ops oa, ob;
            int r = jhi - 1;
            fetch(buf + (size_t)(r - jlo)*RECB, oa, r < PT.b);
            for (; r - 1 >= jlo; r -= 2) {
                step(r, oa, buf + (size_t)(r - 1 - jlo)*RECB, ob, r - 1 < PT.b);
                step(r - 1, ob, r - 2 >= jlo ? buf + (size_t)(r - 2 - jlo)*RECB : nullptr, oa, r - 2 < PT.b);
            }
            if (r >= jlo) step(r, oa, nullptr, ob, false);
        }
        __syncthreads();
    }
}

__global__ void k_bandp_dp(Work W) {
    const LmState *st = W.st;
    if (st->done | st->lin_done) return;
    const int a = blockIdx.x*blockDim.x + threadIdx.x;
    if (a >= W.n_kf) return;
    const int ia = W.fidx[a];
#pragma unroll
    for (int k = 0; k < 6; k++) W.dp[6*a + k] = (ia >= 0 && !st->step_fail) ? -W.Sy[6*ia + k] : 0.0;
}
