// Kernels of a linearisation: reductions, start / pass set-up, mu-sigma and label rasters, k_linearize, k_mid, pose sums, k_postlin.  (part of the single translation unit tsba.hip: included there, in this order)
#pragma once
// ------------------------------------------------------------------------------------------------ kernels
__device__ __forceinline__ double wave_sum1(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int NT>
__device__ __forceinline__ double block_sum(double v, double *lds) {     // deterministic (fixed order), NT threads, all get the result
    const int t = threadIdx.x;
    lds[t] = v; __syncthreads();
    if (t < 64) {
        double s = lds[t];
#pragma unroll
        for (int k = 64; k < NT; k += 64) s += lds[t + k];
        s = wave_sum1(s);
        if (t == 0) lds[0] = s;
    }
    __syncthreads();
    const double r = lds[0]; __syncthreads();
    return r;
}
template <int NT>
__device__ __forceinline__ double block_max(double v, double *lds) {
    const int t = threadIdx.x;
    lds[t] = v; __syncthreads();
    if (t < 64) {
        double s = lds[t];
#pragma unroll
        for (int k = 64; k < NT; k += 64) s = fmax(s, lds[t + k]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s = fmax(s, __shfl_xor(s, o, 64));
        if (t == 0) lds[0] = s;
    }
    __syncthreads();
    const double r = lds[0]; __syncthreads();
    return r;
}

// Sum over a variable-length gather list with U entries (index, then value) in flight per round trip instead of one:
// val(idx) is evaluated for clamped indices and masked, the summation order is the list order.
template <int U, class F>
__device__ __forceinline__ double gather_sum(const int *list, int n, F &&val) {
    double s = 0.0;
    for (int base = 0; base < n; base += U) {
        int idx[U]; double v[U];
#pragma unroll
        for (int u = 0; u < U; u++) idx[u] = list[min(base + u, n - 1)];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = val(idx[u]);
#pragma unroll
        for (int u = 0; u < U; u++) s += base + u < n ? v[u] : 0.0;
    }
    return s;
}

// contiguous range with U loads in flight per round trip, summed in index order
template <int U>
__device__ __forceinline__ double range_sum(const double *v, int i0, int i1) {
    double s = 0.0;
    for (int base = i0; base < i1; base += U) {
        double x[U];
#pragma unroll
        for (int u = 0; u < U; u++) x[u] = v[min(base + u, i1 - 1)];
#pragma unroll
        for (int u = 0; u < U; u++) s += base + u < i1 ? x[u] : 0.0;
    }
    return s;
}

// ---- start point of a solve: parameters (both buffers), inlier flags, LM state
struct ResetSrc { const double *pose0, *rho0, *theta0; const uint8_t *sg0, *tg0, *tf0; long long n_pose, n_rho, n_theta, n_sg, n_tg, n_tf; };
__global__ __launch_bounds__(256) void k_reset_state(Work W, ResetSrc A) {
    const long long t = (long long)blockIdx.x*blockDim.x + threadIdx.x, n = (long long)gridDim.x*blockDim.x;
    for (long long k = t; k < A.n_pose; k += n) { const double v = A.pose0[k]; W.pose[0][k] = v; W.pose[1][k] = v; }
    for (long long k = t; k < A.n_rho; k += n) { const double v = A.rho0[k]; W.rho[0][k] = v; W.rho[1][k] = v; }
    for (long long k = t; k < A.n_theta; k += n) { const double v = A.theta0[k]; W.theta[0][k] = v; W.theta[1][k] = v; }
    for (long long k = t; k < A.n_sg; k += n) W.sgood[k] = A.sg0[k];
    for (long long k = t; k < A.n_tg; k += n) W.tobs_good[k] = A.tg0[k];
    for (long long k = t; k < A.n_tf; k += n) W.tfgood[k] = A.tf0[k];
    for (long long k = t; k < W.n_kf; k += n) W.kf_in[k] = 0;             // (what k_pass_reset clears: the first pass of a window starts with k_pass_begin, tsba_kernels_pass.h)
    for (long long k = t; k < W.n_pt; k += n) W.act_pt[k] = 0;
    for (long long k = t; k < W.n_text; k += n) W.act_tx[k] = 0;
    if (t == 0) { memset(W.st, 0, sizeof(LmState)); if (W.poll0) *W.poll0 = ts_poll_giveups; }
}

// ---- pass initialisation
__global__ void k_pass_reset(Work W, double radius0, int max_it) {
    LmState *s = W.st;
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        int cur = s->cur; long long nl = s->n_lin, nc = s->n_cost;
        memset(s, 0, sizeof(LmState));
        s->cur = cur; s->n_lin = nl; s->n_cost = nc;
        s->radius = radius0; s->decrease_factor = 2.0; s->need_lin = 1; s->first = 1; s->max_it = max_it;
        if (W.hprog) { *W.hprog = (unsigned long long)W.pass_seq << 32; __threadfence_system(); }
    }
    int t = blockIdx.x*blockDim.x + threadIdx.x, n = gridDim.x*blockDim.x;
    for (int k = t; k < W.n_kf; k += n) { W.kf_in[k] = 0; W.kf_const[k] = 0; }
    for (int k = t; k < W.n_pt; k += n) W.act_pt[k] = 0;
    for (int k = t; k < W.n_text; k += n) W.act_tx[k] = 0;
}

// which candidates are active (good flags), which keyframes participate (FLAG_KFIN, optimizer.cc:1410-1411,1428,1514-1515)
__device__ __forceinline__ void participation_wg(const Work &W, const LevelDev &L, const int bid, int partials) {
    // workgroups 0 .. nb_sc-1: one scene candidate per thread; the rest: one (KF, text) group per WAVE, its features on the lanes
    // (a thread walking the 64 features of a group alone was most of this kernel's 14 us)
    __shared__ int cnt_s, cnt_t;
    if (threadIdx.x == 0) { cnt_s = 0; cnt_t = 0; }
    __syncthreads();
    const int nb_sc = (L.n_sc + 255) >> 8, lane = threadIdx.x & 63;
    if (bid < nb_sc) {
        const int t = bid*256 + threadIdx.x;
        bool act = false;
        if (t < L.n_sc) {
            act = !W.filter_good || W.sgood[L.sc_flag[t]];
            if (act) {
                int pt = L.sc_pt[t], h = W.pt_host[pt];
                W.kf_in[L.sc_kf[t]] = 1;
                if (h >= 0) { W.kf_in[h] = 1; W.act_pt[pt] = 1; }
            }
        }
        const int nw = __popcll(__ballot(act));
        if (lane == 0 && nw) atomicAdd(&cnt_s, nw);
    } else {
        const int g = (bid - nb_sc)*4 + (threadIdx.x >> 6);
        if (g < L.n_tg) {
            const int tb = L.tg_tobs[g], j = L.tg_text[g];
            if (!W.filter_good || W.tobs_good[tb]) {
                const int f0 = L.tfeat_off[j], f1 = L.tfeat_off[j+1], fg = W.tobs_fgood_off[tb];
                int cnt = 0;
                for (int f = f0 + lane; f < f1; f += 64) if (!W.filter_good || W.tfgood[fg + L.tfeat_raw[f]]) cnt++;
                cnt = (int)wave_sum1((double)cnt);
                if (lane == 0 && cnt > 0) {
                    const int h = W.text_host[j];
                    W.kf_in[L.tg_kf[g]] = 1;
                    if (h >= 0) { W.kf_in[h] = 1; W.act_tx[j] = 1; }
                    atomicAdd(&cnt_t, cnt);
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // single GPU: per-workgroup partials, summed by the gauge kernel; multi-GPU: the counts are all-reduced before the gauge
        // kernel runs, so they go straight to the state
        if (partials) { W.cntpart[2*bid] = cnt_s; W.cntpart[2*bid + 1] = cnt_t; }
        else { if (cnt_s) atomicAdd(&W.st->ns_active, cnt_s); if (cnt_t) atomicAdd(&W.st->nt_active, cnt_t); }
    }
}
__global__ __launch_bounds__(256) void k_participation(Work W, LevelDev L, int partials) { participation_wg(W, L, (int)blockIdx.x, partials); }
// block counts of k_participation -> LM state (called by the gauge kernels' first wave / all threads)
__device__ __forceinline__ void sum_counts(const Work &W, int ncp, int tid, int nthreads, int *lds2 /* [2] zeroed */) {
    int a = 0, b = 0;
    for (int k = tid; k < ncp; k += nthreads) { a += W.cntpart[2*k]; b += W.cntpart[2*k + 1]; }
    if (a) atomicAdd(&lds2[0], a);
    if (b) atomicAdd(&lds2[1], b);
}
// gauge fixing, optimizer.cc:1562-1588 / :1825-1830
__global__ void k_gauge(Work W, const uint8_t *kf_initial, int state, int ncp, const int *order) {
    __shared__ int s_cnt2[2];
    if (threadIdx.x == 0) { s_cnt2[0] = 0; s_cnt2[1] = 0; }
    __syncthreads();
    if (ncp) { sum_counts(W, ncp, threadIdx.x, blockDim.x, s_cnt2); __syncthreads(); if (threadIdx.x == 0) { W.st->ns_active = s_cnt2[0]; W.st->nt_active = s_cnt2[1]; } }
    if (threadIdx.x || blockIdx.x) return;
    int cnt = 0;
    for (int k = 0; k < W.n_kf; k++) { if (kf_initial[k] && W.kf_in[k]) W.kf_const[k] = 1; cnt += W.kf_in[k]; }
    if (state == TSBA_STATE_LOCAL && cnt > 3) {
        int fixed = 0;
        for (int k = 0; k < W.n_kf && fixed < 3; k++) if (W.kf_in[k]) { W.kf_const[k] = 1; fixed++; }
    }
    int nf = 0;                                                // rows of S: free poses in keyframe order, or in the plan's order
    int row0 = 0;
    for (int i = 0; i < W.n_kf; i++) { const int k = order ? order[i] : i; if (i == W.ring_k0) row0 = nf; W.fidx[k] = (W.kf_in[k] && !W.kf_const[k]) ? nf++ : -1; }
    W.nfree[0] = nf; W.nfree[1] = row0;                         // (ring maps: free poses before the loop's first keyframe)
}

// windows of up to 64 keyframes: one lane per keyframe, ballots instead of the serial walk (7.8 -> ~2 us per pass)
__global__ __launch_bounds__(64) void k_gauge_wave(Work W, const uint8_t *kf_initial, int state, int ncp) {
    __shared__ int s_cnt2[2];
    const int k = threadIdx.x;
    if (k == 0) { s_cnt2[0] = 0; s_cnt2[1] = 0; }
    __syncthreads();
    if (ncp) { sum_counts(W, ncp, k, 64, s_cnt2); __syncthreads(); if (k == 0) { W.st->ns_active = s_cnt2[0]; W.st->nt_active = s_cnt2[1]; } }
    const bool on = k < W.n_kf;
    const int in = on ? W.kf_in[k] : 0, ini = on ? kf_initial[k] : 0;
    const unsigned long long m_in = __ballot(in != 0);
    int cst = (ini && in) ? 1 : 0;
    if (state == TSBA_STATE_LOCAL && __popcll(m_in) > 3) {
        const int before = __popcll(m_in & ((1ull << k) - 1));      // participating keyframes with a smaller index
        if (in && before < 3) cst = 1;                               // the first three of them are held constant
    }
    const bool fre = in && !cst;
    const unsigned long long m_free = __ballot(fre);
    if (on) { W.kf_const[k] = cst; W.fidx[k] = fre ? __popcll(m_free & ((1ull << k) - 1)) : -1; }
    if (k == 0) { W.nfree[0] = __popcll(m_free); W.nfree[1] = 0; }
}
// the same for large maps: 1024 threads, consecutive keyframes per thread, one block-wide exclusive scan for the compressed indices
// (the single-thread walk above costs 1.4 ms at 5000 keyframes)
__global__ __launch_bounds__(1024) void k_gauge_par(Work W, const uint8_t *kf_initial, int state, int ncp, const int *order) {
    __shared__ int s_scan[1024]; __shared__ int s_first[3]; __shared__ int s_cnt; __shared__ int s_cnt2[2];
    if (threadIdx.x == 0) { s_cnt2[0] = 0; s_cnt2[1] = 0; }
    __syncthreads();
    if (ncp) { sum_counts(W, ncp, threadIdx.x, 1024, s_cnt2); __syncthreads(); if (threadIdx.x == 0) { W.st->ns_active = s_cnt2[0]; W.st->nt_active = s_cnt2[1]; } }
    const int tid = threadIdx.x, per = (W.n_kf + 1023)/1024, k0 = tid*per, k1 = min(W.n_kf, k0 + per);
    if (tid == 0) {                                           // STATE_LOCAL: the first three participating keyframes are held constant
        int f = 0; s_first[0] = s_first[1] = s_first[2] = -1;
        if (state == TSBA_STATE_LOCAL) for (int k = 0; k < W.n_kf && f < 3; k++) if (W.kf_in[k]) s_first[f++] = k;
    }
    int cin = 0;
    for (int k = k0; k < k1; k++) cin += W.kf_in[k];
    s_scan[tid] = cin; __syncthreads();
    for (int d = 512; d > 0; d >>= 1) { if (tid < d) s_scan[tid] += s_scan[tid + d]; __syncthreads(); }
    if (tid == 0) s_cnt = s_scan[0];
    __syncthreads();
    const bool fix3 = state == TSBA_STATE_LOCAL && s_cnt > 3;
    int nfree = 0;                                             // from here on a thread's range is a range of POSITIONS (= keyframes without a plan order)
    for (int i = k0; i < k1; i++) {
        const int k = order ? order[i] : i;
        int cst = (kf_initial[k] && W.kf_in[k]) ? 1 : 0;
        if (fix3 && (k == s_first[0] || k == s_first[1] || k == s_first[2])) cst = 1;
        W.kf_const[k] = cst;
        nfree += (W.kf_in[k] && !cst) ? 1 : 0;
    }
    __syncthreads();
    s_scan[tid] = nfree; __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) { const int t = tid >= d ? s_scan[tid - d] : 0; __syncthreads(); s_scan[tid] += t; __syncthreads(); }
    int at = s_scan[tid] - nfree;                              // exclusive prefix
    if (tid == 0) W.nfree[1] = 0;
    __syncthreads();
    for (int i = k0; i < k1; i++) { const int k = order ? order[i] : i; if (i == W.ring_k0 && i > 0) W.nfree[1] = at; W.fidx[k] = (W.kf_in[k] && !W.kf_const[k]) ? at++ : -1; }
    if (tid == 1023) W.nfree[0] = s_scan[1023];
}

// ---- mu / sigma of a projected text box: tool::GetProjText x4 + tool::CalTextinfo (src/tool.cc:1178-1262,1655-1728)
// with cv::fillPoly's scan conversion (boundary Bresenham lines + 16.16 fixed-point scanline spans).  One workgroup per
// (KF, text) observation; the polygon mask of the clamped bounding box lives in LDS as a bit field.
#define MS_THREADS 256
__device__ __forceinline__ void musigma_wg(const Work &W, const LevelDev &L, const int g, const double *pose, const double *theta) {
    __shared__ unsigned mask[MS_MASK_WORDS];
    __shared__ unsigned hist[256];
    __shared__ int s_xy[8], s_bb[4];
    __shared__ double s_red[MS_THREADS];
    const int tid = threadIdx.x;
    // (round 6) the group's static record (tobs, keyframe, text, host: one 16-byte load) instead of three lists and then the host; what hangs off it -- the good
    // flag, both poses, the plane, the corner, the keyframe's image pointer -- requested together by every thread BEFORE the flag is looked at (a request under a
    // lane-dependent branch is waited for at the end of that branch): one dependent round trip where there were three, and the image pointer is there when the
    // histogram needs it
#ifdef MID_STAMPS                           // (tools/mid_stamps.sh: cycles of a mu / sigma workgroup by phase into W.dbg[56..63] -- the slots of k_schur_t's gradient rows in that build)
    const long long us_t0 = clock64();
#define MS_STAMP(slot) do { if (threadIdx.x == 0) atomicAdd((unsigned long long *)&W.dbg[56 + (slot)], (unsigned long long)(clock64() - us_t0)); } while (0)
#else
#define MS_STAMP(slot) do { } while (0)
#endif
    const int4 ra = ((const int4 *)L.tg_rec)[2*g];
    const int tb = ra.x, kf = ra.y, j = ra.z, h = ra.w;
    const uint8_t good_t = W.tobs_good[tb];
    const uint8_t *img = L.img[kf];
    const int bq_ = tid & 3;
    double pc[7], ph[12], th[3], mx, my;
#pragma unroll
    for (int k = 0; k < 7; k++) pc[k] = pose[7*kf + k];
#pragma unroll
    for (int k = 0; k < 12; k++) ph[k] = h >= 0 ? (k < 7 ? pose[7*h + k] : 0.0) : W.text_Twr[12*(size_t)j + k];
#pragma unroll
    for (int k = 0; k < 3; k++) th[k] = theta[3*j + k];
    mx = W.text_box[(j*4 + bq_)*2]; my = W.text_box[(j*4 + bq_)*2 + 1];
    if (W.filter_good && !good_t) { if (tid == 0) { W.musig[2*tb] = 0; W.musig[2*tb+1] = 0; } return; }
    MS_STAMP(0);                                              // (operands there)
    const int w = L.img_w, hh = L.img_h;
    __shared__ int s_c[16];
    if (tid < 4) {                                            // one box corner per lane (the serial walk over the four cost ~1.5 us of divisions)
        const int b = tid;
        Pose C; load_pose(pc, C);
        PairT T;
        if (h >= 0) { Pose Hs; load_pose(ph, Hs); pair_from_poses(C, Hs, T); }
        else pair_from_Twr(C, ph, T);
        double invz = -(mx*th[0] + my*th[1] + th[2]);
        double m[3] = { mx, my, 1.0 }, Rm[3]; mat3_vec(T.Rcr, m, Rm);
        double X = Rm[0]/invz + T.tq[0] + C.t[0], Y = Rm[1]/invz + T.tq[1] + C.t[1], Z = Rm[2]/invz + T.tq[2] + C.t[2];
        double cu = L.K[0]*X/Z + L.K[2], cv = L.K[1]*Y/Z + L.K[3];
        s_xy[2*b] = (int)cu; s_xy[2*b+1] = (int)cv;
        // the reference updates xMax / xMin only on strict improvement, starting from -1 / w + 1: a corner that does not improve
        // contributes nothing -- the same as taking max / min over the corners that do
        s_c[4*b] = cu > -1.0 ? (int)ceil(cu) : -1;            // candidate for xMax (initial value -1)
        s_c[4*b + 1] = cu < (double)(w + 1) ? (int)floor(cu) : w + 1;
        s_c[4*b + 2] = cv > -1.0 ? (int)ceil(cv) : -1;
        s_c[4*b + 3] = cv < (double)(hh + 1) ? (int)floor(cv) : hh + 1;
    }
    __syncthreads();
    if (tid == 0) {
        int xMax = max(max(s_c[0], s_c[4]), max(s_c[8], s_c[12])), xMin = min(min(s_c[1], s_c[5]), min(s_c[9], s_c[13]));
        int yMax = max(max(s_c[2], s_c[6]), max(s_c[10], s_c[14])), yMin = min(min(s_c[3], s_c[7]), min(s_c[11], s_c[15]));
        if (xMin < 0) xMin = 0;
        if (xMin >= w) xMin = w - 1;
        if (yMin < 0) yMin = 0;
        if (yMin >= hh) yMin = hh - 1;
        if (xMax >= w) xMax = w - 1;
        if (xMax < 0) xMax = 0;
        if (yMax >= hh) yMax = hh - 1;
        if (yMax < 0) yMax = 0;
        s_bb[0] = xMin; s_bb[1] = xMax; s_bb[2] = yMin; s_bb[3] = yMax;
    }
    MS_STAMP(1);                                              // (corners projected)
    for (int k = tid; k < min((w*hh + 31) >> 5, MS_MASK_WORDS); k += MS_THREADS) mask[k] = 0;
    hist[tid] = 0;
    __syncthreads();
    const int xMin = s_bb[0], xMax = s_bb[1], yMin = s_bb[2], yMax = s_bb[3];
    MS_STAMP(2);                                              // (mask cleared)
#ifdef MID_STAMPS
    raster_quad(mask, s_xy, w, hh, tid, MS_THREADS, W.dbg + 48);
#else
    raster_quad(mask, s_xy, w, hh, tid, MS_THREADS);
#endif
    __syncthreads();
    MS_STAMP(3);                                              // (quad rasterised)
    // histogram of masked pixels inside the clamped bounding box (tool.cc:1217-1232)
    int bw = xMax - xMin + 1, bh = yMax - yMin + 1;
    {   // four pixels per thread and round with their loads in flight together; (x, y) advance without a division per pixel
        const int npx = bw*bh, dx = MS_THREADS % bw, dy = MS_THREADS / bw;
        int x = tid % bw, y = tid / bw;
        for (int k0 = tid; k0 < npx; k0 += 4*MS_THREADS) {
            int bit[4]; bool in[4]; unsigned px[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                bit[u] = (yMin + y)*w + xMin + x;
                in[u] = k0 + u*MS_THREADS < npx && (mask[bit[u] >> 5] & (1u << (bit[u] & 31)));
                x += dx; y += dy; if (x >= bw) { x -= bw; y++; }
            }
#pragma unroll
            for (int u = 0; u < 4; u++) px[u] = in[u] ? img[bit[u]] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) if (in[u]) atomicAdd(&hist[px[u]], 1u);
        }
    }
    __syncthreads();
    MS_STAMP(4);                                              // (histogram)
    double cnt = (double)hist[tid], sum = (double)hist[tid]*(double)tid;
    double n = block_sum<MS_THREADS>(cnt, s_red), sm = block_sum<MS_THREADS>(sum, s_red);
    if (n < 2.0) { if (tid == 0) { W.musig[2*tb] = 0; W.musig[2*tb+1] = 0; } return; }
    double mu = sm/n;
    double d = (double)tid - mu;
    double ss = block_sum<MS_THREADS>((double)hist[tid]*d*d, s_red);
    if (tid == 0) { W.musig[2*tb] = mu; W.musig[2*tb+1] = sqrt(ss/(n - 1.0)); }
    MS_STAMP(5);
#ifdef MID_STAMPS
    if (threadIdx.x == 0) atomicAdd((unsigned long long *)&W.dbg[63], 1ull);
#endif
}

__global__ __launch_bounds__(MS_THREADS) void k_musigma(Work W, LevelDev L) {
    musigma_wg(W, L, blockIdx.x, W.pose[W.st->cur], W.theta[W.st->cur]);
}

// ---- text label image of one keyframe (optimizer::ShowBAReproj_TextBox -> tool::TextBoxWithFill, optimizer.cc:2508-2582,
// tool.cc:2103-2166): background -1, then every text observation of the keyframe in observation order fills its projected quad
// with its rank; later quads overwrite earlier ones, so ONE workgroup walks the observations sequentially (a keyframe sees a few
// dozen planes) and only the rasterisation of each quad is parallel.
#define LBL_THREADS 1024
__global__ __launch_bounds__(LBL_THREADS) void k_label(Work W, int kf, int w, int hh, double fx, double fy, double cx, double cy, float *out) {
    __shared__ unsigned mask[MS_MASK_WORDS];
    __shared__ int s_xy[8], s_bb[4];
    const int tid = threadIdx.x;
    const double *pose = W.pose[W.st->cur], *theta = W.theta[W.st->cur];
    for (int k = tid; k < w*hh; k += LBL_THREADS) out[k] = -1.0f;
    int rank = 0;
    for (int t = 0; t < W.n_tobs; t++) {
        if (W.tobs_kf[t] != kf) continue;                   // (uniform)
        const int j = W.tobs_text[t], h = W.text_host[j];
        if (tid == 0) {
            Pose C; load_pose(pose + 7*kf, C);
            PairT T;
            if (h >= 0) { Pose Hs; load_pose(pose + 7*h, Hs); pair_from_poses(C, Hs, T); }
            else pair_from_Twr(C, W.text_Twr + 12*j, T);
            const double th[3] = { theta[3*j], theta[3*j+1], theta[3*j+2] };
            int xMin = w, xMax = -1, yMin = hh, yMax = -1;
            for (int b = 0; b < 4; b++) {
                const double mx = W.text_box[(j*4 + b)*2], my = W.text_box[(j*4 + b)*2 + 1];
                const double invz = -(mx*th[0] + my*th[1] + th[2]);
                double m[3] = { mx, my, 1.0 }, Rm[3]; mat3_vec(T.Rcr, m, Rm);
                const double X = Rm[0]/invz + T.tq[0] + C.t[0], Y = Rm[1]/invz + T.tq[1] + C.t[1], Z = Rm[2]/invz + T.tq[2] + C.t[2];
                const double cu = fx*X/Z + cx, cv = fy*Y/Z + cy;
                const int iu = (int)cu, iv = (int)cv;                 // cv::Point(double, double): truncation
                s_xy[2*b] = iu; s_xy[2*b+1] = iv;
                xMin = min(xMin, iu); xMax = max(xMax, iu); yMin = min(yMin, iv); yMax = max(yMax, iv);
            }
            s_bb[0] = max(xMin, 0); s_bb[1] = min(xMax, w - 1); s_bb[2] = max(yMin, 0); s_bb[3] = min(yMax, hh - 1);
        }
        for (int k = tid; k < MS_MASK_WORDS; k += LBL_THREADS) mask[k] = 0;
        __syncthreads();
        raster_quad(mask, s_xy, w, hh, tid, LBL_THREADS);
        __syncthreads();
        const int x0 = s_bb[0], x1 = s_bb[1], y0 = s_bb[2], y1 = s_bb[3];     // the filled set lies inside the corners' bounding box
        const int bw = x1 - x0 + 1, bh = y1 - y0 + 1;
        if (bw > 0 && bh > 0)
            for (int k = tid; k < bw*bh; k += LBL_THREADS) {
                const int x = x0 + k % bw, y = y0 + k / bw, bit = y*w + x;
                if (mask[bit >> 5] & (1u << (bit & 31))) out[bit] = (float)rank;
            }
        rank++;
        __syncthreads();
    }
}

// ---- linearisation / cost.  grid = n_pair (scene waves) + n_tg (text waves), 64 threads each.
#define MODE_FULL 0
#define MODE_COST 1
// transpose-sum of N <= 28 per-lane values of ONE wave inside a two-wave workgroup: lane l (< N) returns the total of acc[l].
// reg: this wave's 28*65 doubles.  Both waves of the workgroup must call it (two workgroup barriers).
template <int N>
__device__ __forceinline__ double wave_sum_to_lane_mw(const double *acc, double *reg, int lane) {
#pragma unroll
    for (int i = 0; i < N; i++) reg[i*65 + lane] = acc[i];
    __syncthreads();
    double s = 0.0;
    if (lane < N) {
        const double *row = reg + lane*65;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int q = 0; q < 64; q += 4) { s0 += row[q]; s1 += row[q + 1]; s2 += row[q + 2]; s3 += row[q + 3]; }
        s = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    return s;
}
// the same for FOUR 16-lane groups per wave (small pairs share a wave): lane (group g, sub s) returns the group totals of acc[s] and
// acc[s + 16] (the latter only for s + 16 < N).  Both waves of the workgroup must call it.
template <int N>
__device__ __forceinline__ void wave_sum_groups16_mw(const double *acc, double *reg, int lane, double &t0, double &t1) {
#pragma unroll
    for (int i = 0; i < N; i++) reg[i*65 + lane] = acc[i];
    __syncthreads();
    const int g16 = lane & 48, sub = lane & 15;
    {
        const double *row = reg + sub*65 + g16;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int q = 0; q < 16; q += 4) { s0 += row[q]; s1 += row[q + 1]; s2 += row[q + 2]; s3 += row[q + 3]; }
        t0 = (s0 + s1) + (s2 + s3);
    }
    t1 = 0.0;
    if (sub + 16 < N) {
        const double *row = reg + (sub + 16)*65 + g16;
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int q = 0; q < 16; q += 4) { s0 += row[q]; s1 += row[q + 1]; s2 += row[q + 2]; s3 += row[q + 3]; }
        t1 = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
}
// 128-thread workgroups: a scene workgroup takes two (target, host) pairs (one per wave) -- or, PPW = 4 for maps whose pairs hold a
// dozen scene blocks (thousands of keyframes: a 64-lane wave per pair was 85 % idle), eight pairs, one per 16-lane group; a text workgroup one (KF, text)
// observation with every feature on TWO lanes (4 taps each): the text lanes' instruction stream (~450 instructions per tap at
// one instruction per ~4.5 cycles) is what bounds the kernel.  <= 256 VGPRs so that all ~730 workgroups of C4 are resident at once.
#ifndef LIN_TPL
#define LIN_TPL 4                        // photometric taps per lane: a feature's 8 taps sit on 8 / LIN_TPL neighbouring lanes.  4 = two lanes per
                                         // feature, 128-thread workgroups.  2 (four lanes per feature, 256 threads) was measured in round 2: the C4
                                         // level-0 launch went from 12.6 to 14.8 us -- the 55-value workgroup reduction is paid per wave, and
                                         // halving a lane's tap loop does not pay for twice the waves
#endif
#define LIN_T (64*(8/LIN_TPL))           // 64 features per text workgroup
#define LIN_NWV (LIN_T/64)
#define MID_U 4                          // slot records of a point that k_mid keeps in flight per round trip
// TEXT = false: levels without text planes (the reference's GlobalBA): the scene path alone needs far fewer registers than the text path.
template <int MODE, int PPW = 1, bool TEXT = true>
__device__ __forceinline__ void lin_body(const Work &W, const LevelDev &L, const int spec) {
    // spec = 0: linearise at x (pass start); spec = 1: speculative linearisation at the LM candidate, into the other LinBuf
    const LmState *st = W.st;
    constexpr int NWV = LIN_T/64, TPL = LIN_TPL, LPF = 8/LIN_TPL;   // waves per workgroup; taps per lane; lanes per feature
    __shared__ double lds[NWV*28*65 + NWV*64];
    __shared__ unsigned s_px[TEXT ? TPL*LIN_T : 1];          // the text path's pixel quads (four bytes): indexed by tap at run time (not registers)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double *reg = lds + wave*28*65, *xw = lds + NWV*28*65;
    // static indices of this workgroup first: in flight together with the LM state
    constexpr int LPP = 64/PPW;                              // lanes per pair
    const int nb_sc = (L.n_pair + NWV*PPW - 1)/(NWV*PPW);
    const int sub = PPW == 1 ? lane : (lane & (LPP - 1));
    // (scene-only launches: the workgroups of ONE XCD -- workgroup i runs on XCD i mod 8 -- take neighbouring pairs: the 64-byte slot records
    // of a landmark's observers share 128-byte lines, and neighbouring pairs observe the same landmarks.  With text groups the same mapping
    // was measured SLOWER on C4 (13.6 vs 13.0 us): the groups are the heavy workgroups and sit at the end of the index range -- contiguous
    // ranges put all of them on the last two XCDs.)  The grid is a multiple of 8 workgroups either way.
    // (Dispatching the text groups FIRST -- lowest workgroup indices -- was no better either: 13.2 us.)
    const int bq = TEXT ? (int)blockIdx.x : ((int)blockIdx.x & 7)*((int)gridDim.x >> 3) + ((int)blockIdx.x >> 3);
    if (bq >= nb_sc + (TEXT ? L.n_tg : 0)) return;
    const int pr = (NWV*bq + wave)*PPW + (PPW == 1 ? 0 : lane/LPP), prc = min(pr, max(L.n_pair - 1, 0));
    int pi = 0, ph = 0, pbeg = 0, pend = 0, tgpp = 0; int4 ra = {0, 0, 0, 0}, rb = {0, 0, 0, 0};
    if (bq < nb_sc) { pi = L.pair_i[prc]; ph = L.pair_h[prc]; pbeg = L.pair_sc_off[prc]; pend = pr < L.n_pair ? L.pair_sc_off[prc+1] : pbeg; }
    else if (TEXT) { ra = ((const int4 *)L.tg_rec)[2*(bq - nb_sc)]; rb = ((const int4 *)L.tg_rec)[2*(bq - nb_sc) + 1]; tgpp = L.tg_ppos[bq - nb_sc]; }   // one static record per group
    if (st->done) return;
    if (!spec && !st->need_lin) return;
    if (spec && st->step_fail) return;
#ifdef MID_STAMPS                           // (tools/mid_stamps.sh: cycles of a workgroup of the linearisation by kind -- scene pairs / text group -- into W.dbg[0..15])
    const long long ls_t0 = clock64(); const int ls_kind = bq < nb_sc ? 0 : 1;
#define LIN_STAMP(slot) do { if (threadIdx.x == 0 && spec) atomicAdd((unsigned long long *)&W.dbg[8*ls_kind + (slot)], (unsigned long long)(clock64() - ls_t0)); } while (0)
#else
#define LIN_STAMP(slot) do { } while (0)
#endif
    const int sel = spec ? (st->cur ^ 1) : st->cur;
    const LinBuf &B = W.lb[spec ? (st->lcur ^ 1) : st->lcur];
    const double *pose = W.pose[sel], *rho = W.rho[sel], *theta = W.theta[sel];
    const int b = bq;
    if (b < nb_sc) {
        // ---------------- scene observations of pair (i, h)
        const int i = pi, h = ph;
        Pose C; load_pose(pose + 7*i, C);
        PairT T;
        if (h >= 0) { Pose Hs; load_pose(pose + 7*h, Hs); pair_from_poses(C, Hs, T); }
        const bool fixed = (h < 0) && W.kf_const[i];           // all parameter blocks constant: not in the reduced program
        double acc[28];
#pragma unroll
        for (int k = 0; k < 28; k++) acc[k] = 0.0;
        const int beg = pbeg, end = pend;
#pragma unroll 1
        for (int c = beg + sub; c < end; c += LPP) {
            const int slot = L.sc_slot[c], pt = L.sc_pt[c];
            const bool act = !fixed && (!W.filter_good || W.sgood[L.sc_flag[c]]);
            // the slot record of an inactive candidate is zeros: one store sequence for both cases (a second, branchy one
            // costs the kernel ~170 VGPRs of live ranges)
            double wv[8];
#pragma unroll
            for (int k = 0; k < 8; k++) wv[k] = 0.0;
            if (act) {
                if (h < 0) pair_from_Trw(C, W.pt_Trw + 12*(size_t)pt, T);
                const double mx = W.pt_ray[2*pt], my = W.pt_ray[2*pt+1], rh = rho[pt];
                const double uo = L.sc_uv[2*c], vo = L.sc_uv[2*c+1];
                double r[2], jt[2][6], jl[2];
                scene_block(T, C.t, mx, my, rh, uo, vo, W.K0[0], W.K0[1], W.K0[2], W.K0[3], W.w_sx, W.w_sy, r, jt, jl);
                double wgt; acc[27] += 0.5*huber(r[0]*r[0] + r[1]*r[1], W.huber_s, wgt);
                int q = 0;
#pragma unroll
                for (int a = 0; a < 6; a++)
#pragma unroll
                    for (int cc = a; cc < 6; cc++) { acc[q] += wgt*(jt[0][a]*jt[0][cc] + jt[1][a]*jt[1][cc]); q++; }
#pragma unroll
                for (int a = 0; a < 6; a++) acc[21 + a] += wgt*(jt[0][a]*r[0] + jt[1][a]*r[1]);
#pragma unroll
                for (int a = 0; a < 6; a++) wv[a] = wgt*(jt[0][a]*jl[0] + jt[1][a]*jl[1]);
                wv[6] = wgt*(jl[0]*jl[0] + jl[1]*jl[1]);
                wv[7] = wgt*(jl[0]*r[0] + jl[1]*r[1]);
            }
            if (slot >= 0) {                         // (the host column -Q^T w is a function of w and the pair's R_cr: k_mid forms it)
#pragma unroll
                for (int k = 0; k < 8; k++) B.w_pt[(size_t)(slot)*PT_REC + k] = wv[k];
            }
        }
        LIN_STAMP(1);                                           // (the pair's candidates evaluated, slot records stored)
        if (MODE == MODE_COST) {
            double cs = acc[27];
            if (PPW == 1) cs = wave_sum1(cs);
            else {
#pragma unroll
                for (int o = LPP/2; o > 0; o >>= 1) cs += __shfl_xor(cs, o, LPP);
            }
            if (sub == 0 && pr < L.n_pair) B.pairCost[pr] = cs;
        } else if (PPW > 1) {
            double t0, t1;
            wave_sum_groups16_mw<28>(acc, reg, lane, t0, t1);
            if (pr < L.n_pair) {
                B.pairM[(size_t)sub*L.n_pair + pr] = t0;                       // values 0 .. 15
                if (sub + 16 < 27) B.pairM[(size_t)(sub + 16)*L.n_pair + pr] = t1;
                else if (sub + 16 == 27) B.pairCost[pr] = t1;
                if (h >= 0 && sub == 0) {
#pragma unroll
                    for (int k = 0; k < 9; k++) PAIRR(B, pr, k, L.n_pair) = T.Rcr[k];      // [pair][9]: k_mid reads a pair's rotation per landmark slot -- nine scattered loads with [9][pair]
                }
            }
        } else {
            double tot = wave_sum_to_lane_mw<28>(acc, reg, lane);
            if (pr < L.n_pair) {
                if (lane < 27) B.pairM[(size_t)lane*L.n_pair + pr] = tot;
                else if (lane == 27) B.pairCost[pr] = tot;
                if (h >= 0 && lane == 0) {
#pragma unroll
                    for (int k = 0; k < 9; k++) PAIRR(B, pr, k, L.n_pair) = T.Rcr[k];      // [pair][9]: k_mid reads a pair's rotation per landmark slot -- nine scattered loads with [9][pair]
                }
            }
        }
        LIN_STAMP(3);
#ifdef MID_STAMPS
        if (threadIdx.x == 0 && spec) atomicAdd((unsigned long long *)&W.dbg[8*ls_kind + 7], 1ull);
#endif
    } else if constexpr (TEXT) {
        // ---------------- photometric blocks of one (KF, text) observation: thread = (feature tid / LPF, tap group tid % LPF)
        const int tb = ra.x, i = ra.y, j = ra.z, h = ra.w, slot = rb.x, f0 = rb.y, f1 = rb.z, fg = rb.w;
        const double mu = W.musig[2*tb], sigma = W.musig[2*tb+1];
        const bool act_g = (!W.filter_good || W.tobs_good[tb]) && !((h < 0) && W.kf_const[i]) && sigma != 0.0;
        // static data of this thread's first feature: fetched together with the level-2 operands, not after them
        const int fl = tid/LPF, tp = tid % LPF;
        int f = f0 + fl, raw = 0; double fu = 0.0, fv = 0.0, refv[TPL];
#pragma unroll
        for (int k = 0; k < TPL; k++) refv[k] = 0.0;
        if (f1 > f0) {
            const int fc = min(f, f1 - 1);
            raw = L.tfeat_raw[fc]; fu = L.tfeat_uv[2*fc]; fv = L.tfeat_uv[2*fc+1];
#pragma unroll
            for (int k = 0; k < TPL; k++) refv[k] = L.tfeat_ref[8*(size_t)fc + TPL*tp + k];
        }
        PairT T;
        // poses / plane / image pointer do not wait for the activity test (h is known from the record)
        Pose C; load_pose(pose + 7*i, C);
        if (h >= 0) { Pose Hs; load_pose(pose + 7*h, Hs); pair_from_poses(C, Hs, T); }
        else pair_from_Twr(C, W.text_Twr + 12*(size_t)j, T);
        const double th[3] = { theta[3*j], theta[3*j+1], theta[3*j+2] };
        const uint8_t *img = L.img[i];
        const double inv_sigma = 1.0/sigma;
        const double ifx = 1.0/L.K[0], ify = 1.0/L.K[1];
        double tot = 0.0;                       // lane l < 55 of wave 0: running total of value l
        // chunks of 64 features (one chunk unless the plane has more).  One accumulator set only: the weighted block of the
        // thread's half feature is reduced per chunk, so that it stays inside the architectural VGPRs
        for (int fb = f0; fb == f0 || fb < f1; fb += 64, f += 64) {
            double blk[55];
#pragma unroll
            for (int k = 0; k < 55; k++) blk[k] = 0.0;
            if (act_g) {                                               // (uniform; the shuffle below needs both lanes of a feature)
                const bool in = f < f1;
                if (fb != f0 && in) {
                    raw = L.tfeat_raw[f]; fu = L.tfeat_uv[2*f]; fv = L.tfeat_uv[2*f+1];
#pragma unroll
                    for (int k = 0; k < TPL; k++) refv[k] = L.tfeat_ref[8*(size_t)f + TPL*tp + k];
                }
                const uint8_t good = in ? (W.filter_good ? W.tfgood[fg + raw] : (uint8_t)1) : (uint8_t)0;   // in flight with the pixel fetches
                // the pixel-pair fetches of all the thread's taps in flight before the first residual; the quads wait in LDS so that
                // the residual loop can stay rolled (unrolled, its live state does not fit 256 VGPRs and spills to scratch)
#pragma unroll
                for (int k = 0; k < TPL; k++) {
                    const int kt = TPL*tp + k;
                    const double mx = (fu + TAP_DX[kt] - L.K[2])*ifx, my = (fv + TAP_DY[kt] - L.K[3])*ify;   // tool.cc:1561
                    const TapPx q = tap_fetch(T, C.t, th, mx, my, L.K[0], L.K[1], L.K[2], L.K[3], img, L.img_w, L.img_h);
                    s_px[k*LIN_T + tid] = (unsigned)q.I00 | ((unsigned)q.I01 << 8) | ((unsigned)q.I10 << 16) | ((unsigned)q.I11 << 24);
                }
                LIN_STAMP(1);                                   // (operands there, the taps' pixel quads requested)
                double s = 0.0;
#pragma unroll 1
                for (int k = 0; k < TPL; k++) {
                    const int kt = TPL*tp + k;
                    const double mx = (fu + TAP_DX[kt] - L.K[2])*ifx, my = (fv + TAP_DY[kt] - L.K[3])*ify;
                    const unsigned q4 = s_px[k*LIN_T + tid];
                    const TapPx pxk = { (int)(q4 & 0xff), (int)((q4 >> 8) & 0xff), (int)((q4 >> 16) & 0xff), (int)(q4 >> 24) };
                    double rf = refv[0];
#pragma unroll
                    for (int q = 1; q < TPL; q++) if (k == q) rf = refv[q];
                    double jt[6], jl[3];
                    double r = text_tap_px(T, C.t, th, mx, my, L.K[0], L.K[1], L.K[2], L.K[3], pxk, L.img_w, L.img_h,
                                           mu, sigma, inv_sigma, rf, W.w_t, true, jt, jl);
                    s += r*r;
                    int q = 0;
#pragma unroll
                    for (int a = 0; a < 6; a++)
#pragma unroll
                        for (int cc = a; cc < 6; cc++) { blk[q] += jt[a]*jt[cc]; q++; }
#pragma unroll
                    for (int a = 0; a < 6; a++) blk[21 + a] += jt[a]*r;
#pragma unroll
                    for (int a = 0; a < 6; a++)
#pragma unroll
                        for (int cc = 0; cc < 3; cc++) blk[27 + a*3 + cc] += jt[a]*jl[cc];
                    blk[45] += jl[0]*jl[0]; blk[46] += jl[0]*jl[1]; blk[47] += jl[0]*jl[2];
                    blk[48] += jl[1]*jl[1]; blk[49] += jl[1]*jl[2]; blk[50] += jl[2]*jl[2];
                    blk[51] += jl[0]*r; blk[52] += jl[1]*r; blk[53] += jl[2]*r;
                }
                double s8 = s;                                      // the block's squared norm: its 8 taps sit on LPF neighbouring lanes
#pragma unroll
                for (int q = 1; q < LPF; q <<= 1) s8 += __shfl_xor(s8, q, 64);
                double wgt; const double rho_h = 0.5*huber(s8, W.huber_t, wgt);
                const double wg = good ? wgt : 0.0;
#pragma unroll
                for (int k = 0; k < 54; k++) blk[k] *= wg;
                blk[54] = (good && tp == 0) ? rho_h : 0.0;
            }
            LIN_STAMP(2);                                       // (the thread's taps evaluated and weighted)
            // 55 sums over the workgroup's threads: per wave two transposes (28 + 27 values), then the waves (fixed order)
            const double t0 = wave_sum_to_lane_mw<28>(blk, reg, lane);
            const double t1 = wave_sum_to_lane_mw<27>(blk + 28, reg, lane);
            if (lane < 28) xw[wave*64 + lane] = t0;
            if (lane < 27) xw[wave*64 + 28 + lane] = t1;
            __syncthreads();
            if (lane < 55) { double part = xw[lane];
#pragma unroll
                for (int q = 1; q < NWV; q++) part += xw[q*64 + lane];
                tot += part; }
            __syncthreads();
        }
        if (wave > 0) return;
        LIN_STAMP(3);
#ifdef MID_STAMPS
        if (threadIdx.x == 0 && spec) atomicAdd((unsigned long long *)&W.dbg[8*ls_kind + 7], 1ull);
#endif
        // wave 0 alone from here (LDS accesses of one wave are ordered; the fence keeps the compiler honest)
        if (lane < 27) B.tgM[(size_t)lane*L.n_tg + tgpp] = tot;          // pair-major rank: k_mid sums a contiguous range
        else if (lane < 45) { if (slot >= 0) B.w_tx[(size_t)(slot)*TX_REC + (lane - 27)] = tot; }
        else if (lane < 54) { if (slot >= 0) B.w_tx[(size_t)(slot)*TX_REC + 18 + (lane - 45)] = tot; }
        else if (lane == 54) B.tgCost[tgpp] = tot;                       // pair-major rank as well: k_mid adds it to its pair's cost
        // (the host column of W, -blkdiag(R,R)^T W, is formed by k_mid from W and the pair's R_cr; an inactive group leaves W = 0)
    }
}

template <int MODE, int PPW = 1, bool TEXT = true>
__global__ __launch_bounds__(LIN_T, TEXT ? 2 : 3) void k_linearize(Work W, LevelDev L, int spec) { lin_body<MODE, PPW, TEXT>(W, L, spec); }

// ---- per landmark: V, b, host column of W (= -sum Q^T w);  per pair: host-side products.  256-thread blocks.
__device__ __forceinline__ double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }
// NT threads per block (256; windows of the k_lin_mid experiment: 128 -- MID_TW -- so that a block is what a workgroup of k_linearize can take over): block b takes
// NT points / NT/8 planes / NT/4 pairs and leaves one partial (gradient max, |x|^2, cost) in B.lmpart.  clear_next: the block also zeroes the next trial's failure flag.
// Round 6 (stamps: tools/mid_stamps.sh, docs/ledger_r06.md): the kernel was the sum of its dependent round trips -- a point with more than four observers took
// three after its offsets (records | the fifth slot's pair | its record), a plane six (pair, record: two slots at a time), a pair one per text group -- 21.8 k
// cycles for the pair blocks, 20.4 k for the plane blocks, 16.1 k for the point blocks of C4.  Now every kind has TWO: the static offsets (with the LM state),
// then everything else at once -- U = 6 slot records of a point in flight (windows; maps keep 4: there the kernel is a throughput kernel and registers are
// occupancy), a plane on MID_PL = 8 lanes with one slot record each, a pair on MID_PR = 4 lanes that share its text groups; the lanes' sums meet by xor shuffles
// (a fixed order), the block's three partials in one reduction.
#define MID_PL 8
#define MID_PR_MIN 4
__device__ __forceinline__ void mid_lds_fence() {           // order this wave's LDS writes before its following LDS reads (one-wave blocks: no barrier needed)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
#define PT_PAIRN 6                          // entries of L.pt_pair4 per point
// MID_PR lanes per pair (4.  16 was measured on C4 -- ~90 pairs at a level, up to ~20 text groups each --: 8.65 against 7.63 us: two more shuffle steps over 27 values and
// stores predicated sixteen ways cost a lone wave more than the group loop's extra round trip; tools/mid_stamps.sh)
template <int NT, int U, int MID_PR>
__device__ __forceinline__ void mid_block(const Work &W, const LevelDev &L, const int nb_pt, const int nb_tx, const int spec, const int b, const bool clear_next, double *red /* [NT] */) {
    static_assert(U <= PT_PAIRN && NT % 64 == 0, "k_mid: the static pair list holds PT_PAIRN entries per point");
    const LmState *st = W.st;
#ifdef MID_STAMPS                           // (make-time experiment, tools/mid_stamps.sh: cycles a block of each kind spends in the kernel, summed over blocks and launches into W.dbg)
    const long long ms_t0 = clock64(); const int ms_kind = b < nb_pt ? 0 : b < nb_pt + nb_tx ? 1 : 2;
#define MID_STAMP(slot) do { if (threadIdx.x == 0 && spec) atomicAdd((unsigned long long *)&W.dbg[16 + 8*ms_kind + (slot)], (unsigned long long)(clock64() - ms_t0)); } while (0)
#else
#define MID_STAMP(slot) do { } while (0)
#endif
    const int tid = threadIdx.x;
    // static offsets of this thread's landmark / pair first: in flight together with the LM state
    int o = 0, e = 0, tq0 = 0, tq1 = 0, ph_ = -1, hp_ = -1, act_ = 0, pr0[U], prt = 0;
    double sg0 = 1.0, sgt[3] = {1.0, 1.0, 1.0};             // the Jacobi scales as they are (a first linearisation replaces them below)
#pragma unroll
    for (int u = 0; u < U; u++) pr0[u] = 0;
    // One-wave blocks (windows, NT == 64) give a plane 32 lanes and a pair the whole wave, a lane per VALUE (the 27 of a plane / a pair, then a lane per output):
    // no cross-lane sums, no predicated stores -- see the two W1 branches below.  Wider blocks (maps: throughput kernels) keep a plane on 8 and a pair on 4 lanes.
    constexpr bool W1 = NT == 64;
    const int jt = W1 ? (b - nb_pt)*2 + (tid >> 5) : (b - nb_pt)*(NT/MID_PL) + (tid >> 3), ut = W1 ? (tid & 31) : (tid & (MID_PL - 1));            // plane blocks: plane, lane
    const int pp = W1 ? b - nb_pt - nb_tx : (b - nb_pt - nb_tx)*(NT/MID_PR) + tid/MID_PR, up = W1 ? tid : (tid & (MID_PR - 1));    // pair blocks: pair, lane
    int prs[6] = {0, 0, 0, 0, 0, 0};
    if (b < nb_pt) { const int j = b*NT + tid; if (j < W.n_pt) { o = L.pls_off[j]; e = L.pls_off[j+1]; act_ = W.act_pt[j]; sg0 = W.sig_pt[j];
#pragma unroll
        for (int u = 0; u < U; u++) pr0[u] = L.pt_pair4[PT_PAIRN*(size_t)j + u]; } }
    else if (b < nb_pt + nb_tx) { if (jt < W.n_text) { o = L.tls_off[jt]; e = L.tls_off[jt+1]; act_ = W.act_tx[jt];
        if constexpr (W1) {
#pragma unroll
            for (int u = 0; u < 6; u++) prs[u] = L.tx_pair8[MID_PL*(size_t)jt + u];
            if (ut == 18 || ut == 21 || ut == 23) sgt[0] = W.sig_tx[(size_t)(ut == 18 ? 0 : ut == 21 ? 1 : 2)*W.n_text + jt];
        } else { prt = L.tx_pair8[MID_PL*(size_t)jt + ut];
#pragma unroll
        for (int k = 0; k < 3; k++) sgt[k] = W.sig_tx[(size_t)k*W.n_text + jt]; } } }
    else { if (pp < L.n_pair) { tq0 = L.pair_tg_off[pp]; tq1 = L.pair_tg_off[pp+1]; ph_ = L.pair_h[pp]; hp_ = L.pair_hpos[pp]; } }
    if (clear_next && W.st_next && b == 0 && tid == 0) W.st_next->step_fail = 0;      // (the next trial's k_schur_t takes this trial's decision into that copy of the state, every field but this one: its own workgroups may raise it)
    if (st->done) return;
    if (!spec && !st->need_lin) return;
    if (spec && st->step_fail) return;
    const LinBuf &B = W.lb[spec ? (st->lcur ^ 1) : st->lcur];
    const int sel = spec ? (st->cur ^ 1) : st->cur;
    const double *rho_x = W.rho[sel], *theta_x = W.theta[sel];
    const bool first = st->first != 0;
    double gm = 0.0, xn = 0.0, cs = 0.0;                        // cs: cost of this thread's pair and of its text groups
    MID_STAMP(0);                                               // (the state and the static offsets are here)
    if (b < nb_pt) {
        const int j = b*NT + tid;
        if (e > o) {
            double acc[8] = {0,0,0,0,0,0,0,0};                       // V, b, host column -sum Q^T w
            for (int s0 = o; s0 < e - 1; s0 += U) {                  // U slot records (and their pairs' R_cr) in flight per round trip
                int pr[U]; double v[U][8], R[U][9];
#pragma unroll
                for (int u = 0; u < U; u++) pr[u] = s0 == o ? pr0[u] : L.pslot_pair[min(s0 + u, e - 2)];
#pragma unroll
                for (int u = 0; u < U; u++) {
#pragma unroll
                    for (int k = 0; k < 8; k++) v[u][k] = B.w_pt[(size_t)(min(s0 + u, e - 2))*PT_REC + k];
#pragma unroll
                    for (int k = 0; k < 9; k++) R[u][k] = PAIRR(B, pr[u], k, L.n_pair);
                }
#pragma unroll
                for (int u = 0; u < U; u++) if (s0 + u < e - 1) {
                    double qa[3], qc[3]; mat3T_vec(R[u], v[u], qa); mat3T_vec(R[u], v[u] + 3, qc);
                    acc[0] += v[u][6]; acc[1] += v[u][7];
#pragma unroll
                    for (int a = 0; a < 3; a++) { acc[2 + a] += -qa[a]; acc[5 + a] += -qc[a]; }
                }
            }
#pragma unroll
            for (int k = 0; k < 6; k++) B.w_pt[(size_t)(e - 1)*PT_REC + k] = acc[2 + k];
            const double V = acc[0];
            double sg = sg0;
            if (first) { sg = 1.0/(1.0 + sqrt(V)); W.sig_pt[j] = sg; }
            VDB_STORE(B, j, W.n_pt, V, clampd(sg*sg*V, W.min_diag, W.max_diag)/(sg*sg), acc[1]);
            if (act_) { gm = fabs(acc[1]); xn = rho_x[j]*rho_x[j]; }
        }
    } else if (W1 && b < nb_pt + nb_tx) {
        // a plane on 32 lanes, lane k < 27 = value k of its record (W 0..17 | V6 18..23 | b3 24..26): the V / b lanes add the slots' values up, a W lane (half, rr, cc)
        // forms its entry of the host column -blkdiag(R,R)^T W from three values and three rotation entries per slot -- six slots in flight, nothing crosses lanes
        const int j = jt, k = ut;
        if (e > o && k < 27) {
            const int hr_ = k/3, cc_ = k - 3*hr_, half_ = hr_/3, rr_ = hr_ - 3*half_;      // (W lanes)
            double acc = 0.0;
            for (int s0 = o; s0 < e - 1; s0 += 6) {
                double x[6][3], Rv[6][3];
#pragma unroll
                for (int u = 0; u < 6; u++) {
                    const int sc = min(s0 + u, e - 2);
                    const int pr = s0 == o ? prs[u] : L.tslot_pair[sc];
                    // (the same requests on every lane, the V / b lanes' at addresses they do not use: a request under a lane-dependent branch is waited for at the end
                    // of that branch -- two kinds of lane were two round trips)
#pragma unroll
                    for (int q = 0; q < 3; q++) { x[u][q] = B.w_tx[(size_t)sc*TX_REC + (k >= 18 ? k : (half_*3 + q)*3 + cc_)]; Rv[u][q] = PAIRR(B, pr, k >= 18 ? 0 : q*3 + rr_, L.n_pair); }
                }
#pragma unroll
                for (int u = 0; u < 6; u++) if (s0 + u < e - 1) acc += k >= 18 ? x[u][0] : -(Rv[u][0]*x[u][0] + Rv[u][1]*x[u][1] + Rv[u][2]*x[u][2]);
            }
            if (k < 18) B.w_tx[(size_t)(e - 1)*TX_REC + k] = acc;
            else if (k < 24) B.V_tx[(size_t)(k - 18)*W.n_text + j] = acc;
            else B.b_tx[(size_t)(k - 24)*W.n_text + j] = acc;
            if (k == 18 || k == 21 || k == 23) {                     // the diagonal of V: Jacobi scale and LM diagonal of the plane's three parameters
                const int kd = k == 18 ? 0 : k == 21 ? 1 : 2;
                double sg = sgt[0];
                if (first) { sg = 1.0/(1.0 + sqrt(acc)); W.sig_tx[(size_t)kd*W.n_text + j] = sg; }
                B.dgs_tx[(size_t)kd*W.n_text + j] = clampd(sg*sg*acc, W.min_diag, W.max_diag)/(sg*sg);
            }
            if (act_ && k >= 24) { gm = fabs(acc); const double tv = theta_x[3*j + (k - 24)]; xn = tv*tv; }
        }
    } else if (b < nb_pt + nb_tx) {
        const int j = jt;
        if (e > o) {                                                  // (uniform over a plane's eight lanes)
            double acc[27];                                          // V6, b3, host column -blkdiag(R,R)^T W (18): this lane's slots
#pragma unroll
            for (int k = 0; k < 27; k++) acc[k] = 0.0;
            for (int s0 = o; s0 < e - 1; s0 += MID_PL) {             // one slot record per lane and round trip (a plane seen from more than eight keyframes goes round again)
                const int sc = min(s0 + ut, e - 2);
                const int pr = s0 == o ? prt : L.tslot_pair[sc];
                double v[27], R[9];
#pragma unroll
                for (int k = 0; k < 27; k++) v[k] = B.w_tx[(size_t)sc*TX_REC + k];
#pragma unroll
                for (int k = 0; k < 9; k++) R[k] = PAIRR(B, pr, k, L.n_pair);
                if (s0 + ut < e - 1) {
#pragma unroll
                    for (int k = 0; k < 9; k++) acc[k] += v[18 + k];
#pragma unroll
                    for (int half = 0; half < 2; half++)
#pragma unroll
                        for (int rr = 0; rr < 3; rr++)
#pragma unroll
                            for (int cc = 0; cc < 3; cc++)
                                acc[9 + (half*3 + rr)*3 + cc] += -(R[0*3 + rr]*v[(half*3 + 0)*3 + cc] + R[1*3 + rr]*v[(half*3 + 1)*3 + cc] + R[2*3 + rr]*v[(half*3 + 2)*3 + cc]);
                }
            }
#pragma unroll
            for (int k = 0; k < 27; k++) {                           // the eight lanes' sums (lanes 8 q .. 8 q + 7 of a wave), fixed order
                acc[k] += __shfl_xor(acc[k], 1, 64); acc[k] += __shfl_xor(acc[k], 2, 64); acc[k] += __shfl_xor(acc[k], 4, 64);
            }
            if (ut == 0) {
#pragma unroll
                for (int k = 0; k < 18; k++) B.w_tx[(size_t)(e - 1)*TX_REC + k] = acc[9 + k];
#pragma unroll
                for (int k = 0; k < 6; k++) B.V_tx[(size_t)k*W.n_text + j] = acc[k];
#pragma unroll
                for (int k = 0; k < 3; k++) B.b_tx[(size_t)k*W.n_text + j] = acc[6 + k];
                const double dv[3] = { acc[0], acc[3], acc[5] };
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    double sg = sgt[k];
                    if (first) { sg = 1.0/(1.0 + sqrt(dv[k])); W.sig_tx[(size_t)k*W.n_text + j] = sg; }
                    B.dgs_tx[(size_t)k*W.n_text + j] = clampd(sg*sg*dv[k], W.min_diag, W.max_diag)/(sg*sg);
                }
                if (act_) for (int k = 0; k < 3; k++) { gm = fmax(gm, fabs(acc[6 + k])); xn += theta_x[3*j + k]*theta_x[3*j + k]; }
            }
        }
    } else if (W1) {
        // a pair on one wave: lane k < 27 = row k of its sums M (21) | c (6), lane 27 its cost -- the scene blocks' sums plus the pair's text groups IN ORDER, eight groups
        // in flight; then a lane per OUTPUT: 27 copies, 36 entries of M Q, 21 of Q^T M Q, 6 of Q^T c, the operands crossing through 80 doubles of LDS
        const int p = pp, k = tid;
        double *xs = red;                                           // [0..26] M | c, [32..40] R, [44..79] M Q
        if (p < L.n_pair) {
            const int ntg = L.n_tg;
            // every lane makes the same nine requests (lanes 28..63 at addresses whose values they do not use, lanes 32..40 the rotation in place of the scene sums):
            // a request under a lane-dependent branch is waited for at the end of that branch, and the sums and the rotation were two round trips
            const bool isM = k < 27, isS = k < 28, isR = k >= 32 && k < 41;
            const double *row = isM ? B.tgM + (size_t)k*ntg : B.tgCost;
            const double *a0 = isM ? B.pairM + (size_t)k*L.n_pair + p : (isR ? &PAIRR(B, p, k - 32, L.n_pair) : B.pairCost + p);
            double g8[8];
            const double m0 = *a0;
#pragma unroll
            for (int u = 0; u < 8; u++) g8[u] = ntg > 0 ? row[min(tq0 + u, ntg - 1)] : 0.0;
            double m = m0;
#pragma unroll
            for (int u = 0; u < 8; u++) m += (isS && tq0 + u < tq1) ? g8[u] : 0.0;
            for (int q0 = tq0 + 8; q0 < tq1; q0 += 8) {
#pragma unroll
                for (int u = 0; u < 8; u++) g8[u] = row[min(q0 + u, ntg - 1)];
#pragma unroll
                for (int u = 0; u < 8; u++) m += (isS && q0 + u < tq1) ? g8[u] : 0.0;
            }
            if (k == 27) cs = m;
            if (k < 27) { xs[k] = m; B.pairOut[(size_t)k*L.n_pair + p] = m; }
            if (isR) xs[k] = ph_ >= 0 ? m : 0.0;
            if (ph_ >= 0) {                                          // (uniform)
                double *out = B.pairOut;
                const double *Rs = xs + 32; double *MQs = xs + 44;
                mid_lds_fence();
                if (k < 36) {                                        // M * blkdiag(R,R), entry (r, half*3 + cc)
                    const int r = k/6, col = k - 6*r, half = col/3, cc = col - 3*half;
                    const double mq = xs[sym6(r, half*3)]*Rs[cc] + xs[sym6(r, half*3 + 1)]*Rs[3 + cc] + xs[sym6(r, half*3 + 2)]*Rs[6 + cc];
                    MQs[k] = mq; out[(size_t)(27 + k)*L.n_pair + p] = mq;
                } else if (k < 42) {                                 // Q^T c
                    const int e6 = k - 36, half = e6/3, a = e6 - 3*half;
                    const double *c = xs + 21 + 3*half;
                    out[(size_t)(84 + e6)*L.n_pair + hp_] = Rs[a]*c[0] + Rs[3 + a]*c[1] + Rs[6 + a]*c[2];
                }
                mid_lds_fence();
                if (k < 36) {                                        // Q^T (M Q): upper triangle, rows 63.. stored host-major
                    const int r = k/6, cc = k - 6*r, hr = r/3, rr = r - 3*hr;
                    if (cc >= r) out[(size_t)(63 + sym6(r, cc))*L.n_pair + hp_] = Rs[rr]*MQs[(hr*3)*6 + cc] + Rs[3 + rr]*MQs[(hr*3 + 1)*6 + cc] + Rs[6 + rr]*MQs[(hr*3 + 2)*6 + cc];
                }
                mid_lds_fence();                                    // (the block's partials go through the same LDS words below)
            }
        }
    } else {
        const int p = pp;
        if (p < L.n_pair) {
            double M[27];                                            // M (21) | c (6): lane 0 starts from the scene blocks' sums, every lane adds its share of the pair's text groups
            double R[9];
            // every load of the pair issued before the first use (a load under a lane-dependent branch is waited for at the end of that branch: the scene sums,
            // the rotation and each text group were a round trip each): the four lanes read the same scene sums and rotation, masked afterwards; a lane's first
            // two text groups (clamped, masked) ride along -- a pair with more than eight text groups goes round again
            const int ntg = L.n_tg, q1 = tq0 + up, q2 = q1 + MID_PR;
            double m0[27], g1[27], g2[27], c0, c1 = 0.0, c2 = 0.0;
#pragma unroll
            for (int k = 0; k < 27; k++) m0[k] = B.pairM[(size_t)k*L.n_pair + p];
            c0 = B.pairCost[p];
#pragma unroll
            for (int k = 0; k < 9; k++) R[k] = PAIRR(B, p, k, L.n_pair);
            if (ntg > 0) {                                           // (uniform)
                const int q1c = min(q1, ntg - 1), q2c = min(q2, ntg - 1);
#pragma unroll
                for (int k = 0; k < 27; k++) { g1[k] = B.tgM[(size_t)k*ntg + q1c]; g2[k] = B.tgM[(size_t)k*ntg + q2c]; }
                c1 = B.tgCost[q1c]; c2 = B.tgCost[q2c];
            } else {
#pragma unroll
                for (int k = 0; k < 27; k++) { g1[k] = 0.0; g2[k] = 0.0; }
            }
            const bool h1 = q1 < tq1, h2 = q2 < tq1;
#pragma unroll
            for (int k = 0; k < 27; k++) M[k] = ((up == 0 ? m0[k] : 0.0) + (h1 ? g1[k] : 0.0)) + (h2 ? g2[k] : 0.0);
            MID_STAMP(3);
            cs = ((up == 0 ? c0 : 0.0) + (h1 ? c1 : 0.0)) + (h2 ? c2 : 0.0);
            for (int q = q2 + MID_PR; q < tq1; q += MID_PR) {
#pragma unroll
                for (int k = 0; k < 27; k++) M[k] += B.tgM[(size_t)k*ntg + q];
                cs += B.tgCost[q];
            }
#pragma unroll
            for (int k = 0; k < 27; k++)
#pragma unroll
                for (int of = 1; of < MID_PR; of <<= 1) M[k] += __shfl_xor(M[k], of, 64);      // all lanes of the pair hold its sums
            double cs_pair = cs;
#pragma unroll
            for (int of = 1; of < MID_PR; of <<= 1) cs_pair += __shfl_xor(cs_pair, of, 64);
            cs = up == 0 ? cs_pair : 0.0;                       // (counted once per pair)
            MID_STAMP(4);
            const double *c = M + 21;
            double *out = B.pairOut;      // [90][n_pair]: M(21) c(6) MQ(36) by pair | QMQ(21) Qc(6) by host-major rank.  Every lane forms everything; lane u stores entries k = u mod MID_PR
#pragma unroll
            for (int k = 0; k < 27; k++) if ((k & (MID_PR - 1)) == up) out[(size_t)k*L.n_pair + p] = M[k];
            if (ph_ >= 0) {
                const int hp = hp_;                        // rows 63..89 are stored host-major
                double Mf[36];
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int cc = 0; cc < 6; cc++) Mf[r*6 + cc] = M[sym6(r, cc)];
                double MQ[36];                         // M * blkdiag(R,R)
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int half = 0; half < 2; half++)
#pragma unroll
                        for (int cc = 0; cc < 3; cc++)
                            MQ[r*6 + half*3 + cc] = Mf[r*6 + half*3]*R[cc] + Mf[r*6 + half*3 + 1]*R[3 + cc] + Mf[r*6 + half*3 + 2]*R[6 + cc];
#pragma unroll
                for (int k = 0; k < 36; k++) if ((k & (MID_PR - 1)) == up) out[(size_t)(27 + k)*L.n_pair + p] = MQ[k];
#pragma unroll
                for (int r = 0; r < 6; r++)
#pragma unroll
                    for (int cc = r; cc < 6; cc++) {
                        const int hr = r/3, rr = r % 3;
                        double v = R[0*3 + rr]*MQ[(hr*3 + 0)*6 + cc] + R[1*3 + rr]*MQ[(hr*3 + 1)*6 + cc] + R[2*3 + rr]*MQ[(hr*3 + 2)*6 + cc];
                        if ((sym6(r, cc) & (MID_PR - 1)) == up) out[(size_t)(63 + sym6(r, cc))*L.n_pair + hp] = v;
                    }
                MID_STAMP(5);
                double a[3], d[3]; mat3T_vec(R, c, a); mat3T_vec(R, c + 3, d);
                if (up == MID_PR - 1) { out[(size_t)84*L.n_pair + hp] = a[0]; out[(size_t)85*L.n_pair + hp] = a[1]; out[(size_t)86*L.n_pair + hp] = a[2]; }
                if (up == MID_PR - 2) { out[(size_t)87*L.n_pair + hp] = d[0]; out[(size_t)88*L.n_pair + hp] = d[1]; out[(size_t)89*L.n_pair + hp] = d[2]; }
            }
        }
    }
    MID_STAMP(1);                                               // (this thread's records summed and stored)
    // the block's partials in ONE reduction: waves by shuffles, the NT / 64 waves' results through LDS, fixed order (the cost as per-block partials: k_postlin /
    // k_decide add a few hundred numbers instead of walking 40 k pairs at 5000 keyframes)
#pragma unroll
    for (int of = 32; of > 0; of >>= 1) { gm = fmax(gm, __shfl_xor(gm, of, 64)); xn += __shfl_xor(xn, of, 64); cs += __shfl_xor(cs, of, 64); }
    if ((tid & 63) == 0) { red[3*(tid >> 6)] = gm; red[3*(tid >> 6) + 1] = xn; red[3*(tid >> 6) + 2] = cs; }
    __syncthreads();
    if (tid == 0) {
#pragma unroll
        for (int w = 1; w < NT/64; w++) { gm = fmax(gm, red[3*w]); xn += red[3*w + 1]; cs += red[3*w + 2]; }
        B.lmpart[3*b] = gm; B.lmpart[3*b + 1] = xn; B.lmpart[3*b + 2] = cs;
    }
    MID_STAMP(2);
#ifdef MID_STAMPS
    if (tid == 0 && spec) atomicAdd((unsigned long long *)&W.dbg[16 + 8*ms_kind + 7], 1ull);      // blocks counted
#endif
}
#define MID_TW LIN_T                         // (128 with LIN_TPL = 4)
template <int NT, int U, int PR>
__global__ __launch_bounds__(NT) void k_mid(Work W, LevelDev L, int nb_pt, int nb_tx, int spec) {
    __shared__ double red[NT < 80 ? 80 : NT];
    mid_block<NT, U, PR>(W, L, nb_pt, nb_tx, spec, (int)blockIdx.x, true, red);
}

// ---- windows: the speculative linearisation of an LM trial and k_mid in ONE launch.  k_mid as a launch of its own was 10.6 us per trial on C4: a launch, the
// state's round trip, the offsets' round trip, and only then its two rounds of records.  Here every workgroup of the linearisation takes a ticket when its
// outputs are stored (fence, then one atomic on a counter that only ever grows: `base` is its value before this launch -- the host keeps count --, so nothing is
// ever reset); the LAST nb_lm arrivals stay: each requests the static offsets of "its" k_mid block, waits until the counter says that all n workgroups have
// stored (they are its own launch's earlier finishers: running or done, whatever the dispatch order -- nobody waits for a workgroup that has not started),
// fences and runs the block.  Same block size as k_mid<MID_TW>, same sums in the same order: bit-identical to the two launches
// (tsba_debug_options.trial_launches = 1).  A wait that outlasts its bound (other work holding the last workgroups off the device for tens of ms) is
// counted and fails the step, as every give-up does.
#define LM_SPIN_MAX (1 << 16)
__global__ __launch_bounds__(LIN_T, 2) void k_lin_mid(Work W, LevelDev L, int nb_pt, int nb_tx, int nb_lm, unsigned long long *ticket, unsigned long long base) {
    static_assert(LIN_T == MID_TW, "a workgroup of the linearisation takes over a k_mid block");
    __shared__ double red[MID_TW]; __shared__ long long s_rank;
    const int tid = threadIdx.x;
    if (W.st_next && blockIdx.x == 0 && tid == 0) W.st_next->step_fail = 0;      // (what k_mid's block 0 does: the next trial's failure flag starts clear)
    lin_body<MODE_FULL, 1, true>(W, L, 1);
    __syncthreads();                                            // (every wave's stores are issued; a wave that left lin_body early is here too)
    if (tid == 0) { __threadfence(); s_rank = (long long)(atomicAdd(ticket, 1ull) - base); }
    __syncthreads();
    const long long n = gridDim.x, m = nb_lm < n ? nb_lm : n, first = s_rank - (n - m);
    if (first < 0) return;
    if (tid == 0) {
        unsigned long long v = __hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int spins = 0; v - base < (unsigned long long)n && spins < LM_SPIN_MAX; spins++) { __builtin_amdgcn_s_sleep(2); v = __hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        if (v - base < (unsigned long long)n) { atomicAdd(&ts_poll_giveups, 1u); if (W.st_next) W.st_next->step_fail = 1; }
    }
    __syncthreads();
    __threadfence();                                            // (the other workgroups' records)
    for (long long b = first; b < nb_lm; b += m) { mid_block<MID_TW, 6, MID_PR_MIN>(W, L, nb_pt, nb_tx, 1, (int)b, false, red); __syncthreads(); }
}

// ---- after a linearisation (256 threads of one block), in two stages so that a multi-GPU run can all-reduce in between:
//   sums_local : pose diagonal / gradient from the pair sums (-> B.Hd, B.bp), landmark gradient max / |x|^2, cost
//   pose_scale : Jacobi scaling, LM diagonal, gradient max and |x|^2 of the free poses (from the possibly all-reduced Hd / bp)
__device__ void sums_local(const Work &W, const LevelDev &L, const LinBuf &B, double *dHd, double *dbp, int nb_lm, double *red,
                           double &gmax_lm, double &xn_lm, double &cost, bool skip_pose = false) {
    const int tid = threadIdx.x;
    gmax_lm = 0.0; xn_lm = 0.0; cost = 0.0;
    const double *out = B.pairOut;
    const size_t np = L.n_pair;
    for (int task = tid; task < (skip_pose ? 0 : 12*W.n_kf); task += 256) {          // (pose, component): diag H (6) | b (6)   (large maps: k_pose_sums_raw did it)
        const int a = task/12, k = task - 12*a;
        const int t0 = L.pose_t_off[a], t1 = L.pose_t_off[a+1], h0 = L.pose_h_off[a], h1 = L.pose_h_off[a+1];
        if (k < 6) {
            const double h = range_sum<24>(out + (size_t)sym6(k, k)*np, t0, t1) + range_sum<24>(out + (size_t)(63 + sym6(k, k))*np, h0, h1);
            dHd[6*a + k] = h;
        } else {
            const double g = range_sum<24>(out + (size_t)(21 + k - 6)*np, t0, t1) - range_sum<24>(out + (size_t)(84 + k - 6)*np, h0, h1);
            dbp[6*a + k - 6] = g; B.bp_loc[6*a + k - 6] = g;
        }
    }
    __threadfence_block();                                             // pose_scale reads these through other threads
    for (int k = tid; k < nb_lm; k += 256) { gmax_lm = fmax(gmax_lm, B.lmpart[3*k]); xn_lm += B.lmpart[3*k + 1]; cost += B.lmpart[3*k + 2]; }
    gmax_lm = block_max<256>(gmax_lm, red); xn_lm = block_sum<256>(xn_lm, red); cost = block_sum<256>(cost, red);
}
__device__ void pose_scale(const Work &W, const LinBuf &B, const double *sHd, const double *sbp, const double *pose, bool first,
                           double *red, double &gmax_p, double &xn_p) {
    const int tid = threadIdx.x;
    gmax_p = 0.0; xn_p = 0.0;
    for (int a = tid; a < W.n_kf; a += 256) {
        const bool fre = W.fidx[a] >= 0;
#pragma unroll
        for (int k = 0; k < 6; k++) {
            const double h = sHd[6*a + k], g = sbp[6*a + k];
            B.Hd[6*a + k] = h; B.bp[6*a + k] = g;                 // (multi-GPU: the all-reduced values replace the local ones)
            if (first) W.sig_p[6*a + k] = 1.0/(1.0 + sqrt(h));
            const double sg = W.sig_p[6*a + k];
            B.dgs_p[6*a + k] = clampd(sg*sg*h, W.min_diag, W.max_diag)/(sg*sg);
            if (fre) gmax_p = fmax(gmax_p, fabs(g));
        }
        if (fre) for (int k = 0; k < 7; k++) xn_p += pose[7*a + k]*pose[7*a + k];
    }
    gmax_p = block_max<256>(gmax_p, red); xn_p = block_sum<256>(xn_p, red);
}
// Single-GPU path of k_postlin / k_decide: everything one linearisation contributes to the LM decision, with the independent
// loads of all parts issued before the first wait and ONE five-value block reduction (a global round trip from this lone
// workgroup costs ~0.6 us, a block reduction ~0.3 us: the old sequence had a dozen of the former and seven of the latter).
//   out5 = { max |gradient|, |x|^2, cost, step^2 (nb_back partials), model cost change (nb_back partials) }   (thread 0)
// write: this workgroup stores the poses' diagonal / gradient / damping rows (every workgroup of a launch that takes the decision redundantly computes
// them; one stores); keep (may be null): [2][6 n_kf] in LDS, the damping rows and the gradient rows for the caller's own use
// light (round 6): only the partials of cost / step / model cost change -- what the accept / reject decision and the trust region are made of; the poses' sums
// (gradient max and |x|^2: the tolerance exits, kept by the one workgroup that runs the full version) are skipped and out5[0], out5[1] are not meaningful
__device__ void postlin_fused(const Work &W, const LevelDev &L, const LinBuf &B, const double *pose, bool first, int nb_lm, int nb_back,
                              double *red /*[5*256]*/, double *xch /*[252]*/, double out5[5], int npp = 0, bool write = true, double *keep = nullptr, bool light = false) {
    const int tid = threadIdx.x;
    double gmax = 0.0, xn = 0.0, cost = 0.0, step2 = 0.0, mcc = 0.0;
#ifdef TSBA_SOLVE_STAMPS
    long long q0_ = clock64(), q1_ = 0, q2_ = 0, q3_ = 0, q4_ = 0;
#endif
    {   // landmark / cost / step partials (one entry per k_mid / k_back workgroup): three per thread in flight, the (rare) rest in a plain loop
        double lc[3], lg[3], lx[3], ps[3], pm[3];
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int k = tid + 256*u;
            lg[u] = B.lmpart[3*min(k, max(nb_lm - 1, 0))]; lx[u] = B.lmpart[3*min(k, max(nb_lm - 1, 0)) + 1]; lc[u] = B.lmpart[3*min(k, max(nb_lm - 1, 0)) + 2];
            ps[u] = W.partial[2*min(k, max(nb_back - 1, 0))]; pm[u] = W.partial[2*min(k, max(nb_back - 1, 0)) + 1];
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
            const int k = tid + 256*u;
            if (k < nb_lm) { gmax = fmax(gmax, lg[u]); xn += lx[u]; cost += lc[u]; }
            if (k < nb_back) { step2 += ps[u]; mcc += pm[u]; }
        }
        for (int k = tid + 768; k < nb_lm; k += 256) { gmax = fmax(gmax, B.lmpart[3*k]); xn += B.lmpart[3*k + 1]; cost += B.lmpart[3*k + 2]; }
        for (int k = tid + 768; k < nb_back; k += 256) { step2 += W.partial[2*k]; mcc += W.partial[2*k + 1]; }
    }
#ifdef TSBA_SOLVE_STAMPS
    q1_ = clock64();
#endif
    // poses, 21 per round: thread (pose, component) sums one entry of diag(H_pp) (6) or of the gradient (6) over the pose's
    // pairs -- target side by pair, host side host-major, both contiguous -- then the six diag threads finish the pose
    const double *out = B.pairOut; const size_t np = L.n_pair;
    // (large maps: k_pose_sums did the per-pose work on many workgroups; only its partials are left to add)
    for (int k = tid; k < npp; k += 256) { gmax = fmax(gmax, W.posepart[2*k]); xn += W.posepart[2*k + 1]; }
    for (int a0 = 0; a0 < (npp > 0 || light ? 0 : W.n_kf); a0 += 21) {
        const int al = tid/12, k = tid - 12*al, a = a0 + al;
        const bool on = tid < 252 && a < W.n_kf;
        const int ac = min(a, W.n_kf - 1);
        const int t0 = L.pose_t_off[ac], t1 = L.pose_t_off[ac+1], h0 = L.pose_h_off[ac], h1 = L.pose_h_off[ac+1];
        const int kk = k < 6 ? k : k - 6;
        const double sgp = first ? 0.0 : W.sig_p[6*ac + kk]; const int fre = W.fidx[ac];
        const double px = pose[7*ac + kk], px6 = pose[7*ac + 6];
        const double *rt = out + (size_t)(k < 6 ? sym6(kk, kk) : 21 + kk)*np, *rh = out + (size_t)(k < 6 ? 63 + sym6(kk, kk) : 84 + kk)*np;
        const double vt = range_sum<24>(rt, t0, t1), vh = range_sum<24>(rh, h0, h1);
        const double val = k < 6 ? vt + vh : vt - vh;
#ifdef TSBA_SOLVE_STAMPS
        q2_ = clock64();
#endif
        if (on) xch[tid] = val;
        __syncthreads();
        if (on && k < 6) {
            const double h = val, g = xch[tid + 6];
            if (write) { B.Hd[6*a + k] = h; B.bp[6*a + k] = g; B.bp_loc[6*a + k] = g; }
            double sg = sgp;
            if (first) { sg = 1.0/(1.0 + sqrt(h)); if (write) W.sig_p[6*a + k] = sg; }
            const double dgv = clampd(sg*sg*h, W.min_diag, W.max_diag)/(sg*sg);
            if (write) B.dgs_p[6*a + k] = dgv;
            if (keep) { keep[6*a + k] = dgv; keep[6*W.n_kf + 6*a + k] = g; }
            if (fre >= 0) { gmax = fmax(gmax, fabs(g)); xn += px*px + (k == 0 ? px6*px6 : 0.0); }
        }
        __syncthreads();
    }
#ifdef TSBA_SOLVE_STAMPS
    q3_ = clock64();
#endif
    // one reduction for the five values: wave w reduces value w, wave 0 also value 4
    red[tid] = gmax; red[256 + tid] = xn; red[512 + tid] = cost; red[768 + tid] = step2; red[1024 + tid] = mcc;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    auto reduce_one = [&](int v) -> double {
        const double *r = red + 256*v;
        double x;
        if (v == 0) {
            x = fmax(fmax(r[lane], r[lane + 64]), fmax(r[lane + 128], r[lane + 192]));
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o, 64));
        } else { x = (r[lane] + r[lane + 64]) + (r[lane + 128] + r[lane + 192]); x = wave_sum1(x); }
        return x;
    };
    const double x0 = reduce_one(wave), x4 = wave == 0 ? reduce_one(4) : 0.0;
    __syncthreads();
    if (lane == 0) { red[256*wave] = x0; if (wave == 0) red[1024] = x4; }
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 5; v++) out5[v] = red[256*v];
#ifdef TSBA_SOLVE_STAMPS
    q4_ = clock64();
    if (tid == 0) { W.dbg[40] = q1_ - q0_; W.dbg[41] = q2_ - q1_; W.dbg[42] = q3_ - q2_; W.dbg[43] = q4_ - q3_; }
#endif
}
// The pose part of postlin_fused for large maps (hundreds of keyframes and more): 21 poses per workgroup instead of 21 per ROUND of
// the single postlin / decide workgroup (238 rounds, 1.1 ms per LM iteration at 5000 keyframes).
__global__ __launch_bounds__(256) void k_pose_sums(Work W, LevelDev L, int spec) {
    const LmState *st = W.st;
    if (st->done) return;
    if (!spec && !st->need_lin) return;
    if (spec && st->step_fail) return;
    __shared__ double xch[256], red[256];
    const LinBuf &B = W.lb[spec ? (st->lcur ^ 1) : st->lcur];
    const double *pose = W.pose[spec ? (st->cur ^ 1) : st->cur];
    const bool first = !spec && st->first != 0;
    const int tid = threadIdx.x;
    const double *out = B.pairOut; const size_t np = L.n_pair;
    const int al = tid/12, k = tid - 12*al, a = blockIdx.x*21 + al;
    const bool on = tid < 252 && a < W.n_kf;
    const int ac = min(a, W.n_kf - 1);
    const int t0 = L.pose_t_off[ac], t1 = L.pose_t_off[ac+1], h0 = L.pose_h_off[ac], h1 = L.pose_h_off[ac+1];
    const int kk = k < 6 ? k : k - 6;
    const double sgp = first ? 0.0 : W.sig_p[6*ac + kk]; const int fre = W.fidx[ac];
    const double px = pose[7*ac + kk], px6 = pose[7*ac + 6];
    const double *rt = out + (size_t)(k < 6 ? sym6(kk, kk) : 21 + kk)*np, *rh = out + (size_t)(k < 6 ? 63 + sym6(kk, kk) : 84 + kk)*np;
    const double vt = range_sum<24>(rt, t0, t1), vh = range_sum<24>(rh, h0, h1);
    const double val = k < 6 ? vt + vh : vt - vh;
    if (on) xch[tid] = val;
    __syncthreads();
    double gmax = 0.0, xn = 0.0;
    if (on && k < 6) {
        const double h = val, g = xch[tid + 6];
        B.Hd[6*a + k] = h; B.bp[6*a + k] = g; B.bp_loc[6*a + k] = g;
        double sg = sgp;
        if (first) { sg = 1.0/(1.0 + sqrt(h)); W.sig_p[6*a + k] = sg; }
        B.dgs_p[6*a + k] = clampd(sg*sg*h, W.min_diag, W.max_diag)/(sg*sg);
        if (fre >= 0) { gmax = fabs(g); xn = px*px + (k == 0 ? px6*px6 : 0.0); }
    }
    gmax = block_max<256>(gmax, red); xn = block_sum<256>(xn, red);
    if (tid == 0) { W.posepart[2*blockIdx.x] = gmax; W.posepart[2*blockIdx.x + 1] = xn; }
}
// The same in a sharded (multi-GPU) run, in two stages around the all-reduce of the exchange buffer cb = [Hd | bp | scalars]:
//   k_pose_sums_raw    this rank's part of diag(H_pp) and of the pose gradient, 21 poses per workgroup  -> cb, B.bp_loc
//   k_pose_scale_multi from the all-reduced cb: B.Hd / B.bp, Jacobi scale (first linearisation), LM diagonal, per-workgroup partials of
//                      the gradient maximum and |x|^2 of the free poses -> W.posepart
// (one workgroup walking 5000 poses cost 0.7 ms per linearisation: more than everything the sharding saves)
__global__ __launch_bounds__(256) void k_pose_sums_raw(Work W, LevelDev L, int spec) {
    const LmState *st = W.st;
    if (st->done) return;
    if (!spec && !st->need_lin) return;
    if (spec && st->step_fail) return;
    const LinBuf &B = W.lb[spec ? (st->lcur ^ 1) : st->lcur];
    const int tid = threadIdx.x;
    const double *out = B.pairOut; const size_t np = L.n_pair;
    const int al = tid/12, k = tid - 12*al, a = blockIdx.x*21 + al;
    if (tid >= 252 || a >= W.n_kf) return;
    const int t0 = L.pose_t_off[a], t1 = L.pose_t_off[a+1], h0 = L.pose_h_off[a], h1 = L.pose_h_off[a+1];
    const int kk = k < 6 ? k : k - 6;
    const double *rt = out + (size_t)(k < 6 ? sym6(kk, kk) : 21 + kk)*np, *rh = out + (size_t)(k < 6 ? 63 + sym6(kk, kk) : 84 + kk)*np;
    const double vt = range_sum<24>(rt, t0, t1), vh = range_sum<24>(rh, h0, h1);
    if (k < 6) W.cb[6*a + kk] = vt + vh;
    else { const double g = vt - vh; W.cb[W.N + 6*a + kk] = g; B.bp_loc[6*a + kk] = g; }
}
__global__ __launch_bounds__(256) void k_pose_scale_multi(Work W, int spec) {
    const LmState *st = W.st;
    if (st->done) return;
    if (!spec && !st->need_lin) return;
    if (spec && st->step_fail) return;
    __shared__ double red[256];
    const LinBuf &B = W.lb[spec ? (st->lcur ^ 1) : st->lcur];
    const double *pose = W.pose[spec ? (st->cur ^ 1) : st->cur];
    const bool first = !spec && st->first != 0;
    const int tid = threadIdx.x, al = tid/6, k = tid - 6*al, a = blockIdx.x*21 + al;
    double gmax = 0.0, xn = 0.0;
    if (tid < 126 && a < W.n_kf) {
        const double h = W.cb[6*a + k], g = W.cb[W.N + 6*a + k];
        B.Hd[6*a + k] = h; B.bp[6*a + k] = g;
        double sg;
        if (first) { sg = 1.0/(1.0 + sqrt(h)); W.sig_p[6*a + k] = sg; } else sg = W.sig_p[6*a + k];
        B.dgs_p[6*a + k] = clampd(sg*sg*h, W.min_diag, W.max_diag)/(sg*sg);
        if (W.fidx[a] >= 0) { gmax = fabs(g); const double px = pose[7*a + k]; xn = px*px; if (k == 0) { const double p6 = pose[7*a + 6]; xn += p6*p6; } }
    }
    gmax = block_max<256>(gmax, red); xn = block_sum<256>(xn, red);
    if (tid == 0) { W.posepart[2*blockIdx.x] = gmax; W.posepart[2*blockIdx.x + 1] = xn; }
}
// the pose part of k_postlin / k_decide in a sharded run on a large map: the partials k_pose_scale_multi left
__device__ void pose_parts_multi(const Work &W, int npp, double *red, double &gmax_p, double &xn_p) {
    gmax_p = 0.0; xn_p = 0.0;
    for (int k = threadIdx.x; k < npp; k += 256) { gmax_p = fmax(gmax_p, W.posepart[2*k]); xn_p += W.posepart[2*k + 1]; }
    gmax_p = block_max<256>(gmax_p, red); xn_p = block_sum<256>(xn_p, red);
}
__global__ __launch_bounds__(256) void k_postlin(Work W, LevelDev L, double grad_tol, int nb_lm, int multi, int npp) {
    LmState *st = W.st;
    if (W.dp_poll) for (int k = threadIdx.x; k <= W.N; k += 256) W.dp[k] = __builtin_nan("");       // (k_solve_back: "not there yet" for the blocks that poll the step)
    if (W.st_next && threadIdx.x == 0) W.st_next->step_fail = 0;
    if (st->done || !st->need_lin) return;
    __shared__ double red[5*256], xch[256];
    double gmax, xn, cost;
    const LinBuf &B = W.lb[st->lcur];
    if (!multi) { double o5[5]; postlin_fused(W, L, B, W.pose[st->cur], st->first != 0, nb_lm, 0, red, xch, o5, npp); gmax = o5[0]; xn = o5[1]; cost = o5[2]; }
    else { double gp, xp;
           if (npp) pose_parts_multi(W, npp, red, gp, xp); else pose_scale(W, B, W.cb, W.cb + W.N, W.pose[st->cur], st->first != 0, red, gp, xp);
           const double *sc = W.cb + 2*(size_t)W.N; cost = sc[0]; xn = sc[1] + xp; gmax = fmax(W.cbm[0], gp); }
    if (threadIdx.x == 0) {
        st->x_cost = cost; st->x_norm = sqrt(xn); st->gmax = gmax;
        if (st->first) st->cost0 = cost;
        st->first = 0; st->need_lin = 0; st->n_lin++;
        if (gmax <= grad_tol) { st->done = 1; st->term = 3; }
        if (W.hprog) { *W.hprog = ((unsigned long long)W.pass_seq << 32) | ((unsigned long long)st->it << 1) | (st->done ? 1u : 0u); __threadfence_system(); }
    }
}

