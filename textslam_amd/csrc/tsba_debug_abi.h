// Debug / timing / test-hook entry points of the C ABI (inside extern "C").  (part of the single translation unit tsba.hip: included there, in this order)
#pragma once
// debug / test aid: first linearisation of pass 0 + reduced system for `radius`; copies S (N x N), g (N), cost, kf flags
int tsba_debug_reduced_system(void *ctx, double radius, double *S, double *g, double *cost, int32_t *kf_free, double *dp) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    if (!c->uploaded) return TSBA_ERR_STATE;
    hipSetDevice(c->device);
    int rc = reset_state(c); if (rc) return rc;
    tsba_options saved = c->opt; c->opt.initial_radius = radius;
    const LevelDev &D = c->lev[c->opt.levels[0]];
    { int rca = set_solver_attrs(c); if (rca) return rca; }
    launch_pass_init(c, D, 0);
    launch_linearize(c, D, 0);
    Work &W = c->W;
    if ((int64_t)D.n_sb < (int64_t)c->n_kf*(c->n_kf + 1)/2) { hipMemsetAsync(c->S_alloc, 0, sizeof(double)*c->S_count, c->stream); c->S_stale = false;
        if (D.far_B > 0) hipMemsetAsync(W.Sfar, 0, sizeof(double)*36*(size_t)std::max(D.n_far, 1), c->stream); }
    // split (multi-GPU) sequence: this shard's PARTIAL S and g, before any exchange and without the pose damping (which is added
    // once after the all-reduce) -- the parts of all shards sum to the unsharded system; dp is not computed
    launch_schur(c, D, (int)is_multi(c));
    if (!is_multi(c)) launch_solve_full(c, D); else hipMemsetAsync(W.dp, 0, sizeof(double)*W.N, c->stream);
    c->opt = saved;
    CK(hipStreamSynchronize(c->stream)); CK(hipGetLastError());
    if (S) {
        if (!W.band) CK(hipMemcpy(S, W.S, sizeof(double)*(size_t)W.N*W.N, hipMemcpyDeviceToHost));
        else { std::vector<double> hb(c->S_count); CK(hipMemcpy(hb.data(), c->S_alloc, sizeof(double)*c->S_count, hipMemcpyDeviceToHost));
            const long long N = W.N, LDB = W.ldS + 1, Wb = LDB - c->S_up;                // band -> dense (entries outside the band are zero)
            for (long long i = 0; i < N; i++) for (long long j = 0; j < N; j++)
                S[i*N + j] = (j >= i - Wb && j <= i + c->S_up - 1) ? hb[(size_t)(Wb + i*(LDB - 1) + j)] : 0.0;
            if (W.ring) {                     // the loop-closure blocks: ghost row 6 nfree + r stands for row r of the first poses (lower triangle: (late pose, early pose))
                int nfr[2] = {0, 0}; CK(hipMemcpy(nfr, W.nfree, 2*sizeof(int), hipMemcpyDeviceToHost));
                const long long n6 = 6LL*nfr[0], r06 = 6LL*nfr[1], ng = std::min<long long>(6LL*W.ring_b, N);
                for (long long r = 0; r < ng && n6 + r < (long long)(c->S_count/LDB); r++) for (long long j = std::max(0LL, n6 + r - Wb); j < n6; j++) {
                    const double v = hb[(size_t)(Wb + (n6 + r)*(LDB - 1) + j)]; if (v != 0.0 && j > r06 + r) S[j*N + r06 + r] = v; }
            }
            if (D.far_B > 0 && D.n_far > 0) {        // the blocks outside the band (lower triangle: rows of the later keyframe)
                const HostPlan &H = c->hplan[D.level];
                std::vector<double> hf(36*(size_t)D.n_far); std::vector<int> fi(c->n_kf);
                CK(hipMemcpy(hf.data(), W.Sfar, sizeof(double)*hf.size(), hipMemcpyDeviceToHost)); CK(hipMemcpy(fi.data(), W.fidx, sizeof(int)*c->n_kf, hipMemcpyDeviceToHost));
                for (int q = 0; q < D.n_far; q++) { const long long ia = fi[H.far_a[q]], ic = fi[H.far_b[q]]; if (ia < 0 || ic < 0) continue;
                    for (int r = 0; r < 6; r++) for (int cc = 0; cc < 6; cc++) S[(6*ic + cc)*N + 6*ia + r] = hf[36*(size_t)q + 6*r + cc]; }
            } }
    }
    if (g) CK(hipMemcpy(g, W.g, sizeof(double)*W.N, hipMemcpyDeviceToHost));
    if (dp) CK(hipMemcpy(dp, W.dp, sizeof(double)*W.N, hipMemcpyDeviceToHost));
    LmState st; CK(hipMemcpy(&st, W.st, sizeof(st), hipMemcpyDeviceToHost));
    if (cost) *cost = st.x_cost;
    if (kf_free) { std::vector<int> in(c->n_kf), cs(c->n_kf);
        CK(hipMemcpy(in.data(), W.kf_in, sizeof(int)*c->n_kf, hipMemcpyDeviceToHost)); CK(hipMemcpy(cs.data(), W.kf_const, sizeof(int)*c->n_kf, hipMemcpyDeviceToHost));
        for (int k = 0; k < c->n_kf; k++) kf_free[k] = in[k] && !cs[k]; }
    return TSBA_OK;
}

// The same for LARGE maps, where the dense (6 n_kf)^2 copy is not an option (7.2 GB at 5000 keyframes): the band of the
// compressed (free-pose) system in LAPACK lower-band storage, ab[(i - j)*n + j] = S(i, j) for j <= i <= j + bw -- what
// scipy.linalg.solveh_banded(lower=True) takes.  Call once with ab = NULL to get n (rows) and bw, then with buffers.
int tsba_debug_reduced_band(void *ctx, double radius, int32_t *n_out, int32_t *bw_out, double *ab, double *g, double *dp) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    if (!c->uploaded) return TSBA_ERR_STATE;
    if (!c->W.band) { set_err(c, "the uploaded problem keeps a dense reduced system: use tsba_debug_reduced_system"); return TSBA_ERR_STATE; }
    if (c->W.ring) { set_err(c, "ring-shaped map: the loop-closure blocks live in ghost rows outside the band (tsba_debug_set no_ring for the reordered band)"); return TSBA_ERR_STATE; }
    int rc = tsba_debug_reduced_system(ctx, radius, nullptr, nullptr, nullptr, nullptr, nullptr); if (rc) return rc;
    Work &W = c->W;
    int nfree = 0; CK(hipMemcpy(&nfree, W.nfree, sizeof(int), hipMemcpyDeviceToHost));
    const long long n = 6LL*nfree, LDB = W.ldS + 1, Wb = LDB - c->S_up;
    const int bw = std::max(6, c->lev[c->opt.levels[0]].bw_rows) + 5;              // rows below the diagonal that can be non-zero (block-aligned band)
    if (n_out) *n_out = (int32_t)n; if (bw_out) *bw_out = bw;
    if (ab) {
        std::vector<double> hb(c->S_count); CK(hipMemcpy(hb.data(), c->S_alloc, sizeof(double)*c->S_count, hipMemcpyDeviceToHost));
        for (long long d = 0; d <= bw; d++) for (long long j = 0; j < n; j++) { const long long i = j + d;
            ab[d*n + j] = (i < n && j >= i - Wb) ? hb[(size_t)(Wb + i*(LDB - 1) + j)] : 0.0; }
    }
    if (g) CK(hipMemcpy(g, W.g, sizeof(double)*n, hipMemcpyDeviceToHost));
    if (dp) CK(hipMemcpy(dp, W.dp, sizeof(double)*W.N, hipMemcpyDeviceToHost));
    return TSBA_OK;
}


// The reduced system of the first linearisation as 6x6 blocks keyed by KEYFRAME pairs, whatever the storage behind it (band rows in any keyframe
// order, the ghost rows of a ring map, the blocks outside the band of a map with long-range coupling): block q couples keyframes kf_r[q], kf_c[q]
// and holds S(rows of kf_r, columns of kf_c), row-major.  A keyframe pair may appear more than once (a block of E inside the band): the parts add.
// First call with kf_r == NULL: *nblk = number of blocks (the system is assembled, solved and downloaded then and kept until the second call).
// g_kf, dp_kf [6 n_kf] by keyframe (0 for constant poses), cost = the cost at the linearisation point.  Band storage only.
int tsba_debug_reduced_blocks(void *ctx, double radius, int32_t *nblk, int32_t *kf_r, int32_t *kf_c, double *val, double *g_kf, double *dp_kf, double *cost) {
    Ctx *c = (Ctx *)ctx; if (!c || !nblk) return TSBA_ERR_ARG;
    if (!c->uploaded) return TSBA_ERR_STATE;
    if (!c->W.band) { set_err(c, "the uploaded problem keeps a dense reduced system: use tsba_debug_reduced_system"); return TSBA_ERR_STATE; }
    if (!kf_r) {
        int rc = tsba_debug_reduced_system(ctx, radius, nullptr, nullptr, nullptr, nullptr, nullptr); if (rc) return rc;
        Work &W = c->W; const LevelDev &D = c->lev[c->opt.levels[0]];
        c->rb_r.clear(); c->rb_c.clear(); c->rb_v.clear();
        int nfr[2] = {0, 0}; CK(hipMemcpy(nfr, W.nfree, 2*sizeof(int), hipMemcpyDeviceToHost));
        std::vector<int> fi(c->n_kf), kf_of(std::max(1, nfr[0]), -1);
        CK(hipMemcpy(fi.data(), W.fidx, sizeof(int)*c->n_kf, hipMemcpyDeviceToHost));
        for (int k = 0; k < c->n_kf; k++) if (fi[k] >= 0 && fi[k] < nfr[0]) kf_of[fi[k]] = k;
        std::vector<double> hb(c->S_count); CK(hipMemcpy(hb.data(), c->S_alloc, sizeof(double)*c->S_count, hipMemcpyDeviceToHost));
        const long long LDB = W.ldS + 1, Wb = LDB - c->S_up, nrows = (long long)(c->S_count/LDB);
        auto at = [&](long long i, long long j) { return hb[(size_t)(Wb + i*(LDB - 1) + j)]; };
        auto emit = [&](int kr, int kc, const double *b) { c->rb_r.push_back(kr); c->rb_c.push_back(kc); c->rb_v.insert(c->rb_v.end(), b, b + 36); };
        const int wbb = (int)(Wb/6);                                   // block columns left of the diagonal block a row block can reach
        for (int rb = 0; rb < nfr[0]; rb++) for (int cb = std::max(0, rb - wbb); cb <= rb; cb++) {
            double b[36]; bool nz = false;
            for (int r = 0; r < 6; r++) for (int q = 0; q < 6; q++) { const long long i = 6LL*rb + r, j = 6LL*cb + q;
                const double v = (j <= i && j >= i - Wb) ? at(i, j) : (j > i && cb == rb ? at(j, i) : 0.0); b[6*r + q] = v; nz |= v != 0.0; }
            if (nz) emit(kf_of[rb], kf_of[cb], b);
        }
        if (W.ring) {                                                  // ghost row 6 nfree + r = row r of the loop's first poses: the closure blocks (late pose, early pose)
            const long long n6 = 6LL*nfr[0]; const int eb0 = nfr[1];
            for (int g = 0; g < W.ring_b && n6 + 6LL*g + 5 < nrows; g++) for (int cb = 0; cb < nfr[0]; cb++) {
                if (cb <= eb0 + g) continue;
                double b[36]; bool nz = false;
                for (int r = 0; r < 6; r++) for (int q = 0; q < 6; q++) { const long long i = n6 + 6LL*g + r, j = 6LL*cb + q;     // S(j, first + r): rows of the late pose cb
                    const double v = (j >= i - Wb && j >= 0) ? at(i, j) : 0.0; b[6*q + r] = v; nz |= v != 0.0; }
                if (nz) emit(kf_of[cb], kf_of[eb0 + g], b);
            }
        }
        if (D.far_B > 0 && D.n_far > 0) {                              // the blocks outside the band: stored with rows = the earlier keyframe a
            const HostPlan &H = c->hplan[D.level];
            std::vector<double> hf(36*(size_t)D.n_far); CK(hipMemcpy(hf.data(), W.Sfar, sizeof(double)*hf.size(), hipMemcpyDeviceToHost));
            for (int q = 0; q < D.n_far; q++) { if (fi[H.far_a[q]] < 0 || fi[H.far_b[q]] < 0) continue;
                bool nz = false; for (int e = 0; e < 36; e++) nz |= hf[36*(size_t)q + e] != 0.0;
                if (nz) emit(H.far_a[q], H.far_b[q], hf.data() + 36*(size_t)q); }
        }
        *nblk = (int32_t)c->rb_r.size();
        return TSBA_OK;
    }
    if ((size_t)*nblk != c->rb_r.size()) { set_err(c, "tsba_debug_reduced_blocks: call with kf_r == NULL first"); return TSBA_ERR_STATE; }
    memcpy(kf_r, c->rb_r.data(), sizeof(int32_t)*c->rb_r.size()); memcpy(kf_c, c->rb_c.data(), sizeof(int32_t)*c->rb_c.size());
    if (val) memcpy(val, c->rb_v.data(), sizeof(double)*c->rb_v.size());
    Work &W = c->W;
    std::vector<int> fi(c->n_kf); CK(hipMemcpy(fi.data(), W.fidx, sizeof(int)*c->n_kf, hipMemcpyDeviceToHost));
    if (g_kf) { std::vector<double> gr(W.N); CK(hipMemcpy(gr.data(), W.g, sizeof(double)*W.N, hipMemcpyDeviceToHost));
        for (int k = 0; k < c->n_kf; k++) for (int e = 0; e < 6; e++) g_kf[6*k + e] = fi[k] >= 0 ? gr[6*(size_t)fi[k] + e] : 0.0; }
    if (dp_kf) CK(hipMemcpy(dp_kf, W.dp, sizeof(double)*W.N, hipMemcpyDeviceToHost));
    if (cost) { LmState st; CK(hipMemcpy(&st, W.st, sizeof(st), hipMemcpyDeviceToHost)); *cost = st.x_cost; }
    c->rb_r.clear(); c->rb_c.clear(); c->rb_v.clear(); c->rb_r.shrink_to_fit(); c->rb_c.shrink_to_fit(); c->rb_v.shrink_to_fit();
    return TSBA_OK;
}
// Per LM trial of pass `pass` of the last solve: out[4 k] = candidate cost (NaN: invalid step), [4 k + 1] = model cost change, [4 k + 2] = radius
// after the decision, [4 k + 3] = 1 accepted / 0 rejected / -1 invalid step / 2 tolerance exit on this trial.  Returns the number of trials
// recorded (min(iterations of the pass, cap, TSBA_TRACE_CAP)) or a negative error.  (The fused pose-only kernel keeps no trace.)
int tsba_debug_lm_trace(void *ctx, int pass, double *out, int cap) {
    Ctx *c = (Ctx *)ctx; if (!c || !out || pass < 0 || pass >= TSBA_MAX_LEVELS || cap <= 0) return TSBA_ERR_ARG;
    if (!c->uploaded || !c->W.trace) return TSBA_ERR_STATE;
    hipSetDevice(c->device); CK(hipStreamSynchronize(c->stream));
    const int n = std::min({cap, TSBA_TRACE_CAP, (int)c->st_host[pass].it});
    if (n > 0) CK(hipMemcpy(out, c->W.trace + 4*(size_t)pass*TSBA_TRACE_CAP, sizeof(double)*4*(size_t)n, hipMemcpyDeviceToHost));
    return n;
}

int tsba_time_linearize(void *ctx, int level, int n, double *avg_ms, double *algo_bytes) {
    Ctx *c = (Ctx *)ctx; if (!c || n <= 0) return TSBA_ERR_ARG;
    if (!c->uploaded || level < 0 || level >= c->n_levels || !c->lev_built[level]) { set_err(c, "level not uploaded"); return TSBA_ERR_STATE; }
    hipSetDevice(c->device);
    int rc = reset_state(c); if (rc) return rc;
    const LevelDev &D = c->lev[level];
    int ps = 0; for (int k = 0; k < c->opt.n_passes; k++) if (c->opt.levels[k] == level) ps = k;
    launch_pass_init(c, D, ps);
    launch_linearize(c, D, 0);                                 // warm-up (also leaves need_lin = 0)
    CK(hipStreamSynchronize(c->stream));
    LmState st; CK(hipMemcpy(&st, c->W.st, sizeof(st), hipMemcpyDeviceToHost));
    st.need_lin = 1; st.done = 0; st.first = 0;
    CK(hipMemcpy(c->W.st, &st, sizeof(st), hipMemcpyHostToDevice));   // k_linearize never clears need_lin itself
    CK(hipEventRecord(c->ev0, c->stream));
    for (int k = 0; k < n; k++) {
        if (lin_small_pairs(c, D) && D.n_tg == 0) hipLaunchKernelGGL((k_linearize<MODE_FULL, 4, false>), dim3((((D.n_pair + 4*LIN_NWV - 1)/(4*LIN_NWV) + 7)/8)*8), dim3(LIN_T), 0, c->stream, c->W, D, 0);
        else if (lin_small_pairs(c, D)) hipLaunchKernelGGL((k_linearize<MODE_FULL, 4>), dim3((((D.n_pair + 4*LIN_NWV - 1)/(4*LIN_NWV) + D.n_tg + 7)/8)*8), dim3(LIN_T), 0, c->stream, c->W, D, 0);
        else hipLaunchKernelGGL((k_linearize<MODE_FULL, 1>), dim3((((D.n_pair + LIN_NWV - 1)/LIN_NWV + D.n_tg + 7)/8)*8), dim3(LIN_T), 0, c->stream, c->W, D, 0);
    }
    CK(hipEventRecord(c->ev1, c->stream));
    CK(hipEventSynchronize(c->ev1));
    float ms = 0; CK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    if (avg_ms) *avg_ms = (double)ms/n;
    if (algo_bytes) {   // 44 B / scene block, 128 B / text block, 16 B / (KF,text) pair, parameters once
        int npairs_text = 0; std::vector<int> kin;
        (void)kin;
        for (int g = 0; g < D.n_tg; g++) npairs_text++;
        *algo_bytes = 44.0*st.ns_active + 128.0*st.nt_active + 16.0*npairs_text + 56.0*c->n_kf + 8.0*c->n_pt + 24.0*c->n_text;
    }
    return TSBA_OK;
}

int tsba_text_label_image(void *ctx, int kf, int level, float *out) {
    Ctx *c = (Ctx *)ctx; if (!c || !out) return TSBA_ERR_ARG;
    if (!c->uploaded) { set_err(c, "no problem uploaded"); return TSBA_ERR_STATE; }
    if (kf < 0 || kf >= c->n_kf || level < 0 || level >= c->n_levels || !c->lev_built[level]) { set_err(c, "keyframe / level out of range or level not uploaded"); return TSBA_ERR_ARG; }
    hipSetDevice(c->device);
    const LevelDev &D = c->lev[level];
    if (D.img_w <= 0 || D.img_h <= 0 || (size_t)D.img_w*D.img_h > (size_t)MS_MASK_WORDS*32) { set_err(c, "no image geometry for this level"); return TSBA_ERR_ARG; }
    const size_t npx = (size_t)D.img_w*D.img_h;
    if (c->lbl_cap < npx) { if (c->lbl_dev) hipFree(c->lbl_dev); if (c->lbl_host) hipHostFree(c->lbl_host); c->lbl_cap = 0;
        CK(hipMalloc((void **)&c->lbl_dev, npx*sizeof(float))); CK(hipHostMalloc((void **)&c->lbl_host, npx*sizeof(float), hipHostMallocDefault)); c->lbl_cap = npx; }
    hipLaunchKernelGGL(k_label, dim3(1), dim3(LBL_THREADS), 0, c->stream, c->W, kf, D.img_w, D.img_h, D.K[0], D.K[1], D.K[2], D.K[3], c->lbl_dev);
    CK(hipMemcpyAsync(c->lbl_host, c->lbl_dev, npx*sizeof(float), hipMemcpyDeviceToHost, c->stream));
    CK(hipStreamSynchronize(c->stream)); CK(hipGetLastError());
    memcpy(out, c->lbl_host, npx*sizeof(float));
    return TSBA_OK;
}

// which kernels the uploaded problem runs through (so that a test can assert that it exercises the path it means to):
// out[0] reduced system in LDS (k_solve_t / k_solve_col)   [1] band storage   [2] streaming band solver   [3] interiors P
// [4] separator system by cyclic reduction   [5] band rows   [6] four pairs per wave in the linearisation of the first pass's level
// [7] fused pose-only kernel   [8] one-wave Schur blocks + k_pose_sums (large maps)   [9] world size   [10] rank
// [11..14] size of this rank's plan of the first pass's level: (target, host) pairs, S blocks, scene candidates, point slots
// [15] the rows of S follow a reverse Cuthill-McKee order of the keyframes instead of the keyframe index
int tsba_debug_solver_info(void *ctx, int32_t *out, int n) {
    Ctx *c = (Ctx *)ctx; if (!c || !out || n < 16) return TSBA_ERR_ARG;
    if (!c->uploaded) return TSBA_ERR_STATE;
    int use_lds; solve_lds_bytes(c, &use_lds);
    int bwmax = 0; for (int l = 0; l < c->n_levels; l++) if (c->lev_built[l]) bwmax = std::max(bwmax, c->lev[l].bw_rows);
    out[0] = use_lds; out[1] = c->W.band; out[2] = c->band_stream; out[3] = c->band_stream ? c->band_parts : 0; out[4] = c->sep_cr ? 1 : 0; out[5] = bwmax;
    out[6] = lin_small_pairs(c, c->lev[c->opt.levels[0]]) ? 1 : 0; out[7] = c->pose_only ? 1 : 0; out[8] = c->n_kf > 126 ? 1 : 0;
    out[9] = c->world; out[10] = c->rank;
    { const LevelDev &D0 = c->lev[c->opt.levels[0]]; out[11] = D0.n_pair; out[12] = D0.n_sb; out[13] = D0.n_sc; out[14] = D0.n_pslot; out[15] = D0.kf_order ? 1 : 0; }
    if (n >= 17) out[16] = c->W.ring;
    if (n >= 19) { out[17] = c->far_B; out[18] = c->n_far; }
    return TSBA_OK;
}
// Iterative reduced-system solves of the last tsba_solve on a map with long-range coupling (tsba_pcg.h): out[0] conjugate-gradient iterations in
// total, [1] reduced systems solved (LM trials), [2] most iterations of one system, [3] systems that hit the iteration cap.  Zeros otherwise.
int tsba_debug_pcg_stats(void *ctx, int32_t out[4]) {
    Ctx *c = (Ctx *)ctx; if (!c || !out) return TSBA_ERR_ARG;
    if (!c->uploaded) return TSBA_ERR_STATE;
    out[0] = out[1] = out[2] = out[3] = 0;
    if (c->far_B <= 0) return TSBA_OK;
    hipSetDevice(c->device); CK(hipStreamSynchronize(c->stream));
    CK(hipMemcpy(out, c->W.pc_stat, 4*sizeof(int32_t), hipMemcpyDeviceToHost));
    return TSBA_OK;
}
// The 6x6 blocks outside the band after tsba_debug_reduced_system / a solve: keyframes a < b of block q and its 36 values (row-major, rows = a);
// the number of blocks is solver_info [18].  Any output may be NULL.
int tsba_debug_far_blocks(void *ctx, int32_t *a, int32_t *b, double *blocks) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    if (!c->uploaded || c->far_B <= 0) return TSBA_ERR_STATE;
    const LevelDev &D = c->lev[c->opt.levels[0]]; const HostPlan &H = c->hplan[D.level];
    hipSetDevice(c->device); CK(hipStreamSynchronize(c->stream));
    if (a) memcpy(a, H.far_a.data(), sizeof(int32_t)*H.far_a.size());
    if (b) memcpy(b, H.far_b.data(), sizeof(int32_t)*H.far_b.size());
    if (blocks && D.n_far > 0) CK(hipMemcpy(blocks, c->W.Sfar, sizeof(double)*36*(size_t)D.n_far, hipMemcpyDeviceToHost));
    return TSBA_OK;
}
// plane cache of the context (tsba_problem.kf_id): keyframes found on the device / copied, over the context's lifetime
int tsba_debug_img_cache_stats(void *ctx, int64_t out[2]) {
    Ctx *c = (Ctx *)ctx; if (!c || !out) return TSBA_ERR_ARG;
    out[0] = c->ic.hits; out[1] = c->ic.misses; return TSBA_OK;
}
// Test hook of the multi-right-hand-side solve phase (tsba_bandms.h): M X = R with the band factor the last tsba_debug_reduced_system / solve left
// behind.  R, X: [6 nfree][T] row-major (compressed free-pose rows).  TSBA_ERR_STATE unless the problem runs through the partitioned band
// solver with the cyclic-reduction separator system on a chain.
int tsba_debug_multi_solve(void *ctx, int T, const double *R, double *X) {       // T = -1: one column through the single-vector solve phase (tsba_bandsv.h)
    Ctx *c = (Ctx *)ctx; if (!c || (T < 1 && T != -1) || !R || !X) return TSBA_ERR_ARG;
    if (!c->uploaded || !ms_available(c)) { if (c) set_err(c, "multi-right-hand-side solve: needs the partitioned band solver with cyclic reduction on a chain"); return TSBA_ERR_STATE; }
    hipSetDevice(c->device);
    int nfree = 0; CK(hipMemcpy(&nfree, c->W.nfree, sizeof(int), hipMemcpyDeviceToHost));
    const bool single = T == -1; if (single) T = 1;
    int rc = single ? sv_reserve(c) : ms_reserve(c, T); if (rc) return rc;
    { int rca = set_solver_attrs(c); if (rca) return rca; }
    LmState st; CK(hipMemcpy(&st, c->W.st, sizeof(st), hipMemcpyDeviceToHost));
    st.done = 0; st.step_fail = 0; st.lin_done = 0; CK(hipMemcpy(c->W.st, &st, sizeof(st), hipMemcpyHostToDevice));
    const MsBuf &M = single ? c->sv : c->ms;
    CK(hipMemcpy(M.R, R, sizeof(double)*6*(size_t)nfree*T, hipMemcpyHostToDevice));
    if (single) { launch_sv_prepare(c, nullptr); launch_sv_solve(c, M.R, 1.0); }
    else { const bool mx = sv_reserve(c) == TSBA_OK; if (mx) launch_sv_prepare(c, nullptr); launch_ms_solve(c, mx); }
    CK(hipStreamSynchronize(c->stream)); CK(hipGetLastError());
    CK(hipMemcpy(X, M.X, sizeof(double)*6*(size_t)nfree*T, hipMemcpyDeviceToHost));
#ifdef SV_STAMPS
    if (single) { std::vector<double> st(8*(size_t)c->band_parts); CK(hipMemcpy(st.data(), M.Wm, sizeof(double)*st.size(), hipMemcpyDeviceToHost));
        for (int p : {0, 1, c->band_parts/2, c->band_parts - 1}) { fprintf(stderr, "sv stamps interior %d:", p); for (int k = 0; k < 8; k++) fprintf(stderr, " %.0f", st[8*(size_t)p + k]); fprintf(stderr, "\n"); } }
#endif
    return TSBA_OK;
}
// row block of every keyframe in the compressed reduced system of the last pass set-up (-1: constant / not participating); with a
// plan order (reverse Cuthill-McKee, tsba_plan.h) this is not monotone in the keyframe index
int tsba_debug_row_of_kf(void *ctx, int32_t *rowblk) {
    Ctx *c = (Ctx *)ctx; if (!c || !rowblk) return TSBA_ERR_ARG;
    if (!c->uploaded) return TSBA_ERR_STATE;
    hipSetDevice(c->device); CK(hipStreamSynchronize(c->stream));
    CK(hipMemcpy(rowblk, c->W.fidx, sizeof(int32_t)*c->n_kf, hipMemcpyDeviceToHost));
    return TSBA_OK;
}
// host only (no device needed): the plan's band bound of one level, with or without the keyframe reordering; order_out [n_kf] gets the
// row order (identity when the plan keeps the keyframe order)
int tsba_debug_plan_band(const tsba_problem *p, const tsba_options *o, int level, int reorder, int32_t *bw_pose, int32_t *order_out) {
    if (!p || !o || !bw_pose || level < 0 || level >= p->n_levels) return TSBA_ERR_ARG;
    HostPlan H; build_plan(p, o, level, H, false, reorder != 0);
    *bw_pose = H.bw_pose;
    if (order_out) for (int k = 0; k < p->n_kf; k++) order_out[k] = H.kf_order.empty() ? k : H.kf_order[(size_t)k];
    return TSBA_OK;
}
static int tsba_plan_time_dev_pairs = 0;    // (tsba_debug_plan_knob 4) tsba_debug_plan_time builds the plan as an upload does since round 6: the point slot pairs left to the device
int tsba_debug_plan_time(const tsba_problem *p, const tsba_options *o, int level, int reps, double *avg_ms) {   // host only: plan construction
    if (!p || !o || reps == 0) return TSBA_ERR_ARG;
    const bool laps = reps < 0; if (laps) reps = -reps;           // reps < 0: one recycled plan object (as a context does), lap times of the last build on stderr
    HostPlan R;
    if (laps) build_plan(p, o, level, R, false, true, CR_SMAX/6);
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; k++) { if (laps) build_plan(p, o, level, R, k == reps - 1, true, CR_SMAX/6, 0, false, tsba_plan_time_dev_pairs != 0); else { HostPlan H; build_plan(p, o, level, H, false, true, 0, 0, false, tsba_plan_time_dev_pairs != 0); } }
    *avg_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()/reps;
    return TSBA_OK;
}
int tsba_debug_time_solve(void *ctx, int n, double *avg_ms) {     // n back-to-back launches of the dense solve on the last S, g
    Ctx *c = (Ctx *)ctx; if (!c || n <= 0 || !c->uploaded) return TSBA_ERR_ARG;
    hipSetDevice(c->device);
    CK(hipStreamSynchronize(c->stream));
    LmState st; CK(hipMemcpy(&st, c->W.st, sizeof(st), hipMemcpyDeviceToHost));
    st.done = 0; st.step_fail = 0;
    CK(hipMemcpy(c->W.st, &st, sizeof(st), hipMemcpyHostToDevice));
    launch_solve(c);
    CK(hipEventRecord(c->ev0, c->stream));
    for (int k = 0; k < n; k++) launch_solve(c);
    CK(hipEventRecord(c->ev1, c->stream));
    CK(hipEventSynchronize(c->ev1));
    float ms = 0; CK(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *avg_ms = (double)ms/n;
    return TSBA_OK;
}

int tsba_debug_copy_S(void *ctx, double *out) {      // (6 n_kf + 1) x (6 n_kf): factored S and the rhs row after a solve
    Ctx *c = (Ctx *)ctx; if (!c || !c->uploaded) return TSBA_ERR_STATE;
    hipSetDevice(c->device); hipStreamSynchronize(c->stream);
    const Work &W = c->W; const long long N = W.N;
    if (!W.band) { if (hipMemcpy(out, W.S, sizeof(double)*(size_t)N*N, hipMemcpyDeviceToHost) != hipSuccess) return TSBA_ERR_DEVICE; }
    else { std::vector<double> hb(c->S_count); if (hipMemcpy(hb.data(), c->S_alloc, sizeof(double)*c->S_count, hipMemcpyDeviceToHost) != hipSuccess) return TSBA_ERR_DEVICE;
        const long long LDB = W.ldS + 1, Wb = LDB - c->S_up;
        for (long long i = 0; i < N; i++) for (long long j = 0; j < N; j++)
            out[i*N + j] = (j >= i - Wb && j <= i + c->S_up - 1) ? hb[(size_t)(Wb + i*(LDB - 1) + j)] : 0.0; }
    // row N = the rhs row: of the large-system solver if that ran, else unused (the LDS solver keeps it on chip)
    return hipMemcpy(out + (size_t)N*N, W.Sy, sizeof(double)*(size_t)N, hipMemcpyDeviceToHost) == hipSuccess ? 0 : TSBA_ERR_DEVICE;
}

int tsba_debug_band_factor(void *ctx, double *lcol, long long n_lcol, double *ldbuf, long long n_ld) {      // test hook: streaming band solver's factor
    Ctx *c = (Ctx *)ctx; if (!c || !c->uploaded || !c->Lcol) return TSBA_ERR_STATE;
    hipSetDevice(c->device); hipStreamSynchronize(c->stream);
    if (hipMemcpy(lcol, c->Lcol, sizeof(double)*n_lcol, hipMemcpyDeviceToHost) != hipSuccess) return TSBA_ERR_DEVICE;
    return hipMemcpy(ldbuf, c->W.LDbuf, sizeof(double)*n_ld, hipMemcpyDeviceToHost) == hipSuccess ? 0 : TSBA_ERR_DEVICE;
}
// host-side index arithmetic of the partitioned band solver, for the CPU test-suite (no device needed):
// out5 = { P, a, b, has_left, has_right } of interior p;  block index of (br, bc) in the cyclic-reduction pool and the pool size
// host-only: FNV-1a over the Schur slot-pair lists of the plan of `level`, built with `threads` host threads in the parallel sections
// (0 = the production choice): the plan must not depend on the number of threads
static int tsba_plan_checksum_ring = 0;       // ring_max_blocks the checksum hook builds its plan with (knob 2)
void tsba_debug_plan_knob(int which, int value) { if (which == 0) tsba_plan_threads = value; else if (which == 1) tsba_plan_mark_mt = value; else if (which == 2) tsba_plan_checksum_ring = value; else if (which == 3) tsba_plan_pin = value; else if (which == 4) tsba_plan_time_dev_pairs = value; }   // host-only measurement knobs
