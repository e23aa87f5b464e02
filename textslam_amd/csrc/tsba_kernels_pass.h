// The start and the end of a pass of a small window in ONE launch each.  (part of the single translation unit tsba.hip: included there, in this order)
//
// A pass (one pyramid level of optimizer::PyrBA, optimizer.cc:282-289 / :1593-1605) used to begin with four launches -- k_pass_reset, k_participation,
// k_gauge_wave, k_musigma -- and to end with k_outlier and a device-to-device copy of the LM state; the solve ended with a device-to-host copy.  On a
// 20-keyframe window each of them is a launch, a round trip or two and no work to speak of (41 + 20 us per pass, three passes per solve).  Here:
//
//   k_pass_begin   workgroups [0, nwg): k_participation's blocks, taken in turn (good flags -> which landmarks / keyframes take part, block counts); the LAST of them to finish
//                  (ticket) keeps the previous pass's final state for the report, resets the LM state and fixes the gauge (k_gauge_wave's ballots);
//                  workgroups [nwg, nwg + n_ms): the text observations' mu / sigma (k_musigma) -- unless the previous pass's k_pass_end has computed them
//   k_pass_end     the outlier pass (four of k_outlier's waves per workgroup)  |  mu / sigma of the NEXT pass's level at the final parameters of this one
//                  (into the other of two buffers: the outlier pass still reads this level's)  |  one workgroup that clears the participation arrays
//   k_solve_end    the passes' final states -> pinned host memory (no copy engine, no staging)
//
// What orders what: the participation arrays are cleared by the launch BEFORE k_pass_begin (k_reset_state at the start of a solve, k_pass_end after a pass);
// nothing in k_pass_begin reads the LM state but `cur`, which the reset leaves alone; the ticket is the only inter-workgroup hand-over and it is the
// classic one (results, fence, atomic; the last arrival fences and reads): no workgroup ever waits for another.
#pragma once

// windows of up to 64 keyframes: one lane per keyframe, ballots instead of a serial walk (the body of k_gauge_wave)
__device__ __forceinline__ void gauge_wave_body(const Work &W, const uint8_t *kf_initial, int state, int k) {
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

#ifndef PB_WG
#define PB_WG 128                           // workgroups that walk k_participation's blocks in k_pass_begin.  Round 6: 24 -> 128 (C4 level 0: 215 blocks, nine in turn per workgroup at ~2 us
                                            // each were the kernel once mu / sigma stopped being it: 20.7 us at 24, 15.9 at 48, 14.4 at 72, 13.6 at 128; the ticket's 128 arrivals cost ~4 us of that)
#endif
__global__ __launch_bounds__(MS_THREADS) void k_pass_begin(Work W, LevelDev L, double radius0, int max_it, const uint8_t *kf_initial, int state,
                                                           int npb, int nwg, int n_ms, LmState *log_prev, int *ticket) {
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b >= nwg) { if (b - nwg < n_ms) musigma_wg(W, L, b - nwg, W.pose[W.st->cur], W.theta[W.st->cur]); return; }
    for (int vb = b; vb < npb; vb += nwg) { participation_wg(W, L, vb, 1); __syncthreads(); }       // (k_participation's block vb: its partial counts go to cntpart[vb])
    __shared__ int s_last; __shared__ int s_cnt2[2];
    if (tid == 0) { __threadfence(); s_last = atomicAdd(ticket, 1) == nwg - 1; s_cnt2[0] = 0; s_cnt2[1] = 0; }
    __syncthreads();
    if (!s_last) return;
    __threadfence();                                              // (the other workgroups' flags and counts)
    sum_counts(W, npb, tid, MS_THREADS, s_cnt2);
    __syncthreads();
    if (tid == 0) {
        LmState *s = W.st;
        if (log_prev) *log_prev = *s;                           // the pass before this one, with its outlier counts (k_pass_end ran in between)
        s->radius = radius0; s->decrease_factor = 2.0; s->x_cost = 0; s->x_norm = 0; s->cand_cost = 0; s->model_change = 0;      // (every field but cur / n_lin / n_cost, which carry over)
        s->step_norm = 0; s->gmax = 0; s->cost0 = 0;
        s->done = 0; s->need_lin = 1; s->first = 1; s->it = 0; s->accepted = 0; s->term = 0; s->invalid = 0; s->max_it = max_it;
        s->step_fail = 0; s->lcur = 0; s->lin_done = 0; s->pad2 = 0;
        s->ns_active = s_cnt2[0]; s->nt_active = s_cnt2[1]; s->n_bad_scene = 0; s->n_bad_tfeat = 0; s->n_bad_text = 0;
        if (W.hprog) { *W.hprog = (unsigned long long)W.pass_seq << 32; __threadfence_system(); }
        *ticket = 0;
    }
    if (tid < 64) gauge_wave_body(W, kf_initial, state, tid);
}

// nb_out: k_outlier's blocks of this pass ((n_sc + 63)/64 + n_tg; 0: no outlier pass).  n_ms: text observations of the NEXT pass's level Ln whose mu / sigma
// go to ms_next (0: there is no next pass, or its level is not on the device yet -- k_pass_begin computes them then)
__global__ __launch_bounds__(MS_THREADS) void k_pass_end(Work W, LevelDev L, LevelDev Ln, int nb_out, int n_ms, double *ms_next,
                                                         double chi2_mono, double chi2_text, double bad_ratio, int do_scene, int do_text) {
    const int b = blockIdx.x, tid = threadIdx.x, nbo = (nb_out + 3) >> 2;
    if (b < nbo) { const int ob = 4*b + (tid >> 6); if (ob < nb_out) outlier_wave(W, L, ob, tid & 63, chi2_mono, chi2_text, bad_ratio, do_scene, do_text, nullptr); return; }
    if (b < nbo + n_ms) { Work Wn = W; Wn.musig = ms_next; musigma_wg(Wn, Ln, b - nbo, W.pose[W.st->cur], W.theta[W.st->cur]); return; }
    for (int k = tid; k < W.n_kf; k += MS_THREADS) W.kf_in[k] = 0;
    for (int k = tid; k < W.n_pt; k += MS_THREADS) W.act_pt[k] = 0;
    for (int k = tid; k < W.n_text; k += MS_THREADS) W.act_tx[k] = 0;
}

// the participation arrays cleared by a launch of its own (after a pass that ended with the launches of rounds 1-4 when a later one begins with k_pass_begin)
__global__ __launch_bounds__(256) void k_part_clear(Work W) {
    const int t = blockIdx.x*256 + threadIdx.x, n = gridDim.x*256;
    for (int k = t; k < W.n_kf; k += n) W.kf_in[k] = 0;
    for (int k = t; k < W.n_pt; k += n) W.act_pt[k] = 0;
    for (int k = t; k < W.n_text; k += n) W.act_tx[k] = 0;
}

// the passes' final states into pinned host memory: st_log[0 .. n - 2] were kept by the following pass's k_pass_begin (log_last: the last one is still in
// W.st; null: every pass's state was copied as it ended)
__global__ __launch_bounds__(64) void k_solve_end(Work W, LmState *st_log, int n, int log_last, LmState *host, unsigned int *host_giveups) {
    static_assert(sizeof(LmState) % 4 == 0, "LmState is copied as words");
    constexpr int NW = sizeof(LmState)/4;
    const int tid = threadIdx.x;
    for (int p = 0; p < n; p++) {
        const int *src = (const int *)((log_last && p == n - 1) ? W.st : st_log + p);
        int *dst = (int *)(host + p);
        for (int k = tid; k < NW; k += 64) dst[k] = src[k];
    }
    if (tid == 0 && host_giveups) *host_giveups = W.poll0 ? ts_poll_giveups - *W.poll0 : 0u;      // polls that ran into their bound while this solve was on the device
    __threadfence_system();
}
