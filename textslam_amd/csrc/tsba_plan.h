// Host-side "plan" of one pyramid pass: the static index structures the kernels walk.
//
// This is the flat counterpart of the reference's problem construction (row B1 of SURVEY.md 8a:
// optimizer.cc:1366-1557 PyrBA, :1126-1207 PyrPoseOptim, :1727-1765 PyrGlobalBA).  The reference adds residual
// blocks one by one to a ceres::Problem; here every block that COULD be added (ignoring the good flags, which
// change between passes on the device) becomes a candidate, grouped so that each later reduction is a gather:
//
//   pair   p = (target KF i, host KF h | -1)   all scene candidates of a pair are contiguous; text groups hang off it
//   group  g = one (KF, text) observation      = up to 64 photometric blocks that share (i, h, theta)
//   slot   s = one (landmark, observing pose) entry of W = J_p^T J_l; the last slot of a landmark is its host pose
//   sblock   = one 6x6 block (a <= b) of the reduced camera system with the slot pairs that feed it
#pragma once
#include <vector>
#include <algorithm>
#include <thread>
#include <mutex>
#include <condition_variable>
#include <atomic>
#include <functional>
#include <sched.h>
#include <pthread.h>
#include <cstdint>
#include <map>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../include/tsba.h"

static int tsba_plan_threads = 0;              // 0: by problem size; > 0: host threads of the plan builder's parallel sections (tests)
static int tsba_plan_pin = 0;                  // 1 (debug knob; tsba_options.host_plan_pin per call): the WORKER threads of a multi-threaded build are pinned to the CPUs that share
                                               // the caller's L3 -- every pass hands the lists the last pass wrote to other threads: through one L3 that is a cache hit, across the
                                               // 32 L3 domains of the GPU box's two sockets a remote-cache miss per line (5000 keyframes: 21.0 -> 13.3 ms).  Off by default: a
                                               // drop-in library does not touch scheduling unasked; the CALLING thread's affinity is never changed
static int tsba_plan_mark_mt = 1;              // S-block keys of the slot pairs marked by the same threads (0: by the calling thread; measurements)

struct PlanCand { int obs, kf, pt, host; };     // a scene candidate: observation, target keyframe, point, host keyframe (-1: frozen)
// work lists of build_plan that live with the plan object: a context builds a plan per call, and 45 MB of fresh lists per build of a
// 5000-keyframe map (page faults, unmapping under sixteen running threads) cost more than filling them
struct PlanScratch {
    std::vector<std::vector<PlanCand> > cs;     // per thread
    std::vector<std::vector<int32_t> > hist, kc, tlo, thi;
    std::vector<int32_t> sc_pair;
    void threads(int T) { if ((int)cs.size() < T) { cs.resize((size_t)T); hist.resize((size_t)T); kc.resize((size_t)T); tlo.resize((size_t)T); thi.resize((size_t)T); } }
};
#define WB_MAXKF_PLAN 128                   // keyframes a loop closure's blocks may touch for the low-rank correction (tsba_wb.h: WB_MAXKF)
struct HostPlan {
    PlanScratch scratch;                        // (kept by recycle())
    int level = 0;
    int bw_pose = 0;                            // half bandwidth of the reduced camera matrix in pose blocks, fill included
    int ring_k0 = 0;                            // ring: the loop starts at this keyframe (0: the whole trajectory is the loop; > 0: a tail before it)
    int ring = 0;                               // 1: the co-visibility graph is a RING (one loop closure between the last keyframes and keyframe ring_k0): bw_pose is the band of
                                                // the chain unrolled past its end -- the loop's first bw_pose poses re-appear as ghost rows behind the last pose (tsba_bandp.h)
    std::vector<int32_t> kf_order;              // empty: S is ordered by keyframe index; else kf_order[i] = keyframe at position i (reverse Cuthill-McKee)
    // Band + long-range coupling (far_B > 0): S = M + E.  Every landmark's poses fall into CLUSTERS that span at most far_B keyframes -- cluster 0: the
    // host and the observers up to far_B keyframes after it, the other poses greedily in ascending order.  M takes a landmark's contribution
    // restricted to each cluster (a principal submatrix of a positive semidefinite matrix: M stays positive definite whatever the map looks like
    // -- simply cutting S off at a band does not), a band of far_B pose blocks in keyframe order: the sb_* lists below.  E takes what couples
    // DIFFERENT clusters of a landmark (points seen again much later, loop closures; also the rare observer before its host): the far_* / fb_*
    // lists.  The system is solved by conjugate gradients preconditioned with M (tsba_pcg.h).  The block positions far_a < far_b are a property of
    // the WHOLE problem (every rank of a sharded solve derives the same list, sorted); fb_* are this rank's contributions in the layout of sb_*.
    int far_B = 0;
    std::vector<int32_t> far_a, far_b;          // [n_far] keyframes of a block of E
    std::vector<int32_t> far_off, far_ent;      // per keyframe: its blocks of E as (index << 1 | 1 if the keyframe is far_b), ascending
    std::vector<int32_t> wb_kf, wb_idx;         // the keyframes a block of E touches, when they are few (<= WB_MAXKF_PLAN: loop closures -- the low-rank correction of tsba_wb.h), and every keyframe's index in that list (-1)
    std::vector<int32_t> fb_id, fb_pab, fb_pba, fb_pt_off, fb_pt_s1, fb_pt_s2, fb_pt_lm, fb_tx_off, fb_tx_s1, fb_tx_s2, fb_tx_lm;    // fb_id: 0 .. n_far - 1
    int n_far() const { return (int)far_a.size(); }
    // Large maps: the slot pairs of the POINT landmarks by S block (sb_pt_off / _s1 / _s2 / _lm: 2 M entries at 5000 keyframes, 9 of the plan's 21 ms
    // on sixteen host threads) are built on the device by tsba_devplan.h from lists that are uploaded anyway; the host then only counts them.
    // dev_pt_pairs >= 0: their number (the four host lists stay empty); cl_pt_dev: cluster of every point slot when the band / long-range split is on
    int64_t dev_pt_pairs = -1; std::vector<int32_t> cl_pt_dev;
    // scene candidates (sorted by pair)
    std::vector<int32_t> sc_obs, sc_kf, sc_pt, sc_flag, sc_slot;
    std::vector<double>  sc_uv;                 // [n_sc][2]
    // pairs
    std::vector<int32_t> pair_i, pair_h, pair_hpos, pair_sc_off, pair_tg_off, pair_tg;   // pair_hpos: rank of the pair in host-major order (-1: frozen host)
    // text groups
    std::vector<int32_t> tg_tobs, tg_kf, tg_text, tg_pair, tg_slot;
    std::vector<int32_t> pt_pose6;      // poses of the first 6 slots of every point (clamped): k_back needs them one round trip earlier
    std::vector<int32_t> pt_pair4;      // pairs of the first PT_PAIRN (6) observer slots of every point (clamped): k_mid fetches their R_cr together with the slot records
    std::vector<int32_t> tx_pair8;      // pairs of the first 8 observer slots of every plane (clamped): k_mid gives a plane eight lanes, one slot record each
    std::vector<int32_t> tg_ppos;       // rank of the group in pair-major order (pair_tg is the inverse): k_mid sums contiguous ranges
    std::vector<int32_t> pf_g, pf_f;    // single-keyframe problems (pose-only path): flat list of (group, feature) over all groups
    std::vector<int32_t> tg_rec;        // per group, one 32-byte record: tobs, kf, text, host, slot, f0, f1, fgood offset (all static)
    // landmark slots
    std::vector<int32_t> pls_off, pslot_pose, pslot_pair, pslot_lm;     // points
    std::vector<int32_t> tls_off, tslot_pose, tslot_pair, tslot_lm;     // text planes
    // reduced-system blocks
    std::vector<int32_t> sb_a, sb_b, sb_pab, sb_pba;
    std::vector<int32_t> sb_rng;       // per block, four entries: a diagonal block's ranges of pair products (pose_t_off[a], pose_t_off[a+1], pose_h_off[a], pose_h_off[a+1]) -- k_schur_t requests them with the block itself
    std::vector<int32_t> sb_pt_off, sb_pt_s1, sb_pt_s2, sb_pt_lm, sb_tx_off, sb_tx_s1, sb_tx_s2, sb_tx_lm;
    // per pose: pairs where it is target / host; slots it owns
    std::vector<int32_t> pose_t_off, pose_t, pose_h_off, pose_h;
    std::vector<int32_t> pose_ps_off, pose_ps, pose_ps_lm, pose_ts_off, pose_ts, pose_ts_lm;
    // text blocks flattened (group, feature) for the eval hook / outlier pass
    int n_sc() const { return (int)sc_obs.size(); }
    int n_pair() const { return (int)pair_i.size(); }
    int n_tg() const { return (int)tg_tobs.size(); }
    int n_pslot() const { return (int)pslot_pose.size(); }
    int n_tslot() const { return (int)tslot_pose.size(); }
    int n_sb() const { return (int)sb_a.size(); }
    // a new plan into the same object (a context builds one per call: per keyframe for the local BA): scalars reset, every list emptied with its
    // storage kept -- no allocation, no page faults for lists of the size the last call needed
    void recycle() {
        level = 0; bw_pose = 0; ring = 0; ring_k0 = 0; far_B = 0;
        for (std::vector<int32_t> *v : { &wb_kf, &wb_idx, &far_a, &far_b, &far_off, &far_ent, &fb_id, &fb_pab, &fb_pba, &fb_pt_off, &fb_pt_s1, &fb_pt_s2, &fb_pt_lm, &fb_tx_off, &fb_tx_s1, &fb_tx_s2, &fb_tx_lm, &kf_order, &sc_obs, &sc_kf, &sc_pt, &sc_flag, &sc_slot, &pair_i, &pair_h, &pair_hpos, &pair_sc_off, &pair_tg_off, &pair_tg,
                                         &tg_tobs, &tg_kf, &tg_text, &tg_pair, &tg_slot, &pt_pose6, &pt_pair4, &tx_pair8, &tg_ppos, &pf_g, &pf_f, &tg_rec,
                                         &pls_off, &pslot_pose, &pslot_pair, &pslot_lm, &tls_off, &tslot_pose, &tslot_pair, &tslot_lm,
                                         &sb_a, &sb_b, &sb_pab, &sb_pba, &sb_rng, &sb_pt_off, &sb_pt_s1, &sb_pt_s2, &sb_pt_lm, &sb_tx_off, &sb_tx_s1, &sb_tx_s2, &sb_tx_lm,
                                         &pose_t_off, &pose_t, &pose_h_off, &pose_h, &pose_ps_off, &pose_ps, &pose_ps_lm, &pose_ts_off, &pose_ts, &pose_ts_lm }) v->clear();
        sc_uv.clear(); dev_pt_pairs = -1; cl_pt_dev.clear();
    }
};

// Dense id of a sparse set of integer keys, ids in key order.  Small key ranges (a 20-keyframe window has 420 possible pairs)
// use a direct table -- no sort, no binary search: plan construction is on the critical path of every cold tsba_local_ba call.
struct KeyIndex {
    // three modes by key range: direct table (<= 4 M keys: a 20-keyframe window has 420 possible pairs), bitmap + rank directory
    // (<= 2^31 keys: 5000 keyframes have 25 M possible pose pairs -- 3 MB of bits, O(1) id by popcount, no sort and no binary search
    // over millions of slot pairs), sorted list beyond that
    int64_t range = 0; std::vector<int32_t> table; std::vector<int64_t> keys;       // table[key] = id (dense mode)
    std::vector<uint64_t> bits; std::vector<uint32_t> rank;                           // bitmap mode: rank[w] = set bits before word w
    bool dense() const { return !table.empty(); }
    bool bitmap() const { return !bits.empty(); }
    void begin(int64_t key_range) {
        range = key_range; table.clear(); keys.clear(); bits.clear(); rank.clear();
        if (range <= (int64_t)1 << 22) table.assign((size_t)range, -1);
        else if (range <= (int64_t)1 << 31) bits.assign((size_t)((range + 63) >> 6), 0);
    }
    void add(int64_t k) { if (dense()) table[(size_t)k] = 0; else if (bitmap()) bits[(size_t)(k >> 6)] |= (uint64_t)1 << (k & 63); else keys.push_back(k); }
    bool concurrent() const { return dense() || bitmap(); }                           // add_mt may be called from several threads at once
    void add_mt(int64_t k) { if (dense()) { int32_t *e = &table[(size_t)k]; if (__atomic_load_n(e, __ATOMIC_RELAXED) != 0) __atomic_store_n(e, 0, __ATOMIC_RELAXED); }
                             else { uint64_t *w = &bits[(size_t)(k >> 6)]; const uint64_t m = (uint64_t)1 << (k & 63);      // (test first: most slot pairs hit a bit that is set already, and a line that is only read stays shared)
                                    if (!(__atomic_load_n(w, __ATOMIC_RELAXED) & m)) __atomic_fetch_or(w, m, __ATOMIC_RELAXED); } }
    int finish() {                                                                   // returns the number of distinct keys
        if (dense()) { int n = 0; keys.clear(); for (int64_t k = 0; k < range; k++) if (table[(size_t)k] == 0) { table[(size_t)k] = n++; keys.push_back(k); } return n; }
        if (bitmap()) {
            rank.resize(bits.size()); keys.clear(); uint32_t n = 0;
            for (size_t w = 0; w < bits.size(); w++) { rank[w] = n; uint64_t b = bits[w];
                while (b) { const int t = __builtin_ctzll(b); keys.push_back((int64_t)(w << 6) + t); b &= b - 1; n++; } }
            return (int)n;
        }
        std::sort(keys.begin(), keys.end()); keys.erase(std::unique(keys.begin(), keys.end()), keys.end()); return (int)keys.size();
    }
    int id(int64_t k) const {
        if (dense()) return table[(size_t)k];
        if (bitmap()) { const size_t w = (size_t)(k >> 6); return (int)(rank[w] + (uint32_t)__builtin_popcountll(bits[w] & (((uint64_t)1 << (k & 63)) - 1))); }
        return (int)(std::lower_bound(keys.begin(), keys.end(), k) - keys.begin());
    }
};
// stable counting sort: order[] = indices 0..n-1 sorted by bucket[], off[] = CSR offsets (n_bucket + 1)
inline void bucket_order(const std::vector<int> &bucket, int n_bucket, std::vector<int> &order, std::vector<int32_t> &off) {
    off.assign((size_t)n_bucket + 1, 0);
    for (int b : bucket) off[(size_t)b + 1]++;
    for (int q = 0; q < n_bucket; q++) off[q+1] += off[q];
    std::vector<int32_t> cur(off.begin(), off.end() - 1);
    order.resize(bucket.size());
    for (size_t i = 0; i < bucket.size(); i++) order[(size_t)cur[bucket[i]]++] = (int)i;
}

// Reverse Cuthill-McKee ordering of the keyframe co-visibility graph (nb: adjacency lists, deduplicated).  After a loop closure the
// graph is a ring -- keyframe n-1 shares landmarks with keyframe 0 -- and in keyframe order the envelope of S spans the whole matrix
// (7.2 GB dense at 5000 keyframes); walked breadth-first from a peripheral keyframe the two arms of the ring interleave and the band
// is twice the local one.  Components in turn, start = a pseudo-peripheral node (two breadth-first sweeps), neighbours by degree.
inline void rcm_order(int n, const std::vector<std::vector<int> > &nb, std::vector<int32_t> &order) {
    order.clear(); order.reserve(n);
    std::vector<char> seen(n, 0); std::vector<int> level(n, 0), q;
    auto bfs_last = [&](int s0) {                               // farthest node of minimum degree from s0 (within its component)
        std::vector<int> fr(1, s0), mark; std::vector<char> vis(n, 0); vis[s0] = 1; int last = s0;
        while (!fr.empty()) { std::vector<int> nx; int best = -1;
            for (int u : fr) { if (best < 0 || nb[u].size() < nb[best].size()) best = u; for (int v : nb[u]) if (!vis[v] && !seen[v]) { vis[v] = 1; nx.push_back(v); } }
            last = best; fr.swap(nx); }
        return last;
    };
    for (int s0 = 0; s0 < n; s0++) {
        if (seen[s0]) continue;
        int start = s0;
        if (!nb[s0].empty()) { start = bfs_last(s0); start = bfs_last(start); }
        const size_t head0 = order.size();
        order.push_back(start); seen[start] = 1;
        for (size_t head = head0; head < order.size(); head++) {
            const int u = order[head]; q.clear();
            for (int v : nb[u]) if (!seen[v]) { seen[v] = 1; q.push_back(v); }
            std::sort(q.begin(), q.end(), [&](int a, int b) { return nb[a].size() != nb[b].size() ? nb[a].size() < nb[b].size() : a < b; });
            for (int v : q) order.push_back(v);
        }
        std::reverse(order.begin() + (long)head0, order.end());
    }
}

// Which rank of a sharded global BA keeps a residual block: the rank whose KEYFRAME RANGE holds the landmark's host (ranges of
// n_kf / nshard consecutive keyframes).  Co-visibility is local in keyframe index, so a rank's blocks touch only the (target, host)
// pairs and the 6x6 blocks of S near its own range -- its linearisation and Schur assembly shrink with 1 / nshard (a landmark-index
// stride would leave every rank with every pair and every S block, each 1 / nshard full).  A frozen landmark (host outside the map)
// couples nothing: each of its observations goes with its target keyframe.
__host__ __device__ inline int tsba_shard_of(int host, int target_kf, int n_kf, int nshard) {
    const int k = host >= 0 ? host : target_kf;
    return (int)(((long long)k*nshard)/n_kf);
}

// ---- host threads of the plan builder.  A 5000-keyframe map has 0.5 M observations and 2 M slot pairs, and its plan is 2/3 of a cold
// tsba_global_ba call: every pass over them runs on a small fork-join pool (the calling thread is member 0; the others spin between the
// parallel sections of ONE build_plan call -- they live for some ten milliseconds).  Every section is a contiguous split of its items and
// every placement is a stable bucket sort, so the plan does not depend on the number of threads.
// the CPUs that share the last-level cache with the CPU this thread runs on, within the process's affinity mask (false: unknown / fewer than 2)
inline bool plan_l3_cpuset(cpu_set_t *set) {
    const int cpu = sched_getcpu(); if (cpu < 0) return false;
    char path[128]; snprintf(path, sizeof path, "/sys/devices/system/cpu/cpu%d/cache/index3/shared_cpu_list", cpu);
    FILE *f = fopen(path, "r"); if (!f) return false;
    char buf[1024]; const bool got = fgets(buf, sizeof buf, f) != nullptr; fclose(f); if (!got) return false;
    CPU_ZERO(set);
    for (const char *q = buf; *q && *q != '\n'; ) { char *e; long a = strtol(q, &e, 10); if (e == q) break; long b = a; q = e;
        if (*q == '-') { b = strtol(q + 1, &e, 10); q = e; }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++) if (c >= 0) CPU_SET((int)c, set);
        if (*q == ',') q++; }
    cpu_set_t allowed; if (sched_getaffinity(0, sizeof allowed, &allowed) == 0) CPU_AND(set, set, &allowed);
    return CPU_COUNT(set) >= 2;
}
// CPUs this process may run on at once: the affinity mask, cut by the cgroup's CPU quota (read once)
inline int plan_usable_cpus() {
    static const int n = []() { int c = (int)std::max(1u, std::thread::hardware_concurrency());
        cpu_set_t m; if (sched_getaffinity(0, sizeof m, &m) == 0) c = std::min(c, std::max(1, CPU_COUNT(&m)));
        if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { char q[64]; long per = 0; if (fscanf(f, "%63s %ld", q, &per) == 2 && per > 0 && strcmp(q, "max") != 0) c = std::min(c, std::max(1, (int)((atol(q) + per/2)/per))); fclose(f); }
        return c; }();
    return n;
}
struct PlanPool {
    int T; std::vector<std::thread> th; std::atomic<int> gen{0}, done{0}; std::atomic<bool> stop{false};
    const std::function<void(int)> *job = nullptr;
    std::mutex m; std::condition_variable cv;          // workers park here after a short spin (several contexts may build plans at once: no indefinite busy-waiting)
    explicit PlanPool(int T_, bool pin = false) : T(std::max(1, T_)) {
        cpu_set_t l3; const bool pinned = pin && T > 1 && plan_l3_cpuset(&l3);
        if (pinned) T = std::min(T, CPU_COUNT(&l3));
        for (int t = 1; t < T; t++) { th.emplace_back([this, t]() { worker(t); });
            if (pinned) pthread_setaffinity_np(th.back().native_handle(), sizeof l3, &l3); }       // the workers only: the caller keeps its own mask
    }
    ~PlanPool() { { std::lock_guard<std::mutex> lk(m); stop.store(true); gen.fetch_add(1, std::memory_order_release); } cv.notify_all(); for (auto &x : th) x.join(); }
    PlanPool(const PlanPool &) = delete; PlanPool &operator=(const PlanPool &) = delete;
    static void cpu_relax() {
#if defined(__x86_64__) || defined(__i386__)
        __builtin_ia32_pause();
#elif defined(__aarch64__)
        __asm__ __volatile__("yield");
#else
        std::this_thread::yield();
#endif
    }
    void worker(int t) { int seen = 0;
        for (;;) { int spins = 0, g;
            while ((g = gen.load(std::memory_order_acquire)) == seen) {
                if (++spins < 4096) { cpu_relax(); continue; }
                std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return gen.load(std::memory_order_acquire) != seen; }); }
            seen = g; if (stop.load()) return; (*job)(t); done.fetch_add(1, std::memory_order_release); } }
    template <class F> void run(F &&f) {                                // f(t) for t = 0 .. T-1, returns when all are done
        if (T <= 1) { f(0); return; }
        const std::function<void(int)> fn = [&f](int t) { f(t); };
        job = &fn; done.store(0, std::memory_order_relaxed);
        { std::lock_guard<std::mutex> lk(m); gen.fetch_add(1, std::memory_order_release); } cv.notify_all();
        f(0);
        int spins = 0; while (done.load(std::memory_order_acquire) != T - 1) { if (++spins > 2048) std::this_thread::yield(); else cpu_relax(); }
    }
    void range(size_t n, int t, size_t &a, size_t &b) const { a = n*(size_t)t/(size_t)T; b = n*(size_t)(t + 1)/(size_t)T; }
};
// Stable bucket placement in two passes over items the caller enumerates twice in the same order (thread t its own contiguous share):
// pass 1 count(t, key) -- key < 0: the item takes no place --, offsets(), pass 2 next(t, e) = the item's position (e: the thread's running
// item ordinal).  Thread-major within a bucket = item order, whatever the number of threads.
struct BucketPlacer {
    PlanPool &pool; int nb; std::vector<std::vector<int32_t> > &hist, &kc;       // (the plan's scratch lists: one placer at a time)
    BucketPlacer(PlanPool &pl, int n_bucket, PlanScratch &sc) : pool(pl), nb(n_bucket), hist(sc.hist), kc(sc.kc) {}
    void begin(int t, size_t expect) { hist[t].assign((size_t)nb, 0); kc[t].clear(); kc[t].reserve(expect); }
    void count(int t, int key) { kc[t].push_back(key); if (key >= 0) hist[t][(size_t)key]++; }
    // off[q] = start of bucket q (nb + 1 entries); a non-empty bucket is `extra` entries longer than its items (a landmark's host slot)
    void offsets(std::vector<int32_t> &off, int extra = 0) {
        const int T = pool.T; off.assign((size_t)nb + 1, 0);
        pool.run([&](int t) { size_t a, b; pool.range((size_t)nb, t, a, b);
            for (size_t q = a; q < b; q++) { int32_t tot = 0; for (int u = 0; u < T; u++) tot += hist[u][q]; off[q + 1] = tot > 0 ? tot + extra : 0; } });
        for (int q = 0; q < nb; q++) off[(size_t)q + 1] += off[(size_t)q];
        pool.run([&](int t) { size_t a, b; pool.range((size_t)nb, t, a, b);
            for (size_t q = a; q < b; q++) { int32_t run = off[q]; for (int u = 0; u < T; u++) { const int32_t c = hist[u][q]; hist[u][q] = run; run += c; } } });
    }
    int key(int t, size_t e) const { return kc[t][e]; }
    int next(int t, size_t &e) { const int k = kc[t][e++]; return k < 0 ? -1 : hist[t][(size_t)k]++; }
};

// The plan of a SINGLE-FRAME problem whose landmarks are all frozen in hosts outside it (optimizer::PoseOptim, optimizer.cc:135-195 -- the call the tracking
// thread makes per frame): one (target, frozen host) pair, every observation a candidate in its own order, no landmark slot, one diagonal block.  build_plan
// arrives at exactly these lists through its generic passes (candidate keys, bucket placements, slot pairs: 0.03 - 0.07 ms per level, on threads of their own in
// a one-shot call); here they are written down directly -- a copy of the input arrays -- in a few microseconds on the calling thread, so that a one-shot
// tsba_pose_optim call plans and stages all its levels before the solve (round 6; tests/test_band_partition.py compares the checksum over every list with build_plan's).
inline bool plan_is_single_frame(const tsba_problem *p, const tsba_options *o) {
    if (p->n_kf != 1 || o->lm_nshard > 1) return false;
    for (int j = 0; j < p->n_pt; j++) if (p->pt_host[j] >= 0) return false;
    for (int j = 0; j < p->n_text; j++) if (p->text_host[j] >= 0) return false;
    return true;
}
inline void build_plan_single_frame(const tsba_problem *p, const tsba_options *o, int L, HostPlan &P) {
    P.recycle(); P.level = L;
    const int n_pt = p->n_pt, n_text = p->n_text, n_sc = p->n_sobs[L], n_tg = o->use_text ? p->n_tobs : 0, n_pair = (n_sc > 0 || n_tg > 0) ? 1 : 0;
    if (n_pair) { P.pair_i.assign(1, 0); P.pair_h.assign(1, -1); P.pair_hpos.assign(1, -1); }
    P.pair_sc_off.assign((size_t)n_pair + 1, 0); P.pair_tg_off.assign((size_t)n_pair + 1, 0);
    if (n_pair) { P.pair_sc_off[1] = n_sc; P.pair_tg_off[1] = n_tg; }
    P.sc_obs.resize((size_t)n_sc); P.sc_kf.resize((size_t)n_sc); P.sc_pt.resize((size_t)n_sc); P.sc_flag.resize((size_t)n_sc); P.sc_slot.assign((size_t)n_sc, -1); P.sc_uv.resize(2*(size_t)n_sc);
    for (int c = 0; c < n_sc; c++) { P.sc_obs[(size_t)c] = c; P.sc_kf[(size_t)c] = p->sobs_kf[L][c]; P.sc_pt[(size_t)c] = p->sobs_pt[L][c]; P.sc_flag[(size_t)c] = p->sobs_flag[L][c]; }
    if (n_sc > 0) memcpy(P.sc_uv.data(), p->sobs_uv0[L], 2*(size_t)n_sc*sizeof(double));
    P.tg_tobs.resize((size_t)n_tg); P.tg_kf.resize((size_t)n_tg); P.tg_text.resize((size_t)n_tg); P.tg_pair.assign((size_t)n_tg, 0); P.tg_slot.assign((size_t)n_tg, -1);
    P.pair_tg.resize((size_t)n_tg); P.tg_ppos.resize((size_t)n_tg); P.tg_rec.resize(8*(size_t)n_tg);
    for (int g = 0; g < n_tg; g++) { const int j = p->tobs_text[g];
        P.tg_tobs[(size_t)g] = g; P.tg_kf[(size_t)g] = p->tobs_kf[g]; P.tg_text[(size_t)g] = j; P.pair_tg[(size_t)g] = g; P.tg_ppos[(size_t)g] = g;
        int32_t *r = &P.tg_rec[8*(size_t)g];
        r[0] = g; r[1] = p->tobs_kf[g]; r[2] = j; r[3] = p->text_host[j]; r[4] = -1;
        r[5] = p->tfeat_off[L] ? p->tfeat_off[L][j] : 0; r[6] = p->tfeat_off[L] ? p->tfeat_off[L][j+1] : 0; r[7] = p->tobs_fgood_off[g];
        for (int f = r[5]; f < r[6]; f++) { P.pf_g.push_back(g); P.pf_f.push_back(f); } }
    P.pls_off.assign((size_t)n_pt + 1, 0); P.tls_off.assign((size_t)n_text + 1, 0);
    P.pt_pose6.assign(6*(size_t)n_pt, 0); P.pt_pair4.assign(6*(size_t)n_pt, 0); P.tx_pair8.assign(8*(size_t)n_text, 0);
    P.sb_a.assign(1, 0); P.sb_b.assign(1, 0); P.sb_pab.assign(1, -1); P.sb_pba.assign(1, -1);
    P.sb_pt_off.assign(2, 0); P.sb_tx_off.assign(2, 0);
    P.pose_t_off.assign(2, 0); P.pose_t_off[1] = n_pair; if (n_pair) P.pose_t.assign(1, 0);
    P.pose_h_off.assign(2, 0); P.pose_ps_off.assign(2, 0); P.pose_ts_off.assign(2, 0);
    P.sb_rng.assign(4, 0); P.sb_rng[1] = n_pair;
}

// far_max_blocks > 0: maps whose envelope no band solver reaches may be split into a band of at most far_max_blocks pose blocks + long-range
// blocks (HostPlan::far_B); far_force: take the split whenever the map is eligible, without trying the keyframe reordering first
inline void build_plan(const tsba_problem *p, const tsba_options *o, int L, HostPlan &P, bool dbg_plan = false, bool allow_reorder = true, int ring_max_blocks = 0,
                       int far_max_blocks = 0, bool far_force = false, bool dev_pairs = false) {
    auto tp0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) { if (!dbg_plan) return; auto t = std::chrono::steady_clock::now(); fprintf(stderr, "[build_plan] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(t - tp0).count()); tp0 = t; };
    P.recycle();
    P.level = L;
    const int n_kf = p->n_kf, n_pt = p->n_pt, n_text = p->n_text, n_obs = p->n_sobs[L];
    int T = 1;
    if ((size_t)n_obs > 100000) T = std::max(1, std::min(std::min(16, plan_usable_cpus()), (int)(std::thread::hardware_concurrency()/2)));
    if (tsba_plan_threads > 0) T = tsba_plan_threads;
    if ((int64_t)n_kf*(n_kf + 1) > ((int64_t)1 << 31)) T = 1;       // (the key sets below are shared bitmaps only up to 2^31 keys)
    PlanPool pool(T, tsba_plan_pin != 0 || o->host_plan_pin != 0);
    T = pool.T;
    typedef PlanCand Cand;
    PlanScratch &SC = P.scratch; SC.threads(T);
    std::vector<std::vector<Cand> > &cs = SC.cs;                      // scene candidates in observation order, thread-major
    for (auto &v : cs) v.clear();
    // The envelope of the reduced camera matrix is a property of the WHOLE problem: every rank of a sharded solve must derive the
    // same band (storage layout, solver choice, exchange counts) -- so it is taken from all observations, before the shard filter.
    // A landmark couples every pair of its poses (observers + host): column lo = min pose reaches down to hi = max pose; the fill
    // closure below (running maximum) covers the poses in between.
    std::vector<int> reach(n_kf);
    for (int a = 0; a < n_kf; a++) reach[a] = a;
    std::vector<int32_t> lm_lo((size_t)n_pt + n_text, n_kf), lm_hi((size_t)n_pt + n_text, -1);
    std::vector<std::vector<int32_t> > &tlo = SC.tlo, &thi = SC.thi;  // per thread (points): a shared array would bounce its lines between the threads' caches on every observation
    const bool keep_inc = n_kf > 64 && allow_reorder;                 // large maps: the landmarks' pose lists may be needed (reordering)
    // ---- pairs: key = kf*(n_kf+1) + (host+1)
    auto key_of = [&](int kf, int host) { return (int64_t)kf*(n_kf + 1) + (host + 1); };
    KeyIndex pk; pk.begin((int64_t)n_kf*(n_kf + 1));
    const bool sharded = o->lm_nshard > 1;
    pool.run([&](int t) { size_t s0, s1; pool.range((size_t)n_obs, t, s0, s1); std::vector<Cand> &v = cs[(size_t)t]; v.reserve(s1 - s0);
        int32_t *lo_t = lm_lo.data(), *hi_t = lm_hi.data();
        if (T > 1) { tlo[(size_t)t].assign((size_t)n_pt, n_kf); thi[(size_t)t].assign((size_t)n_pt, -1); lo_t = tlo[(size_t)t].data(); hi_t = thi[(size_t)t].data(); }
        for (size_t s = s0; s < s1; s++) {
            const int kf = p->sobs_kf[L][s], pt = p->sobs_pt[L][s], host = p->pt_host[pt];
            if (host >= 0 && host == kf) continue;                    // optimizer.cc:1393 "Host != Target"
            if (host >= 0) { lo_t[pt] = std::min(lo_t[pt], std::min(kf, host)); hi_t[pt] = std::max(hi_t[pt], std::max(kf, host)); }   // (a frozen landmark couples nothing)
            if (sharded && tsba_shard_of(host, kf, n_kf, o->lm_nshard) != o->lm_shard) continue;   // multi-GPU: landmark shard
            v.push_back({ (int)s, kf, pt, host >= 0 ? host : -1 });
            if (T > 1) pk.add_mt(key_of(kf, v.back().host)); else pk.add(key_of(kf, v.back().host)); } });
    if (T > 1) pool.run([&](int t) { size_t j0, j1; pool.range((size_t)n_pt, t, j0, j1);
        for (size_t j = j0; j < j1; j++) { int lo = n_kf, hi = -1; for (int u = 0; u < T; u++) { lo = std::min(lo, tlo[(size_t)u][j]); hi = std::max(hi, thi[(size_t)u][j]); } lm_lo[j] = lo; lm_hi[j] = hi; } });
    struct Grp { int tobs, kf, text, host; };
    std::vector<Grp> gs;
    if (o->use_text) for (int t = 0; t < p->n_tobs; t++) {
        int kf = p->tobs_kf[t], j = p->tobs_text[t], host = p->text_host[j];
        if (host >= 0 && host == kf) continue;                    // optimizer.cc:1484
        if (host >= 0) { lm_lo[n_pt + j] = std::min(lm_lo[n_pt + j], std::min(kf, host)); lm_hi[n_pt + j] = std::max(lm_hi[n_pt + j], std::max(kf, host)); }
        if (sharded && tsba_shard_of(host, kf, n_kf, o->lm_nshard) != o->lm_shard) continue;
        gs.push_back({ t, kf, j, host >= 0 ? host : -1 });
        pk.add(key_of(kf, gs.back().host));
    }
    for (size_t lm = 0; lm < lm_lo.size(); lm++) if (lm_hi[lm] >= 0) reach[lm_lo[lm]] = std::max(reach[lm_lo[lm]], lm_hi[lm]);
    const int n_pair = pk.finish();
    const std::vector<int64_t> &keys = pk.keys;
    auto pair_of = [&](int kf, int host) { return pk.id(key_of(kf, host)); };
    P.pair_i.resize(n_pair); P.pair_h.resize(n_pair);
    for (int q = 0; q < n_pair; q++) { P.pair_i[q] = (int)(keys[q]/(n_kf + 1)); P.pair_h[q] = (int)(keys[q] % (n_kf + 1)) - 1; }
    lap("candidates + pair keys");
    // ---- sort scene candidates by pair (stable: keeps the reference order inside a pair)
    size_t n_sc_ = 0; for (auto &v : cs) n_sc_ += v.size();
    const int n_sc = (int)n_sc_;
    std::vector<int32_t> &sc_pair = SC.sc_pair; sc_pair.resize((size_t)n_sc);      // pair of a sorted candidate
    {   BucketPlacer bp(pool, n_pair, SC);
        pool.run([&](int t) { const std::vector<Cand> &v = cs[(size_t)t]; bp.begin(t, v.size()); for (const Cand &k : v) bp.count(t, pair_of(k.kf, k.host)); });
        bp.offsets(P.pair_sc_off);
        P.sc_obs.resize(n_sc); P.sc_kf.resize(n_sc); P.sc_pt.resize(n_sc); P.sc_flag.resize(n_sc); P.sc_slot.assign(n_sc, -1); P.sc_uv.resize(2*(size_t)n_sc);
        pool.run([&](int t) { size_t e = 0;
            for (const Cand &k : cs[(size_t)t]) { const int q = bp.key(t, e); const int c = bp.next(t, e);
                P.sc_obs[c] = k.obs; P.sc_kf[c] = k.kf; P.sc_pt[c] = k.pt; P.sc_flag[c] = p->sobs_flag[L][k.obs]; sc_pair[(size_t)c] = q;
                P.sc_uv[2*(size_t)c] = p->sobs_uv0[L][2*(size_t)k.obs]; P.sc_uv[2*(size_t)c+1] = p->sobs_uv0[L][2*(size_t)k.obs+1]; } });
    }
    lap("candidate order");
    // ---- text groups and their pair CSR
    const int n_tg = (int)gs.size();
    P.tg_tobs.resize(n_tg); P.tg_kf.resize(n_tg); P.tg_text.resize(n_tg); P.tg_pair.resize(n_tg); P.tg_slot.assign(n_tg, -1);
    P.pair_tg_off.assign(n_pair + 1, 0);
    for (int g = 0; g < n_tg; g++) {
        P.tg_tobs[g] = gs[g].tobs; P.tg_kf[g] = gs[g].kf; P.tg_text[g] = gs[g].text; P.tg_pair[g] = pair_of(gs[g].kf, gs[g].host);
        P.pair_tg_off[P.tg_pair[g] + 1]++;
    }
    for (int q = 0; q < n_pair; q++) P.pair_tg_off[q+1] += P.pair_tg_off[q];
    P.pair_tg.resize(n_tg);
    { std::vector<int> cur(P.pair_tg_off.begin(), P.pair_tg_off.end() - 1);
      for (int g = 0; g < n_tg; g++) P.pair_tg[cur[P.tg_pair[g]]++] = g; }
    lap("  text groups");
    // ---- landmark slots (landmark-major; last slot of a landmark = its host pose): the sorted candidates placed by point
    {   BucketPlacer bl(pool, n_pt, SC);
        pool.run([&](int t) { size_t c0, c1; pool.range((size_t)n_sc, t, c0, c1); bl.begin(t, c1 - c0);
            for (size_t c = c0; c < c1; c++) { const int j = P.sc_pt[c]; bl.count(t, p->pt_host[j] >= 0 ? j : -1); } });
        bl.offsets(P.pls_off, 1);
        const int ns = P.pls_off[n_pt];
        P.pslot_pose.assign(ns, -1); P.pslot_pair.assign(ns, -1); P.pslot_lm.assign(ns, -1);
        pool.run([&](int t) { size_t c0, c1; pool.range((size_t)n_sc, t, c0, c1); size_t e = 0;
            for (size_t c = c0; c < c1; c++) { const int s = bl.next(t, e); if (s < 0) continue;
                P.sc_slot[c] = s; P.pslot_pose[s] = P.sc_kf[c]; P.pslot_pair[s] = sc_pair[c]; P.pslot_lm[s] = P.sc_pt[c]; } });
        pool.run([&](int t) { size_t j0, j1; pool.range((size_t)n_pt, t, j0, j1);
            for (size_t j = j0; j < j1; j++) if (P.pls_off[j+1] > P.pls_off[j]) { const int s = P.pls_off[j+1] - 1; P.pslot_pose[s] = p->pt_host[j]; P.pslot_lm[s] = (int)j; } });
    }
    lap("  point slots");
    {
        std::vector<int> cnt(n_text + 1, 0);
        for (int g = 0; g < n_tg; g++) if (p->text_host[P.tg_text[g]] >= 0) cnt[P.tg_text[g]]++;
        P.tls_off.assign(n_text + 1, 0);
        for (int j = 0; j < n_text; j++) P.tls_off[j+1] = P.tls_off[j] + (cnt[j] > 0 ? cnt[j] + 1 : 0);
        int ns = P.tls_off[n_text];
        P.tslot_pose.assign(ns, -1); P.tslot_pair.assign(ns, -1); P.tslot_lm.assign(ns, -1);
        std::vector<int> cur(P.tls_off.begin(), P.tls_off.end() - 1);
        for (int g = 0; g < n_tg; g++) { int j = P.tg_text[g]; if (p->text_host[j] < 0) continue;
            int s = cur[j]++; P.tg_slot[g] = s; P.tslot_pose[s] = P.tg_kf[g]; P.tslot_pair[s] = P.tg_pair[g]; P.tslot_lm[s] = j; }
        for (int j = 0; j < n_text; j++) if (cnt[j] > 0) { int s = P.tls_off[j+1] - 1; P.tslot_pose[s] = p->text_host[j]; P.tslot_lm[s] = j; }
    }
    lap("  text slots");
    {   // the point slots of every pose (0.57 M slots at 5000 keyframes)
        const size_t n_ps = (size_t)P.n_pslot();
        BucketPlacer bq(pool, n_kf, SC);
        pool.run([&](int t) { size_t s0, s1; pool.range(n_ps, t, s0, s1); bq.begin(t, s1 - s0); for (size_t s = s0; s < s1; s++) bq.count(t, P.pslot_pose[s]); });
        bq.offsets(P.pose_ps_off);
        P.pose_ps.resize(n_ps); P.pose_ps_lm.resize(n_ps);
        pool.run([&](int t) { size_t s0, s1; pool.range(n_ps, t, s0, s1); size_t e = 0;
            for (size_t s = s0; s < s1; s++) { const int at = bq.next(t, e); P.pose_ps[at] = (int)s; P.pose_ps_lm[at] = P.pslot_lm[s]; } });
    }
    lap("groups + landmark slots");
    // ---- reduced-system blocks: key = a*n_kf + b (a <= b)
    auto bkey = [&](int a, int b) { return (int64_t)a*n_kf + b; };
    // slot pairs (s1, s2) of every landmark with pose(s1) <= pose(s2), in landmark-major generation order; they are visited three times
    // (mark the block, count per block, place) instead of being materialised and sorted: 2 M pairs at 5000 keyframes.  Thread t takes the
    // landmarks [lo[t], lo[t+1]): about the same number of slots each.
    auto split_landmarks = [&](const std::vector<int32_t> &loff, int n_lm) { std::vector<int> lo((size_t)T + 1, 0); const size_t n_slot = n_lm > 0 ? (size_t)loff[n_lm] : 0;
        for (int t = 1; t < T; t++) lo[t] = (int)(std::lower_bound(loff.begin(), loff.begin() + n_lm + 1, (int32_t)(n_slot*t/T)) - loff.begin());
        lo[T] = n_lm; return lo; };
    // cl (may be null): cluster of every slot within its landmark (band + long-range split, below): only pairs of ONE cluster feed the band
    auto range_pairs = [&](const std::vector<int32_t> &loff, const std::vector<int32_t> &pose, const int32_t *cl, int j0, int j1, auto &&f) {
        for (int j = j0; j < j1; j++) for (int s1 = loff[j]; s1 < loff[j+1]; s1++) for (int s2 = loff[j]; s2 < loff[j+1]; s2++) {
            const int a = pose[s1], b2 = pose[s2]; if (a > b2 || (cl && cl[s1] != cl[s2])) continue; f(bkey(a, b2), s1, s2); } };
    const std::vector<int> lo_pt = split_landmarks(P.pls_off, n_pt), lo_tx = split_landmarks(P.tls_off, n_text);
    // pair_far (may be null): (target, host) pairs whose cross term H_th lies outside the band part (the target is not in the host's cluster)
    auto build_blocks = [&](const int32_t *cl_pt, const int32_t *cl_tx, const char *pair_far) {
    KeyIndex bk; bk.begin((int64_t)n_kf*n_kf);
    // marking the blocks of the point slots: by POSE (thread t the poses [a0, a1) with about the same number of slots: every slot of pose a, every
    // slot of its landmark at a pose b >= a) -- a thread then writes its own rows of the key bitmap only; landmark-major, every thread wrote
    // every row, and each of the 45 k first-time writes took the line out of fifteen other caches
    if (n_kf <= 64) { /* small windows: every (a, b) is a block (added below) -- nothing to mark */ }
    else if (T > 1 && tsba_plan_mark_mt) {
        const size_t n_ps = (size_t)P.n_pslot();
        pool.run([&](int t) {
            const int a0 = (int)(std::lower_bound(P.pose_ps_off.begin(), P.pose_ps_off.end(), (int32_t)(n_ps*(size_t)t/(size_t)T)) - P.pose_ps_off.begin());
            const int a1 = t + 1 == T ? n_kf : (int)(std::lower_bound(P.pose_ps_off.begin(), P.pose_ps_off.end(), (int32_t)(n_ps*(size_t)(t + 1)/(size_t)T)) - P.pose_ps_off.begin());
            for (int a = std::min(a0, n_kf); a < std::min(a1, n_kf); a++)
                for (int x = P.pose_ps_off[a]; x < P.pose_ps_off[a+1]; x++) { const int j = P.pose_ps_lm[x], sa = P.pose_ps[x];
                    for (int s2 = P.pls_off[j]; s2 < P.pls_off[j+1]; s2++) { const int b = P.pslot_pose[s2]; if (b >= a && (!cl_pt || cl_pt[sa] == cl_pt[s2])) bk.add_mt(bkey(a, b)); } } });
    } else range_pairs(P.pls_off, P.pslot_pose, cl_pt, 0, n_pt, [&](int64_t k, int, int) { bk.add(k); });
    if (n_kf > 64) range_pairs(P.tls_off, P.tslot_pose, cl_tx, 0, n_text, [&](int64_t k, int, int) { bk.add(k); });
    for (int q = 0; q < n_pair; q++) { int i = P.pair_i[q], h = P.pair_h[q]; bk.add(bkey(i, i));
        if (h >= 0) { bk.add(bkey(h, h)); if (!pair_far || !pair_far[q]) bk.add(bkey(std::min(i, h), std::max(i, h))); } }
    if (n_kf <= 64) for (int a = 0; a < n_kf; a++) for (int b = a; b < n_kf; b++) bk.add(bkey(a, b));   // small windows: dense S, no memset
    lap("slot pair generation");
    const int n_sb = bk.finish();
    lap("block key index");
    const std::vector<int64_t> &bkeys = bk.keys;
    auto blk_of = [&](int64_t k) { return bk.id(k); };
    P.sb_a.resize(n_sb); P.sb_b.resize(n_sb); P.sb_pab.assign(n_sb, -1); P.sb_pba.assign(n_sb, -1);
    for (int q = 0; q < n_sb; q++) { P.sb_a[q] = (int)(bkeys[q]/n_kf); P.sb_b[q] = (int)(bkeys[q] % n_kf); }
    for (int q = 0; q < n_pair; q++) { int i = P.pair_i[q], h = P.pair_h[q]; if (h < 0 || (pair_far && pair_far[q])) continue;
        int bl = blk_of(bkey(std::min(i, h), std::max(i, h)));
        if (i < h) P.sb_pab[bl] = q; else P.sb_pba[bl] = q; }     // pab: target = a, host = b;  pba: target = b, host = a
    // the slot pairs placed by block, stable: the landmark-major generation order is kept within a block
    auto fill_tri = [&](const std::vector<int32_t> &loff, const std::vector<int32_t> &pose, const int32_t *cl, const std::vector<int32_t> &lm_of, const std::vector<int> &lo,
                        std::vector<int32_t> &off, std::vector<int32_t> &s1v, std::vector<int32_t> &s2v, std::vector<int32_t> &lmv) {
        const int n_lm = lo[(size_t)T];
        const size_t n_slot = n_lm > 0 ? (size_t)loff[n_lm] : 0;
        if (n_slot == 0) { off.assign((size_t)n_sb + 1, 0); s1v.clear(); s2v.clear(); lmv.clear(); return; }
        if (dev_pairs && &off == &P.sb_pt_off) {                       // the device builds these lists (tsba_devplan.h): count only
            std::vector<int64_t> part((size_t)T, 0); std::vector<char> dup((size_t)T, 0);
            pool.run([&](int t) { int64_t n = 0;
                // the device walk relies on ONE slot per (landmark, pose) -- what the reference's maps have; a landmark listed twice at a keyframe
                // (two slots at one pose) keeps the host lists, which take any input.  A landmark's observer slots come in ascending keyframe order (its candidates were
                // placed in pair order) with its host last: strictly ascending observers that differ from the host = no pose twice, and then the pairs with
                // pose(s1) <= pose(s2) are k (k + 1) / 2 whatever the order -- one pass over the slots instead of two over all their pairs (0.08 of the 0.33 ms a C4
                // window's first plan takes on the calling thread).  Anything else, and the cluster split, through the pairs themselves.
                for (int j = lo[t]; j < lo[t+1] && !dup[(size_t)t]; j++) { const int a = loff[j], b = loff[j+1], k = b - a; if (k <= 0) continue;
                    bool plain = cl == nullptr; for (int s1 = a; plain && s1 + 2 < b; s1++) plain = pose[s1] < pose[s1 + 1];
                    for (int s1 = a; plain && s1 + 1 < b; s1++) plain = pose[s1] != pose[b - 1];
                    if (plain) { n += (int64_t)k*(k + 1)/2; continue; }
                    range_pairs(loff, pose, cl, j, j + 1, [&](int64_t, int, int) { n++; });
                    for (int s1 = a; s1 < b; s1++) for (int s2 = s1 + 1; s2 < b; s2++) if (pose[s1] == pose[s2]) dup[(size_t)t] = 1; }
                part[(size_t)t] = n; });
            bool any_dup = false; for (char d : dup) any_dup |= d != 0;
            if (!any_dup) {
                int64_t tot = 0; for (int64_t v : part) tot += v;
                P.dev_pt_pairs = tot; off.clear(); s1v.clear(); s2v.clear(); lmv.clear();
                if (cl) P.cl_pt_dev.assign(cl, cl + n_slot); else P.cl_pt_dev.clear();
                lap("  slot pairs: counted for the device build");
                return; }
            P.dev_pt_pairs = -1; P.cl_pt_dev.clear(); }
        BucketPlacer bp(pool, n_sb, SC);
        pool.run([&](int t) { bp.begin(t, 4*n_slot/(size_t)T + 1024); range_pairs(loff, pose, cl, lo[t], lo[t+1], [&](int64_t k, int, int) { bp.count(t, blk_of(k)); }); });
        lap("  slot pairs: count");
        bp.offsets(off);
        const size_t tot = (size_t)off[n_sb];
        s1v.resize(tot); s2v.resize(tot); lmv.resize(tot);
        lap("  slot pairs: offsets + resize");
        pool.run([&](int t) { size_t e = 0; range_pairs(loff, pose, cl, lo[t], lo[t+1], [&](int64_t, int s1, int s2) { const int at = bp.next(t, e); s1v[at] = s1; s2v[at] = s2; lmv[at] = lm_of[s1]; }); });
    };
    fill_tri(P.pls_off, P.pslot_pose, cl_pt, P.pslot_lm, lo_pt, P.sb_pt_off, P.sb_pt_s1, P.sb_pt_s2, P.sb_pt_lm);      // (lm: saves one dependent gather in k_schur)
    fill_tri(P.tls_off, P.tslot_pose, cl_tx, P.tslot_lm, lo_tx, P.sb_tx_off, P.sb_tx_s1, P.sb_tx_s2, P.sb_tx_lm);
    lap("slot pairs by block");
    };
    build_blocks(nullptr, nullptr, nullptr);
    const int n_sb = P.n_sb();
    // ---- per-pose lists
    auto csr = [&](int n, const std::vector<std::pair<int,int>> &items, std::vector<int32_t> &off, std::vector<int32_t> &val) {
        off.assign(n + 1, 0); val.resize(items.size());
        for (auto &it : items) off[it.first + 1]++;
        for (int k = 0; k < n; k++) off[k+1] += off[k];
        std::vector<int> cur(off.begin(), off.end() - 1);
        for (auto &it : items) val[cur[it.first]++] = it.second;
    };
    P.pt_pose6.resize(6*(size_t)n_pt); P.pt_pair4.resize(6*(size_t)n_pt);
    pool.run([&](int t) { size_t j0, j1; pool.range((size_t)n_pt, t, j0, j1);
        for (size_t j = j0; j < j1; j++) { const int o = P.pls_off[j], e = P.pls_off[j+1];       // observer slots [o, e-1), host slot e-1
            for (int u = 0; u < 6; u++) P.pt_pose6[6*j + u] = e > o ? P.pslot_pose[std::min(o + u, e - 1)] : 0;
            for (int u = 0; u < 6; u++) P.pt_pair4[6*j + u] = e - 1 > o ? P.pslot_pair[std::min(o + u, e - 2)] : 0; } });
    P.tx_pair8.resize(8*P.tls_off.size() > 8 ? 8*(P.tls_off.size() - 1) : 0);
    for (size_t j = 0; j + 1 < P.tls_off.size(); j++) { const int o = P.tls_off[j], e = P.tls_off[j+1];
        for (int u = 0; u < 8; u++) P.tx_pair8[8*j + u] = e - 1 > o ? P.tslot_pair[std::min(o + u, e - 2)] : 0; }
    P.tg_ppos.assign(n_tg, 0);
    for (size_t k = 0; k < P.pair_tg.size(); k++) P.tg_ppos[P.pair_tg[k]] = (int)k;
    P.tg_rec.resize(8*(size_t)n_tg);
    for (int g = 0; g < n_tg; g++) {
        const int tb = P.tg_tobs[g], j = P.tg_text[g];
        int32_t *r = &P.tg_rec[8*(size_t)g];
        r[0] = tb; r[1] = P.tg_kf[g]; r[2] = j; r[3] = p->text_host[j]; r[4] = P.tg_slot[g];
        r[5] = p->tfeat_off[L] ? p->tfeat_off[L][j] : 0; r[6] = p->tfeat_off[L] ? p->tfeat_off[L][j+1] : 0; r[7] = p->tobs_fgood_off[tb];
    }
    if (n_kf == 1)
        for (int g = 0; g < n_tg; g++)
            for (int f = P.tg_rec[8*(size_t)g + 5]; f < P.tg_rec[8*(size_t)g + 6]; f++) { P.pf_g.push_back(g); P.pf_f.push_back(f); }
    {   // envelope of S: column a reaches down to its last coupled pose; Cholesky fill closes the profile under the running
        // maximum (a column inherits the reach of every earlier column that reaches it).  Compressing the fixed poses out
        // (device side) only shrinks distances, so this is an upper bound for the system that is actually factored.
        std::vector<int> &cm = reach;                             // (all ranks' observations: see the top of this function)
        for (int q = 0; q < n_sb; q++) cm[P.sb_a[q]] = std::max(cm[P.sb_a[q]], (int)P.sb_b[q]);
        auto closed_bw = [&](const std::vector<int> &c) { int run = -1, bw = 0;
            for (int k = 0; k < n_kf; k++) { const int reach = (run >= k) ? std::max(c[k], run) : c[k]; run = std::max(run, reach); bw = std::max(bw, reach - k); }
            return bw; };
        P.bw_pose = closed_bw(cm);
        // A wide envelope (beyond the streaming band solvers: 26 pose blocks) on a large map: try the reverse Cuthill-McKee order of the
        // co-visibility graph (all ranks' observations, so that every rank of a sharded solve derives the same order)
        if (keep_inc && P.bw_pose > 26) {
            // poses of every landmark (observers + host), sorted and distinct, as one CSR list -- from ALL observations (a second pass over
            // them: only maps with a wide envelope come here)
            const size_t n_lm = (size_t)n_pt + n_text;
            std::vector<int32_t> po_off(n_lm + 1, 0), po_n(n_lm, 0), po;
            auto each_incidence = [&](auto &&f) {
                for (int s = 0; s < n_obs; s++) { const int kf = p->sobs_kf[L][s], pt = p->sobs_pt[L][s], host = p->pt_host[pt]; if (host >= 0 && host != kf) f((size_t)pt, kf); }
                if (o->use_text) for (int t = 0; t < p->n_tobs; t++) { const int kf = p->tobs_kf[t], j = p->tobs_text[t], host = p->text_host[j]; if (host >= 0 && host != kf) f((size_t)n_pt + j, kf); } };
            each_incidence([&](size_t lm, int) { po_off[lm + 1]++; });
            for (size_t j = 0; j < n_lm; j++) { if (lm_hi[j] >= 0) po_off[j + 1]++; po_off[j + 1] += po_off[j]; }
            po.resize((size_t)po_off[n_lm]);
            for (size_t j = 0; j < n_lm; j++) if (lm_hi[j] >= 0) po[(size_t)po_off[j] + po_n[j]++] = j < (size_t)n_pt ? p->pt_host[j] : p->text_host[j - n_pt];
            each_incidence([&](size_t lm, int kf) { po[(size_t)po_off[lm] + po_n[lm]++] = kf; });
            for (size_t j = 0; j < n_lm; j++) { if (po_n[j] < 2) continue; int32_t *b = &po[(size_t)po_off[j]];
                std::sort(b, b + po_n[j]); po_n[j] = (int32_t)(std::unique(b, b + po_n[j]) - b); }
            struct PoseList { const int32_t *p; size_t n; size_t size() const { return n; } int operator[](size_t i) const { return p[i]; } int back() const { return p[n - 1]; }
                              const int32_t *begin() const { return p; } const int32_t *end() const { return p + n; } };
            struct PoseLists { const std::vector<int32_t> &off, &cnt, &val;
                struct It { const PoseLists *L; size_t j; bool operator!=(const It &o) const { return j != o.j; } void operator++() { j++; }
                            PoseList operator*() const { return PoseList{ L->val.data() + L->off[j], (size_t)L->cnt[j] }; } };
                It begin() const { return It{ this, 0 }; } It end() const { return It{ this, cnt.size() }; } };
            const PoseLists poses_of{ po_off, po_n, po };
            std::vector<std::vector<int> > nb((size_t)n_kf);
            // A single loop closure between the END and the START of the trajectory makes the graph a ring: every landmark's poses
            // fit an arc of a few keyframes, some arcs wrap from the last keyframes to the first.  Unrolled past its end (a wrapping
            // landmark's early poses k count as n_kf + k) the matrix is a band again -- bw_pose of the open chain, not twice that
            // as under any reordering -- and the solver ties the ghost rows to the first poses at the root of its separator tree.
            if (ring_max_blocks > 0) {
                const int RB = ring_max_blocks, nu = n_kf + RB + 1;
                std::vector<int> cu(nu); for (int k = 0; k < nu; k++) cu[k] = k;
                bool ok = true, wraps = false;
                for (const PoseList v : poses_of) { if (v.size() < 2) continue;
                    int best = v[0] + n_kf - v.back(); size_t at = 0;                 // largest cyclic gap (at = 0: the wrap between last and first)
                    for (size_t x = 1; x < v.size(); x++) if (v[x] - v[x - 1] > best) { best = v[x] - v[x - 1]; at = x; }
                    const int lo = at ? v[at] : v[0], hi = at ? v[at - 1] + n_kf : v.back();
                    if (hi - lo > RB || hi >= n_kf + RB) { ok = false; break; }
                    if (at) wraps = true;
                    cu[lo] = std::max(cu[lo], hi); }
                if (ok && wraps) {
                    int run = -1, bwr = 0;
                    for (int k = 0; k < nu; k++) { const int r = (run >= k) ? std::max(cu[k], run) : cu[k]; run = std::max(run, r); bwr = std::max(bwr, r - k); }
                    if (bwr >= 1 && bwr <= RB && n_kf >= 4*(3*bwr + 2) + bwr) { P.ring = 1; P.bw_pose = bwr; P.ring_k0 = 0; }
                }
                // The usual loop closure: the last keyframes meet keyframe k0 > 0 -- a ring with a tail.  The rows stay in keyframe order; a landmark that
                // spans the closure counts its early poses (the loop's first ones) as n_kf + (k - k0): the ghost rows behind the last pose.
                if (!P.ring) {
                    int k0 = n_kf, klow = -1, ncl = 0; bool okc = true;
                    for (const PoseList v : poses_of) { if (v.size() < 2 || v.back() - v[0] <= RB) continue;
                        size_t at = 1; for (size_t x = 2; x < v.size(); x++) if (v[x] - v[x - 1] > v[at] - v[at - 1]) at = x;
                        if (v[at - 1] - v[0] > RB || v.back() - v[at] > RB || v[at] < n_kf - 2*RB) { okc = false; break; }
                        k0 = std::min(k0, v[0]); klow = std::max(klow, v[at - 1]); ncl++; }
                    if (okc && ncl > 0 && klow - k0 < RB && k0 >= 3*RB + 8 && n_kf - k0 >= 4*(3*RB + 2) + RB) {
                        const int nu2 = n_kf + RB + 1;
                        std::vector<int> c2(nu2); for (int k = 0; k < nu2; k++) c2[k] = k;
                        bool ok2 = true; int gmax = -1;
                        for (const PoseList v : poses_of) { if (v.size() < 2) continue;
                            int lo = v[0], hi = v.back();
                            if (hi - lo > RB) { lo = nu2; hi = -1;                       // spans the closure: its early poses as ghosts
                                for (int k : v) { const int u = (k >= k0 && k < k0 + RB) ? n_kf + (k - k0) : k; if (u != k) gmax = std::max(gmax, k - k0);
                                    lo = std::min(lo, u); hi = std::max(hi, u); } }
                            if (hi - lo > RB) { ok2 = false; break; }
                            c2[lo] = std::max(c2[lo], hi); }
                        if (ok2) {
                            int run = -1, bwr = 0;
                            for (int k = 0; k < nu2; k++) { const int r = (run >= k) ? std::max(c2[k], run) : c2[k]; run = std::max(run, r); bwr = std::max(bwr, r - k); }
                            if (bwr >= 1 && bwr <= RB && gmax < bwr) { P.ring = 1; P.bw_pose = bwr; P.ring_k0 = k0; }
                        }
                    }
                }
            }
            // Band + long-range blocks: most landmarks span a few consecutive keyframes (the local band), a small fraction couples keyframes far
            // apart (points seen again much later, several loop closures) -- no order of the keyframes makes that a band.  The local band is the
            // largest span that a noticeable share of the landmarks has; what lies outside goes to the long-range list.
            int far_B = 0;
            if (!P.ring && far_max_blocks > 0) {
                std::vector<int64_t> hist((size_t)far_max_blocks + 2, 0); int64_t n_lm2 = 0;
                for (const PoseList v : poses_of) { if (v.size() < 2) continue; n_lm2++; hist[(size_t)std::min(v.back() - v[0], far_max_blocks + 1)]++; }
                int Bm = 0; for (int s = 1; s <= far_max_blocks; s++) if (hist[(size_t)s]*500 >= n_lm2) Bm = s;
                // (a handful of landmarks just beyond that span -- an observer skipped here and there -- would scatter small blocks of E over the whole map: the
                // band takes them in unless that costs more than one or two extra pose blocks)
                for (int grow = 0; Bm >= 1 && Bm < far_max_blocks && grow < 2; grow++) { int64_t nbeyond = 0; for (int s = Bm + 1; s <= far_max_blocks; s++) nbeyond += hist[(size_t)s];
                    if (nbeyond <= 4) break; Bm++; }
                int64_t n_wide = 0; for (int s = Bm + 1; s <= far_max_blocks + 1; s++) n_wide += hist[(size_t)s];
                if (Bm >= 1 && n_wide*5 <= n_lm2 && n_kf >= 4*(3*Bm + 2) + Bm) far_B = Bm;
            }
            // clusters of a landmark's poses (ascending; a host slot may come last): 0 = the host and what follows it within far_B keyframes
            auto clusters = [&](auto &&pose_at, int n, int host, int32_t *cl) { int id = 0, start = 0;
                for (int x = 0; x < n; x++) { const int a = pose_at(x);
                    if (a >= host && a - host <= far_B) { cl[x] = 0; continue; }
                    if (id == 0 || a - start > far_B) { id++; start = a; }
                    cl[x] = id; } };
            // keyframes that the coupling outside the band part would touch: a few dozen = loop closures (the low-rank correction of tsba_wb.h applies)
            auto far_touched = [&]() { std::vector<char> hit((size_t)n_kf, 0); std::vector<int32_t> cl; size_t lm = 0; int n = 0;
                for (const PoseList v : poses_of) { const size_t j = lm++; if (v.size() < 2) continue;
                    const int host = j < (size_t)n_pt ? p->pt_host[j] : p->text_host[j - n_pt];
                    cl.resize(v.size()); clusters([&](int x) { return v[(size_t)x]; }, (int)v.size(), host, cl.data());
                    bool wide = false; for (size_t x = 0; x < v.size(); x++) wide |= cl[x] != 0;
                    if (wide) for (size_t x = 0; x < v.size(); x++) if (!hit[(size_t)v[x]]) { hit[(size_t)v[x]] = 1; n++; } }
                return n; };
            auto take_far = [&]() {
                P.far_B = far_B; P.bw_pose = far_B; P.kf_order.clear();
                // block positions of E: from ALL observations
                KeyIndex fk; fk.begin((int64_t)n_kf*n_kf);
                { std::vector<int32_t> cl; size_t lm = 0;
                  for (const PoseList v : poses_of) { const size_t j = lm++; if (v.size() < 2) continue;
                    const int host = j < (size_t)n_pt ? p->pt_host[j] : p->text_host[j - n_pt];
                    cl.resize(v.size()); clusters([&](int x) { return v[(size_t)x]; }, (int)v.size(), host, cl.data());
                    bool wide = false; for (size_t x = 0; x < v.size(); x++) wide |= cl[x] != 0;
                    if (!wide) continue;
                    for (size_t x = 0; x < v.size(); x++) for (size_t y = x + 1; y < v.size(); y++) if (cl[x] != cl[y]) fk.add(bkey(v[x], v[y])); } }
                const int n_far = fk.finish();
                P.far_a.resize((size_t)n_far); P.far_b.resize((size_t)n_far); P.fb_id.resize((size_t)n_far);
                for (int q = 0; q < n_far; q++) { P.far_a[(size_t)q] = (int32_t)(fk.keys[(size_t)q]/n_kf); P.far_b[(size_t)q] = (int32_t)(fk.keys[(size_t)q] % n_kf); P.fb_id[(size_t)q] = q; }
                P.far_off.assign((size_t)n_kf + 1, 0);
                for (int q = 0; q < n_far; q++) { P.far_off[(size_t)P.far_a[(size_t)q] + 1]++; P.far_off[(size_t)P.far_b[(size_t)q] + 1]++; }
                for (int k = 0; k < n_kf; k++) P.far_off[(size_t)k + 1] += P.far_off[(size_t)k];
                P.far_ent.resize((size_t)P.far_off[(size_t)n_kf]);
                { std::vector<int32_t> cur(P.far_off.begin(), P.far_off.end() - 1);
                  for (int q = 0; q < n_far; q++) { P.far_ent[(size_t)cur[(size_t)P.far_a[(size_t)q]]++] = q << 1; P.far_ent[(size_t)cur[(size_t)P.far_b[(size_t)q]]++] = (q << 1) | 1; } }
                { int nu = 0; for (int k = 0; k < n_kf; k++) nu += P.far_off[(size_t)k + 1] > P.far_off[(size_t)k];
                  if (nu > 0 && nu <= WB_MAXKF_PLAN) { P.wb_idx.assign((size_t)n_kf, -1);
                      for (int k = 0; k < n_kf; k++) if (P.far_off[(size_t)k + 1] > P.far_off[(size_t)k]) { P.wb_idx[(size_t)k] = (int32_t)P.wb_kf.size(); P.wb_kf.push_back(k); } } }
                // this rank's slots: cluster of every slot, the band part M rebuilt from the pairs within a cluster, E from the pairs across clusters
                std::vector<int32_t> cl_pt((size_t)P.n_pslot(), 0), cl_tx((size_t)P.n_tslot(), 0);
                std::vector<char> wide_pt((size_t)n_pt, 0), wide_tx((size_t)n_text, 0), pair_far((size_t)n_pair, 0);
                pool.run([&](int t) { size_t j0, j1; pool.range((size_t)n_pt, t, j0, j1);
                    for (size_t j = j0; j < j1; j++) { const int o0 = P.pls_off[j], n = P.pls_off[j + 1] - o0; if (n < 2) continue;
                        clusters([&](int x) { return P.pslot_pose[(size_t)(o0 + x)]; }, n, p->pt_host[j], &cl_pt[(size_t)o0]);
                        for (int x = 0; x < n; x++) if (cl_pt[(size_t)(o0 + x)]) wide_pt[j] = 1; } });
                for (int j = 0; j < n_text; j++) { const int o0 = P.tls_off[j], n = P.tls_off[j + 1] - o0; if (n < 2) continue;
                    clusters([&](int x) { return P.tslot_pose[(size_t)(o0 + x)]; }, n, p->text_host[j], &cl_tx[(size_t)o0]);
                    for (int x = 0; x < n; x++) if (cl_tx[(size_t)(o0 + x)]) wide_tx[(size_t)j] = 1; }
                for (int q = 0; q < n_pair; q++) { const int i = P.pair_i[q], h = P.pair_h[q]; pair_far[(size_t)q] = h >= 0 && !(i >= h && i - h <= far_B); }
                build_blocks(cl_pt.data(), cl_tx.data(), pair_far.data());
                P.fb_pab.assign((size_t)n_far, -1); P.fb_pba.assign((size_t)n_far, -1);
                for (int q = 0; q < n_pair; q++) { if (!pair_far[(size_t)q]) continue; const int i = P.pair_i[q], h = P.pair_h[q];
                    const int bl = fk.id(bkey(std::min(i, h), std::max(i, h)));
                    if (i < h) P.fb_pab[(size_t)bl] = q; else P.fb_pba[(size_t)bl] = q; }
                auto fill_far = [&](const std::vector<int32_t> &loff, const std::vector<int32_t> &pose, const std::vector<int32_t> &cl, const std::vector<char> &wide, int n_lm,
                                    std::vector<int32_t> &off, std::vector<int32_t> &s1v, std::vector<int32_t> &s2v, std::vector<int32_t> &lmv) {
                    off.assign((size_t)n_far + 1, 0);
                    auto each = [&](auto &&f) { for (int j = 0; j < n_lm; j++) { if (!wide[(size_t)j]) continue;
                        for (int s1 = loff[j]; s1 < loff[j + 1]; s1++) for (int s2 = loff[j]; s2 < loff[j + 1]; s2++)
                            if (pose[s1] < pose[s2] && cl[(size_t)s1] != cl[(size_t)s2]) f(fk.id(bkey(pose[s1], pose[s2])), s1, s2, j); } };
                    each([&](int id, int, int, int) { off[(size_t)id + 1]++; });
                    for (int q = 0; q < n_far; q++) off[(size_t)q + 1] += off[(size_t)q];
                    s1v.resize((size_t)off[(size_t)n_far]); s2v.resize(s1v.size()); lmv.resize(s1v.size());
                    std::vector<int32_t> cur(off.begin(), off.end() - 1);
                    each([&](int id, int s1, int s2, int j) { const int at = cur[(size_t)id]++; s1v[(size_t)at] = s1; s2v[(size_t)at] = s2; lmv[(size_t)at] = j; }); };
                fill_far(P.pls_off, P.pslot_pose, cl_pt, wide_pt, n_pt, P.fb_pt_off, P.fb_pt_s1, P.fb_pt_s2, P.fb_pt_lm);
                fill_far(P.tls_off, P.tslot_pose, cl_tx, wide_tx, n_text, P.fb_tx_off, P.fb_tx_s1, P.fb_tx_s2, P.fb_tx_lm);
            };
            // (large maps: the reordering costs more host time than it can save; loop closures -- few keyframes touched -- are solved directly on the
            // band of the keyframe order, which beats the doubled band of any reordering)
            if (far_B > 0 && (far_force || n_kf > 2000 || (n_kf >= 600 && far_touched() <= WB_MAXKF_PLAN))) take_far();
            if (!P.ring && !P.far_B) {
            if ((int64_t)n_kf*n_kf <= ((int64_t)1 << 28)) {       // adjacency through a bitmap over pose pairs: set bits come out sorted and distinct
                std::vector<uint64_t> bm((((size_t)n_kf*n_kf) >> 6) + 1, 0);
                for (const PoseList v : poses_of) { if (v.size() < 2) continue;
                    for (size_t x = 0; x < v.size(); x++) for (size_t y = 0; y < v.size(); y++) if (x != y) { const size_t k = (size_t)v[x]*n_kf + v[y]; bm[k >> 6] |= (uint64_t)1 << (k & 63); } }
                for (size_t w = 0; w < bm.size(); w++) { uint64_t b = bm[w];
                    while (b) { const size_t k = (w << 6) + (size_t)__builtin_ctzll(b); b &= b - 1; nb[k/n_kf].push_back((int)(k % n_kf)); } }
            } else {
            for (const PoseList v : poses_of) { if (v.size() < 2) continue;
                for (size_t x = 0; x < v.size(); x++) for (size_t y = 0; y < v.size(); y++) if (x != y) nb[(size_t)v[x]].push_back(v[y]); }
            for (auto &v : nb) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }
            }
            std::vector<int32_t> order; rcm_order(n_kf, nb, order);
            std::vector<int> pos(n_kf), c2(n_kf);
            for (int i = 0; i < n_kf; i++) { pos[order[i]] = i; c2[i] = i; }
            for (const PoseList v : poses_of) { if (v.size() < 2) continue; int lo = n_kf, hi = -1;
                for (int k : v) { lo = std::min(lo, pos[k]); hi = std::max(hi, pos[k]); } c2[lo] = std::max(c2[lo], hi); }
            for (int q = 0; q < n_sb; q++) { const int pa = pos[P.sb_a[q]], pb = pos[P.sb_b[q]]; c2[std::min(pa, pb)] = std::max(c2[std::min(pa, pb)], std::max(pa, pb)); }
            const int bw2 = closed_bw(c2);
            if (far_B > 0 && bw2 > 26) take_far();                 // no order of the keyframes brings the envelope within the band solvers' reach
            else if (bw2*10 < P.bw_pose*7) { P.kf_order = order; P.bw_pose = bw2; }
            }
        }
    }
    std::vector<std::pair<int,int>> it_t, it_h, it_ts;
    for (int q = 0; q < n_pair; q++) { it_t.push_back({ P.pair_i[q], q }); if (P.pair_h[q] >= 0) it_h.push_back({ P.pair_h[q], q }); }
    for (int s = 0; s < P.n_tslot(); s++) it_ts.push_back({ P.tslot_pose[s], s });
    csr(n_kf, it_t, P.pose_t_off, P.pose_t); csr(n_kf, it_h, P.pose_h_off, P.pose_h);
    // pairs are sorted by (target, host): pose_t[k] == k, and the host-side products are stored host-major so that the per-pose
    // sums of both kinds read contiguous ranges
    P.pair_hpos.assign(n_pair, -1);
    for (size_t k = 0; k < P.pose_h.size(); k++) P.pair_hpos[P.pose_h[k]] = (int)k;
    P.sb_rng.assign(4*(size_t)P.n_sb(), 0);
    for (int b = 0; b < P.n_sb(); b++) if (P.sb_a[b] == P.sb_b[b]) { const int a = P.sb_a[b];
        P.sb_rng[4*b] = P.pose_t_off[a]; P.sb_rng[4*b + 1] = P.pose_t_off[a+1]; P.sb_rng[4*b + 2] = P.pose_h_off[a]; P.sb_rng[4*b + 3] = P.pose_h_off[a+1]; }
    csr(n_kf, it_ts, P.pose_ts_off, P.pose_ts);
    P.pose_ts_lm.resize(P.pose_ts.size());
    for (size_t k = 0; k < P.pose_ts.size(); k++) P.pose_ts_lm[k] = P.tslot_lm[P.pose_ts[k]];
    lap("per-pose lists");
}
