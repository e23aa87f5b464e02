// libtsba.so -- MI355X (gfx950) bundle adjustment / pose optimisation for TextSLAM's optimizer:: hot path.
// C ABI in include/tsba.h.  One translation unit: kernels + Levenberg-Marquardt driver + ABI.
//
// Kernel sequence of one LM iteration (all launched back-to-back on one stream, no host sync; the LM state machine
// lives in device memory and every kernel starts by reading it):
//   k_schur   one wave per 6x6 block of the reduced camera system S (gather over landmark slot pairs) + reduced gradient
//   k_solve   one workgroup: blocked Cholesky of S in LDS, pose step
//   k_back    landmark back-substitution, candidate parameters x (+) dx, step norm, model cost change
//   k_linearize<COST>  candidate cost (residuals + Huber only)
//   k_decide  step quality, trust-region update, accept / reject, convergence tests (Ceres 1.x semantics)
//   k_linearize<FULL>  (only after an accepted step) residual + analytic Jacobian + IRLS weight + per-pair / per-group
//                      J^T J, J^T r, J^T J_landmark sums: one wave per (target KF, host KF) pair of scene observations,
//                      one wave per (KF, text) observation of up to 64 photometric blocks
//   k_mid     per landmark V, b and the host-pose column of W; per pair the host-side products
//   k_postlin pose diagonal / gradient, Jacobi scaling, cost, gradient tolerance
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <chrono>
#include <thread>
#include <mutex>
#include <map>
#include <condition_variable>
#include <atomic>
#include <dlfcn.h>
#include <rccl/rccl.h>      // types only: the library is dlopen()ed by tsba_comm_init, single-GPU use never touches RCCL
#include "../../include/tsba.h"
#include "../../include/tsba_debug.h"
#include "tsba_device.h"
#include "tsraster.h"
#include "tsba_plan.h"

#include "tsba_types.h"
#include "tsba_kernels_lin.h"
#include "tsba_kernels_schur.h"
#include "tsba_solve.h"
#include "tsba_solve_la.h"
#include "tsba_chol.h"
#include "tsba_band.h"
#include "tsba_bandp.h"
#include "tsba_bandcr.h"
#include "tsba_bandcre.h"
#include "tsba_bandms.h"
#include "tsba_bandsv.h"
#include "tsba_bandmx.h"
#include "tsba_pcg.h"
#include "tsba_wb.h"
#include "tsba_pose.h"

#include "tsba_kernels_step.h"
#include "tsba_kernels_pass.h"
#include "tsba_devplan.h"
// ---- LDS hygiene (tests): what a kernel finds in LDS is whatever the last workgroup on that compute unit left -- its own context's kernels when the device is
// otherwise idle (the same bytes every run: a read of never-written LDS goes unnoticed), another context's next to it (TextSLAM extracts ORB features while
// a bundle adjustment runs).  tsba_debug_options.lds_poison = 1 / 2 / 3 fills the LDS of every compute unit with NaNs / 1e300 / 0x5a bytes before EVERY launch
// of a solve: a kernel that reads what it has not written shows up as a changed result (tests/test_gpu_polling.py).
__global__ __launch_bounds__(256) void k_lds_poison(int pattern, int ndbl) {
    extern __shared__ __attribute__((aligned(16))) double pz[];
    const double v = pattern == 1 ? __longlong_as_double(0x7ff8000000000000LL) : pattern == 2 ? 1e300 : __longlong_as_double(0x5a5a5a5a5a5a5a5aLL);
    for (int k = threadIdx.x; k < ndbl; k += 256) pz[k] = v;
    __syncthreads();
    for (int k = 0; k < 40; k++) __builtin_amdgcn_s_sleep(127);      // (~2 us: long enough for every compute unit to be handed one of the workgroups)
}
static thread_local int g_lds_poison = 0;
static void lds_poison_hook(hipStream_t st) {
    if (!g_lds_poison) return;
    static std::once_flag once; std::call_once(once, [] { hipFuncSetAttribute((const void *)k_lds_poison, hipFuncAttributeMaxDynamicSharedMemorySize, 160*1024 - 256); });
    hipLaunchKernelGGL(k_lds_poison, dim3(1024), dim3(256), 160*1024 - 256, st, g_lds_poison, (160*1024 - 256)/8);
}
#define LAUNCHK(kern, grid, block, lds, st, ...) do { lds_poison_hook(st); hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__); } while (0)
// ------------------------------------------------------------------------------------------------ host side
static int pose_grid(const LevelDev &D) { return std::max(1, (D.n_sc + 255)/256 + (D.n_pf + 31)/32); }    // workgroups of k_pose_iter
struct DevBuf {
    void *p = nullptr; size_t bytes = 0;
};
struct PassRecord { LmState st; };

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    bool uploaded = false;
    tsba_options opt;
    int n_kf = 0, n_pt = 0, n_text = 0, n_tobs = 0, n_sgood = 0, n_tfgood = 0, n_levels = 0;
    Work W;
    std::vector<HostPlan> hplan;                  // per level (only levels used by the options are built)
    std::vector<LevelDev> lev;
    std::vector<int> lev_built;                   // level l is on the device (plan lists, reference features, image table)
    // one-shot calls on small windows stage a level when its pass begins (the coarse passes run while the plan of level 0 is still being built)
    std::vector<std::thread> planners;            // planners[l]: the host thread that builds level l's plan (joined by stage_level)
    std::vector<int> lev_planned;                 // a plan of level l was started for this upload
    const tsba_problem *stage_p = nullptr;        // the caller's problem while levels may still be staged from it (one-shot calls only)
    std::vector<int> ic_slot; bool use_img_cache = false;      // plane cache: slot of every keyframe of this upload
    std::atomic<int> plan_done[TSBA_MAX_LEVELS];  // set by a plan thread when its plan is complete: a level is staged ahead of its pass only when that costs no wait
    hipStream_t copy_stream = nullptr; hipEvent_t ev_stage[TSBA_MAX_LEVELS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_upload = nullptr; bool upload_pending = false;      // one-shot calls end their upload without a synchronisation: the copy stream's first staging waits for this event (the slabs' clearing runs on the compute stream)
    bool stage_async = false; int lev_wait[TSBA_MAX_LEVELS] = {0, 0, 0, 0};     // levels staged during a solve go over the copy stream; their pass waits for the event
    // restart copies
    double *pose0 = nullptr, *rho0 = nullptr, *theta0 = nullptr; uint8_t *sgood0 = nullptr, *tobs_good0 = nullptr, *tfgood0 = nullptr;
    uint8_t *kf_initial = nullptr;
    LmState *st_host = nullptr;                   // pinned: per-pass snapshots
    LmState *st_log = nullptr;                    // device [MAX passes]
    int nb_back_max = 0;
    LmState *st_base = nullptr;                   // W.st / W.st_next are st_base and st_base + 1 in the order of the moment
    double *musig2[2] = {nullptr, nullptr}; int musig_sel = 0;      // mu / sigma of the text observations, two buffers: k_pass_end fills the next pass's while the outlier pass reads this one's
    int *ticket = nullptr;                        // k_pass_begin: arrival counter of its participation workgroups (zero between launches)
    unsigned long long *lin_ticket = nullptr, lin_base = 0;      // k_lin_mid: arrivals of its workgroups over all launches since the upload (device), the same count on the host
    size_t lds_limit = 0;
    int n_cu = 0;                                 // compute units of the device
    std::vector<std::pair<std::pair<const void *, size_t>, int>> occ_cache;      // (kernel, dynamic LDS) -> workgroups of it one compute unit holds
    std::vector<uint8_t *> img_dev[TSBA_MAX_LEVELS];
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // RCCL (global BA sharded over the GPUs of one node): one process per GPU, communicator created by tsba_comm_init
    bool pose_only = false;                       // one keyframe, every landmark frozen in its host: the fused pose-only LM kernel applies
    double *S_alloc = nullptr; size_t S_count = 0;
    int S_up = CH_NB;                               // band storage: columns stored right of the diagonal + 1
    bool S_stale = true;                            // band storage: entries of another pass may be left in S (cleared before the next assembly)
    double *S_xchg = nullptr; int xchg_wp = 0;      // multi-GPU, band storage: packed band rows for the exchange (k_band_pack)
    bool sep_cr = false;                  // separator system by cyclic reduction on the compact block pool (tsba_bandcr.h)
    int band_parts = 1; double *Lb = nullptr, *Tbuf = nullptr, *Bpart = nullptr, *Ssep = nullptr, *Lcol_sep = nullptr, *CRcontrib = nullptr, *CRfac = nullptr; int *CRgate = nullptr; int cre_epoch = 0; int nsep_ld = 0; Work Wsep;   // partitioned band solver (tsba_bandp.h)
    double *Lcol = nullptr; int band_stream = 0;  // streaming band solver (tsba_band.h): L by block column; 1 = every built level fits it     // storage behind W.S (dense or band)
    float *lbl_dev = nullptr, *lbl_host = nullptr; size_t lbl_cap = 0;   // text label image staging
    unsigned long long *hprog = nullptr; unsigned int pass_seq = 0;   // pinned progress word written by k_postlin / k_decide
    std::vector<struct Slab> slabs; int cur_slab = 0;
    int run_slab = 0; size_t run_off = 0, run_len = 0;   // pending contiguous host-to-device range
    int cur_bw_rows = 1 << 30;                     // band bound of the level being solved (set by launch_pass_init)
    // device cache of keyframe pyramid planes (tsba_problem.kf_id): one slot per keyframe, all levels of a keyframe contiguous
    struct ImgCache { uint8_t *dev = nullptr, *stage = nullptr; size_t stage_cap = 0, slot = 0, lvl_off[TSBA_MAX_LEVELS] = {0,0,0,0};
                      int w[TSBA_MAX_LEVELS] = {0,0,0,0}, h[TSBA_MAX_LEVELS] = {0,0,0,0}; unsigned lvl_mask = 0;
                      long long id[TSBA_IMG_CACHE_KF]; unsigned long long used[TSBA_IMG_CACHE_KF]; bool full[TSBA_IMG_CACHE_KF]; unsigned long long tick = 0;
                      long long hits = 0, misses = 0; } ic;
    WbBuf wb{}; Work Wk{}; double *wb_alloc = nullptr; size_t wb_bytes = 0;      // low-rank correction for loop closures (tsba_wb.h): its buffers, the k x k dense system as a second Work
    EcgBuf ecg{}; double *ecg_alloc = nullptr; size_t ecg_bytes = 0;      // enlarged conjugate gradients (tsba_pcg.h)
    bool pose_retry = false;                      // tsba_solve is running the pose-only solve again with a launch per LM step (after a poll give-up in k_pose_pass)
    bool pack_in_solve = false, packed = false;   // one-shot calls: tsba_solve packs the results behind its last kernel (enqueue_pack); packed: the block in dl_host is that solve's
    unsigned char *dl_dev = nullptr, *dl_host = nullptr; size_t dl_bytes = 0;       // results of a solve as one block (k_pack_results): one device-to-host copy per download
    MsBuf sv{}; double *sv_alloc = nullptr; size_t sv_bytes = 0; bool sv_prepared = false;      // single-vector solve phase (tsba_bandsv.h); sv_prepared: k_sv_linv has run on the current factorisation
    MsBuf ms{}; double *ms_alloc = nullptr; size_t ms_bytes = 0; int ms_cap = 0;      // multi-right-hand-side solve phase of the partitioned band solver (tsba_bandms.h)
    std::vector<int32_t> rb_r, rb_c; std::vector<double> rb_v;      // tsba_debug_reduced_blocks: the blocks between its two calls
    int cov_text = -1; double *cov_log = nullptr;     // tsba_theta_optim: V of this plane at the end of every pass [TSBA_MAX_LEVELS][6]
    int far_B = 0, n_far = 0, pcg_parts = 0; unsigned int pcg_seq = 0;      // band + long-range blocks (tsba_pcg.h): band of M in pose blocks, blocks outside it, partial sums per vector kernel
    int rank = 0, world = 1; bool force_multi = false;
    tsba_debug_options dbg{};                      // test / diagnostics switches (tsba_debug_set), all zero in production
    struct LocalGroup *lgroup = nullptr;           // in-process communicator (tsba_comm_init_local)
    bool in_solve = false, has_token = false;      // local group: this rank is inside tsba_solve / holds the group's device token
    size_t x_acc = 0, x_trial = 0, x_lin = 0, x_pass = 0;   // bytes handed to collectives: running total; last LM trial / linearisation / pass set-up
    decltype(&ncclCommCount) p_count = nullptr;
    void *rccl_so = nullptr; ncclComm_t comm = nullptr;
    decltype(&ncclGetUniqueId) p_getid = nullptr; decltype(&ncclCommInitRank) p_init = nullptr;
    decltype(&ncclAllReduce) p_allreduce = nullptr; decltype(&ncclCommDestroy) p_destroy = nullptr;
    decltype(&ncclGetErrorString) p_errstr = nullptr;
};

static void set_err(Ctx *c, const std::string &s) { c->err = s; }
// hipFuncAttributeMaxDynamicSharedMemorySize of a kernel, once per (kernel, device) and size class instead of before every solve (a global-BA solve made ~45 of these
// calls).  And with a retry: beside another host thread that was initialising an RCCL communicator the call has been seen to fail with "invalid device function" for a
// kernel it had accepted a thousand times (the runtime's function table while another library's code objects are being registered; the busy-context suite of round 6's
// final tree, profiles/r06_gpu_suite_beside_busy_contexts_final.txt) -- a transient of the runtime must not fail a solve.
static int set_attr_cached(Ctx *c, const void *fn, int bytes, const char *what) {
    static std::mutex mu; static std::map<std::pair<const void *, int>, int> granted;
    std::lock_guard<std::mutex> lk(mu);
    const auto key = std::make_pair(fn, c->device); const auto it = granted.find(key);
    if (it != granted.end() && it->second >= bytes) return 0;
    hipError_t e = hipSuccess;
    for (int t = 0; t < 6; t++) {
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) break;
        (void)hipGetLastError(); std::this_thread::sleep_for(std::chrono::microseconds(100 << t));
    }
    if (e != hipSuccess) { set_err(c, std::string("hipFuncSetAttribute(") + what + "): " + hipGetErrorString(e)); return TSBA_ERR_DEVICE; }
    granted[key] = bytes; return 0;
}
#define ATTR(fn, bytes) do { const int rc_ = set_attr_cached(c, (const void *)(fn), (int)(bytes), #fn); if (rc_) return rc_; } while (0)
static bool is_multi(const Ctx *c);
// Kernels whose workgroups wait for each other inside ONE launch (value polling: k_solve_back, k_sv_cre_tree, k_sv_tree_back, k_cre_back_tree) make progress
// only if every workgroup of the launch has a compute unit.  They are chosen only where the whole grid fits the device at once (occupancy of that kernel
// with its dynamic LDS x compute units); otherwise the launch-per-step path that they replaced runs.  Other work on the device (a second context's
// kernels) can only DELAY the dispatch of a workgroup -- it ends without waiting for anything here -- and a wait that outlasts the polling bound is counted
// (ts_poll_giveups -> tsba_report.poll_timeouts) on top of failing the linear solve.
static bool grid_resident(Ctx *c, const void *fn, int threads, size_t lds, int grid) {
    int nb = -1;
    for (auto &e : c->occ_cache) if (e.first.first == fn && e.first.second == lds) nb = e.second;
    if (nb < 0) { if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, threads, lds) != hipSuccess) nb = 0;
        c->occ_cache.push_back({{fn, lds}, nb}); }
    return (long long)nb*(c->dbg.assume_cus > 0 ? c->dbg.assume_cus : c->n_cu) >= grid;
}

// Device memory comes from a few large slabs (bump allocation, 256-byte aligned) that persist across uploads: the ~200 arrays
// of a problem cost no hipMalloc / hipFree / hipMemset each (one memset per slab and upload), and host data is staged through a
// pinned mirror of the slab so that consecutive uploads leave as ONE host-to-device copy (a plan is ~50 arrays per level).
struct Slab { char *dev = nullptr, *host = nullptr; size_t size = 0, used = 0; };
static void flush_run(Ctx *c) {
    if (c->run_len) { Slab &sl = c->slabs[c->run_slab];
        hipMemcpyAsync(sl.dev + c->run_off, sl.host + c->run_off, c->run_len, hipMemcpyHostToDevice, c->stage_async && c->copy_stream ? c->copy_stream : c->stream); c->run_len = 0; }
}
static int slab_take(Ctx *c, size_t bytes, int *slab, size_t *off) {
    for (;;) {
        if (c->cur_slab < (int)c->slabs.size()) { Slab &sl = c->slabs[c->cur_slab];
            if (sl.size - sl.used >= bytes) { *slab = c->cur_slab; *off = sl.used; sl.used += bytes; return 0; }
            c->cur_slab++; continue; }
        Slab sl; sl.size = std::max<size_t>(bytes, (size_t)64 << 20);
        hipError_t e = hipMalloc((void **)&sl.dev, sl.size);
        if (e != hipSuccess) { set_err(c, std::string("hipMalloc: ") + hipGetErrorString(e)); return TSBA_ERR_DEVICE; }
        hipMemsetAsync(sl.dev, 0, sl.size, c->stream);
        if (c->stage_async) hipStreamSynchronize(c->stream);     // (a slab born during a solve: the copy stream must not overtake its clearing)
        c->slabs.push_back(sl);
    }
}
// A piece that receives no host data (an allocation, an empty list) right behind the pending run of staged bytes: when it is small the run goes on across it --
// zeros in the pinned mirror, which is what the slab's clearing left on the device -- instead of ending there.  Round 6 (rocprofv3 timeline of a one-shot
// tsba_pose_optim call): a single-frame plan has some thirty empty lists, each of them cut the run, and the call left as ~30 copies of a few hundred bytes
// (3 - 4 us of host time and 5 - 10 us of device time EACH, one after the other in front of the first kernel) instead of three.
#define RUN_BRIDGE ((size_t)32 << 10)
static void run_bridge(Ctx *c, int si, size_t off, size_t bytes) {
    Slab &sl = c->slabs[si];
    if (c->run_len && c->run_slab == si && c->run_off + c->run_len == off && bytes <= RUN_BRIDGE && sl.host) { memset(sl.host + off, 0, bytes); c->run_len += bytes; }
}
template <typename T>
static int dev_alloc(Ctx *c, T **out, size_t n) {
    const size_t bytes = (std::max<size_t>(n, 1)*sizeof(T) + 255) & ~(size_t)255;
    int si; size_t off; int rc = slab_take(c, bytes, &si, &off); if (rc) return rc;
    *out = (T *)(c->slabs[si].dev + off); run_bridge(c, si, off, bytes); return 0;
}
template <typename T>
static int dev_upload(Ctx *c, const T **out, const T *src, size_t n) {
    const size_t bytes = (std::max<size_t>(n, 1)*sizeof(T) + 255) & ~(size_t)255;
    int si; size_t off; int rc = slab_take(c, bytes, &si, &off); if (rc) return rc;
    Slab &sl = c->slabs[si];
    *out = (const T *)(sl.dev + off);
    if (!n || !src) { run_bridge(c, si, off, bytes); return 0; }
    if (!sl.host) { hipError_t e = hipHostMalloc((void **)&sl.host, sl.size, hipHostMallocDefault);
        if (e != hipSuccess) { set_err(c, std::string("hipHostMalloc: ") + hipGetErrorString(e)); return TSBA_ERR_DEVICE; } }
    if (c->run_len && (c->run_slab != si || c->run_off + c->run_len != off)) flush_run(c);
    if (!c->run_len) { c->run_slab = si; c->run_off = off; }
    memcpy(sl.host + off, src, n*sizeof(T));
    if (bytes > n*sizeof(T)) memset(sl.host + off + n*sizeof(T), 0, bytes - n*sizeof(T));
    c->run_len = off + bytes - c->run_off;
    return 0;
}
template <typename T>
static int dev_upload_vec(Ctx *c, const T **out, const std::vector<T> &v) { return dev_upload(c, out, v.data(), v.size()); }

static void join_planners(Ctx *c) { for (auto &t : c->planners) if (t.joinable()) t.join(); c->planners.clear(); }
static void free_problem(Ctx *c) {
    join_planners(c); c->stage_p = nullptr;
    hipStreamSynchronize(c->stream);
    if (c->copy_stream) hipStreamSynchronize(c->copy_stream);       // (a level staged ahead of a pass that a failed solve never reached: its copy reads the slabs' pinned mirror)
    for (int l = 0; l < TSBA_MAX_LEVELS; l++) c->lev_wait[l] = 0;
    // the slabs stay (tsba_destroy frees them): zero what the last problem used, restart the bump allocation
    for (Slab &sl : c->slabs) { if (sl.used) hipMemsetAsync(sl.dev, 0, sl.used, c->stream); sl.used = 0; }
    c->cur_slab = 0; c->run_len = 0;
    c->uploaded = false; c->lev.clear(); c->lev_built.clear();      // (the host plans keep their storage for the next upload: HostPlan::recycle)
    for (int l = 0; l < TSBA_MAX_LEVELS; l++) c->img_dev[l].clear();
}

extern "C" {

void tsba_default_options_local(tsba_options *o) {
    memset(o, 0, sizeof(*o));
    o->w_sx = o->w_sy = 1.0/1.2; o->w_t = 1.0/0.2;                  // optimizer.cc:1350-1351
    o->huber_scene = sqrt(5.991); o->huber_text = 3.0;              // :1369, :1454
    o->n_passes = 3;
    for (int i = 0; i < 3; i++) { o->levels[i] = 2 - i; o->its[i] = 10; o->chi2_mono[i] = 12.25; o->chi2_text[i] = i == 2 ? 0.95 : 0.5; }  // :282-289
    o->text_bad_ratio = 0.99; o->state = TSBA_STATE_LOCAL; o->outlier_scene = o->outlier_text = 1;
    o->use_text = 1; o->filter_good = 1; o->text_jacobian = 0;
    o->initial_radius = 1e4; o->max_radius = 1e16; o->min_radius = 1e-32; o->min_relative_decrease = 1e-3;
    o->function_tolerance = 1e-6; o->gradient_tolerance = 1e-10; o->parameter_tolerance = 1e-8;
    o->min_diagonal = 1e-6; o->max_diagonal = 1e32; o->lm_shard = 0; o->lm_nshard = 1;
}
void tsba_default_options_pose(tsba_options *o) { tsba_default_options_local(o); o->state = TSBA_STATE_NOTREACHWIN; }
void tsba_default_options_global(tsba_options *o) {
    tsba_default_options_local(o);
    o->w_sx = o->w_sy = o->w_t = 1.0; o->n_passes = 1; o->levels[0] = 0; o->its[0] = 20; o->chi2_mono[0] = 18.0;   // :411-414
    o->state = TSBA_STATE_GLOBAL; o->outlier_scene = o->outlier_text = 0; o->use_text = 0; o->filter_good = 0;
}

void tsba_default_options_init(tsba_options *o) {                    // optimizer.cc:960-1056
    tsba_default_options_local(o);
    o->w_sx = o->w_sy = o->w_t = 1.0; o->huber_scene = 3.0; o->huber_text = 3.0;
    o->n_passes = 4; for (int i = 0; i < 4; i++) { o->levels[i] = 3 - i; o->its[i] = 10; }
    o->state = TSBA_STATE_NOTREACHWIN; o->outlier_scene = o->outlier_text = 0; o->filter_good = 0;
}
void tsba_default_options_landmarker(tsba_options *o) {              // optimizer.cc:531-541,1861,1873,1922
    tsba_default_options_local(o);
    o->w_sx = o->w_sy = o->w_t = 1.0; o->huber_scene = sqrt(5.991); o->huber_text = 2.0;
    o->n_passes = 4; for (int i = 0; i < 4; i++) { o->levels[i] = 3 - i; o->its[i] = 50; o->chi2_mono[i] = 18.0; o->chi2_text[i] = 1.5; }
    o->state = TSBA_STATE_NOTREACHWIN; o->outlier_scene = 1; o->outlier_text = 0;
}
void tsba_default_options_theta(tsba_options *o) {                   // optimizer.cc:610-615,2176,2203-2209
    tsba_default_options_local(o);
    o->w_sx = o->w_sy = o->w_t = 1.0; o->huber_text = 1e300;          // LossFunction* = nullptr
    for (int i = 0; i < 3; i++) o->its[i] = 50;
    o->state = TSBA_STATE_NOTREACHWIN; o->outlier_scene = o->outlier_text = 0; o->filter_good = 0;
}

int tsba_create(void **ctx, int device) {
    if (!ctx) return TSBA_ERR_ARG;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return TSBA_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return TSBA_ERR_DEVICE;
    Ctx *c = new Ctx(); c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) { delete c; return TSBA_ERR_DEVICE; }
    for (int l = 0; l < TSBA_MAX_LEVELS; l++) c->plan_done[l].store(0);
    if (hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking) != hipSuccess) c->copy_stream = nullptr;     // (optional: staging then shares the compute stream)
    for (int l = 0; l < TSBA_MAX_LEVELS; l++) if (hipEventCreateWithFlags(&c->ev_stage[l], hipEventDisableTiming) != hipSuccess) c->ev_stage[l] = nullptr;
    if (hipEventCreateWithFlags(&c->ev_upload, hipEventDisableTiming) != hipSuccess) c->ev_upload = nullptr;
    hipDeviceProp_t prop;
    const bool ok = hipEventCreate(&c->ev0) == hipSuccess && hipEventCreate(&c->ev1) == hipSuccess
        && hipHostMalloc((void **)&c->st_host, sizeof(LmState)*TSBA_MAX_LEVELS + 64, 0) == hipSuccess
        && hipMalloc((void **)&c->st_log, sizeof(LmState)*TSBA_MAX_LEVELS) == hipSuccess
        && hipGetDeviceProperties(&prop, device) == hipSuccess;
    if (!ok) {                                      // nothing half-built leaves this function
        if (c->ev0) hipEventDestroy(c->ev0); if (c->ev1) hipEventDestroy(c->ev1);
        if (c->st_host) hipHostFree(c->st_host); if (c->st_log) hipFree(c->st_log);
        hipStreamDestroy(c->stream); delete c; return TSBA_ERR_DEVICE;
    }
    if (hipHostMalloc((void **)&c->hprog, 64, hipHostMallocDefault) != hipSuccess) c->hprog = nullptr; else memset(c->hprog, 0, 64);   // (optional: early-exit polling only)
    c->n_cu = prop.multiProcessorCount;
    c->lds_limit = prop.sharedMemPerBlock;       // 64 KiB default static limit; dynamic up to 160 KiB on gfx950
    if (c->lds_limit < 160*1024) c->lds_limit = 160*1024;
    *ctx = c; return TSBA_OK;
}
static void lgroup_forget(Ctx *c);              // (defined with LocalGroup below)
int tsba_destroy(void *ctx) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    hipSetDevice(c->device);
    lgroup_forget(c);
    free_problem(c);
    hipStreamSynchronize(c->stream);
    for (Slab &sl : c->slabs) { hipFree(sl.dev); if (sl.host) hipHostFree(sl.host); }
    c->slabs.clear();
    if (c->comm && c->p_destroy) c->p_destroy(c->comm);
    hipHostFree(c->st_host); hipFree(c->st_log); if (c->hprog) hipHostFree(c->hprog);
    if (c->lbl_dev) hipFree(c->lbl_dev); if (c->lbl_host) hipHostFree(c->lbl_host);
    if (c->ic.dev) hipFree(c->ic.dev); if (c->ic.stage) hipHostFree(c->ic.stage);
    if (c->ms_alloc) hipFree(c->ms_alloc);
    if (c->sv_alloc) hipFree(c->sv_alloc);
    if (c->dl_dev) hipFree(c->dl_dev);
    if (c->dl_host) hipHostFree(c->dl_host);
    if (c->ecg_alloc) hipFree(c->ecg_alloc);
    if (c->wb_alloc) hipFree(c->wb_alloc);
    for (int l = 0; l < TSBA_MAX_LEVELS; l++) if (c->ev_stage[l]) hipEventDestroy(c->ev_stage[l]);
    if (c->ev_upload) hipEventDestroy(c->ev_upload);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    hipEventDestroy(c->ev0); hipEventDestroy(c->ev1); hipStreamDestroy(c->stream);
    delete c; return TSBA_OK;
}
int tsba_abi_version(void) { return TSBA_ABI_VERSION; }
const char *tsba_last_error(void *ctx) { return ctx ? ((Ctx *)ctx)->err.c_str() : "null ctx"; }

// Every index the plan builder and the kernels dereference is range-checked here, once, in O(problem size): a bad index from the
// adapter becomes TSBA_ERR_ARG instead of a host out-of-bounds read or a GPU memory fault (the library never aborts the process).
static int check_problem(Ctx *c, const tsba_problem *p, const tsba_options *o) {
    auto bad = [&](const char *what) { set_err(c, std::string("invalid problem: ") + what); return TSBA_ERR_ARG; };
    if (!p || !o) return bad("null problem / options");
    if (p->n_kf <= 0 || p->n_levels < 1 || p->n_levels > TSBA_MAX_LEVELS) return bad("n_kf / n_levels");
    if (p->n_pt < 0 || p->n_text < 0 || p->n_tobs < 0 || p->n_sgood < 0) return bad("negative count");
    if (o->n_passes < 1 || o->n_passes > TSBA_MAX_LEVELS) return bad("n_passes");
    for (int i = 0; i < o->n_passes; i++) if (o->levels[i] < 0 || o->levels[i] >= p->n_levels) return bad("pass level out of range");
    if (o->text_jacobian != 0) { set_err(c, "text_jacobian=1 (numeric diff) is oracle-only"); return TSBA_ERR_ARG; }
    if (o->lm_nshard < 1 || o->lm_shard < 0 || o->lm_shard >= o->lm_nshard) return bad("lm_shard / lm_nshard");
    if (!p->pose) return bad("pose is null");
    if (p->n_pt > 0 && (!p->rho || !p->pt_ray || !p->pt_host)) return bad("null point array");
    if (p->n_text > 0 && (!p->theta || !p->text_host || !p->text_box_ray)) return bad("null text-plane array");
    if (p->n_sgood > 0 && !p->sgood) return bad("sgood is null");
    bool frozen = false;
    for (int j = 0; j < p->n_pt; j++) { const int h = p->pt_host[j]; if (h < -1 || h >= p->n_kf) return bad("pt_host out of range"); frozen |= h < 0; }
    if (frozen && !p->pt_host_Trw) return bad("pt_host_Trw is null but a point has a frozen host");
    frozen = false;
    for (int j = 0; j < p->n_text; j++) { const int h = p->text_host[j]; if (h < -1 || h >= p->n_kf) return bad("text_host out of range"); frozen |= h < 0; }
    if (frozen && !p->text_host_Twr) return bad("text_host_Twr is null but a plane has a frozen host");
    if (p->n_tobs > 0) {
        if (!p->tobs_kf || !p->tobs_text || !p->tobs_good || !p->tobs_fgood_off) return bad("null text-observation array");
        if (p->tobs_fgood_off[0] < 0) return bad("tobs_fgood_off[0] < 0");
        for (int t = 0; t < p->n_tobs; t++) {
            if (p->tobs_kf[t] < 0 || p->tobs_kf[t] >= p->n_kf) return bad("tobs_kf out of range");
            if (p->tobs_text[t] < 0 || p->tobs_text[t] >= p->n_text) return bad("tobs_text out of range");
            if (p->tobs_fgood_off[t+1] < p->tobs_fgood_off[t]) return bad("tobs_fgood_off not monotone");
        }
        if (p->tobs_fgood_off[p->n_tobs] > 0 && !p->tfgood) return bad("tfgood is null");
    }
    std::vector<char> seen(p->n_levels, 0);
    std::vector<int32_t> maxraw;
    for (int i = 0; i < o->n_passes; i++) {
        const int l = o->levels[i]; if (seen[l]) continue; seen[l] = 1;
        const int ns = p->n_sobs[l];
        if (ns < 0) return bad("n_sobs < 0");
        if (ns > 0) {
            if (!p->sobs_kf[l] || !p->sobs_pt[l] || !p->sobs_flag[l] || !p->sobs_uv0[l]) return bad("null scene-observation array");
            const int32_t *kf = p->sobs_kf[l], *pt = p->sobs_pt[l], *fl = p->sobs_flag[l];
            for (int s = 0; s < ns; s++) {
                if ((unsigned)kf[s] >= (unsigned)p->n_kf) return bad("sobs_kf out of range");
                if ((unsigned)pt[s] >= (unsigned)p->n_pt) return bad("sobs_pt out of range");
                if ((unsigned)fl[s] >= (unsigned)p->n_sgood) return bad("sobs_flag out of range");
            }
        }
        if (p->n_text > 0 && p->tfeat_off[l]) {             // (a level without text features passes tfeat_off = NULL)
            const int32_t *off = p->tfeat_off[l]; const int nf = p->n_tfeat[l];
            if (nf < 0 || off[0] < 0 || off[p->n_text] > nf) return bad("tfeat_off out of range");
            for (int j = 0; j < p->n_text; j++) if (off[j+1] < off[j]) return bad("tfeat_off not monotone");
            if (off[p->n_text] > off[0] && (!p->tfeat_raw[l] || !p->tfeat_uv[l] || !p->tfeat_ref[l])) return bad("null text-feature array");
            maxraw.assign(p->n_text, -1);                  // a feature's flag is tfgood[tobs_fgood_off[t] + raw]: raw must fit every observation's span
            for (int j = 0; j < p->n_text; j++) for (int f = off[j]; f < off[j+1]; f++) {
                const int r = p->tfeat_raw[l][f]; if (r < 0) return bad("tfeat_raw < 0"); maxraw[j] = std::max(maxraw[j], (int32_t)r); }
            for (int t = 0; t < p->n_tobs; t++)
                if (maxraw[p->tobs_text[t]] >= p->tobs_fgood_off[t+1] - p->tobs_fgood_off[t]) return bad("tfeat_raw exceeds the observation's flag span");
        }
        if (o->use_text && p->n_tobs > 0)
            if (!p->img[l] || p->img_w[l] <= 0 || p->img_h[l] <= 0 || p->img_w[l]*p->img_h[l] > MS_MASK_WORDS*32) { set_err(c, "missing image level or image larger than 640x480"); return TSBA_ERR_ARG; }
    }
    return 0;
}

static int stage_level(Ctx *c, const tsba_problem *p, int l, double *t_plan, double *t_img);
// lazy: the caller (a one-shot entry point) runs the solve right away and keeps *p alive until it returns -- on small windows only the first
// pass's level is staged here, the others when their pass begins (tsba_solve), so that the coarse passes run on the device while the host
// still builds and stages the plan of level 0 (the largest: ~1 ms of a 20-keyframe window's cold call)
static int upload_impl(void *ctx, const tsba_problem *p, const tsba_options *o, bool lazy) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    hipSetDevice(c->device);
    int rc = check_problem(c, p, o); if (rc) return rc;
    const bool tdbg = c->dbg.verbose != 0;
    auto tu0 = std::chrono::steady_clock::now(); double t_plan = 0.0, t_img = 0.0;
    free_problem(c);
    auto tu1 = std::chrono::steady_clock::now();
    c->opt = *o;
    if (c->world > 1) { c->opt.lm_shard = c->rank; c->opt.lm_nshard = c->world; }
    o = &c->opt;
    c->n_kf = p->n_kf; c->n_pt = p->n_pt; c->n_text = p->n_text; c->n_tobs = p->n_tobs; c->n_sgood = p->n_sgood; c->n_levels = p->n_levels;
    c->n_tfgood = p->n_tobs > 0 ? p->tobs_fgood_off[p->n_tobs] : 0;
    Work &W = c->W; memset(&W, 0, sizeof(W));
    W.n_kf = p->n_kf; W.n_pt = p->n_pt; W.n_text = p->n_text; W.n_tobs = p->n_tobs; W.N = 6*p->n_kf;
    for (int k = 0; k < 4; k++) W.K0[k] = p->K[k];
    W.w_sx = o->w_sx; W.w_sy = o->w_sy; W.w_t = o->w_t; W.huber_s = o->huber_scene; W.huber_t = o->huber_text;
    W.filter_good = o->filter_good; W.min_diag = o->min_diagonal; W.max_diag = o->max_diagonal;
    W.rank = c->rank; W.world = c->world;
    // the per-level plans are independent of each other and of the uploads below: one host thread per level builds them while this
    // thread stages the parameter / observation arrays (C4: 1.6 ms of plan construction in sequence -> the largest level, overlapped)
    c->hplan.resize(p->n_levels); c->lev.resize(p->n_levels); c->lev_built.assign(p->n_levels, 0); c->lev_planned.assign(p->n_levels, 0);
    c->planners.clear(); c->planners.resize(p->n_levels);
    std::vector<std::thread> &planners = c->planners;
    struct Joiner { Ctx *c; bool armed; ~Joiner() { if (armed) join_planners(c); } } joiner{c, true};   // on the error returns
    int n_lev_used = 0; { std::vector<char> sn(p->n_levels, 0); for (int q = 0; q < o->n_passes; q++) if (!sn[o->levels[q]]) { sn[o->levels[q]] = 1; n_lev_used++; } }
    const bool small_window = solve_lds_doubles(W.N)*sizeof(double) <= 160*1024 - 64;
    // (round 6: also the single-frame problems of tsba_pose_optim -- the plans of the later passes' levels are built and staged while the first pass runs)
    // (round 6: a single-frame problem with every landmark frozen -- tsba_pose_optim, per frame -- has a plan that is a copy of its input: written down directly
    // on this thread for every level (build_plan_single_frame, microseconds), no plan threads, every level staged before the solve)
    const bool single_frame = plan_is_single_frame(p, o) && c->dbg.host_pair_lists != 1;
    const bool defer = lazy && small_window && n_lev_used > 1 && !is_multi(c) && !single_frame;
    for (int l = 0; l < TSBA_MAX_LEVELS; l++) { c->plan_done[l].store(0); c->lev_wait[l] = 0; }
    std::function<void()> first_plan;
    {   auto tp0 = std::chrono::steady_clock::now();
        std::vector<char> seen(p->n_levels, 0);
        const int n_lev = n_lev_used;
        const bool reorder = !c->dbg.no_kf_reorder;
        // ring maps (one loop closure): a single-level, single-GPU solve through the partitioned solver with the cyclic-reduction separator tree
        const int ring_max = (n_lev == 1 && !c->dbg.no_ring && !c->dbg.no_band_stream && c->dbg.sep_solver != 1 && c->dbg.sep_solver != 3 && c->dbg.band_parts != 1) ? CR_SMAX/6 : 0;
        // maps with long-range coupling (several loop closures, points seen again much later): band + blocks outside it, preconditioned conjugate gradients
        const int far_max = (n_lev == 1 && c->dbg.far_solver != 1 && !c->dbg.no_band_stream && (!c->dbg.no_ring || c->dbg.far_solver >= 2)) ? CR_SMAX/6 : 0;     // (no_ring asks for the reordering path)
        const bool far_force = c->dbg.far_solver == 2 || c->dbg.far_solver == 3;
        // the point slot pairs by S block are built on the device (tsba_devplan.h) -- since round 4 on maps of more than 126 keyframes (2 M pairs: 9 of the plan's 21 ms), since round 6 on
        // windows as well: they are the largest part of a window's plan (C4 level 2: 0.18 of 0.5 ms on the calling thread of a one-shot call, level 0: 0.7 of 1.5 ms), the three small
        // launches that build them cost the pass ~15 us, the lists are the same entries in the same order (same bits: test_schur_lists_built_on_the_device_equal_the_host_lists)
        // (... on problems of at least 4096 scene observations: below, the three launches cost a level more than the host's lists -- InitBA's pair, a landmark's refinement, a plane's:
        // 15 us per level of calls that take 1.3 - 2.3 ms in launch-bound trials of ~32 us; tsba_debug_options.host_pair_lists = 2 asks for the device lists at any size)
        const bool dev_pairs = p->n_kf > 1 && c->dbg.host_pair_lists != 1 && (c->dbg.host_pair_lists == 2 || p->n_sobs[0] >= 4096);
        for (int ps = o->n_passes - 1; ps >= 0; ps--) { const int l = o->levels[ps]; if (seen[l]) continue;
            bool later = false; for (int q = 0; q < ps; q++) later |= o->levels[q] == l;       // (a level used by an earlier pass is started with that pass)
            if (later) continue;
            seen[l] = 1;
            HostPlan *H = &c->hplan[l];
            c->lev_planned[l] = 1;
            std::atomic<int> *done = &c->plan_done[l];
            // the levels of the later passes first (the largest plans); a deferring call builds the first pass's (small) plan on this thread:
            // it is needed at once, and a thread's start costs as much as that plan
            if (single_frame) { build_plan_single_frame(p, o, l, *H); done->store(1); continue; }
            if (defer && ps == 0) { first_plan = [=]() { build_plan(p, o, l, *H, tdbg, reorder, ring_max, far_max, far_force, dev_pairs); done->store(1); }; continue; }    // (built below, behind the problem arrays' copy)
            planners[l] = std::thread([p, o, l, H, tdbg, reorder, ring_max, far_max, far_force, dev_pairs, done]() { build_plan(p, o, l, *H, tdbg, reorder, ring_max, far_max, far_force, dev_pairs); done->store(1, std::memory_order_release); }); }
        t_plan += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count();
    }
#define UP(dst, src, n) do { rc = dev_upload(c, &(dst), (src), (size_t)(n)); if (rc) return rc; } while (0)
#define AL(dst, n) do { rc = dev_alloc(c, &(dst), (size_t)(n)); if (rc) return rc; } while (0)
    const double *cd; const uint8_t *cu;
    UP(cd, p->pose, 7*(size_t)p->n_kf); c->pose0 = (double *)cd;
    UP(cd, p->rho, p->n_pt); c->rho0 = (double *)cd;
    UP(cd, p->theta, 3*(size_t)p->n_text); c->theta0 = (double *)cd;
    UP(cu, p->sgood, p->n_sgood); c->sgood0 = (uint8_t *)cu;
    UP(cu, p->tobs_good, p->n_tobs); c->tobs_good0 = (uint8_t *)cu;
    UP(cu, p->tfgood, c->n_tfgood); c->tfgood0 = (uint8_t *)cu;
    std::vector<uint8_t> ki(p->n_kf, 0); if (p->kf_initial) memcpy(ki.data(), p->kf_initial, p->n_kf);
    UP(cu, ki.data(), p->n_kf); c->kf_initial = (uint8_t *)cu;
    for (int b = 0; b < 2; b++) { AL(W.pose[b], 7*(size_t)p->n_kf); AL(W.rho[b], p->n_pt); AL(W.theta[b], 3*(size_t)p->n_text); }
    AL(W.sgood, p->n_sgood); AL(W.tobs_good, p->n_tobs); AL(W.tfgood, c->n_tfgood);
    UP(W.pt_ray, p->pt_ray, 2*(size_t)p->n_pt); UP(W.pt_host, p->pt_host, p->n_pt);
    { std::vector<double> z; const double *src = p->pt_host_Trw; if (!src) { z.assign(12*(size_t)p->n_pt, 0.0); src = z.data(); } UP(W.pt_Trw, src, 12*(size_t)p->n_pt); }
    UP(W.text_host, p->text_host, p->n_text);
    { std::vector<double> z; const double *src = p->text_host_Twr; if (!src) { z.assign(12*(size_t)p->n_text, 0.0); src = z.data(); } UP(W.text_Twr, src, 12*(size_t)p->n_text); }
    UP(W.text_box, p->text_box_ray, 8*(size_t)p->n_text);
    UP(W.tobs_kf, p->tobs_kf, p->n_tobs); UP(W.tobs_text, p->tobs_text, p->n_tobs); UP(W.tobs_fgood_off, p->tobs_fgood_off, (size_t)p->n_tobs + 1);
    AL(c->musig2[0], 2*(size_t)p->n_tobs); AL(c->musig2[1], 2*(size_t)p->n_tobs); c->musig_sel = 0; W.musig = c->musig2[0];
    AL(c->ticket, 4); W.poll0 = (unsigned int *)(c->ticket + 1);
    AL(c->lin_ticket, 2); c->lin_base = 0;                         // (slab memory: zero)
    AL(W.kf_in, p->n_kf); AL(W.kf_const, p->n_kf); AL(W.act_pt, p->n_pt); AL(W.act_tx, p->n_text);
    AL(W.fidx, p->n_kf); AL(W.nfree, 2); AL(W.dbg, 64); AL(W.trace, 4*(size_t)TSBA_TRACE_CAP*TSBA_MAX_LEVELS); AL(W.LDbuf, 32*((size_t)p->n_kf + BAND_BW_MAX/6 + 1));     // (+ the ghost blocks of a ring map)
    // ---- plane cache (tsba_problem.kf_id): the keyframes of this call get their slots; the planes of those not seen before are staged and copied
    std::vector<int> &ic_slot = c->ic_slot; ic_slot.clear();
    bool &use_img_cache = c->use_img_cache; use_img_cache = false;
    if (p->kf_id && !o->img_on_device && o->use_text && p->n_tobs > 0 && p->n_kf <= TSBA_IMG_CACHE_KF) {
        Ctx::ImgCache &IC = c->ic;
        unsigned mask = 0; size_t off = 0, lo[TSBA_MAX_LEVELS] = {0,0,0,0}; bool same = IC.dev != nullptr;
        { std::vector<char> seen(p->n_levels, 0);
          for (int ps = 0; ps < o->n_passes; ps++) { const int l = o->levels[ps]; if (seen[l] || !p->img[l]) continue; seen[l] = 1; mask |= 1u << l; }
          for (int l = 0; l < p->n_levels; l++) if (mask >> l & 1) { lo[l] = off; off += ((size_t)p->img_w[l]*p->img_h[l] + 255) & ~(size_t)255;
              same = same && IC.w[l] == p->img_w[l] && IC.h[l] == p->img_h[l]; } }
        same = same && IC.lvl_mask == mask && IC.slot == off;
        if (mask && off) {
            if (!same) {                                     // new geometry (or first use): an empty cache of TSBA_IMG_CACHE_KF slots
                hipStreamSynchronize(c->stream);
                if (IC.dev) { hipFree(IC.dev); IC.dev = nullptr; }
                if (hipMalloc((void **)&IC.dev, off*TSBA_IMG_CACHE_KF) != hipSuccess) { IC.dev = nullptr; set_err(c, "hipMalloc (plane cache)"); return TSBA_ERR_DEVICE; }
                IC.slot = off; IC.lvl_mask = mask;
                for (int l = 0; l < TSBA_MAX_LEVELS; l++) { IC.lvl_off[l] = lo[l]; IC.w[l] = l < p->n_levels ? p->img_w[l] : 0; IC.h[l] = l < p->n_levels ? p->img_h[l] : 0; }
                for (int q = 0; q < TSBA_IMG_CACHE_KF; q++) { IC.id[q] = 0; IC.used[q] = 0; IC.full[q] = false; }
            }
            for (int l = 0; l < p->n_levels; l++) if (mask >> l & 1) for (int k = 0; k < p->n_kf; k++)      // before any slot changes hands: a failed call must not leave a slot marked full without its planes
                if (!p->img[l][k]) { set_err(c, "null image pointer"); return TSBA_ERR_ARG; }
            IC.tick++;
            ic_slot.assign((size_t)p->n_kf, -1);
            std::vector<int> miss;
            for (int k = 0; k < p->n_kf; k++) { for (int q = 0; q < TSBA_IMG_CACHE_KF; q++) if (IC.full[q] && IC.id[q] == p->kf_id[k]) { ic_slot[(size_t)k] = q; IC.used[q] = IC.tick; break; }
                if (ic_slot[(size_t)k] < 0) miss.push_back(k); }
            for (int k : miss) { int best = -1;              // least recently used slot that this call does not use
                for (int q = 0; q < TSBA_IMG_CACHE_KF; q++) if (IC.used[q] != IC.tick && (best < 0 || !IC.full[q] || (IC.full[best] && IC.used[q] < IC.used[best]))) { best = q; if (!IC.full[q]) break; }
                IC.id[best] = p->kf_id[k]; IC.full[best] = true; IC.used[best] = IC.tick; ic_slot[(size_t)k] = best; }
            IC.hits += p->n_kf - (long long)miss.size(); IC.misses += (long long)miss.size();
            if (!miss.empty()) {
                const size_t need = miss.size()*off;
                if (IC.stage_cap < need) { if (IC.stage) hipHostFree(IC.stage); IC.stage = nullptr; IC.stage_cap = 0;
                    if (hipHostMalloc((void **)&IC.stage, need, hipHostMallocDefault) != hipSuccess) { IC.stage = nullptr;
                        for (int k : miss) IC.full[ic_slot[(size_t)k]] = false;          // (their planes were never copied)
                        set_err(c, "hipHostMalloc (plane cache staging)"); return TSBA_ERR_DEVICE; }
                    IC.stage_cap = need; }
                for (size_t m = 0; m < miss.size(); m++) { const int k = miss[m];
                    for (int l = 0; l < p->n_levels; l++) if (mask >> l & 1) memcpy(IC.stage + m*off + lo[l], p->img[l][k], (size_t)p->img_w[l]*p->img_h[l]);
                    hipMemcpyAsync(IC.dev + (size_t)ic_slot[(size_t)k]*off, IC.stage + m*off, off, hipMemcpyHostToDevice, c->stream); }
            }
            use_img_cache = true;
        }
    }
    // ---- per-level plans
    if (first_plan) {                              // a deferring call: the first pass's plan on this thread, while the problem arrays (and a new keyframe's planes) cross the bus
        flush_run(c);
        auto tp0 = std::chrono::steady_clock::now(); first_plan();
        t_plan += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count(); }
    size_t mx_pair = 1, mx_tg = 1, mx_pslot = 1, mx_tslot = 1, mx_cnt = 1;
    for (int ps = 0; ps < o->n_passes; ps++) {
        const int l = o->levels[ps]; if (c->lev_built[l]) continue;
        if (defer && ps > 0) {                     // staged when its pass begins: buffers by what the level can hold at most
            const size_t nsc = (size_t)p->n_sobs[l], ntg = (size_t)p->n_tobs;
            mx_pair = std::max(mx_pair, std::min((size_t)p->n_kf*((size_t)p->n_kf + 1), nsc + ntg)); mx_tg = std::max(mx_tg, ntg);
            mx_pslot = std::max(mx_pslot, nsc + (size_t)p->n_pt); mx_tslot = std::max(mx_tslot, ntg + (size_t)p->n_text); mx_cnt = std::max(mx_cnt, nsc + ntg);
            continue; }
        rc = stage_level(c, p, l, &t_plan, &t_img); if (rc) return rc;
        const LevelDev &D = c->lev[l];
        mx_pair = std::max(mx_pair, (size_t)D.n_pair); mx_tg = std::max(mx_tg, (size_t)D.n_tg);
        mx_pslot = std::max(mx_pslot, (size_t)D.n_pslot); mx_tslot = std::max(mx_tslot, (size_t)D.n_tslot); mx_cnt = std::max(mx_cnt, (size_t)D.n_sc + D.n_tg);
    }
    c->stage_p = defer ? p : nullptr;
    c->nb_back_max = back_blocks_pt(p->n_pt) + back_blocks_tx(p->n_text) + (p->n_kf + 255)/256;      // k_back's blocks (at least k_mid's landmark blocks)
    {   bool po = p->n_kf == 1;
        for (int j = 0; po && j < p->n_pt; j++) po = p->pt_host[j] < 0;
        for (int j = 0; po && j < p->n_text; j++) po = p->text_host[j] < 0;
        c->pose_only = po && !c->dbg.no_pose_kernel;
        W.pst = nullptr; W.ppart = nullptr;
        if (c->pose_only) {
            size_t gmax = 1;
            for (int l = 0; l < p->n_levels; l++) {
                if (c->lev_built[l]) gmax = std::max(gmax, (size_t)pose_grid(c->lev[l]));
                else if (c->lev_planned[l]) {                // a level staged later (deferred): its grid is bounded by what the level can hold
                    size_t npf = 0;
                    if (p->tfeat_off[l]) for (int q = 0; q < p->n_tobs; q++) { const int j = p->tobs_text[q]; npf += (size_t)(p->tfeat_off[l][j + 1] - p->tfeat_off[l][j]); }
                    gmax = std::max(gmax, ((size_t)p->n_sobs[l] + 255)/256 + (npf + 31)/32 + 1); } }
            AL(W.pst, 2); AL(W.ppart, 3*28*gmax);
        } }
    for (int b = 0; b < 2; b++) {
        LinBuf &B = W.lb[b];
        AL(B.pairM, 27*mx_pair); AL(B.pairCost, mx_pair); AL(B.pairR, 9*mx_pair); AL(B.pairOut, 90*mx_pair);
        AL(B.tgM, 27*mx_tg); AL(B.tgCost, mx_tg);
        AL(B.w_pt, PT_REC*mx_pslot); AL(B.vdb_pt, PT_VDB*(size_t)p->n_pt);
        AL(B.w_tx, TX_REC*mx_tslot); AL(B.V_tx, 6*(size_t)p->n_text); AL(B.b_tx, 3*(size_t)p->n_text); AL(B.dgs_tx, 3*(size_t)p->n_text);
        AL(B.Hd, W.N); AL(B.bp, W.N); AL(B.bp_loc, W.N); AL(B.dgs_p, W.N);
        AL(B.lmpart, 3*((size_t)p->n_pt/64 + (size_t)p->n_text/2 + mx_pair + 8));      // (one partial per k_mid block, at its smallest block size)      // (one partial per k_mid block: at most n_pt / 128 + n_text / 128 + pairs / 128 + 3, and nb_back_max >= n_pt / 64 + n_text / 16)
    }
    AL(W.sig_pt, p->n_pt); AL(W.sig_tx, 3*(size_t)p->n_text); AL(W.sig_p, W.N);
    AL(W.cb, 2*(size_t)W.N + 8); AL(W.cbm, 1);
    {   // reduced camera matrix: dense for the LDS solver; for the large-system Cholesky only its band (rows overlap in a skewed
        // view: S(i,j) = base[i*(LDB-1) + j], LDB = band + 96 columns of the diagonal block's upper triangle, where the inverse
        // diagonal factors are kept) -- 80 MB instead of 7.2 GB at 5000 keyframes, and what the ranks all-reduce
        const int use_lds_ = solve_lds_doubles(W.N)*sizeof(double) <= 160*1024 - 64;        // (as solve_lds_bytes)
        W.dp_poll = use_lds_ && !is_multi(c) && !c->pose_only && (c->dbg.solve_variant == 0 || c->dbg.solve_variant == 5);        // solver and back-substitution in one launch (k_solve_back)
        if (W.dp_poll) {                           // ... whose workgroups poll the solver's: only where the whole grid is resident at once (else k_solve_t + k_back + k_decide)
            const int nb_all_ = back_blocks_pt(p->n_pt) + back_blocks_tx(p->n_text) + (p->n_kf + 255)/256;
            const int ldsb_ = std::max((int)(solve_lds_doubles(W.N)*sizeof(double)), (int)((768 + W.N + 2)*sizeof(double)));
            ATTR(k_solve_back<true>, ldsb_);
            if (!grid_resident(c, (const void *)k_solve_back<true>, SOLVE_THREADS, (size_t)ldsb_, 1 + (nb_all_ + 2)/3)) W.dp_poll = 0;
        }
        int bwmax = 0; for (int l = 0; l < p->n_levels; l++) if (c->lev_built[l]) bwmax = std::max(bwmax, c->lev[l].bw_rows);
        // band storage: row i holds the columns [i - Wb, i + up) (skewed view S(i, j) = base[i (LDB - 1) + j]).  The blocked Cholesky of
        // tsba_chol.h writes 96-wide blocks on both sides of the diagonal (up = CH_NB, Wb = band + CH_NB - 1); the streaming / partitioned
        // solvers read the band only (up = 6: the diagonal pose block is stored square) -- 72 instead of 251 columns per row at a band of
        // 60, and the band is cleared before every Schur assembly (60 MB per LM trial at 5000 keyframes with the wide rows)
        const bool stream_ok = bwmax >= 6 && bwmax <= BAND_BW_MAX && band_chunk_blocks(bwmax) > 0 && !c->dbg.no_band_stream;
        int ring = 0, ring_k0 = 0; for (int l = 0; l < p->n_levels; l++) if (c->lev_built[l] && c->hplan[l].ring) { ring = 1; ring_k0 = c->hplan[l].ring_k0; }     // (a ring plan is only built for single-level solves)
        int ring_G = 0;
        const size_t nrow = (size_t)W.N + (ring ? bwmax : 0);                                                 // + the ghost rows of the first separator
        c->S_up = stream_ok ? 6 : CH_NB;
        const size_t LDB = stream_ok ? (size_t)bwmax + 12 : (size_t)bwmax + 2*CH_NB - 1;
        if (use_lds_ || (size_t)bwmax + 2*CH_NB - 1 >= (size_t)W.N) { c->S_count = (size_t)(W.N + 1)*W.N; AL(c->S_alloc, c->S_count); W.S = c->S_alloc; W.ldS = W.N; W.band = 0; }
        else { c->S_count = nrow*LDB + LDB; AL(c->S_alloc, c->S_count); W.S = c->S_alloc + (LDB - c->S_up); W.ldS = (int)LDB - 1; W.band = 1; }
        W.ring = 0; W.ring_g = 0; W.ring_b = 0; W.ring_k0 = -1;
        c->Lcol = nullptr; c->band_stream = 0; c->sep_cr = false;
        if (W.band && stream_ok) {
            AL(c->Lcol, ((size_t)p->n_kf + bwmax/6 + 1)*bwmax*6); c->band_stream = 1;
            // substructuring: P interiors on P workgroups + a separator system (again a band, 2 bw - 6 wide)
            // number of interiors: the interiors run in parallel (n_kf / P blocks each, ~3.5 us per block, 5 us once the border makes the
            // panel waves take two rounds), the separator system is sequential again ((P - 1) B blocks at ~4.5 us, 5.5 us when its band
            // exceeds 115 rows): the sum is smallest near sqrt(n_kf t_f / (B t_s))
            const int Bq = bwmax/6;
            const double t_f = bwmax > 57 ? 5.0 : 3.5, t_s = 2*bwmax - 6 > 115 ? 5.5 : 4.5;
            int P = (int)lround(sqrt((double)p->n_kf*t_f/((double)std::max(Bq, 1)*t_s)));
            bool want_cr = false;
            if (bwmax <= CR_SMAX && c->dbg.sep_solver != 1) {
                // separator system by cyclic reduction (tsba_bandcre.h): its cost grows with log2(P) only (~45 us per level: one elimination
                // and one back-substitution launch; 130 us with the three kernels of round 1), so many more, shorter interiors pay.
                // Measured at 5000 keyframes / band 10 (ms per 20-iteration solve): P = 64 / 80 / 96 / 112 / 127 / 150 -> 20.6 / 20.3 / 19.4 /
                // 18.6 / 18.0 / 18.8 (150: an eighth level)
                double best = 1e300; int bestP = P;
                for (int q = 4; q <= BANDP_MAXP; q++) {
                    if ((p->n_kf - (q - 1)*Bq)/q < 2*Bq + 2) break;          // (the kernels need 2 B + 2 blocks per interior; until round 6 this loop stopped at 2 B + 8 -- C5: 11 interiors, 9.39 ms; 13: 8.76 ms, tools/diag/gpu_sweep_parts.py)
                    int lev = 1; for (int hh = 1; hh < q - 1; hh <<= 1) lev++;
                    const double cost = (double)p->n_kf/q*t_f + 45.0*lev;
                    if (cost < best) { best = cost; bestP = q; }
                }
                // (few, long interiors -- some hundred keyframes -- are still cheaper with the sequential separator solve: compare)
                const int Ps = std::max(1, std::min(P, BANDP_MAXP));
                const double cost_seq = (double)p->n_kf/Ps*t_f + (double)(Ps - 1)*Bq*t_s;
                if (best < cost_seq) { P = bestP; want_cr = true; }
            }
            if (c->dbg.sep_solver >= 2 && bwmax <= CR_SMAX) want_cr = true;
            if (c->dbg.band_parts > 0) P = c->dbg.band_parts;
            P = std::max(1, std::min(P, BANDP_MAXP));
            if (ring) {                       // ring: a power of two interiors in the loop (the separator tree ends in its first separator and the ghost), cyclic reduction only
                const int cap = c->dbg.band_parts > 0 ? c->dbg.band_parts : 128, nloop = p->n_kf - ring_k0;
                int Pr = 4; while (2*Pr <= cap && (nloop - 2*Pr*Bq)/(2*Pr) >= 2*Bq + 8) Pr *= 2;
                ring_G = Pr;
                int Pt = 0;                   // a tail before the loop: interiors of about the loop's size
                if (ring_k0 > 0) { const int ql = (nloop - Pr*Bq)/Pr; Pt = std::max(1, std::min(std::min(RING_OFF - 1, BANDP_MAXP - Pr), (ring_k0 + ql/2)/(ql + Bq)));
                    while (Pt > 1 && (ring_k0 - (Pt - 1)*Bq)/Pt < 2*Bq + 8) Pt--; }
                P = Pr + Pt; want_cr = true;
            }
            while (!ring && P > 1 && (p->n_kf - (P - 1)*Bq)/P < ((c->dbg.band_parts > 0 || want_cr) ? 2*Bq + 2 : 4*Bq + 4)) P--;     // (2 B + 2: the least the kernels take; the sequential separator solve pays only for interiors of a few bands)
            if (P > 1 && bandp_chunk_blocks(bwmax) > 0 && 2*bwmax - 6 <= BAND_BW_MAX && band_chunk_blocks(2*bwmax - 6) > 0) {
                const int nsepb = cr_mmax(ring, P, ring_G);                     // separator labels (ring: the last one is the ghost of the loop's first separator)
                const int nsep = nsepb*bwmax, bws = 2*bwmax - 6;
                c->nsep_ld = nsep;
                AL(c->Lb, ((size_t)p->n_kf + bwmax/6 + 1)*bwmax*6); AL(c->Tbuf, (size_t)P*((size_t)4*bwmax*bwmax + 2*bwmax));
                AL(c->Bpart, (size_t)P*BANDP_NS*((size_t)bwmax*bwmax + bwmax));
                c->sep_cr = want_cr && P >= 4;
                if (c->sep_cr) { AL(c->Ssep, cr_pool_blocks(nsepb)*(size_t)bwmax*bwmax); AL(c->CRcontrib, (size_t)nsepb*cre_contrib_doubles(bwmax)); AL(c->CRfac, (size_t)nsepb*cre_rec_doubles(bwmax)); AL(c->CRgate, (size_t)(nsepb + 2)*TSBA_CRE_KMAX); c->cre_epoch = 0; }
                else AL(c->Ssep, (size_t)nsep*nsep + nsep);
                AL(c->Lcol_sep, (size_t)(nsep/6 + 1)*bws*6);
                Work &Ws = c->Wsep; memset(&Ws, 0, sizeof(Ws));
                Ws.N = nsep; Ws.n_kf = 0; Ws.S = c->Ssep; Ws.ldS = nsep; Ws.band = 1; Ws.st = nullptr;       // (st is set at launch: W.st is allocated below)
                AL(Ws.Sy, nsep); AL(Ws.g, nsep); AL(Ws.dp, nsep); AL(Ws.LDbuf, 32*(size_t)(nsep/6 + 1)); AL(Ws.nfree, 1); AL(Ws.fidx, 1);
                c->band_parts = P;
                W.ring = (ring && c->sep_cr) ? 1 : 0; W.ring_g = ring_G; W.ring_b = bwmax/6; W.ring_k0 = W.ring ? ring_k0 : -1;
            } else c->band_parts = 1;
        }
        if (ring && !W.ring) { set_err(c, "ring-shaped map: the partitioned band solver is not available for this plan"); return TSBA_ERR_STATE; }
        AL(W.Sy, W.N + BAND_BW_MAX);                            // (+ the ghost rows of a ring map)
        c->far_B = 0; c->n_far = 0; c->pcg_parts = 0;
        for (int l = 0; l < p->n_levels; l++) if (c->lev_built[l] && c->lev[l].far_B > 0) { c->far_B = c->lev[l].far_B; c->n_far = std::max(c->n_far, c->lev[l].n_far); }
        if (c->far_B > 0) {
            if (!W.band || !c->band_stream) { set_err(c, "map with long-range coupling: the band solvers are not available for this plan"); return TSBA_ERR_STATE; }
            c->pcg_parts = (p->n_kf + 31)/32;
            AL(W.Sfar, 36*(size_t)std::max(c->n_far, 1));
            AL(W.pc_x, W.N); AL(W.pc_r, W.N); AL(W.pc_p[0], W.N); AL(W.pc_p[1], W.N); AL(W.pc_q, W.N); AL(W.pc_g0, W.N);
            AL(W.pc_part, 5*(size_t)c->pcg_parts + 16 + 144); AL(W.pcs, 2); AL(W.pc_stat, 8);       // (pc_part: + partial r.z per interior of the solve phase, tsba_bandsv.h)
        }
        c->S_xchg = nullptr; c->xchg_wp = 0;
        if (W.band && is_multi(c)) { c->xchg_wp = std::min(W.N, bwmax + 6); AL(c->S_xchg, ((size_t)W.N + bwmax)*c->xchg_wp); }
    }
    AL(W.g, W.N); AL(W.dp, W.N + 2); AL(W.dl_pt, p->n_pt); AL(W.dl_tx, 3*(size_t)p->n_text);
    AL(W.partial, 2*(size_t)c->nb_back_max);
    AL(W.posepart, 2*((size_t)p->n_kf/21 + 2));
    AL(W.cntpart, 2*(mx_cnt/4 + mx_cnt/256 + 4));
    AL(W.st, 2); c->st_base = W.st;
    W.st_next = (W.dp_poll && p->n_kf <= SCHUR_KEEP_KF) ? W.st + 1 : nullptr;        // windows on one GPU: k_schur_t takes the previous trial's decision itself (the two copies of the state swap roles after that launch)
    AL(c->cov_log, 6*TSBA_MAX_LEVELS);
    flush_run(c);
    auto tu2 = std::chrono::steady_clock::now();
    // (a one-shot call goes straight on to the solve on the same stream: nothing touches the staged bytes before free_problem's synchronisation at the next upload)
    c->upload_pending = false;
    if (lazy && c->ev_upload && (!c->stage_p || c->copy_stream)) { if (c->stage_p) { hipEventRecord(c->ev_upload, c->stream); c->upload_pending = true; } }
    else if (hipStreamSynchronize(c->stream) != hipSuccess) { set_err(c, "upload sync failed"); return TSBA_ERR_DEVICE; }
    if (tdbg) { auto tu3 = std::chrono::steady_clock::now(); auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        fprintf(stderr, "[tsba_upload] free %.2f ms, host total %.2f ms (waiting for the plan threads %.2f ms, image section %.2f ms), final sync %.2f ms\n", ms(tu0, tu1), ms(tu1, tu2), t_plan, t_img, ms(tu2, tu3)); }
    c->uploaded = true;
    joiner.armed = false;                          // (deferred levels: their plan threads are joined by stage_level / free_problem)
    if (!c->stage_p) join_planners(c);
    return TSBA_OK;
}
int tsba_upload(void *ctx, const tsba_problem *p, const tsba_options *o) { return upload_impl(ctx, p, o, false); }

// One pyramid level onto the device: the plan's lists (its host thread is joined here), the level's reference features, the table of image planes.
static int stage_level(Ctx *c, const tsba_problem *p, int l, double *t_plan, double *t_img) {
    const tsba_options *o = &c->opt; int rc;
    if (c->stage_async && c->copy_stream && c->upload_pending) { hipStreamWaitEvent(c->copy_stream, c->ev_upload, 0); c->upload_pending = false; }
    {   auto tp0 = std::chrono::steady_clock::now();            // in pass order: the coarse levels are ready first and are staged while level 0 is still being built
        if (l < (int)c->planners.size() && c->planners[l].joinable()) c->planners[l].join();
        if (t_plan) *t_plan += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tp0).count(); }
    HostPlan &H = c->hplan[l];
    LevelDev &D = c->lev[l]; memset(&D, 0, sizeof(D));
    D.level = l; D.n_sc = H.n_sc(); D.n_pair = H.n_pair(); D.n_tg = H.n_tg(); D.n_pslot = H.n_pslot(); D.n_tslot = H.n_tslot(); D.n_sb = H.n_sb(); D.bw_rows = 6*H.bw_pose;
    for (int k = 0; k < 4; k++) { double v = p->K[k]; for (int q = 0; q < l; q++) v *= 0.5; D.K[k] = v; }
    D.img_w = p->img_w[l]; D.img_h = p->img_h[l];
#define UV(field) do { rc = dev_upload_vec(c, &D.field, H.field); if (rc) return rc; } while (0)
    UV(sc_obs); UV(sc_kf); UV(sc_pt); UV(sc_flag); UV(sc_slot); UV(sc_uv);
    UV(tg_rec); UV(tg_ppos); UV(pt_pose6); UV(pt_pair4); UV(tx_pair8); UV(pair_i); UV(pair_h); UV(pair_hpos); UV(pair_sc_off); UV(pair_tg_off); UV(pair_tg);
    UV(tg_tobs); UV(tg_kf); UV(tg_text); UV(tg_pair); UV(tg_slot);
    UV(pf_g); UV(pf_f); D.n_pf = (int)H.pf_g.size();
    if (!H.kf_order.empty()) UV(kf_order); else D.kf_order = nullptr;
    D.far_B = H.far_B; D.n_far = H.n_far();
    D.sb_far = nullptr;
    D.n_wb = (int)H.wb_kf.size(); D.wb_kf = nullptr; D.wb_idx = nullptr;
    if (H.far_B > 0 && D.n_wb > 0) { UV(wb_kf); UV(wb_idx); }
    D.far_rec = nullptr; D.n_far_ent = H.far_B > 0 ? (int)H.far_ent.size() : 0;
    if (H.far_B > 0) { rc = dev_alloc(c, &D.far_rec, std::max<size_t>(1, H.far_ent.size())); if (rc) return rc; }
    if (H.far_B > 0) { UV(far_a); UV(far_b); UV(far_off); UV(far_ent); UV(fb_id); UV(fb_pab); UV(fb_pba); UV(fb_pt_off); UV(fb_pt_s1); UV(fb_pt_s2); UV(fb_pt_lm);
        UV(fb_tx_off); UV(fb_tx_s1); UV(fb_tx_s2); UV(fb_tx_lm); }
    UV(pls_off); UV(pslot_pose); UV(pslot_pair); UV(pslot_lm); UV(tls_off); UV(tslot_pose); UV(tslot_pair); UV(tslot_lm);
    UV(sb_a); UV(sb_b); UV(sb_pab); UV(sb_pba); UV(sb_rng); UV(sb_tx_off); UV(sb_tx_s1); UV(sb_tx_s2); UV(sb_tx_lm);
    int *dp_off = nullptr, *dp_s1 = nullptr, *dp_s2 = nullptr, *dp_lm = nullptr; const int32_t *dp_cl = nullptr;
    if (H.dev_pt_pairs >= 0) {                                     // large maps: the point slot pairs by block are built on the device (tsba_devplan.h)
        rc = dev_alloc(c, &dp_off, (size_t)D.n_sb + 2); if (rc) return rc;
        const size_t tot = (size_t)std::max<int64_t>(H.dev_pt_pairs, 1);
        rc = dev_alloc(c, &dp_s1, tot); if (rc) return rc; rc = dev_alloc(c, &dp_s2, tot); if (rc) return rc; rc = dev_alloc(c, &dp_lm, tot); if (rc) return rc;
        if (!H.cl_pt_dev.empty()) { rc = dev_upload_vec(c, &dp_cl, H.cl_pt_dev); if (rc) return rc; }
        D.sb_pt_off = dp_off; D.sb_pt_s1 = dp_s1; D.sb_pt_s2 = dp_s2; D.sb_pt_lm = dp_lm;
    } else { UV(sb_pt_off); UV(sb_pt_s1); UV(sb_pt_s2); UV(sb_pt_lm); }
    UV(pose_t_off); UV(pose_t); UV(pose_h_off); UV(pose_h); UV(pose_ps_off); UV(pose_ps); UV(pose_ps_lm); UV(pose_ts_off); UV(pose_ts); UV(pose_ts_lm);
#undef UV
    if (p->n_text > 0 && p->tfeat_off[l]) {
        D.n_tfeat = p->n_tfeat[l];
        UP(D.tfeat_off, p->tfeat_off[l], (size_t)p->n_text + 1); UP(D.tfeat_raw, p->tfeat_raw[l], p->n_tfeat[l]);
        UP(D.tfeat_uv, p->tfeat_uv[l], 2*(size_t)p->n_tfeat[l]); UP(D.tfeat_ref, p->tfeat_ref[l], 8*(size_t)p->n_tfeat[l]);
    } else { std::vector<int32_t> z((size_t)p->n_text + 1, 0); UP(D.tfeat_off, z.data(), z.size()); }
    auto ti0 = std::chrono::steady_clock::now();
    if (o->use_text && p->n_tobs > 0 && p->img[l]) {
        std::vector<const uint8_t *> ptrs(p->n_kf, nullptr);
        size_t npx = (size_t)p->img_w[l]*p->img_h[l];
        const bool cached = c->use_img_cache;
        for (int k = 0; k < p->n_kf; k++) {                 // through the pinned staging mirror: one copy for the whole level
            if (!p->img[l][k]) { set_err(c, "null image pointer"); return TSBA_ERR_ARG; }
            if (o->img_on_device) { ptrs[k] = p->img[l][k]; continue; }   // resident pyramid plane (tsframe_level_ptr): used in place
            if (cached) { ptrs[k] = c->ic.dev + (size_t)c->ic_slot[(size_t)k]*c->ic.slot + c->ic.lvl_off[l]; continue; }   // the keyframe's slot of the plane cache (filled by the upload)
            rc = dev_upload(c, &ptrs[k], p->img[l][k], npx); if (rc) return rc;
        }
        const uint8_t *const *dptr = nullptr;
        rc = dev_upload(c, &dptr, (const uint8_t *const *)ptrs.data(), ptrs.size()); if (rc) return rc;
        D.img = (const uint8_t *const *)dptr;
    }
    if (t_img) *t_img += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - ti0).count();
    if (c->stage_async || H.dev_pt_pairs >= 0) flush_run(c);        // (a level staged with the upload: its bytes leave with the upload's last copy -- one copy for all levels of a single-frame call)
    if (H.dev_pt_pairs >= 0 && D.n_sb > 0) {                       // (after the copies it reads, on the stream they went over)
        hipStream_t sq = c->stage_async && c->copy_stream ? c->copy_stream : c->stream;
        hipMemsetAsync(dp_off, 0, sizeof(int)*((size_t)D.n_sb + 2), sq);
        LAUNCHK(k_sb_pairs<0>, dim3(D.n_sb), dim3(64), 0, sq, D.n_sb, D.sb_a, D.sb_b, D.pose_ps_off, D.pose_ps, D.pose_ps_lm, D.pls_off, D.pslot_pose, (const int *)dp_cl, dp_off, dp_s1, dp_s2, dp_lm);
        LAUNCHK(k_sb_scan, dim3(1), dim3(1024), 0, sq, D.n_sb, dp_off);
        LAUNCHK(k_sb_pairs<1>, dim3(D.n_sb), dim3(64), 0, sq, D.n_sb, D.sb_a, D.sb_b, D.pose_ps_off, D.pose_ps, D.pose_ps_lm, D.pls_off, D.pslot_pose, (const int *)dp_cl, dp_off, dp_s1, dp_s2, dp_lm);
    }
    if (c->stage_async && c->copy_stream && c->ev_stage[l]) { hipEventRecord(c->ev_stage[l], c->copy_stream); c->lev_wait[l] = 1; }      // the level's pass waits for this copy
    c->lev_built[l] = 1;
    return TSBA_OK;
}
// during a solve: levels of later passes whose plans are complete go to the device now, over the copy stream, next to the running pass
static int stage_ahead(Ctx *c, int ps) {
    if (!c->stage_p) return TSBA_OK;
    for (int q = ps + 1; q < c->opt.n_passes; q++) { const int l = c->opt.levels[q];
        if (c->lev_built[l] || !c->lev_planned[l]) continue;
        if (!c->plan_done[l].load(std::memory_order_acquire)) break;           // in pass order
        c->stage_async = true; const int rc = stage_level(c, c->stage_p, l, nullptr, nullptr); c->stage_async = false;
        if (rc) return rc; }
    return TSBA_OK;
}

static int reset_state(Ctx *c) {                 // one launch instead of ten small copies (each ~2.5 us on the stream)
    Work &W = c->W;
    ResetSrc A = { (const double *)c->pose0, (const double *)c->rho0, (const double *)c->theta0,
                   (const uint8_t *)c->sgood0, (const uint8_t *)c->tobs_good0, (const uint8_t *)c->tfgood0,
                   (long long)7*c->n_kf, (long long)c->n_pt, (long long)3*c->n_text, (long long)c->n_sgood, (long long)c->n_tobs, (long long)c->n_tfgood };
    long long mx = std::max(std::max(A.n_pose, A.n_rho), std::max(std::max(A.n_theta, A.n_sg), std::max(A.n_tg, A.n_tf)));
    const int nb = (int)std::min<long long>(1024, std::max<long long>(1, (mx + 255)/256));
    LAUNCHK(k_reset_state, dim3(nb), dim3(256), 0, c->stream, W, A);
    return 0;
}

static bool is_multi(const Ctx *c) { return c->world > 1 || c->force_multi; }

// In-process communicator (tsba_comm_init_local): `world` contexts of one process, one host thread each.  A collective is
// stream-sync -> device-to-host -> barrier -> rank 0 reduces in rank order (deterministic) -> barrier -> host-to-device.
// It exists so that the N > 1 code path -- sharded upload, split kernel sequence, every exchange -- runs under the test-suite on a
// one-GPU box; production multi-GPU runs use RCCL (tsba_comm_init).
struct LocalGroup {
    int world = 1; bool broken = false;
    std::vector<Ctx *> members;                    // contexts that joined (tsba_comm_init_local): their lgroup is cleared when the group goes away
    std::mutex m; std::condition_variable cv; int arrived = 0; unsigned long long gen = 0;
    std::vector<std::vector<char>> stage; std::vector<char> result;
    // The ranks of a group usually share ONE device.  A rank holds this token while it has kernels in flight (from the moment it leaves a
    // collective until its stream has drained at the next one), so that the ranks' launches do not overlap on the device: kernel
    // durations under a profiler are then those of a rank that has the GPU to itself, as in a real multi-GPU run.
    std::mutex gpu_token;
    void fail() { std::lock_guard<std::mutex> lk(m); broken = true; cv.notify_all(); }     // a member gives up: the others must not wait for it
    bool barrier() {                               // false: a member never arrived (it failed before the collective) -- do not hang
        std::unique_lock<std::mutex> lk(m);
        if (broken) return false;
        const unsigned long long g = gen;
        if (++arrived == world) { arrived = 0; gen++; cv.notify_all(); return true; }
        if (!cv.wait_for(lk, std::chrono::seconds(120), [&] { return gen != g || broken; })) { broken = true; cv.notify_all(); return false; }
        return !broken;
    }
};
static void lgroup_forget(Ctx *c) {              // the context leaves its in-process group (destroyed, or joined to an RCCL communicator)
    if (!c->lgroup) return;
    { std::lock_guard<std::mutex> lk(c->lgroup->m); for (Ctx *&m : c->lgroup->members) if (m == c) m = nullptr; }
    c->lgroup = nullptr;
}
static void local_allreduce(Ctx *c, void *buf, size_t count, ncclDataType_t dt, ncclRedOp_t op) {
    LocalGroup *G = c->lgroup;
    const size_t bytes = count*(dt == ncclDouble ? sizeof(double) : sizeof(int));
    auto fail = [&](const char *what) { c->err = std::string("ncclAllReduce (local group): ") + what; G->fail(); };
    if (hipStreamSynchronize(c->stream) != hipSuccess) { fail("stream"); return; }
    std::vector<char> &mine = G->stage[c->rank];
    mine.resize(bytes);
    if (hipMemcpy(mine.data(), buf, bytes, hipMemcpyDeviceToHost) != hipSuccess) { fail("device-to-host"); return; }
    if (c->has_token) { c->has_token = false; G->gpu_token.unlock(); }          // this rank's kernels have drained: the next rank may run
    struct Retake { Ctx *c; LocalGroup *G; ~Retake() { if (c->in_solve && !c->has_token) { G->gpu_token.lock(); c->has_token = true; } } } retake{c, G};
    if (!G->barrier()) { fail("a rank did not arrive"); return; }
    if (c->rank == 0) {
        G->result = G->stage[0];
        for (int r = 1; r < G->world; r++) {
            if (G->stage[r].size() != bytes) { G->fail(); break; }              // ranks disagree on the count: a layout bug, never sum garbage
            if (dt == ncclDouble) { double *a = (double *)G->result.data(); const double *b = (const double *)G->stage[r].data();
                if (op == ncclMax) for (size_t k = 0; k < count; k++) a[k] = a[k] > b[k] ? a[k] : b[k]; else for (size_t k = 0; k < count; k++) a[k] += b[k]; }
            else { int *a = (int *)G->result.data(); const int *b = (const int *)G->stage[r].data(); for (size_t k = 0; k < count; k++) a[k] += b[k]; }
        }
    }
    if (!G->barrier()) { fail("ranks disagree on the element count, or a rank did not arrive"); return; }
    if (hipMemcpy(buf, G->result.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) fail("host-to-device");
}
static void allreduce(Ctx *c, void *buf, size_t count, ncclDataType_t dt, ncclRedOp_t op) {
    c->x_acc += count*(dt == ncclDouble ? sizeof(double) : sizeof(int));
    if (c->lgroup) { local_allreduce(c, buf, count, dt, op); return; }
    if (!c->comm) return;                          // force_multi without a communicator: exercises the split kernels only
    ncclResult_t r = c->p_allreduce(buf, buf, count, dt, op, c->comm, c->stream);
    if (r != ncclSuccess) c->err = std::string("ncclAllReduce: ") + c->p_errstr(r);
}
static void launch_pass_init(Ctx *c, const LevelDev &D, int pass) {
    const size_t x0 = c->x_acc;
    struct XP { Ctx *c; size_t x0; ~XP() { c->x_pass = c->x_acc - x0; } } xp{c, x0};
    c->cur_bw_rows = D.bw_rows; c->S_stale = true;             // (a new pass: other free poses, other entries of S)
    c->W.hprog = c->hprog; c->W.pass_seq = ++c->pass_seq; c->W.trace_pass = pass;
    Work &W = c->W; const tsba_options &o = c->opt;
    LAUNCHK(k_pass_reset, dim3(64), dim3(256), 0, c->stream, W, o.initial_radius, o.its[pass]);
    int n = D.n_sc + D.n_tg;
    const int npb = (D.n_sc + 255)/256 + (D.n_tg + 3)/4;                    // k_participation workgroups: scene candidates | text groups
    const int ncp = (n > 0 && !is_multi(c)) ? npb : 0;                      // count partials (single GPU)
    if (n > 0) LAUNCHK(k_participation, dim3(npb), dim3(256), 0, c->stream, W, D, ncp ? 1 : 0);
    if (is_multi(c)) {                             // participation and block counts are global properties
        allreduce(c, W.kf_in, c->n_kf, ncclInt32, ncclSum);
        allreduce(c, &W.st->ns_active, 2, ncclInt32, ncclSum);
        LAUNCHK(k_kfin_multi, dim3((c->n_kf + 255)/256), dim3(256), 0, c->stream, W);
    }
    if (c->n_kf <= 64) LAUNCHK(k_gauge_wave, dim3(1), dim3(64), 0, c->stream, W, (const uint8_t *)c->kf_initial, o.state, ncp);
    else if (c->n_kf > 256) LAUNCHK(k_gauge_par, dim3(1), dim3(1024), 0, c->stream, W, (const uint8_t *)c->kf_initial, o.state, ncp, D.kf_order);
    else LAUNCHK(k_gauge, dim3(1), dim3(64), 0, c->stream, W, (const uint8_t *)c->kf_initial, o.state, ncp, D.kf_order);
    if (D.far_B > 0 && D.far_rec) { const int ne = D.n_far_ent;      // the blocks outside the band by keyframe, with the other keyframe's row (the gauge is fixed now)
        if (ne > 0) LAUNCHK(k_far_rows, dim3((ne + 255)/256), dim3(256), 0, c->stream, W, D, ne); }
    if (D.n_tg > 0) LAUNCHK(k_musigma, dim3(D.n_tg), dim3(MS_THREADS), 0, c->stream, W, D);
}
// k_mid's blocks: 256 landmarks / pairs each.  tsba_debug_options.trial_launches = 1 / 2 (the k_lin_mid experiment and its comparison partner): 128 (MID_TW: what a
// workgroup of the linearisation can take over)
// Windows (round 6): ONE wave per block -- what bounds a block is the number of cache lines its gathers pull through its compute unit's vector cache (256 points x 5 slots
// x (a 64-byte record + a 72-byte rotation) took ~10 k cycles on each of 20 compute units); 64-thread blocks put the same gathers on four times as many
static int mid_threads(const Ctx *c) { return c->n_kf <= SCHUR_KEEP_KF ? ((c->dbg.trial_launches == 1 || c->dbg.trial_launches == 2) ? MID_TW : 64) : 256; }
// (round 6: a plane has MID_PL = 8 lanes, one slot record each, a pair MID_PR = 4 lanes that share its text groups: the three kinds of block each end after
// two dependent round trips instead of up to eight)
static void mid_blocks(const Ctx *c, const LevelDev &D, int &nb_pt, int &nb_tx, int &nb_pr) { const int t = mid_threads(c);
    const int pr = MID_PR_MIN;
    nb_pt = (c->n_pt + t - 1)/t;
    if (t == 64) { nb_tx = (c->n_text + 1)/2; nb_pr = D.n_pair; }      // one-wave blocks: a plane on 32 lanes, a pair on the whole wave (a lane per value / output)
    else { nb_tx = (c->n_text + t/MID_PL - 1)/(t/MID_PL); nb_pr = (D.n_pair + t/pr - 1)/(t/pr); } }
static int pose_parts(const Ctx *c) { return c->n_kf > 126 ? (c->n_kf + 20)/21 : 0; }    // k_pose_sums workgroups (0: the pose sums stay in k_postlin / k_decide)
// pairs with a dozen scene blocks (large maps): four pairs per wave
static bool lin_small_pairs(const Ctx *c, const LevelDev &D) { return !c->dbg.no_small_pairs && D.n_pair > 0 && (long long)D.n_sc <= 24LL*D.n_pair; }
static void launch_linearize(Ctx *c, const LevelDev &D, int spec, bool skip_postlin = false) {      // skip_postlin: windows -- the first k_schur_t of the pass does k_postlin's work (SchurDec.on = 2)
    struct XL { Ctx *c; size_t x0; ~XL() { c->x_lin = c->x_acc - x0; } } xl{c, c->x_acc};
    Work &W = c->W;
    int nb_pt, nb_tx, nb_pr; mid_blocks(c, D, nb_pt, nb_tx, nb_pr); const int nb_kf = (c->n_kf + 255)/256;
    // EXPERIMENT, off by default (trial_launches = 2): k_mid inside the speculative linearisation's launch (k_lin_mid: its last workgroups to finish take the k_mid
    // blocks).  Measured in round 5: 40.7 us per launch against 13.4 + 10.6 us for the two launches -- 736 workgroups telling each other that they are done costs
    // more than the kernel boundary it replaces (tools/ticket_bench.hip, docs/ledger_r05.md 14.2)
    const unsigned lm_grid = (unsigned)((((D.n_pair + LIN_NWV - 1)/LIN_NWV + D.n_tg + 7)/8)*8);
    if (spec && W.st_next && c->lin_ticket && c->dbg.trial_launches == 2 && !is_multi(c) && mid_threads(c) == MID_TW && D.n_pair + D.n_tg > 0 && !lin_small_pairs(c, D)
        && grid_resident(c, (const void *)k_lin_mid, LIN_T, 0, (int)lm_grid)) {      // (its retained workgroups wait for every arrival: only where the whole grid is resident, as every launch that waits)
        const unsigned grid = lm_grid;
        LAUNCHK(k_lin_mid, dim3(grid), dim3(LIN_T), 0, c->stream, W, D, nb_pt, nb_tx, nb_pt + nb_tx + nb_pr, c->lin_ticket, c->lin_base);
        c->lin_base += grid;
        return;
    }
    if (D.n_pair + D.n_tg > 0) {
        if (lin_small_pairs(c, D) && D.n_tg == 0) LAUNCHK((k_linearize<MODE_FULL, 4, false>), dim3((((D.n_pair + 4*LIN_NWV - 1)/(4*LIN_NWV) + 7)/8)*8), dim3(LIN_T), 0, c->stream, W, D, spec);
        else if (lin_small_pairs(c, D)) LAUNCHK((k_linearize<MODE_FULL, 4>), dim3((((D.n_pair + 4*LIN_NWV - 1)/(4*LIN_NWV) + D.n_tg + 7)/8)*8), dim3(LIN_T), 0, c->stream, W, D, spec);
        else LAUNCHK((k_linearize<MODE_FULL, 1>), dim3((((D.n_pair + LIN_NWV - 1)/LIN_NWV + D.n_tg + 7)/8)*8), dim3(LIN_T), 0, c->stream, W, D, spec);
    }
    if (mid_threads(c) == MID_TW) LAUNCHK((k_mid<MID_TW, 6, MID_PR_MIN>), dim3(nb_pt + nb_tx + nb_pr), dim3(MID_TW), 0, c->stream, W, D, nb_pt, nb_tx, spec);
    else if (mid_threads(c) == 64) LAUNCHK((k_mid<64, 6, MID_PR_MIN>), dim3(nb_pt + nb_tx + nb_pr), dim3(64), 0, c->stream, W, D, nb_pt, nb_tx, spec);
    else LAUNCHK((k_mid<256, 4, MID_PR_MIN>), dim3(nb_pt + nb_tx + nb_pr), dim3(256), 0, c->stream, W, D, nb_pt, nb_tx, spec);
    const int multi = is_multi(c);
    const int npp = pose_parts(c);
    if (multi) {
        // local sums -> exchange buffer -> all-reduce; the consumer (k_postlin / k_decide) installs them into the right LinBuf
        if (npp) LAUNCHK(k_pose_sums_raw, dim3(npp), dim3(256), 0, c->stream, W, D, spec);
        LAUNCHK(k_sums_multi, dim3(1), dim3(256), 0, c->stream, W, D, spec, nb_pt + nb_tx + nb_pr, back_blocks_pt(c->n_pt) + back_blocks_tx(c->n_text) + nb_kf, back_blocks_pt(c->n_pt) + back_blocks_tx(c->n_text), npp);      // (k_back's blocks: its partial sums)
        allreduce(c, W.cb, 2*(size_t)W.N + 8, ncclDouble, ncclSum);
        allreduce(c, W.cbm, 1, ncclDouble, ncclMax);
        if (npp) LAUNCHK(k_pose_scale_multi, dim3(npp), dim3(256), 0, c->stream, W, spec);
    } else if (npp) LAUNCHK(k_pose_sums, dim3(npp), dim3(256), 0, c->stream, W, D, spec);
    if (!spec && !skip_postlin) LAUNCHK(k_postlin, dim3(1), dim3(256), 0, c->stream, W, D, c->opt.gradient_tolerance, nb_pt + nb_tx + nb_pr, multi, npp);
}
static int solve_lds_bytes(Ctx *c, int *use_lds) {
    size_t bytes = solve_lds_doubles(c->W.N)*sizeof(double);                                // worst case: every pose free
    *use_lds = bytes <= 160*1024 - 64;                                                      // gfx950: 160 KB of LDS per workgroup
    return *use_lds ? (int)bytes : 0;
}
static void launch_schur(Ctx *c, const LevelDev &D, int multi, SchurDec dec = SchurDec{0, 0, 0, tsba_options{}}) {
    if (c->n_kf > 126 && !c->dbg.no_schur_quad) {               // large maps: four S blocks per wave, then one wave per pose for the reduced gradient
        const int nq = D.n_sb > 0 ? (((D.n_sb + 3)/4 + 7)/8)*8 : 0, ng = ((c->n_kf + 7)/8)*8;           // (multiples of 8 workgroups: the kernel's XCD-aware mappings)
        if (D.n_tg > 0) LAUNCHK(k_schur_quad<true>, dim3(nq + ng), dim3(64), 0, c->stream, c->W, D, multi, nq, ng);
        else LAUNCHK(k_schur_quad<false>, dim3(nq + ng), dim3(64), 0, c->stream, c->W, D, multi, nq, ng);
    } else if (c->n_kf > 126) LAUNCHK(k_schur_t<1>, dim3(D.n_sb + c->n_kf), dim3(64), 0, c->stream, c->W, D, multi, 0, SchurDec{0, 0, 0, tsba_options{}});
    else LAUNCHK(k_schur_t<4>, dim3(D.n_sb + c->n_kf), dim3(256), 0, c->stream, c->W, D, multi, 0, dec);
    if (D.far_B > 0 && D.n_far > 0) {            // the blocks of E (what couples different clusters of a landmark): the same kernels on the fb_* lists, stored to W.Sfar
        LevelDev E = D;
        E.n_sb = D.n_far; E.sb_a = D.far_a; E.sb_b = D.far_b; E.sb_pab = D.fb_pab; E.sb_pba = D.fb_pba; E.sb_far = D.fb_id;
        E.sb_pt_off = D.fb_pt_off; E.sb_pt_s1 = D.fb_pt_s1; E.sb_pt_s2 = D.fb_pt_s2; E.sb_pt_lm = D.fb_pt_lm;
        E.sb_tx_off = D.fb_tx_off; E.sb_tx_s1 = D.fb_tx_s1; E.sb_tx_s2 = D.fb_tx_s2; E.sb_tx_lm = D.fb_tx_lm;
        if (c->n_kf > 126 && !c->dbg.no_schur_quad) { const int nq = (((E.n_sb + 3)/4 + 7)/8)*8;
            if (D.n_tg > 0) LAUNCHK(k_schur_quad<true>, dim3(nq), dim3(64), 0, c->stream, c->W, E, multi, nq, 0);
            else LAUNCHK(k_schur_quad<false>, dim3(nq), dim3(64), 0, c->stream, c->W, E, multi, nq, 0); }
        else if (c->n_kf > 126) LAUNCHK(k_schur_t<1>, dim3(E.n_sb), dim3(64), 0, c->stream, c->W, E, multi, 0, SchurDec{0, 0, 0, tsba_options{}});
        else LAUNCHK(k_schur_t<4>, dim3(E.n_sb), dim3(256), 0, c->stream, c->W, E, multi, 0, SchurDec{0, 0, 0, tsba_options{}});
    }
}
// kernels with more than 64 KB of dynamic LDS need the attribute once per process
static int set_solver_attrs(Ctx *c) {
    int use_lds; int lds = solve_lds_bytes(c, &use_lds);
    if (use_lds) { ATTR(k_solve_t<false>, lds);
        ATTR((k_solve_t<false, false>), lds);
        ATTR(k_solve_back<true>, std::max(lds, (int)((768 + c->W.N + 2)*sizeof(double))));
        ATTR(k_solve_back<false>, std::max(lds, (int)((768 + c->W.N + 2)*sizeof(double))));
        ATTR(k_solve_la, (int)std::min<size_t>(solve_la_lds_doubles(c->W.N)*sizeof(double), 160*1024 - 64)); }
    else {
        ATTR(k_band_solve, 156*1024);
        ATTR(k_bandp_factor, 156*1024);
        ATTR(k_cr_pivot, 156*1024);
        ATTR(k_cr_update, 156*1024);
        ATTR(k_cr_back, 156*1024);
        ATTR(k_cre_elim, 156*1024);
        ATTR(k_bandp_sepf, 156*1024);
        ATTR(k_cre_back, 156*1024);
        ATTR(k_ms_cre_back, 159*1024);
        ATTR(k_ms_cre_fwd, 100*1024);
        ATTR(k_ms_cre_root, 100*1024);
        ATTR(k_sv_linv, 156*1024);
#define MX_ATTR(SS) ATTR(k_mx_cre_fwd<SS>, 156*1024); ATTR(k_mx_cre_root<SS>, 156*1024); ATTR(k_mx_cre_back<SS>, 156*1024);
        MX_ATTR(36) MX_ATTR(42) MX_ATTR(48) MX_ATTR(54) MX_ATTR(60) MX_ATTR(66)
#undef MX_ATTR
        ATTR(k_sv_fwd_int<1>, 156*1024);
        ATTR(k_sv_fwd_int<2>, 156*1024);
        ATTR(k_sv_back_int<1>, 156*1024);
        ATTR(k_sv_back_int<2>, 156*1024);
        ATTR(k_sv_tree_back<1>, 150*1024);
        ATTR(k_sv_tree_back<2>, 150*1024);
        ATTR(k_bandp_backsub<1>, 156*1024);
        ATTR(k_bandp_backsub<2>, 156*1024);
        ATTR(k_band_backsub<1>, 156*1024);
        ATTR(k_band_backsub<2>, 156*1024);
        ATTR(k_band_backsub<3>, 156*1024);
        ATTR(k_solve_t<true>, (int)(solve_diag_lds_doubles()*sizeof(double)));
        ATTR(k_chol_panel, (CH_NB + 64)*(CH_NB + 1)*(int)sizeof(double));
        ATTR(k_chol_update, 2*64*(CH_NB + 1)*(int)sizeof(double));
        ATTR(k_chol_backsub, (CH_NB*(CH_NB + 1) + 2*CH_NB + 8*CH_NB)*(int)sizeof(double));
        ATTR(k_chol_fwd, (CH_NB*(CH_NB + 1) + 2*CH_NB + 8*CH_NB)*(int)sizeof(double));
    }
    return 0;
}
static void launch_dense_chol(Ctx *c, Work &W, int bw);
// dense solve of the reduced camera system: LDS kernel for small windows, multi-workgroup blocked Cholesky otherwise
static bool ms_available(const Ctx *c);
static int sv_reserve(Ctx *c);
static void launch_sv_prepare(Ctx *c, double *xreset);
static void launch_solve(Ctx *c) {
    Work &W = c->W;
    c->sv_prepared = false;
    int use_lds; int lds = solve_lds_bytes(c, &use_lds);
    if (use_lds) {                                  // small windows: one workgroup, S in LDS.  solve_variant 1: the two-panel-wave schedule of tsba_solve.h (A/B runs)
        const size_t la = solve_la_lds_doubles(W.N)*sizeof(double);
        if ((c->dbg.solve_variant == 1 || c->dbg.solve_variant == 2) && la <= 160*1024 - 64) LAUNCHK(k_solve_la, dim3(1), dim3(SOLVE_THREADS), (int)la, c->stream, W, c->dbg.solve_variant == 2 ? 0 : 1);
        else if (c->dbg.solve_variant == 4) LAUNCHK((k_solve_t<false, false>), dim3(1), dim3(SOLVE_THREADS), lds, c->stream, W, 0);      // (the diagonal blocks through the LDS scratch: A/B and bit-identity runs)
        else LAUNCHK(k_solve_t<false>, dim3(1), dim3(SOLVE_THREADS), lds, c->stream, W, 0);
        return; }
    if (c->band_stream && c->band_parts > 1) {      // partitioned: interiors in parallel + separator system (tsba_bandp.h)
        const int bwp = std::max(6, c->cur_bw_rows), cbp = bandp_chunk_blocks(bwp), P = c->band_parts;
        const int bwsep = 2*bwp - 6, cbs = band_chunk_blocks(bwsep);
        Work &Ws = c->Wsep; Ws.st = W.st; Ws.ldS = (P - 1)*bwp; Ws.N = (P - 1)*bwp;
        if (!c->sep_cr) hipMemsetAsync(c->Ssep, 0, sizeof(double)*((size_t)Ws.ldS*Ws.ldS + Ws.ldS), c->stream);
        LAUNCHK(k_bandp_factor, dim3(P), dim3(BANDP_T), (int)(bandp_lds_doubles(bwp, cbp)*sizeof(double)), c->stream, W, bwp, cbp, P, c->Lcol, c->Lb, c->Tbuf);
        if (c->sep_cr && (W.ring || (c->dbg.sep_solver != 3 && c->dbg.sep_solver != 4)))      // block pool: border products + separator assembly in one launch (4: the three launches, for A/B runs)
            LAUNCHK(k_bandp_sepf, dim3(W.ring ? P + 1 : P - 1), dim3(BSF_T), (int)(bandp_sepf_lds_doubles()*sizeof(double)), c->stream, W, bwp, P, (const double *)c->Tbuf, (const double *)c->Lb, c->Ssep, Ws.g, Ws.nfree);
        else {
        hipMemsetAsync(c->Bpart, 0, sizeof(double)*(size_t)P*BANDP_NS*((size_t)bwp*bwp + bwp), c->stream);        // (slices of short interiors stay empty)
        LAUNCHK(k_bandp_border, dim3(P, BANDP_NS), dim3(256), (int)((2*(size_t)BANDP_JC*bwp*6 + 6*BANDP_JC)*sizeof(double)), c->stream, W, bwp, P, (const double *)c->Lb, c->Bpart);
        LAUNCHK(k_bandp_sep, dim3(P - 1), dim3(256), 0, c->stream, W, bwp, P, (const double *)c->Tbuf, (const double *)c->Bpart, c->Ssep, Ws.ldS, Ws.g, Ws.nfree, (int)c->sep_cr);
        }
        if (c->sep_cr) {                  // separator system by block cyclic reduction (tsba_bandcr.h): log2(P - 1) levels
            const int mmax = cr_mmax(W.ring, P, W.ring_g);
            int mlev = mmax;                  // levels h < mlev.  Ring: the loop's separators need h <= G/2 (the root and the ghost are merged at the root, no level for them),
            if (W.ring) { mlev = W.ring_g;    // a tail's separator RING_OFF - j the level of the lowest set bit of j (j < the number of tail interiors)
                for (int hh = 1; hh < P - W.ring_g; hh <<= 1) mlev = std::max(mlev, 2*hh); }
            const int lab0 = W.ring && P > W.ring_g ? RING_OFF - (P - W.ring_g) + 1 : 0;          // lowest separator label (a ring with a tail counts down from RING_OFF)
            const int lp = (int)(cr_pivot_lds_doubles(bwp)*sizeof(double)), lu = (int)(cr_update_lds_doubles(bwp)*sizeof(double)), lb = (int)(cr_back_lds_doubles(bwp)*sizeof(double));
            int htop = 1;
            if (c->dbg.sep_solver != 3 || W.ring) {      // one launch per level (tsba_bandcre.h); 3: the pivot / update / back kernels of tsba_bandcr.h
                const int le = (int)(cre_elim_lds_doubles(bwp)*sizeof(double)), lbk = (int)(cre_back_lds_doubles(bwp)*sizeof(double));
                c->cre_epoch++;                                   // (this factorisation's ordinal: what the K workgroups of a pivot tell each other they have loaded for, k_cre_elim)
                auto pivots = [&](int h, int &kb) { kb = lab0/(2*h); const int klast = (mmax - 1 - h)/(2*h); return std::max(0, klast - kb + 1); };     // pivots (2 k + 1) h, k = kb ..
                for (int h = 1; h < mlev; h <<= 1) {
                    int kb; const int npiv = pivots(h, kb); if (npiv <= 0) { htop = h; continue; }
                    const int K = std::max(1, std::min(TSBA_CRE_KMAX, 224/npiv));     // workgroups per pivot (they share its product and stores)
                    LAUNCHK(k_cre_elim, dim3(npiv*K), dim3(CRE_T), le, c->stream, W, Ws, bwp, P, h, 0, K, kb, c->CRcontrib, c->CRfac, c->CRgate, c->cre_epoch); htop = h; }
                LAUNCHK(k_cre_elim, dim3(1), dim3(CRE_T), le, c->stream, W, Ws, bwp, P, 0, W.ring ? 2 : 1, 1, 0, c->CRcontrib, c->CRfac, c->CRgate, c->cre_epoch);
                // back substitution: a launch per level -- or one launch through the inverse factors and products of the solve phase (k_sv_linv + k_cre_back_tree)
                // where the iterative path needs those anyway (maps with long-range blocks) or the tree is deep enough to pay for k_sv_linv (28 us at 48-row
                // separators against 10.5 us per level)
                if ((c->far_B > 0 || (htop >= 32 && bwp <= 60)) && ms_available(c) && !(c->dbg.sv_per_level & 2) && c->dbg.pcg_refactor != 1 && mmax >= 2 && grid_resident(c, (const void *)k_cre_back_tree, SV_CT, 0, mmax - 1) && sv_reserve(c) == TSBA_OK) {
                    launch_sv_prepare(c, Ws.Sy);
                    LAUNCHK(k_cre_back_tree, dim3(mmax - 1), dim3(SV_CT), 0, c->stream, W, Ws, bwp, P, (const double *)c->CRfac, c->sv);
                } else
                for (int h = htop; h >= 1; h >>= 1) { int kb; const int npiv = pivots(h, kb); if (npiv > 0) LAUNCHK(k_cre_back, dim3(npiv), dim3(CRE_BT), lbk, c->stream, W, Ws, bwp, P, h, kb, (const double *)c->CRfac); }
            } else {
            for (int h = 1; h < mmax; h <<= 1) {
                const int npiv = (mmax + 2*h - 1)/(2*h);           // >= the pivots (2k + 1) h < m; workgroups past the end return
                LAUNCHK(k_cr_pivot, dim3(npiv), dim3(CR_T), lp, c->stream, W, Ws, bwp, P, h, 0);
                LAUNCHK(k_cr_update, dim3(2*npiv + 1), dim3(CR_T), lu, c->stream, W, Ws, bwp, P, h, npiv);
                htop = h;
            }
            LAUNCHK(k_cr_pivot, dim3(1), dim3(CR_T), lp, c->stream, W, Ws, bwp, P, 0, 1);
            LAUNCHK(k_cr_back, dim3(1), dim3(CR_T), lb, c->stream, W, Ws, bwp, P, 0, 1);
            for (int h = htop; h >= 1; h >>= 1)
                LAUNCHK(k_cr_back, dim3((mmax + 2*h - 1)/(2*h)), dim3(CR_T), lb, c->stream, W, Ws, bwp, P, h, 0);
            }
        } else {
        const int ldss = (int)(band_lds_doubles(bwsep, cbs)*sizeof(double)), nus = (bwsep + 63)/64;
        LAUNCHK(k_band_solve, dim3(1), dim3(SOLVE_THREADS), ldss, c->stream, Ws, bwsep, cbs, c->Lcol_sep);
        if (nus <= 1) LAUNCHK(k_band_backsub<1>, dim3(1), dim3(BAND_BS_T), ldss, c->stream, Ws, bwsep, (const double *)c->Lcol_sep);
        else if (nus == 2) LAUNCHK(k_band_backsub<2>, dim3(1), dim3(BAND_BS_T), ldss, c->stream, Ws, bwsep, (const double *)c->Lcol_sep);
        else LAUNCHK(k_band_backsub<3>, dim3(1), dim3(BAND_BS_T), ldss, c->stream, Ws, bwsep, (const double *)c->Lcol_sep);
        }
        const int nup = (bwp + 63)/64, ldsp = (int)((2*(size_t)BAND_CK*(2*(size_t)bwp*6 + 32) + 6*BAND_RINGB + 2*bwp + 64)*sizeof(double));
        if (nup <= 1) LAUNCHK(k_bandp_backsub<1>, dim3(P), dim3(BAND_BS_T), ldsp, c->stream, W, bwp, P, (const double *)c->Lcol, (const double *)c->Lb, (const double *)Ws.Sy);
        else LAUNCHK(k_bandp_backsub<2>, dim3(P), dim3(BAND_BS_T), ldsp, c->stream, W, bwp, P, (const double *)c->Lcol, (const double *)c->Lb, (const double *)Ws.Sy);
        LAUNCHK(k_bandp_dp, dim3((W.n_kf + 255)/256), dim3(256), 0, c->stream, W);
        return;
    }
    if (c->band_stream) {                                          // narrow band: one workgroup streams down the band (tsba_band.h)
        const int bws = std::max(6, c->cur_bw_rows), cb = band_chunk_blocks(bws);
        if (c->dbg.verbose) fprintf(stderr, "[launch_solve] band stream bw %d cb %d lds %zu B\n", bws, cb, band_lds_doubles(bws, cb)*sizeof(double));
        LAUNCHK(k_band_solve, dim3(1), dim3(SOLVE_THREADS), (int)(band_lds_doubles(bws, cb)*sizeof(double)), c->stream, W, bws, cb, c->Lcol);
        const int nu = (bws + 63)/64, ldsb = (int)(band_lds_doubles(bws, cb)*sizeof(double));      // tasks per lane of the back substitution
        if (nu <= 1) LAUNCHK(k_band_backsub<1>, dim3(1), dim3(BAND_BS_T), ldsb, c->stream, W, bws, (const double *)c->Lcol);
        else if (nu == 2) LAUNCHK(k_band_backsub<2>, dim3(1), dim3(BAND_BS_T), ldsb, c->stream, W, bws, (const double *)c->Lcol);
        else LAUNCHK(k_band_backsub<3>, dim3(1), dim3(BAND_BS_T), ldsb, c->stream, W, bws, (const double *)c->Lcol);
        return;
    }
    launch_dense_chol(c, W, std::min(c->cur_bw_rows, W.N));
}
// multi-workgroup blocked Cholesky (tsba_chol.h) of the system in `W` (band bound bw rows below a pose block; bw = N: dense)
static void launch_dense_chol(Ctx *c, Work &W, int bw) {
    const int N = W.N;                                             // worst case: every keyframe free
    LAUNCHK(k_chol_rhs, dim3((N + 255)/256), dim3(256), 0, c->stream, W);
    const int lds_diag = (int)(solve_diag_lds_doubles()*sizeof(double));
    const int lds_panel = (CH_NB + 64)*(CH_NB + 1)*(int)sizeof(double);
    const int lds_upd = 2*64*(CH_NB + 1)*(int)sizeof(double);
    for (int j0 = 0; j0 < N; j0 += CH_NB) {
        LAUNCHK(k_solve_t<true>, dim3(1), dim3(SOLVE_THREADS), lds_diag, c->stream, W, j0);
        // the host only knows the worst case n = N; a shorter last block (nb < NB) still has the rhs row below it
        const int wr = std::max(0, std::min(bw, N - (j0 + 6)));        // band rows below the block, + 1 for the rhs row
        LAUNCHK(k_chol_panel, dim3(wr/64 + 1), dim3(CH_T), lds_panel, c->stream, W, j0, bw);
        const int nt = (wr + 1 + 63)/64;
        if (wr > 0) LAUNCHK(k_chol_update, dim3(nt*(nt + 1)/2), dim3(CH_T), lds_upd, c->stream, W, j0, bw);
    }
    const int lds_bs = (CH_NB*(CH_NB + 1) + 2*CH_NB + 8*CH_NB)*(int)sizeof(double);
    LAUNCHK(k_chol_backsub, dim3(1), dim3(1024), lds_bs, c->stream, W, bw);
}

// ---- solve phase of the partitioned band solver for T right-hand sides (tsba_bandms.h): needs the factor of the last launch_solve of this level
static bool ms_available(const Ctx *c) { return c->band_stream && c->band_parts > 1 && c->sep_cr && !c->W.ring && c->dbg.sep_solver != 3; }
static int ms_reserve(Ctx *c, int T) {           // buffers for T columns (kept until a larger request or another problem size)
    const size_t n6 = (size_t)c->W.N, labels = (size_t)cr_mmax(0, c->band_parts, 0) + 1, sdim = (size_t)std::max(6, c->cur_bw_rows);
    const size_t per = 4*n6 + 6*labels*sdim, need = per*(size_t)T*sizeof(double);
    if (need > c->ms_bytes) { if (c->ms_alloc) { hipStreamSynchronize(c->stream); hipFree(c->ms_alloc); } c->ms_alloc = nullptr; c->ms_bytes = 0;
        if (hipMalloc((void **)&c->ms_alloc, need) != hipSuccess) { set_err(c, "hipMalloc (multi-right-hand-side buffers)"); return TSBA_ERR_DEVICE; }
        c->ms_bytes = need; }
    double *q = c->ms_alloc; MsBuf &M = c->ms; M.T = T;
    M.R = q; q += n6*T; M.Wm = q; q += n6*T; M.V = q; q += n6*T; M.X = q; q += n6*T;
    M.G = q; q += labels*sdim*T; M.Z = q; q += labels*sdim*T; M.Xs = q; q += labels*sdim*T; M.Cg = q; q += 2*labels*sdim*T; M.G2 = q;
    c->ms_cap = T;
    return TSBA_OK;
}
static void launch_ms_solve(Ctx *c, bool mx = false) {            // M.R -> M.X.  mx: the separators in product form (tsba_bandmx.h) -- c->sv holds the inverse factors of this factorisation
    Work &W = c->W; const MsBuf &M = c->ms;
    const int bwp = std::max(6, c->cur_bw_rows), P = c->band_parts, B = bwp/6, ncg = (M.T + 63)/64;
    Work &Ws = c->Wsep; Ws.st = W.st;
    mx = mx && bwp >= 36 && bwp <= MX_SMAX && bwp % 6 == 0;
    const size_t ldsf = ms_cre_lds_doubles(bwp, 1)*sizeof(double), ldsb = (ms_cre_lds_doubles(bwp, 3) + 8*(size_t)(bwp + 2))*sizeof(double), ldsx = mx_lds_doubles(bwp)*sizeof(double);
    LAUNCHK(k_ms_fwd_int, dim3(P, ncg), dim3(64), 0, c->stream, W, bwp, P, (const double *)c->Lcol, M);
    LAUNCHK(k_ms_sep_rhs, dim3(P - 1, ncg), dim3(64*B), 0, c->stream, W, bwp, P, (const double *)c->Lcol, (const double *)c->Lb, M);
    const int mmax = cr_mmax(0, P, 0);
    auto pivots = [&](int h, int &kb) { kb = 0; const int klast = (mmax - 1 - h)/(2*h); return mmax - 1 - h < 0 ? 0 : std::max(0, klast + 1); };
    int htop = 0;
    const double *Li = c->sv.Li, *Lid = c->sv.Lid;
    // (the product-form kernels are instantiated per separator size: compile-time loop bounds and LDS offsets)
#define MX_CASES(CALL) switch (bwp) { case 36: CALL(36) break; case 42: CALL(42) break; case 48: CALL(48) break; case 54: CALL(54) break; case 60: CALL(60) break; case 66: CALL(66) break; default: break; }
    for (int h = 1; h < mmax; h <<= 1) { int kb; const int npiv = pivots(h, kb); if (npiv <= 0) continue;
        if (mx) {
#define MX_FWD(SS) LAUNCHK(k_mx_cre_fwd<SS>, dim3(npiv, ncg), dim3(MX_T), ldsx, c->stream, W, Ws, bwp, P, h, kb, M, Li, Lid);
            MX_CASES(MX_FWD)
#undef MX_FWD
        } else LAUNCHK(k_ms_cre_fwd, dim3(npiv, ncg), dim3(MS_CT), ldsf, c->stream, W, Ws, bwp, P, h, kb, (const double *)c->CRfac, M);
        htop = h; }
    if (mx) {
#define MX_ROOT(SS) LAUNCHK(k_mx_cre_root<SS>, dim3(1, ncg), dim3(MX_T), ldsx, c->stream, W, bwp, P, M, Li, Lid);
        MX_CASES(MX_ROOT)
#undef MX_ROOT
    } else LAUNCHK(k_ms_cre_root, dim3(1, ncg), dim3(256), ldsf, c->stream, W, Ws, bwp, P, (const double *)c->CRfac, M);
    for (int h = htop; h >= 1; h >>= 1) { int kb; const int npiv = pivots(h, kb);
        if (npiv <= 0) continue;
        if (mx) {
#define MX_BACK(SS) LAUNCHK(k_mx_cre_back<SS>, dim3(npiv, ncg), dim3(MX_T), ldsx, c->stream, W, Ws, bwp, P, h, kb, M, Li);
            MX_CASES(MX_BACK)
#undef MX_BACK
        } else LAUNCHK(k_ms_cre_back, dim3(npiv, ncg), dim3(MS_CT), ldsb, c->stream, W, Ws, bwp, P, h, kb, (const double *)c->CRfac, M); }
#undef MX_CASES
    LAUNCHK(k_ms_back_border, dim3(P, ncg), dim3(BB_T), 0, c->stream, W, bwp, P, (const double *)c->Lb, M);
    LAUNCHK(k_ms_back_int, dim3(P, ncg), dim3(64), 0, c->stream, W, bwp, P, (const double *)c->Lcol, M);
}

// ---- the same for ONE right-hand side (tsba_bandsv.h): x = M^-1 (rs * r) into c->sv.X.  launch_sv_prepare once per factorisation (the
// separators' inverse factors), then any number of launch_sv_solve
static int sv_reserve(Ctx *c) {
    const size_t n6 = ((size_t)c->W.N + 1) & ~(size_t)1, labels = (size_t)cr_mmax(0, c->band_parts, 0) + 1, sdim = (size_t)std::max(6, c->cur_bw_rows);
    const size_t need = (4*n6 + 7*labels*sdim + 3*labels*sdim*sdim)*sizeof(double);
    if (need > c->sv_bytes) { if (c->sv_alloc) { hipStreamSynchronize(c->stream); hipFree(c->sv_alloc); } c->sv_alloc = nullptr; c->sv_bytes = 0;
        if (hipMalloc((void **)&c->sv_alloc, need) != hipSuccess) { set_err(c, "hipMalloc (solve-phase buffers)"); return TSBA_ERR_DEVICE; }
        c->sv_bytes = need; }
    double *q = c->sv_alloc; MsBuf &M = c->sv; M.T = 1;
    M.Li = q; q += labels*sdim*sdim; M.Pp = q; q += 2*labels*sdim*sdim;
    M.R = q; q += n6; M.Wm = q; q += n6; M.V = q; q += n6; M.X = q; q += n6;
    M.G = q; q += labels*sdim; M.Z = q; q += labels*sdim; M.Xs = q; q += labels*sdim; M.Cg = q; q += 2*labels*sdim; M.G2 = q; q += labels*sdim; M.Lid = q;
    return TSBA_OK;
}
// bound of an interior's length in pose blocks (bandp_part: the device partitions the FREE poses -- at most n_kf -- into at most band_parts interiors of at
// least 2 B + 2 blocks; where it has to take fewer interiors they stay below twice that)
static int sv_lmax_of(int n_kf, int B, int P) { return std::max(n_kf/std::max(1, P) + 2, 5*B + 8); }
static int sv_lmax(const Ctx *c) { return sv_lmax_of(c->n_kf, std::max(6, c->cur_bw_rows)/6, c->band_parts); }
static void launch_sv_prepare(Ctx *c, double *xreset) {
    const int bwp = std::max(6, c->cur_bw_rows), P = c->band_parts, mmax = cr_mmax(0, P, 0);
    if (mmax > 0) LAUNCHK(k_sv_linv, dim3(mmax), dim3(SV_LT), sv_linv_lds_doubles(bwp)*sizeof(double), c->stream, c->W, bwp, P, (const double *)c->CRfac, (const double *)c->Ssep, c->sv, xreset);
    c->sv_prepared = true;
}
static void launch_sv_solve(Ctx *c, const double *r, double rs, const double *rdot = nullptr, double *rz_part = nullptr, SvUpd upd = SvUpd{0, 0, 0, 0}) {
    Work &W = c->W; const MsBuf &M = c->sv;
    const int bwp = std::max(6, c->cur_bw_rows), P = c->band_parts, B = bwp/6, lmax = sv_lmax(c);
    Work &Ws = c->Wsep; Ws.st = W.st;
    const size_t ldf = sv_fwd_lds_doubles(B)*sizeof(double), ldb = sv_back_lds_doubles(B, lmax)*sizeof(double);
    const int mmax = cr_mmax(0, P, 0);
    auto pivots = [&](int h) { const int klast = (mmax - 1 - h)/(2*h); return mmax - 1 - h < 0 ? 0 : std::max(0, klast + 1); };
    int htop = 0;
    for (int h = 1; h < mmax; h <<= 1) if (pivots(h) > 0) htop = h;
    // the highest level has one pivot (3 h >= 2 h >= the number of separators): its forward step, the root and its back substitution are one workgroup's work
    const bool fuse_top = htop > 0 && pivots(htop) == 1;
    const bool tb_fits = B <= 10 ? grid_resident(c, (const void *)k_sv_tree_back<1>, SV_T, ldb, P) : grid_resident(c, (const void *)k_sv_tree_back<2>, SV_T, ldb, P);
    const int tree = fuse_top && !(c->dbg.sv_per_level & 1) && grid_resident(c, (const void *)k_sv_cre_tree, SV_CT, 0, mmax - 1);           // the whole tree in one launch (k_sv_cre_tree): its workgroups poll each other
    if (B <= 10) LAUNCHK(k_sv_fwd_int<1>, dim3(P), dim3(SV_T), ldf, c->stream, W, bwp, P, (const double *)c->Lcol, (const double *)c->Lb, r, rs, M, tree, upd);
    else LAUNCHK(k_sv_fwd_int<2>, dim3(P), dim3(SV_T), ldf, c->stream, W, bwp, P, (const double *)c->Lcol, (const double *)c->Lb, r, rs, M, tree, upd);
    const bool tree_back = tree && !(c->dbg.sv_per_level & 8) && tb_fits;        // ... and the interiors' back substitution in the tree's launch (k_sv_tree_back)
    if (tree_back) {
        if (B <= 10) LAUNCHK(k_sv_tree_back<1>, dim3(P), dim3(SV_T), ldb, c->stream, W, bwp, P, htop, lmax, (const double *)c->Lcol, (const double *)c->Lb, M, rdot, rz_part);
        else LAUNCHK(k_sv_tree_back<2>, dim3(P), dim3(SV_T), ldb, c->stream, W, bwp, P, htop, lmax, (const double *)c->Lcol, (const double *)c->Lb, M, rdot, rz_part);
        return; }
    if (tree) LAUNCHK(k_sv_cre_tree, dim3(mmax - 1), dim3(SV_CT), 0, c->stream, W, Ws, bwp, P, htop, M);
    else {
        for (int h = 1; h <= htop; h <<= 1) { const int npiv = pivots(h); if (npiv <= 0 || (fuse_top && h == htop)) continue;
            LAUNCHK(k_sv_cre_fwd, dim3(npiv), dim3(SV_CT), 0, c->stream, W, Ws, bwp, P, h, 0, M); }
        if (fuse_top) LAUNCHK(k_sv_cre_top, dim3(1), dim3(SV_CT), 0, c->stream, W, Ws, bwp, P, htop, M);
        else LAUNCHK(k_sv_cre_root, dim3(1), dim3(SV_CT), 0, c->stream, W, bwp, P, M);
        for (int h = htop; h >= 1; h >>= 1) { const int npiv = pivots(h);
            if (npiv > 0 && !(fuse_top && h == htop)) LAUNCHK(k_sv_cre_back, dim3(npiv), dim3(SV_CT), 0, c->stream, W, Ws, bwp, P, h, 0, M); }
    }
    if (B <= 10) LAUNCHK(k_sv_back_int<1>, dim3(P), dim3(SV_T), ldb, c->stream, W, bwp, P, lmax, (const double *)c->Lcol, (const double *)c->Lb, M, rdot, rz_part);
    else LAUNCHK(k_sv_back_int<2>, dim3(P), dim3(SV_T), ldb, c->stream, W, bwp, P, lmax, (const double *)c->Lcol, (const double *)c->Lb, M, rdot, rz_part);
}

// The reduced system of one LM trial: a direct solve, or -- band + long-range blocks -- conjugate gradients preconditioned with the band
// solver (tsba_pcg.h).  The host enqueues iteration k only once the device has reached iteration k - 2 (pinned progress word), so a solve
// that converges wastes two iterations of empty launches; every rank of a sharded run iterates on its own copy of the summed system.
static void launch_solve_full(Ctx *c, const LevelDev &D) {
    launch_solve(c);
    if (D.far_B <= 0) return;
    Work &W = c->W;
    const int nbp = c->pcg_parts, B = std::max(6, c->cur_bw_rows)/6;
    const int nmv = (c->n_kf + PCG_MW - 1)/PCG_MW, pq_off = 3*nbp + 8;          // the matvec's workgroups (a wave per keyframe) and where their partial p.q go (nmv <= 2 nbp)
    const int cap = c->dbg.pcg_max_it > 0 ? c->dbg.pcg_max_it : 200;
    const double tol = c->dbg.pcg_tol_exp > 0 ? pow(10.0, -(double)c->dbg.pcg_tol_exp) : 1e-10, tol2 = tol*tol;
    const unsigned int seq = ++c->pcg_seq;
    // the inverse factors of the separators (k_sv_linv), once per factorisation: the single-vector solve phase and the product form of the many-column one use them
    const bool svok = ms_available(c) && c->dbg.pcg_refactor != 1 && sv_reserve(c) == TSBA_OK;       // (pcg_refactor = 3: as 0 with r.z by its own kernel, for A/B runs)
    if (svok && !c->sv_prepared) launch_sv_prepare(c, nullptr);       // (the direct solve of a chain has run it already: its back substitution uses the same products)
    // Enlarged conjugate gradients on the many-column solve phase of the band solver (ECG_T columns per application of M^-1): an option (pcg_block = 2).
    // It halves the iterations where the coupling outside the band is a few hundred blocks (outlying eigenvalues, captured 32 at a time), but an
    // application costs 0.8 ms at 5000 keyframes against 0.13 ms of the single-vector solve phase (tsba_bandsv.h) -- measured when the single-vector
    // iteration still re-ran the factorisation (0.57 ms), ms per solve single / enlarged: two loop closures 410 / 303, 1 % long-range points 247 / 320
    const bool want_block = c->dbg.pcg_block == 2;
    if (ms_available(c) && want_block && ms_reserve(c, std::max(ECG_T, c->ms_cap)) == TSBA_OK) {
        const int nch = (c->n_kf + ECG_CH - 1)/ECG_CH; const size_t n6 = (size_t)W.N;
        const size_t need = (2*n6*ECG_T + (size_t)nch*2*(ECG_T*ECG_T + 1) + 4*(size_t)ECG_T*ECG_T + 4*ECG_T + 16)*sizeof(double);
        bool ok = true;
        if (need > c->ecg_bytes) { if (c->ecg_alloc) { hipStreamSynchronize(c->stream); hipFree(c->ecg_alloc); } c->ecg_alloc = nullptr; c->ecg_bytes = 0;
            ok = hipMalloc((void **)&c->ecg_alloc, need) == hipSuccess; if (ok) c->ecg_bytes = need; }
        if (ok) {
            EcgBuf &E = c->ecg; double *q = c->ecg_alloc;
            E.P = q; q += n6*ECG_T; E.Q = q; q += n6*ECG_T; E.part = q; q += (size_t)nch*2*(ECG_T*ECG_T + 1); E.Cm = q; q += ECG_T*ECG_T; E.Lm = q; q += ECG_T*ECG_T + ECG_T;
            E.Y = q; q += ECG_T*ECG_T; E.y1 = q; q += ECG_T; E.scal = q; E.nchunk = nch;
            const int Tk = c->ms.T; c->ms.T = ECG_T; const MsBuf M = c->ms;
            auto finished_e = [&](int it) {
                if (!c->hprog || it < 2) return false;
                const auto tw = std::chrono::steady_clock::now();
                for (int spin = 0;; spin++) {
                    const unsigned long long w = ((volatile unsigned long long *)c->hprog)[1];
                    if ((unsigned int)(w >> 32) == seq) { if (w & 1) return true; if ((int)((w & 0xffffffffu) >> 1) + 2 >= it) return false; }
                    PlanPool::cpu_relax();
                    if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - tw > std::chrono::seconds(5)) return false;
                }
            };
            LAUNCHK(k_ecg_begin, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, M);
            launch_ms_solve(c, svok);
            LAUNCHK(k_ecg_gram, dim3(nch), dim3(256), 0, c->stream, W, (const double *)M.X, (const double *)M.X, (const double *)nullptr, (const double *)M.R, (const double *)M.X, E);
            LAUNCHK(k_ecg_small, dim3(1), dim3(1024), 0, c->stream, W, E, 0, 0, seq, tol2);
            LAUNCHK(k_ecg_update, dim3(nbp), dim3(256), 0, c->stream, W, M, E, 2, 1);
            int it = 0;
            for (; it < cap; it++) {
                if (finished_e(it)) break;
                LAUNCHK(k_ecg_matvec, dim3(nbp), dim3(256), 0, c->stream, W, D, B, E);
                LAUNCHK(k_ecg_gram, dim3(nch), dim3(256), 0, c->stream, W, (const double *)E.P, (const double *)E.Q, (const double *)M.R, (const double *)nullptr, (const double *)nullptr, E);
                LAUNCHK(k_ecg_small, dim3(1), dim3(1024), 0, c->stream, W, E, 1, it, seq, tol2);
                LAUNCHK(k_ecg_update, dim3(nbp), dim3(256), 0, c->stream, W, M, E, 1, 0);
                launch_ms_solve(c, svok);
                LAUNCHK(k_ecg_gram, dim3(nch), dim3(256), 0, c->stream, W, (const double *)E.Q, (const double *)M.X, (const double *)nullptr, (const double *)M.R, (const double *)M.X, E);
                LAUNCHK(k_ecg_small, dim3(1), dim3(1024), 0, c->stream, W, E, 2, it, seq, tol2);
                LAUNCHK(k_ecg_update, dim3(nbp), dim3(256), 0, c->stream, W, M, E, 2, 0);
            }
            LAUNCHK(k_ecg_finish, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, it);
            c->ms.T = Tk;
            return;
        }
    }
    // Loop closures (E touches a few dozen keyframes): the band solve corrected by the low-rank part exactly (tsba_wb.h) is the preconditioner --
    // set up once per trial (one solve phase with k columns, the k x k matrix), then a band solve and a k x k Cholesky per application
    bool wb = D.n_wb > 0 && ms_available(c) && c->dbg.far_solver != 3 && ms_reserve(c, std::max(6*D.n_wb, c->ms_cap)) == TSBA_OK;
    const int kk = 6*D.n_wb;
    if (wb) {
        const size_t need = (3*(size_t)kk*kk + 2*(size_t)kk + (size_t)W.N + ((size_t)kk + 1)*kk + 4*(size_t)kk + 64 + 2*(size_t)D.n_wb + 16)*sizeof(double);
        if (need > c->wb_bytes) { if (c->wb_alloc) { hipStreamSynchronize(c->stream); hipFree(c->wb_alloc); } c->wb_alloc = nullptr; c->wb_bytes = 0;
            if (hipMalloc((void **)&c->wb_alloc, need) == hipSuccess) c->wb_bytes = need; else wb = false; }
    }
    double *K2 = nullptr;
    if (wb) {
        WbBuf &Bw = c->wb; double *q = c->wb_alloc;
        Bw.k = kk; Bw.n_u = D.n_wb; Bw.wb_kf = D.wb_kf; Bw.wb_idx = D.wb_idx;
        Bw.Gm = q; q += (size_t)kk*kk; Bw.T1 = q; q += (size_t)kk*kk; K2 = q; q += (size_t)kk*kk; Bw.xu = q; q += kk; Bw.vu = q; q += kk; Bw.z = q; q += W.N;
        Work &Wk = c->Wk; memset(&Wk, 0, sizeof(Wk));
        Wk.N = kk; Wk.n_kf = D.n_wb; Wk.ldS = kk; Wk.band = 0; Wk.st = W.st;
        Wk.S = q; q += ((size_t)kk + 1)*kk; Wk.Sy = q; q += kk + 8; Wk.g = q; q += kk; Wk.dp = q; q += kk; Wk.LDbuf = q; q += kk + 8;
        Wk.fidx = (int *)q; Wk.nfree = Wk.fidx + D.n_wb + 2;
        const int Tk = c->ms.T; c->ms.T = kk; const MsBuf M = c->ms;
        LAUNCHK(k_wb_init, dim3(1), dim3(64), 0, c->stream, Wk.fidx, Wk.nfree, D.n_wb);
        LAUNCHK(k_wb_units, dim3(1024), dim3(256), 0, c->stream, W, M, Bw);
        launch_ms_solve(c, svok);
        LAUNCHK(k_wb_gather, dim3(std::min(1024, (kk*kk + 255)/256)), dim3(256), 0, c->stream, W, M, Bw);
        LAUNCHK(k_wb_EG, dim3(D.n_wb), dim3(256), 0, c->stream, W, D, Bw);
        LAUNCHK(k_wb_K2, dim3(std::min(2048, (kk*kk + 255)/256)), dim3(256), 0, c->stream, W, Bw, K2);
        c->ms.T = Tk;
    }
    bool wb_factored = false;
    auto correct = [&](const double *yp, double ys) {              // z = M_W^-1 r from y = M^-1 r = ys * yp[]
        WbBuf &Bw = c->wb; Work &Wk = c->Wk; MsBuf M = c->ms; M.T = kk;
        LAUNCHK(k_wb_rhs, dim3(1), dim3(512), 0, c->stream, W, Bw, yp, ys, Wk.g);
        if (!wb_factored) {                                        // once per LM trial: the k x k factor (in place, over a copy), with this right-hand side riding along
            hipMemcpyAsync(Wk.S, K2, sizeof(double)*(size_t)kk*kk, hipMemcpyDeviceToDevice, c->stream);
            launch_dense_chol(c, Wk, kk); wb_factored = true;
        } else {                                                   // later applications: the two substitutions on that factor (0.28 ms of factorisation each before)
            const int lds_bs = (CH_NB*(CH_NB + 1) + 2*CH_NB + 8*CH_NB)*(int)sizeof(double);
            LAUNCHK(k_chol_rhs, dim3((kk + 255)/256), dim3(256), 0, c->stream, Wk);
            LAUNCHK(k_chol_fwd, dim3(1), dim3(1024), lds_bs, c->stream, Wk, kk);
            LAUNCHK(k_chol_backsub, dim3(1), dim3(1024), lds_bs, c->stream, Wk, kk);
        }
        LAUNCHK(k_wb_Gw, dim3(1), dim3(512), 0, c->stream, W, Bw, (const double *)Wk.dp);
        LAUNCHK(k_wb_Ex, dim3(D.n_wb), dim3(64), 0, c->stream, W, D, Bw, (const double *)Bw.xu);
        LAUNCHK(k_wb_apply, dim3(512), dim3(256), 0, c->stream, W, M, Bw, yp, ys);
    };
    if (wb) { correct(W.Sy, -1.0); LAUNCHK(k_pcg_begin, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, (const double *)c->wb.z, 1.0); }
    else LAUNCHK(k_pcg_begin, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, (const double *)W.Sy, -1.0);
    if (wb) LAUNCHK(k_pcg_rcheck, dim3(1), dim3(64), 0, c->stream, W, -1, nbp, 0.0);
    auto finished = [&](int it) {                                  // true: the device reported convergence (or the end of the pass); else waits until it is within two iterations
        if (!c->hprog || it < 2) return false;
        const auto tw = std::chrono::steady_clock::now();
        for (int spin = 0;; spin++) {
            const unsigned long long w = ((volatile unsigned long long *)c->hprog)[1];
            if ((unsigned int)(w >> 32) == seq) { if (w & 1) return true; if ((int)((w & 0xffffffffu) >> 1) + 2 >= it) return false; }
            PlanPool::cpu_relax();
            if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - tw > std::chrono::seconds(5)) return false;      // never hang on it
        }
    };
    // M^-1 on the residual: the solve phase of the partitioned band solver on the factor this trial's first solve left (tsba_bandms.h); where
    // that is not available (a single interior, the sequential separator solve) the factorisation is run again with the residual as right-hand side
    // (measured at 5000 keyframes, one column: 1.3 ms per application against 0.57 ms for the factorisation re-run -- the solve phase pays for 64
    // columns whether it has them or not; it is the default only for the block variants.  pcg_refactor = 2 selects it for the single-vector iteration)
    const bool ms = !wb && ms_available(c) && c->dbg.pcg_refactor == 2 && ms_reserve(c, std::max(1, c->ms_cap)) == TSBA_OK;
    const bool sv = !ms && svok && (c->dbg.pcg_refactor == 0 || c->dbg.pcg_refactor == 3);      // (the single-vector solve phase, tsba_bandsv.h: the default)
    const bool fused_dot = sv && !wb && c->band_parts <= 144 && c->dbg.pcg_refactor == 0; const int rz2_off = 5*nbp + 16;
    const double *zp = wb ? c->wb.z : W.Sy; double zs = wb ? 1.0 : -1.0;
    int it = 0;
    for (; it < cap; it++) {
        if (finished(it)) break;
        LAUNCHK(k_pcg_matvec, dim3(nmv), dim3(64*PCG_MW), 0, c->stream, W, D, it, seq, B, tol2, (it > 0 && fused_dot) ? rz2_off : 0, (it > 0 && fused_dot) ? c->band_parts : nbp, pq_off, zp, zs);
        if (ms) { LAUNCHK(k_pcg_update, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, it, nbp, pq_off, nmv, c->ms.R, 1.0);
            const int Tk = c->ms.T; c->ms.T = 1; launch_ms_solve(c, svok); c->ms.T = Tk; zp = c->ms.X; zs = 1.0; }
        else if (sv && fused_dot && !(c->dbg.sv_per_level & 4)) {      // the iteration's update step (alpha; x, r) inside the first kernel of the preconditioner application
            launch_sv_solve(c, W.pc_r, 1.0, W.pc_r, W.pc_part + rz2_off, SvUpd{1, it, nmv, pq_off});
            zp = c->sv.X; zs = 1.0; }
        else if (sv) { LAUNCHK(k_pcg_update, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, it, nbp, pq_off, nmv, c->sv.R, 1.0);
            if (wb) LAUNCHK(k_pcg_rcheck, dim3(1), dim3(64), 0, c->stream, W, it, nbp, 1e-20);      // |r| <= 1e-10 |b|
            if (fused_dot) launch_sv_solve(c, c->sv.R, 1.0, c->sv.R, W.pc_part + rz2_off);      // (r.z comes along: no k_pcg_dot)
            else launch_sv_solve(c, c->sv.R, 1.0);
            zp = c->sv.X; zs = 1.0;
            if (wb) { correct(c->sv.X, 1.0); zp = c->wb.z; } }
        else { LAUNCHK(k_pcg_update, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, it, nbp, pq_off, nmv, W.g, -1.0);
            if (wb) LAUNCHK(k_pcg_rcheck, dim3(1), dim3(64), 0, c->stream, W, it, nbp, 1e-20);      // |r| <= 1e-10 |b|
            launch_solve(c);
            if (wb) { correct(W.Sy, -1.0); zp = c->wb.z; zs = 1.0; } }
        if (!fused_dot) LAUNCHK(k_pcg_dot, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, zp, zs);
    }
    LAUNCHK(k_pcg_finish, dim3(nbp), dim3(PCG_ET), 0, c->stream, W, it);
}

static void launch_decide(Ctx *c, const LevelDev &D) {
    Work &W = c->W;
    int nb_pt, nb_tx, nb_pr; mid_blocks(c, D, nb_pt, nb_tx, nb_pr); const int nb_kf = (c->n_kf + 255)/256;
    const int nb_all = back_blocks_pt(c->n_pt) + back_blocks_tx(c->n_text) + nb_kf;
    LAUNCHK(k_decide, dim3(1), dim3(256), 0, c->stream, W, D, nb_all, nb_pt + nb_tx + nb_pr, c->opt, (int)is_multi(c), pose_parts(c));
}
// one LM iteration: reduced system -> pose step -> back-substitution / candidate -> speculative linearisation at the
// candidate -> decision (on acceptance the speculative LinBuf simply becomes the current one)
static void launch_step(Ctx *c, const LevelDev &D, bool decide_prev = false, bool first_fused = false) {
    struct XT { Ctx *c; size_t x0; ~XT() { c->x_trial = c->x_acc - x0; } } xt{c, c->x_acc};
    Work &W = c->W;
    int nb_pt, nb_tx, nb_pr; mid_blocks(c, D, nb_pt, nb_tx, nb_pr); const int nb_kf = (c->n_kf + 255)/256;
    // block-sparse S: what no block of the plan covers must read as zero.  The streaming / partitioned band solvers leave S intact and a pass
    // writes the same entries in every trial (the free poses are fixed at its start), so the band is cleared once per pass; the in-place
    // Cholesky of the wide-band path needs it before every assembly -- and so does a sharded run (a rank assembles only its own blocks; the
    // other entries hold the sums the last exchange unpacked)
    if ((int64_t)D.n_sb < (int64_t)c->n_kf*(c->n_kf + 1)/2 && (!c->band_stream || c->S_stale || is_multi(c))) {
        hipMemsetAsync(c->S_alloc, 0, sizeof(double)*c->S_count, c->stream); c->S_stale = false;
        if (D.far_B > 0) hipMemsetAsync(W.Sfar, 0, sizeof(double)*36*(size_t)std::max(D.n_far, 1), c->stream); }
    const int bb_pt = back_blocks_pt(c->n_pt), bb_tx = back_blocks_tx(c->n_text), nb_all = bb_pt + bb_tx + nb_kf;      // k_back's blocks
    const bool fused_decide = W.st_next != nullptr && D.far_B <= 0;       // the decision on a trial is taken by the NEXT trial's k_schur_t (the last trial's by k_decide after the loop)
    if (fused_decide && (decide_prev || first_fused)) {     // (first_fused: the pass's first trial -- workgroup 0 of the assembly does k_postlin's work on the first linearisation, nobody decides anything)
        launch_schur(c, D, 0, SchurDec{decide_prev ? 1 : 2, nb_all, nb_pt + nb_tx + nb_pr, c->opt});
        std::swap(W.st, W.st_next);                 // from here on the launches see the state that launch wrote
    } else launch_schur(c, D, (int)is_multi(c));
    if (is_multi(c)) {                             // one exchange per LM trial: the reduced normal equations
        if (c->S_xchg) {                           // band storage: only the band's entries travel
            const size_t nx = ((size_t)W.N + (W.ring ? c->xchg_wp - 6 : 0))*c->xchg_wp;
            const int nbp = (int)std::min<size_t>(2048, (nx + 255)/256);
            LAUNCHK(k_band_pack, dim3(nbp), dim3(256), 0, c->stream, W, c->S_xchg, c->xchg_wp, 0);
            allreduce(c, c->S_xchg, nx, ncclDouble, ncclSum);
            LAUNCHK(k_band_pack, dim3(nbp), dim3(256), 0, c->stream, W, c->S_xchg, c->xchg_wp, 1);
        } else allreduce(c, c->S_alloc, c->S_count, ncclDouble, ncclSum);
        if (D.far_B > 0 && D.n_far > 0) allreduce(c, W.Sfar, 36*(size_t)D.n_far, ncclDouble, ncclSum);      // the blocks outside the band
        allreduce(c, W.g, W.N, ncclDouble, ncclSum);
        LAUNCHK(k_damp_multi, dim3((c->n_kf + 255)/256), dim3(256), 0, c->stream, W);
    }
    if (W.dp_poll && D.far_B <= 0) {               // small window: solver (workgroup 0) and back-substitution (three blocks per workgroup, polling the step) in one launch
        int use_lds; const int lds = solve_lds_bytes(c, &use_lds);
        const int ldsb = std::max(lds, (int)((768 + W.N + 2)*sizeof(double)));
        if (c->dbg.solve_variant == 5) LAUNCHK(k_solve_back<false>, dim3(1 + (nb_all + 2)/3), dim3(SOLVE_THREADS), ldsb, c->stream, W, D, bb_pt, bb_tx, nb_all);      // (A/B: the diagonal blocks through the LDS scratch)
        else LAUNCHK(k_solve_back<true>, dim3(1 + (nb_all + 2)/3), dim3(SOLVE_THREADS), ldsb, c->stream, W, D, bb_pt, bb_tx, nb_all);
    } else {
        launch_solve_full(c, D);
        LAUNCHK(k_back, dim3(nb_all), dim3(256), 0, c->stream, W, D, bb_pt, bb_tx);
    }
    launch_linearize(c, D, 1);
    if (!fused_decide) launch_decide(c, D);
}

static int enqueue_pack(Ctx *c);
int tsba_solve(void *ctx, tsba_report *r) {
    Ctx *c = (Ctx *)ctx; if (!c || !r) return TSBA_ERR_ARG;
    if (c) c->packed = false;
    if (!c->uploaded) { set_err(c, "no problem uploaded"); return TSBA_ERR_STATE; }
    hipSetDevice(c->device);
    memset(r, 0, sizeof(*r));
    const tsba_options &o = c->opt;
    { int rca = set_solver_attrs(c); if (rca) return rca; }
    struct Poison { Poison(int v) { g_lds_poison = v; } ~Poison() { g_lds_poison = 0; } } poison(c->dbg.lds_poison);
    struct Token { Ctx *c; Token(Ctx *c_) : c(c_) { if (c->lgroup) { c->in_solve = true; c->lgroup->gpu_token.lock(); c->has_token = true; } }
                   ~Token() { if (c->lgroup) { c->in_solve = false; if (c->has_token) { hipStreamSynchronize(c->stream); c->has_token = false; c->lgroup->gpu_token.unlock(); } } } } token(c);
    auto t0 = std::chrono::steady_clock::now();
    int rc = reset_state(c); if (rc) return rc;
    if (c->far_B > 0) hipMemsetAsync(c->W.pc_stat, 0, 8*sizeof(int), c->stream);
    // windows on one GPU (the same contexts that take a trial's decision inside the next trial's assembly): a pass begins and ends with one launch each
    auto fastp = [&](const LevelDev &D) { return c->W.st_next != nullptr && c->n_kf <= 64 && D.far_B <= 0 && !is_multi(c) && !c->pose_only && !c->dbg.pass_launches && D.n_sc + D.n_tg > 0; };
    bool fast_any = false; for (int ps = 0; ps < o.n_passes; ps++) if (c->lev_built[o.levels[ps]] ? fastp(c->lev[o.levels[ps]]) : (c->W.st_next != nullptr && c->n_kf <= 64 && !c->dbg.pass_launches)) fast_any = true;
    bool log_pending = false;                     // the pass before this one left its final state in W.st only
    int ms_ahead = -1;                            // the pass whose mu / sigma the previous pass's k_pass_end has computed
    for (int ps = 0; ps < o.n_passes; ps++) {
        if (!c->lev_built[o.levels[ps]]) {            // a level the upload left for now (one-shot call on a small window): its plan is ready or nearly so
            if (!c->stage_p) { set_err(c, "level not staged"); return TSBA_ERR_STATE; }
            c->stage_async = true; rc = stage_level(c, c->stage_p, o.levels[ps], nullptr, nullptr); c->stage_async = false; if (rc) return rc; }
        if (c->lev_wait[o.levels[ps]]) { hipStreamWaitEvent(c->stream, c->ev_stage[o.levels[ps]], 0); c->lev_wait[o.levels[ps]] = 0; }
        const LevelDev &D = c->lev[o.levels[ps]];
        const bool pose_path = c->pose_only && !is_multi(c);
        // PoseOptim: the pass's LM steps in one launch (k_pose_pass) where its workgroups are all resident at once, otherwise a launch per step
        const bool pose_one_launch = pose_path && !c->dbg.pass_launches && grid_resident(c, (const void *)k_pose_pass, POSE_WG, 0, pose_grid(D));
        if (pose_path) {                                     // k_pass_reset + k_participation + k_gauge + k_musigma in one launch
            c->W.hprog = c->hprog; c->W.pass_seq = ++c->pass_seq;
            LAUNCHK(k_pose_begin, dim3(D.n_tg + 1), dim3(MS_THREADS), 0, c->stream, c->W, D, o.initial_radius, o.its[ps], (const uint8_t *)c->kf_initial,
                    pose_one_launch ? c->W.ppart : (double *)nullptr, 3*28*pose_grid(D), log_pending ? c->st_log + ps - 1 : (LmState *)nullptr);
            log_pending = false;
        } else if (fastp(D)) {
            // windows: k_pass_begin (tsba_kernels_pass.h).  The participation arrays are clear (k_reset_state / the last pass's k_pass_end); the text
            // observations' mu / sigma are there already if the last pass's k_pass_end computed them for this level
            c->cur_bw_rows = D.bw_rows; c->S_stale = true; c->x_pass = 0;
            c->W.hprog = c->hprog; c->W.pass_seq = ++c->pass_seq; c->W.trace_pass = ps;
            // (at most PB_WG workgroups walk k_participation's npb blocks: every arrival at the ticket is a device-wide fence and an atomic on one word --
            // 30 - 40 ns each, one after the other: tools/ticket_bench.hip)
            const int npb = (D.n_sc + 255)/256 + (D.n_tg + 3)/4, nwg = std::min(npb, PB_WG), n_ms = ms_ahead == ps ? 0 : D.n_tg;
            LAUNCHK(k_pass_begin, dim3(nwg + n_ms), dim3(MS_THREADS), 0, c->stream, c->W, D, o.initial_radius, o.its[ps], (const uint8_t *)c->kf_initial, o.state,
                               npb, nwg, n_ms, log_pending ? c->st_log + ps - 1 : (LmState *)nullptr, c->ticket);
            log_pending = false;
        } else {
            if (log_pending) { CK(hipMemcpyAsync(c->st_log + ps - 1, c->W.st, sizeof(LmState), hipMemcpyDeviceToDevice, c->stream)); log_pending = false; }
            launch_pass_init(c, D, ps);
        }
        // The kernels of an LM iteration return at once when the pass has converged, but each still costs a launch (~4 us):
        // the host reads the pinned progress word and stays at most two iterations ahead of the device -- no API call, no
        // synchronisation -- so a pass that converges early wastes two iterations of empty launches instead of all the rest.
        auto converged = [&](int it) {
            if (!c->hprog || is_multi(c)) return false;
            const auto tw = std::chrono::steady_clock::now();
            for (int spin = 0;; spin++) {
                const unsigned long long w = *(volatile unsigned long long *)c->hprog;
                if ((unsigned int)(w >> 32) == c->W.pass_seq) { if (w & 1) return true; if ((int)((w & 0xffffffffu) >> 1) + 2 > it) break; }
                else if (it < 2) break;                              // the device has not reached this pass yet
                PlanPool::cpu_relax();
                if ((spin & 1023) == 1023 && std::chrono::steady_clock::now() - tw > std::chrono::seconds(2)) break;   // never hang on it
            }
            return false;
        };
        if (pose_path) {                                     // PoseOptim: one launch per LM iteration (tsba_pose.h)
            const int G = pose_grid(D);
            int k_last = -1;                                 // (k_pose_pass leaves the final state in pst[0])
            if (pose_one_launch) LAUNCHK(k_pose_pass, dim3(G), dim3(POSE_WG), 0, c->stream, c->W, D, o, G, o.its[ps]);
            else {
                LAUNCHK(k_pose_iter, dim3(G), dim3(POSE_WG), 0, c->stream, c->W, D, o, -1, G);
                for (int k = 0; k <= o.its[ps]; k++) {       // launch k decides trial k - 1 and prepares trial k
                    if (k >= 1 && converged(k - 1)) break;
                    LAUNCHK(k_pose_iter, dim3(G), dim3(POSE_WG), 0, c->stream, c->W, D, o, k, G);
                    k_last = k;
                }
                if (k_last < 0) k_last = 0;
            }
            // outlier pass + installation of the pass's result (one extra workgroup) in one launch
            LAUNCHK(k_outlier, dim3((D.n_sc + 63)/64 + D.n_tg + 1), dim3(64), 0, c->stream, c->W, D, o.chi2_mono[ps], o.chi2_text[ps],
                               o.text_bad_ratio, o.outlier_scene, o.outlier_text, (const PoseState *)(c->W.pst + ((k_last + 1) & 1)));
            log_pending = true;                              // (kept by the next pass's k_pose_begin, or by k_solve_end)
            rc = stage_ahead(c, ps); if (rc) return rc;      // (one-shot calls: the later passes' levels, over the copy stream while this pass runs)
            continue;
        }
        // windows: k_postlin (the first linearisation's scaling, cost, gradient test: 8.6 us of one workgroup) inside the first trial's assembly
        const bool fuse_first = fastp(D) && o.its[ps] > 0 && c->dbg.trial_launches != 3;
        launch_linearize(c, D, 0, fuse_first);
        int n_trials = 0;
        for (int it = 0; it < o.its[ps]; it++) {
            if (converged(it) && !(fuse_first && n_trials == 0)) break;
            launch_step(c, D, n_trials > 0, fuse_first && n_trials == 0); n_trials++;
            if (it >= 1) { rc = stage_ahead(c, ps); if (rc) return rc; }      // (with two iterations queued the device does not run dry while the host stages)
        }
        if (n_trials > 0 && c->W.st_next != nullptr && D.far_B <= 0) launch_decide(c, D);      // windows: the decision on the last trial (the others were taken by the following trial's k_schur_t)
        if (fastp(D)) {
            // windows: the outlier pass, the NEXT pass's mu / sigma (when its level is on the device already) and the clearing of the participation arrays
            // in one launch; this pass's final state is kept by the next pass's k_pass_begin (or by k_solve_end)
            const bool outl = (o.outlier_scene || o.outlier_text) && D.n_sc + D.n_tg > 0;
            const int nb_out = outl ? (D.n_sc + 63)/64 + D.n_tg : 0;
            const LevelDev *Dn = nullptr;
            if (ps + 1 < o.n_passes && c->lev_built[o.levels[ps + 1]] && fastp(c->lev[o.levels[ps + 1]]) && c->lev[o.levels[ps + 1]].n_tg > 0) { const int ln = o.levels[ps + 1];
                if (c->lev_wait[ln]) { hipStreamWaitEvent(c->stream, c->ev_stage[ln], 0); c->lev_wait[ln] = 0; }      // (staged over the copy stream during this pass's trials: long since there)
                Dn = &c->lev[ln]; }
            const int n_ms = Dn ? Dn->n_tg : 0;
            LAUNCHK(k_pass_end, dim3((nb_out + 3)/4 + n_ms + 1), dim3(MS_THREADS), 0, c->stream, c->W, D, Dn ? *Dn : D, nb_out, n_ms, c->musig2[c->musig_sel ^ 1],
                               o.chi2_mono[ps], o.chi2_text[ps], o.text_bad_ratio, o.outlier_scene, o.outlier_text);
            if (Dn) { c->musig_sel ^= 1; c->W.musig = c->musig2[c->musig_sel]; ms_ahead = ps + 1; }
            if (c->cov_text >= 0 && c->cov_text < c->n_text && ps < TSBA_MAX_LEVELS) LAUNCHK(k_record_vtx, dim3(1), dim3(64), 0, c->stream, c->W, c->cov_text, c->cov_log + 6*ps);
            log_pending = true;
            continue;
        }
        if (o.outlier_scene || o.outlier_text)
            if (D.n_sc + D.n_tg > 0) LAUNCHK(k_outlier, dim3((D.n_sc + 63)/64 + D.n_tg), dim3(64), 0, c->stream, c->W, D,
                                                          o.chi2_mono[ps], o.chi2_text[ps], o.text_bad_ratio, o.outlier_scene, o.outlier_text, (const PoseState *)nullptr);
        if (c->cov_text >= 0 && c->cov_text < c->n_text && ps < TSBA_MAX_LEVELS) LAUNCHK(k_record_vtx, dim3(1), dim3(64), 0, c->stream, c->W, c->cov_text, c->cov_log + 6*ps);
        CK(hipMemcpyAsync(c->st_log + ps, c->W.st, sizeof(LmState), hipMemcpyDeviceToDevice, c->stream));
        if (fast_any) LAUNCHK(k_part_clear, dim3(8), dim3(256), 0, c->stream, c->W);       // (a later pass may begin with k_pass_begin)
    }
    if (c->world > 1) {                           // every landmark was optimised by its owner only
        int nl = c->n_pt + 3*c->n_text;
        if (nl > 0) {
            LAUNCHK(k_delta_multi, dim3((nl + 255)/256), dim3(256), 0, c->stream, c->W, (const double *)c->rho0, (const double *)c->theta0, 0);
            if (c->n_pt) allreduce(c, c->W.dl_pt, c->n_pt, ncclDouble, ncclSum);
            if (c->n_text) allreduce(c, c->W.dl_tx, 3*(size_t)c->n_text, ncclDouble, ncclSum);
            LAUNCHK(k_delta_multi, dim3((nl + 255)/256), dim3(256), 0, c->stream, c->W, (const double *)c->rho0, (const double *)c->theta0, 1);
        }
    }
    // the passes' final states straight into pinned host memory (a kernel's stores: no copy engine, no staging)
    LAUNCHK(k_solve_end, dim3(1), dim3(64), 0, c->stream, c->W, c->st_log, o.n_passes, log_pending ? 1 : 0, c->st_host, (unsigned int *)(c->st_host + TSBA_MAX_LEVELS) + 8);
    int *pcg_host = (int *)(c->st_host + TSBA_MAX_LEVELS);          // (the pinned block has room for 8 ints behind the pass snapshots)
    if (c->far_B > 0) CK(hipMemcpyAsync(pcg_host, c->W.pc_stat, 8*sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (c->pack_in_solve) { rc = enqueue_pack(c); if (rc) return rc; }
    CK(hipStreamSynchronize(c->stream));
    CK(hipGetLastError());
    c->packed = c->pack_in_solve;
    if (!c->err.empty() && c->err.rfind("ncclAllReduce", 0) == 0) return TSBA_ERR_COMM;
    auto t1 = std::chrono::steady_clock::now();
    r->t_solve_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    r->n_passes = o.n_passes;
    long long prev_lin = 0, prev_cost = 0;
    for (int ps = 0; ps < o.n_passes; ps++) {
        const LmState &s = c->st_host[ps];
        r->iters[ps] = s.it; r->accepted[ps] = s.accepted; r->termination[ps] = s.term;
        r->cost0[ps] = s.cost0; r->cost1[ps] = s.x_cost;
        r->n_sblock[ps] = s.ns_active; r->n_tblock[ps] = s.nt_active;
        r->n_bad_scene[ps] = s.n_bad_scene; r->n_bad_tfeat[ps] = s.n_bad_tfeat; r->n_bad_text[ps] = s.n_bad_text;
        long long evals = (s.n_lin - prev_lin) + (s.n_cost - prev_cost);
        r->n_resid_evals += evals*(2LL*s.ns_active + 8LL*s.nt_active);
        prev_lin = s.n_lin; prev_cost = s.n_cost;
        if (s.term == 5) r->status = TSBA_ERR_NUMERIC;
    }
    { int use_lds; solve_lds_bytes(c, &use_lds); const LevelDev &Dl = c->lev[o.levels[o.n_passes - 1]];
      r->solver_path = (c->pose_only && !is_multi(c)) ? TSBA_SOLVER_POSE : use_lds ? TSBA_SOLVER_LDS
          : Dl.far_B > 0 ? (Dl.n_wb > 0 && ms_available(c) && c->dbg.far_solver != 3 ? TSBA_SOLVER_BAND_LOWRANK : TSBA_SOLVER_BAND_PCG)
          : !c->band_stream ? TSBA_SOLVER_DENSE : c->band_parts <= 1 ? TSBA_SOLVER_BAND : c->W.ring ? TSBA_SOLVER_RING : c->sep_cr ? TSBA_SOLVER_BAND_CR : TSBA_SOLVER_BAND_PART; }
    r->poll_timeouts = (int32_t)((const unsigned int *)pcg_host)[8];
    // PoseOptim in one launch per pass (k_pose_pass): every workgroup keeps the LM state redundantly and assumes that all of them read the same polled sums -- a poll that ran
    // into its bound in ONE workgroup (a dispatch stall of ~100 ms beside another context) would let the copies part silently (round-5 advisor).  A give-up anywhere
    // during such a solve: the answer is not used, the solve runs again from its start point with a launch per LM step (k_pose_iter: no workgroup waits for another)
    if (c->pose_only && !is_multi(c) && r->poll_timeouts != 0 && !c->dbg.pass_launches && !c->pose_retry) {
        c->pose_retry = true; c->dbg.pass_launches = 1;
        const int32_t gave_up = r->poll_timeouts;
        const int rc2 = tsba_solve(ctx, r);
        c->dbg.pass_launches = 0; c->pose_retry = false;
        if (rc2 == TSBA_OK) r->poll_timeouts += gave_up;       // (reported: the caller sees that the first attempt was abandoned)
        return rc2;
    }
    if (c->far_B > 0) { r->pcg_iterations = pcg_host[0]; r->pcg_systems = pcg_host[1]; r->pcg_max_iterations = pcg_host[2]; r->pcg_unconverged = pcg_host[3]; r->pcg_stagnated = pcg_host[4]; }
    return TSBA_OK;
}

// the results of a solve -- LM state, the current poses / inverse depths / plane parameters, the three flag arrays -- gathered into one block on the
// device (which of the two parameter buffers is current is device-side state) and brought over by ONE copy into pinned memory: seven synchronous
// copies into pageable memory, the first of them only to learn an index, were 0.22 ms of a 3.8 ms local-BA call
struct DlLayout { size_t o_pose, o_rho, o_theta, o_sg, o_to, o_tf, total; };
static DlLayout dl_layout(const Ctx *c) {
    DlLayout L; auto up = [](size_t v) { return (v + 15) & ~(size_t)15; };
    L.o_pose = up(sizeof(LmState)); L.o_rho = L.o_pose + up(sizeof(double)*7*(size_t)c->n_kf); L.o_theta = L.o_rho + up(sizeof(double)*(size_t)c->n_pt);
    L.o_sg = L.o_theta + up(sizeof(double)*3*(size_t)c->n_text); L.o_to = L.o_sg + up((size_t)c->n_sgood); L.o_tf = L.o_to + up((size_t)c->n_tobs); L.total = L.o_tf + up((size_t)c->n_tfgood);
    return L;
}
__global__ __launch_bounds__(256) void k_pack_results(Work W, DlLayout L, unsigned char *out, int n_kf, int n_pt, int n_text, int n_sg, int n_to, int n_tf) {
    const LmState *st = W.st; const int cur = st->cur & 1;
    const size_t t = (size_t)blockIdx.x*256 + threadIdx.x, nt = (size_t)gridDim.x*256;
    double *o_pose = (double *)(out + L.o_pose), *o_rho = (double *)(out + L.o_rho), *o_theta = (double *)(out + L.o_theta);
    if (t < sizeof(LmState)/4) ((int *)out)[t] = ((const int *)st)[t];
    for (size_t k = t; k < 7*(size_t)n_kf; k += nt) o_pose[k] = W.pose[cur][k];
    for (size_t k = t; k < (size_t)n_pt; k += nt) o_rho[k] = W.rho[cur][k];
    for (size_t k = t; k < 3*(size_t)n_text; k += nt) o_theta[k] = W.theta[cur][k];
    for (size_t k = t; k < (size_t)n_sg; k += nt) out[L.o_sg + k] = W.sgood[k];
    for (size_t k = t; k < (size_t)n_to; k += nt) out[L.o_to + k] = W.tobs_good[k];
    for (size_t k = t; k < (size_t)n_tf; k += nt) out[L.o_tf + k] = W.tfgood[k];
}
// The one-shot entry points (round 6) pack the results INSIDE the solve, behind its last kernel and before its one synchronisation -- small blocks (a pose-only
// call: 5 KB, a window: 60 KB) as the kernel's own stores into pinned host memory, larger ones through the device block and one copy: tsba_download then
// only copies out of the pinned block (no launch, no second synchronisation: 0.06 ms of a 0.66 ms tsba_pose_optim call).
#define TSBA_PACK_DIRECT_MAX ((size_t)256 << 10)
static int dl_reserve(Ctx *c, const DlLayout &L) {
    if (L.total > c->dl_bytes) {
        CK(hipStreamSynchronize(c->stream));
        if (c->dl_dev) hipFree(c->dl_dev); if (c->dl_host) hipHostFree(c->dl_host);
        c->dl_dev = nullptr; c->dl_host = nullptr; c->dl_bytes = 0;
        const size_t cap = L.total + L.total/4;
        CK(hipMalloc((void **)&c->dl_dev, cap)); CK(hipHostMalloc((void **)&c->dl_host, cap, hipHostMallocDefault)); c->dl_bytes = cap;
    }
    return TSBA_OK;
}
static int enqueue_pack(Ctx *c) {                 // (no synchronisation: the caller's)
    const DlLayout L = dl_layout(c);
    int rc = dl_reserve(c, L); if (rc) return rc;
    const bool direct = L.total <= TSBA_PACK_DIRECT_MAX;
    const size_t work = std::max<size_t>({7*(size_t)c->n_kf, (size_t)c->n_pt, 3*(size_t)c->n_text, (size_t)c->n_sgood, (size_t)c->n_tobs, (size_t)c->n_tfgood, 64});
    LAUNCHK(k_pack_results, dim3((unsigned)std::min<size_t>(direct ? 64 : 1024, (work + 255)/256)), dim3(256), 0, c->stream, c->W, L, direct ? c->dl_host : c->dl_dev, c->n_kf, c->n_pt, c->n_text, c->n_sgood, c->n_tobs, c->n_tfgood);
    if (!direct) CK(hipMemcpyAsync(c->dl_host, c->dl_dev, L.total, hipMemcpyDeviceToHost, c->stream));
    return TSBA_OK;
}
int tsba_download(void *ctx, tsba_problem *p) {
    Ctx *c = (Ctx *)ctx; if (!c || !p) return TSBA_ERR_ARG;
    if (!c->uploaded) { set_err(c, "no problem uploaded"); return TSBA_ERR_STATE; }
    hipSetDevice(c->device);
    static_assert(sizeof(LmState) % 4 == 0, "LmState is copied as words");
    const DlLayout L = dl_layout(c);
    if (c->packed) c->packed = false;             // (the solve that has just ended left the block in pinned memory)
    else { int rc = enqueue_pack(c); if (rc) return rc;
        CK(hipStreamSynchronize(c->stream)); CK(hipGetLastError()); }
    memcpy(p->pose, c->dl_host + L.o_pose, sizeof(double)*7*(size_t)c->n_kf);
    if (c->n_pt) memcpy(p->rho, c->dl_host + L.o_rho, sizeof(double)*(size_t)c->n_pt);
    if (c->n_text) memcpy(p->theta, c->dl_host + L.o_theta, sizeof(double)*3*(size_t)c->n_text);
    if (c->n_sgood) memcpy(p->sgood, c->dl_host + L.o_sg, (size_t)c->n_sgood);
    if (c->n_tobs) memcpy(p->tobs_good, c->dl_host + L.o_to, (size_t)c->n_tobs);
    if (c->n_tfgood) memcpy(p->tfgood, c->dl_host + L.o_tf, (size_t)c->n_tfgood);
    return TSBA_OK;
}

static int one_shot(void *ctx, tsba_problem *p, const tsba_options *o, tsba_report *r) {
    auto t0 = std::chrono::steady_clock::now();
    int rc = upload_impl(ctx, p, o, true); if (rc) return rc;
    auto t1 = std::chrono::steady_clock::now();
    ((Ctx *)ctx)->pack_in_solve = true;
    rc = tsba_solve(ctx, r);
    ((Ctx *)ctx)->pack_in_solve = false;
    { Ctx *c = (Ctx *)ctx; join_planners(c); c->stage_p = nullptr;      // (*p is the caller's: nothing may be staged from it after this call)
      for (int l = 0; l < (int)c->lev_planned.size(); l++) if (c->lev_planned[l] && !c->lev_built[l]) c->uploaded = false; }   // a failed solve left a level unstaged: upload again before anything else
    if (rc) return rc;
    auto t2 = std::chrono::steady_clock::now();
    rc = tsba_download(ctx, p); if (rc) return rc;
    auto t3 = std::chrono::steady_clock::now();
    r->t_upload_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    r->t_download_ms = std::chrono::duration<double, std::milli>(t3 - t2).count();
    return r->status;
}
int tsba_local_ba(void *ctx, tsba_problem *p, const tsba_options *o, tsba_report *r) { return one_shot(ctx, p, o, r); }
int tsba_pose_optim(void *ctx, tsba_problem *p, const tsba_options *o, tsba_report *r) {
    if (p && p->n_kf != 1) { if (ctx) set_err((Ctx *)ctx, "tsba_pose_optim needs n_kf == 1"); return TSBA_ERR_ARG; }
    return one_shot(ctx, p, o, r);
}
int tsba_global_ba(void *ctx, tsba_problem *p, const tsba_options *o, tsba_report *r) { return one_shot(ctx, p, o, r); }
int tsba_theta_optim(void *ctx, tsba_problem *p, const tsba_options *o, int text, double cov[9], tsba_report *r) {
    Ctx *c = (Ctx *)ctx;
    if (!c || !p || !o || !cov || !r || text < 0 || text >= p->n_text) return TSBA_ERR_ARG;
    c->cov_text = text;
    int rc = one_shot(ctx, p, o, r);
    c->cov_text = -1;
    if (rc) return rc;
    // Information matrix of theta[text] = V of the linearisation at the end of a pass (undamped, loss-corrected J^T J).  The reference
    // runs ceres::Covariance after EVERY pyramid pass and keeps the last one that succeeds (optimizer.cc:2219-2238: thetaVariance is only
    // overwritten when Compute returns true): the passes are tried from the last to the first.  A pass fails when V is not positive
    // definite or its reciprocal condition number is below 1e-14 (Ceres' min_reciprocal_condition_number; recalled, oracle/RECALLED.md).
    // No pass succeeds: as the reference (PyrThetaOptim still returns true, optimizer.cc:2224-2241) not an error, cov[] untouched.
    double Vall[6*TSBA_MAX_LEVELS];
    CK(hipMemcpy(Vall, c->cov_log, sizeof(Vall), hipMemcpyDeviceToHost));
    r->cov_valid = 0;
    for (int ps = std::min(o->n_passes, TSBA_MAX_LEVELS) - 1; ps >= 0; ps--) {
        const double *V = Vall + 6*ps;
        const double a = V[0], b = V[1], cc = V[2], e = V[3], f = V[4], i = V[5];
        const double A = e*i - f*f, B = -(b*i - cc*f), C = b*f - cc*e, det = a*A + b*B + cc*C;
        if (!(det > 0.0) || !(a > 0.0) || !(a*e - b*b > 0.0)) continue;
        // eigenvalues of the symmetric 3x3 (trigonometric form): reciprocal condition number
        const double q = (a + e + i)/3.0, p1 = b*b + cc*cc + f*f, p2 = (a - q)*(a - q) + (e - q)*(e - q) + (i - q)*(i - q) + 2.0*p1, pp = sqrt(p2/6.0);
        double lmin = q, lmax = q;
        if (pp > 0.0) { const double ip = 1.0/pp, b00 = (a - q)*ip, b01 = b*ip, b02 = cc*ip, b11 = (e - q)*ip, b12 = f*ip, b22 = (i - q)*ip;
            double hr = 0.5*(b00*(b11*b22 - b12*b12) - b01*(b01*b22 - b12*b02) + b02*(b01*b12 - b11*b02));
            hr = hr < -1.0 ? -1.0 : (hr > 1.0 ? 1.0 : hr);
            const double phi = acos(hr)/3.0; lmax = q + 2.0*pp*cos(phi); lmin = q + 2.0*pp*cos(phi + 2.0943951023931953); }
        if (!(lmin > 1e-14*lmax)) continue;
        const double id = 1.0/det;
        cov[0] = A*id; cov[1] = B*id; cov[2] = C*id; cov[3] = B*id; cov[4] = (a*i - cc*cc)*id; cov[5] = -(a*f - b*cc)*id;
        cov[6] = C*id; cov[7] = cov[5]; cov[8] = (a*e - b*b)*id;
        r->cov_valid = 1;
        break;
    }
    return TSBA_OK;
}

int tsba_eval(void *ctx, const tsba_problem *p, const tsba_options *o, int level,
              double *resid, double *jac, double *musigma, int64_t *ns_out, int64_t *nt_out) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    if (!p || level < 0 || level >= p->n_levels) return TSBA_ERR_ARG;
    tsba_options oo = *o; oo.n_passes = 1; oo.levels[0] = level;
    int rc = tsba_upload(ctx, p, &oo); if (rc) return rc;
    rc = reset_state(c); if (rc) return rc;
    const LevelDev &D = c->lev[level]; const HostPlan &H = c->hplan[level];
    launch_pass_init(c, D, 0);
    // reference block order: scene candidates by observation index, then text blocks by (tobs, feature)
    std::vector<int> out_idx(H.n_sc(), -1);
    { std::vector<int> by_obs(p->n_sobs[level], -1);
      for (int q = 0; q < H.n_sc(); q++) by_obs[H.sc_obs[q]] = q;
      int n = 0;
      for (int s = 0; s < p->n_sobs[level]; s++) { int q = by_obs[s]; if (q < 0) continue;
          if (oo.filter_good && !p->sgood[H.sc_flag[q]]) continue; out_idx[q] = n++; }
      *ns_out = n; }
    std::vector<int> bg, bf;
    for (int g = 0; g < H.n_tg(); g++) {
        int tb = H.tg_tobs[g], j = H.tg_text[g];
        if (oo.filter_good && !p->tobs_good[tb]) continue;
        for (int f = p->tfeat_off[level][j]; f < p->tfeat_off[level][j+1]; f++) {
            if (oo.filter_good && !p->tfgood[p->tobs_fgood_off[tb] + p->tfeat_raw[level][f]]) continue;
            bg.push_back(g); bf.push_back(f);
        }
    }
    *nt_out = (int64_t)bg.size();
    int64_t ns = *ns_out, nt = *nt_out;
    if (resid || jac) {
        const int *d_oi, *d_bg, *d_bf; double *d_r, *d_j = nullptr;
        rc = dev_upload_vec(c, &d_oi, out_idx); if (rc) return rc;
        rc = dev_upload_vec(c, &d_bg, bg); if (rc) return rc;
        rc = dev_upload_vec(c, &d_bf, bf); if (rc) return rc;
        rc = dev_alloc(c, &d_r, (size_t)(2*ns + 8*nt)); if (rc) return rc;
        if (jac) { rc = dev_alloc(c, &d_j, (size_t)(26*ns + 120*nt)); if (rc) return rc; }
        flush_run(c);                                  // staged uploads leave as one copy
        if (H.n_sc() > 0) LAUNCHK(k_eval_scene, dim3((H.n_sc() + 255)/256), dim3(256), 0, c->stream, c->W, D, d_oi, d_r, d_j);
        if (nt > 0) LAUNCHK(k_eval_text, dim3(((int)nt + 255)/256), dim3(256), 0, c->stream, c->W, D, (int)nt, d_bg, d_bf, (int)ns, d_r, d_j);
        CK(hipStreamSynchronize(c->stream)); CK(hipGetLastError());
        if (resid) CK(hipMemcpy(resid, d_r, sizeof(double)*(size_t)(2*ns + 8*nt), hipMemcpyDeviceToHost));
        if (jac) CK(hipMemcpy(jac, d_j, sizeof(double)*(size_t)(26*ns + 120*nt), hipMemcpyDeviceToHost));
    }
    CK(hipStreamSynchronize(c->stream));
    if (musigma && p->n_tobs) CK(hipMemcpy(musigma, c->W.musig, sizeof(double)*2*p->n_tobs, hipMemcpyDeviceToHost));
    return TSBA_OK;
}

#include "tsba_debug_abi.h"
} // extern "C"
static unsigned long long plan_checksum(const HostPlan &H) {           // over EVERY list of the plan
    unsigned long long h = 1469598103934665603ull;
    auto mix = [&](const std::vector<int32_t> &v) { for (int32_t x : v) { h ^= (unsigned int)x; h *= 1099511628211ull; } h ^= v.size(); h *= 1099511628211ull; };
    mix(H.sb_a); mix(H.sb_b); mix(H.sb_pt_off); mix(H.sb_pt_s1); mix(H.sb_pt_s2); mix(H.sb_pt_lm); mix(H.sb_tx_off); mix(H.sb_tx_s1); mix(H.sb_tx_s2); mix(H.sb_tx_lm);
    for (const std::vector<int32_t> *v : { &H.kf_order, &H.sc_obs, &H.sc_kf, &H.sc_pt, &H.sc_flag, &H.sc_slot, &H.pair_i, &H.pair_h, &H.pair_hpos, &H.pair_sc_off, &H.pair_tg_off, &H.pair_tg,
                                           &H.tg_tobs, &H.tg_kf, &H.tg_text, &H.tg_pair, &H.tg_slot, &H.pt_pose6, &H.pt_pair4, &H.tx_pair8, &H.tg_ppos, &H.pf_g, &H.pf_f, &H.tg_rec,
                                           &H.pls_off, &H.pslot_pose, &H.pslot_pair, &H.pslot_lm, &H.tls_off, &H.tslot_pose, &H.tslot_pair, &H.tslot_lm, &H.sb_pab, &H.sb_pba,
                                           &H.pose_t_off, &H.pose_t, &H.pose_h_off, &H.pose_h, &H.pose_ps_off, &H.pose_ps, &H.pose_ps_lm, &H.pose_ts_off, &H.pose_ts, &H.pose_ts_lm, &H.sb_rng }) mix(*v);
    if (H.far_B > 0) { h ^= (unsigned long long)H.far_B; h *= 1099511628211ull;
        for (const std::vector<int32_t> *v : { &H.far_a, &H.far_b, &H.far_off, &H.far_ent, &H.fb_id, &H.fb_pab, &H.fb_pba, &H.fb_pt_off, &H.fb_pt_s1, &H.fb_pt_s2, &H.fb_pt_lm, &H.fb_tx_off, &H.fb_tx_s1, &H.fb_tx_s2, &H.fb_tx_lm }) mix(*v); }
    for (double x : H.sc_uv) { unsigned long long u; memcpy(&u, &x, 8); h ^= u; h *= 1099511628211ull; }
    h ^= (unsigned long long)(H.bw_pose*4 + H.ring*2) + 8ull*(unsigned)H.ring_k0; h *= 1099511628211ull;
    return h;
}
extern "C" {
unsigned long long tsba_debug_plan_checksum(const tsba_problem *p, const tsba_options *o, int level, int threads) {
    if (!p || !o || level < 0 || level >= p->n_levels) return 0;
    const int saved = tsba_plan_threads; tsba_plan_threads = threads;
    HostPlan H; build_plan(p, o, level, H, false, true, tsba_plan_checksum_ring);
    tsba_plan_threads = saved;
    return plan_checksum(H);
}
// the plan of a single-frame problem (every landmark frozen) as build_plan_single_frame writes it down: the same checksum as tsba_debug_plan_checksum's; 0: not such a problem.
// recycled != 0: into a plan object that held the generic plan of the same problem's level 0 before
unsigned long long tsba_debug_plan_checksum_single_frame(const tsba_problem *p, const tsba_options *o, int level, int recycled) {
    if (!p || !o || level < 0 || level >= p->n_levels || !plan_is_single_frame(p, o)) return 0;
    HostPlan H; if (recycled) build_plan(p, o, 0, H);
    build_plan_single_frame(p, o, level, H);
    return plan_checksum(H);
}
// the plan of (p, o, level) built into a plan object that held the plan of (warm, ow, warm_level) before -- as a context does from call to call
unsigned long long tsba_debug_plan_checksum_recycled(const tsba_problem *warm, const tsba_options *ow, int warm_level, int warm_threads,
                                                     const tsba_problem *p, const tsba_options *o, int level, int threads) {
    if (!warm || !ow || !p || !o || level < 0 || level >= p->n_levels || warm_level < 0 || warm_level >= warm->n_levels) return 0;
    const int saved = tsba_plan_threads;
    HostPlan H;
    tsba_plan_threads = warm_threads; build_plan(warm, ow, warm_level, H, false, true, tsba_plan_checksum_ring);
    tsba_plan_threads = threads; build_plan(p, o, level, H, false, true, tsba_plan_checksum_ring);
    tsba_plan_threads = saved;
    return plan_checksum(H);
}
// host-only: the band + long-range split of the plan of `level` (tsba_plan.h: HostPlan::far_* / fb_*) when bands of up to far_max_blocks pose blocks are allowed.
// out[0] band of the preconditioner M in pose blocks (0: no split: ring, reordering or plain band), [1] 6x6 blocks of E (all ranks'), [2] those this rank
// contributes to, [3] the plan's band bound, [4] ring, [5] keyframes reordered, [6] checksum of the block positions (low 31 bits), [7] slot pairs of E, [8] keyframes touched by E when few enough for the low-rank correction (else 0).
// Checks what the split promises -- every block of M within the band, every slot pair of a landmark in exactly one of M / E, the positions of E
// sorted -- and returns TSBA_ERR_STATE if not.
int tsba_debug_plan_far(const tsba_problem *p, const tsba_options *o, int level, int far_max_blocks, int force, int ring_max_blocks, int32_t out[9]) {
    if (!p || !o || !out || level < 0 || level >= p->n_levels) return TSBA_ERR_ARG;
    HostPlan H; build_plan(p, o, level, H, false, true, ring_max_blocks, far_max_blocks, force != 0);
    int mine = 0; long long npairs = 0;
    if (H.far_B > 0) {
        for (int q = 0; q < H.n_sb(); q++) if (H.sb_b[q] - H.sb_a[q] > H.far_B) return TSBA_ERR_STATE;
        for (int q = 0; q < H.n_far(); q++) { if (H.far_a[q] >= H.far_b[q]) return TSBA_ERR_STATE;
            if (q > 0 && !(H.far_a[q-1] < H.far_a[q] || (H.far_a[q-1] == H.far_a[q] && H.far_b[q-1] < H.far_b[q]))) return TSBA_ERR_STATE;
            const bool has = H.fb_pt_off[q+1] > H.fb_pt_off[q] || H.fb_tx_off[q+1] > H.fb_tx_off[q] || H.fb_pab[q] >= 0 || H.fb_pba[q] >= 0; mine += has;
            for (int e = H.fb_pt_off[q]; e < H.fb_pt_off[q+1]; e++) if (H.pslot_pose[H.fb_pt_s1[e]] != H.far_a[q] || H.pslot_pose[H.fb_pt_s2[e]] != H.far_b[q] || H.pslot_lm[H.fb_pt_s1[e]] != H.pslot_lm[H.fb_pt_s2[e]]) return TSBA_ERR_STATE; }
        npairs = (long long)H.fb_pt_s1.size() + (long long)H.fb_tx_s1.size();
        // slot pairs with pose(s1) <= pose(s2): those of M + those of E = all of them
        long long all = 0; for (int j = 0; j < p->n_pt; j++) { const long long n = H.pls_off[j+1] - H.pls_off[j]; all += n*(n + 1)/2; }
        for (int j = 0; j < p->n_text; j++) { const long long n = H.tls_off[j+1] - H.tls_off[j]; all += n*(n + 1)/2; }
        if ((long long)H.sb_pt_s1.size() + (long long)H.sb_tx_s1.size() + npairs != all) return TSBA_ERR_STATE;
    }
    unsigned long long h = 1469598103934665603ull;
    for (const std::vector<int32_t> *v : { &H.far_a, &H.far_b, &H.far_off, &H.far_ent }) for (int32_t x : *v) { h ^= (unsigned int)x; h *= 1099511628211ull; }
    out[0] = H.far_B; out[1] = H.n_far(); out[2] = mine; out[3] = H.bw_pose; out[4] = H.ring; out[5] = H.kf_order.empty() ? 0 : 1; out[6] = (int32_t)(h & 0x7fffffffu);
    out[7] = (int32_t)npairs; out[8] = (int32_t)H.wb_kf.size();
    return TSBA_OK;
}
// host-only: does the plan of `level` take the ring path (one loop closure between the last and the first keyframes) when separators of up
// to ring_max_blocks pose blocks are allowed?  Returns 1 / 0 (< 0: error); *bw_pose = the band of the plan either way
int tsba_debug_plan_ring(const tsba_problem *p, const tsba_options *o, int level, int ring_max_blocks, int32_t *bw_pose) {
    if (!p || !o || !bw_pose || level < 0 || level >= p->n_levels) return TSBA_ERR_ARG;
    HostPlan H; build_plan(p, o, level, H, false, true, ring_max_blocks);
    *bw_pose = H.bw_pose;
    return H.ring ? 1 + 16*H.ring_k0 : 0;     // (the first keyframe of the loop in the upper bits: 0 = the whole trajectory)
}
void tsba_debug_bandp_part_ring(int nb, int B, int Pmax, int p, int *out5) { const BandpPart r = bandp_part_ring(nb - B, 0, B, Pmax, Pmax, p); out5[0] = r.P; out5[1] = r.a; out5[2] = r.b; out5[3] = r.has_left; out5[4] = r.has_right; }
// ring with a tail: nf free poses, the loop starts at free row row0; out8 = P, a, b, has_left, has_right, G, Pt, label of the left separator
void tsba_debug_bandp_part_ring2(int nf, int row0, int B, int Pmax, int Gmax, int p, int *out8) { const BandpPart r = bandp_part_ring(nf, row0, B, Pmax, Gmax, p);
    out8[0] = r.P; out8[1] = r.a; out8[2] = r.b; out8[3] = r.has_left; out8[4] = r.has_right; out8[5] = r.G; out8[6] = r.Pt; out8[7] = r.lblL; }
int tsba_debug_sv_lmax(int n_kf, int B, int Pmax) { return sv_lmax_of(n_kf, B, Pmax); }      // the bound the solve phase sizes its LDS with (tests/test_band_partition.py)
void tsba_debug_bandp_part(int nb, int B, int Pmax, int p, int *out5) { const BandpPart r = bandp_part(nb, B, Pmax, p); out5[0] = r.P; out5[1] = r.a; out5[2] = r.b; out5[3] = r.has_left; out5[4] = r.has_right; }
long long tsba_debug_cr_blk_index(int mmax, int br, int bc) { return (long long)cr_blk_index(mmax, br, bc); }
long long tsba_debug_cr_pool_blocks(int mmax) { return (long long)cr_pool_blocks(mmax); }
int tsba_debug_stamps(void *ctx, long long *out64) {
    Ctx *c = (Ctx *)ctx; if (!c || !c->uploaded) return TSBA_ERR_STATE;
    hipSetDevice(c->device); hipStreamSynchronize(c->stream);
    return hipMemcpy(out64, c->W.dbg, 64*sizeof(long long), hipMemcpyDeviceToHost) == hipSuccess ? 0 : TSBA_ERR_DEVICE;
}
#ifdef TSBA_SOLVE_STAMPS
int tsba_debug_step_stamps(void *ctx, long long *out128) {      // stamps build only: per factorisation step of the last k_solve_t launch
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    hipSetDevice(c->device); hipStreamSynchronize(c->stream);
    return hipMemcpyFromSymbol(out128, HIP_SYMBOL(ts_step_stamps), 128*sizeof(long long)) == hipSuccess ? 0 : TSBA_ERR_DEVICE;
}
#endif

// The library is loaded ONCE per process, under a lock -- and should be loaded (tsba_comm_load) before other host threads launch kernels: loading a HIP library registers its code
// objects with the runtime, and a kernel launch or hipFuncSetAttribute on another thread at that moment has been seen to fail ("invalid device function") and to crash (a segmentation
// fault inside another library's launch): the busy-context suite of round 6, test_multi_gpu_kernel_sequence_single_process beside an ORB extractor and a bundle adjustment looping on
// their own threads, once in about five runs (docs/ledger_r06.md 15.14).
static std::mutex g_rccl_mu; static void *g_rccl_so = nullptr; static std::string g_rccl_tried;
static void *rccl_handle() {
    std::lock_guard<std::mutex> lk(g_rccl_mu);
    if (g_rccl_so) return g_rccl_so;
    // RCCL must sit on the SAME HIP runtime as this library: streams and device pointers do not cross runtimes.  A process can hold
    // two (a PyTorch wheel bundles its own libamdhip64 + librccl next to /opt/rocm's, and which one this library is bound to
    // depends on the load order), and a bare dlopen("librccl.so.1") returns whichever copy was loaded first.  So: find the
    // runtime our own HIP calls resolve to and take the librccl next to it, by full path.
    g_rccl_tried.clear();
    Dl_info di;
    if (dladdr((void *)&hipStreamSynchronize, &di) && di.dli_fname) {
        std::string dir(di.dli_fname); const size_t sl = dir.rfind('/'); dir = sl == std::string::npos ? std::string(".") : dir.substr(0, sl);
        for (const char *n : { "/librccl.so.1", "/librccl.so" }) {
            const std::string path = dir + n; g_rccl_tried += path + " ";
            g_rccl_so = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL); if (g_rccl_so) break; }
    }
    if (!g_rccl_so) for (const char *n : { "librccl.so.1", "librccl.so" }) { g_rccl_tried += std::string(n) + " "; g_rccl_so = dlopen(n, RTLD_NOW | RTLD_LOCAL); if (g_rccl_so) break; }
    return g_rccl_so;
}
static int load_rccl(Ctx *c) {
    if (c->rccl_so) return 0;
    c->rccl_so = rccl_handle();
    if (!c->rccl_so) { const char *e = dlerror(); set_err(c, std::string("dlopen(librccl) failed, tried: ") + g_rccl_tried + ": " + (e ? e : "")); return TSBA_ERR_COMM; }
    c->p_getid = (decltype(c->p_getid))dlsym(c->rccl_so, "ncclGetUniqueId");
    c->p_init = (decltype(c->p_init))dlsym(c->rccl_so, "ncclCommInitRank");
    c->p_allreduce = (decltype(c->p_allreduce))dlsym(c->rccl_so, "ncclAllReduce");
    c->p_destroy = (decltype(c->p_destroy))dlsym(c->rccl_so, "ncclCommDestroy");
    c->p_errstr = (decltype(c->p_errstr))dlsym(c->rccl_so, "ncclGetErrorString");
    c->p_count = (decltype(c->p_count))dlsym(c->rccl_so, "ncclCommCount");
    if (!c->p_getid || !c->p_init || !c->p_allreduce || !c->p_destroy || !c->p_errstr) { set_err(c, "librccl: missing symbols"); return TSBA_ERR_COMM; }
    return 0;
}
int tsba_comm_load(void) { return rccl_handle() ? TSBA_OK : TSBA_ERR_COMM; }
int tsba_comm_unique_id(void *ctx, void *id128) {
    Ctx *c = (Ctx *)ctx; if (!c || !id128) return TSBA_ERR_ARG;
    int rc = load_rccl(c); if (rc) return rc;
    ncclUniqueId id;
    ncclResult_t r = c->p_getid(&id);
    if (r != ncclSuccess) { set_err(c, std::string("ncclGetUniqueId: ") + c->p_errstr(r)); return TSBA_ERR_COMM; }
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, 128);
    return TSBA_OK;
}
void *tsba_local_group_create(int world) {
    if (world < 1) return nullptr;
    LocalGroup *G = new LocalGroup(); G->world = world; G->stage.resize(world); return G;
}
void tsba_local_group_destroy(void *group) {
    LocalGroup *G = (LocalGroup *)group; if (!G) return;
    G->fail();
    for (Ctx *m : G->members) if (m && m->lgroup == G) { m->lgroup = nullptr; m->rank = 0; m->world = 1; m->uploaded = false; }   // (a resident problem was sharded for the group)
    delete G;
}
int tsba_comm_init_local(void *ctx, void *group, int rank, int world) {
    Ctx *c = (Ctx *)ctx; LocalGroup *G = (LocalGroup *)group;
    if (!c || !G || world != G->world || rank < 0 || rank >= world) return TSBA_ERR_ARG;
    hipSetDevice(c->device);
    free_problem(c);                               // any resident problem was sharded for the old world size
    c->lgroup = G; c->rank = rank; c->world = world;
    { std::lock_guard<std::mutex> lk(G->m); G->members.push_back(c); }
    return TSBA_OK;
}
int tsba_comm_stats(void *ctx, int32_t *ranks, int64_t bytes[3]) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    if (ranks) { int n = c->lgroup ? c->lgroup->world : 1;
        if (c->comm && c->p_count && c->p_count(c->comm, &n) != ncclSuccess) return TSBA_ERR_COMM;
        *ranks = n; }
    if (bytes) { bytes[0] = (int64_t)c->x_trial; bytes[1] = (int64_t)c->x_lin; bytes[2] = (int64_t)c->x_pass; }
    return TSBA_OK;
}
int tsba_debug_set(void *ctx, const tsba_debug_options *d) {
    Ctx *c = (Ctx *)ctx; if (!c) return TSBA_ERR_ARG;
    if (d) c->dbg = *d; else memset(&c->dbg, 0, sizeof(c->dbg));
    return TSBA_OK;
}
int tsba_comm_init(void *ctx, const void *id128, int rank, int world) {
    Ctx *c = (Ctx *)ctx; if (!c || world < 1 || rank < 0 || rank >= world) return TSBA_ERR_ARG;
    hipSetDevice(c->device);
    lgroup_forget(c);                              // (an RCCL communicator replaces an in-process group)
    if (!id128) { c->force_multi = world >= 1; c->rank = 0; c->world = 1; return TSBA_OK; }   // test hook: split kernels, no communicator
    int rc = load_rccl(c); if (rc) return rc;
    ncclUniqueId id; memcpy(&id, id128, 128);
    ncclResult_t r = c->p_init(&c->comm, world, id, rank);
    if (r != ncclSuccess) { set_err(c, std::string("ncclCommInitRank: ") + c->p_errstr(r)); return TSBA_ERR_COMM; }
    c->rank = rank; c->world = world;
    free_problem(c);                               // any resident problem was sharded for the old world size
    return TSBA_OK;
}

} // extern "C"
