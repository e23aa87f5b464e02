// TEST INFRASTRUCTURE: the C ABI of libtsba.so driven from C++ through the SAME gather / scatter templates a TextSLAM maintainer
// compiles against the real object graph (adapter/tsba_gather.hpp), here over plain structs of the same shape (mock_textslam.hpp).
//
//   abi_from_cxx <dump.bin> <mode> <out.bin>
//     1. read a flat problem (written by tests/test_cxx_adapter.py from textslam_amd.synth),
//     2. build the object graph from it: keyframes with vObvPts / vSceneObv2d[level] / vObvGoodPts / vFrameImg, map points with
//        their host keyframes, text planes with vRefFeature / vTextDeteRay / observations and their flags,
//     3. run the adapter's gather and compare EVERY flat array with the dump (identical, poses to 1e-15: q -> R -> q round trip),
//     4. if a HIP device is there: call the entry point of <mode> (local | global | landmarker | pose | init), scatter the result
//        back into the object graph exactly as optimizer.cc does, and write the graph's parameters and flags to <out.bin>.
//   exit code 0 = all of it, 3 = gather verified but no device (tsba_create returned TSBA_ERR_DEVICE), anything else = failure.
//
//   abi_from_cxx <dump.bin> time_local <out.json> [reps]
//     what optimizer::LocalBundleAdjustment costs its caller end to end (optimizer.cc:197-331): gather (pack_map over ALL map points / texts)
//     -> tsba_local_ba (plan + upload + solve + download) -> scatter_map, each phase timed; every repetition on a freshly built object graph
//     (the scatter moves the poses: a second call on the same graph would converge in fewer iterations).  One JSON line to <out.json>.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <map>
#include <vector>
#include "mock_textslam.hpp"
#include "dump_io.hpp"
#include "tsba_gather.hpp"

using namespace mock;
typedef tsba_adapter::Packed Packed;

static std::string L(const char *base, int l) { char b[64]; snprintf(b, sizeof b, "%s_%d", base, l); return b; }

// ---- the object graph of one problem
struct Graph {
    mock::map M;
    std::vector<keyframe *> kfs;          // the window / the map's keyframes, index = flat keyframe index
    std::vector<mapPts *> pts; std::vector<mapText *> texts;
    frame F;                              // pose mode
    int n_levels; double K[4];
};
static void set_T34(keyframe *kf, const double *T34, bool is_Twr) {      // outside host: T_rw (scene) or T_wr (text), row-major 3x4
    Mat44 Tm; Tm.setIdentity();
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) Tm(r, c) = T34[4*r + c];
    if (!is_Twr) { kf->SetPose(Tm); return; }
    kf->mTwc = Tm;                         // (only mTwc is read for a frozen text host, optimizer.cc:1524)
}
static void pose_to_frame(frame &fr, const double *pose) {
    double q[4] = { pose[0], pose[1], pose[2], pose[3] };
    Mat33 R; quat_to_R(q, R);
    Mat44 Tm; Tm.setIdentity();
    for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Tm(r, c) = R(r, c); Tm(r, 3) = pose[4 + r]; }
    fr.SetPose(Tm);
}
static void fill_images(frame &fr, const Dump &d, int n_levels, size_t k) {
    fr.vFrameImg.resize((size_t)n_levels);
    for (int l = 0; l < n_levels; l++) {
        const uint8_t *im = U8(d, L("img", l)); const int32_t *wh = I32(d, L("img_wh", l));
        if (!im || !wh) continue;
        const size_t npx = (size_t)wh[0]*wh[1];
        fr.vFrameImg[(size_t)l].cols = wh[0]; fr.vFrameImg[(size_t)l].rows = wh[1];
        fr.vFrameImg[(size_t)l].data.assign(im + k*npx, im + (k + 1)*npx);
    }
}
static mapText *make_text(const Dump &d, int n_levels, size_t j, keyframe *ref, const double *theta) {
    mapText *t = new mapText(); t->STATE = TEXTGOOD; t->mnId = j; t->RefKF = ref;
    Mat31 N; N(0) = theta[0]; N(1) = theta[1]; N(2) = theta[2];
    t->nidx = (int)ref->mNcr.size(); ref->mNcr.push_back(N);
    const double *box = F64(d, "text_box_ray") + 8*j;
    for (int b = 0; b < 4; b++) { Vec2 v; v(0) = box[2*b]; v(1) = box[2*b + 1]; t->vTextDeteRay.push_back(v); }
    t->vRefFeature.resize((size_t)n_levels);
    for (int l = 0; l < n_levels; l++) {
        const int32_t *off = I32(d, L("tfeat_off", l)); if (!off) continue;
        for (int f = off[j]; f < off[j + 1]; f++) {
            TextFeature *tf = new TextFeature(); tf->level = l; tf->IdxToRaw = I32(d, L("tfeat_raw", l))[f];
            tf->u = F64(d, L("tfeat_uv", l))[2*f]; tf->v = F64(d, L("tfeat_uv", l))[2*f + 1]; tf->feature(0) = tf->u; tf->feature(1) = tf->v;
            tf->neighbourNInten.assign(F64(d, L("tfeat_ref", l)) + 8*(size_t)f, F64(d, L("tfeat_ref", l)) + 8*(size_t)f + 8);
            t->vRefFeature[(size_t)l].push_back(tf);
        }
    }
    return t;
}
static void add_scene_obs(frame &fr, const Dump &d, int n_levels, int k, int flag_off, int n_raw, const std::vector<mapPts *> &pts) {
    fr.vObvPts.assign((size_t)n_raw, nullptr); fr.vObvGoodPts.assign((size_t)n_raw, true);
    fr.vSceneObv2d.resize((size_t)n_levels);
    for (int l = 0; l < n_levels; l++) {
        const int32_t *kf = I32(d, L("sobs_kf", l)), *pt = I32(d, L("sobs_pt", l)), *fl = I32(d, L("sobs_flag", l)); const double *uv = F64(d, L("sobs_uv0", l));
        const size_t n = CNT(d, L("sobs_kf", l));
        for (size_t s = 0; s < n; s++) { if (kf[s] != k) continue;
            const int raw = fl[s] - flag_off;
            SceneFeature *sf = new SceneFeature(); sf->level = l; sf->IdxToRaw = raw;
            const double sc = 1.0/(double)(1 << l);
            sf->u = uv[2*s]*sc; sf->v = uv[2*s + 1]*sc; sf->feature(0) = l == 0 ? uv[2*s] : sf->u; sf->feature(1) = l == 0 ? uv[2*s + 1] : sf->v;
            fr.vSceneObv2d[(size_t)l].push_back(sf);
            if (l == 0) { SceneObservation *so = new SceneObservation(); so->pt = pts[(size_t)pt[s]]; so->idx = raw; fr.vObvPts[(size_t)raw] = so; }
        }
    }
    const uint8_t *sg = U8(d, "sgood");
    for (int i = 0; i < n_raw; i++) fr.vObvGoodPts[(size_t)i] = sg[flag_off + i] != 0;
}
static void add_text_obs(frame &fr, const Dump &d, int k, const std::vector<mapText *> &texts) {
    const size_t n = CNT(d, "tobs_kf"); const int32_t *tk = I32(d, "tobs_kf"), *tt = I32(d, "tobs_text"), *fo = I32(d, "tobs_fgood_off");
    for (size_t t = 0; t < n; t++) { if (tk[t] != k) continue;
        TextObservation *o = new TextObservation(); o->obj = texts[(size_t)tt[t]]; o->cos = 1.0;
        fr.vObvText.push_back(o); fr.vObvGoodTexts.push_back(U8(d, "tobs_good")[t] != 0);
        std::vector<bool> fg; for (int f = fo[t]; f < fo[t + 1]; f++) fg.push_back(U8(d, "tfgood")[f] != 0);
        fr.vObvGoodTextFeats.push_back(fg); }
}

static void build_graph(const Dump &d, const std::string &mode, Graph &G) {
    G.n_levels = I32(d, "n_levels")[0]; for (int k = 0; k < 4; k++) G.K[k] = F64(d, "K")[k];
    const size_t n_kf = CNT(d, "pose")/7, n_pt = CNT(d, "rho"), n_text = CNT(d, "theta")/3;
    const double *pose = F64(d, "pose"), *rho = F64(d, "rho"), *theta = F64(d, "theta"), *ray = F64(d, "pt_ray");
    const int32_t *ph = I32(d, "pt_host"), *th = I32(d, "text_host"), *koff = I32(d, "kf_flag_off");
    long unsigned next_id = 2, outside_id = (long unsigned)n_kf + 2;
    int n_initial = 0;
    if (mode != "pose") for (size_t k = 0; k < n_kf; k++) {
        keyframe *kf = new keyframe();
        // the gauge marks of the flat problem come from mnId 0 / 1 (optimizer.cc:274-275)
        kf->mnId = (U8(d, "kf_initial")[k] && mode != "landmarker" && n_initial < 2) ? (long unsigned)n_initial++ : next_id++;
        pose_to_frame(*kf, pose + 7*k); fill_images(*kf, d, G.n_levels, k);
        G.kfs.push_back(kf);
    }
    auto outside = [&](const double *T34, bool twr) { keyframe *kf = new keyframe(); kf->mnId = outside_id++; set_T34(kf, T34, twr); return kf; };
    for (size_t j = 0; j < n_pt; j++) {
        mapPts *p = new mapPts(); p->mnId = j; p->rho = rho[j]; p->ray(0) = ray[2*j]; p->ray(1) = ray[2*j + 1]; p->ray(2) = 1.0;
        p->RefKF = (ph[j] >= 0 && mode != "pose") ? G.kfs[(size_t)ph[j]] : outside(F64(d, "pt_host_Trw") + 12*j, false);
        G.pts.push_back(p);
    }
    for (size_t j = 0; j < n_text; j++) {
        keyframe *ref = (th[j] >= 0 && mode != "pose") ? G.kfs[(size_t)th[j]] : outside(F64(d, "text_host_Twr") + 12*j, true);
        G.texts.push_back(make_text(d, G.n_levels, j, ref, theta + 3*j));
    }
    if (mode == "pose") {
        pose_to_frame(G.F, pose); fill_images(G.F, d, G.n_levels, 0);
        add_scene_obs(G.F, d, G.n_levels, 0, 0, (int)CNT(d, "sgood"), G.pts);
        add_text_obs(G.F, d, 0, G.texts);
        return;
    }
    for (size_t k = 0; k < n_kf; k++) {
        add_scene_obs(*G.kfs[k], d, G.n_levels, (int)k, koff[k], koff[k + 1] - koff[k], G.pts);
        add_text_obs(*G.kfs[k], d, (int)k, G.texts);
    }
    if (mode == "init") {                          // InitBA: F1 lists every landmark it hosts (vObvPts / vObvText), F2 holds the observations
        keyframe *F1 = G.kfs[0];
        for (size_t j = 0; j < n_pt; j++) { SceneObservation *so = new SceneObservation(); so->pt = G.pts[j]; so->idx = (int)j; F1->vObvPts.push_back(so); }
        for (size_t j = 0; j < n_text; j++) { TextObservation *o = new TextObservation(); o->obj = G.texts[j]; o->cos = 1.0; F1->vObvText.push_back(o); }
        F1->iNTextObj = (int)n_text;
    }
    for (size_t j = 0; j < n_text; j++) {          // mapText::vObvkeyframe (ThetaOptimMultiFs walks it)
        const size_t n = CNT(d, "tobs_kf");
        for (size_t t = 0; t < n; t++) if ((size_t)I32(d, "tobs_text")[t] == j) G.texts[j]->vObvkeyframe[G.kfs[(size_t)I32(d, "tobs_kf")[t]]] = std::vector<int>(1, 0);
    }
    G.M.vMapPoints = G.pts; G.M.vMapTextObjs = G.texts; G.M.vKeyframes = G.kfs;
    G.M.imapPts = (int)n_pt; G.M.imapText = (int)n_text; G.M.imapkfs = (int)outside_id;
}

// ---- comparison of a gathered problem with the dump
static int n_bad = 0;
template <class TT> static void same(const char *what, const TT *a, const TT *b, size_t n, double tol = 0.0) {
    if (n && (!a || !b)) { fprintf(stderr, "MISMATCH %s: missing array\n", what); n_bad++; return; }
    for (size_t i = 0; i < n; i++) if (std::fabs((double)a[i] - (double)b[i]) > tol) { fprintf(stderr, "MISMATCH %s[%zu]: %.17g vs %.17g\n", what, i, (double)a[i], (double)b[i]); n_bad++; return; }
}
static void size_is(const char *what, size_t a, size_t b) { if (a != b) { fprintf(stderr, "MISMATCH %s: %zu entries vs %zu\n", what, a, b); n_bad++; } }
static void compare(const Packed &P, const Dump &d, const std::string &mode) {
    const tsba_problem &p = P.p;
    size_is("n_kf", (size_t)p.n_kf, CNT(d, "pose")/7); size_is("n_pt", (size_t)p.n_pt, CNT(d, "rho")); size_is("n_text", (size_t)p.n_text, CNT(d, "theta")/3);
    if (n_bad) return;
    same("pose", p.pose, F64(d, "pose"), 7*(size_t)p.n_kf, 1e-15); same("rho", p.rho, F64(d, "rho"), (size_t)p.n_pt); same("theta", p.theta, F64(d, "theta"), 3*(size_t)p.n_text);
    same("kf_initial", p.kf_initial, U8(d, "kf_initial"), (size_t)p.n_kf);
    same("pt_ray", p.pt_ray, F64(d, "pt_ray"), 2*(size_t)p.n_pt); same("pt_host", p.pt_host, I32(d, "pt_host"), (size_t)p.n_pt);
    same("text_host", p.text_host, I32(d, "text_host"), (size_t)p.n_text); same("text_box_ray", p.text_box_ray, F64(d, "text_box_ray"), 8*(size_t)p.n_text);
    for (int j = 0; j < p.n_pt; j++) if (p.pt_host[j] < 0) same("pt_host_Trw", p.pt_host_Trw + 12*j, F64(d, "pt_host_Trw") + 12*j, 12);
    for (int j = 0; j < p.n_text; j++) if (p.text_host[j] < 0) same("text_host_Twr", p.text_host_Twr + 12*j, F64(d, "text_host_Twr") + 12*j, 12);
    size_is("n_sgood", (size_t)p.n_sgood, CNT(d, "sgood")); same("sgood", p.sgood, U8(d, "sgood"), CNT(d, "sgood"));
    for (int l = 0; l < p.n_levels; l++) {
        size_is(L("n_sobs", l).c_str(), (size_t)p.n_sobs[l], CNT(d, L("sobs_kf", l)));
        same(L("sobs_kf", l).c_str(), p.sobs_kf[l], I32(d, L("sobs_kf", l)), (size_t)p.n_sobs[l]); same(L("sobs_pt", l).c_str(), p.sobs_pt[l], I32(d, L("sobs_pt", l)), (size_t)p.n_sobs[l]);
        same(L("sobs_flag", l).c_str(), p.sobs_flag[l], I32(d, L("sobs_flag", l)), (size_t)p.n_sobs[l]); same(L("sobs_uv0", l).c_str(), p.sobs_uv0[l], F64(d, L("sobs_uv0", l)), 2*(size_t)p.n_sobs[l]);
        if (p.n_text == 0 || !I32(d, L("tfeat_off", l))) continue;
        size_is(L("n_tfeat", l).c_str(), (size_t)p.n_tfeat[l], CNT(d, L("tfeat_raw", l)));
        same(L("tfeat_off", l).c_str(), p.tfeat_off[l], I32(d, L("tfeat_off", l)), (size_t)p.n_text + 1); same(L("tfeat_raw", l).c_str(), p.tfeat_raw[l], I32(d, L("tfeat_raw", l)), (size_t)p.n_tfeat[l]);
        same(L("tfeat_uv", l).c_str(), p.tfeat_uv[l], F64(d, L("tfeat_uv", l)), 2*(size_t)p.n_tfeat[l]); same(L("tfeat_ref", l).c_str(), p.tfeat_ref[l], F64(d, L("tfeat_ref", l)), 8*(size_t)p.n_tfeat[l]);
        if (p.img[l] && I32(d, L("img_wh", l))) { const int32_t *wh = I32(d, L("img_wh", l)); size_is("img_w", (size_t)p.img_w[l], (size_t)wh[0]); size_is("img_h", (size_t)p.img_h[l], (size_t)wh[1]);
            const size_t npx = (size_t)wh[0]*wh[1];
            for (int k = 0; k < p.n_kf; k++) if (memcmp(p.img[l][k], U8(d, L("img", l)) + (size_t)k*npx, npx)) { fprintf(stderr, "MISMATCH image level %d keyframe %d\n", l, k); n_bad++; break; } }
    }
    size_is("n_tobs", (size_t)p.n_tobs, CNT(d, "tobs_kf"));
    same("tobs_kf", p.tobs_kf, I32(d, "tobs_kf"), (size_t)p.n_tobs); same("tobs_text", p.tobs_text, I32(d, "tobs_text"), (size_t)p.n_tobs);
    same("tobs_good", p.tobs_good, U8(d, "tobs_good"), (size_t)p.n_tobs); same("tobs_fgood_off", p.tobs_fgood_off, I32(d, "tobs_fgood_off"), (size_t)p.n_tobs + 1);
    size_is("tfgood", P.tfgood.size(), CNT(d, "tfgood"));
    if (!n_bad) same("tfgood", p.tfgood, U8(d, "tfgood"), CNT(d, "tfgood"));
    (void)mode;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s dump.bin local|global|landmarker|pose|init out.bin\n", argv[0]); return 2; }
    Dump d; if (!read_dump(argv[1], d)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    const std::string mode = argv[2];
    if (mode == "time_local") {
        const int reps = argc > 4 ? atoi(argv[4]) : 5;
        void *cx = nullptr; int r0 = tsba_create(&cx, 0);
        if (r0 == TSBA_ERR_DEVICE) { printf("no HIP device\n"); return 3; }
        if (r0) return 1;
        const bool wt = CNT(d, "tobs_kf") > 0;
        std::vector<double> tg, ts, tc, tt; int its[TSBA_MAX_LEVELS] = {0}; double sol = 0, upl = 0, dwn = 0;
        tsba_adapter::GatherCache cache;                                      // as the adapter's: what the 19 keyframes that stay contributed to the last call
        const bool use_cache = !(argc > 5 && std::string(argv[5]) == "nocache");
        Packed Pkeep;                                                        // (the adapter keeps one per thread)
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        for (int rep_i = 0; rep_i < reps + 1; rep_i++) {                    // (the first call of a fresh context allocates: not counted)
            Graph G; build_graph(d, "local", G);
            G.kfs.back()->mnId = 100000 + (long unsigned)rep_i; G.M.imapkfs = 100000 + reps + 8;     // the window's newest keyframe is new to the context in every call: its planes cross the bus, the other 19 keyframes' do not
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<mapPts *> vP = G.M.GetAllMapPoints(); std::vector<mapText *> vT = G.M.GetAllMapTexts(TEXTGOOD);
            Packed Pnew; Packed &P = use_cache ? Pkeep : Pnew; if (use_cache) P.reset();
            tsba_adapter::pack_map<Traits>(&G.M, G.kfs, vP, vT, 0, G.n_levels, G.K, wt, P, nullptr, nullptr, use_cache ? &cache : nullptr);
            const auto t1 = std::chrono::steady_clock::now();
            tsba_options o; tsba_report rp; tsba_default_options_local(&o); o.state = I32(d, "state") ? I32(d, "state")[0] : TSBA_STATE_LOCAL;
            const int rc = tsba_local_ba(cx, &P.p, &o, &rp);
            const auto t2 = std::chrono::steady_clock::now();
            if (rc) { fprintf(stderr, "tsba_local_ba: %d (%s)\n", rc, tsba_last_error(cx)); return 1; }
            tsba_adapter::scatter_map<Traits>(P, G.kfs, vP, vT, true, true);
            const auto t3 = std::chrono::steady_clock::now();
            if (rep_i == 0) continue;
            tg.push_back(ms(t0, t1)); tc.push_back(ms(t1, t2)); ts.push_back(ms(t2, t3)); tt.push_back(ms(t0, t3));
            for (int k = 0; k < rp.n_passes; k++) its[k] = rp.iters[k];
            sol = rp.t_solve_ms; upl = rp.t_upload_ms; dwn = rp.t_download_ms;
        }
        auto mn = [](std::vector<double> &v) { double m = v[0]; for (double x : v) m = x < m ? x : m; return m; };
        FILE *f = fopen(argv[3], "w"); if (!f) return 2;
        fprintf(f, "{\"local_ba_adapter_call_ms\": %.4f, \"gather_ms\": %.4f, \"tsba_local_ba_ms\": %.4f, \"scatter_ms\": %.4f, \"of_which_upload_plan_ms\": %.4f, \"solve_ms\": %.4f, \"download_ms\": %.4f, "
                   "\"reps\": %d, \"lm_iterations\": [%d, %d, %d], \"keyframes\": %zu, \"map_points\": %zu, \"text_planes\": %zu, \"gather_cache\": %s, \"gather_cache_hits\": %lld, \"gather_cache_misses\": %lld}\n",
                mn(tt), mn(tg), mn(tc), mn(ts), upl, sol, dwn, reps, its[0], its[1], its[2], CNT(d, "pose")/7, CNT(d, "rho"), CNT(d, "theta")/3, use_cache ? "true" : "false", cache.hits, cache.misses);
        fclose(f); tsba_destroy(cx);
        printf("adapter call: %.3f ms (gather %.3f, tsba_local_ba %.3f, scatter %.3f)\n", mn(tt), mn(tg), mn(tc), mn(ts));
        return 0;
    }
    if (mode == "time_gather") {            // CPU only: what pack_map costs with and without the segment cache (same graph, keyframe ids as a sliding window would have them)
        Graph G; build_graph(d, "local", G);
        const bool wt = CNT(d, "tobs_kf") > 0;
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        tsba_adapter::GatherCache cache; double best[3] = {1e30, 1e30, 1e30}; Packed Pkeep;
        for (int rep_i = 0; rep_i < 30; rep_i++) for (int cc = 0; cc < 3; cc++) {
            const auto t0 = std::chrono::steady_clock::now();
            std::vector<mapPts *> vP = G.M.GetAllMapPoints(); std::vector<mapText *> vT = G.M.GetAllMapTexts(TEXTGOOD);
            Packed Pnew; Packed &P = cc == 2 ? Pkeep : Pnew; if (cc == 2) P.reset();
            tsba_adapter::pack_map<Traits>(&G.M, G.kfs, vP, vT, 0, G.n_levels, G.K, wt, P, nullptr, nullptr, cc ? &cache : nullptr);
            const double t = ms(t0, std::chrono::steady_clock::now());
            if (t < best[cc]) best[cc] = t;
        }
        printf("gather: %.3f ms from scratch, %.3f ms with the segment cache, %.3f ms with the cache into a Packed kept between calls\n", best[0], best[1], best[2]);
        return 0;
    }
    if (mode == "slide_check") {
        // A window that slides: sub-windows of W keyframes of the dump's map, moved on by one keyframe per call (tracking.cc:826-842), gathered (a) from
        // scratch and (b) with ONE GatherCache kept over the calls -- every flat array of (b) has to equal (a)'s, element for element; then the same after
        // the graph changed in the ways the reference allows between two calls: a keyframe gained an observation (keyframe::AddSceneObserv, the lists grow),
        // flags flipped, parameters moved, a plane left the TEXTGOOD set.  CPU only.
        Graph G; build_graph(d, "local", G);
        const bool wt = CNT(d, "tobs_kf") > 0;
        const size_t nk = G.kfs.size(), Wn = nk > 6 ? nk - 5 : nk;
        tsba_adapter::GatherCache cache; int calls = 0;
        auto same = [&](const Packed &A, const Packed &B) {
            bool ok = A.pose == B.pose && A.rho == B.rho && A.theta == B.theta && A.pt_ray == B.pt_ray && A.pt_Trw == B.pt_Trw && A.text_Twr == B.text_Twr && A.text_box == B.text_box
                && A.pt_host == B.pt_host && A.text_host == B.text_host && A.tobs_kf == B.tobs_kf && A.tobs_text == B.tobs_text && A.tobs_fgood_off == B.tobs_fgood_off
                && A.kf_initial == B.kf_initial && A.sgood == B.sgood && A.tobs_good == B.tobs_good && A.tfgood == B.tfgood && A.kf_id == B.kf_id && A.kf_flag_off == B.kf_flag_off && A.tobs_raw == B.tobs_raw;
            for (int l = 0; l < TSBA_MAX_LEVELS; l++) ok = ok && A.sobs_uv0[l] == B.sobs_uv0[l] && A.tfeat_uv[l] == B.tfeat_uv[l] && A.tfeat_ref[l] == B.tfeat_ref[l] && A.sobs_kf[l] == B.sobs_kf[l]
                && A.sobs_pt[l] == B.sobs_pt[l] && A.sobs_flag[l] == B.sobs_flag[l] && A.tfeat_off[l] == B.tfeat_off[l] && A.tfeat_raw[l] == B.tfeat_raw[l] && A.img[l] == B.img[l];
            return ok && A.p.n_kf == B.p.n_kf && A.p.n_pt == B.p.n_pt && A.p.n_text == B.p.n_text && A.p.n_tobs == B.p.n_tobs && A.p.n_sgood == B.p.n_sgood;
        };
        auto check = [&](size_t w0, const char *what) {
            std::vector<keyframe *> win(G.kfs.begin() + (long)w0, G.kfs.begin() + (long)(w0 + Wn));
            std::vector<mapPts *> vP = G.M.GetAllMapPoints(); std::vector<mapText *> vT = G.M.GetAllMapTexts(TEXTGOOD);
            Packed A, B;
            tsba_adapter::pack_map<Traits>(&G.M, win, vP, vT, 0, G.n_levels, G.K, wt, A);
            tsba_adapter::pack_map<Traits>(&G.M, win, vP, vT, 0, G.n_levels, G.K, wt, B, nullptr, nullptr, &cache);
            calls++;
            if (!same(A, B)) { fprintf(stderr, "slide_check: cached gather differs from a fresh one (%s, window at %zu)\n", what, w0); return false; }
            return true;
        };
        for (size_t w0 = 0; w0 + Wn <= nk; w0++) if (!check(w0, "slide")) return 1;
        const long long hits_slide = cache.hits, miss_slide = cache.misses;
        // the graph moves on between calls
        keyframe *kf = G.kfs[nk - 2];
        if (!G.pts.empty()) {                                                   // keyframe::AddSceneObserv (keyframe.cc:96-114): one more entry of vObvPts / vObvGoodPts and one feature per pyramid level
            kf->vObvPts.push_back(new SceneObservation{G.pts[0], 0}); kf->vObvGoodPts.push_back(true);
            const int raw = (int)kf->vObvPts.size() - 1;
            for (size_t l = 0; l < kf->vSceneObv2d.size(); l++) { const double sc = 1.0/(double)(1 << l); Vec2 f; f(0) = 101.5*sc; f(1) = 77.25*sc;
                kf->vSceneObv2d[l].push_back(new SceneFeature{f(0), f(1), f, (int)l, raw}); } }
        for (size_t i = 0; i < kf->vObvGoodPts.size(); i += 7) kf->vObvGoodPts[i] = !kf->vObvGoodPts[i];
        for (size_t j = 0; j < G.pts.size(); j += 3) { double r = G.pts[j]->GetInverD()*1.01; G.pts[j]->SetRho(r); }
        if (G.texts.size() > 1) G.texts[1]->STATE = TEXTIMMATURE;
        if (!check(nk - Wn, "after the graph changed")) return 1;
        if (!check(nk - Wn, "the same window again")) return 1;
        cache.invalidate();
        if (!check(nk - Wn, "after invalidate()")) return 1;
        printf("slide_check: %d gathers identical with and without the cache; while sliding %lld segments reused, %lld built\n", calls, hits_slide, miss_slide);
        if (!(hits_slide > miss_slide)) { fprintf(stderr, "slide_check: the cache was not used\n"); return 1; }
        return 0;
    }
    Graph G; build_graph(d, mode, G);
    const bool with_text = CNT(d, "tobs_kf") > 0;
    Packed P;
    std::vector<mapPts *> vPts = G.M.GetAllMapPoints(); std::vector<mapText *> vTexts = G.M.GetAllMapTexts(TEXTGOOD);
    if (mode == "local") tsba_adapter::pack_map<Traits>(&G.M, G.kfs, vPts, vTexts, 0, G.n_levels, G.K, with_text, P);
    else if (mode == "global") tsba_adapter::pack_map<Traits>(&G.M, G.M.GetAllKeyFrame(), vPts, vTexts, 1, G.n_levels, G.K, with_text, P);
    else if (mode == "landmarker") tsba_adapter::pack_map<Traits>(&G.M, G.M.GetAllKeyFrame(), vPts, vTexts, 2, G.n_levels, G.K, with_text, P);
    else if (mode == "pose") tsba_adapter::pack_pose<Traits>(G.F, G.n_levels, G.K, with_text, P);
    else if (mode == "init") tsba_adapter::pack_init<Traits>(*G.kfs[0], *G.kfs[1], G.n_levels, G.K, P);
    else if (mode == "theta") {                    // ThetaOptimMultiFs(F, obj): plane 0, the last keyframe standing in for the current frame
        frame Fcur = *G.kfs.back();
        tsba_adapter::pack_theta<Traits>(Fcur, *G.texts[0], G.n_levels, G.K, P);
        if (P.p.n_text != 1 || P.p.n_tobs != P.p.n_kf - 1 || P.p.text_host[0] != 0) { fprintf(stderr, "pack_theta: unexpected shape\n"); return 1; }
        printf("gather identical: theta problem of %d frames\n", P.p.n_kf);
        void *cx = nullptr; int r0 = tsba_create(&cx, 0);
        if (r0 == TSBA_ERR_DEVICE) { printf("no HIP device: gather verified only\n"); return 3; }
        tsba_options ot; tsba_report rt; double cov[9] = {0};
        tsba_default_options_theta(&ot);
        r0 = tsba_theta_optim(cx, &P.p, &ot, 0, cov, &rt);
        if (r0) { fprintf(stderr, "tsba_theta_optim: %d (%s)\n", r0, tsba_last_error(cx)); return 1; }
        Traits::set_theta(*G.texts[0], P.p.theta);                       // obj->RefKF->SetN(thetaNew, ...), obj->Covariance = cov (optimizer.cc:619-621)
        FILE *ft = fopen(argv[3], "wb"); if (!ft) return 2;
        int32_t cv = rt.cov_valid; put(ft, "theta", 0, P.p.theta, 3); put(ft, "cov", 0, cov, 9); put(ft, "cov_valid", 1, &cv, 1); fclose(ft);
        tsba_destroy(cx);
        printf("solve + scatter done: theta\n");
        return 0;
    }
    else { fprintf(stderr, "unknown mode %s\n", mode.c_str()); return 2; }
    compare(P, d, mode);
    if (n_bad) { fprintf(stderr, "gather differs from the flat problem in %d array(s)\n", n_bad); return 1; }
    printf("gather identical: %d keyframes, %d points, %d planes, %d text observations, %d levels\n", P.p.n_kf, P.p.n_pt, P.p.n_text, P.p.n_tobs, P.p.n_levels);

    void *ctx = nullptr;
    int rc = tsba_create(&ctx, 0);
    if (rc == TSBA_ERR_DEVICE) { printf("no HIP device: gather verified only\n"); return 3; }
    if (rc) { fprintf(stderr, "tsba_create: %d\n", rc); return 1; }
    tsba_options o; tsba_report rep;
    const int state = I32(d, "state") ? I32(d, "state")[0] : TSBA_STATE_LOCAL;
    if (mode == "local") { tsba_default_options_local(&o); o.state = state; rc = tsba_local_ba(ctx, &P.p, &o, &rep); }
    else if (mode == "global") { tsba_default_options_global(&o); o.use_text = with_text; rc = tsba_global_ba(ctx, &P.p, &o, &rep); }
    else if (mode == "landmarker") { tsba_default_options_landmarker(&o); rc = tsba_local_ba(ctx, &P.p, &o, &rep); }
    else if (mode == "pose") { tsba_default_options_pose(&o); rc = tsba_pose_optim(ctx, &P.p, &o, &rep); }
    else { tsba_default_options_init(&o); rc = tsba_local_ba(ctx, &P.p, &o, &rep); }
    if (rc) { fprintf(stderr, "solve failed: %d (%s)\n", rc, tsba_last_error(ctx)); return 1; }
    // scatter into the object graph as optimizer.cc does, then read the graph back for the comparison on the Python side
    if (mode == "pose") tsba_adapter::scatter_pose<Traits>(P, G.F);
    else if (mode == "init") tsba_adapter::scatter_init<Traits>(P, *G.kfs[0], *G.kfs[1]);
    else tsba_adapter::scatter_map<Traits>(P, G.kfs, vPts, vTexts, mode != "landmarker", mode != "global");
    std::vector<double> pose, rho, theta; std::vector<uint8_t> sg, tg, tf;
    if (mode == "pose") tsba_adapter::push_pose<Traits>(G.F, pose); else for (size_t k = 0; k < G.kfs.size(); k++) tsba_adapter::push_pose<Traits>(*G.kfs[k], pose);
    for (size_t j = 0; j < G.pts.size(); j++) rho.push_back(G.pts[j]->GetInverD());
    for (size_t j = 0; j < G.texts.size(); j++) { Mat31 N = G.texts[j]->RefKF->mNcr[(size_t)G.texts[j]->GetNidx()]; theta.push_back(N(0)); theta.push_back(N(1)); theta.push_back(N(2)); }
    std::vector<frame *> frames; if (mode == "pose") frames.push_back(&G.F); else for (size_t k = 0; k < G.kfs.size(); k++) frames.push_back(G.kfs[k]);
    for (size_t k = 0; k < frames.size(); k++) {
        for (size_t i = 0; i < frames[k]->vObvGoodPts.size(); i++) sg.push_back(frames[k]->vObvGoodPts[i]);
        for (size_t i = 0; i < frames[k]->vObvGoodTexts.size(); i++) { tg.push_back(frames[k]->vObvGoodTexts[i]); for (size_t f = 0; f < frames[k]->vObvGoodTextFeats[i].size(); f++) tf.push_back(frames[k]->vObvGoodTextFeats[i][f]); }
    }
    FILE *f = fopen(argv[3], "wb"); if (!f) return 2;
    put(f, "pose", 0, pose.data(), pose.size()); put(f, "rho", 0, rho.data(), rho.size()); put(f, "theta", 0, theta.data(), theta.size());
    put(f, "sgood", 2, sg.data(), sg.size()); put(f, "tobs_good", 2, tg.data(), tg.size()); put(f, "tfgood", 2, tf.data(), tf.size());
    int32_t it[TSBA_MAX_LEVELS]; for (int k = 0; k < TSBA_MAX_LEVELS; k++) it[k] = rep.iters[k];
    put(f, "iters", 1, it, (size_t)rep.n_passes); put(f, "cost1", 0, rep.cost1, (size_t)rep.n_passes);
    fclose(f);
    tsba_destroy(ctx);
    printf("solve + scatter done: passes %d, iterations", rep.n_passes); for (int k = 0; k < rep.n_passes; k++) printf(" %d", rep.iters[k]); printf("\n");
    return 0;
}
