// TEST INFRASTRUCTURE: the C ABI of libtsloop.so driven from C++ through the gather / scatter templates a TextSLAM maintainer compiles
// (adapter/tsloop_gather.hpp), over plain structs with the shape of the reference's loop-closing data (mock_textslam.hpp).
//
//   loop_from_cxx <dump.bin> sim3|loop <out.bin>
//     sim3: vFeat1 / vFeat2 (FeatureConvert: posObv, obv2d.pt), vbInliers, Sim12 from the dump -> pack_sim3 must reproduce the flat arrays
//           (P1, P2, uv, inliers identical; Sim12 with its quaternion normalised) -> with a HIP device: tsloop_optimize_sim3, scatter_sim3.
//     loop: keyframes with mTcw from the drifted estimate, NormConnections / LoopConnections (std::map<keyframe *, std::set<keyframe *>>),
//           vConnectKFs (corrected Sim3 of the current keyframe's neighbourhood), mScw, KF, LoopKF, a few map points and text planes ->
//           pack_loop: the gathered arrays are written out (the Python side compares them with the flat problem of textslam_amd.synth
//           connection by connection) -> with a HIP device: tsloop_optimize_loop, scatter_loop (SetPose([R | t/s]), rho *= s, theta *= s),
//           and the object graph is written out.
//   exit code 0 = all of it, 3 = gather done but no device (tsloop_create returned TSLOOP_ERR_DEVICE), anything else = failure.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <map>
#include <set>
#include <string>
#include <vector>
#include "mock_textslam.hpp"
#include "dump_io.hpp"
#include "tsloop_gather.hpp"

using namespace mock;

static int n_bad = 0;
static void same(const char *what, const void *a, const void *b, size_t bytes) { if (bytes && memcmp(a, b, bytes) != 0) { fprintf(stderr, "gather: %s differs\n", what); n_bad++; } }
static void close_to(const char *what, const double *a, const double *b, size_t n, double tol) {
    for (size_t i = 0; i < n; i++) if (!(std::fabs(a[i] - b[i]) <= tol)) { fprintf(stderr, "gather: %s[%zu] = %.17g, expected %.17g\n", what, i, a[i], b[i]); n_bad++; return; } }

static int run_sim3(const Dump &d, const char *out_path) {
    const size_t n = CNT(d, "inliers");
    const double *P1 = F64(d, "P1"), *P2 = F64(d, "P2"), *uv1 = F64(d, "uv1"), *uv2 = F64(d, "uv2"), *sim0 = F64(d, "sim0"), *K = F64(d, "K");
    const uint8_t *inl = U8(d, "inliers");
    if (!sim0 || !K || (n && (!P1 || !P2 || !uv1 || !uv2 || !inl))) { fprintf(stderr, "sim3 dump incomplete\n"); return 2; }
    std::vector<FeatureConvert> vFeat1(n), vFeat2(n); std::vector<bool> vbInliers(n);
    for (size_t i = 0; i < n; i++) {
        FeatureConvert a; memset(&a, 0, sizeof a); FeatureConvert b = a;
        for (int c = 0; c < 3; c++) { a.posObv(c) = P1[3*i + c]; b.posObv(c) = P2[3*i + c]; }
        a.obv2d.pt.x = (float)uv1[2*i]; a.obv2d.pt.y = (float)uv1[2*i + 1]; b.obv2d.pt.x = (float)uv2[2*i]; b.obv2d.pt.y = (float)uv2[2*i + 1];
        vFeat1[i] = a; vFeat2[i] = b; vbInliers[i] = inl[i] != 0;
    }
    Sim3_loop Sim12; for (int a = 0; a < 4; a++) Sim12.r[a] = sim0[a]; for (int a = 0; a < 3; a++) Sim12.t(a) = sim0[4 + a]; Sim12.s = sim0[7];   // (r as handed over: not normalised)
    tsloop_adapter::PackedSim3 P;
    tsloop_adapter::pack_sim3<Traits>(vFeat1, vFeat2, vbInliers, Sim12, K, P);
    if ((size_t)P.p.n != n) { fprintf(stderr, "gather: n\n"); return 1; }
    same("P1", P.p.P1, P1, 24*n); same("P2", P.p.P2, P2, 24*n); same("inliers", P.p.inlier, inl, n); same("K", P.p.K, K, 32);
    for (size_t i = 0; i < 2*n; i++) if (P.p.uv1[i] != (float)uv1[i] || P.p.uv2[i] != (float)uv2[i]) { fprintf(stderr, "gather: uv[%zu]\n", i); n_bad++; break; }
    double qn = 0; for (int a = 0; a < 4; a++) qn += sim0[a]*sim0[a]; qn = std::sqrt(qn);
    double expect[8]; for (int a = 0; a < 4; a++) expect[a] = sim0[a]/qn; for (int a = 4; a < 8; a++) expect[a] = sim0[a];
    close_to("sim", P.p.sim, expect, 8, 1e-15);
    if (n_bad) return 1;
    printf("gather identical: sim3 problem of %zu matches\n", n);

    void *ctx = nullptr; int rc = tsloop_create(0, &ctx);
    if (rc == TSLOOP_ERR_DEVICE) { printf("no HIP device: gather verified only\n"); return 3; }
    if (rc) { fprintf(stderr, "tsloop_create: %d\n", rc); return 1; }
    tsloop_options o; tsloop_default_options_sim3(&o); tsloop_report rep;
    rc = tsloop_optimize_sim3(ctx, &P.p, &o, &rep);
    if (rc != TSLOOP_OK && rc != TSLOOP_ERR_NUMERIC) { fprintf(stderr, "tsloop_optimize_sim3: %d (%s)\n", rc, tsloop_last_error(ctx)); return 1; }
    const int n_in = tsloop_adapter::scatter_sim3<Traits>(P, rep, vbInliers, Sim12);
    double sim[8]; Traits::sim_get(Sim12, false, sim);
    std::vector<uint8_t> vb(n); for (size_t i = 0; i < n; i++) vb[i] = vbInliers[i] ? 1 : 0;
    int32_t meta[3] = { n_in, rep.iters, rep.termination };
    FILE *f = fopen(out_path, "wb"); if (!f) return 2;
    put(f, "sim", 0, sim, 8); put(f, "inliers", 2, vb.data(), n); put(f, "meta", 1, meta, 3); put(f, "cost1", 0, &rep.cost1, 1);
    fclose(f); tsloop_destroy(ctx);
    printf("solve + scatter done: sim3, %d inliers\n", n_in);
    return 0;
}

static int run_loop(const Dump &d, const char *out_path) {
    const size_t n = CNT(d, "est")/8;
    const double *est = F64(d, "est"), *conn_sim = F64(d, "conn_sim"), *mScw8 = F64(d, "mScw"), *rho0 = F64(d, "rho"), *theta0 = F64(d, "theta");
    const int32_t *conn_idx = I32(d, "conn_idx"), *ni = I32(d, "norm_i"), *nj = I32(d, "norm_j"), *li = I32(d, "loop_i"), *lj = I32(d, "loop_j"), *ids = I32(d, "ids");
    const int32_t *pt_host = I32(d, "pt_host"), *text_host = I32(d, "text_host");
    if (!est || !ids || n < 3) { fprintf(stderr, "loop dump incomplete\n"); return 2; }
    const int kf_cur = ids[0], kf_loop = ids[1];
    // keyframes in ONE array: pointer order = keyframe order, so the std::map / std::set iteration of the gather is reproducible
    std::vector<keyframe> store(n); std::vector<keyframe *> vKFs(n);
    for (size_t k = 0; k < n; k++) {
        keyframe &kf = store[k]; kf.mnId = (long unsigned int)k; vKFs[k] = &kf;
        Mat33 R; quat_to_R(est + 8*k, R); Mat44 Tm; Tm.setIdentity();
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Tm(r, c) = R(r, c); Tm(r, 3) = est[8*k + 4 + r]; }
        kf.SetPose(Tm);
    }
    std::map<keyframe *, std::set<keyframe *> > NormConnections, LoopConnections;
    for (size_t e = 0; e < CNT(d, "norm_i"); e++) NormConnections[&store[(size_t)ni[e]]].insert(&store[(size_t)nj[e]]);
    for (size_t e = 0; e < CNT(d, "loop_i"); e++) LoopConnections[&store[(size_t)lj[e]]].insert(&store[(size_t)li[e]]);     // keyed by KFj (the current side), :826
    std::map<keyframe *, Sim3_loop> vConnectKFs;
    for (size_t c = 0; c < CNT(d, "conn_idx"); c++) vConnectKFs[&store[(size_t)conn_idx[c]]] = Traits::sim_make(conn_sim + 8*c, conn_sim + 8*c + 4, conn_sim[8*c + 7]);
    const Sim3_loop mScw = Traits::sim_make(mScw8, mScw8 + 4, mScw8[7]);
    // a few landmarks for the map update
    const size_t n_pt = CNT(d, "pt_host"), n_tx = CNT(d, "text_host");
    std::vector<mapPts> pts(n_pt); std::vector<mapText> txs(n_tx); std::vector<mapPts *> vPts(n_pt); std::vector<mapText *> vObjs(n_tx);
    for (size_t j = 0; j < n_pt; j++) { pts[j].mnId = (long unsigned int)j; pts[j].RefKF = &store[(size_t)pt_host[j]]; pts[j].rho = rho0[j]; pts[j].ray(0) = 0.1; pts[j].ray(1) = -0.2; pts[j].ray(2) = 1.0; vPts[j] = &pts[j]; }
    for (size_t j = 0; j < n_tx; j++) { keyframe &h = store[(size_t)text_host[j]]; txs[j].STATE = TEXTGOOD; txs[j].mnId = (long unsigned int)j; txs[j].RefKF = &h; txs[j].nidx = (int)h.mNcr.size();
        Mat31 N; for (int a = 0; a < 3; a++) N(a) = theta0[3*j + a]; h.mNcr.push_back(N); vObjs[j] = &txs[j]; }

    tsloop_adapter::PackedLoop P;
    if (!tsloop_adapter::pack_loop<Traits>(vKFs, LoopConnections, NormConnections, &store[(size_t)kf_cur], &store[(size_t)kf_loop], vConnectKFs, mScw, P)) { fprintf(stderr, "pack_loop refused the graph\n"); return 1; }
    FILE *f = fopen(out_path, "wb"); if (!f) return 2;
    put(f, "g_pose", 0, P.pose.data(), P.pose.size()); put(f, "g_fixed", 2, P.fixed.data(), P.fixed.size());
    put(f, "g_edge_i", 1, P.edge_i.data(), P.edge_i.size()); put(f, "g_edge_j", 1, P.edge_j.data(), P.edge_j.size()); put(f, "g_meas", 0, P.meas.data(), P.meas.size());
    printf("gather done: loop problem of %d keyframes, %d connections\n", P.p.n_kf, P.p.n_edge);

    void *ctx = nullptr; int rc = tsloop_create(0, &ctx);
    if (rc == TSLOOP_ERR_DEVICE) { fclose(f); printf("no HIP device: gather only\n"); return 3; }
    if (rc) { fclose(f); fprintf(stderr, "tsloop_create: %d\n", rc); return 1; }
    tsloop_options o; tsloop_default_options_loop(&o); tsloop_report rep;
    rc = tsloop_optimize_loop(ctx, &P.p, &o, &rep);
    if (rc != TSLOOP_OK) { fclose(f); fprintf(stderr, "tsloop_optimize_loop: %d (%s)\n", rc, tsloop_last_error(ctx)); return 1; }
    tsloop_adapter::scatter_loop<Traits>(P, vKFs, vPts, vObjs);
    std::vector<double> T34, rho, theta;
    for (size_t k = 0; k < n; k++) for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) T34.push_back(store[k].mTcw(r, c));
    for (size_t j = 0; j < n_pt; j++) rho.push_back(pts[j].GetInverD());
    for (size_t j = 0; j < n_tx; j++) { const Mat31 &N = txs[j].RefKF->mNcr[(size_t)txs[j].GetNidx()]; for (int a = 0; a < 3; a++) theta.push_back(N(a)); }
    int32_t meta[2] = { rep.iters, rep.termination };
    put(f, "pose_solved", 0, P.pose.data(), P.pose.size()); put(f, "T34", 0, T34.data(), T34.size()); put(f, "rho", 0, rho.data(), rho.size());
    put(f, "theta", 0, theta.data(), theta.size()); put(f, "meta", 1, meta, 2); put(f, "cost1", 0, &rep.cost1, 1);
    fclose(f); tsloop_destroy(ctx);
    printf("solve + scatter done: loop, %d iterations\n", rep.iters);
    return 0;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s dump.bin sim3|loop out.bin\n", argv[0]); return 2; }
    Dump d; if (!read_dump(argv[1], d)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    const std::string mode = argv[2];
    if (mode == "sim3") return run_sim3(d, argv[3]);
    if (mode == "loop") return run_loop(d, argv[3]);
    fprintf(stderr, "unknown mode %s\n", mode.c_str()); return 2;
}
