// optimizer_tsba.cc -- drop-in bodies of TextSLAM's optimizer:: methods over libtsba.so (include/tsba.h).
//
// Lives in the TextSLAM tree next to src/optimizer.cc and replaces the bodies of the six BA / pose methods of src/optimizer.h:57-66
// (signatures, callers -- tracking.cc:447,521,561,833,839, loopClosing.cc:589 -- and the host-side bookkeeping stay as they are):
// each body is gather (adapter/tsba_gather.hpp) -> one C-ABI call -> scatter -> the reference's own UpdateTrackedText* call.
// OptimizeSim3 / OptimizeLoop map onto include/tsloop.h the same way.  Build: add this file instead of the Pyr*
// methods, -I<this repo>/include -I<this repo>/adapter, link -ltsba.  Not compiled by this repository (needs TextSLAM + Eigen +
// OpenCV); the gather / scatter templates it instantiates are compiled and tested here through tests/cxx/abi_from_cxx.cpp.
#include <optimizer.h>
#include "tsba.h"
#include "tsba_gather.hpp"
#include "textslam_traits.hpp"

namespace TextSLAM {
using tsba_adapter::Packed;
typedef tsba_adapter::TextSlamTraits TT;

namespace {
// One context per calling THREAD: TextSLAM's tracking, mapping and loop-closing threads call these methods concurrently (tracking.cc:447-561,
// loopClosing.cc:589) and a tsba context serves one caller at a time (include/tsba.h).  thread_local keeps this file free of header changes;
// with a header change the context would be a member of the optimizer object guarded like its other state.
// The holder destroys the context when its thread exits (slabs, plane cache of 64 pyramid slots and streams go with it).  The label image below
// reads the state of the last solve OF THE CALLING THREAD: UpdateTrackedTextBA runs on the thread that ran the BA (optimizer.cc:322-326), as here.
struct CtxHolder { void *ctx = nullptr; ~CtxHolder() { if (ctx) tsba_destroy(ctx); } };
void *tsba_ctx() { static thread_local CtxHolder h;
    if (!h.ctx && tsba_abi_version() != TSBA_ABI_VERSION) { std::cerr << "libtsba.so: ABI version " << tsba_abi_version() << ", this adapter was built against " << TSBA_ABI_VERSION << std::endl; exit(-1); }
    if (!h.ctx && tsba_create(&h.ctx, 0) != TSBA_OK) { std::cerr << "tsba_create: no usable HIP device" << std::endl; exit(-1); } return h.ctx; }
// what the window's keyframes and planes contributed to the last calls of THIS thread (adapter/tsba_gather.hpp: GatherCache).  Cleared by the entry points
// that run where observations are re-targeted without their lists changing length (loop closing: mapPts::Replace before GlobalBA, loopClosing.cc:560-591)
tsba_adapter::GatherCache &gather_cache() { static thread_local tsba_adapter::GatherCache c; return c; }
void k_of(const Mat33 &K, double out[4]) { out[0] = K(0, 0); out[1] = K(1, 1); out[2] = K(0, 2); out[3] = K(1, 2); }
// TSBA_ERR_NUMERIC = the linear solver broke down in some LM trial (Ceres: termination FAILURE): the one-shot entry points have still
// downloaded the LAST ACCEPTED state and the flags of the passes that ran (tsba.hip: one_shot returns the status after tsba_download), so
// the scatter below writes a consistent state -- as Ceres leaves the parameter blocks at the best iterate.  Everything else is an error.
bool failed(int rc, const char *what) { if (rc != TSBA_OK && rc != TSBA_ERR_NUMERIC) { std::cerr << what << ": " << tsba_last_error(tsba_ctx()) << std::endl; return true; } return false; }
// label image of keyframe `kf` for the state left by the last solve: the TextLabelImg of ShowBAReproj_TextBox (optimizer.cc:2508-2582)
cv::Mat label_image(int kf, const cv::Size &size) { cv::Mat lab(size, CV_32F); tsba_text_label_image(tsba_ctx(), kf, 0, (float *)lab.data); return lab; }
}

void optimizer::LocalBundleAdjustment(map *mpMap, vector<keyframe *> vKFs, const BAStatus &STATE) {
    vector<mapPts *> vMapPts = mpMap->GetAllMapPoints();
    vector<mapText *> vMapTexts = mpMap->GetAllMapTexts(TEXTGOOD);
    double K[4]; k_of(vK[0], K);
    static thread_local Packed P_keep; Packed &P = P_keep; P.reset();                                                       // (one per thread: the arrays' storage stays)
    tsba_adapter::pack_map<TT>(mpMap, vKFs, vMapPts, vMapTexts, /*local*/0, /*levels 0..2*/3, K, !bFlag_noText, P, nullptr, nullptr, &gather_cache());           // optimizer.cc:201-279, :1366-1557
    tsba_options o; tsba_default_options_local(&o);                                                                        // :282-289
    o.state = STATE == LOCAL ? TSBA_STATE_LOCAL : STATE == GLOBAL ? TSBA_STATE_GLOBAL : TSBA_STATE_NOTREACHWIN;              // :1571-1588
    o.use_text = !bFlag_noText; o.outlier_scene = o.outlier_text = !bFlag_rapid;                                             // :1339-1345
    tsba_report rep;
    if (failed(tsba_local_ba(tsba_ctx(), &P.p, &o, &rep), "tsba_local_ba")) return;
    tsba_adapter::scatter_map<TT>(P, vKFs, vMapPts, vMapTexts, /*poses*/true, /*flags*/true);                               // :292-326
    keyframe *KFCur = vKFs[vKFs.size() - 1];
    cv::Mat ImgTextLabel = label_image((int)vKFs.size() - 1, KFCur->vFrameImg[0].size());
    vector<int> vIdxGOOD2Raw = GetNewIdxForTextState(KFCur->vObvText, TEXTGOOD);
    UpdateTrackedTextBA(KFCur->vObvText, vIdxGOOD2Raw, ImgTextLabel, KFCur, false);                                         // :328-329
}

void optimizer::GlobalBA(map *mpMap) {
    vector<keyframe *> vKFs = mpMap->GetAllKeyFrame();
    vector<mapPts *> vMapPts = mpMap->GetAllMapPoints(false);
    vector<mapText *> vMapTexts = mpMap->GetAllMapTexts(TEXTGOOD);
    double K[4]; k_of(vK[0], K);
    Packed P;
    gather_cache().invalidate();                                                                                           // (a loop closure re-targets observations: what the local windows cached is stale)
    tsba_adapter::pack_map<TT>(mpMap, vKFs, vMapPts, vMapTexts, /*global*/1, 1, K, /*FLAG_TEXT = false, :1707*/false, P);
    tsba_options o; tsba_default_options_global(&o);                                                                       // :411-414
    tsba_report rep;
    const int rc_g = tsba_global_ba(tsba_ctx(), &P.p, &o, &rep);
    if (failed(rc_g, "tsba_global_ba") || rc_g == TSBA_ERR_NUMERIC) exit(-1);                                              // (the reference exits when the solve is not usable, :1842-1845)
    tsba_adapter::scatter_map<TT>(P, vKFs, vMapPts, vMapTexts, true, false);                                              // :417-451
}

void optimizer::OptimizeLandmarker(map *mpMap) {
    vector<keyframe *> vKFs = mpMap->GetAllKeyFrame();
    vector<mapPts *> vMapPts = mpMap->GetAllMapPoints(false);
    vector<mapText *> vMapTexts = mpMap->GetAllMapTexts(TEXTGOOD);
    double K[4]; k_of(vK[0], K);
    Packed P;
    tsba_adapter::pack_map<TT>(mpMap, vKFs, vMapPts, vMapTexts, /*landmarker: every pose constant*/2, 4, K, !bFlag_noText, P);
    tsba_options o; tsba_default_options_landmarker(&o);                                                                  // :531-541
    tsba_report rep;
    if (failed(tsba_local_ba(tsba_ctx(), &P.p, &o, &rep), "tsba_local_ba(landmarker)")) return;
    tsba_adapter::scatter_map<TT>(P, vKFs, vMapPts, vMapTexts, /*poses stay*/false, true);                                // :543-556
    for (int back = 2; back >= 1; back--) {                                                                                // the two newest keyframes, :558-561
        keyframe *KF = vKFs[vKFs.size() - back];
        cv::Mat lab = label_image((int)vKFs.size() - back, KF->vFrameImg[0].size());
        vector<int> vIdx = GetNewIdxForTextState(KF->vObvText, TEXTGOOD);
        UpdateTrackedTextBA(KF->vObvText, vIdx, lab, KF, false);
    }
}

void optimizer::PoseOptim(frame &F) {
    double K[4]; k_of(vK[0], K);
    Packed P;
    tsba_adapter::pack_pose<TT>(F, 4, K, !bFlag_noText, P);                                                               // :139-172, :1104-1207
    tsba_options o; tsba_default_options_pose(&o);                                                                        // levels 2,1,0
    if (bFlag_rapid) {                                                                                                     // :181-182: level 3 first, no outlier passes
        o.n_passes = 4; for (int i = 0; i < 4; i++) { o.levels[i] = 3 - i; o.its[i] = 10; o.chi2_mono[i] = 12.25; o.chi2_text[i] = i == 3 ? 0.95 : 0.5; }
        o.outlier_scene = o.outlier_text = 0;
    }
    o.use_text = !bFlag_noText;
    tsba_report rep;
    if (failed(tsba_pose_optim(tsba_ctx(), &P.p, &o, &rep), "tsba_pose_optim")) return;
    tsba_adapter::scatter_pose<TT>(P, F);                                                                                 // :188-190
    vector<TextObservation *> TextObjs;
    for (size_t i = 0; i < F.vObvText.size(); i++) if (F.vObvText[i]->obj->STATE == TEXTGOOD) TextObjs.push_back(F.vObvText[i]);
    cv::Mat ImgTextLabel = label_image(0, F.vFrameImg[0].size());
    UpdateTrackedTextPOSE(TextObjs, ImgTextLabel, F);                                                                      // :192
}

void optimizer::InitBA(keyframe *F1, keyframe *F2) {
    double K[4]; k_of(vK[0], K);
    Packed P;
    tsba_adapter::pack_init<TT>(*F1, *F2, 4, K, P);                                                                       // :59-104, :978-1030
    tsba_options o; tsba_default_options_init(&o);
    tsba_report rep;
    if (failed(tsba_local_ba(tsba_ctx(), &P.p, &o, &rep), "tsba_local_ba(init)")) return;
    tsba_adapter::scatter_init<TT>(P, *F1, *F2);                                                                          // :119-129
    cv::Mat ImgTextLabel = label_image(1, F2->vFrameImg[0].size());
    UpdateTrackedTextBA(F2->vObvText, ImgTextLabel, F2, true);                                                             // :131
}

bool optimizer::ThetaOptimMultiFs(const frame &F, mapText *&obj) {
    double K[4]; k_of(vK[0], K);
    Packed P;
    tsba_adapter::pack_theta<TT>(F, *obj, 3, K, P);                                                                       // :565-603
    tsba_options o; tsba_default_options_theta(&o);                                                                       // levels 2,1,0 x 50, no loss
    tsba_report rep; double cov[9];
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) cov[3*r + c] = obj->Covariance(r, c);                          // kept when the information matrix is singular
    const int rc = tsba_theta_optim(tsba_ctx(), &P.p, &o, 0, cov, &rep);
    if (rc != TSBA_OK) { cout << "PyrThetaOptim failed, return false." << endl; return false; }                            // :605-618
    TT::set_theta(*obj, P.p.theta);                                                                                        // :620-621
    Mat33 thetaVariance; for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) thetaVariance(r, c) = cov[3*r + c];
    obj->Covariance = thetaVariance;                                                                                       // :622
    return true;
}

}  // namespace TextSLAM
