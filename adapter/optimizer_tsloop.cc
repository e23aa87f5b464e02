// optimizer_tsloop.cc -- drop-in bodies of TextSLAM's two loop-closing optimisers over libtsloop.so (include/tsloop.h).
//
// Lives in the TextSLAM tree next to src/optimizer.cc and replaces the bodies of optimizer::OptimizeSim3 (src/optimizer.cc:626-731,
// called from loopClosing::ComputeSim3) and optimizer::OptimizeLoop (:733-957, called from loopClosing::CorrectLoop); signatures and
// callers stay as they are.  Each body is gather (adapter/tsloop_gather.hpp) -> one C-ABI call -> scatter.  Together with
// adapter/optimizer_tsba.cc these are the eight public methods of src/optimizer.h:57-70.  Build: -I<this repo>/include
// -I<this repo>/adapter, link -ltsloop.  Not compiled by this repository (needs TextSLAM + Eigen + OpenCV); the templates it instantiates
// are compiled and tested here through tests/cxx/loop_from_cxx.cpp.
#include <optimizer.h>
#include "tsloop.h"
#include "tsloop_gather.hpp"
#include "textslam_traits.hpp"

namespace TextSLAM {
typedef tsba_adapter::TextSlamTraits TT;

namespace {
struct CtxHolder { void *ctx = nullptr; ~CtxHolder() { if (ctx) tsloop_destroy(ctx); } };          // one context per calling thread, destroyed with it
void *tsloop_ctx() { static thread_local CtxHolder h; if (!h.ctx && tsloop_create(0, &h.ctx) != TSLOOP_OK) { std::cerr << "tsloop_create: no usable HIP device" << std::endl; exit(-1); } return h.ctx; }
}

int optimizer::OptimizeSim3(vector<FeatureConvert> &vFeat1, vector<FeatureConvert> &vFeat2, vector<bool> &vbInliers, Sim3_loop &Sim12, const float th2) {
    (void)th2;                                                                                   // unused by the reference too (threshOutlier = 4.0, :629)
    const double Kf[4] = { K(0, 0), K(1, 1), K(0, 2), K(1, 2) };                                 // K1 = K2 = K, :633-634
    tsloop_adapter::PackedSim3 P;
    tsloop_adapter::pack_sim3<TT>(vFeat1, vFeat2, vbInliers, Sim12, Kf, P);                      // :636-669
    tsloop_options o; tsloop_default_options_sim3(&o);                                           // HuberLoss(sqrt(10)), 20 iterations, 4 px
    tsloop_report rep;
    if (tsloop_optimize_sim3(tsloop_ctx(), &P.p, &o, &rep) != TSLOOP_OK) { std::cerr << "tsloop_optimize_sim3: " << tsloop_last_error(tsloop_ctx()) << std::endl; return 0; }
    return tsloop_adapter::scatter_sim3<TT>(P, rep, vbInliers, Sim12);                           // :683-729
}

void optimizer::OptimizeLoop(std::map<keyframe *, set<keyframe *>> &LoopConnections, std::map<keyframe *, set<keyframe *>> &NormConnections,
                             keyframe *KF, keyframe *LoopKF,
                             std::map<keyframe *, Sim3_loop, std::less<keyframe *>, Eigen::aligned_allocator<std::pair<keyframe *, Sim3_loop>>> &vConnectKFs,
                             Sim3_loop &mScw, map *mpMap) {
    vector<keyframe *> vKFs = mpMap->GetAllKeyFrame();                                           // :741
    tsloop_adapter::PackedLoop P;
    if (!tsloop_adapter::pack_loop<TT>(vKFs, LoopConnections, NormConnections, KF, LoopKF, vConnectKFs, mScw, P)) {   // :745-869
        std::cerr << "OptimizeLoop: a keyframe id does not index the pose table" << std::endl; exit(-1); }
    tsloop_options o; tsloop_default_options_loop(&o);                                           // 20 iterations, no loss, :871-877
    tsloop_report rep;
    if (tsloop_optimize_loop(tsloop_ctx(), &P.p, &o, &rep) != TSLOOP_OK) { cerr << "Loop pose Optimize failed !"; exit(-1); }   // :880-883
    vector<mapPts *> vPts = mpMap->GetAllMapPoints();
    vector<mapText *> vObjs = mpMap->GetAllMapTexts();
    tsloop_adapter::scatter_loop<TT>(P, vKFs, vPts, vObjs);                                      // :884-956
}

}  // namespace TextSLAM
