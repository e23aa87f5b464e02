// tsloop_gather.hpp -- gather / scatter between TextSLAM's loop-closing data and the flat problems of include/tsloop.h.
// Header-only, C++11, no third-party include; like adapter/tsba_gather.hpp the templates reach the reference's objects through the
// member names the reference uses, so they compile against the real types (adapter/textslam_traits.hpp + adapter/optimizer_tsloop.cc,
// inside the TextSLAM tree) and against plain structs of the same shape (tests/cxx/mock_textslam.hpp, tests/cxx/loop_from_cxx.cpp).
//
// What each function restates (citations relative to the TextSLAM tree):
//   pack_sim3      optimizer::OptimizeSim3   src/optimizer.cc:633-669   parameter block Sim12 = (q normalised | t | s), one pair of residual blocks
//                                                                       per match that is still an inlier
//   scatter_sim3                             :683-729                   Sim12 = (q normalised, t, s); vbInliers; the return value
//   pack_loop      optimizer::OptimizeLoop   :745-869                   vScwIni / pose[mnId], the normal and the loop connections in the iteration
//                                                                       order of the reference's std::map / std::set, keyframes 0, 1 and LoopKF constant
//   scatter_loop                             :884-956                   SetPose([R | t/s]), rho *= s(RefKF), theta *= s(RefKF)
// The solves in between -- ceres::Solve over auto_sim / auto_siminv, and over numer_loop_ver2 -- are what libtsloop.so replaces.
//
// Traits, in addition to those of tsba_gather.hpp:
//   Sim3                                                        the reference's Sim3_loop (setting.h:129-171): members r, t, s; inverse(); operator*
//   Sim3 sim_of_pose(const R33 &R, const V3 &t, double s)       Sim3_loop(R, t, s)
//   Sim3 sim_make(const double q[4], const double t[3], double s)   Sim3_loop(Quaterniond(w, x, y, z).normalized(), t, s)
//   void sim_get(const Sim3 &S, bool normalise, double out[8])  (w x y z | t | s); normalise: q = q.normalized() first (:637-638, :766-767)
//   void set_pose_sim(KeyFrame &, const double pose[8])         q normalised, T = [R(q) | t / s], SetPose(T)            (:887-906)
#ifndef TSLOOP_GATHER_HPP
#define TSLOOP_GATHER_HPP

#include <cstdint>
#include <cstring>
#include <map>
#include <vector>
#include "tsloop.h"

namespace tsloop_adapter {

// ---------------------------------------------------------------------------------------------------------------- OptimizeSim3
struct PackedSim3 {
    std::vector<double> P1, P2; std::vector<float> uv1, uv2; std::vector<uint8_t> inlier;
    tsloop_sim3_problem p;
    PackedSim3() { std::memset(&p, 0, sizeof(p)); }
};

// vFeat1 / vFeat2: std::vector<FeatureConvert> (posObv: the matched point in its own camera frame, obv2d.pt: the keypoint);
// vbInliers: std::vector<bool>; Sim12: the initial Sim3 (camera 2 -> camera 1); K = (fx, fy, cx, cy) of optimizer::K
template <class T, class FeatVec, class BoolVec>
inline void pack_sim3(const FeatVec &vFeat1, const FeatVec &vFeat2, const BoolVec &vbInliers, const typename T::Sim3 &Sim12, const double K[4], PackedSim3 &P) {
    const size_t n = vFeat2.size();                                        // assert(vFeat1.size() == vFeat2.size()), optimizer.cc:655
    P.P1.resize(3*n); P.P2.resize(3*n); P.uv1.resize(2*n); P.uv2.resize(2*n); P.inlier.resize(n);
    for (size_t i = 0; i < n; i++) {
        for (int a = 0; a < 3; a++) { P.P1[3*i + a] = vFeat1[i].posObv(a, 0); P.P2[3*i + a] = vFeat2[i].posObv(a, 0); }
        P.uv1[2*i] = vFeat1[i].obv2d.pt.x; P.uv1[2*i + 1] = vFeat1[i].obv2d.pt.y;              // Vec2(vFeat1[ip2].obv2d.pt.x, .y), :662
        P.uv2[2*i] = vFeat2[i].obv2d.pt.x; P.uv2[2*i + 1] = vFeat2[i].obv2d.pt.y;              // :667
        P.inlier[i] = vbInliers[i] ? 1 : 0;                                                    // if (!vbInliers[ip2]) continue, :657
    }
    std::memset(&P.p, 0, sizeof(P.p));
    P.p.n = (int32_t)n; P.p.P1 = P.P1.data(); P.p.P2 = P.P2.data(); P.p.uv1 = P.uv1.data(); P.p.uv2 = P.uv2.data(); P.p.inlier = P.inlier.data();
    for (int k = 0; k < 4; k++) P.p.K[k] = K[k];                                                // K1 = K2 = K, :633-634
    T::sim_get(Sim12, true, P.p.sim);                                                          // q = Sim12.r.normalized(), :637-646
}
// after tsloop_optimize_sim3: Sim12 and vbInliers as the reference leaves them; returns numInlier
template <class T, class BoolVec>
inline int scatter_sim3(const PackedSim3 &P, const tsloop_report &rep, BoolVec &vbInliers, typename T::Sim3 &Sim12) {
    Sim12 = T::sim_make(P.p.sim, P.p.sim + 4, P.p.sim[7]);                                      // gScmRes(q12.normalized(), t12, s12), :683-701
    for (size_t i = 0; i < P.inlier.size(); i++) if (!P.inlier[i]) vbInliers[i] = false;        // :719-722 (a match never comes back)
    return rep.n_inlier;                                                                        // :728
}

// ---------------------------------------------------------------------------------------------------------------- OptimizeLoop
struct PackedLoop {
    std::vector<double> pose, meas; std::vector<uint8_t> fixed; std::vector<int32_t> edge_i, edge_j;
    tsloop_graph_problem p;
    PackedLoop() { std::memset(&p, 0, sizeof(p)); }
};

// LoopConnections / NormConnections: std::map<keyframe *, std::set<keyframe *>>; vConnectKFs: std::map<keyframe *, Sim3_loop, ...> (the corrected
// Sim3 of the current keyframe's neighbourhood); mScw: the corrected Sim3 of KF itself.  Returns false when a keyframe id does not index the
// pose table (the reference indexes pose[] of vKFs.size() entries by mnId and would write out of bounds).
// (UseEssential = false in the reference, :738: the covisibility-weight filter of :807-811 / :843-847 is off and is not restated.)
template <class T, class ConnMap, class SimMap>
inline bool pack_loop(const std::vector<typename T::KeyFrame *> &vKFs, ConnMap &LoopConnections, ConnMap &NormConnections,
                      typename T::KeyFrame *KF, typename T::KeyFrame *LoopKF, SimMap &vConnectKFs, const typename T::Sim3 &mScw, PackedLoop &P) {
    typedef typename T::KeyFrame KeyFrame; typedef typename T::Sim3 Sim3;
    const size_t n = vKFs.size();
    P.pose.assign(8*n, 0.0); P.fixed.assign(n, 0); P.edge_i.clear(); P.edge_j.clear(); P.meas.clear();
    std::map<KeyFrame *, Sim3> vScwIni;
    for (size_t k = 0; k < n; k++) {                                                            // :745-778
        KeyFrame *kf = vKFs[k];
        const size_t id = (size_t)kf->mnId; if (id >= n) return false;
        double q[4]; T::quat_of(kf->mRcw, q);                                                   // Eigen::Quaterniond q(Rcw); q = q.normalized()
        double t[3] = { kf->mtcw(0, 0), kf->mtcw(1, 0), kf->mtcw(2, 0) }; double s = 1.0;
        Sim3 S = T::sim_make(q, t, s);
        if (vConnectKFs.count(kf)) { double a[8]; T::sim_get(vConnectKFs[kf], true, a); S = T::sim_make(a, a + 4, a[7]); }   // UseSTrans, :762-770
        vScwIni[kf] = S;
        T::sim_get(S, false, &P.pose[8*id]);
    }
    auto add_edge = [&](size_t i, size_t j, const Sim3 &Sji) { double a[8]; T::sim_get(Sji, false, a);          // numer_loop_ver2::Create(Sji.r, Sji.t, Sji.s)
        P.edge_i.push_back((int32_t)i); P.edge_j.push_back((int32_t)j); P.meas.insert(P.meas.end(), a, a + 8); };
    for (typename ConnMap::iterator it = NormConnections.begin(); it != NormConnections.end(); ++it) {          // normal edges, :788-820
        KeyFrame *KFi = it->first;
        const Sim3 Siw = T::sim_of_pose(KFi->mRcw, KFi->mtcw, 1.0);
        for (typename ConnMap::mapped_type::const_iterator sit = it->second.begin(); sit != it->second.end(); ++sit) {
            KeyFrame *KFj = *sit;
            const Sim3 Sjw = T::sim_of_pose(KFj->mRcw, KFj->mtcw, 1.0);
            if ((size_t)KFi->mnId >= n || (size_t)KFj->mnId >= n) return false;
            add_edge((size_t)KFi->mnId, (size_t)KFj->mnId, Sjw*Siw.inverse());
        }
    }
    for (typename ConnMap::iterator it = LoopConnections.begin(); it != LoopConnections.end(); ++it) {          // loop edges, :823-858
        KeyFrame *KFj = it->first;
        Sim3 Sjw = vScwIni[KFj];
        if (KFj->mnId == KF->mnId) Sjw = mScw;
        for (typename ConnMap::mapped_type::const_iterator sit = it->second.begin(); sit != it->second.end(); ++sit) {
            KeyFrame *KFi = *sit;
            if ((size_t)KFi->mnId >= n || (size_t)KFj->mnId >= n) return false;
            add_edge((size_t)KFi->mnId, (size_t)KFj->mnId, Sjw*vScwIni[KFi].inverse());
        }
    }
    const size_t fix[3] = { 0, 1, (size_t)LoopKF->mnId };                                       // vNeedFix, :861-869
    for (int a = 0; a < 3; a++) { if (fix[a] >= n) return false; P.fixed[fix[a]] = 1; }
    std::memset(&P.p, 0, sizeof(P.p));
    P.p.n_kf = (int32_t)n; P.p.n_edge = (int32_t)P.edge_i.size();
    P.p.pose = P.pose.data(); P.p.fixed = P.fixed.data(); P.p.edge_i = P.edge_i.data(); P.p.edge_j = P.edge_j.data(); P.p.meas = P.meas.data();
    return true;
}
// after tsloop_optimize_loop: the reference's map update (:884-956) on the returned poses -- keyframe poses with the scale folded into the
// translation, inverse depths and plane parameters of every landmark scaled by its host keyframe's s
template <class T>
inline void scatter_loop(const PackedLoop &P, const std::vector<typename T::KeyFrame *> &vKFs,
                         const std::vector<typename T::MapPt *> &vPts, const std::vector<typename T::MapText *> &vObjs) {
    for (size_t k = 0; k < vKFs.size(); k++) T::set_pose_sim(*vKFs[k], &P.pose[8*(size_t)vKFs[k]->mnId]);     // :886-907
    for (size_t j = 0; j < vPts.size(); j++) {                                                                   // :912-941 (pwCorr / rhoCorr there are computed and not used)
        double rho = vPts[j]->GetInverD();
        rho *= P.pose[8*(size_t)vPts[j]->RefKF->mnId + 7];
        vPts[j]->SetRho(rho);
    }
    for (size_t j = 0; j < vObjs.size(); j++) {                                                                  // :943-949
        const double s = P.pose[8*(size_t)vObjs[j]->RefKF->mnId + 7];
        auto theta = vObjs[j]->RefKF->mNcr[(size_t)vObjs[j]->GetNidx()];
        const double th[3] = { theta(0, 0)*s, theta(1, 0)*s, theta(2, 0)*s };
        T::set_theta(*vObjs[j], th);
    }
}

}  // namespace tsloop_adapter
#endif
