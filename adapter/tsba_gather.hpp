// tsba_gather.hpp -- gather / scatter between TextSLAM's object graph and the flat tsba_problem of include/tsba.h
// (rows B1 and O2 of SURVEY.md 8a).  Header-only, C++11, no third-party include: the object graph is reached through the
// member names the reference uses (keyframe::mRcw, mapPts::GetInverD(), mapText::vRefFeature ...), so the same templates
// compile against
//   * the real TextSLAM types  -- adapter/textslam_traits.hpp + adapter/optimizer_tsba.cc, built inside the TextSLAM tree, and
//   * plain structs of the same shape -- tests/cxx/mock_textslam.hpp, built and run by this repository's test-suite
//     (tests/cxx/abi_from_cxx.cpp: the C ABI driven from the language the reference is written in).
//
// What each function restates (citations relative to the TextSLAM tree):
//   pack_local       optimizer::LocalBundleAdjustment  src/optimizer.cc:197-279 + the problem construction of PyrBA :1366-1557
//   pack_global      optimizer::GlobalBA               :334-410            + PyrGlobalBA :1727-1765
//   pack_landmarker  optimizer::OptimizeLandmarker     :456-530            + PyrLandmarkers :1880-2060
//   pack_pose        optimizer::PoseOptim              :135-172            + PyrPoseOptim :1104-1207
//   pack_init        optimizer::InitBA                 :57-104             + PyrIniBA :978-1030
//   pack_theta       optimizer::ThetaOptimMultiFs      :565-603            + PyrThetaOptim :2170-2199
//   scatter_*        the write-back halves: :292-326 (local / global), :119-129 (init), :188-190 (pose), :531-547 (landmarker)
// The numeric core between gather and scatter -- the Pyr* methods -- is what libtsba.so replaces.
//
// Traits (a struct of static functions, see the two implementations named above):
//   Map, KeyFrame, Frame, MapPt, MapText            the object-graph types
//   int  text_good()                                 the TEXTGOOD enumerator
//   void quat_of(const R33 &R, double q[4])          Eigen::Quaterniond(R).normalized() as (w, x, y, z)        optimizer.cc:84-90
//   void set_pose(KeyFrame / Frame &, const double pose[7])   normalise q, build Tcw, SetPose(Tcw)              :292-312
//   void set_theta(MapText &, const double th[3])    RefKF->SetN(Mat31(th), GetNidx())                         :321-325
//   const uint8_t *img(const Image &), int img_w(const Image &), int img_h(const Image &)    continuous CV_8UC1 (nume_BAText.h:25)
#ifndef TSBA_GATHER_HPP
#define TSBA_GATHER_HPP

#include <cstdint>
#include <cstring>
#include <map>
#include <atomic>
#include <vector>
#include "tsba.h"

namespace tsba_adapter {

// Owns the flat arrays of one call; `p` points into them (finish()).
struct Packed {
    std::vector<double> pose, rho, theta, pt_ray, pt_Trw, text_Twr, text_box;
    std::vector<double> sobs_uv0[TSBA_MAX_LEVELS], tfeat_uv[TSBA_MAX_LEVELS], tfeat_ref[TSBA_MAX_LEVELS];
    std::vector<int32_t> pt_host, text_host, sobs_kf[TSBA_MAX_LEVELS], sobs_pt[TSBA_MAX_LEVELS], sobs_flag[TSBA_MAX_LEVELS];
    std::vector<int32_t> tfeat_off[TSBA_MAX_LEVELS], tfeat_raw[TSBA_MAX_LEVELS], tobs_kf, tobs_text, tobs_fgood_off;
    std::vector<uint8_t> kf_initial, sgood, tobs_good, tfgood;
    std::vector<const uint8_t *> img[TSBA_MAX_LEVELS];
    std::vector<int64_t> kf_id;             // keyframe::mnId of every keyframe (tsba_problem.kf_id: the context keeps the pyramid planes of keyframes it has seen)
    // bookkeeping for the scatter
    std::vector<int32_t> kf_flag_off;       // sgood offset of keyframe k's vObvGoodPts
    std::vector<int32_t> tobs_raw;          // index of text observation t in its keyframe's vObvText (vObvGoodTexts / vObvGoodTextFeats row)
    tsba_problem p;

    Packed() { std::memset(&p, 0, sizeof(p)); }
    // empty, with every array's storage kept: an adapter that packs a window per keyframe reuses ONE Packed (3 MB of arrays at 20 keyframes x 5000 points --
    // allocating and first-touching them was a quarter of a cached gather)
    void reset() {
        pose.clear(); rho.clear(); theta.clear(); pt_ray.clear(); pt_Trw.clear(); text_Twr.clear(); text_box.clear();
        pt_host.clear(); text_host.clear(); tobs_kf.clear(); tobs_text.clear(); tobs_fgood_off.clear();
        kf_initial.clear(); sgood.clear(); tobs_good.clear(); tfgood.clear(); kf_id.clear(); kf_flag_off.clear(); tobs_raw.clear();
        for (int l = 0; l < TSBA_MAX_LEVELS; l++) { sobs_uv0[l].clear(); tfeat_uv[l].clear(); tfeat_ref[l].clear(); sobs_kf[l].clear(); sobs_pt[l].clear(); sobs_flag[l].clear();
            tfeat_off[l].clear(); tfeat_raw[l].clear(); img[l].clear(); }
        std::memset(&p, 0, sizeof(p));
    }

    // wire the pointers of tsba_problem to the vectors (after the last push_back)
    void finish(int n_levels, const double K[4], const int img_w[TSBA_MAX_LEVELS], const int img_h[TSBA_MAX_LEVELS]) {
        p.n_kf = (int32_t)(pose.size()/7); p.n_pt = (int32_t)rho.size(); p.n_text = (int32_t)(theta.size()/3); p.n_levels = n_levels;
        for (int k = 0; k < 4; k++) p.K[k] = K[k];
        p.pose = pose.data(); p.rho = rho.data(); p.theta = theta.data(); p.kf_initial = kf_initial.data();
        p.pt_ray = pt_ray.data(); p.pt_host = pt_host.data(); p.pt_host_Trw = pt_Trw.data();
        p.text_host = text_host.data(); p.text_host_Twr = text_Twr.data(); p.text_box_ray = text_box.data();
        p.n_sgood = (int32_t)sgood.size(); p.sgood = sgood.data();
        p.n_tobs = (int32_t)tobs_kf.size(); p.tobs_kf = tobs_kf.data(); p.tobs_text = tobs_text.data(); p.tobs_good = tobs_good.data();
        if (tobs_fgood_off.empty()) tobs_fgood_off.push_back(0);
        p.tobs_fgood_off = tobs_fgood_off.data(); p.tfgood = tfgood.data();
        p.kf_id = (kf_id.size() == pose.size()/7 && !kf_id.empty()) ? kf_id.data() : nullptr;
        for (int l = 0; l < n_levels; l++) {
            p.n_sobs[l] = (int32_t)sobs_kf[l].size();
            p.sobs_kf[l] = sobs_kf[l].data(); p.sobs_pt[l] = sobs_pt[l].data(); p.sobs_flag[l] = sobs_flag[l].data(); p.sobs_uv0[l] = sobs_uv0[l].data();
            p.n_tfeat[l] = (int32_t)tfeat_raw[l].size();
            p.tfeat_off[l] = tfeat_off[l].empty() ? nullptr : tfeat_off[l].data();
            p.tfeat_raw[l] = tfeat_raw[l].data(); p.tfeat_uv[l] = tfeat_uv[l].data(); p.tfeat_ref[l] = tfeat_ref[l].data();
            p.img[l] = img[l].empty() ? nullptr : img[l].data(); p.img_w[l] = img_w[l]; p.img_h[l] = img_h[l];
        }
    }
};

// ---- what a keyframe / a text plane contributes to EVERY window it is part of, kept between calls -----------------------------------
// optimizer::LocalBundleAdjustment runs once per new keyframe on a window that slid by one (tracking.cc:826-842): 19 of its 20 keyframes were in the
// last call, and walking their observation lists again -- obs[s]->IdxToRaw, vObvPts[raw]->pt->mnId, vSceneObv2d[0][raw]->feature: three dependent
// pointer hops per observation and level -- is most of the gather's time (1.1 ms of a 4.6 ms call at 20 keyframes x 5000 points).  A keyframe's
// observation lists are append-only in the reference (keyframe::AddSceneObserv, keyframe.cc:96-114: the only writers besides the constructor; called
// by loop closing, loopClosing.cc:1259), and a plane's reference features are fixed when the plane is created (mapText.cc:64-107).  So a segment is
// cached under the object's mnId and reused while the lengths of the lists it was built from are unchanged; what a call still does per observation
// is one table lookup (map point id -> index in THIS call's point list) and four stores.  Point / plane parameters, poses and every flag are read
// from the object graph in every call -- only topology is kept.  invalidate() after anything that re-targets observations without changing the lists'
// lengths (mapPts::Replace at a loop closure, mapPts.cc:170-190): the adapter's GlobalBA / loop entry points call it.
// The arrays a cached gather produces are the arrays pack_map produces without a cache, element for element (tests/cxx: slide_check).
// A process-wide topology epoch: whatever re-targets observations WITHOUT changing list lengths (mapPts::Replace / keyframe::ReplaceMapPt at a loop closure, on
// whichever thread it runs) bumps it once; every thread's cache compares it in begin_call() and drops what it holds (round-5 advisor: a thread_local cache
// invalidated only on the thread that ran GlobalBA would keep packing stale point ids on another).
inline std::atomic<unsigned long long> &gather_topology_epoch() { static std::atomic<unsigned long long> e(0); return e; }
inline void gather_topology_changed() { gather_topology_epoch().fetch_add(1, std::memory_order_acq_rel); }
struct GatherCache {
    unsigned long long seen_epoch = 0;
    struct KfSeg { size_t n_obvpts; std::vector<size_t> n_obs; std::vector<std::vector<int32_t> > raw, ptid; std::vector<std::vector<double> > uv0; unsigned long long used; };
    struct TextSeg { std::vector<size_t> n_feat; std::vector<std::vector<int32_t> > raw; std::vector<std::vector<double> > uv, ref; unsigned long long used; };
    std::map<long long, KfSeg> kf; std::map<long long, TextSeg> text;
    unsigned long long tick; long long hits, misses;
    GatherCache() : tick(0), hits(0), misses(0) {}
    void invalidate() { kf.clear(); text.clear(); gather_topology_changed(); seen_epoch = gather_topology_epoch().load(std::memory_order_acquire); }      // (also tells every other thread's cache)
    // entries no call has used for `keep` calls leave (the window moved on)
    void begin_call(unsigned long long keep = 4) { tick++;
        { const unsigned long long e = gather_topology_epoch().load(std::memory_order_acquire); if (e != seen_epoch) { kf.clear(); text.clear(); seen_epoch = e; } }
        for (std::map<long long, KfSeg>::iterator it = kf.begin(); it != kf.end();) { if (it->second.used + keep < tick) kf.erase(it++); else ++it; }
        for (std::map<long long, TextSeg>::iterator it = text.begin(); it != text.end();) { if (it->second.used + keep < tick) text.erase(it++); else ++it; } }
    template <class KeyFrameT>
    const KfSeg &scene(const KeyFrameT &K, int n_levels) {
        KfSeg &S = kf[(long long)K.mnId];
        bool ok = S.n_obs.size() == (size_t)n_levels && S.n_obvpts == K.vObvPts.size();
        for (int l = 0; ok && l < n_levels; l++) ok = S.n_obs[(size_t)l] == ((size_t)l < K.vSceneObv2d.size() ? K.vSceneObv2d[(size_t)l].size() : 0);
        S.used = tick;
        if (ok) { hits++; return S; }
        misses++;
        S.n_obvpts = K.vObvPts.size(); S.n_obs.assign((size_t)n_levels, 0); S.raw.assign((size_t)n_levels, std::vector<int32_t>()); S.ptid = S.raw; S.uv0.assign((size_t)n_levels, std::vector<double>());
        for (int l = 0; l < n_levels; l++) {
            if ((size_t)l >= K.vSceneObv2d.size()) continue;
            const auto &obs = K.vSceneObv2d[(size_t)l];
            S.n_obs[(size_t)l] = obs.size(); S.raw[(size_t)l].reserve(obs.size()); S.ptid[(size_t)l].reserve(obs.size()); S.uv0[(size_t)l].reserve(2*obs.size());
            for (size_t s = 0; s < obs.size(); s++) {
                const int raw = obs[s]->IdxToRaw;
                S.raw[(size_t)l].push_back(raw); S.ptid[(size_t)l].push_back((int32_t)K.vObvPts[(size_t)raw]->pt->mnId);
                const auto uv0 = K.vSceneObv2d[0][(size_t)raw]->feature;
                S.uv0[(size_t)l].push_back(uv0(0)); S.uv0[(size_t)l].push_back(uv0(1));
            }
        }
        return S;
    }
    template <class MapTextT>
    const TextSeg &features(const MapTextT &Tx, int n_levels) {
        TextSeg &S = text[(long long)Tx.mnId];
        bool ok = S.n_feat.size() == (size_t)n_levels;
        for (int l = 0; ok && l < n_levels; l++) ok = S.n_feat[(size_t)l] == ((size_t)l < Tx.vRefFeature.size() ? Tx.vRefFeature[(size_t)l].size() : 0);
        S.used = tick;
        if (ok) { hits++; return S; }
        misses++;
        S.n_feat.assign((size_t)n_levels, 0); S.raw.assign((size_t)n_levels, std::vector<int32_t>()); S.uv.assign((size_t)n_levels, std::vector<double>()); S.ref = S.uv;
        for (int l = 0; l < n_levels; l++) {
            if ((size_t)l >= Tx.vRefFeature.size()) continue;
            const auto &feats = Tx.vRefFeature[(size_t)l];
            S.n_feat[(size_t)l] = feats.size();
            for (size_t f = 0; f < feats.size(); f++) {
                S.raw[(size_t)l].push_back((int32_t)feats[f]->IdxToRaw);
                S.uv[(size_t)l].push_back(feats[f]->u); S.uv[(size_t)l].push_back(feats[f]->v);
                for (int k = 0; k < TSBA_NTAP; k++) S.ref[(size_t)l].push_back(feats[f]->neighbourNInten[k]);
            }
        }
        return S;
    }
};

// ---- small pieces shared by every problem type -----------------------------------------------------------------------------
template <class T, class PoseHolder>
inline void push_pose(const PoseHolder &kf, std::vector<double> &pose) {                // optimizer.cc:84-90, :264-271
    double q[4]; T::quat_of(kf.mRcw, q);
    pose.push_back(q[0]); pose.push_back(q[1]); pose.push_back(q[2]); pose.push_back(q[3]);
    pose.push_back(kf.mtcw(0, 0)); pose.push_back(kf.mtcw(1, 0)); pose.push_back(kf.mtcw(2, 0));
}
template <class M44>
inline void push_mat34(const M44 &Tm, std::vector<double> &out) {                       // row-major 3x4 of a 4x4 transform
    for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) out.push_back(Tm(r, c));
}
inline void push_zero12(std::vector<double> &out) { out.insert(out.end(), 12, 0.0); }

// text features of one pyramid level, grouped by plane (mapText::vRefFeature[level], mapText.cc:64-107): centre, IdxToRaw, the 8
// normalised host intensities (neighbourNInten).  The 8 tap rays are not shipped: the library rebuilds them from the centre with the
// reference's own expression (tool.cc:1550-1566) -- 16 B instead of 192 B per feature.
template <class MapTextT>
inline void push_text_features(const std::vector<MapTextT *> &texts, int level, Packed &P, GatherCache *cache = nullptr, int n_levels = 0) {
    P.tfeat_off[level].assign(1, 0);
    for (size_t j = 0; j < texts.size(); j++) {
        if (cache) {                                                 // the plane's features of this level as three block copies
            const GatherCache::TextSeg &S = cache->features(*texts[j], n_levels);
            P.tfeat_raw[level].insert(P.tfeat_raw[level].end(), S.raw[(size_t)level].begin(), S.raw[(size_t)level].end());
            P.tfeat_uv[level].insert(P.tfeat_uv[level].end(), S.uv[(size_t)level].begin(), S.uv[(size_t)level].end());
            P.tfeat_ref[level].insert(P.tfeat_ref[level].end(), S.ref[(size_t)level].begin(), S.ref[(size_t)level].end());
        } else if ((size_t)level < texts[j]->vRefFeature.size()) {
            const auto &feats = texts[j]->vRefFeature[level];
            for (size_t f = 0; f < feats.size(); f++) {
                P.tfeat_raw[level].push_back((int32_t)feats[f]->IdxToRaw);
                P.tfeat_uv[level].push_back(feats[f]->u); P.tfeat_uv[level].push_back(feats[f]->v);
                for (int k = 0; k < TSBA_NTAP; k++) P.tfeat_ref[level].push_back(feats[f]->neighbourNInten[k]);
            }
        }
        P.tfeat_off[level].push_back((int32_t)P.tfeat_raw[level].size());
    }
}
template <class MapTextT>
inline void push_text_box(const MapTextT &obj, std::vector<double> &box) {              // vTextDeteRay: 4 corner rays (mapText.cc:87-90)
    for (int b = 0; b < 4; b++) { box.push_back(obj.vTextDeteRay[b](0)); box.push_back(obj.vTextDeteRay[b](1)); }
}

// ---- map-level problems: LocalBundleAdjustment / GlobalBA / OptimizeLandmarker ---------------------------------------------------
// mode 0 local (landmarks hosted outside vKFs are frozen: vMapPtOptim / vMapTextOptim = false, optimizer.cc:236-262)
// mode 1 global, mode 2 landmarker (every keyframe of the map; landmarker: kf_initial = all ones => every pose constant)
template <class T>
inline void pack_map(typename T::Map *mpMap, const std::vector<typename T::KeyFrame *> &vKFs,
                     const std::vector<typename T::MapPt *> &vMapPts, const std::vector<typename T::MapText *> &vMapTexts,
                     int mode, int n_levels, const double K[4], bool with_text, Packed &P,
                     std::vector<int> *mnId2Pts_out = nullptr, std::vector<int> *mnId2Texts_out = nullptr, GatherCache *cache = nullptr) {
    if (cache) cache->begin_call();
    std::vector<int> mnId2Pts((size_t)mpMap->imapPts, -1), mnId2Texts((size_t)mpMap->imapText, -1), mnId2KFs((size_t)mpMap->imapkfs, -1);
    for (size_t k = 0; k < vKFs.size(); k++) mnId2KFs[(size_t)vKFs[k]->mnId] = (int)k;                                 // :229-233
    {   // one allocation per array instead of a doubling chain (this runs once per keyframe, on the caller's thread)
        const size_t nk = vKFs.size(), np = vMapPts.size(), nt = vMapTexts.size();
        P.pose.reserve(7*nk); P.kf_initial.reserve(nk); P.kf_id.reserve(nk); P.kf_flag_off.reserve(nk);
        P.rho.reserve(np); P.pt_ray.reserve(2*np); P.pt_host.reserve(np); P.pt_Trw.reserve(12*np);
        P.theta.reserve(3*nt); P.text_host.reserve(nt); P.text_Twr.reserve(12*nt); P.text_box.reserve(8*nt);
        size_t nflag = 0; for (size_t k = 0; k < nk; k++) nflag += vKFs[k]->vObvGoodPts.size();
        P.sgood.reserve(nflag);
        for (int l = 0; l < n_levels; l++) { size_t n = 0; for (size_t k = 0; k < nk; k++) if ((size_t)l < vKFs[k]->vSceneObv2d.size()) n += vKFs[k]->vSceneObv2d[(size_t)l].size();
            P.sobs_kf[l].reserve(n); P.sobs_pt[l].reserve(n); P.sobs_flag[l].reserve(n); P.sobs_uv0[l].reserve(2*n);
            if (with_text) { size_t nf = 0; for (size_t j = 0; j < nt; j++) if ((size_t)l < vMapTexts[j]->vRefFeature.size()) nf += vMapTexts[j]->vRefFeature[(size_t)l].size();
                P.tfeat_off[l].reserve(nt + 1); P.tfeat_raw[l].reserve(nf); P.tfeat_uv[l].reserve(2*nf); P.tfeat_ref[l].reserve(TSBA_NTAP*nf); } }
    }
    // poses + gauge marks
    for (size_t k = 0; k < vKFs.size(); k++) {
        push_pose<T>(*vKFs[k], P.pose);
        P.kf_id.push_back((int64_t)vKFs[k]->mnId);
        P.kf_initial.push_back(mode == 2 ? 1 : (vKFs[k]->mnId == 0 || vKFs[k]->mnId == 1) ? 1 : 0);                   // :274-275
    }
    // scene points: rho, ray, host (or the frozen host's T_rw)
    {   // (arrays sized once and written by index: a push_back per value -- 17 per point -- was a third of this function's time)
        const size_t np = vMapPts.size();
        P.rho.resize(np); P.pt_ray.resize(2*np); P.pt_host.resize(np); P.pt_Trw.assign(12*np, 0.0);
        for (size_t j = 0; j < np; j++) {
            typename T::MapPt *pt = vMapPts[j];
            P.rho[j] = pt->GetInverD();
            const auto ray = pt->GetRaydir();
            P.pt_ray[2*j] = ray(0); P.pt_ray[2*j + 1] = ray(1);
            const int h = mnId2KFs[(size_t)pt->RefKF->mnId];
            P.pt_host[j] = h;
            if (h < 0) { const auto &Tm = pt->RefKF->mTcw; double *o = &P.pt_Trw[12*j];                                    // :1416-1417 (a host inside the window: zeros)
                for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) o[4*r + c] = Tm(r, c); }
            mnId2Pts[(size_t)pt->mnId] = (int)j;
        }
    }
    // text planes: theta = n/d in the host frame, host (or the frozen host's T_wr), detection box rays
    for (size_t j = 0; j < vMapTexts.size(); j++) {
        typename T::MapText *obj = vMapTexts[j];
        const auto N = obj->RefKF->mNcr[(size_t)obj->GetNidx()];
        P.theta.push_back(N(0, 0)); P.theta.push_back(N(1, 0)); P.theta.push_back(N(2, 0));
        const int h = mnId2KFs[(size_t)obj->RefKF->mnId];
        P.text_host.push_back(h);
        if (h < 0) push_mat34(obj->RefKF->mTwc, P.text_Twr); else push_zero12(P.text_Twr);                             // :1523-1524
        push_text_box(*obj, P.text_box);
        mnId2Texts[(size_t)obj->mnId] = (int)j;
    }
    // good flags of the scene observations: the keyframes' vObvGoodPts, concatenated
    {   size_t nflag = 0;
        for (size_t k = 0; k < vKFs.size(); k++) { P.kf_flag_off.push_back((int32_t)nflag); nflag += vKFs[k]->vObvGoodPts.size(); }
        P.sgood.resize(nflag);
        for (size_t k = 0; k < vKFs.size(); k++) { uint8_t *o = nflag ? &P.sgood[(size_t)P.kf_flag_off[k]] : nullptr; const auto &g = vKFs[k]->vObvGoodPts;
            for (size_t i = 0; i < g.size(); i++) o[i] = g[i] ? 1 : 0; }
    }
    // scene observations per level, in the reference's residual order: keyframe-major, then vSceneObv2d[level] (:1366-1435).
    // The residual always uses the LEVEL-0 pixel of the observation (SceneUse0Pyr, :1336,1402-1403).
    for (int l = 0; l < n_levels; l++)
        for (size_t k = 0; k < vKFs.size(); k++) {
            if ((size_t)l >= vKFs[k]->vSceneObv2d.size()) continue;
            if (cache) {                                               // the keyframe's cached segment: one lookup and four stores per observation
                const GatherCache::KfSeg &S = cache->scene(*vKFs[k], n_levels);
                const std::vector<int32_t> &raws = S.raw[(size_t)l], &ids = S.ptid[(size_t)l]; const std::vector<double> &uv = S.uv0[(size_t)l];
                const int32_t f0 = P.kf_flag_off[k];
                const size_t w0 = P.sobs_kf[l].size(), ns = raws.size(); size_t w = w0;
                P.sobs_kf[l].resize(w0 + ns, (int32_t)k); P.sobs_pt[l].resize(w0 + ns); P.sobs_flag[l].resize(w0 + ns); P.sobs_uv0[l].resize(2*(w0 + ns));
                int32_t *opt = P.sobs_pt[l].data(), *ofl = P.sobs_flag[l].data(); double *ouv = P.sobs_uv0[l].data();
                for (size_t s = 0; s < ns; s++) {
                    const int j = mnId2Pts[(size_t)ids[s]];
                    if (j < 0) continue;                               // (a point the map no longer lists)
                    opt[w] = j; ofl[w] = f0 + raws[s]; ouv[2*w] = uv[2*s]; ouv[2*w + 1] = uv[2*s + 1]; w++;
                }
                if (w != w0 + ns) { P.sobs_kf[l].resize(w); P.sobs_pt[l].resize(w); P.sobs_flag[l].resize(w); P.sobs_uv0[l].resize(2*w); }
                continue;
            }
            const auto &obs = vKFs[k]->vSceneObv2d[l];
            for (size_t s = 0; s < obs.size(); s++) {
                const int raw = obs[s]->IdxToRaw;
                const int j = mnId2Pts[(size_t)vKFs[k]->vObvPts[(size_t)raw]->pt->mnId];
                if (j < 0) continue;                                   // (a point the map no longer lists)
                P.sobs_kf[l].push_back((int32_t)k); P.sobs_pt[l].push_back(j); P.sobs_flag[l].push_back(P.kf_flag_off[k] + raw);
                const auto uv0 = vKFs[k]->vSceneObv2d[0][(size_t)raw]->feature;
                P.sobs_uv0[l].push_back(uv0(0)); P.sobs_uv0[l].push_back(uv0(1));
            }
        }
    int iw[TSBA_MAX_LEVELS] = {0, 0, 0, 0}, ih[TSBA_MAX_LEVELS] = {0, 0, 0, 0};
    if (with_text) {
        for (int l = 0; l < n_levels; l++) push_text_features(vMapTexts, l, P, cache, n_levels);
        // text observations, keyframe-major over GetStateTextObvs(TEXTGOOD) (:1462-1475): good flag of the observation and one flag
        // per LEVEL-0 feature of the plane (vObvGoodTextFeats[raw], indexed by IdxToRaw)
        P.tobs_fgood_off.assign(1, 0);
        for (size_t k = 0; k < vKFs.size(); k++) {
            std::vector<int> vNew2Raw;
            const auto vText = vKFs[k]->GetStateTextObvs(T::text_good(), vNew2Raw);
            for (size_t i = 0; i < vText.size(); i++) {
                const int raw = vNew2Raw[i], j = mnId2Texts[(size_t)vText[i]->obj->mnId];
                if (j < 0) continue;
                P.tobs_kf.push_back((int32_t)k); P.tobs_text.push_back(j); P.tobs_raw.push_back(raw);
                P.tobs_good.push_back(vKFs[k]->vObvGoodTexts[(size_t)raw] ? 1 : 0);
                const auto &fg = vKFs[k]->vObvGoodTextFeats[(size_t)raw];
                const size_t f0 = P.tfgood.size(); P.tfgood.resize(f0 + fg.size());
                for (size_t f = 0; f < fg.size(); f++) P.tfgood[f0 + f] = fg[f] ? 1 : 0;
                P.tobs_fgood_off.push_back((int32_t)P.tfgood.size());
            }
        }
        for (int l = 0; l < n_levels; l++) {
            for (size_t k = 0; k < vKFs.size(); k++) P.img[l].push_back(T::img(vKFs[k]->vFrameImg[(size_t)l]));
            if (!vKFs.empty()) { iw[l] = T::img_w(vKFs[0]->vFrameImg[(size_t)l]); ih[l] = T::img_h(vKFs[0]->vFrameImg[(size_t)l]); }
        }
    } else for (int l = 0; l < n_levels; l++) P.tfeat_off[l].assign(vMapTexts.size() + 1, 0);
    P.finish(n_levels, K, iw, ih);
    if (mnId2Pts_out) *mnId2Pts_out = mnId2Pts;
    if (mnId2Texts_out) *mnId2Texts_out = mnId2Texts;
}

// write-back of the map-level problems (optimizer.cc:292-326): poses (not for the landmarker), rho, theta, and the flags the outlier
// passes cleared (PyrBA writes vObvGoodPts / vObvGoodTexts / vObvGoodTextFeats in place, :1626-1686)
template <class T>
inline void scatter_map(const Packed &P, const std::vector<typename T::KeyFrame *> &vKFs, const std::vector<typename T::MapPt *> &vMapPts,
                        const std::vector<typename T::MapText *> &vMapTexts, bool poses, bool flags) {
    if (poses) for (size_t k = 0; k < vKFs.size(); k++) T::set_pose(*vKFs[k], &P.pose[7*k]);
    for (size_t j = 0; j < vMapPts.size(); j++) { double r = P.rho[j]; vMapPts[j]->SetRho(r); }
    for (size_t j = 0; j < vMapTexts.size(); j++) T::set_theta(*vMapTexts[j], &P.theta[3*j]);
    if (!flags) return;
    for (size_t k = 0; k < vKFs.size(); k++)
        for (size_t i = 0; i < vKFs[k]->vObvGoodPts.size(); i++) vKFs[k]->vObvGoodPts[i] = P.sgood[(size_t)P.kf_flag_off[k] + i] != 0;
    for (size_t t = 0; t < P.tobs_kf.size(); t++) {
        typename T::KeyFrame *kf = vKFs[(size_t)P.tobs_kf[t]]; const size_t raw = (size_t)P.tobs_raw[t];
        kf->vObvGoodTexts[raw] = P.tobs_good[t] != 0;
        for (size_t f = 0; f < kf->vObvGoodTextFeats[raw].size(); f++) kf->vObvGoodTextFeats[raw][f] = P.tfgood[(size_t)P.tobs_fgood_off[t] + f] != 0;
    }
}

// ---- PoseOptim: one frame, every landmark frozen in its host (optimizer.cc:135-172, PyrPoseOptim :1104-1207) -------------------------
// points = F.vObvPts in order (sobs_pt = IdxToRaw), planes = the TEXTGOOD entries of F.vObvText (FLAGTextObjs keeps their raw index)
template <class T>
inline void pack_pose(typename T::Frame &F, int n_levels, const double K[4], bool with_text, Packed &P) {
    push_pose<T>(F, P.pose);
    P.kf_initial.push_back(0);
    for (size_t j = 0; j < F.vObvPts.size(); j++) {
        typename T::MapPt *pt = F.vObvPts[j]->pt;
        const auto rr = pt->GetPtInv();                                   // (mx, my, rho), :1130
        P.pt_ray.push_back(rr(0)); P.pt_ray.push_back(rr(1)); P.rho.push_back(rr(2));
        P.pt_host.push_back(-1); push_mat34(pt->RefKF->mTcw, P.pt_Trw);   // Trw = RefKF->mTcw, :1131
    }
    P.kf_flag_off.push_back(0);
    for (size_t i = 0; i < F.vObvGoodPts.size(); i++) P.sgood.push_back(F.vObvGoodPts[i] ? 1 : 0);
    for (int l = 0; l < n_levels; l++) {
        if ((size_t)l >= F.vSceneObv2d.size()) continue;
        const auto &obs = F.vSceneObv2d[l];
        for (size_t s = 0; s < obs.size(); s++) {
            const int raw = obs[s]->IdxToRaw;
            P.sobs_kf[l].push_back(0); P.sobs_pt[l].push_back(raw); P.sobs_flag[l].push_back(raw);
            const auto uv0 = F.vSceneObv2d[0][(size_t)raw]->feature;
            P.sobs_uv0[l].push_back(uv0(0)); P.sobs_uv0[l].push_back(uv0(1));
        }
    }
    int iw[TSBA_MAX_LEVELS] = {0, 0, 0, 0}, ih[TSBA_MAX_LEVELS] = {0, 0, 0, 0};
    std::vector<typename T::MapText *> texts;
    P.tobs_fgood_off.assign(1, 0);
    if (with_text) for (size_t i = 0; i < F.vObvText.size(); i++) {
        typename T::MapText *obj = F.vObvText[i]->obj;
        if (obj->STATE != T::text_good()) continue;                      // :146-155
        const auto N = obj->RefKF->mNcr[(size_t)obj->GetNidx()];          // thetaFix, :1178
        P.theta.push_back(N(0, 0)); P.theta.push_back(N(1, 0)); P.theta.push_back(N(2, 0));
        P.text_host.push_back(-1); push_mat34(obj->RefKF->mTwc, P.text_Twr); push_text_box(*obj, P.text_box);
        P.tobs_kf.push_back(0); P.tobs_text.push_back((int32_t)texts.size()); P.tobs_raw.push_back((int32_t)i);
        P.tobs_good.push_back(F.vObvGoodTexts[i] ? 1 : 0);
        for (size_t f = 0; f < F.vObvGoodTextFeats[i].size(); f++) P.tfgood.push_back(F.vObvGoodTextFeats[i][f] ? 1 : 0);
        P.tobs_fgood_off.push_back((int32_t)P.tfgood.size());
        texts.push_back(obj);
    }
    for (int l = 0; l < n_levels; l++) {
        push_text_features(texts, l, P);
        if (with_text && !texts.empty()) { P.img[l].push_back(T::img(F.vFrameImg[(size_t)l])); iw[l] = T::img_w(F.vFrameImg[(size_t)l]); ih[l] = T::img_h(F.vFrameImg[(size_t)l]); }
    }
    P.finish(n_levels, K, iw, ih);
}
template <class T>
inline void scatter_pose(const Packed &P, typename T::Frame &F) {                        // :188-190 + the flags PyrPoseOptim cleared
    T::set_pose(F, &P.pose[0]);
    for (size_t i = 0; i < F.vObvGoodPts.size(); i++) F.vObvGoodPts[i] = P.sgood[i] != 0;
    for (size_t t = 0; t < P.tobs_kf.size(); t++) { const size_t raw = (size_t)P.tobs_raw[t];
        F.vObvGoodTexts[raw] = P.tobs_good[t] != 0;
        for (size_t f = 0; f < F.vObvGoodTextFeats[raw].size(); f++) F.vObvGoodTextFeats[raw][f] = P.tfgood[(size_t)P.tobs_fgood_off[t] + f] != 0; }
}

// ---- InitBA: two keyframes, F1 (identity, constant) hosts every landmark, F2 observes (optimizer.cc:57-104, PyrIniBA :978-1030) ----------
template <class T>
inline void pack_init(typename T::KeyFrame &F1, typename T::KeyFrame &F2, int n_levels, const double K[4], Packed &P) {
    push_pose<T>(F1, P.pose); push_pose<T>(F2, P.pose);
    P.kf_initial.push_back(1); P.kf_initial.push_back(0);
    for (size_t j = 0; j < F1.vObvPts.size(); j++) {
        typename T::MapPt *pt = F1.vObvPts[j]->pt;
        P.rho.push_back(pt->GetInverD());
        const auto ray = pt->GetRaydir(); P.pt_ray.push_back(ray(0)); P.pt_ray.push_back(ray(1));
        P.pt_host.push_back(0); push_zero12(P.pt_Trw);
    }
    P.kf_flag_off.push_back(0); P.kf_flag_off.push_back(0);
    P.sgood.assign(F1.vObvPts.size() > 0 ? F1.vObvPts.size() : 1, 1);      // InitBA has no flags: every observation takes part
    for (int l = 0; l < n_levels; l++) {
        if ((size_t)l >= F2.vSceneObv2d.size()) continue;
        const auto &obs = F2.vSceneObv2d[l];
        for (size_t s = 0; s < obs.size(); s++) {
            const int raw = obs[s]->IdxToRaw;
            P.sobs_kf[l].push_back(1); P.sobs_pt[l].push_back(raw); P.sobs_flag[l].push_back(raw);
            const auto uv0 = F2.vSceneObv2d[0][(size_t)raw]->feature;
            P.sobs_uv0[l].push_back(uv0(0)); P.sobs_uv0[l].push_back(uv0(1));
        }
    }
    std::vector<typename T::MapText *> texts;
    P.tobs_fgood_off.assign(1, 0);
    for (size_t i = 0; i < F1.vObvText.size(); i++) {
        typename T::MapText *obj = F1.vObvText[i]->obj;
        const auto N = F1.mNcr[(size_t)obj->GetNidx()];
        P.theta.push_back(N(0, 0)); P.theta.push_back(N(1, 0)); P.theta.push_back(N(2, 0));
        P.text_host.push_back(0); push_zero12(P.text_Twr); push_text_box(*obj, P.text_box);
        P.tobs_kf.push_back(1); P.tobs_text.push_back((int32_t)i); P.tobs_raw.push_back((int32_t)i); P.tobs_good.push_back(1);
        const size_t nf0 = obj->vRefFeature.empty() ? 0 : obj->vRefFeature[0].size();
        P.tfgood.insert(P.tfgood.end(), nf0, 1);
        P.tobs_fgood_off.push_back((int32_t)P.tfgood.size());
        texts.push_back(obj);
    }
    int iw[TSBA_MAX_LEVELS] = {0, 0, 0, 0}, ih[TSBA_MAX_LEVELS] = {0, 0, 0, 0};
    for (int l = 0; l < n_levels; l++) {
        push_text_features(texts, l, P);
        P.img[l].push_back(T::img(F1.vFrameImg[(size_t)l])); P.img[l].push_back(T::img(F2.vFrameImg[(size_t)l]));
        iw[l] = T::img_w(F2.vFrameImg[(size_t)l]); ih[l] = T::img_h(F2.vFrameImg[(size_t)l]);
    }
    P.finish(n_levels, K, iw, ih);
}
template <class T>
inline void scatter_init(const Packed &P, typename T::KeyFrame &F1, typename T::KeyFrame &F2) {          // :119-129
    T::set_pose(F2, &P.pose[7]);
    for (size_t j = 0; j < F1.vObvPts.size(); j++) { double r = P.rho[j]; F1.vObvPts[j]->pt->SetRho(r); }
    for (size_t i = 0; i < F1.vObvText.size(); i++) T::set_theta(*F1.vObvText[i]->obj, &P.theta[3*i]);
}

// ---- ThetaOptimMultiFs: one plane, its observing keyframes (without the host) + the current frame, all poses constant -------------------------
// (optimizer.cc:565-603).  The reference works with T_cr = T_cw T_rw^-1 per frame; the flat problem gets the same geometry from the host
// as keyframe 0 (constant) and the observers' own T_cw.
template <class T>
inline void pack_theta(const typename T::Frame &F, typename T::MapText &obj, int n_levels, const double K[4], Packed &P) {
    std::vector<const typename T::KeyFrame *> kfs;
    kfs.push_back(obj.RefKF);
    for (typename std::map<typename T::KeyFrame *, std::vector<int> >::const_iterator it = obj.vObvkeyframe.begin(); it != obj.vObvkeyframe.end(); ++it)
        if (it->first->mnId != obj.RefKF->mnId) kfs.push_back(it->first);                              // :580-583
    for (size_t k = 0; k < kfs.size(); k++) push_pose<T>(*kfs[k], P.pose);
    push_pose<T>(F, P.pose);                                                                             // the current frame, last
    const size_t n_kf = kfs.size() + 1;
    P.kf_initial.assign(n_kf, 1);
    const auto N = obj.RefKF->mNcr[(size_t)obj.GetNidx()];
    P.theta.push_back(N(0, 0)); P.theta.push_back(N(1, 0)); P.theta.push_back(N(2, 0));
    P.text_host.push_back(0); push_zero12(P.text_Twr); push_text_box(obj, P.text_box);
    std::vector<typename T::MapText *> texts(1, &obj);
    P.tobs_fgood_off.assign(1, 0);
    const size_t nf0 = obj.vRefFeature.empty() ? 0 : obj.vRefFeature[0].size();
    for (size_t k = 1; k < n_kf; k++) {
        P.tobs_kf.push_back((int32_t)k); P.tobs_text.push_back(0); P.tobs_raw.push_back(0); P.tobs_good.push_back(1);
        P.tfgood.insert(P.tfgood.end(), nf0, 1); P.tobs_fgood_off.push_back((int32_t)P.tfgood.size());
    }
    P.sgood.assign(1, 1);
    int iw[TSBA_MAX_LEVELS] = {0, 0, 0, 0}, ih[TSBA_MAX_LEVELS] = {0, 0, 0, 0};
    for (int l = 0; l < n_levels; l++) {
        push_text_features(texts, l, P);
        for (size_t k = 0; k < kfs.size(); k++) P.img[l].push_back(T::img(kfs[k]->vFrameImg[(size_t)l]));
        P.img[l].push_back(T::img(F.vFrameImg[(size_t)l]));
        iw[l] = T::img_w(F.vFrameImg[(size_t)l]); ih[l] = T::img_h(F.vFrameImg[(size_t)l]);
    }
    P.finish(n_levels, K, iw, ih);
}

}  // namespace tsba_adapter
#endif
