// TEST INFRASTRUCTURE.  Plain structs with the SHAPE of TextSLAM's object graph -- the member names and signatures the optimizer
// touches (src/frame.h:88-141, src/keyframe.h:39-155, src/mapPts.h:30-68, src/mapText.h:32-90, src/map.h:31-67, src/setting.h:48-95)
// -- without Eigen / OpenCV, so that adapter/tsba_gather.hpp (the templates a TextSLAM maintainer compiles against the real types)
// is compiled and exercised by this repository's own test-suite, from the language the reference is written in.
#ifndef MOCK_TEXTSLAM_HPP
#define MOCK_TEXTSLAM_HPP
#include <cmath>
#include <cstdint>
#include <map>
#include <vector>

namespace mock {

struct Vec2 { double v[2]; double operator()(int i) const { return v[i]; } double &operator()(int i) { return v[i]; } };
struct Vec3 {                                   // stands in for Eigen Vec3 and Mat31: v(i) and v(i, 0)
    double v[3];
    double operator()(int i) const { return v[i]; } double &operator()(int i) { return v[i]; }
    double operator()(int i, int) const { return v[i]; } double &operator()(int i, int) { return v[i]; }
};
typedef Vec3 Mat31;
struct Mat33 { double m[9]; double operator()(int r, int c) const { return m[3*r + c]; } double &operator()(int r, int c) { return m[3*r + c]; } };
struct Mat44 { double m[16]; double operator()(int r, int c) const { return m[4*r + c]; } double &operator()(int r, int c) { return m[4*r + c]; }
               void setIdentity() { for (int i = 0; i < 16; i++) m[i] = (i % 5 == 0) ? 1.0 : 0.0; } };
struct Image { std::vector<uint8_t> data; int cols, rows; Image() : cols(0), rows(0) {} };       // cv::Mat CV_8UC1, continuous

enum TextStatus { TEXTGOOD = 0, TEXTIMMATURE = 1, TEXTBAD = 2 };
enum BAStatus { NOTREACHWIN = 0, LOCAL = 1, GLOBAL = 2 };

struct SceneFeature { double u, v; Vec2 feature; int level, IdxToRaw; };
struct TextFeature { double u, v; Vec2 feature; int level, IdxToRaw; std::vector<double> neighbourNInten; };
struct keyframe; struct mapPts; struct mapText;
struct SceneObservation { mapPts *pt; int idx; };
struct TextObservation { mapText *obj; std::vector<int> idx; double cos; };

struct frame {
    Mat44 mTcw, mTwc; Mat33 mRcw, mRwc; Mat31 mtcw, mtwc;
    std::vector<Mat31> mNcr;
    int iNTextObj;
    std::vector<Image> vFrameImg;
    std::vector<TextObservation *> vObvText;
    std::vector<SceneObservation *> vObvPts;
    std::vector<std::vector<SceneFeature *> > vSceneObv2d;
    std::vector<bool> vObvGoodPts, vObvGoodTexts;
    std::vector<std::vector<bool> > vObvGoodTextFeats;
    frame() : iNTextObj(0) {}
    void SetPose(const Mat44 &Tcw) {                                  // frame.cc: Tcw -> R, t and the inverse
        mTcw = Tcw;
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) { mRcw(r, c) = Tcw(r, c); mRwc(c, r) = Tcw(r, c); } mtcw(r) = Tcw(r, 3); }
        for (int r = 0; r < 3; r++) mtwc(r) = -(mRwc(r, 0)*mtcw(0) + mRwc(r, 1)*mtcw(1) + mRwc(r, 2)*mtcw(2));
        mTwc.setIdentity();
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) mTwc(r, c) = mRwc(r, c); mTwc(r, 3) = mtwc(r); }
    }
};
struct keyframe : frame {
    long unsigned int mnId;
    keyframe() : mnId(0) {}
    void SetN(Mat31 &N, int idx) { mNcr[(size_t)idx] = N; }
    std::vector<TextObservation *> GetStateTextObvs(const TextStatus &need, std::vector<int> &vNew2Raw);
};
struct mapPts {
    long unsigned int mnId; keyframe *RefKF; double rho; Vec3 ray;
    double GetInverD() { return rho; }
    Mat31 GetRaydir() { return ray; }
    Vec3 GetPtInv() { Vec3 r; r(0) = ray(0); r(1) = ray(1); r(2) = rho; return r; }
    void SetRho(double &r) { rho = r; }
};
struct mapText {
    TextStatus STATE; long unsigned int mnId; keyframe *RefKF; int nidx;
    std::vector<Vec2> vTextDeteRay;
    std::vector<std::vector<TextFeature *> > vRefFeature;
    std::map<keyframe *, std::vector<int> > vObvkeyframe;
    int GetNidx() { return nidx; }
};
inline std::vector<TextObservation *> keyframe::GetStateTextObvs(const TextStatus &need, std::vector<int> &vNew2Raw) {
    std::vector<TextObservation *> out; vNew2Raw.clear();
    for (size_t i = 0; i < vObvText.size(); i++) if (vObvText[i]->obj->STATE == need) { out.push_back(vObvText[i]); vNew2Raw.push_back((int)i); }
    return out;
}
struct map {
    std::vector<mapPts *> vMapPoints; std::vector<mapText *> vMapTextObjs; std::vector<keyframe *> vKeyframes;
    int imapPts, imapText, imapkfs;
    map() : imapPts(0), imapText(0), imapkfs(0) {}
    std::vector<mapPts *> GetAllMapPoints() { return vMapPoints; }
    std::vector<mapPts *> GetAllMapPoints(const bool &) { return vMapPoints; }
    std::vector<mapText *> GetAllMapTexts(const TextStatus &s) { std::vector<mapText *> o; for (size_t i = 0; i < vMapTextObjs.size(); i++) if (vMapTextObjs[i]->STATE == s) o.push_back(vMapTextObjs[i]); return o; }
    std::vector<keyframe *> GetAllKeyFrame() { return vKeyframes; }
};

// quaternion (w, x, y, z) -> rotation matrix, as Eigen::Quaterniond::toRotationMatrix
inline void quat_to_R(const double q[4], Mat33 &R) {
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double tx = 2*x, ty = 2*y, tz = 2*z, twx = tx*w, twy = ty*w, twz = tz*w, txx = tx*x, txy = ty*x, txz = tz*x, tyy = ty*y, tyz = tz*y, tzz = tz*z;
    R(0, 0) = 1 - (tyy + tzz); R(0, 1) = txy - twz; R(0, 2) = txz + twy;
    R(1, 0) = txy + twz; R(1, 1) = 1 - (txx + tzz); R(1, 2) = tyz - twx;
    R(2, 0) = txz - twy; R(2, 1) = tyz + twx; R(2, 2) = 1 - (txx + tyy);
}

// ---- loop closing: setting.h:129-171 (Sim3_loop), :173-189 (FeatureConvert), cv::KeyPoint::pt
struct Sim3_loop {
    double r[4]; Vec3 t; double s;                                    // r = (w, x, y, z)
    Sim3_loop() : s(1.0) { r[0] = 1; r[1] = r[2] = r[3] = 0; t(0) = t(1) = t(2) = 0; }
    static void qmul(const double a[4], const double b[4], double o[4]) {
        o[0] = a[0]*b[0] - a[1]*b[1] - a[2]*b[2] - a[3]*b[3]; o[1] = a[0]*b[1] + a[1]*b[0] + a[2]*b[3] - a[3]*b[2];
        o[2] = a[0]*b[2] - a[1]*b[3] + a[2]*b[0] + a[3]*b[1]; o[3] = a[0]*b[3] + a[1]*b[2] - a[2]*b[1] + a[3]*b[0]; }
    static Vec3 rot(const double q[4], const Vec3 &v) { Mat33 R; quat_to_R(q, R); Vec3 o; for (int i = 0; i < 3; i++) o(i) = R(i, 0)*v(0) + R(i, 1)*v(1) + R(i, 2)*v(2); return o; }
    Sim3_loop inverse() const { Sim3_loop o; o.r[0] = r[0]; o.r[1] = -r[1]; o.r[2] = -r[2]; o.r[3] = -r[3];
        Vec3 u; for (int i = 0; i < 3; i++) u(i) = (-1.0/s)*t(i); o.t = rot(o.r, u); o.s = 1.0/s; return o; }       // (r*, r*((-1/s) t), 1/s)
    Sim3_loop operator*(const Sim3_loop &b) const { Sim3_loop o; qmul(r, b.r, o.r); Vec3 u = rot(r, b.t); for (int i = 0; i < 3; i++) o.t(i) = s*u(i) + t(i); o.s = s*b.s; return o; }
};
struct KeyPoint { struct Pt { float x, y; } pt; };
struct FeatureConvert { Mat31 posWorld, posObv; int FlagTS; mapText *obj; mapPts *pt; keyframe *KF; int idx2d; KeyPoint obv2d; Vec2 obv2dPred; };

// The Traits adapter/tsba_gather.hpp asks for (adapter/textslam_traits.hpp is the same over the real types)
struct Traits {
    typedef mock::map Map; typedef mock::keyframe KeyFrame; typedef mock::frame Frame; typedef mock::mapPts MapPt; typedef mock::mapText MapText;
    static TextStatus text_good() { return TEXTGOOD; }
    // Eigen::Quaterniond(R).normalized(): the trace / largest-diagonal branches of Eigen's quaternion-from-matrix
    static void quat_of(const Mat33 &R, double q[4]) {
        double t = R(0, 0) + R(1, 1) + R(2, 2);
        if (t > 0) { t = std::sqrt(t + 1.0); q[0] = 0.5*t; t = 0.5/t; q[1] = (R(2, 1) - R(1, 2))*t; q[2] = (R(0, 2) - R(2, 0))*t; q[3] = (R(1, 0) - R(0, 1))*t; }
        else { int i = 0; if (R(1, 1) > R(0, 0)) i = 1; if (R(2, 2) > R(i, i)) i = 2; const int j = (i + 1) % 3, k = (j + 1) % 3;
            t = std::sqrt(R(i, i) - R(j, j) - R(k, k) + 1.0); q[1 + i] = 0.5*t; t = 0.5/t; q[0] = (R(k, j) - R(j, k))*t; q[1 + j] = (R(j, i) + R(i, j))*t; q[1 + k] = (R(k, i) + R(i, k))*t; }
        const double n = std::sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
        for (int a = 0; a < 4; a++) q[a] /= n;
    }
    template <class PoseHolder> static void set_pose(PoseHolder &kf, const double pose[7]) {       // optimizer.cc:292-312
        const double n = std::sqrt(pose[0]*pose[0] + pose[1]*pose[1] + pose[2]*pose[2] + pose[3]*pose[3]);
        const double q[4] = { pose[0]/n, pose[1]/n, pose[2]/n, pose[3]/n };
        Mat33 R; quat_to_R(q, R);
        Mat44 Tcw; Tcw.setIdentity();
        for (int r = 0; r < 3; r++) { for (int c = 0; c < 3; c++) Tcw(r, c) = R(r, c); Tcw(r, 3) = pose[4 + r]; }
        kf.SetPose(Tcw);
    }
    static void set_theta(mapText &obj, const double th[3]) { Mat31 N; N(0) = th[0]; N(1) = th[1]; N(2) = th[2]; obj.RefKF->SetN(N, obj.GetNidx()); }
    // loop closing (adapter/tsloop_gather.hpp)
    typedef mock::Sim3_loop Sim3;
    static Sim3 sim_make(const double q[4], const double t[3], double s) { Sim3 S; const double n = std::sqrt(q[0]*q[0] + q[1]*q[1] + q[2]*q[2] + q[3]*q[3]);
        for (int a = 0; a < 4; a++) S.r[a] = q[a]/n;
        for (int a = 0; a < 3; a++) S.t(a) = t[a];
        S.s = s; return S; }
    static Sim3 sim_of_pose(const Mat33 &R, const Mat31 &t, double s) { double q[4]; quat_of(R, q); const double tt[3] = { t(0), t(1), t(2) }; return sim_make(q, tt, s); }
    static void sim_get(const Sim3 &S, bool normalise, double out[8]) { const double n = normalise ? std::sqrt(S.r[0]*S.r[0] + S.r[1]*S.r[1] + S.r[2]*S.r[2] + S.r[3]*S.r[3]) : 1.0;
        for (int a = 0; a < 4; a++) out[a] = S.r[a]/n;
        for (int a = 0; a < 3; a++) out[4 + a] = S.t(a);
        out[7] = S.s; }
    static void set_pose_sim(keyframe &kf, const double pose[8]) {                                  // optimizer.cc:887-906: T = [R(q / |q|) | t / s]
        const double p7[7] = { pose[0], pose[1], pose[2], pose[3], pose[4]/pose[7], pose[5]/pose[7], pose[6]/pose[7] };
        set_pose(kf, p7); }
    static const uint8_t *img(const Image &im) { return im.data.data(); }
    static int img_w(const Image &im) { return im.cols; }
    static int img_h(const Image &im) { return im.rows; }
};

}  // namespace mock
#endif
