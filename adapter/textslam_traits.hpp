// textslam_traits.hpp -- the Traits of adapter/tsba_gather.hpp over the REAL TextSLAM types.  Lives in the TextSLAM tree (it needs
// TextSLAM's headers, Eigen and OpenCV: not compiled by this repository -- the same templates are compiled and run here against
// tests/cxx/mock_textslam.hpp, whose Traits has exactly these members).
#ifndef TEXTSLAM_TRAITS_HPP
#define TEXTSLAM_TRAITS_HPP
#include <Eigen/Geometry>
#include <opencv2/core/core.hpp>
#include <setting.h>
#include <frame.h>
#include <keyframe.h>
#include <mapPts.h>
#include <mapText.h>
#include <map.h>

namespace tsba_adapter {

struct TextSlamTraits {
    typedef TextSLAM::map Map; typedef TextSLAM::keyframe KeyFrame; typedef TextSLAM::frame Frame;
    typedef TextSLAM::mapPts MapPt; typedef TextSLAM::mapText MapText;
    static TextSLAM::TextStatus text_good() { return TextSLAM::TEXTGOOD; }
    // optimizer.cc:84-90 / :264-271: Eigen::Quaterniond q(Rcw); q = q.normalized();  pose = (w, x, y, z, t)
    static void quat_of(const TextSLAM::Mat33 &R, double q[4]) {
        Eigen::Quaterniond e(R); e = e.normalized();
        q[0] = e.w(); q[1] = e.x(); q[2] = e.y(); q[3] = e.z();
    }
    // optimizer.cc:292-312: normalise the quaternion, build Tcw, SetPose
    template <class PoseHolder> static void set_pose(PoseHolder &kf, const double pose[7]) {
        Eigen::Quaterniond qcw; qcw.w() = pose[0]; qcw.x() = pose[1]; qcw.y() = pose[2]; qcw.z() = pose[3];
        qcw = qcw.normalized();
        TextSLAM::Mat33 Rcw(qcw);
        TextSLAM::Mat44 Tcw; Tcw.setIdentity();
        Tcw.block<3, 3>(0, 0) = Rcw; Tcw.block<3, 1>(0, 3) = TextSLAM::Mat31(pose[4], pose[5], pose[6]);
        kf.SetPose(Tcw);
    }
    // optimizer.cc:321-325
    static void set_theta(MapText &obj, const double th[3]) { TextSLAM::Mat31 N(th[0], th[1], th[2]); obj.RefKF->SetN(N, obj.GetNidx()); }
    // ---- loop closing (adapter/tsloop_gather.hpp): the reference's own Sim3_loop (setting.h:129-171) does the algebra
    typedef TextSLAM::Sim3_loop Sim3;
    static Sim3 sim_make(const double q[4], const double t[3], double s) {
        Eigen::Quaterniond e(q[0], q[1], q[2], q[3]); e = e.normalized();
        return Sim3(e, Eigen::Vector3d(t[0], t[1], t[2]), s);
    }
    static Sim3 sim_of_pose(const TextSLAM::Mat33 &R, const TextSLAM::Mat31 &t, double s) { return Sim3(R, t, s); }     // optimizer.cc:794-796
    static void sim_get(const Sim3 &S, bool normalise, double out[8]) {
        Eigen::Quaterniond e = S.r; if (normalise) e = e.normalized();
        out[0] = e.w(); out[1] = e.x(); out[2] = e.y(); out[3] = e.z(); out[4] = S.t(0); out[5] = S.t(1); out[6] = S.t(2); out[7] = S.s;
    }
    // optimizer.cc:887-906: q normalised, T = [R(q) | t / s], SetPose(T)
    static void set_pose_sim(KeyFrame &kf, const double pose[8]) {
        const double p7[7] = { pose[0], pose[1], pose[2], pose[3], pose[4]/pose[7], pose[5]/pose[7], pose[6]/pose[7] };
        set_pose(kf, p7);
    }
    // nume_BAText.h:25: the cost functors index cv::Mat::data directly, i.e. continuous CV_8UC1 with step == cols
    static const uint8_t *img(const cv::Mat &im) { CV_Assert(im.type() == CV_8UC1 && im.isContinuous()); return im.data; }
    static int img_w(const cv::Mat &im) { return im.cols; }
    static int img_h(const cv::Mat &im) { return im.rows; }
};

}  // namespace tsba_adapter
#endif
