// ORBextractor_tsorb.hpp -- TextSLAM's ORBextractor (src/ORBextractor.h:51-88) over libtsorb.so (include/tsorb.h).
// Lives in the TextSLAM tree: the class keeps its name, constructor, operator() and getters, frame::FeatExtraScene (frame.cc:328-331)
// and the two 3000-feature initialisation extractors (tracking.cc:38-39) call it unchanged.  Everything that does not need OpenCV is
// adapter/tsorb_extractor_core.hpp (compiled and tested by this repository: tests/cxx/orb_from_cxx.cpp); this file adds the cv::Mat /
// cv::KeyPoint conversions and is not compiled here.
#ifndef ORBEXTRACTOR_TSORB_HPP
#define ORBEXTRACTOR_TSORB_HPP
#include <iostream>
#include <vector>
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include "tsorb_extractor_core.hpp"

namespace TextSLAM {

class ORBextractor : public tsorb_adapter::ExtractorCore {
public:
    ORBextractor(int nfeatures_, float scaleFactor_, int nlevels_, int iniThFAST_, int minThFAST_)
        : tsorb_adapter::ExtractorCore(nfeatures_, scaleFactor_, nlevels_, iniThFAST_, minThFAST_, /*device*/0) {
        if (!ok()) { std::cerr << "tsorb_create failed (" << create_rc() << "): no usable HIP device" << std::endl; exit(-1); }
    }

    // ORBextractor.cc:1054-1116: keypoints (level-major, coordinates scaled back to level 0) + 32-byte descriptors
    void operator()(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors) {
        if (image.empty()) return;
        cv::Mat im = image.getMat();
        CV_Assert(im.type() == CV_8UC1);
        std::vector<float> kp; std::vector<uint8_t> d;
        const int n = extract(im.data, im.cols, im.rows, (int)im.step, kp, d);
        if (n <= 0) { keypoints.clear(); descriptors.release(); return; }
        keypoints.resize((size_t)n);
        for (int i = 0; i < n; i++) keypoints[(size_t)i] = cv::KeyPoint(kp[6*i], kp[6*i + 1], kp[6*i + 2], kp[6*i + 3], kp[6*i + 4], (int)kp[6*i + 5]);
        cv::Mat(n, 32, CV_8U, d.data()).copyTo(descriptors);
    }

    // the getters of ORBextractor.h:61-83 return by value there
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }
};

}  // namespace TextSLAM
#endif
