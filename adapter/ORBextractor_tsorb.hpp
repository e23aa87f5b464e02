// ORBextractor_tsorb.hpp -- TextSLAM's ORBextractor (src/ORBextractor.h:51-88) over libtsorb.so (include/tsorb.h).
// Lives in the TextSLAM tree: the class keeps its name, constructor, operator() and getters, frame::FeatExtraScene (frame.cc:328-331)
// and the two 3000-feature initialisation extractors (tracking.cc:38-39) call it unchanged.  Not compiled by this repository (OpenCV).
#ifndef ORBEXTRACTOR_TSORB_HPP
#define ORBEXTRACTOR_TSORB_HPP
#include <vector>
#include <opencv2/core/core.hpp>
#include <opencv2/features2d/features2d.hpp>
#include "tsorb.h"

namespace TextSLAM {

class ORBextractor {
public:
    ORBextractor(int nfeatures_, float scaleFactor_, int nlevels_, int iniThFAST_, int minThFAST_)
        : nfeatures(nfeatures_), scaleFactor(scaleFactor_), nlevels(nlevels_), iniThFAST(iniThFAST_), minThFAST(minThFAST_), ctx_(nullptr) {
        if (tsorb_create(&ctx_, nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, /*device*/0) != 0) { std::cerr << "tsorb_create failed" << std::endl; exit(-1); }
        mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;                                     // ORBextractor.cc:415-430
        for (int i = 1; i < nlevels; i++) { mvScaleFactor[i] = mvScaleFactor[i - 1]*scaleFactor; mvLevelSigma2[i] = mvScaleFactor[i]*mvScaleFactor[i]; }
        for (int i = 0; i < nlevels; i++) { mvInvScaleFactor[i] = 1.0f/mvScaleFactor[i]; mvInvLevelSigma2[i] = 1.0f/mvLevelSigma2[i]; }
    }
    ~ORBextractor() { if (ctx_) tsorb_destroy(ctx_); }

    // ORBextractor.cc:1054-1116: keypoints (level-major, coordinates scaled back to level 0) + 32-byte descriptors
    void operator()(cv::InputArray image, cv::InputArray /*mask*/, std::vector<cv::KeyPoint> &keypoints, cv::OutputArray descriptors) {
        if (image.empty()) return;
        cv::Mat im = image.getMat();
        CV_Assert(im.type() == CV_8UC1);
        const int cap = nfeatures + 8*nlevels + 64;
        std::vector<float> kp(6*(size_t)cap); cv::Mat d(cap, 32, CV_8U); int32_t n = 0;
        if (tsorb_extract_batch(ctx_, im.data, 1, im.cols, im.rows, (int)im.step, kp.data(), d.data, &n, cap) != 0) { keypoints.clear(); descriptors.release(); return; }
        keypoints.resize((size_t)n);
        for (int i = 0; i < n; i++) keypoints[(size_t)i] = cv::KeyPoint(kp[6*i], kp[6*i + 1], kp[6*i + 2], kp[6*i + 3], kp[6*i + 4], (int)kp[6*i + 5]);
        if (n == 0) descriptors.release(); else d.rowRange(0, n).copyTo(descriptors);
    }

    int inline GetLevels() { return nlevels; }
    float inline GetScaleFactor() { return scaleFactor; }
    std::vector<float> inline GetScaleFactors() { return mvScaleFactor; }
    std::vector<float> inline GetInverseScaleFactors() { return mvInvScaleFactor; }
    std::vector<float> inline GetScaleSigmaSquares() { return mvLevelSigma2; }
    std::vector<float> inline GetInverseScaleSigmaSquares() { return mvInvLevelSigma2; }

    void *tsorb_context() { return ctx_; }     // for the window search on the resident features (tsorb_match_*, include/tsorb.h)

protected:
    int nfeatures; double scaleFactor; int nlevels, iniThFAST, minThFAST;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    void *ctx_;
};

}  // namespace TextSLAM
#endif
