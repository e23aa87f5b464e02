// tsorb_extractor_core.hpp -- the OpenCV-free half of TextSLAM's ORBextractor over libtsorb.so (include/tsorb.h): the constructor's
// tables (src/ORBextractor.cc:410-430) and one operator() call as flat arrays.  adapter/ORBextractor_tsorb.hpp wraps it with the
// cv::Mat / cv::KeyPoint conversions inside the TextSLAM tree; this half is compiled and run by this repository's own tests
// (tests/cxx/orb_from_cxx.cpp): the class from the language the reference is written in.
#ifndef TSORB_EXTRACTOR_CORE_HPP
#define TSORB_EXTRACTOR_CORE_HPP
#include <cstdint>
#include <vector>
#include "tsorb.h"

namespace tsorb_adapter {

class ExtractorCore {
public:
    ExtractorCore(int nfeatures_, float scaleFactor_, int nlevels_, int iniThFAST_, int minThFAST_, int device = 0)
        : nfeatures(nfeatures_), scaleFactor(scaleFactor_), nlevels(nlevels_), iniThFAST(iniThFAST_), minThFAST(minThFAST_), ctx_(nullptr), create_rc_(0) {
        mvScaleFactor.resize(nlevels); mvLevelSigma2.resize(nlevels); mvInvScaleFactor.resize(nlevels); mvInvLevelSigma2.resize(nlevels);
        mvScaleFactor[0] = 1.0f; mvLevelSigma2[0] = 1.0f;                                      // ORBextractor.cc:415-423 (scaleFactor is a double member)
        for (int i = 1; i < nlevels; i++) { mvScaleFactor[i] = mvScaleFactor[i - 1]*scaleFactor; mvLevelSigma2[i] = mvScaleFactor[i]*mvScaleFactor[i]; }
        for (int i = 0; i < nlevels; i++) { mvInvScaleFactor[i] = 1.0f/mvScaleFactor[i]; mvInvLevelSigma2[i] = 1.0f/mvLevelSigma2[i]; }   // :425-430
        create_rc_ = tsorb_create(&ctx_, nfeatures, scaleFactor_, nlevels, iniThFAST, minThFAST, device);   // the quota per level, umax and the pattern live in the library
        if (create_rc_ != TSORB_OK) ctx_ = nullptr;
    }
    ~ExtractorCore() { if (ctx_) tsorb_destroy(ctx_); }
    ExtractorCore(const ExtractorCore &) = delete; ExtractorCore &operator=(const ExtractorCore &) = delete;

    bool ok() const { return ctx_ != nullptr; }
    int create_rc() const { return create_rc_; }           // TSORB_ERR_DEVICE: no usable GPU (there is no CPU path)
    int capacity() const { return nfeatures + 8*nlevels + 64; }          // the quadtree returns at most a few keypoints more than asked for per level

    // ORBextractor::operator() (ORBextractor.cc:1054-1116) on one CV_8UC1 image of `rows` rows of `step` bytes: the keypoints level-major with
    // their coordinates scaled back to level 0, as 6 floats (pt.x, pt.y, size, angle, response, octave) each, and their 32-byte descriptors.
    // Returns the number of keypoints, < 0 on error (tsorb_last_error()).
    int extract(const uint8_t *data, int cols, int rows, int step, std::vector<float> &kp6, std::vector<uint8_t> &desc) {
        if (!ctx_) return create_rc_ ? create_rc_ : TSORB_ERR_DEVICE;
        const int cap = capacity();
        kp6.resize(6*(size_t)cap); desc.resize(32*(size_t)cap); int32_t n = 0;
        const int rc = tsorb_extract_batch(ctx_, data, 1, cols, rows, step, kp6.data(), desc.data(), &n, cap);
        if (rc != TSORB_OK) { kp6.clear(); desc.clear(); return rc; }
        kp6.resize(6*(size_t)n); desc.resize(32*(size_t)n);
        return n;
    }
    const char *last_error() const { return ctx_ ? tsorb_last_error(ctx_) : "tsorb_create failed"; }

    int GetLevels() const { return nlevels; }                                                  // ORBextractor.h:61-83
    float GetScaleFactor() const { return (float)scaleFactor; }
    const std::vector<float> &GetScaleFactors() const { return mvScaleFactor; }
    const std::vector<float> &GetInverseScaleFactors() const { return mvInvScaleFactor; }
    const std::vector<float> &GetScaleSigmaSquares() const { return mvLevelSigma2; }
    const std::vector<float> &GetInverseScaleSigmaSquares() const { return mvInvLevelSigma2; }
    void *tsorb_context() { return ctx_; }                 // for the window search on the resident features (tsorb_match_*, include/tsorb.h)

protected:
    int nfeatures; double scaleFactor; int nlevels, iniThFAST, minThFAST;
    std::vector<float> mvScaleFactor, mvInvScaleFactor, mvLevelSigma2, mvInvLevelSigma2;
    void *ctx_; int create_rc_;
};

}  // namespace tsorb_adapter
#endif
