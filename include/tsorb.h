/*
 * tsorb.h -- C ABI of the MI355X-native ORB front-end (the ORBextractor hot path of TextSLAM).
 *
 *   tsorb_create         <- ORBextractor::ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)   src/ORBextractor.cc:410-471
 *   tsorb_extract_batch  <- ORBextractor::operator()(image, mask, keypoints, descriptors)                      src/ORBextractor.cc:1054-1116
 *                           = ComputePyramid (:1118-1143) + ComputeKeyPointsOctTree (:766-854: per-cell cv::FAST 20 -> 7,
 *                             DistributeOctTree :540-764, IC_Angle :77-104) + GaussianBlur 7x7 sigma 2 + computeOrbDescriptor (:108-147)
 *
 * The reference extracts one frame per call on one CPU thread; this ABI takes a batch of frames (frame::FeatExtraScene calls
 * it once per frame, frame.cc:328-331 -- the adapter simply passes n = 1, or batches the two initialisation frames).
 * Keypoints come back in the reference's order (level-major, quadtree list order inside a level) as 6 floats
 * (x, y, size, angle, response, octave) = the cv::KeyPoint fields the reference fills; descriptors as 32 bytes each.
 * All pointers are HOST pointers owned by the caller.  Return 0 = OK, negative = error; never exits.
 */
#ifndef TSORB_H
#define TSORB_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TSORB_OK          0
#define TSORB_ERR_ARG    -1
#define TSORB_ERR_DEVICE -2
#define TSORB_MAX_LEVELS  8

int tsorb_create(void **ctx, int nfeatures, float scale_factor, int nlevels, int ini_th_fast, int min_th_fast, int device);
int tsorb_destroy(void *ctx);
const char *tsorb_last_error(void *ctx);

/* Getters of the reference class (ORBextractor.h:63-88): scale factors / per-level feature quota. */
int tsorb_get_levels(void *ctx);
int tsorb_get_scale_factors(void *ctx, float *sf /*[nlevels]*/, float *inv_sf /*[nlevels]*/);
int tsorb_get_features_per_level(void *ctx, int32_t *n /*[nlevels]*/);

/* imgs: n grayscale frames, each h rows of `stride` bytes (w <= stride).
 * kp [n][cap][6], desc [n][cap][32], count [n]: per frame at most cap keypoints (the reference returns <= ~nfeatures + a few). */
int tsorb_extract_batch(void *ctx, const uint8_t *imgs, int n, int w, int h, int stride,
                        float *kp, uint8_t *desc, int32_t *count, int cap);

/* Staged form for resident benchmarking: upload once, run the device pipeline any number of times, download. */
int tsorb_upload(void *ctx, const uint8_t *imgs, int n, int w, int h, int stride, int cap);
int tsorb_run(void *ctx);
int tsorb_download(void *ctx, float *kp, uint8_t *desc, int32_t *count);

/* Test hook: pyramid level `level` (with its 19-px BORDER_REFLECT_101 frame) of frame f after a run: (h_l + 38) x (w_l + 38) bytes;
 * blurred = 1 returns the 7x7 Gaussian-blurred level (no frame): h_l x w_l. */
int tsorb_debug_level(void *ctx, int frame, int level, int blurred, uint8_t *out, int32_t *w_out, int32_t *h_out);
/* Test / diagnostics hook: the shape of the FAST launch(es).  -1 (default): chosen by the batch size -- from 24 frames on, the levels whose cells fit a 40 x 40
 * tile run two waves per cell in a launch of their own, the others (and every level of a smaller batch) four waves per cell on a 72 x 72 tile; 0: every level through
 * the general instance whatever the batch size; 2: the split whatever the batch size; 1 / 3: the split with four waves / one wave per cell on the small tile (A/B runs).
 * The output does not depend on it (tests/test_gpu_orb.py).  Takes effect at the next upload. */
int tsorb_debug_fast_shape(void *ctx, int shape);
/* Test / diagnostics hook: how the pyramid is formed.  -1 (default): chosen by the batch size -- for a few frames a tile of a level is formed from a base level several
 * levels up inside one workgroup (k_pyramid_one: two launches, levels 0 .. 3 from the input image and levels 4 .. from level 3; the per-frame call of frame.cc:328-331 is
 * a chain of eight dependent launches otherwise), larger batches take a launch per level; 0: always a launch per level; 1: always the two launches; 2: every level from
 * the input image in ONE launch (when the geometry fits the kernel's buffers; a launch per level otherwise); 3: two levels per launch at any batch size (an experiment:
 * slower on a batch); 100 + s: the two launches split at level s (takes effect at the next upload); 200 / 201: a batch's orientation and blur as two launches / one (default).
 * Up to 5 frames take the few-frames plan (where it still beats the batch plan, which gained from the same session's work).  The output is the same bytes every way (tests/test_gpu_orb.py). */
int tsorb_debug_pyramid(void *ctx, int shape);
/* Test hook: the number of runs of this context in which a (frame, level) did not fit the LDS quadtree (more than 4096 candidates, 1024 nodes) and the serial pass
 * (k_octree_serial, then orientation and descriptors once more) was launched behind the first synchronisation. */
int tsorb_debug_fallbacks(void *ctx);

/* ---- Window / projection search: the step between the extractor and PoseOptim.
 *   tsorb_match_set_frame / _set_features  <- frame::AssignFeaturesToGrid + PosInGrid                       src/frame.cc:372-407
 *   tsorb_match_search                     <- frame::GetFeaturesInArea (src/frame.cc:415-468; keyframe.cc:217-256 with qlev = -1,-1)
 *                                             + tracking::DescriptorDistance (src/tracking.cc:2762-2778) + the best / second-best scan
 *                                             of tracking::SearchFrom3D / SearchFrom3DAdd / SearchFrom3DLocalTrack (:1109-1345)
 * The searched frame is frame `frame` of the resident batch (its keypoints and descriptors never leave the device) or an explicit
 * feature set (kp6 [n][6] = x,y,size,angle,response,octave as the extractor returns them; desc [n][32]).  min/max x/y are the
 * frame's mnMinX.. (frame.cc:115-125); the grid is FRAME_GRID_COLS x FRAME_GRID_ROWS = 64 x 48.
 * Per query (x, y, radius r, octave range qlev = {minLevel, maxLevel}, NULL = no level check, 32-byte descriptor):
 *   cand_idx / cand_dist [nq][max_cand]: the candidates in the reference's order with their Hamming distances (cand_cnt = how many
 *   there were, possibly > max_cand), best_idx / best_dist: the first minimum (strict <, as the reference's loop; -1 / INT_MAX if
 *   none), best_dist2: the runner-up distance.  Stateful variants (SearchForInitializ's running vMatchDist filter, the
 *   first-come claim of a feature) stay in the caller, on the candidate lists.  Output pointers may be NULL. */
int tsorb_match_set_frame(void *ctx, int frame, double min_x, double max_x, double min_y, double max_y);
int tsorb_match_set_features(void *ctx, const float *kp6, const uint8_t *desc, int n, double min_x, double max_x, double min_y, double max_y);
int tsorb_match_search(void *ctx, int nq, const float *qxy, const float *qr, const int32_t *qlev, const uint8_t *qdesc, int max_cand,
                       int32_t *cand_idx, int32_t *cand_dist, int32_t *cand_cnt, int32_t *best_idx, int32_t *best_dist, int32_t *best_dist2);

#ifdef __cplusplus
}
#endif
#endif
