/* tsframe.h -- C ABI of the BA-pyramid / reference-feature front-end (libtsframe.so, gfx950).  SURVEY.md 8f rank 3.
 *
 * Replaces, on the device and bit-exactly (integer image arithmetic; fp64 sampling without FMA contraction):
 *   frame::GetPyrMat                    /root/reference/src/frame.cc:178-204     -> tsframe_set_image
 *   tool::GetPyramidPts (text, scene)   /root/reference/src/tool.cc:564-710,862-980 -> tsframe_pyramid_pts
 *   tool::CalNormvec / GetNeighbour     /root/reference/src/tool.cc:1342-1364,1540-1566 (INTERVAL8) -> tsframe_neighbours
 *   tool::GetBoxAllPixs                 /root/reference/src/tool.cc:1264-1337     -> tsframe_box_pixels
 * The pyramid stays resident in HBM: tsframe_level_ptr hands the device pointers to the BA library, so the four levels of a
 * keyframe need no host round trip between GetPyrMat and the photometric residuals.
 * All functions return 0 on success, a negative TSFRAME_ERR_* otherwise; tsframe_last_error gives the text. */
#ifndef TSFRAME_H
#define TSFRAME_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define TSFRAME_OK 0
#define TSFRAME_ERR_ARG (-1)
#define TSFRAME_ERR_DEVICE (-2)
#define TSFRAME_ERR_STATE (-3)
#define TSFRAME_MAX_LEVELS 8

enum { TSFRAME_IMG = 0, TSFRAME_GRAD = 1, TSFRAME_GRADX = 2, TSFRAME_GRADY = 3 };

int tsframe_create(int device, void **ctx);
void tsframe_destroy(void *ctx);
const char *tsframe_last_error(void *ctx);

/* frame::GetPyrMat: level 0 = img (w x h, 8-bit), level l = cv::pyrDown(level l-1); per level cv::Sobel x / y (CV_8U) and their
 * addWeighted(.5, .5) blend.  The image is copied through a pinned staging buffer; everything else happens on the device. */
int tsframe_set_image(void *ctx, const uint8_t *img, int w, int h, int n_levels);
int tsframe_level_size(void *ctx, int level, int *w, int *h);
/* device pointer of a resident plane (which = TSFRAME_IMG / GRAD / GRADX / GRADY); valid until the next tsframe_set_image */
int tsframe_level_ptr(void *ctx, int level, int which, const uint8_t **dev);
int tsframe_get_level(void *ctx, int level, int which, uint8_t *out);

/* tool::GetPyramidPts.  mode 0: text features, grid over the detection box box = {PMin.x, PMin.y, PMax.x, PMax.y} (level-0 pixels);
 * mode 1: scene features, grid over the image (box ignored).  xy = n raw features (float x, y at level 0), inv_scale[n_levels].
 * Outputs are level-major, level l in [level_off[l], level_off[l+1]); capacity of every output array: n * n_levels.
 * u, v: level coordinates; idx: IdxToRaw; inten: bilinear intensity on the level image; in: the bilinear sample was inside. */
int tsframe_pyramid_pts(void *ctx, int mode, const float *xy, int n, const double *box, const double *inv_scale,
                        int32_t *level_off, double *u, double *v, int32_t *idx, double *inten, uint8_t *in);

/* tool::CalNormvec -> GetNeighbour(INTERVAL8) on the resident level image: for n features (uv, level coordinates) the 8 neighbour
 * intensities, raw and (I - mu) / sigma; in[j] = inside flag of the last tap (what the reference leaves in feat->IN).
 * sigma == 0: TSFRAME_ERR_ARG (the reference's CalNormvec returns false). */
int tsframe_neighbours(void *ctx, int level, const double *uv, int n, double mu, double sigma,
                       double *inten8, double *ninten8, uint8_t *in);

/* tool::GetBoxAllPixs (called for level 0 by mapText's constructor, mapText.cc:103): every pixel of the level image inside the filled
 * detection quad (4 corners x, y in level pixels; cv::Point truncation + cv::fillPoly scan conversion, boundary included), in row-major
 * order of the clamped bounding box; entry i is the TextFeature with IdxToRaw = i: u, v = pixel, inten = I(v, u), ninten = (I - mu) / sigma
 * (the ray ((u - cx) / fx, (v - cy) / fy, 1) is left to the caller).  *n_out = number of pixels; cap = capacity of the four output
 * arrays, cap == 0 only counts (outputs may be NULL); n_out > cap: TSFRAME_ERR_ARG with *n_out set. */
int tsframe_box_pixels(void *ctx, int level, const double *quad, double mu, double sigma, int cap, int32_t *n_out,
                       int32_t *u, int32_t *v, double *inten, double *ninten);

#ifdef __cplusplus
}
#endif
#endif
