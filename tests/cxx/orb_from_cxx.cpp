// TEST INFRASTRUCTURE: the ORBextractor class of the adapter (its OpenCV-free half, adapter/tsorb_extractor_core.hpp) driven from C++.
//
//   orb_from_cxx <dump.bin> <out.bin>
//     dump: "img" (u8, rows x cols), "wh" (i32: cols, rows, step), "args" (i32: nfeatures, nlevels, iniThFAST, minThFAST), "scale" (f64)
//     1. construct the extractor: the scale tables of the constructor (ORBextractor.cc:415-430) are written out whatever the device;
//     2. with a HIP device: one operator() call on the image; keypoints (6 floats each) and descriptors are written out.
//   exit code 0 = all of it, 3 = tables only (no device), anything else = failure.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dump_io.hpp"
#include "tsorb_extractor_core.hpp"

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s dump.bin out.bin\n", argv[0]); return 2; }
    Dump d; if (!read_dump(argv[1], d)) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    const uint8_t *img = U8(d, "img"); const int32_t *wh = I32(d, "wh"), *args = I32(d, "args"); const double *scale = F64(d, "scale");
    if (!img || !wh || !args || !scale || CNT(d, "img") < (size_t)wh[2]*wh[1]) { fprintf(stderr, "dump incomplete\n"); return 2; }
    tsorb_adapter::ExtractorCore ex(args[0], (float)scale[0], args[1], args[2], args[3]);
    FILE *f = fopen(argv[2], "wb"); if (!f) return 2;
    std::vector<double> tab;                               // (the record files hold f64 / i32 / u8: floats widened exactly)
    for (int pass = 0; pass < 4; pass++) { const std::vector<float> &v = pass == 0 ? ex.GetScaleFactors() : pass == 1 ? ex.GetInverseScaleFactors() : pass == 2 ? ex.GetScaleSigmaSquares() : ex.GetInverseScaleSigmaSquares();
        for (size_t i = 0; i < v.size(); i++) tab.push_back((double)v[i]); }
    int32_t lev = ex.GetLevels(); double sf = (double)ex.GetScaleFactor();
    put(f, "tables", 0, tab.data(), tab.size()); put(f, "levels", 1, &lev, 1); put(f, "scale_factor", 0, &sf, 1);
    printf("tables written: %d levels\n", lev);
    if (!ex.ok()) { fclose(f);
        if (ex.create_rc() == TSORB_ERR_DEVICE) { printf("no HIP device: tables only\n"); return 3; }
        fprintf(stderr, "tsorb_create: %d\n", ex.create_rc()); return 1; }
    std::vector<float> kp; std::vector<uint8_t> desc;
    const int n = ex.extract(img, wh[0], wh[1], wh[2], kp, desc);
    if (n < 0) { fclose(f); fprintf(stderr, "extract: %d (%s)\n", n, ex.last_error()); return 1; }
    std::vector<double> kpd(kp.begin(), kp.end()); int32_t n32 = n;
    put(f, "n", 1, &n32, 1); put(f, "kp", 0, kpd.data(), kpd.size()); put(f, "desc", 2, desc.data(), desc.size());
    fclose(f);
    printf("extract done: %d keypoints\n", n);
    return 0;
}
