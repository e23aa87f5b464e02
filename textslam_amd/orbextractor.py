"""Host-side mirror of TextSLAM's ORBextractor (src/ORBextractor.h:48-88) over the C ABI of libtsorb.so (include/tsorb.h).

`ORBextractor(nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST)` and `__call__(image)` keep the reference's
constructor / operator() meaning; `extract_batch` is the batched form the GPU wants.  No CPU fallback."""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("TSORB_LIB", os.path.join(_HERE, "libtsorb.so"))     # TSORB_LIB: instrumented build for diagnostics
_lib = None

EXPORTED_SYMBOLS = ["tsorb_create", "tsorb_destroy", "tsorb_last_error", "tsorb_get_levels", "tsorb_get_scale_factors",
                    "tsorb_get_features_per_level", "tsorb_extract_batch", "tsorb_upload", "tsorb_run", "tsorb_download",
                    "tsorb_debug_level", "tsorb_debug_fast_shape", "tsorb_debug_pyramid", "tsorb_debug_fallbacks", "tsorb_match_set_frame", "tsorb_match_set_features", "tsorb_match_search"]


class TsorbError(RuntimeError):
    pass


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIBPATH):
            raise TsorbError(f"{_LIBPATH} not found: build the HIP extension first")
        L = C.CDLL(_LIBPATH)
        vp, up, fp, ip = C.c_void_p, C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_int32)
        L.tsorb_create.argtypes = [C.POINTER(vp), C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, C.c_int]
        L.tsorb_destroy.argtypes = [vp]
        L.tsorb_last_error.argtypes = [vp]; L.tsorb_last_error.restype = C.c_char_p
        L.tsorb_get_levels.argtypes = [vp]
        L.tsorb_get_scale_factors.argtypes = [vp, fp, fp]
        L.tsorb_get_features_per_level.argtypes = [vp, ip]
        L.tsorb_extract_batch.argtypes = [vp, up, C.c_int, C.c_int, C.c_int, C.c_int, fp, up, ip, C.c_int]
        L.tsorb_upload.argtypes = [vp, up, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]
        L.tsorb_run.argtypes = [vp]
        L.tsorb_download.argtypes = [vp, fp, up, ip]
        L.tsorb_debug_level.argtypes = [vp, C.c_int, C.c_int, C.c_int, up, ip, ip]
        L.tsorb_debug_fast_shape.argtypes = [vp, C.c_int]
        L.tsorb_debug_pyramid.argtypes = [vp, C.c_int]
        L.tsorb_debug_fallbacks.argtypes = [vp]
        L.tsorb_match_set_frame.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
        L.tsorb_match_set_features.argtypes = [vp, fp, up, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double]
        L.tsorb_match_search.argtypes = [vp, C.c_int, fp, fp, ip, up, C.c_int, ip, ip, ip, ip, ip, ip]
        _lib = L
    return _lib


class ORBextractor:
    def __init__(self, nfeatures=1000, scaleFactor=1.2, nlevels=8, iniThFAST=20, minThFAST=7, device=0):
        self.lib = load_library()
        self.ctx = C.c_void_p()
        rc = self.lib.tsorb_create(C.byref(self.ctx), nfeatures, scaleFactor, nlevels, iniThFAST, minThFAST, device)
        if rc != 0:
            raise TsorbError(f"tsorb_create failed with {rc}: no usable HIP device" if rc == -2 else f"tsorb_create failed with {rc}")
        self.nlevels, self.cap = nlevels, nfeatures + 8 * nlevels + 64
        self._shape = None

    def close(self):
        if self.ctx:
            self.lib.tsorb_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.tsorb_last_error(self.ctx)
            raise TsorbError(f"{what} failed with {rc}: {msg.decode() if msg else ''}")

    # getters of the reference class
    def GetLevels(self):
        return self.lib.tsorb_get_levels(self.ctx)

    def GetScaleFactors(self):
        sf = np.zeros(self.nlevels, np.float32)
        self._check(self.lib.tsorb_get_scale_factors(self.ctx, sf.ctypes.data_as(C.POINTER(C.c_float)), None), "tsorb_get_scale_factors")
        return sf

    def GetFeaturesPerLevel(self):
        n = np.zeros(self.nlevels, np.int32)
        self._check(self.lib.tsorb_get_features_per_level(self.ctx, n.ctypes.data_as(C.POINTER(C.c_int32))), "tsorb_get_features_per_level")
        return n

    def upload(self, imgs):
        imgs = np.ascontiguousarray(imgs, np.uint8)
        if imgs.ndim == 2:
            imgs = imgs[None]
        n, h, w = imgs.shape
        self._check(self.lib.tsorb_upload(self.ctx, imgs.ctypes.data_as(C.POINTER(C.c_uint8)), n, w, h, w, self.cap), "tsorb_upload")
        self._shape = (n, h, w)

    def run(self):
        self._check(self.lib.tsorb_run(self.ctx), "tsorb_run")

    def download(self):
        n = self._shape[0]
        kp = np.zeros((n, self.cap, 6), np.float32)
        desc = np.zeros((n, self.cap, 32), np.uint8)
        cnt = np.zeros(n, np.int32)
        self._check(self.lib.tsorb_download(self.ctx, kp.ctypes.data_as(C.POINTER(C.c_float)), desc.ctypes.data_as(C.POINTER(C.c_uint8)),
                                            cnt.ctypes.data_as(C.POINTER(C.c_int32))), "tsorb_download")
        return [(kp[i, :cnt[i]].copy(), desc[i, :cnt[i]].copy()) for i in range(n)]

    def extract_batch(self, imgs):
        """imgs [n, h, w] uint8 -> list of (keypoints [k, 6] = x, y, size, angle, response, octave ; descriptors [k, 32])."""
        self.upload(imgs)
        self.run()
        return self.download()

    def __call__(self, image, mask=None):
        """ORBextractor::operator()(image, mask, keypoints, descriptors) -- the mask is ignored, as in the reference."""
        return self.extract_batch(image)[0]

    # ---- window / projection search (frame::GetFeaturesInArea + tracking::DescriptorDistance)
    def match_set_frame(self, frame, bounds):
        """Search in frame `frame` of the resident batch (features stay on the device); bounds = (mnMinX, mnMaxX, mnMinY, mnMaxY)."""
        self._check(self.lib.tsorb_match_set_frame(self.ctx, int(frame), *[float(b) for b in bounds]), "tsorb_match_set_frame")

    def match_set_features(self, kp6, desc, bounds):
        kp6 = np.ascontiguousarray(kp6, np.float32); desc = np.ascontiguousarray(desc, np.uint8)
        self._check(self.lib.tsorb_match_set_features(self.ctx, kp6.ctypes.data_as(C.POINTER(C.c_float)), desc.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                      kp6.shape[0], *[float(b) for b in bounds]), "tsorb_match_set_features")

    def match_search(self, qxy, qr, qlev, qdesc, max_cand=64):
        qxy = np.ascontiguousarray(qxy, np.float32); qr = np.ascontiguousarray(qr, np.float32); qdesc = np.ascontiguousarray(qdesc, np.uint8)
        nq = qxy.shape[0]
        qlev_p = None if qlev is None else np.ascontiguousarray(qlev, np.int32)
        ci = np.full((nq, max_cand), -1, np.int32); cd = np.full((nq, max_cand), -1, np.int32)
        cc = np.zeros(nq, np.int32); bi = np.zeros(nq, np.int32); bd = np.zeros(nq, np.int32); bd2 = np.zeros(nq, np.int32)
        ip_ = lambda a: a.ctypes.data_as(C.POINTER(C.c_int32))
        self._check(self.lib.tsorb_match_search(self.ctx, nq, qxy.ctypes.data_as(C.POINTER(C.c_float)), qr.ctypes.data_as(C.POINTER(C.c_float)),
                                                None if qlev_p is None else ip_(qlev_p), qdesc.ctypes.data_as(C.POINTER(C.c_uint8)), max_cand,
                                                ip_(ci), ip_(cd), ip_(cc), ip_(bi), ip_(bd), ip_(bd2)), "tsorb_match_search")
        return dict(cand_idx=ci, cand_dist=cd, cand_cnt=cc, best_idx=bi, best_dist=bd, best_dist2=bd2)

    def debug_fast_shape(self, shape=-1):
        """Diagnostics: the shape of the FAST launches (include/tsorb.h); -1 = chosen by the batch size."""
        self._check(self.lib.tsorb_debug_fast_shape(self.ctx, int(shape)), "tsorb_debug_fast_shape")

    def debug_pyramid(self, shape=-1):
        """Diagnostics: how the pyramid is formed (include/tsorb.h): 0 a launch per level, 1 every level from the input image in one launch, -1 by the batch size."""
        self._check(self.lib.tsorb_debug_pyramid(self.ctx, int(shape)), "tsorb_debug_pyramid")

    def debug_fallbacks(self):
        """Runs of this extractor that took the serial quadtree pass (include/tsorb.h)."""
        return int(self.lib.tsorb_debug_fallbacks(self.ctx))

    def debug_level(self, frame, level, blurred=False):
        n, h, w = self._shape
        out = np.zeros((h + 38) * (w + 38), np.uint8)
        lw, lh = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.tsorb_debug_level(self.ctx, frame, level, int(blurred), out.ctypes.data_as(C.POINTER(C.c_uint8)),
                                               C.byref(lw), C.byref(lh)), "tsorb_debug_level")
        if blurred:
            return out[:lw.value * lh.value].reshape(lh.value, lw.value).copy()
        return out[:(lw.value + 38) * (lh.value + 38)].reshape(lh.value + 38, lw.value + 38).copy()


def synthetic_frame(seed, w=640, h=480):
    """Seeded test frame: flat rectangles, discs and noise (corners at every contrast and scale)."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 100.0)
    for _ in range(120):
        x0, y0 = rng.integers(0, w), rng.integers(0, h)
        ww, hh = rng.integers(8, 80), rng.integers(8, 60)
        img[y0:y0 + hh, x0:x0 + ww] = rng.uniform(20, 235)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(60):
        cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(3, 25)
        img[(xx - cx) ** 2 + (yy - cy) ** 2 < r * r] = rng.uniform(10, 245)
    img += rng.normal(0, 3, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)
