"""Host-side mirror of the reference's BA-pyramid / reference-feature construction on the MI355X (libtsframe.so, include/tsframe.h).

  Frame.GetPyrMat(img, iScaleLevels)                  frame::GetPyrMat          /root/reference/src/frame.cc:178-204
  Frame.GetPyramidPts(...) / GetPyramidPtsScene(...)  tool::GetPyramidPts       /root/reference/src/tool.cc:564-710, 862-980
  Frame.CalNormvec(level, uv, mu, std)                tool::CalNormvec          /root/reference/src/tool.cc:1342-1364 (GetNeighbour INTERVAL8)
  Frame.GetBoxAllPixs(level, vTextDete, mu, std, K)   tool::GetBoxAllPixs       /root/reference/src/tool.cc:1264-1337

No CPU fallback: without the HIP library / a GPU every call raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, "libtsframe.so")
EXPORTED_SYMBOLS = ["tsframe_create", "tsframe_destroy", "tsframe_last_error", "tsframe_set_image", "tsframe_level_size", "tsframe_level_ptr",
                    "tsframe_get_level", "tsframe_pyramid_pts", "tsframe_neighbours", "tsframe_box_pixels"]
IMG, GRAD, GRADX, GRADY = 0, 1, 2, 3


class FrameError(RuntimeError):
    pass


def _load():
    if not os.path.exists(_LIBPATH):
        raise FrameError("libtsframe.so is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); there is no CPU fallback")
    L = C.CDLL(_LIBPATH)
    vp, dp, ip, up = C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    L.tsframe_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.tsframe_destroy.argtypes = [vp]; L.tsframe_destroy.restype = None
    L.tsframe_last_error.argtypes = [vp]; L.tsframe_last_error.restype = C.c_char_p
    L.tsframe_set_image.argtypes = [vp, up, C.c_int, C.c_int, C.c_int]
    L.tsframe_level_size.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.tsframe_level_ptr.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    L.tsframe_get_level.argtypes = [vp, C.c_int, C.c_int, up]
    L.tsframe_pyramid_pts.argtypes = [vp, C.c_int, C.POINTER(C.c_float), C.c_int, dp, dp, ip, dp, dp, ip, dp, up]
    L.tsframe_neighbours.argtypes = [vp, C.c_int, dp, C.c_int, C.c_double, C.c_double, dp, dp, up]
    L.tsframe_box_pixels.argtypes = [vp, C.c_int, dp, C.c_double, C.c_double, C.c_int, ip, ip, ip, dp, dp]
    return L


def _dp(a): return a.ctypes.data_as(C.POINTER(C.c_double))
def _up(a): return a.ctypes.data_as(C.POINTER(C.c_uint8))


class Frame:
    """The image side of a `frame` / `keyframe`: pyramid and gradient planes resident in HBM."""

    def __init__(self, device: int = 0):
        self.lib = _load()
        self.ctx = C.c_void_p()
        rc = self.lib.tsframe_create(device, C.byref(self.ctx))
        if rc != 0:
            raise FrameError("tsframe_create failed (%d): no usable GPU %d" % (rc, device))
        self.n_levels = 0

    def __del__(self):
        if getattr(self, "ctx", None) and self.ctx:
            self.lib.tsframe_destroy(self.ctx); self.ctx = None

    def _check(self, rc, what):
        if rc != 0:
            raise FrameError("%s failed (%d): %s" % (what, rc, self.lib.tsframe_last_error(self.ctx).decode()))

    def GetPyrMat(self, img, iScaleLevels: int):
        img = np.ascontiguousarray(img, np.uint8)
        self._check(self.lib.tsframe_set_image(self.ctx, _up(img), img.shape[1], img.shape[0], iScaleLevels), "tsframe_set_image")
        self.n_levels = iScaleLevels

    def level_shape(self, level):
        w, h = C.c_int(0), C.c_int(0)
        self._check(self.lib.tsframe_level_size(self.ctx, level, C.byref(w), C.byref(h)), "tsframe_level_size")
        return h.value, w.value

    def level(self, level, which=IMG):
        out = np.zeros(self.level_shape(level), np.uint8)
        self._check(self.lib.tsframe_get_level(self.ctx, level, which, _up(out)), "tsframe_get_level")
        return out

    def level_device_ptr(self, level, which=IMG) -> int:
        p = C.c_void_p()
        self._check(self.lib.tsframe_level_ptr(self.ctx, level, which, C.byref(p)), "tsframe_level_ptr")
        return p.value

    def _pts(self, mode, xy, box, inv_scale):
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2); n = len(xy); nl = self.n_levels; cap = max(1, n*nl)
        inv = np.ascontiguousarray(inv_scale, np.float64); assert len(inv) == nl
        bx = np.ascontiguousarray(box, np.float64) if box is not None else None
        off = np.zeros(nl + 1, np.int32); u = np.zeros(cap); v = np.zeros(cap); idx = np.zeros(cap, np.int32); I = np.zeros(cap); inn = np.zeros(cap, np.uint8)
        self._check(self.lib.tsframe_pyramid_pts(self.ctx, mode, xy.ctypes.data_as(C.POINTER(C.c_float)), n, _dp(bx) if bx is not None else None, _dp(inv),
                                                 off.ctypes.data_as(C.POINTER(C.c_int32)), _dp(u), _dp(v), idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(I), _up(inn)),
                    "tsframe_pyramid_pts")
        m = int(off[nl])
        return {"level_off": off, "u": u[:m], "v": v[:m], "idx": idx[:m], "inten": I[:m], "in": inn[:m]}

    def GetPyramidPts(self, vObvRaw, PMin, PMax, vInvScalefactor):
        """Text features of one detection box: vObvRaw = keypoint (x, y) at level 0."""
        return self._pts(0, vObvRaw, [PMin[0], PMin[1], PMax[0], PMax[1]], vInvScalefactor)

    def GetPyramidPtsScene(self, vObvRaw, vInvScalefactor):
        return self._pts(1, vObvRaw, None, vInvScalefactor)

    def CalNormvec(self, level, uv, mu, std):
        """Returns (neighbourInten [n, 8], neighbourNInten [n, 8], IN [n]); std == 0 raises (the reference returns false)."""
        uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2); n = len(uv)
        I = np.zeros((n, 8)); N = np.zeros((n, 8)); inn = np.zeros(n, np.uint8)
        self._check(self.lib.tsframe_neighbours(self.ctx, level, _dp(uv), n, float(mu), float(std), _dp(I), _dp(N), _up(inn)), "tsframe_neighbours")
        return I, N, inn

    def GetBoxAllPixs(self, level, vTextDete, mu, std, K=None):
        """All pixels of the level image inside the detection quad vTextDete (4 x (x, y)); returns a dict with u, v (int32), featureInten,
        featureNInten and -- when K = (fx, fy, cx, cy) is given -- ray [n, 3]; entry i has IdxToRaw = i, level 0, IN = True."""
        quad = np.ascontiguousarray(vTextDete, np.float64).reshape(4, 2)
        ip = C.POINTER(C.c_int32); n = C.c_int32(0)
        self._check(self.lib.tsframe_box_pixels(self.ctx, level, _dp(quad), float(mu), float(std), 0, C.byref(n), None, None, None, None), "tsframe_box_pixels")
        m = n.value; cap = max(1, m)
        u = np.zeros(cap, np.int32); v = np.zeros(cap, np.int32); I = np.zeros(cap); N = np.zeros(cap)
        if m > 0:
            self._check(self.lib.tsframe_box_pixels(self.ctx, level, _dp(quad), float(mu), float(std), cap, C.byref(n), u.ctypes.data_as(ip), v.ctypes.data_as(ip),
                                                    _dp(I), _dp(N)), "tsframe_box_pixels")
        out = {"u": u[:m], "v": v[:m], "featureInten": I[:m], "featureNInten": N[:m]}
        if K is not None:
            fx, fy, cx, cy = K
            out["ray"] = np.stack([(out["u"] - cx)/fx, (out["v"] - cy)/fy, np.ones(m)], 1)
        return out
