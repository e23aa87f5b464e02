"""Host-side mirror of the reference's loop-closure optimisers on the MI355X (libtsloop.so, include/tsloop.h).

  LoopOptimizer.OptimizeSim3(...)     optimizer::OptimizeSim3     /root/reference/src/optimizer.cc:626-731
  LoopOptimizer.OptimizeLoop(...)     optimizer::OptimizeLoop     /root/reference/src/optimizer.cc:733-957 (the solve; the map update stays with the caller)

No CPU fallback: without the HIP library / a GPU every call raises.
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.path.join(_HERE, "libtsloop.so")
EXPORTED_SYMBOLS = ["tsloop_default_options_sim3", "tsloop_default_options_loop", "tsloop_create", "tsloop_destroy", "tsloop_last_error",
                    "tsloop_optimize_sim3", "tsloop_optimize_loop"]


class TsloopOptions(C.Structure):
    _fields_ = [("max_it", C.c_int32), ("pad", C.c_int32), ("huber_delta", C.c_double), ("thresh_outlier", C.c_double),
                ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double), ("min_relative_decrease", C.c_double),
                ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
                ("min_diagonal", C.c_double), ("max_diagonal", C.c_double)]


class TsloopReport(C.Structure):
    _fields_ = [("iters", C.c_int32), ("accepted", C.c_int32), ("termination", C.c_int32), ("n_inlier", C.c_int32),
                ("cost0", C.c_double), ("cost1", C.c_double), ("t_ms", C.c_double)]


class TsloopSim3Problem(C.Structure):
    _fields_ = [("n", C.c_int32), ("pad", C.c_int32), ("P1", C.POINTER(C.c_double)), ("P2", C.POINTER(C.c_double)),
                ("uv1", C.POINTER(C.c_float)), ("uv2", C.POINTER(C.c_float)), ("inlier", C.POINTER(C.c_uint8)),
                ("K", C.c_double*4), ("sim", C.c_double*8)]


class TsloopGraphProblem(C.Structure):
    _fields_ = [("n_kf", C.c_int32), ("n_edge", C.c_int32), ("pose", C.POINTER(C.c_double)), ("fixed", C.POINTER(C.c_uint8)),
                ("edge_i", C.POINTER(C.c_int32)), ("edge_j", C.POINTER(C.c_int32)), ("meas", C.POINTER(C.c_double))]


def make_graph_problem(pose, fixed, edge_i, edge_j, meas):
    pose = np.ascontiguousarray(pose, np.float64).reshape(-1, 8).copy(); fixed = np.ascontiguousarray(fixed, np.uint8)
    ei = np.ascontiguousarray(edge_i, np.int32); ej = np.ascontiguousarray(edge_j, np.int32); meas = np.ascontiguousarray(meas, np.float64).reshape(-1, 8)
    assert len(fixed) == len(pose) and len(ei) == len(ej) == len(meas)
    p = TsloopGraphProblem()
    p.n_kf = len(pose); p.n_edge = len(ei)
    p.pose = pose.ctypes.data_as(C.POINTER(C.c_double)); p.fixed = fixed.ctypes.data_as(C.POINTER(C.c_uint8))
    p.edge_i = ei.ctypes.data_as(C.POINTER(C.c_int32)); p.edge_j = ej.ctypes.data_as(C.POINTER(C.c_int32)); p.meas = meas.ctypes.data_as(C.POINTER(C.c_double))
    return p, (fixed, ei, ej, meas), pose


class LoopError(RuntimeError):
    pass


def make_sim3_problem(P1, P2, uv1, uv2, inliers, sim, K):
    """Packs numpy arrays into the ABI struct; returns (struct, keep-alive tuple, inlier array)."""
    P1 = np.ascontiguousarray(P1, np.float64).reshape(-1, 3); P2 = np.ascontiguousarray(P2, np.float64).reshape(-1, 3)
    uv1 = np.ascontiguousarray(uv1, np.float32).reshape(-1, 2); uv2 = np.ascontiguousarray(uv2, np.float32).reshape(-1, 2)
    inl = np.ascontiguousarray(inliers, np.uint8).copy()
    assert len(P1) == len(P2) == len(uv1) == len(uv2) == len(inl)          # assert((int)vFeat1.size()==(int)vFeat2.size()), optimizer.cc:655
    p = TsloopSim3Problem()
    p.n = len(P1)
    p.P1 = P1.ctypes.data_as(C.POINTER(C.c_double)); p.P2 = P2.ctypes.data_as(C.POINTER(C.c_double))
    p.uv1 = uv1.ctypes.data_as(C.POINTER(C.c_float)); p.uv2 = uv2.ctypes.data_as(C.POINTER(C.c_float))
    p.inlier = inl.ctypes.data_as(C.POINTER(C.c_uint8))
    for k in range(4): p.K[k] = float(K[k])
    for k in range(8): p.sim[k] = float(sim[k])
    return p, (P1, P2, uv1, uv2), inl


def report_dict(r):
    return {k: getattr(r, k) for k, _ in TsloopReport._fields_}


class LoopOptimizer:
    def __init__(self, device: int = 0):
        if not os.path.exists(_LIBPATH):
            raise LoopError("libtsloop.so is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950); there is no CPU fallback")
        L = self.lib = C.CDLL(_LIBPATH)
        L.tsloop_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        L.tsloop_destroy.argtypes = [C.c_void_p]; L.tsloop_destroy.restype = None
        L.tsloop_last_error.argtypes = [C.c_void_p]; L.tsloop_last_error.restype = C.c_char_p
        L.tsloop_default_options_sim3.argtypes = [C.POINTER(TsloopOptions)]; L.tsloop_default_options_sim3.restype = None
        L.tsloop_optimize_sim3.argtypes = [C.c_void_p, C.POINTER(TsloopSim3Problem), C.POINTER(TsloopOptions), C.POINTER(TsloopReport)]
        L.tsloop_default_options_loop.argtypes = [C.POINTER(TsloopOptions)]; L.tsloop_default_options_loop.restype = None
        L.tsloop_optimize_loop.argtypes = [C.c_void_p, C.POINTER(TsloopGraphProblem), C.POINTER(TsloopOptions), C.POINTER(TsloopReport)]
        self.ctx = C.c_void_p()
        rc = L.tsloop_create(device, C.byref(self.ctx))
        if rc != 0:
            raise LoopError("tsloop_create failed (%d): no usable GPU %d" % (rc, device))

    def __del__(self):
        if getattr(self, "ctx", None) and self.ctx:
            self.lib.tsloop_destroy(self.ctx); self.ctx = None

    def default_options_sim3(self):
        o = TsloopOptions(); self.lib.tsloop_default_options_sim3(C.byref(o)); return o

    def OptimizeSim3(self, P1, uv1, P2, uv2, vbInliers, Sim12, K, options=None):
        """vFeat1 = (posObv P1, obv2d uv1), vFeat2 = (P2, uv2); Sim12 = (qw, qx, qy, qz, t, s).
        Returns (numInlier, Sim12 [8], vbInliers, report) -- the reference updates Sim12 / vbInliers in place and returns the count."""
        o = options or self.default_options_sim3()
        p, keep, inl = make_sim3_problem(P1, P2, uv1, uv2, vbInliers, Sim12, K)
        r = TsloopReport()
        rc = self.lib.tsloop_optimize_sim3(self.ctx, C.byref(p), C.byref(o), C.byref(r))
        if rc not in (0, -3):
            raise LoopError("tsloop_optimize_sim3 failed (%d): %s" % (rc, self.lib.tsloop_last_error(self.ctx).decode()))
        rep = report_dict(r); rep["status"] = rc
        return r.n_inlier, np.array(list(p.sim)), inl.astype(bool), rep

    def default_options_loop(self):
        o = TsloopOptions(); self.lib.tsloop_default_options_loop(C.byref(o)); return o

    def OptimizeLoop(self, pose, fixed, edge_i, edge_j, meas, options=None):
        """Sim3 pose graph: pose [n_kf, 8] initial (q | t | s), fixed [n_kf], connections (edge_i, edge_j, meas = Sji).
        Returns (corrected poses [n_kf, 8], report)."""
        o = options or self.default_options_loop()
        p, keep, x = make_graph_problem(pose, fixed, edge_i, edge_j, meas)
        r = TsloopReport()
        rc = self.lib.tsloop_optimize_loop(self.ctx, C.byref(p), C.byref(o), C.byref(r))
        if rc not in (0, -3):
            raise LoopError("tsloop_optimize_loop failed (%d): %s" % (rc, self.lib.tsloop_last_error(self.ctx).decode()))
        rep = report_dict(r); rep["status"] = rc
        return x, rep
