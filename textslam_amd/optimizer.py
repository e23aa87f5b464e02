"""Host-side mirror of TextSLAM's `optimizer` class over the C ABI of libtsba.so (include/tsba.h).

Method names, argument meaning and side effects follow src/optimizer.h:57-70 of the reference:
the caller owns the problem, the optimiser mutates poses / rho / theta and the good-flags in place.
All compute runs in the HIP library; there is no CPU fallback -- a missing library or device raises.
"""
import ctypes as C
import os
import numpy as np

from .abi import (TsbaProblem, TsbaOptions, TsbaReport, TsbaDebugOptions, BAProblem, options_local, options_pose, options_global,
                  options_init, options_landmarker, options_theta, STATE_LOCAL, STATE_NOTREACHWIN)

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBPATH = os.environ.get("TSBA_LIB", os.path.join(_HERE, "libtsba.so"))     # TSBA_LIB: instrumented build for diagnostics
_lib = None


class TsbaError(RuntimeError):
    pass


def load_library():
    """dlopen libtsba.so and declare the ABI of include/tsba.h. Raises if the HIP extension was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIBPATH):
        raise TsbaError(f"{_LIBPATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')")
    L = C.CDLL(_LIBPATH)
    vp, dp = C.c_void_p, C.POINTER(C.c_double)
    L.tsba_create.argtypes = [C.POINTER(vp), C.c_int]
    L.tsba_destroy.argtypes = [vp]
    L.tsba_last_error.argtypes = [vp]
    L.tsba_last_error.restype = C.c_char_p
    for name in ("tsba_local_ba", "tsba_pose_optim", "tsba_global_ba"):
        getattr(L, name).argtypes = [vp, C.POINTER(TsbaProblem), C.POINTER(TsbaOptions), C.POINTER(TsbaReport)]
    L.tsba_upload.argtypes = [vp, C.POINTER(TsbaProblem), C.POINTER(TsbaOptions)]
    L.tsba_solve.argtypes = [vp, C.POINTER(TsbaReport)]
    L.tsba_text_label_image.argtypes = [vp, C.c_int, C.c_int, C.POINTER(C.c_float)]
    L.tsba_download.argtypes = [vp, C.POINTER(TsbaProblem)]
    L.tsba_eval.argtypes = [vp, C.POINTER(TsbaProblem), C.POINTER(TsbaOptions), C.c_int, dp, dp, dp,
                            C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.tsba_time_linearize.argtypes = [vp, C.c_int, C.c_int, dp, dp]
    L.tsba_debug_reduced_system.argtypes = [vp, C.c_double, dp, dp, dp, C.POINTER(C.c_int32), dp]
    L.tsba_debug_reduced_band.argtypes = [vp, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp, dp, dp]
    L.tsba_debug_solver_info.argtypes = [vp, C.POINTER(C.c_int32), C.c_int]
    L.tsba_debug_pcg_stats.argtypes = [vp, C.POINTER(C.c_int32)]
    L.tsba_debug_img_cache_stats.argtypes = [vp, C.POINTER(C.c_int64)]
    L.tsba_debug_multi_solve.argtypes = [vp, C.c_int, dp, dp]
    L.tsba_debug_far_blocks.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), dp]
    L.tsba_debug_row_of_kf.argtypes = [vp, C.POINTER(C.c_int32)]
    L.tsba_debug_time_solve.argtypes = [vp, C.c_int, dp]
    ip32 = C.POINTER(C.c_int32)
    L.tsba_debug_reduced_blocks.argtypes = [vp, C.c_double, ip32, ip32, ip32, dp, dp, dp, dp]
    L.tsba_debug_lm_trace.argtypes = [vp, C.c_int, dp, C.c_int]
    L.tsba_comm_stats.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.tsba_comm_unique_id.argtypes = [vp, vp]
    L.tsba_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.tsba_local_group_create.argtypes = [C.c_int]
    L.tsba_local_group_create.restype = vp
    L.tsba_local_group_destroy.argtypes = [vp]
    L.tsba_local_group_destroy.restype = None
    L.tsba_comm_init_local.argtypes = [vp, vp, C.c_int, C.c_int]
    L.tsba_debug_set.argtypes = [vp, C.POINTER(TsbaDebugOptions)]
    L.tsba_theta_optim.argtypes = [vp, C.POINTER(TsbaProblem), C.POINTER(TsbaOptions), C.c_int, dp, C.POINTER(TsbaReport)]
    for name in ("tsba_default_options_local", "tsba_default_options_pose", "tsba_default_options_global",
                 "tsba_default_options_init", "tsba_default_options_landmarker", "tsba_default_options_theta"):
        getattr(L, name).argtypes = [C.POINTER(TsbaOptions)]
        getattr(L, name).restype = None
    _lib = L
    return L


ABI_VERSION = 5                                   # TSBA_ABI_VERSION of include/tsba.h
EXPORTED_SYMBOLS = [
    "tsba_default_options_local", "tsba_default_options_pose", "tsba_default_options_global",
    "tsba_abi_version", "tsba_create", "tsba_destroy", "tsba_last_error",
    "tsba_default_options_init", "tsba_default_options_landmarker", "tsba_default_options_theta",
    "tsba_local_ba", "tsba_pose_optim", "tsba_global_ba", "tsba_theta_optim", "tsba_text_label_image",
    "tsba_upload", "tsba_solve", "tsba_download", "tsba_eval", "tsba_time_linearize",
    "tsba_comm_unique_id", "tsba_comm_load", "tsba_comm_init", "tsba_comm_init_local", "tsba_local_group_create", "tsba_local_group_destroy",
    "tsba_debug_set", "tsba_debug_reduced_system",
]


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def local_group_create(world):
    g = load_library().tsba_local_group_create(world)
    if not g:
        raise TsbaError("tsba_local_group_create failed")
    return C.c_void_p(g)


def local_group_destroy(group):
    load_library().tsba_local_group_destroy(group)


class Optimizer:
    """Drop-in for the numeric core of TextSLAM's optimizer (one instance per calling thread / GPU)."""

    def __init__(self, device=0):
        self.lib = load_library()
        if self.lib.tsba_abi_version() != ABI_VERSION:           # (the ctypes mirrors in abi.py describe exactly one layout)
            raise TsbaError(f"libtsba.so has ABI version {self.lib.tsba_abi_version()}, abi.py mirrors {ABI_VERSION}")
        self.ctx = C.c_void_p()
        rc = self.lib.tsba_create(C.byref(self.ctx), device)
        if rc != 0:
            raise TsbaError(f"tsba_create(device={device}) failed with {rc}: no usable HIP device")
        self._resident = None

    def close(self):
        if self.ctx:
            self.lib.tsba_destroy(self.ctx)
            self.ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.tsba_last_error(self.ctx)
            raise TsbaError(f"{what} failed with {rc}: {msg.decode() if msg else ''}")

    # ---- optimizer:: replacements (src/optimizer.h:57-70)
    def LocalBundleAdjustment(self, prob: BAProblem, state=STATE_LOCAL, options: TsbaOptions = None):
        o = options or options_local(state)
        return self._one_shot(self.lib.tsba_local_ba, prob, o, "tsba_local_ba")

    def PoseOptim(self, prob: BAProblem, options: TsbaOptions = None):
        o = options or options_pose()
        return self._one_shot(self.lib.tsba_pose_optim, prob, o, "tsba_pose_optim")

    def GlobalBA(self, prob: BAProblem, options: TsbaOptions = None):
        o = options or options_global()
        return self._one_shot(self.lib.tsba_global_ba, prob, o, "tsba_global_ba")

    def InitBA(self, prob: BAProblem, options: TsbaOptions = None):
        """optimizer::InitBA(F1, F2): prob holds the two keyframes, kf_initial = [1, 0] keeps the host (identity) constant."""
        return self._one_shot(self.lib.tsba_local_ba, prob, options or options_init(), "tsba_local_ba(init)")

    def OptimizeLandmarker(self, prob: BAProblem, options: TsbaOptions = None):
        """optimizer::OptimizeLandmarker(map): rho / theta refinement with every pose constant (kf_initial = all ones)."""
        return self._one_shot(self.lib.tsba_local_ba, prob, options or options_landmarker(), "tsba_local_ba(landmarker)")

    def ThetaOptimMultiFs(self, prob: BAProblem, text: int = 0, options: TsbaOptions = None, cov0=None):
        """optimizer::ThetaOptimMultiFs(F, obj): returns (report, 3x3 covariance of theta[text]).  A singular information matrix is
        not an error (the reference keeps obj->Covariance as it was, optimizer.cc:2224-2241): report["cov_valid"] == 0 and the
        returned matrix is `cov0` (the caller's previous covariance, zeros if none was given)."""
        o = options or options_theta()
        s = prob.struct()
        rep = TsbaReport()
        cov = np.zeros(9) if cov0 is None else np.ascontiguousarray(cov0, np.float64).reshape(9).copy()
        self._check(self.lib.tsba_theta_optim(self.ctx, C.byref(s), C.byref(o), text, _dp(cov), C.byref(rep)), "tsba_theta_optim")
        return rep.as_dict(), cov.reshape(3, 3)

    def TextLabelImage(self, kf: int, level: int, shape):
        """Label image (float32 h x w, -1 = background) of keyframe kf for the state left by the last solve
        (the TextLabelImg of optimizer::ShowBAReproj_TextBox)."""
        out = np.zeros(shape, np.float32)
        self._check(self.lib.tsba_text_label_image(self.ctx, int(kf), int(level), out.ctypes.data_as(C.POINTER(C.c_float))), "tsba_text_label_image")
        return out

    def _one_shot(self, fn, prob, o, what):
        s = prob.struct()
        rep = TsbaReport()
        rc = fn(self.ctx, C.byref(s), C.byref(o), C.byref(rep))
        if rc != 0 and rc != -3:
            self._check(rc, what)
        return rep.as_dict()

    # ---- staged interface (problem resident in HBM)
    def upload(self, prob: BAProblem, options: TsbaOptions):
        s = prob.struct()
        self._check(self.lib.tsba_upload(self.ctx, C.byref(s), C.byref(options)), "tsba_upload")
        self._resident = prob

    def solve(self):
        rep = TsbaReport()
        self._check(self.lib.tsba_solve(self.ctx, C.byref(rep)), "tsba_solve")
        return rep.as_dict()

    def download(self, prob: BAProblem = None):
        prob = prob or self._resident
        s = prob.struct()
        self._check(self.lib.tsba_download(self.ctx, C.byref(s)), "tsba_download")
        return prob

    # ---- test hooks
    def evaluate(self, prob: BAProblem, options: TsbaOptions, level: int, jac=True):
        s = prob.struct()
        ns, nt = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.tsba_eval(self.ctx, C.byref(s), C.byref(options), level, None, None, None,
                                       C.byref(ns), C.byref(nt)), "tsba_eval(count)")
        n_s, n_t = ns.value, nt.value
        resid = np.zeros(2 * n_s + 8 * n_t)
        J = np.zeros(26 * n_s + 120 * n_t) if jac else None
        ms = np.zeros((max(prob.n_tobs, 1), 2))
        self._check(self.lib.tsba_eval(self.ctx, C.byref(s), C.byref(options), level, _dp(resid),
                                       _dp(J) if jac else None, _dp(ms), C.byref(ns), C.byref(nt)), "tsba_eval")
        out = {"resid": resid, "ns": n_s, "nt": n_t, "musigma": ms[:prob.n_tobs]}
        if jac:
            out["jac_scene"] = J[:26 * n_s].reshape(n_s, 2, 13)
            out["jac_text"] = J[26 * n_s:].reshape(n_t, 8, 15)
        return out

    def reduced_system(self, radius: float):
        """First linearisation of pass 0 of the uploaded problem: S (6n_kf x 6n_kf, identity rows for fixed poses), g, cost."""
        n = 6 * self._resident.n_kf
        S, g, dpv = np.zeros((n, n)), np.zeros(n), np.zeros(n)
        cost = C.c_double(0)
        free = np.zeros(self._resident.n_kf, np.int32)
        self._check(self.lib.tsba_debug_reduced_system(self.ctx, radius, _dp(S), _dp(g), C.byref(cost),
                                                       free.ctypes.data_as(C.POINTER(C.c_int32)), _dp(dpv)),
                    "tsba_debug_reduced_system")
        return {"S": S, "g": g, "cost": cost.value, "free": free, "dp": dpv}

    def solver_info(self):
        """Which kernel paths the uploaded problem takes (tsba_debug_solver_info)."""
        v = (C.c_int32 * 19)()
        self._check(self.lib.tsba_debug_solver_info(self.ctx, v, 19), "tsba_debug_solver_info")
        keys = ("lds_solver", "band_storage", "band_stream", "interiors", "sep_cr", "band_rows", "small_pairs", "pose_kernel", "large_map", "world", "rank",
                "n_pair", "n_sblock", "n_scene_candidates", "n_point_slots", "kf_reordered", "ring", "far_band_blocks", "far_blocks")
        return dict(zip(keys, [int(x) for x in v]))

    def far_blocks(self):
        """The 6x6 blocks outside the band (tsba_debug_far_blocks): keyframes a < b and values [n][6][6] (rows: keyframe a)."""
        n = self.solver_info()["far_blocks"]
        a, b, v = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros((n, 6, 6))
        ip = C.POINTER(C.c_int32)
        self._check(self.lib.tsba_debug_far_blocks(self.ctx, a.ctypes.data_as(ip), b.ctypes.data_as(ip), _dp(v)), "tsba_debug_far_blocks")
        return a, b, v

    def multi_solve(self, R, single=False):
        """M X = R with the band factor of the last solve (tsba_debug_multi_solve); R: [6 x free poses, T].  single: one column through the
        single-vector solve phase (what a conjugate-gradient iteration applies)."""
        R = np.ascontiguousarray(R, np.float64); X = np.zeros_like(R)
        assert not single or R.shape[1] == 1
        self._check(self.lib.tsba_debug_multi_solve(self.ctx, -1 if single else R.shape[1], _dp(R), _dp(X)), "tsba_debug_multi_solve")
        return X

    def img_cache_stats(self):
        """Plane cache (tsba_problem.kf_id): (keyframes found on the device, keyframes copied) over the context's lifetime."""
        v = (C.c_int64 * 2)()
        self._check(self.lib.tsba_debug_img_cache_stats(self.ctx, v), "tsba_debug_img_cache_stats")
        return int(v[0]), int(v[1])

    def pcg_stats(self):
        """Maps with long-range coupling: conjugate-gradient statistics of the last solve (tsba_debug_pcg_stats)."""
        v = (C.c_int32 * 4)()
        self._check(self.lib.tsba_debug_pcg_stats(self.ctx, v), "tsba_debug_pcg_stats")
        return dict(zip(("iterations", "systems", "max_iterations", "hit_cap"), [int(x) for x in v]))

    def reduced_band(self, radius: float):
        """Large maps: the reduced system of the first linearisation in LAPACK lower-band storage (scipy.linalg.solveh_banded,
        lower=True) over the compressed free-pose rows, with g and the pose step dp (by keyframe)."""
        n, bw = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.tsba_debug_reduced_band(self.ctx, radius, C.byref(n), C.byref(bw), None, None, None), "tsba_debug_reduced_band")
        ab, g, dpv = np.zeros((bw.value + 1, n.value)), np.zeros(n.value), np.zeros(6*self._resident.n_kf)
        self._check(self.lib.tsba_debug_reduced_band(self.ctx, radius, C.byref(n), C.byref(bw), _dp(ab), _dp(g), _dp(dpv)), "tsba_debug_reduced_band")
        free = np.zeros(self._resident.n_kf, np.int32)
        self._check(self.lib.tsba_debug_reduced_system(self.ctx, radius, None, None, None, free.ctypes.data_as(C.POINTER(C.c_int32)), None),
                    "tsba_debug_reduced_system")
        rowblk = np.zeros(self._resident.n_kf, np.int32)
        self._check(self.lib.tsba_debug_row_of_kf(self.ctx, rowblk.ctypes.data_as(C.POINTER(C.c_int32))), "tsba_debug_row_of_kf")
        # dp in the row order of S (the keyframe order unless the plan reordered the keyframes)
        dp_rows = np.zeros(n.value)
        for k in np.nonzero(rowblk >= 0)[0]:
            dp_rows[6*rowblk[k]:6*rowblk[k] + 6] = dpv[6*k:6*k + 6]
        return {"ab": ab, "g": g, "dp": dpv, "dp_rows": dp_rows, "n": n.value, "bw": bw.value, "free": free, "rowblk": rowblk}

    def reduced_blocks(self, radius: float):
        """Large maps, any co-visibility graph: the reduced system of the first linearisation as 6x6 blocks keyed by keyframe pairs
        (tsba_debug_reduced_blocks).  -> dict(blocks {(kf_hi, kf_lo): 6x6 with rows = kf_hi}, g [6 n_kf], dp [6 n_kf], cost)."""
        n = C.c_int32(0); ip = C.POINTER(C.c_int32)
        self._check(self.lib.tsba_debug_reduced_blocks(self.ctx, radius, C.byref(n), None, None, None, None, None, None), "tsba_debug_reduced_blocks")
        nk = self._resident.n_kf
        kr, kc, v = np.zeros(n.value, np.int32), np.zeros(n.value, np.int32), np.zeros((n.value, 6, 6))
        g, dpv, cost = np.zeros(6*nk), np.zeros(6*nk), C.c_double(0)
        self._check(self.lib.tsba_debug_reduced_blocks(self.ctx, radius, C.byref(n), kr.ctypes.data_as(ip), kc.ctypes.data_as(ip), _dp(v), _dp(g), _dp(dpv), C.byref(cost)),
                    "tsba_debug_reduced_blocks")
        blocks = {}
        for q in range(n.value):
            r, c_ = int(kr[q]), int(kc[q])
            key, b = ((r, c_), v[q]) if r >= c_ else ((c_, r), v[q].T)
            if key in blocks:
                blocks[key] = blocks[key] + b
            else:
                blocks[key] = b.copy()
        return {"blocks": blocks, "g": g, "dp": dpv, "cost": cost.value}

    def lm_trace(self, ps: int = 0, cap: int = 64):
        """Per LM trial of pass `ps` of the last solve: [trials][4] = candidate cost (NaN: invalid step), model cost change, radius after
        the decision, 1 accepted / 0 rejected / -1 invalid / 2 tolerance exit (tsba_debug_lm_trace)."""
        out = np.full((cap, 4), np.nan)
        n = self.lib.tsba_debug_lm_trace(self.ctx, int(ps), _dp(out), cap)
        if n < 0:
            self._check(n, "tsba_debug_lm_trace")
        return out[:n].copy()

    # ---- multi-GPU (global BA): one process per GPU, RCCL communicator owned by the library
    def comm_unique_id(self):
        buf = (C.c_char * 128)()
        self._check(self.lib.tsba_comm_unique_id(self.ctx, buf), "tsba_comm_unique_id")
        return bytes(buf)

    def comm_init(self, id128, rank, world):
        self._check(self.lib.tsba_comm_init(self.ctx, id128, rank, world), "tsba_comm_init")

    def comm_init_local(self, group, rank, world):
        """In-process communicator (test hook, include/tsba.h): `group` from local_group_create(world), one thread per rank."""
        self._check(self.lib.tsba_comm_init_local(self.ctx, group, rank, world), "tsba_comm_init_local")

    def debug_set(self, **kw):
        """tsba_debug_set: solver-path switches for tests / diagnostics (no keyword = production behaviour)."""
        d = TsbaDebugOptions()
        for k, v in kw.items():
            setattr(d, k, int(v))
        self._check(self.lib.tsba_debug_set(self.ctx, C.byref(d) if kw else None), "tsba_debug_set")

    def time_solve(self, n: int = 50):
        """Average ms of the reduced-system solve on the S, g of the last solve (tsba_debug_time_solve)."""
        ms = C.c_double(0)
        self._check(self.lib.tsba_debug_time_solve(self.ctx, n, C.byref(ms)), "tsba_debug_time_solve")
        return ms.value

    def exchange_bytes(self):
        """Communicator size and the bytes this rank handed to collectives per LM trial / linearisation / pass set-up."""
        r = C.c_int32(0); b = (C.c_int64 * 3)()
        self._check(self.lib.tsba_comm_stats(self.ctx, C.byref(r), b), "tsba_comm_stats")
        return {"ranks": r.value, "per_trial": int(b[0]), "per_linearisation": int(b[1]), "per_pass": int(b[2])}

    def time_linearize(self, level: int, n: int = 50):
        ms, nbytes = C.c_double(0), C.c_double(0)
        self._check(self.lib.tsba_time_linearize(self.ctx, level, n, C.byref(ms), C.byref(nbytes)), "tsba_time_linearize")
        return ms.value, nbytes.value
