"""ctypes mirror of include/tsba.h (the C ABI of the BA / pose-optimisation back-end).

Holds only plain data definitions: the structs, a numpy-backed problem container and the
reference option sets.  The compute lives in the HIP library (textslam_amd/csrc -> libtsba.so).
"""
import ctypes as C
import math
import numpy as np

MAX_LEVELS = 4
NTAP = 8

STATE_NOTREACHWIN, STATE_LOCAL, STATE_GLOBAL = 0, 1, 2

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_up = C.POINTER(C.c_uint8)
c_upp = C.POINTER(c_up)


class TsbaProblem(C.Structure):
    _fields_ = [
        ("n_kf", C.c_int32), ("n_pt", C.c_int32), ("n_text", C.c_int32), ("n_levels", C.c_int32),
        ("K", C.c_double * 4),
        ("pose", c_dp), ("rho", c_dp), ("theta", c_dp),
        ("kf_initial", c_up),
        ("pt_ray", c_dp), ("pt_host", c_ip), ("pt_host_Trw", c_dp),
        ("text_host", c_ip), ("text_host_Twr", c_dp), ("text_box_ray", c_dp),
        ("n_sobs", C.c_int32 * MAX_LEVELS),
        ("sobs_kf", c_ip * MAX_LEVELS), ("sobs_pt", c_ip * MAX_LEVELS), ("sobs_flag", c_ip * MAX_LEVELS),
        ("sobs_uv0", c_dp * MAX_LEVELS),
        ("n_sgood", C.c_int32), ("sgood", c_up),
        ("n_tfeat", C.c_int32 * MAX_LEVELS),
        ("tfeat_off", c_ip * MAX_LEVELS), ("tfeat_raw", c_ip * MAX_LEVELS),
        ("tfeat_uv", c_dp * MAX_LEVELS), ("tfeat_ref", c_dp * MAX_LEVELS),
        ("n_tobs", C.c_int32),
        ("tobs_kf", c_ip), ("tobs_text", c_ip), ("tobs_good", c_up),
        ("tobs_fgood_off", c_ip), ("tfgood", c_up),
        ("img", c_upp * MAX_LEVELS),
        ("img_w", C.c_int32 * MAX_LEVELS), ("img_h", C.c_int32 * MAX_LEVELS),
        ("kf_id", C.POINTER(C.c_int64)),
    ]


class TsbaOptions(C.Structure):
    _fields_ = [
        ("w_sx", C.c_double), ("w_sy", C.c_double), ("w_t", C.c_double),
        ("huber_scene", C.c_double), ("huber_text", C.c_double),
        ("n_passes", C.c_int32),
        ("levels", C.c_int32 * MAX_LEVELS), ("its", C.c_int32 * MAX_LEVELS),
        ("chi2_mono", C.c_double * MAX_LEVELS), ("chi2_text", C.c_double * MAX_LEVELS),
        ("text_bad_ratio", C.c_double),
        ("state", C.c_int32), ("outlier_scene", C.c_int32), ("outlier_text", C.c_int32),
        ("use_text", C.c_int32), ("filter_good", C.c_int32), ("text_jacobian", C.c_int32),
        ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double), ("parameter_tolerance", C.c_double),
        ("min_diagonal", C.c_double), ("max_diagonal", C.c_double),
        ("lm_shard", C.c_int32), ("lm_nshard", C.c_int32),
        ("img_on_device", C.c_int32), ("host_plan_pin", C.c_int32),
    ]


class TsbaReport(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("n_passes", C.c_int32),
        ("iters", C.c_int32 * MAX_LEVELS), ("accepted", C.c_int32 * MAX_LEVELS), ("termination", C.c_int32 * MAX_LEVELS),
        ("cost0", C.c_double * MAX_LEVELS), ("cost1", C.c_double * MAX_LEVELS),
        ("n_sblock", C.c_int64 * MAX_LEVELS), ("n_tblock", C.c_int64 * MAX_LEVELS),
        ("n_resid_evals", C.c_int64),
        ("n_bad_scene", C.c_int32 * MAX_LEVELS), ("n_bad_tfeat", C.c_int32 * MAX_LEVELS), ("n_bad_text", C.c_int32 * MAX_LEVELS),
        ("t_upload_ms", C.c_double), ("t_solve_ms", C.c_double), ("t_download_ms", C.c_double),
        ("cov_valid", C.c_int32), ("solver_path", C.c_int32),
        ("pcg_iterations", C.c_int32), ("pcg_systems", C.c_int32), ("pcg_max_iterations", C.c_int32), ("pcg_unconverged", C.c_int32), ("pcg_stagnated", C.c_int32),
        ("poll_timeouts", C.c_int32), ("reserved_", C.c_int32 * 2),
    ]

    def as_dict(self):
        n = self.n_passes
        d = {"status": self.status, "n_passes": n, "n_resid_evals": self.n_resid_evals, "cov_valid": self.cov_valid,
             "t_upload_ms": self.t_upload_ms, "t_solve_ms": self.t_solve_ms, "t_download_ms": self.t_download_ms,
             "solver_path": self.solver_path, "pcg_iterations": self.pcg_iterations, "pcg_systems": self.pcg_systems,
             "pcg_max_iterations": self.pcg_max_iterations, "pcg_unconverged": self.pcg_unconverged, "pcg_stagnated": self.pcg_stagnated,
             "poll_timeouts": self.poll_timeouts}
        for k in ("iters", "accepted", "termination", "cost0", "cost1", "n_sblock", "n_tblock",
                  "n_bad_scene", "n_bad_tfeat", "n_bad_text"):
            d[k] = list(getattr(self, k))[:n]
        return d


class TsbaDebugOptions(C.Structure):
    """tsba_debug_options (include/tsba_debug.h): test / diagnostics switches of one context; all zero = production behaviour."""
    _fields_ = [
        ("band_parts", C.c_int32), ("sep_solver", C.c_int32), ("no_band_stream", C.c_int32), ("no_pose_kernel", C.c_int32),
        ("no_small_pairs", C.c_int32), ("verbose", C.c_int32), ("no_kf_reorder", C.c_int32), ("no_schur_quad", C.c_int32), ("no_ring", C.c_int32),
        ("far_solver", C.c_int32), ("pcg_max_it", C.c_int32), ("pcg_tol_exp", C.c_int32), ("pcg_refactor", C.c_int32), ("pcg_block", C.c_int32), ("solve_variant", C.c_int32), ("sv_per_level", C.c_int32), ("host_pair_lists", C.c_int32),
        ("pass_launches", C.c_int32), ("trial_launches", C.c_int32), ("assume_cus", C.c_int32), ("lds_poison", C.c_int32),
    ]


def _lm_defaults(o):
    # Ceres 1.x Solver::Options defaults
    o.initial_radius, o.max_radius, o.min_radius = 1e4, 1e16, 1e-32
    o.min_relative_decrease = 1e-3
    o.function_tolerance, o.gradient_tolerance, o.parameter_tolerance = 1e-6, 1e-10, 1e-8
    o.min_diagonal, o.max_diagonal = 1e-6, 1e32
    o.lm_shard, o.lm_nshard = 0, 1


def options_local(state=STATE_LOCAL):
    """optimizer::LocalBundleAdjustment constants, src/optimizer.cc:282-289,1350-1351,1369,1454."""
    o = TsbaOptions()
    _lm_defaults(o)
    o.w_sx = o.w_sy = 1.0 / 1.2
    o.w_t = 1.0 / 0.2
    o.huber_scene, o.huber_text = math.sqrt(5.991), 3.0
    o.n_passes = 3
    for i in range(3):
        o.levels[i], o.its[i], o.chi2_mono[i] = 2 - i, 10, 12.25
        o.chi2_text[i] = 0.95 if i == 2 else 0.5
    o.text_bad_ratio = 0.99
    o.state = state
    o.outlier_scene = o.outlier_text = 1
    o.use_text, o.filter_good, o.text_jacobian = 1, 1, 0
    return o


def options_pose():
    """optimizer::PoseOptim constants, src/optimizer.cc:174-186 (same weights, no gauge)."""
    return options_local(STATE_NOTREACHWIN)


def options_global():
    """optimizer::GlobalBA constants, src/optimizer.cc:411-414,1707,1724,1751 (scene only, unweighted, 20 its)."""
    o = TsbaOptions()
    _lm_defaults(o)
    o.w_sx = o.w_sy = o.w_t = 1.0
    o.huber_scene, o.huber_text = math.sqrt(5.991), 3.0
    o.n_passes = 1
    o.levels[0], o.its[0], o.chi2_mono[0], o.chi2_text[0] = 0, 20, 18.0, 0.5
    o.text_bad_ratio = 0.99
    o.state = STATE_GLOBAL
    o.use_text, o.filter_good = 0, 0
    return o


def options_init():
    """optimizer::InitBA / PyrIniBA, src/optimizer.cc:960-1056 (mark kf_initial = [1, 0]: the host at identity is constant)."""
    o = options_local(STATE_NOTREACHWIN)
    o.w_sx = o.w_sy = o.w_t = 1.0
    o.huber_scene = o.huber_text = 3.0
    o.n_passes = 4
    for i in range(4):
        o.levels[i], o.its[i] = 3 - i, 10
    o.outlier_scene = o.outlier_text = 0
    o.filter_good = 0
    return o


def options_landmarker():
    """optimizer::OptimizeLandmarker / PyrLandmarkers, src/optimizer.cc:531-541,1861,1873,1922 (every KF constant)."""
    o = options_local(STATE_NOTREACHWIN)
    o.w_sx = o.w_sy = o.w_t = 1.0
    o.huber_scene, o.huber_text = math.sqrt(5.991), 2.0
    o.n_passes = 4
    for i in range(4):
        o.levels[i], o.its[i], o.chi2_mono[i], o.chi2_text[i] = 3 - i, 50, 18.0, 1.5
    o.outlier_scene, o.outlier_text = 1, 0
    return o


def options_theta():
    """optimizer::ThetaOptimMultiFs / PyrThetaOptim, src/optimizer.cc:610-615,2176,2203-2209 (no loss, 50 iterations)."""
    o = options_local(STATE_NOTREACHWIN)
    o.w_sx = o.w_sy = o.w_t = 1.0
    o.huber_text = 1e300
    for i in range(3):
        o.its[i] = 50
    o.outlier_scene = o.outlier_text = 0
    o.filter_good = 0
    return o


def _ptr(a, ctype):
    if a is None or a.size == 0:
        return C.cast(None, C.POINTER(ctype))
    return a.ctypes.data_as(C.POINTER(ctype))


class BAProblem:
    """Numpy-backed flat BA problem; `struct()` returns a TsbaProblem view (arrays are not copied)."""

    F64 = ("pose", "rho", "theta", "pt_ray", "pt_host_Trw", "text_host_Twr", "text_box_ray")
    I32 = ("pt_host", "text_host", "tobs_kf", "tobs_text", "tobs_fgood_off")
    U8 = ("kf_initial", "sgood", "tobs_good", "tfgood")

    def __init__(self):
        self.K = np.zeros(4)
        self.n_levels = 1
        for k in self.F64:
            setattr(self, k, np.zeros(0, np.float64))
        for k in self.I32:
            setattr(self, k, np.zeros(0, np.int32))
        for k in self.U8:
            setattr(self, k, np.zeros(0, np.uint8))
        self.tobs_fgood_off = np.zeros(1, np.int32)
        self.sobs_kf = [np.zeros(0, np.int32) for _ in range(MAX_LEVELS)]
        self.sobs_pt = [np.zeros(0, np.int32) for _ in range(MAX_LEVELS)]
        self.sobs_flag = [np.zeros(0, np.int32) for _ in range(MAX_LEVELS)]
        self.sobs_uv0 = [np.zeros((0, 2), np.float64) for _ in range(MAX_LEVELS)]
        self.tfeat_off = [None] * MAX_LEVELS
        self.tfeat_raw = [np.zeros(0, np.int32) for _ in range(MAX_LEVELS)]
        self.tfeat_uv = [np.zeros((0, 2), np.float64) for _ in range(MAX_LEVELS)]
        self.tfeat_ref = [np.zeros((0, 8), np.float64) for _ in range(MAX_LEVELS)]
        self.img = [None] * MAX_LEVELS          # per level: uint8 array [n_kf, h, w]
        self.img_dev = [None] * MAX_LEVELS      # optional, per level: n_kf device addresses (Frame.level_device_ptr) -- handed over instead
                                                # of the host arrays; pair with TsbaOptions.img_on_device = 1 (self.img keeps the shapes)
        self.kf_id = None                        # optional int64 [n_kf]: identities of the keyframes (tsba_problem.kf_id: the context's plane cache)
        self.truth = {}                          # ground truth (synthetic problems only)
        self._keep = []

    @property
    def n_kf(self):
        return self.pose.reshape(-1, 7).shape[0]

    @property
    def n_pt(self):
        return self.rho.size

    @property
    def n_text(self):
        return self.theta.reshape(-1, 3).shape[0]

    @property
    def n_tobs(self):
        return self.tobs_kf.size

    def normalise(self):
        """Make every array C-contiguous with the ABI dtype."""
        for k in self.F64:
            setattr(self, k, np.ascontiguousarray(getattr(self, k), np.float64))
        for k in self.I32:
            setattr(self, k, np.ascontiguousarray(getattr(self, k), np.int32))
        for k in self.U8:
            setattr(self, k, np.ascontiguousarray(getattr(self, k), np.uint8))
        for l in range(MAX_LEVELS):
            self.sobs_kf[l] = np.ascontiguousarray(self.sobs_kf[l], np.int32)
            self.sobs_pt[l] = np.ascontiguousarray(self.sobs_pt[l], np.int32)
            self.sobs_flag[l] = np.ascontiguousarray(self.sobs_flag[l], np.int32)
            self.sobs_uv0[l] = np.ascontiguousarray(self.sobs_uv0[l], np.float64)
            if self.tfeat_off[l] is None:
                self.tfeat_off[l] = np.zeros(self.n_text + 1, np.int32)
            self.tfeat_off[l] = np.ascontiguousarray(self.tfeat_off[l], np.int32)
            self.tfeat_raw[l] = np.ascontiguousarray(self.tfeat_raw[l], np.int32)
            self.tfeat_uv[l] = np.ascontiguousarray(self.tfeat_uv[l], np.float64)
            self.tfeat_ref[l] = np.ascontiguousarray(self.tfeat_ref[l], np.float64)
            if self.img[l] is not None:
                self.img[l] = np.ascontiguousarray(self.img[l], np.uint8)
        if self.kf_initial.size == 0:
            self.kf_initial = np.zeros(self.n_kf, np.uint8)
        if self.pt_host_Trw.size == 0:
            self.pt_host_Trw = np.zeros((self.n_pt, 12))
        if self.text_host_Twr.size == 0:
            self.text_host_Twr = np.zeros((self.n_text, 12))
        return self

    def struct(self):
        self.normalise()
        s = TsbaProblem()
        s.n_kf, s.n_pt, s.n_text, s.n_levels = self.n_kf, self.n_pt, self.n_text, self.n_levels
        for i in range(4):
            s.K[i] = float(self.K[i])
        s.pose, s.rho, s.theta = _ptr(self.pose, C.c_double), _ptr(self.rho, C.c_double), _ptr(self.theta, C.c_double)
        s.kf_initial = _ptr(self.kf_initial, C.c_uint8)
        s.pt_ray, s.pt_host, s.pt_host_Trw = _ptr(self.pt_ray, C.c_double), _ptr(self.pt_host, C.c_int32), _ptr(self.pt_host_Trw, C.c_double)
        s.text_host, s.text_host_Twr = _ptr(self.text_host, C.c_int32), _ptr(self.text_host_Twr, C.c_double)
        s.text_box_ray = _ptr(self.text_box_ray, C.c_double)
        s.n_sgood, s.sgood = self.sgood.size, _ptr(self.sgood, C.c_uint8)
        s.n_tobs = self.n_tobs
        s.tobs_kf, s.tobs_text, s.tobs_good = _ptr(self.tobs_kf, C.c_int32), _ptr(self.tobs_text, C.c_int32), _ptr(self.tobs_good, C.c_uint8)
        s.tobs_fgood_off, s.tfgood = _ptr(self.tobs_fgood_off, C.c_int32), _ptr(self.tfgood, C.c_uint8)
        keep = []
        for l in range(MAX_LEVELS):
            s.n_sobs[l] = self.sobs_kf[l].size
            s.sobs_kf[l], s.sobs_pt[l], s.sobs_flag[l] = _ptr(self.sobs_kf[l], C.c_int32), _ptr(self.sobs_pt[l], C.c_int32), _ptr(self.sobs_flag[l], C.c_int32)
            s.sobs_uv0[l] = _ptr(self.sobs_uv0[l], C.c_double)
            s.n_tfeat[l] = self.tfeat_raw[l].size
            s.tfeat_off[l], s.tfeat_raw[l] = _ptr(self.tfeat_off[l], C.c_int32), _ptr(self.tfeat_raw[l], C.c_int32)
            s.tfeat_uv[l], s.tfeat_ref[l] = _ptr(self.tfeat_uv[l], C.c_double), _ptr(self.tfeat_ref[l], C.c_double)
            if self.img[l] is not None and l < self.n_levels:
                im = self.img[l]
                arr = (c_up * self.n_kf)()
                for k in range(self.n_kf):
                    arr[k] = im[k].ctypes.data_as(c_up) if self.img_dev[l] is None else C.cast(C.c_void_p(int(self.img_dev[l][k])), c_up)
                keep.append(arr)
                s.img[l] = C.cast(arr, c_upp)
                s.img_h[l], s.img_w[l] = im.shape[1], im.shape[2]
            else:
                s.img[l] = C.cast(None, c_upp)
        kid = getattr(self, "kf_id", None)
        if kid is not None:
            self.kf_id = np.ascontiguousarray(kid, np.int64)
            assert self.kf_id.size == self.n_kf
            s.kf_id = self.kf_id.ctypes.data_as(C.POINTER(C.c_int64))
        self._keep = keep
        return s

    def copy(self):
        import copy
        q = BAProblem()
        for k, v in self.__dict__.items():
            if k == "_keep":
                continue
            setattr(q, k, copy.deepcopy(v))
        return q

    def algorithmic_bytes(self, level, n_sblock, n_tblock, n_pair):
        """bytes one residual+Jacobian evaluation must move."""
        return (44 * n_sblock + 128 * n_tblock + 16 * n_pair
                + 56 * self.n_kf + 8 * self.n_pt + 24 * self.n_text)


def write_dump(path, P, state=STATE_LOCAL):
    """A flat problem as the record file the C++ drivers of tests/cxx read (abi_from_cxx: object graph -> adapter gather -> C ABI -> scatter)."""
    import struct
    P.normalise()
    rec = []

    def put(name, a, dt):
        a = np.ascontiguousarray(a, {0: np.float64, 1: np.int32, 2: np.uint8}[dt]).reshape(-1)
        rec.append(struct.pack("<I", len(name)) + name.encode() + struct.pack("<BQ", dt, a.size) + a.tobytes())
    n_kf = P.n_kf
    put("n_levels", [P.n_levels], 1); put("state", [state], 1); put("K", P.K, 0)
    for k in ("pose", "rho", "theta", "pt_ray", "pt_host_Trw", "text_host_Twr", "text_box_ray"):
        put(k, getattr(P, k), 0)
    for k in ("pt_host", "text_host", "tobs_kf", "tobs_text", "tobs_fgood_off"):
        put(k, getattr(P, k), 1)
    for k in ("kf_initial", "sgood", "tobs_good", "tfgood"):
        put(k, getattr(P, k), 2)
    # the keyframes' flag ranges: level 0 lists every raw observation of a keyframe, in order
    cnt = np.bincount(P.sobs_kf[0], minlength=n_kf) if P.sobs_kf[0].size else np.zeros(n_kf, np.int64)
    off = np.concatenate([[0], np.cumsum(cnt)])
    assert off[-1] == P.sgood.size and np.array_equal(P.sobs_flag[0], np.arange(P.sgood.size))
    put("kf_flag_off", off, 1)
    for l in range(P.n_levels):
        put("sobs_kf_%d" % l, P.sobs_kf[l], 1); put("sobs_pt_%d" % l, P.sobs_pt[l], 1); put("sobs_flag_%d" % l, P.sobs_flag[l], 1)
        put("sobs_uv0_%d" % l, P.sobs_uv0[l], 0)
        if P.n_text:
            put("tfeat_off_%d" % l, P.tfeat_off[l], 1); put("tfeat_raw_%d" % l, P.tfeat_raw[l], 1)
            put("tfeat_uv_%d" % l, P.tfeat_uv[l], 0); put("tfeat_ref_%d" % l, P.tfeat_ref[l], 0)
        if P.img[l] is not None:
            put("img_%d" % l, P.img[l], 2); put("img_wh_%d" % l, [P.img[l].shape[2], P.img[l].shape[1]], 1)
    with open(path, "wb") as f:
        f.write(b"".join(rec))
