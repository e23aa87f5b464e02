"""Generates tests/golden/*.npz with the CPU oracle (run in the build container: python tests/golden/make_golden.py).

The reference ships no golden vectors and cannot be built here (needs Ceres / OpenCV / Eigen), so these fixtures are
outputs of oracle/tsba_oracle.c on seeded synthetic windows (PARITY UNPINNED, see oracle/tsba_oracle.h).  They pin
(a) the oracle against silent regressions / platform drift and (b) the HIP path on the GPU box.
Each file: the generator arguments, a checksum of the generated inputs, and the expected outputs."""
import os
import sys
import hashlib
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from textslam_amd import synth, abi  # noqa: E402
import oracle  # noqa: E402

CASES = {
    "tiny_local": dict(kind="local", gen=dict(seed=7, n_kf=5, n_pt=60, n_text=4)),
    "tiny_scene": dict(kind="scene", gen=dict(seed=11, n_kf=6, n_pt=300, n_text=0)),
    "tiny_pose": dict(kind="pose", gen=dict(seed=13, n_kf=1, n_pt=200, n_text=5, frozen_frac=1.0, max_targets=1, text_targets=1)),
}


def make_case(name):
    c = CASES[name]
    P = synth.tiny(**c["gen"])
    if c["kind"] == "local":
        o = abi.options_local()
    elif c["kind"] == "pose":
        o = abi.options_pose()
    else:
        o = abi.options_local(); o.use_text = 0; o.n_passes = 1; o.levels[0] = 0; o.its[0] = 15
    return P, o


def input_digest(P):
    h = hashlib.sha256()
    for a in (P.pose, P.rho, P.theta, P.pt_ray, P.sobs_uv0[0], P.tfeat_ref[0], P.sgood, P.tfgood):
        h.update(np.ascontiguousarray(a).tobytes())
    if P.img[0] is not None:
        h.update(P.img[0].tobytes())
    return h.hexdigest()


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name in CASES:
        P, o = make_case(name)
        lvl = o.levels[o.n_passes - 1]
        ev = oracle.evaluate(P, o, lvl)
        Q = P.copy()
        rep = oracle.solve(Q, o)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"),
                            digest=input_digest(P), level=lvl,
                            resid=ev["resid"], jac_scene=ev["jac_scene"], jac_text=ev["jac_text"], musigma=ev["musigma"],
                            pose=Q.pose, rho=Q.rho, theta=Q.theta, sgood=Q.sgood, tobs_good=Q.tobs_good, tfgood=Q.tfgood,
                            iters=np.array(rep["iters"]), cost0=np.array(rep["cost0"]), cost1=np.array(rep["cost1"]))
        print(name, rep["iters"], rep["cost1"])


# global BA on maps whose reduced system is not a plain band: the LM run trial by trial (candidate cost, model cost change, radius, decision), the final
# parameters and the reduced gradient of the first linearisation -- for the CPU suite (oracle regression) and for the GPU suite (every solver path of
# tsba_global_ba against a COMMITTED vector, not only against the oracle built on the test box)
GLOBAL_CASES = {
    "mid_global_long_range": dict(n_kf=130, n_pt=4000, band=8, far_frac=0.03),
    "mid_global_ring": dict(n_kf=150, n_pt=3600, band=8, loop=True),
    "mid_global_two_closures": dict(n_kf=160, n_pt=4000, band=8, closures=2),
}


def make_global_case(name):
    P = synth.config_global(**GLOBAL_CASES[name])
    o = abi.options_global(); o.its[0] = 10
    return P, o


def global_digest(P):
    h = hashlib.sha256()
    for a in (P.pose, P.rho, P.pt_ray, P.sobs_uv0[0], P.sobs_kf[0], P.sobs_pt[0], P.pt_host):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def make_global():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    for name in GLOBAL_CASES:
        P, o = make_global_case(name)
        rb = oracle.reduced_blocks(P, o, 0, o.initial_radius)
        Q = P.copy()
        rep, tr = oracle.solve_traced(Q, o)
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), digest=global_digest(P), trace=tr[0], pose=Q.pose, rho=Q.rho,
                            g=rb["g"], free_idx=rb["free_idx"], cost_lin=rb["cost"], n_blocks=len(rb["br"]), abs_sum_S=float(np.abs(rb["val"]).sum()),
                            iters=np.array(rep["iters"]), accepted=np.array(rep["accepted"]), termination=np.array(rep["termination"]),
                            cost0=np.array(rep["cost0"]), cost1=np.array(rep["cost1"]))
        print(name, rep["iters"], rep["accepted"], rep["cost1"], len(rb["br"]))


def make_orb():
    from textslam_amd.orbextractor import synthetic_frame
    seed = 77
    kp, desc = oracle.orb_extract(synthetic_frame(seed))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "orb_frame.npz"), seed=seed, kp=kp, desc=desc)
    print("orb_frame", kp.shape)


if __name__ == "__main__":
    main()
    make_global()
    make_orb()
