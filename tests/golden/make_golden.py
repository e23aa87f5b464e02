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


def make_orb():
    from textslam_amd.orbextractor import synthetic_frame
    seed = 77
    kp, desc = oracle.orb_extract(synthetic_frame(seed))
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "orb_frame.npz"), seed=seed, kp=kp, desc=desc)
    print("orb_frame", kp.shape)


if __name__ == "__main__":
    main()
    make_orb()
