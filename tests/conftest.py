import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def map_cache():
    """Synthetic maps of thousands of keyframes take ~10 s to build: the full-size tests of several modules share them (read-only: every
    solve works on a copy)."""
    from textslam_amd import synth
    cache = {}

    def get(**kw):
        key = tuple(sorted(kw.items()))
        if key not in cache:
            text = kw.pop("n_text", 0)
            if text:
                cache[key] = synth.make_problem(n_text=text, max_targets=8, text_targets=5, frozen_frac=0.0, rot_deg=0.2, trans_m=0.01, **kw)
            else:
                cache[key] = synth.config_global(**kw)
        return cache[key]
    return get


@pytest.fixture(scope="session", autouse=True)
def busy_device():
    """TSBA_TEST_BUSY=orb | ba | both: the whole session runs beside other contexts that keep the device busy from their own host threads (an ORB extractor
    looping on a 16-frame batch; a bundle adjustment of a 700-keyframe map in a loop) -- what TextSLAM's tracking and loop-closing threads do beside the mapping
    thread.  Every parity / bit-identity assertion of the GPU suite then also says "whatever else runs on the device" (profiles/r05_gpu_suite_beside_busy_contexts.txt)."""
    mode = os.environ.get("TSBA_TEST_BUSY", "")
    if not mode:
        yield None
        return
    import threading
    import numpy as np
    from textslam_amd.optimizer import load_library
    load_library().tsba_comm_load()                 # (include/tsba.h: RCCL's code objects are registered before the busy threads launch kernels, not under them by the communicator tests)
    stop = threading.Event(); threads = []; counts = {"orb": 0, "ba": 0}
    if mode in ("orb", "both"):
        from textslam_amd.orbextractor import ORBextractor, synthetic_frame
        ex = ORBextractor(device=0); ex.upload(np.stack([synthetic_frame(s) for s in range(16)]))

        def orb_loop():
            while not stop.is_set():
                ex.run(); counts["orb"] += 1
        threads.append(threading.Thread(target=orb_loop, daemon=True))
    if mode in ("ba", "both"):
        from textslam_amd import synth, abi
        from textslam_amd.optimizer import Optimizer
        g = Optimizer(0); P = synth.config_global(n_kf=700, n_pt=14000, band=8); o = abi.options_global(); o.its[0] = 20
        g.upload(P, o)

        def ba_loop():
            while not stop.is_set():
                g.solve(); counts["ba"] += 1
        threads.append(threading.Thread(target=ba_loop, daemon=True))
    for t in threads:
        t.start()
    yield counts
    stop.set()
    for t in threads:
        t.join(timeout=60)
    print("\nbusy contexts ran", counts)
