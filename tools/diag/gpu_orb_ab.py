import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd.orbextractor import ORBextractor, synthetic_frame
imgs = np.stack([synthetic_frame(s) for s in range(64)])
for n in [int(x) for x in os.environ.get("ORB_AB_FRAMES", "64,1").split(",")]:
    ex = ORBextractor(1000, 1.2, 8, 20, 7, device=0); ex.debug_fast_shape(int(os.environ.get("ORB_AB_SHAPE", "-1"))); ex.debug_pyramid(int(os.environ.get("ORB_AB_PYR", "-1")));
    if "ORB_AB_SPLIT" in os.environ: ex.debug_pyramid(100 + int(os.environ["ORB_AB_SPLIT"]))
    if "ORB_AB_OB" in os.environ: ex.debug_pyramid(200 + int(os.environ["ORB_AB_OB"]))
    ex.upload(imgs[:n])
    for _ in range(5): ex.run()
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); ex.run(); ts.append((time.perf_counter() - t0)*1e3)
    print(os.environ.get("TSORB_LIB", "default"), "shape", os.environ.get("ORB_AB_SHAPE", "-1"), "pyramid", os.environ.get("ORB_AB_PYR", "-1"), "split", os.environ.get("ORB_AB_SPLIT", "-"), "orient+blur", os.environ.get("ORB_AB_OB", "-"), f"frames {n:2d}: median {np.median(ts):.4f} ms  min {min(ts):.4f} ms", flush=True)
    ex.close()
