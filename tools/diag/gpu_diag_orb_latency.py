"""GPU diagnostic (not a pytest): ORB extraction latency for one frame (the per-frame call of the SLAM front-end) and for a batch."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd.orbextractor import ORBextractor, synthetic_frame
ex = ORBextractor(1000, 1.2, 8, 20, 7)
for n in (1, 2, 64):
    imgs = np.stack([synthetic_frame(100 + k) for k in range(n)])
    for rep in range(4):
        t = time.time(); out = ex.extract_batch(imgs); dt = (time.time() - t)*1e3
    ex.upload(imgs)
    ts = []
    for rep in range(5):
        t = time.time(); ex.run(); ts.append((time.time() - t)*1e3)
    print("n=%d: extract_batch (upload+run+download) %.3f ms, resident run %.3f ms, keypoints/frame %d" % (n, dt, min(ts), len(out[0][0]) if isinstance(out, (list, tuple)) else -1))
