"""GPU diagnostic (not a pytest): timing of the window search (grid build + search of 1000 queries in a 1000-feature frame) next to the CPU oracle."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd.orbextractor import ORBextractor, synthetic_frame
import oracle
ex = ORBextractor(1000, 1.2, 8, 20, 7)
(kpA, dA), (kpB, dB) = ex.extract_batch(np.stack([synthetic_frame(40), synthetic_frame(41)]))
bounds = (0.0, 640.0, 0.0, 480.0)
nq = kpA.shape[0]
qxy = kpA[:, :2].copy(); qr = np.full(nq, 40.0, np.float32); oct_ = kpA[:, 5].astype(np.int32); qlev = np.stack([oct_ - 1, oct_ + 1], 1)
for rep in range(3):
    t = time.time(); ex.match_set_frame(1, bounds); t1 = time.time(); out = ex.match_search(qxy, qr, qlev, dA, 32); t2 = time.time()
print("GPU: grid build %.3f ms, search of %d queries %.3f ms (mean candidates %.1f)" % ((t1 - t)*1e3, nq, (t2 - t1)*1e3, out["cand_cnt"].mean()))
t = time.time()
for rep in range(20): ref = oracle.orb_match(kpB, dB, bounds, qxy, qr, qlev, dA, 32)
print("CPU oracle: %.3f ms per call" % ((time.time() - t)*1e3/20))
