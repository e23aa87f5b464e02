import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from textslam_amd.orbextractor import ORBextractor, synthetic_frame
import oracle
n = 4
imgs = np.stack([synthetic_frame(s) for s in range(n)])
ex = ORBextractor()
t = time.time(); res = ex.extract_batch(imgs); print("gpu batch of %d: %.1f ms" % (n, (time.time()-t)*1e3))
for l in range(8):
    a = ex.debug_level(0, l); b = oracle.orb_level(imgs[0], l)
    ab = ex.debug_level(0, l, True); bb = oracle.orb_level(imgs[0], l, blurred=True)
    print("level", l, a.shape, "pyr diff px", int(np.sum(a != b)), "blur diff px", int(np.sum(ab != bb)))
for f in range(n):
    kp_o, d_o = oracle.orb_extract(imgs[f])
    kp_g, d_g = res[f]
    print("frame", f, "n gpu/oracle", len(kp_g), len(kp_o), "per level gpu", np.bincount(kp_g[:,5].astype(int), minlength=8), "ora", np.bincount(kp_o[:,5].astype(int), minlength=8))
    m = min(len(kp_g), len(kp_o))
    same_xy = np.all(kp_g[:m,:2] == kp_o[:m,:2], axis=1)
    print("   same xy %d/%d  angle maxdiff %.3g  resp eq %d  desc eq %d" % (same_xy.sum(), m, np.abs(kp_g[:m,3]-kp_o[:m,3]).max() if m else 0,
          int(np.sum(kp_g[:m,4]==kp_o[:m,4])), int(np.sum(np.all(d_g[:m]==d_o[:m],axis=1)))))
    if not same_xy.all():
        i = int(np.argmin(same_xy)); print("   first mismatch at", i, kp_g[i], kp_o[i])
# timing with resident batch of 64
imgs64 = np.stack([synthetic_frame(100+s) for s in range(64)])
ex.upload(imgs64)
for _ in range(3):
    t = time.time(); ex.run(); print("run 64 frames: %.2f ms" % ((time.time()-t)*1e3))
