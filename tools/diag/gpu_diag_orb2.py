import sys, os
import numpy as np
sys.path.insert(0, '/root/repo')
from textslam_amd.orbextractor import ORBextractor, synthetic_frame
import oracle
orb = ORBextractor()
cases = {"low": (synthetic_frame(50).astype(np.int32)//12+100).astype(np.uint8), "small": np.ascontiguousarray(synthetic_frame(51)[:240,:320]),
         "noise": np.random.default_rng(5).integers(0,256,(480,640)).astype(np.uint8)}
for name, img in cases.items():
    kg, dg = orb(img); ko, do = oracle.orb_extract(img, cap=8192)
    print(name, len(kg), len(ko), np.bincount(kg[:,5].astype(int),minlength=8), np.bincount(ko[:,5].astype(int),minlength=8))
    m=min(len(kg),len(ko)); eq=np.all(kg[:m,:2]==ko[:m,:2],axis=1); print("   xy equal", eq.sum(), "of", m, "first diff", int(np.argmin(eq)) if not eq.all() else -1)
