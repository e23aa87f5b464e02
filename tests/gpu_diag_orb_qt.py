import sys; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from textslam_amd.orbextractor import ORBextractor, synthetic_frame
orb = ORBextractor()
imgs = np.stack([synthetic_frame(100 + i) for i in range(64)])
orb.extract_batch(imgs) if hasattr(orb, "extract_batch") else orb(imgs[0])
