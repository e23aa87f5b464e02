"""GPU diagnostic (not a pytest): call latencies of the BA-pyramid front-end (640x480, 4 levels) next to the CPU restatement."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from textslam_amd.frame import Frame
from textslam_amd.orbextractor import synthetic_frame
img = synthetic_frame(3)
fr = Frame(0)
inv = [1.0, 0.5, 0.25, 0.125]
rng = np.random.default_rng(1)
xy = np.stack([rng.uniform(0, 639, 1000), rng.uniform(0, 479, 1000)], 1).astype(np.float32)
for _ in range(3): fr.GetPyrMat(img, 4); pts = fr.GetPyramidPtsScene(xy, inv); fr.CalNormvec(0, xy.astype(np.float64), 100.0, 30.0)
def t(f, n=50):
    t0 = time.perf_counter()
    for _ in range(n): f()
    return (time.perf_counter() - t0)/n*1e3
print("GetPyrMat (upload 300 KB + 3 pyrDown + 4 x Sobel/blend): %.3f ms" % t(lambda: fr.GetPyrMat(img, 4)))
print("GetPyramidPts scene, 1000 features x 4 levels:          %.3f ms" % t(lambda: fr.GetPyramidPtsScene(xy, inv)))
print("CalNormvec 1000 features x 8 taps:                       %.3f ms" % t(lambda: fr.CalNormvec(0, xy.astype(np.float64), 100.0, 30.0)))
t0 = time.perf_counter(); pyr = oracle.frame_pyramid(img, 4); t1 = time.perf_counter(); oracle.frame_pyramid_pts(1, xy, None, pyr, inv); t2 = time.perf_counter()
print("CPU restatement (1 core): pyramid %.2f ms, pyramid pts %.2f ms" % ((t1 - t0)*1e3, (t2 - t1)*1e3))
