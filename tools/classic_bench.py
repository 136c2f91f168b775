"""ClassicFeatureFrontEnd(ORB, ORB, BF, KNN) on the GPU over the synthetic stream: frames/s of the synchronous stereoCallback
(python tools/classic_bench.py [frames]); run under rocprofv3 --kernel-trace --stats for the per-kernel split."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd"))
import numpy as np
from spvo import host, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
frames, poses, P_l, P_r = synth.stereo_sequence(8, os.path.join(ROOT, "tests", "golden", "images", "0000000000.png"), seed=0)
seq = [frames[i % 8] for i in range(n)]
p, s, sec = host.classic_sequence(seq, P_l, P_r, "KNN", True, 2.0, 4, warm=5)
print("classic front end on the GPU: %.1f stereo frames/s (%.3f ms per pair), keypoints %d, stereo matches %d, inliers %d" % ((n - 5) / sec, 1e3 * sec / (n - 5), np.median(s[5:, 0]), np.median(s[5:, 2]), np.median(s[5:, 3])))
