"""Times the matcher's kernels alone (K12a distance GEMM, K12b exact re-rank) through the C ABI's stage timers.
usage: python tools/match_bench.py [--fp8]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd"))
from spvo import capi  # noqa: E402


def main():
    ctx = capi.Context(max_keypoints=2048)
    if "--fp8" in sys.argv:
        ctx.set_match_fp8(True)
    rng = np.random.RandomState(0)
    for n in (1000, 2048):
        a = rng.randn(n, 256).astype(np.float32)
        a /= np.linalg.norm(a, axis=1, keepdims=True)
        b = rng.randn(n, 256).astype(np.float32)
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        b[: n // 2] = a[: n // 2] + 0.03 * rng.randn(n // 2, 256).astype(np.float32)
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        for _ in range(20):
            ctx.match(a, b)
        ctx.profile_enable(True)
        ctx.profile_reset()
        for _ in range(200):
            ctx.match(a, b)
        prof = ctx.profile()
        ctx.profile_enable(False)
        fl = 2.0 * n * n * 256
        for st in ("match_gemm", "match_rerank", "match"):
            if st in prof and prof[st]["calls"]:
                us = prof[st]["total_ms"] / prof[st]["calls"] * 1e3
                extra = f"  {fl / us / 1e6:7.1f} TFLOP/s = {fl / us / 1e6 / 157.3:.3f} of the fp32 MFMA peak" if st == "match_gemm" else ""
                print(f"n={n:5d} {st:13s} {us:8.2f} us{extra}")
    ctx.close()


if __name__ == "__main__":
    main()
