"""Where the HOST spends a step of bench.py's headline leg (images in HBM, pairs handed over ahead, deferred solve): wall time per call of
the front end -- prefetch (submission + a completed group's launch), addStereoImagePair (waits for the features), the two matchDescriptors
(wait for the matches enqueued with the detector), the collect of the previous frame's solve (waits for the solver), this frame's solve
hand-over -- as mean / median / p90 / max over the timed steps.  The calls are the ones bench.py makes: this script wraps the methods of
spvo.host.FrontEnd with timers and runs bench.main().
usage: python3 tools/step_breakdown.py [bench.py arguments; default --no-cpu-baseline --no-extras --steps 200 --warmup 20]"""
import os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd"))
import numpy as np
from spvo import host

T = {}
EV = []   # (start us, duration us, name) of every wrapped call, in call order


def timed(name, fn):
    def w(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            d = (time.perf_counter() - t0) * 1e6
            T.setdefault(name, []).append(d)
            EV.append((t0 * 1e6, d, name))
    return w


FE = host.FrontEnd
FE.prefetch_device = timed("prefetch_device (submission; every other one launches a group)", FE.prefetch_device)
FE.add_stereo_image_pair_device = timed("addStereoImagePair (device images): waits for the features", FE.add_stereo_image_pair_device)
FE.match_descriptors = timed("matchDescriptors: waits for the matches", FE.match_descriptors)
FE.finish_solve = timed("collect of the previous frame's solve", FE.finish_solve)
_orig_mas = FE._match_and_solve


def _mas(self, deferred):
    t0 = time.perf_counter()
    r = _orig_mas(self, deferred)
    T.setdefault("_match_and_solve total (matches + collect + hand-over of the solve)", []).append((time.perf_counter() - t0) * 1e6)
    return r


FE._match_and_solve = _mas
_sd = FE.step_device
_nstep = [0]
TRACE = os.environ.get("STEP_TRACE")   # "first:last": one line per step in that range, interleaved with the library's own trunk-timing lines


def _step(self, *a, **k):
    i0, t0 = len(EV), time.perf_counter()
    r = _sd(self, *a, **k)
    d = (time.perf_counter() - t0) * 1e6
    T.setdefault("step_device total", []).append(d)
    EV.append((t0 * 1e6, d, "step_device total"))
    _nstep[0] += 1
    if TRACE:
        lo, hi = (int(v) for v in TRACE.split(":"))
        if lo <= _nstep[0] < hi:
            inner = [e for e in EV[i0:-1] if not e[2].startswith("_match")]
            sys.stderr.write(f"T {t0 * 1e6:.0f} step {_nstep[0]:5d} {d:6.0f} us: " + " ".join(f"{e[2][:5]}={e[1]:.0f}" for e in inner) + "\n")
    return r


FE.step_device = _step

import bench

if len(sys.argv) == 1:
    sys.argv += ["--no-cpu-baseline", "--no-extras", "--steps", "200", "--warmup", "20"]
bench.main()
print("host time per call, microseconds (all calls of the run, warm-up included):", file=sys.stderr)
for k, v in T.items():
    a = np.asarray(v[len(v) // 10:])   # drop the first tenth: engine load, first submissions
    print(f"  {k:85s} n={len(a):6d} mean {a.mean():7.1f} median {np.median(a):7.1f} p90 {np.percentile(a, 90):7.1f} max {a.max():8.1f}", file=sys.stderr)

# the slowest steps, call by call (a step = one step_device; its entry is appended when it returns, i.e. AFTER its inner calls)
steps = [i for i, e in enumerate(EV) if e[2] == "step_device total"]
steps = steps[len(steps) // 10:]
def non_wait(i):   # a step's time outside addStereoImagePair / matchDescriptors / the solve's collect (the calls that wait for the device)
    t0, d, _ = EV[i]
    j, w = i - 1, 0.0
    while j >= 0 and EV[j][0] >= t0:
        if EV[j][2].startswith(("addStereo", "matchDesc", "collect")):
            w += EV[j][1]
        j -= 1
    return d - w


nw = np.asarray([non_wait(i) for i in steps])
print(f"time of a step outside the waiting calls: mean {nw.mean():.0f} median {np.median(nw):.0f} p90 {np.percentile(nw, 90):.0f} max {nw.max():.0f} us; "
      f"{int((nw > 500).sum())} of {len(nw)} steps above 500 us", file=sys.stderr)
slow = [i for i, v in zip(steps, nw) if v > 500][:8]
for i in sorted(slow):
    t0, d, _ = EV[i]
    print(f"  step at {t0 - EV[steps[0]][0]:10.0f} us, {d:7.0f} us:", file=sys.stderr)
    j = i - 1
    inner = []
    while j >= 0 and EV[j][0] >= t0:
        inner.append(EV[j]); j -= 1
    for st, dd, nm in reversed(inner):
        if nm.startswith("_match"): continue
        print(f"      +{st - t0:7.0f} us  {dd:7.0f} us  {nm[:60]}", file=sys.stderr)
