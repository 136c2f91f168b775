"""Thread scaling of the CPU restatement's network on the machine it runs on (picks the thread count bench.py's cpu_baseline uses)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd"))
import numpy as np
from oracle import cpu_backend
from spvo import weights

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cgroup cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError:
    pass
lib = cpu_backend.build("native", out=os.path.join(tempfile.mkdtemp(), "l.so"))
plan = weights.vgg_plan(seed=0)
p = os.path.join(tempfile.mkdtemp(), "v.spvw"); weights.save(plan, p)
x = np.random.rand(1, 1, 360, 1176).astype(np.float32)
for nt in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64, 128]:
    c = cpu_backend.CpuBackend(lib, net_height=360, net_width=1176, num_threads=nt)
    c.load_weights(p)
    c.forward(x)
    ts = []
    for _ in range(3):
        t = time.time(); c.forward(x); ts.append(time.time() - t)
    dt = min(ts)
    print(nt, "threads: %.3f s per image  %.0f GFLOP/s  (%.1f per thread)" % (dt, 71.8 / dt, 71.8 / dt / nt), flush=True)
    c.close()
