"""Host calls and device activity of the pipelined leg on one time axis.  Input: the stderr of
    STEP_TRACE=<first>:<last> SPVO_TUNE_TRUNK_TIMING=<first launch to trace> python3 tools/step_breakdown.py
(`T <host clock us> ...` lines of tools/step_breakdown.py = one per step, of the library = one per trunk launch; `G launch ...` lines of the
library = trunk and tail of a launch as the device timed them, on the host's clock).  This is the trace that showed round 4's idle gaps
were host-driven NMS continuations.
usage: trace_merge.py STDERR_FILE [lines = 300]"""
import re, sys
ev = []
for l in open(sys.argv[1]).read().splitlines():
    if l.startswith("T "):
        t = l.split()[1]
        ev.append((float(t), l[3 + len(t):][:170]))
    elif l.startswith("G "):
        m = re.match(r"G launch (\d+) \((\d) pairs\): trunk (\d+) \.\. (\d+), tail (\d+) \.\. (\d+)", l)
        if not m:
            continue
        n, p, b, e, tb, te = map(int, m.groups())
        ev += [(b, f"    GPU trunk {n} ({p} pairs) begins"), (e, f"    GPU trunk {n} ends ({e - b} us)"), (tb, f"        GPU tail {n} begins"), (te, f"        GPU tail {n} ends ({te - tb} us)")]
ev.sort()
for t, s in ev[:int(sys.argv[2]) if len(sys.argv) > 2 else 300]:
    print(f"{t - ev[0][0]:9.0f} {s}")
