"""One steady-state frame of the SYNCHRONOUS drop-in call sequence (tools/sync_leg.py <steps> 0) as the device saw it: every kernel and every
copy of the frame on all streams, start relative to the frame's first device activity, duration, and the idle time of the whole device in
front of it.  Input: the directory of `rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d DIR -o t -- python3 tools/sync_leg.py 60 0`.
usage: sync_timeline.py DIR [frame index from the end = 5]"""
import csv, glob, os, sys
d = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 5
ev = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K", r["Kernel_Name"].split("(")[0].replace("void spvo::", "").replace("spvo::", "")[:60], r.get("Queue_Id", "")))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C", "copy " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", "")) + " B", ""))
ev.sort()
starts = [i for i, e in enumerate(ev) if e[2] == "K" and e[3].startswith("preprocess_kernel")]
if len(starts) < back + 2:
    sys.exit("too few frames in the trace")
# a frame = from the first device activity after the previous frame's last kernel ... to the last event before the next frame's uploads
pre0, pre1 = starts[-back - 1], starts[-back]


def first_of_frame(pre):   # walk back over the copies (image uploads) that precede the preprocess kernel
    i = pre
    while i > 0 and ev[i - 1][2] == "C" and ev[pre][0] - ev[i - 1][0] < 300_000:
        i -= 1
    return i
i0, i1 = first_of_frame(pre0), first_of_frame(pre1)
t0 = ev[i0][0]
busy_end = t0
tot_idle = 0.0
print(f"{'start us':>9s} {'dur us':>8s} {'idle before':>11s}  event")
for s, e, kind, name, q in ev[i0:i1]:
    idle = max(0.0, (s - busy_end) / 1e3)
    tot_idle += idle
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {idle:11.1f}  {name}" + (f"  [queue {q}]" if q else ""))
    busy_end = max(busy_end, e)
print(f"frame period {(ev[i1][0] - t0) / 1e3:.1f} us; device active span {(busy_end - t0) / 1e3:.1f} us; idle inside the span {tot_idle:.1f} us; "
      f"idle between the frame's last device event and the next frame's first {(ev[i1][0] - busy_end) / 1e3:.1f} us (host: result read-back, join, next call)")
