"""HBM traffic of every layer of the conv stack from the PMC counters (north_star: "rocprof reports achieved HBM GB/s on the conv stack").

Inputs: two rocprofv3 passes of `tools/forward_loop.py` -- `--pmc FETCH_SIZE` and, separately, `--pmc WRITE_SIZE`, each with
`--kernel-trace` only (MI355X_MICROARCH.md, rocprofv3 PMC slots: the two counters do not fit one pass) -- and the per-layer timing file of
tools/layer_roofline_json.py (same engine, same size, HIP events, NOT under the profiler).  The dispatches of a forward pass are matched to
the layers by their order (several layers share a kernel name).  traffic = 2 x FETCH_SIZE + WRITE_SIZE: on gfx950 FETCH_SIZE counts the
128-byte requests of wide streaming reads as 64 bytes (same guide; validated with a copy kernel of known size, profiles/r02_pmc.json
`copy_calibration`), WRITE_SIZE is exact; both count Infinity-Cache hits.  Counters are reported in KB.

usage: pmc_layers.py <fetch dir> <write dir> <layer_roofline.json> <out.json> [passes = 30]"""
import collections, csv, glob, json, os, sys

fetch_dir, write_dir, lr_path, out_path = sys.argv[1:5]
passes = int(sys.argv[5]) if len(sys.argv) > 5 else 30
LAYER_KERNELS = ("conv_", "heads_fused", "heads_i8", "l2norm", "dwconv", "dwpw_", "maxpool")


def sequence(d, counter):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in LAYER_KERNELS):
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    return rows


lr = json.load(open(lr_path))
layers = lr["layers"]
out_layers = [dict(layer=l["layer"], shape=l["shape"], kernel=l["kernel"].split(" ")[0], duration_us=l["duration_us"], algorithmic_MB=l["algorithmic_MB"]) for l in layers]
per_counter = {}
for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
    seq = sequence(d, name)
    L = len(layers)
    if len(seq) > passes * L:      # dispatches in front of the loop (an INT8 engine's calibration passes run on an fp32 engine first)
        seq = seq[len(seq) - passes * L:]
    if len(seq) % L or len(seq) // L != passes:
        raise SystemExit(f"{name}: {len(seq)} layer dispatches do not make {passes} passes of {L} layers")
    acc = collections.defaultdict(list)
    knames = {}
    for i, (_, kn, v) in enumerate(seq):
        if i // L < passes // 3:          # the first third of the passes warms the caches and the clocks
            continue
        acc[i % L].append(v)
        knames[i % L] = kn
    per_counter[name] = {k: sum(v) / len(v) for k, v in acc.items()}
    for k, l in enumerate(out_layers):
        l.setdefault("kernel_name", knames[k][:80])
        assert l["kernel_name"] == knames[k][:80], (l, knames[k])      # both passes see the same kernel at the same position
tot_t = tot_a = tot_us = 0.0
for k, l in enumerate(out_layers):
    f_kb, w_kb = per_counter["FETCH_SIZE"][k], per_counter["WRITE_SIZE"][k]
    traffic = (2 * f_kb + w_kb) * 1024
    l.update(FETCH_SIZE_KB=round(f_kb, 1), WRITE_SIZE_KB=round(w_kb, 1), traffic_MB=round(traffic / 1e6, 1),
             traffic_over_algorithmic=round(traffic / 1e6 / l["algorithmic_MB"], 3) if l["algorithmic_MB"] else None,
             hbm_GBps=round(traffic / l["duration_us"] / 1e3, 0), frac_of_hbm_peak=round(traffic / l["duration_us"] / 1e3 / 8000.0, 4))
    tot_t += traffic; tot_a += l["algorithmic_MB"] * 1e6; tot_us += l["duration_us"]
res = {"_how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/forward_loop.py; dispatches matched to layers by "
               "order; traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction of MI355X_MICROARCH.md; counters in KB, Infinity-Cache hits included); durations from "
               + os.path.basename(lr_path) + " (HIP events, not under the profiler)",
       "engine": lr.get("_how", "").split(":")[1].split(",")[0].strip() if ":" in lr.get("_how", "") else "", "layers": out_layers,
       "conv_stack": {"traffic_MB": round(tot_t / 1e6, 1), "algorithmic_MB": round(tot_a / 1e6, 1), "traffic_over_algorithmic": round(tot_t / tot_a, 3),
                      "sum_of_layers_us": round(tot_us, 1), "hbm_GBps": round(tot_t / tot_us / 1e3, 0), "frac_of_hbm_peak": round(tot_t / tot_us / 1e3 / 8000.0, 4)}}
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res["conv_stack"]))
for l in out_layers:
    print(f'{l["layer"][:28]:30s} {l["duration_us"]:7.1f} us  traffic {l["traffic_MB"]:7.1f} MB = {(l["traffic_over_algorithmic"] or 0):5.2f} x algorithmic  {l["hbm_GBps"]:6.0f} GB/s')
