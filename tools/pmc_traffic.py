"""Aggregate two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately as MI355X_MICROARCH.md
prescribes) of `python3 tools/perop.py` into profiles/<out>.json: HBM bytes per launch for every kernel, and the
dominant convolution (conv1b) singled out for bench.py's `roofline.traffic`.

usage on the GPU box (cd /tmp && export TMPDIR=/tmp first):
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT/pmc_fetch -- python3 tools/perop.py
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d OUT/pmc_write -- python3 tools/perop.py
  python3 tools/pmc_traffic.py OUT/pmc_fetch OUT/pmc_write profiles/r01_pmc_conv_traffic.json
"""
import collections, csv, glob, json, os, sys

fetch_dir, write_dir, out_path = sys.argv[1:4]


def collect(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            key = f'{r["Kernel_Name"][:90]} grid={r["Grid_Size"]}'
            acc[key].append(float(r["Counter_Value"]))
    return acc


fetch, write = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
per = {}
for k in sorted(set(fetch) | set(write)):
    per[k] = {"dispatches": len(fetch.get(k, write.get(k))),
              "FETCH_SIZE_KB_mean": round(sum(fetch[k]) / len(fetch[k]), 1) if k in fetch else None,
              "WRITE_SIZE_KB_mean": round(sum(write[k]) / len(write[k]), 1) if k in write else None}
# conv1b = the pooled 3x3 kernel instance (Winograd by default, the direct MFMA kernel with the diagnostic switch "winograd" = 0) with the
# largest fetch volume
cands = [k for k in per if ("conv_wino_kernel<true, true" in k or ("conv_mfma_kernel<3" in k and "true, true" in k)) and per[k]["FETCH_SIZE_KB_mean"]]
dom = max(cands, key=lambda k: per[k]["FETCH_SIZE_KB_mean"])
f_kb, w_kb = per[dom]["FETCH_SIZE_KB_mean"], per[dom]["WRITE_SIZE_KB_mean"]
H, W = 360, 1176
algorithmic = 2 * (64 * H * W * 4 + 64 * (H // 2) * (W // 2) * 4) + 64 * 64 * 9 * 4   # input planes + pooled output planes + weights
out = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --output-format csv -- python3 tools/perop.py",
    "note": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for 16-B-per-lane streaming reads (MI355X_MICROARCH.md, HBM) -> doubled; "
            "WRITE_SIZE taken as is; counters are in KB",
    "dominant_kernel": dom,
    "fetch_KB_raw": f_kb, "write_KB": w_kb,
    "traffic_bytes_per_launch": int((2 * f_kb + w_kb) * 1024),
    "algorithmic_bytes_per_launch": algorithmic,
    "per_kernel": per,
}
json.dump(out, open(out_path, "w"), indent=1)
print(dom, "traffic", out["traffic_bytes_per_launch"], "algorithmic", algorithmic)
