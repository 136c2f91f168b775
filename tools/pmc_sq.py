"""Aggregate rocprofv3 --pmc SQ counter passes of `python3 tools/perop.py` for the dominant convolution (conv1b).
usage on the GPU box (cd /tmp && export TMPDIR=/tmp first), one pass per counter group:
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d OUT/a -- python3 tools/perop.py
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d OUT/b -- python3 tools/perop.py
  python3 tools/pmc_sq.py profiles/<out>.json OUT/a OUT/b"""
import collections, csv, glob, json, os, sys

out_path, dirs = sys.argv[1], sys.argv[2:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
cands = [k for k in acc if "conv_wino_kernel<true, true, 1" in k or ("conv_mfma_kernel<3" in k and k.rstrip(">(spvo::ConvArgs)").endswith(", 1"))]
dom = cands[0]
res = {c: {"mean": sum(v) / len(v), "launches": len(v)} for c, v in acc[dom].items()}
res["_kernel"] = dom
res["_note"] = ("rocprofv3 --pmc, separate passes, python3 tools/perop.py; SQ cycle counters are in units of 4 clocks, summed over all waves / SIMDs; "
                "MFMA busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (number of SIMDs x kernel cycles / 4)")
json.dump(res, open(out_path, "w"), indent=1)
print(dom, {k: v["mean"] for k, v in res.items() if isinstance(v, dict)})
