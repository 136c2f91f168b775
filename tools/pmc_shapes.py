"""Matrix-pipe busy fraction, wave-cycle split and LDS bank conflicts of the F(4x4) Winograd kernel on the deep layers' shapes
(tools/collect_profiles_r04.sh: tools/wino_bench4 stand-alone, one rocprofv3 --pmc group per pass, --kernel-trace only).
usage: pmc_shapes.py <prof dir> <out.json>"""
import collections, csv, glob, json, os, sys
src, out_path = sys.argv[1:3]


def counters(d):
    acc, dur = collections.defaultdict(list), []
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv_wino4_kernel" not in r["Kernel_Name"]:
                continue
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur.append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    mean = lambda v: sum(sorted(v)[len(v) // 5:]) / max(1, len(sorted(v)[len(v) // 5:]))
    return {k: mean(v) for k, v in acc.items()}, (mean(dur) if dur else 0.0)


res = {"_how": "tools/collect_profiles_r04.sh: rocprofv3 --kernel-trace --pmc <group> -- tools/wino_bench4 <shape> (both images, stand-alone, back-to-back launches)",
       "_units": "SQ_* wave counters in quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs; GRBM_GUI_ACTIVE summed over the 8 XCDs"}
for L in ("conv2a", "conv2b", "conv3a", "conv3b", "convPaDa"):
    c1, d1 = counters("shape_sq_" + L)
    c2, _ = counters("shape_sq2_" + L)
    if not c1:
        continue
    e = {"duration_us_under_pmc": round(d1 / 1e3, 2), "clock_GHz_from_GRBM_GUI_ACTIVE": round(c1["GRBM_GUI_ACTIVE"] / 8 / d1, 3),
         "mfma_busy_fraction_of_wall": round(c1["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c1["GRBM_GUI_ACTIVE"] / 8), 4)}
    if c2:
        wc = c2["SQ_WAVE_CYCLES"]
        e["wave_cycle_split"] = {n: round(c2[n] / wc, 4) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if n in c2}
        if c2.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_bank_conflict_fraction"] = round(c2["SQ_LDS_BANK_CONFLICT"] / c2["SQ_LDS_IDX_ACTIVE"], 4)
    res[L] = e
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
