"""Turns the counter passes of tools/collect_profiles_r0N.sh into one JSON (profiles/r0N_pmc.json): per kernel the HBM traffic
(FETCH_SIZE x 2 + WRITE_SIZE, with the x 2 validated by the copy kernel of the SAME session), the algorithmic bytes, the
matrix-pipe busy fraction and the wave-cycle split.

usage: python tools/pmc_summary.py gpurun_out/prof_r02 profiles/r03_pmc.json"""
import collections, csv, glob, json, os, sys

src, out_path = sys.argv[1:3]


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    return acc, dur


def mean(v):
    v = sorted(v)[len(v) // 5:] if len(v) > 5 else v      # skip the warm-up launches' low readings
    return sum(v) / len(v)


def pick(acc, key):
    ks = [k for k in acc if key in k]
    return ks[0] if ks else None


res = {"_how": "tools/collect_profiles_r0N.sh on one MI355X: rocprofv3 --kernel-trace --pmc <group> --output-format csv, one group per pass, stand-alone "
               "binaries tools/wino_bench4 360 1176 64 64 1 (conv1b, both images, Winograd F(4x4,3x3)), tools/match_bench 1000 2 (two 1000 x 1000 jobs), tools/copy_bench 1024",
       "_units": "FETCH_SIZE / WRITE_SIZE in KB; SQ_* wave counters in quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs; "
                 "GRBM_GUI_ACTIVE summed over the 8 XCDs"}
# ---- calibration: known-traffic copy kernel in the same session
f, _ = counters("pmc_fetch_copy"); w, _ = counters("pmc_write_copy")
k = pick(f, "copy_calibration")
known_kb = 1024 * 1024
res["copy_calibration"] = {"kernel": k, "known_read_KB": known_kb, "known_written_KB": known_kb, "FETCH_SIZE_KB": round(mean(f[k]["FETCH_SIZE"]), 1),
                           "WRITE_SIZE_KB": round(mean(w[k]["WRITE_SIZE"]), 1),
                           "fetch_reported_over_known": round(mean(f[k]["FETCH_SIZE"]) / known_kb, 4),
                           "write_reported_over_known": round(mean(w[k]["WRITE_SIZE"]) / known_kb, 4),
                           "conclusion": "FETCH_SIZE reports 1/2 of a 16-byte-per-lane streaming read, WRITE_SIZE the bytes written: traffic = 2 x FETCH_SIZE + WRITE_SIZE"}
fcorr = 1.0 / res["copy_calibration"]["fetch_reported_over_known"]
# the same bytes read through the LDS-DMA path (global_load_lds_dwordx4), which is how the convolution kernels read everything
kd = pick(f, "lds_dma_calibration")
fcorr_dma = fcorr
if kd:
    ratio = mean(f[kd]["FETCH_SIZE"]) / known_kb
    res["lds_dma_calibration"] = {"kernel": kd, "known_read_KB": known_kb, "FETCH_SIZE_KB": round(mean(f[kd]["FETCH_SIZE"]), 1), "fetch_reported_over_known": round(ratio, 4),
                                  "conclusion": "correction factor for kernels that read through LDS-DMA = 1 / this ratio"}
    fcorr_dma = 1.0 / ratio
# ---- kernels
H, W = 360, 1176
NT = 8   # column tiles of 128 train rows at 1000 x 1000
alg = {"conv_wino4_kernel<true, true": 2 * (64 * H * W * 4 + 64 * (H // 2) * (W // 2) * 4) + 36 * 64 * 64 * 4,    # input + pooled output planes + transformed filters
       "match_gemm_kernel<false, false>": 2 * (2 * 1000 * 256 * 4 + 1000 * 1000 * 4),                             # unfused form, two jobs: both descriptor sets + the distance matrix
       "match_rerank_kernel": 2 * (1000 * 1000 * 4 + 1000 * 8),
       "match_gemm_kernel<false, true>": 2 * (2 * 1000 * 256 * 4 + 1000 * NT * (16 + 3 * 8)),                     # fused form: descriptor sets + per (row, tile) 16 bytes of bounds and ~3 entries
       "match_merge_kernel": 2 * (1000 * NT * (16 + 3 * 8) + 1000 * (1 + 3) * 1024 + 1000 * 8)}                   # the lists, the query row + ~3 candidate rows, the result
flops = {"conv_wino4_kernel<true, true": 2.0 * 2 * H * W * 64 * 64 * 9 / 4,   # executed on the matrix pipe: 1/4 of the direct convolution's
         "match_gemm_kernel<false, false>": 2 * 2.0 * 1000 * 1000 * 256,
         "match_gemm_kernel<false, true>": 2 * 2.0 * 1000 * 1000 * 256}
for prog, keys in (("wino", ["conv_wino4_kernel<true, true"]), ("match", ["match_gemm_kernel<false, false>", "match_rerank_kernel", "match_gemm_kernel<false, true>", "match_merge_kernel"])):
    f, _ = counters("pmc_fetch_" + prog); w, _ = counters("pmc_write_" + prog)
    s1, d1 = counters("pmc_sq_" + prog); s2, _ = counters("pmc_sq2_" + prog)
    for key in keys:
        k = pick(f, key)
        if not k:
            continue
        e = {"kernel": k[:120]}
        fk, wk = mean(f[k]["FETCH_SIZE"]), mean(w[pick(w, key)]["WRITE_SIZE"])
        e["FETCH_SIZE_KB"], e["WRITE_SIZE_KB"] = round(fk, 1), round(wk, 1)
        fc = fcorr_dma if key.startswith("conv_wino") else fcorr
        e["fetch_correction"] = round(fc, 3)
        e["traffic_bytes_per_launch"] = int((fc * fk + wk) * 1024)
        e["algorithmic_bytes_per_launch"] = alg[key]
        e["traffic_over_algorithmic"] = round(e["traffic_bytes_per_launch"] / alg[key], 3)
        k1 = pick(s1, key)
        if k1:
            c = {n: mean(v) for n, v in s1[k1].items()}
            dur_ns = mean(d1[k1])
            e["duration_us_under_pmc"] = round(dur_ns / 1e3, 2)
            e["SQ_VALU_MFMA_BUSY_CYCLES"] = c["SQ_VALU_MFMA_BUSY_CYCLES"]
            if dur_ns >= 0.2e6:   # the GRBM quotient reads high (3-4 GHz on a 2.4 GHz part) for dispatches of tens of microseconds: clock and busy-of-wall only for the 0.2 ms+ kernel (2.22 GHz: physical)
                clk_ghz = c["GRBM_GUI_ACTIVE"] / 8 / dur_ns
                e["clock_GHz_from_GRBM_GUI_ACTIVE"] = round(clk_ghz, 3)
                e["mfma_busy_fraction_of_wall"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 4)
            else:   # by time instead: busy cycles per SIMD at the nominal 2.4 GHz against the dispatch's duration
                e["mfma_busy_fraction_of_duration_at_2.4GHz"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / 2.4 / dur_ns, 4)
            if key in flops:
                e["executed_tflops_under_pmc"] = round(flops[key] / dur_ns / 1e3, 2)
        k2 = pick(s2, key)
        if k2:
            c = {n: mean(v) for n, v in s2[k2].items()}
            wc = c["SQ_WAVE_CYCLES"]
            e["wave_cycle_split"] = {n: round(c[n] / wc, 4) for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if n in c}
            if c.get("SQ_LDS_IDX_ACTIVE"):
                e["lds_bank_conflict_fraction"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
        res[key.split("<")[0] + ("_fused" if key.endswith("true>") and key.startswith("match_gemm") else "")] = e
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
