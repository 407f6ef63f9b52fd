"""profiles/r02_heff_traffic.json and r02_heff_mfma_util.json from the four rocprofv3 passes of tools/profile_r02_heff.sh
(kernel trace; FETCH_SIZE; WRITE_SIZE; MFMA busy / GRBM_GUI_ACTIVE).  FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is
doubled on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section).   python tools/heff_profile_summary.py [dir] [round tag]
(round tag r02 by default: input files <tag>_heff_pmc_*.csv in dir, output profiles/<tag>_heff_traffic.json / _mfma_util.json)"""
import collections, csv, json, os, sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r02prof"
TAG = sys.argv[2] if len(sys.argv) > 2 else "r02"
NAMES = ["S1 L.psi (NN)", "S2 heavy row range (NN)", "S2 light row tiles (list kernel)", "S3 .R (NT, 8 K splits)", "S3 ordered combine of the K splits"]


def launches(path, counters):
    rows = [r for r in csv.DictReader(open(path)) if "zgemm" in r["Kernel_Name"]]
    by = collections.OrderedDict()
    for r in rows:
        d = by.setdefault(r["Dispatch_Id"], {"ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
        d[r["Counter_Name"]] = float(r["Counter_Value"])
    seq = list(by.values())
    per = len(NAMES)
    seq = seq[per:]  # the first apply warms up
    n = len(seq) // per
    out = []
    for s in range(per):
        sel = [seq[per * a + s] for a in range(n)]
        out.append({k: sum(x[k] for x in sel) / n for k in ["ms"] + counters})
    return out


fetch = launches(os.path.join(src, TAG + "_heff_pmc_fetch.csv"), ["FETCH_SIZE"])
write = launches(os.path.join(src, TAG + "_heff_pmc_write.csv"), ["WRITE_SIZE"])
util = launches(os.path.join(src, TAG + "_heff_pmc_util.csv"), ["SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])
rd = [2 * 1024 * f["FETCH_SIZE"] for f in fetch]
wr = [1024 * w["WRITE_SIZE"] for w in write]
traffic = {
    "shape": {"D": 1024, "d": 16, "M": 32},
    "unit": "bytes per H_eff apply (C4 interior site, finite-state-machine MPO core: block-sparse W stage, S3 with the contraction split in 8)",
    "command": "tools/profile_" + TAG + "_heff.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/heff_fsm_probe.py 1024 16 32 3 (and WRITE_SIZE in its own pass)",
    "stages": NAMES,
    "ms": [f["ms"] for f in fetch],
    "read_bytes": rd,
    "write_bytes": wr,
    "total_bytes": sum(rd) + sum(wr),
    "algorithmic_bytes_B_H": 1.61e9,
    "earlier_in_round_2_S3_unsplit": {"read_bytes_S3": 179.4e9, "total_bytes": 259.6e9},
    "r01_total_bytes_dense_W_stage": 170.4e9,
    "note": "FETCH_SIZE doubled per the microarch guide; the counters sit on the fabric side of the per-XCD L2s (Infinity-Cache hits included). "
            "S3 unsplit reads 180 GB: its workgroups run 2048 K tiles each, drift apart and stop sharing operand strips in L2; "
            "split in 8 (256 K tiles per workgroup) it reads half of that and runs 2 % faster (tools/longk_probe.sh: 180 / 123 / 94 / 68 / 77 GB "
            "for 1 / 4 / 8 / 16 / 32 splits)",
}
json.dump(traffic, open("profiles/%s_heff_traffic.json" % TAG, "w"), indent=1)
mf = {"shape": traffic["shape"],
      "method": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); v_mfma_f64_16x16x4_f64 = 64 cycles; same command with "
                "--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE",
      "stages": {}}
for name, u in zip(NAMES, util):
    if u["GRBM_GUI_ACTIVE"] > 0:
        mf["stages"][name] = {"ms": u["ms"], "mfma_pipe_busy_frac": u["SQ_VALU_MFMA_BUSY_CYCLES"] / (u["GRBM_GUI_ACTIVE"] / 8 * 1024),
                              "clock_GHz_from_GRBM_GUI_ACTIVE": u["GRBM_GUI_ACTIVE"] / 8 / (u["ms"] * 1e-3) / 1e9}
tot = sum(u["ms"] for u in util)
ex = 0.75 * (2 * 4.398046511104e12 + 0.152 * 2.199023255552e12)
mf["apply"] = {"ms": tot, "executed_mfma_tflops": ex / (tot * 1e-3) / 1e12, "algorithmic_tflops_dense_count": 1.099511627776e13 / (tot * 1e-3) / 1e12}
json.dump(mf, open("profiles/%s_heff_mfma_util.json" % TAG, "w"), indent=1)
print(json.dumps({"total_GB": traffic["total_bytes"] / 1e9, "read_GB": [round(x / 1e9, 2) for x in rd], "write_GB": [round(x / 1e9, 2) for x in wr],
                  "ms": [round(x, 2) for x in traffic["ms"]], "busy": {k: round(v["mfma_pipe_busy_frac"], 3) for k, v in mf["stages"].items()}}))
