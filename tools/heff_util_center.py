"""Matrix-pipe utilisation of the H_eff apply's launches from one rocprofv3 PMC pass over tools/heff_center_probe.py
(SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE): busy fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) per
launch position of the periodic per-apply kernel sequence (the last `reps` applies), clock from GRBM_GUI_ACTIVE / 8 / duration.
    python tools/heff_util_center.py <util.csv> <name> <reps> <tag>  ->  profiles/r04_heff_mfma_util_<name>_<tag>.json"""
import collections, csv, json, sys

path, name, reps, tag = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
APPLY = ("zgemm", "k_copy2d", "k_transpose")
by = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    d = by.setdefault(int(r["Dispatch_Id"]), {"k": r["Kernel_Name"], "ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
seq = [v for _, v in sorted(by.items()) if any(a in v["k"] for a in APPLY)]
names = [v["k"].split("(")[0] for v in seq]
per = None
for p in range(1, len(seq) // reps + 1):
    tail = names[-reps * p:]
    if all(tail[i] == tail[i % p] for i in range(len(tail))) and any("zgemm" in n for n in tail[:p]):
        per = p
        break
assert per, "no periodic tail"
tail = seq[-reps * per:]
out = {"workload": name, "form": tag, "launches_per_apply": per,
       "method": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/heff_center_probe.py " + f"{name} {reps}",
       "launches": []}
tot = 0.0
for s in range(per):
    sel = [tail[per * a + s] for a in range(reps)]
    ms = sum(x["ms"] for x in sel) / reps
    busy = sum(x.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for x in sel) / reps
    gui = sum(x.get("GRBM_GUI_ACTIVE", 0.0) for x in sel) / reps
    tot += ms
    out["launches"].append({"kernel": sel[0]["k"].split("(")[0].replace("mitdvp::", ""), "ms": ms,
                            "mfma_pipe_busy_frac": busy / (gui / 8 * 1024) if gui > 0 else None,
                            "clock_GHz_from_GRBM_GUI_ACTIVE": gui / 8 / (ms * 1e-3) / 1e9 if ms > 0 else None})
out["apply_ms_under_the_profiler"] = tot
import os
rnd = os.environ.get("MITDVP_ROUND", "04")
json.dump(out, open(f"profiles/r{rnd}_heff_mfma_util_{name}_{tag}.json", "w"), indent=1)
print(json.dumps(out)[:1500])
