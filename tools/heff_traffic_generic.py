"""Fabric-side bytes per H_eff apply at any shape from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; KiB units,
FETCH doubled on gfx950 per MI355X_MICROARCH.md): every kernel dispatched by `tools/heff_fsm_probe.py D d M reps` after the
first (warm-up) apply is summed and divided by the number of remaining applies.
    python tools/heff_traffic_generic.py <fetch.csv> <write.csv> D d M reps  ->  profiles/r03_heff_traffic_D{D}_d{d}_M{M}.json"""
import collections, csv, json, sys

fetch_csv, write_csv = sys.argv[1], sys.argv[2]
D, d, M, reps = (int(x) for x in sys.argv[3:7])


def per_apply(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(r["Dispatch_Id"], [r["Kernel_Name"], 0.0])[1] += float(r["Counter_Value"])
    seq = [(k, v) for k, v in by.values() if "zgemm" in k or "k_copy2d" in k]
    n_per = len(seq) // (reps + 1)  # the probe runs one warm-up apply + reps timed ones through the same call
    seq = seq[n_per:]
    return sum(v for _, v in seq) / reps, n_per


rd_kib, n1 = per_apply(fetch_csv, "FETCH_SIZE")
wr_kib, n2 = per_apply(write_csv, "WRITE_SIZE")
B_H = 16.0 * (2 * D * d * D + 2 * D * D * M + M * d * d * M)
out = {"shape": {"D": D, "d": d, "M": M}, "launches_per_apply": n1,
       "unit": "bytes per H_eff apply, interior site, the bench's finite-state-machine MPO core",
       "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/heff_fsm_probe.py {D} {d} {M} {reps} (and WRITE_SIZE in its own pass); tools/heff_traffic_generic.py",
       "read_bytes": 2 * 1024 * rd_kib, "write_bytes": 1024 * wr_kib, "total_bytes": 2 * 1024 * rd_kib + 1024 * wr_kib,
       "algorithmic_bytes_B_H": B_H}
out["ratio_to_algorithmic"] = out["total_bytes"] / B_H
json.dump(out, open(f"profiles/r03_heff_traffic_D{D}_d{d}_M{M}.json", "w"), indent=1)
print(json.dumps(out))
