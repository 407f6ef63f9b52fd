"""Fabric-side bytes per H_eff apply from two rocprofv3 PMC passes over tools/heff_center_probe.py (FETCH_SIZE, WRITE_SIZE in
KiB; FETCH doubled on gfx950 per MI355X_MICROARCH.md).  The kernels of the LAST `reps` applies are taken: every apply
issues the same sequence, found as the shortest period of the trailing kernel-name sequence.
    python tools/heff_traffic_center.py <fetch.csv> <write.csv> <name> <D> <d> <M> <reps> <tag>  ->  profiles/r04_heff_traffic_<name>_<tag>.json"""
import collections, csv, json, os, sys

fetch_csv, write_csv, name = sys.argv[1:4]
D, d, M, reps = (int(x) for x in sys.argv[4:8])
tag = sys.argv[8]
APPLY = ("zgemm", "k_copy2d", "k_transpose", "k_ident_dev")


def per_apply(path, counter):
    by = collections.OrderedDict()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            by.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0])[1] += float(r["Counter_Value"])
    seq = [(k, v) for _, (k, v) in sorted(by.items()) if any(a in k for a in APPLY)]
    names = [k.split("(")[0] for k, _ in seq]
    # period of the tail: the smallest p with the last reps * p names p-periodic
    for p in range(1, len(seq) // reps + 1):
        tail = names[-reps * p:]
        if all(tail[i] == tail[i % p] for i in range(len(tail))) and k_identless(tail[:p]):
            return sum(v for _, v in seq[-reps * p:]) / reps, p, tail[:p]
    raise SystemExit("no periodic tail found")


def k_identless(period):
    return any("zgemm" in n for n in period)


rd_kib, n1, kern = per_apply(fetch_csv, "FETCH_SIZE")
wr_kib, n2, _ = per_apply(write_csv, "WRITE_SIZE")
B_H = 16.0 * (2 * D * d * D + 2 * D * D * M + M * d * d * M)
out = {"shape": {"D": D, "d": d, "M": M}, "workload": name, "form": tag, "launches_per_apply": n1, "kernels_per_apply": kern,
       "unit": "bytes per H_eff apply at the centre of a canonical chain, the bench's MPO generator",
       "command": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE -- python3 tools/heff_center_probe.py {name} {reps} (and WRITE_SIZE in its own pass); tools/heff_traffic_center.py",
       "read_bytes": 2 * 1024 * rd_kib, "write_bytes": 1024 * wr_kib, "total_bytes": 2 * 1024 * rd_kib + 1024 * wr_kib,
       "algorithmic_bytes_B_H": B_H}
out["ratio_to_algorithmic"] = out["total_bytes"] / B_H
rnd = os.environ.get("MITDVP_ROUND", "04")
json.dump(out, open(f"profiles/r{rnd}_heff_traffic_{name}_{tag}.json", "w"), indent=1)
print(json.dumps(out))
