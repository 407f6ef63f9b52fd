"""Fabric-side traffic of the small-bond regime (C2) from two rocprofv3 PMC passes over tools/small_trace.py:
    python tools/c2_traffic.py <fetch.csv> <write.csv> <round>   ->  profiles/r<round>_c2_traffic.json
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH is doubled on gfx950 (MI355X_MICROARCH.md, HBM section: the counter reports half
the bytes of wide coalesced reads; the 8-byte agent-scope loads of k_small_site are an uncalibrated access width, so the
read side is an upper bound).  Counters are fabric-side requests of the XCD L2s: Infinity-Cache hits included."""
import collections, csv, json, sys

fetch_csv, write_csv, rnd = sys.argv[1], sys.argv[2], int(sys.argv[3])
L, d, D, M = 10, 10, 32, 6
NSWEEP = 24  # tools/small_trace.py: 2 + 10 time steps = 24 half-sweeps after the set-up


def total(path, counter, only=None):
    by = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            by[int(r["Dispatch_Id"])] += float(r["Counter_Value"])
            names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    sel = [k for k in by if only is None or only in names[k]]
    return sum(by[k] for k in sel), len(sel)


f_ss, n_ss = total(fetch_csv, "FETCH_SIZE", "k_small_site")
w_ss, _ = total(write_csv, "WRITE_SIZE", "k_small_site")
f_all, _ = total(fetch_csv, "FETCH_SIZE")
w_all, _ = total(write_csv, "WRITE_SIZE")
# applies per sweep (mean Krylov dimensions of the bench's C2 leg: 9.4 per site, 9.5 per bond exponential)
heff_per_sweep = L * 9.4
all_per_sweep = heff_per_sweep + (L - 1) * 9.5 + (L - 1)
bytes_ss = (2 * f_ss + w_ss) * 1024.0
B_H = 16.0 * (2 * D * d * D + 2 * D * D * M + M * d * d * M)
out = {
    "shape": {"L": L, "D": D, "d": d, "M": M},
    "workload": f"C2-like chain L={L} d={d} D={D} M={M} (tools/small_trace.py: 2 + 2 + 20 sweeps incl. set-up)",
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python3 tools/small_trace.py (and a second pass with --pmc WRITE_SIZE); tools/profile_r05.sh, tools/c2_traffic.py",
    "note": "FETCH_SIZE doubled (gfx950 reports 1/2 of wide coalesced reads; the 8-byte agent-scope loads of this kernel are an uncalibrated width: upper bound on the read side); fabric-side requests of the XCD L2s, Infinity-Cache hits included.  k_small_site carries every H_eff / K_eff apply, every environment update and both local exponentials of a sweep.",
    "k_small_site_dispatches": n_ss,
    "k_small_site_fetch_raw_KB": f_ss,
    "k_small_site_write_KB": w_ss,
    "k_small_site_bytes_per_sweep": bytes_ss / NSWEEP,
    "applies_per_sweep_all_kinds": all_per_sweep,
    "heff_applies_per_sweep": heff_per_sweep,
    "bytes_per_heff_apply": bytes_ss / NSWEEP / all_per_sweep,
    "bytes_per_heff_apply_note": "k_small_site traffic / (H_eff + K_eff applies + environment updates): an average over unlike applies",
    "algorithmic_bytes_per_heff_apply": B_H,
    "ratio_to_algorithmic": bytes_ss / NSWEEP / all_per_sweep / B_H,
    "all_kernels_fetch_raw_KB": f_all,
    "all_kernels_write_KB": w_all,
    "total_bytes_per_sweep": (2 * f_all + w_all) * 1024.0 / NSWEEP,
}
json.dump(out, open(f"profiles/r{rnd:02d}_c2_traffic.json", "w"), indent=1)
print(json.dumps(out)[:600])
