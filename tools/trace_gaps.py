"""Idle time between kernels in a rocprofv3 kernel trace (csv): total busy / span, the largest gaps and the kernels around them,
busy time per kernel name.   python tools/trace_gaps.py kernel_trace.csv [skip_first_ms]"""
import collections, csv, sys

rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))))
skip = float(sys.argv[2]) * 1e6 if len(sys.argv) > 2 else 0.0
t0 = rows[0][0]
rows = [r for r in rows if r[0] - t0 >= skip]
span = rows[-1][1] - rows[0][0]
busy = 0
gaps = collections.Counter()
gap_total = 0
end = rows[0][0]
prev = None
for s, e, n in rows:
    if s > end:
        g = s - end
        gap_total += g
        key = ((prev or "")[:50], n[:50])
        gaps[key] += g
        busy += e - s
    else:
        busy += max(0, e - max(s, end))
    if e > end:
        end, prev = e, n
print(f"span {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms ({busy / span:.3f}), idle {gap_total / 1e6:.2f} ms, kernels {len(rows)}")
for (a, b), g in gaps.most_common(12):
    print(f"  idle {g / 1e6:8.3f} ms  after {a!r} before {b!r}")
per = collections.Counter()
cnt = collections.Counter()
for s, e, n in rows:
    per[n[:70]] += e - s
    cnt[n[:70]] += 1
for n, v in per.most_common(14):
    print(f"  {v / 1e6:9.3f} ms {cnt[n]:7d}  {n}")
