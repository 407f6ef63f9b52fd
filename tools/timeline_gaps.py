"""Busy fraction and gaps of a rocprofv3 --kernel-trace csv (last `frac` of the trace = steady sweeps), and the launch
sequence of a window: python tools/timeline_gaps.py trace.csv [frac] [dump_ms]"""
import csv, sys, collections
path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.4
dump_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
lo = t1 - (t1 - t0) * frac
win = [r for r in rows if r[0] >= lo]
busy = 0; cur_end = win[0][0]; gaps = []
for s, e, n in win:
    if s > cur_end: gaps.append((s - cur_end, n))
    busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
span = win[-1][1] - win[0][0]
print("window %.1f ms, %d launches, busy %.3f, gap total %.2f ms" % (span / 1e6, len(win), busy / span, sum(g for g, _ in gaps) / 1e6))
h = collections.Counter()
for g, _ in gaps:
    h["<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<20us" if g < 20000 else "<50us" if g < 50000 else ">=50us"] += g
print("gap time by size (ms):", {k: round(v / 1e6, 3) for k, v in h.items()})
after = collections.Counter()
for g, n in gaps:
    if g >= 10000: after[n.split("(")[0][-60:]] += g
print("gaps >= 10 us by the kernel that follows (ms):")
for k, v in after.most_common(12): print("   %8.3f  %s" % (v / 1e6, k))
if dump_ms > 0:
    s0 = win[len(win) // 2][0]
    prev = s0
    for s, e, n in win[len(win) // 2:]:
        if s - s0 > dump_ms * 1e6: break
        print("%9.1f us  gap %6.1f  dur %7.1f  %s" % ((s - s0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, n.split("(")[0][-70:]))
        prev = e
