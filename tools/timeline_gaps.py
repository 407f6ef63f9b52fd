"""Busy fraction and gaps of a rocprofv3 --kernel-trace csv, per SEGMENT (the trace is cut at idle times >= cut_us: the
host-side pauses between bench.py's warm-up, timed region and profiled repeat), and the launch sequence from the middle
of the segment `which` (counted from the end: 1 = last = the profiled repeat, 2 = the timed region):
    python tools/timeline_gaps.py trace.csv [which=2] [dump_ms=0] [cut_us=400] [min_ms=30] [lo hi]
lo hi: look only at the part of the trace between these fractions of its length (e.g. 0.15 0.45: inside the timed region
when the repeat is as long as warm-up + timed region), as one segment."""
import csv, sys, collections
path = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dump_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
cut_us = float(sys.argv[4]) if len(sys.argv) > 4 else 400.0
min_ms = float(sys.argv[5]) if len(sys.argv) > 5 else 30.0
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
if len(sys.argv) > 7:
    t0, t1 = rows[0][0], rows[-1][1]
    a, b = t0 + (t1 - t0) * float(sys.argv[6]), t0 + (t1 - t0) * float(sys.argv[7])
    rows = [r for r in rows if a <= r[0] <= b]
    cut_us, which = 1e12, 1
segs, cur, cur_end = [], [rows[0]], rows[0][1]
for r in rows[1:]:
    if r[0] - cur_end >= cut_us * 1e3:
        segs.append(cur); cur = []
    cur.append(r); cur_end = max(cur_end, r[1])
segs.append(cur)
segs = [s for s in segs if s[-1][1] - s[0][0] >= min_ms * 1e6]
def short(n): return n.split("(")[0][-64:]
def stats(win, label):
    busy = 0; cur_end = win[0][0]; gaps = []
    for s, e, n in win:
        if s > cur_end: gaps.append((s - cur_end, n))
        busy += max(0, e - max(s, cur_end)); cur_end = max(cur_end, e)
    span = win[-1][1] - win[0][0]
    print("%s: %.1f ms, %d launches, busy %.3f, idle %.2f ms" % (label, span / 1e6, len(win), busy / span, sum(g for g, _ in gaps) / 1e6))
    h = collections.OrderedDict((k, 0) for k in ("<2us", "<5us", "<10us", "<20us", "<50us", ">=50us"))
    for g, _ in gaps:
        h["<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<20us" if g < 20000 else "<50us" if g < 50000 else ">=50us"] += g
    print("   idle time by gap size (ms):", {k: round(v / 1e6, 3) for k, v in h.items()})
    after = collections.Counter(); cnt = collections.Counter()
    for g, n in gaps:
        after[short(n)] += g; cnt[short(n)] += 1
    print("   idle time by the kernel that follows the gap (ms, gaps, mean us):")
    for k, v in after.most_common(10): print("   %8.3f %6d %6.1f  %s" % (v / 1e6, cnt[k], v / 1e3 / cnt[k], k))
    tot = collections.Counter(); num = collections.Counter()
    for s, e, n in win:
        tot[short(n)] += e - s; num[short(n)] += 1
    print("   kernel time (ms, share of the span, launches, mean us):")
    for k, v in tot.most_common(24): print("   %8.3f %5.1f%% %6d %7.1f  %s" % (v / 1e6, 100.0 * v / span, num[k], v / 1e3 / num[k], k))
for i, s in enumerate(segs): stats(s, "segment %d of %d" % (i + 1, len(segs)))
if dump_ms > 0 and len(segs) >= which:
    win = segs[-which]
    s0 = win[len(win) // 2][0]; prev = s0
    print("sequence from the middle of segment %d:" % (len(segs) - which + 1))
    for s, e, n in win[len(win) // 2:]:
        if s - s0 > dump_ms * 1e6: break
        print("%9.1f us  gap %6.1f  dur %7.1f  %s" % ((s - s0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, short(n)))
        prev = e
