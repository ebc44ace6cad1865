"""Kernel timeline of the replayed learner updates in a rocprofv3 --kernel-trace CSV (bench.py --mode train): one update =
the window from the end of an update's last kernel (adam_update_kernel) to the end of the next one's; inside a replayed
group the next update's draw / gather / scan run on the side stream beside the current update's tail, so they show in
the window of the update BEFORE the one that consumes them.  Prints one window from the middle of the run, the median
period over all windows and the median duration of every kernel.  Usage:
  rocprofv3 --kernel-trace -d out -o p -f csv -- python3 bench.py --mode train --steps 300 --warmup 100 --no-cpu-baseline --no-other-modes
  python scripts/timeline_update.py out/**/p_kernel_trace.csv"""
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
ends = [e[1] for e in ev if "adam_update_kernel" in e[2]]
if len(ends) < 8:
    sys.exit("fewer than 8 updates in the trace")
periods = [(b - a) / 1e3 for a, b in zip(ends[:-1], ends[1:])]
typical = [p for p in periods if p < 3 * statistics.median(periods)]      # (episode boundaries: rollout + store in between)
print(f"{len(ends)} updates; period between the ends of consecutive updates: median {statistics.median(periods):.1f} us, "
      f"mean of the {len(typical)} ordinary ones {statistics.mean(typical):.1f} us (the first update of a replayed group does its "
      f"draw / gather / scan itself, the others find them done)")
mid = len(periods) // 2
print("periods of 44 consecutive updates from the middle of the run (us; a replayed group shows as a repeating pattern): "
      + " ".join(f"{p:.0f}" for p in periods[mid - 22:mid + 22]))
# a window whose period is the median one
k = min(range(len(periods) // 3, 2 * len(periods) // 3), key=lambda i: abs(periods[i] - statistics.median(periods)))
w0, w1 = ends[k], ends[k + 1]
win = [(s, e, n, q) for s, e, n, q in ev if w0 - 100 <= s and e <= w1]
print(f"window of update {k + 1} (t = 0 at the previous update's last kernel end; start us, duration us, queue, kernel):")
for s, e, n, q in win:
    print(f"  {(s - w0) / 1e3:7.1f} {(e - s) / 1e3:6.1f}  q{q}  {n[:78]}")
busy = sorted((s, e) for s, e, _, _ in win)
idle, cur = 0, w0
for s, e in busy:
    if s > cur:
        idle += s - cur
    cur = max(cur, e)
print(f"  no kernel running for {idle / 1e3:.1f} us of the window's {(w1 - w0) / 1e3:.1f} us; sum of kernel time "
      f"{sum(e - s for s, e, _, _ in win) / 1e3:.1f} us")
# median duration per kernel name over the whole trace (update kernels only: those seen in the window)
names = []
for _, _, n, _ in win:
    if n not in names:
        names.append(n)
print("median duration over the trace:")
for n in names:
    d = [(e - s) / 1e3 for s, e, nn, _ in ev if nn == n]
    print(f"  {statistics.median(d):6.1f} us x {len(d):5d}  {n[:78]}")
