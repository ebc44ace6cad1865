"""Summarise the kernel timeline of the LAST replayed train step in a rocprofv3 --kernel-trace CSV:
per-queue busy time, the wall span of the step, and the longest gaps.  Usage:
  rocprofv3 --kernel-trace -d out -o p -f csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-gemm-tuning
  python scripts/timeline_update.py out/**/p_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows))
# the timed region ends with the last gather_rows_kernel .. adam_update_kernel pattern: take the last complete update
idx = [i for i, e in enumerate(ev) if "gather_rows_kernel" in e[2]]
bumps = [i for i, e in enumerate(ev) if "adam_update_kernel" in e[2]]
lo = idx[-2] if len(idx) >= 2 and idx[-1] > bumps[-1] else idx[-1]
hi = [b for b in bumps if b > lo][0]
step = ev[lo:hi + 1]
t0 = step[0][0]
print(f"update: {len(step)} kernels, span {(step[-1][1] - t0) / 1e3:.1f} us, sum of kernel time {sum(e[1] - e[0] for e in step) / 1e3:.1f} us")
queues = {}
for s, e, n, q in step:
    queues.setdefault(q, []).append((s, e, n))
for q, lst in queues.items():
    print(f" queue {q}: {len(lst)} kernels, busy {sum(e - s for s, e, _ in lst) / 1e3:.1f} us, from {(lst[0][0] - t0) / 1e3:.1f} to {(lst[-1][1] - t0) / 1e3:.1f} us")
# global coverage: time when NO kernel is running
pts = sorted((s, e) for s, e, _, _ in step)
idle, cur_end, gaps = 0, pts[0][0], []
for s, e in pts:
    if s > cur_end:
        idle += s - cur_end
        gaps.append((s - cur_end, cur_end - t0))
    cur_end = max(cur_end, e)
print(f" no kernel running: {idle / 1e3:.1f} us in {len(gaps)} gaps; largest: " + ", ".join(f"{g / 1e3:.1f}us@{at / 1e3:.0f}" for g, at in sorted(gaps, reverse=True)[:6]))
print(" timeline (start us, dur us, queue, kernel):")
for s, e, n, q in step:
    print(f"  {(s - t0) / 1e3:7.1f} {(e - s) / 1e3:6.1f}  q{q}  {n[:70]}")

# median over ALL complete updates in the trace of the idle time in front of each kernel on its own queue
# (one update's timeline is a single sample; a gap that shows in the median is real)
import statistics
starts = [i for i in idx if any(b > i for b in bumps)]
per_pos = {}
for lo_i in starts:
    hi_i = [b for b in bumps if b > lo_i][0]
    upd = ev[lo_i:hi_i + 1]
    if len(upd) != len(step):
        continue
    last_end = {}
    for pos, (s, e, n, q) in enumerate(upd):
        if q in last_end:
            per_pos.setdefault(pos, []).append((s - last_end[q]) / 1e3)
        last_end[q] = e
print(f" median idle time before each kernel on its queue over {len(per_pos.get(1, []))} updates (only > 3 us shown):")
for pos, gaps_ in sorted(per_pos.items()):
    med = statistics.median(gaps_)
    if med > 3.0:
        print(f"  #{pos:2d} {med:6.1f} us before {step[pos][2][:60]}")

# update-to-update period: start of one update's first kernel to the start of the next one's (the host-side cost of a
# step — index upload, graph launch — shows here as the difference between the period and the span)
firsts = [ev[i][0] for i in starts]
ends = [ev[[b for b in bumps if b > i][0]][1] for i in starts]
per = [(b - a) / 1e3 for a, b in zip(firsts[:-1], firsts[1:])]
gap = [(firsts[k + 1] - ends[k]) / 1e3 for k in range(len(starts) - 1)]
span = [(e - s) / 1e3 for s, e in zip(firsts, ends)]
if per:
    print(f" over {len(per)} consecutive updates: median period {statistics.median(per):.1f} us, median span {statistics.median(span):.1f} us, "
          f"median gap between an update's last kernel and the next update's first {statistics.median(gap):.1f} us")
    # what runs in a typical gap
    k = len(starts) // 2
    inside = [(s, e, n, q) for s, e, n, q in ev if ends[k] <= s < firsts[k + 1]]
    print(f" kernels between update {k} and {k + 1}: " + (", ".join(f"{n[:40]} ({(e - s) / 1e3:.1f} us)" for s, e, n, q in inside) or "none"))

# raw window: every kernel from the end of one update's last launch to the end of the next one's (a branch that starts
# before the gather — the scan — is cut off by the per-update view above)
if len(starts) > 4:
    k = len(starts) // 2 + 1
    w0, w1 = ends[k - 1], ends[k]
    print(f" raw window of update {k} (t = 0 at the previous update's last kernel end):")
    for s_, e_, n_, q_ in ev:
        if w0 - 2000 <= s_ <= w1:
            print(f"  {(s_ - w0) / 1e3:7.1f} {(e_ - s_) / 1e3:6.1f}  q{q_}  {n_[:70]}")
