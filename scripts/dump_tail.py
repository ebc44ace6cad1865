"""The kernel launches of the LAST n_updates learner updates of a rocprofv3 --kernel-trace CSV, and whatever ran in front of
them, with start time / duration / queue / gap (a driver-style run's timed region — `bench.py --steps 20 --warmup 5` —
is short enough to read as a whole; update kernels are summarised one line per update):
    python scripts/dump_tail.py p_kernel_trace.csv [n_updates] [rows in front]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n_upd = int(sys.argv[2]) if len(sys.argv) > 2 else 20
front = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
adam = [i for i, e in enumerate(ev) if "adam_update_kernel" in e[2]]
first = adam[-n_upd - 1] + 1 if len(adam) > n_upd else 0
lo, hi = max(0, first - front), adam[-1] + 1
t0 = ev[first][0]
UPD = ("adam_", "wgrad_", "mixer_fused", "qheads_pair", "qhead_taken", "qhead_double", "gru_sequence", "gather_rows", "td_mask", "td_loss",
       "sample_episodes")
prev_end = ev[lo][0]
upd_start = None
for s, e, name, q in ev[lo:hi]:
    is_upd = any(u in name for u in UPD) and s >= t0
    if is_upd:
        upd_start = s if upd_start is None else upd_start
        if "adam_update_kernel" in name:
            print(f"{(upd_start - t0) / 1e3:9.1f} {(e - upd_start) / 1e3:7.1f}        one update (its kernels from first start to Adam's end)")
            upd_start = None
    else:
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{q}  gap {(s - prev_end) / 1e3:7.1f}  {name[:90]}")
    prev_end = max(prev_end, e)
print(f"timed region (first launch after the previous update to the last Adam): {(ev[hi - 1][1] - t0) / 1e3:.1f} us")
