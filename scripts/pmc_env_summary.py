"""Summarise rocprofv3 --pmc passes of the env-step kernels into the JSON that bench.py's `roofline.traffic` reads
(profiles/r*_env_step_pmc.json).  One directory per (E, counter) pass, named <out>/E<E>_<COUNTER>/ and produced by

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE -d <out>/E4096_FETCH_SIZE -o p -f csv -- python3 bench.py --mode env --batch-envs 4096 \
      --steps 30 --warmup 5 --no-cpu-baseline          (and WRITE_SIZE; and --batch-envs 4194304)

Counters come in KB; on gfx950 FETCH_SIZE reports half of the bytes of a coalesced streaming read
(MI355X_MICROARCH.md, HBM section), so traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per dispatch (medians).
Usage: python scripts/pmc_env_summary.py <out> <round-tag> > profiles/<round-tag>_env_step_pmc.json"""
import csv
import glob
import json
import os
import statistics
import sys

out, tag = sys.argv[1], sys.argv[2]
per_env = len(sys.argv) > 3 and sys.argv[3] == "per-env"
J, R = 3, 4
B_ALG = 89 + ((8 * (6 * R + 3 * J + J * R) + J * R) if per_env else 0)   # bench.py: algorithmic bytes per env-step
runs = {}
# dispatches are told apart by kernel name AND grid size (threads): a per-env pass runs bench.py once at its default
# E = 4096 and picks up the 2^22-env launches of the same kernel from bench.py's large_batch section
for E, kern in ((4096, "env_step_kernel" if per_env else "env_step_slots_kernel"), (4194304, "env_step_kernel")):
    rec = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        files = glob.glob(os.path.join(out, f"E{E}_{ctr}", "**", "*counter_collection.csv"), recursive=True) or \
            glob.glob(os.path.join(out, f"ALL_{ctr}", "**", "*counter_collection.csv"), recursive=True)
        vals = []
        for f in files:
            for r in csv.DictReader(open(f)):
                big = int(r["Grid_Size"]) >= (1 << 22)
                if kern + "<" in r["Kernel_Name"] and r["Counter_Name"] == ctr and big == (E > (1 << 20)):
                    vals.append(float(r["Counter_Value"]))
        if not vals:
            continue
        rec[f"{ctr}_KB_median"], rec[f"{ctr}_n"] = statistics.median(vals), len(vals)
    if len(rec) == 4:
        rec["traffic_bytes_per_launch"] = int((2 * rec["FETCH_SIZE_KB_median"] + rec["WRITE_SIZE_KB_median"]) * 1024)
        rec["algorithmic_bytes_per_launch"] = E * B_ALG
        rec["traffic_over_algorithmic"] = round(rec["traffic_bytes_per_launch"] / (E * B_ALG), 4)
        rec["kernel_name"] = kern
        runs[str(E)] = rec
# the many-step launch of the rollout (macjd_env_step_many: T x E work items in the lane kernel): passes MANY_<ctr> of
# `bench.py --mode rollout`, dispatches recognised by their grid size
if not per_env:
    T_STEPS, E0 = 100, 4096
    # compulsory bytes (bench.py, B_many): actions read + reward / components / terminated written per work item; step
    # counter, episode index (read) and track (written) once per env
    B_MANY = 8 * J + 4 + 12 + 1 + (8 + R) / T_STEPS
    rec = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(os.path.join(out, f"MANY_{ctr}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "env_step_kernel<" in r["Kernel_Name"] and r["Counter_Name"] == ctr and \
                        T_STEPS * E0 <= int(r["Grid_Size"]) < T_STEPS * E0 + 256:
                    vals.append(float(r["Counter_Value"]))
        if vals:
            rec[f"{ctr}_KB_median"], rec[f"{ctr}_n"] = statistics.median(vals), len(vals)
    if len(rec) == 4:
        rec["traffic_bytes_per_launch"] = int((2 * rec["FETCH_SIZE_KB_median"] + rec["WRITE_SIZE_KB_median"]) * 1024)
        rec["algorithmic_bytes_per_launch"] = int(T_STEPS * E0 * B_MANY)
        rec["traffic_over_algorithmic"] = round(rec["traffic_bytes_per_launch"] / (T_STEPS * E0 * B_MANY), 4)
        rec["kernel_name"] = "env_step_kernel (many-step launch: 100 steps x 4096 envs)"
        rec["bytes_per_env_step_algorithmic"] = round(B_MANY, 2)
        runs[f"many_{E0}"] = rec
print(json.dumps({
    "round": tag, "kernel": f"macjd::env_step_kernel<{J},{R},per-env>" if per_env else f"macjd::env_step_slots_kernel<{J},{R}>",
    "workload": f"{J} jammers / {R} radars, Philox in-kernel, info outputs on" + (", per-env scenario tables" if per_env else ""),
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (python3 bench.py --mode env --batch-envs E "
              "--steps 30 --warmup 5); per-dispatch medians; unit KB; gfx950 correction per MI355X_MICROARCH.md section HBM: "
              "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 (scripts/pmc_env_summary.py)",
    "bytes_per_env_step_algorithmic": B_ALG,
    "note": ("lane-per-env kernel streaming its env's table column" if per_env else
             "E = 4096: (env x slot) kernel; E = 2^22: lane-per-env kernel (the default from 2^16 envs)"),
    "runs": runs}, indent=1))
