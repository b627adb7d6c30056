#!/usr/bin/env python3
"""Copy the judged summaries of a tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/
and (re)write profiles/traffic.json, the per-launch HBM traffic bench.py reports.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a coalesced streaming read -- confirmed
here on kernels with a known byte count (k_pull, k_density: FETCH_SIZE*2 == 38*8 B/site), so reads
are doubled; WRITE_SIZE is exact.  FETCH_SIZE counts L2->fabric requests, Infinity-Cache hits included.
usage: tools/make_profiles.py <tag> <workload-key> <schedule> <kernel-name-prefix>[,<prefix>...]
Several prefixes = the kernels of one step (two-pass schedule): their per-launch medians are added.
"""
import csv, glob, json, os, shutil, sys
tag, key, schedule, kname = sys.argv[1:5]
src = os.path.join("gpurun_out", tag)
dst = "profiles"
os.makedirs(dst, exist_ok=True)
for f in ("kernel_stats.csv", "pmc_summary.txt", "bench.json"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
med = {}
for kn in kname.split(","):
    vals = {}
    for f in glob.glob(os.path.join(src, "*", "*", "*counter_collection.csv")):
        for row in csv.DictReader(open(f)):
            if row["Kernel_Name"].startswith(kn) or (" " + kn) in row["Kernel_Name"][:40]:
                vals.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    for k, v in vals.items():
        med[k] = med.get(k, 0.0) + sorted(v)[len(v) // 2]
read_b = med["FETCH_SIZE"] * 1024 * 2
write_b = med["WRITE_SIZE"] * 1024
path = os.path.join(dst, "traffic.json")
t = json.load(open(path)) if os.path.exists(path) else {}
t[f"{key}|{schedule}"] = {"hbm_bytes_per_launch": read_b + write_b, "read_bytes": read_b, "write_bytes": write_b,
                          "source": f"profiles/{tag}_pmc_summary.txt", "kernel": kname,
                          "correction": "FETCH_SIZE KiB x2 (gfx950 half-count, calibrated on k_pull/k_density), WRITE_SIZE KiB x1"}
json.dump(t, open(path, "w"), indent=1)
print(json.dumps(t[f"{key}|{schedule}"]))
