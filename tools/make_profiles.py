#!/usr/bin/env python3
"""Copy the judged summaries of ONE tools/profile_round.sh run from gpurun_out/<tag>/ into profiles/ and
(re)write the entry of profiles/traffic.json that bench.py reports as roofline.traffic.

The traffic is computed from the COMMITTED summary profiles/<tag>_pmc_summary.txt alone (median FETCH_SIZE and
WRITE_SIZE of the kernel's dispatches of that one session), so it can be re-derived by hand:
    hbm bytes per launch = FETCH_SIZE[KiB] x 1024 x 2  +  WRITE_SIZE[KiB] x 1024
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and WRITE_SIZE are in KiB; on
gfx950 FETCH_SIZE reports exactly half of the bytes of a coalesced streaming read -- confirmed here on kernels
with a known byte count (k_pull, k_density: FETCH_SIZE*2 == 38*8 B/site), so reads are doubled; WRITE_SIZE is
exact.  FETCH_SIZE counts L2->fabric requests, Infinity-Cache hits included.
usage: tools/make_profiles.py <tag> <workload-key> <schedule> <kernel-name-prefix>[,<prefix>...]
Several prefixes = the kernels of one step (two-pass schedule): their per-launch medians are added.
"""
import json, os, re, shutil, sys


def parse_summary(path):
    """{kernel name: {counter: median}} of a tools/pmc_summary.py file."""
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"== (.*?)\s+\((\d+) dispatches\)", line)
        if m:
            cur = out.setdefault(m.group(1).strip(), {})
            continue
        m = re.match(r"\s+(\S+)\s+mean\s+(\S+)\s+median\s+(\S+)", line)
        if m and cur is not None:
            cur[m.group(1)] = float(m.group(3))
    return out


def traffic_from_summary(path, prefixes):
    kern = parse_summary(path)
    fetch = write = 0.0
    for pre in prefixes:
        hits = [v for k, v in kern.items() if k.startswith(pre) or (" " + pre) in (" " + k)[:48]]
        if len(hits) != 1:
            raise SystemExit(f"{path}: {len(hits)} kernels match '{pre}': {list(kern)}")
        fetch += hits[0]["FETCH_SIZE"]
        write += hits[0]["WRITE_SIZE"]
    return fetch * 1024 * 2, write * 1024


def kernel_avg_ms(path, prefixes):
    """Sum over the step's kernels of the AverageNs column of a rocprofv3 --kernel-trace --stats summary."""
    import csv
    total = 0.0
    rows = list(csv.DictReader(open(path)))
    for pre in prefixes:
        hits = [r for r in rows if r["Name"].startswith(pre) or (" " + pre) in (" " + r["Name"])[:48]]
        if len(hits) != 1:
            raise SystemExit(f"{path}: {len(hits)} kernels match '{pre}'")
        total += float(hits[0]["AverageNs"]) * 1e-6
    return total


if __name__ == "__main__":
    tag, key, schedule, kname = sys.argv[1:5]
    src, dst = os.path.join("gpurun_out", tag), "profiles"
    os.makedirs(dst, exist_ok=True)
    for f in ("kernel_stats.csv", "pmc_summary.txt", "bench.json", "kernel_timed.json"):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
    summary = os.path.join(dst, f"{tag}_pmc_summary.txt")
    read_b, write_b = traffic_from_summary(summary, kname.split(","))
    path = os.path.join(dst, "traffic.json")
    t = json.load(open(path)) if os.path.exists(path) else {}
    stats = os.path.join(dst, f"{tag}_kernel_stats.csv")
    timed = os.path.join(dst, f"{tag}_kernel_timed.json")
    avg_all = round(kernel_avg_ms(stats, kname.split(",")), 4) if os.path.exists(stats) else None
    # round 4: the average over the bench's timed launches (tools/trace_timed_avg.py) where the session has one; the --stats
    # average over ALL launches (placement probes on discarded allocations included) is kept beside it
    avg = json.load(open(timed))["avg_timed_ms"] if os.path.exists(timed) else avg_all
    t[f"{key}|{schedule}"] = {"hbm_bytes_per_launch": read_b + write_b, "read_bytes": read_b, "write_bytes": write_b,
                              "kernel_avg_ms": avg, "kernel_avg_all_launches_ms": avg_all,
                              "kernel_stats": stats,
                              "source": summary, "kernel": kname,
                              "rule": "median FETCH_SIZE KiB x1024 x2 (gfx950 half-count, calibrated on k_pull/k_density) + median WRITE_SIZE KiB x1024, from the source file alone"}
    json.dump(t, open(path, "w"), indent=1)
    print(json.dumps(t[f"{key}|{schedule}"]))
