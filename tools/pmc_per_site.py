#!/usr/bin/env python3
"""Per-site counter table of tools/round4/r4_pmc_sizes.sh: one row per lattice, counters of the step's kernel (k_fused_ho) divided by
the sites one launch updates.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B: /opt/skills/guides/
MI355X_MICROARCH.md, HBM section), both sizes are KiB.  usage: pmc_per_site.py gpurun_out/r4_pmc_sizes"""
import glob, json, os, re, sys
root = sys.argv[1]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_profiles import parse_summary
rows = []
for d in sorted(glob.glob(os.path.join(root, "*x*x*"))):
    shape = [int(v) for v in os.path.basename(d).split("x")]
    sites = shape[0] * shape[1] * shape[2]
    kern = parse_summary(os.path.join(d, "pmc_summary.txt"))
    k = [v for n, v in kern.items() if "k_fused_ho" in n]
    if len(k) != 1:
        print(f"# {d}: {len(k)} k_fused_ho kernels in the summary: {list(kern)}"); continue
    c = k[0]
    try:
        b = json.loads(open(os.path.join(d, "bench.json")).read().strip().splitlines()[-1])
    except Exception:
        b = {"value": float("nan"), "roofline": {"frac": float("nan")}}
    rd, wr = c["FETCH_SIZE"] * 1024 * 2 / sites, c["WRITE_SIZE"] * 1024 / sites
    rows.append((os.path.basename(d), b["value"], b["roofline"]["frac"], rd, wr, (rd + wr) / 608.0, c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]),
                 c["SQ_INSTS_VALU"] / sites * 64, c["SQ_WAVE_CYCLES"] / sites, c["SQ_WAIT_INST_ANY"] / sites, c["SQ_BUSY_CYCLES"] / sites, c["GRBM_GUI_ACTIVE"] / sites * 1e3, c["SQ_WAVES"]))
print("lattice        MLUPS   frac  read B/site  write B/site  traffic/608  L2 hit  VALU/site  WAVE_CYCLES/site  WAIT_INST_ANY/site  BUSY_CYCLES/site  GUI_ACTIVE/1000 sites  waves")
for r in rows:
    print(f"{r[0]:13s} {r[1]:7.0f} {r[2]:6.3f} {r[3]:11.1f} {r[4]:13.1f} {r[5]:12.3f} {r[6]:7.3f} {r[7]:10.0f} {r[8]:17.1f} {r[9]:19.1f} {r[10]:17.2f} {r[11]:22.2f} {r[12]:6.0f}")
