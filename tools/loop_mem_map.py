#!/usr/bin/env python3
"""Where the vector-memory instructions sit in the steady-state loop of the hand-over kernel: prints the loop as a string
of phases -- runs of global loads (L), global stores (S), and the number of VALU / LDS instructions between them -- from
the device assembly (hipcc --cuda-device-only -S).  usage: tools/loop_mem_map.py file.s [mode 0|1] [rag 0|1]"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rag = int(sys.argv[3]) if len(sys.argv) > 3 else 0
start = [i for i, l in enumerate(lines) if re.match(r"^_Z10k_fused_hoILi4ELi%dELb%dEE.*:" % (mode, rag), l)][0]
end = [i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")][0]
body = lines[start:end]
labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
loops = []
for i, l in enumerate(body):
    m = re.match(r"\s+s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
cands = [lp for lp in loops if sum("global_store" in l for l in body[lp[0]:lp[1]]) >= 38 and sum("global_load" in l for l in body[lp[0]:lp[1]]) >= 38]
lo, hi = min(cands, key=lambda lp: lp[1] - lp[0])
out = []
cnt = {"v": 0, "d": 0}
def flush():
    if cnt["v"] or cnt["d"]:
        out.append("[%dv%s]" % (cnt["v"], (" %dd" % cnt["d"]) if cnt["d"] else ""))
    cnt["v"] = cnt["d"] = 0
for l in body[lo:hi]:
    t = l.strip().split()[0] if l.strip() else ""
    if t.startswith("global_load"): flush(); out.append("L")
    elif t.startswith("global_store"): flush(); out.append("S")
    elif t.startswith("s_barrier"): flush(); out.append("|BAR|")
    elif t.startswith("s_waitcnt") and "vmcnt" in l: flush(); out.append("w%s" % re.search(r"vmcnt\((\d+)\)", l).group(1))
    elif t.startswith("s_cbranch") or t.startswith("s_branch"): flush(); out.append("/")
    elif t.startswith("v_"): cnt["v"] += 1
    elif t.startswith("ds_"): cnt["d"] += 1
flush()
s = " ".join(out)
s = re.sub(r"(?:L ){2,}L", lambda m: "L*%d" % m.group(0).count("L"), s)
s = re.sub(r"(?:S ){2,}S", lambda m: "S*%d" % m.group(0).count("S"), s)
print("loop %d instructions:" % (hi - lo), s)
