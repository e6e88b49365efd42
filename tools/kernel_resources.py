#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage), e.g.
    python tools/kernel_resources.py attention_h2.hip [-fno-slp-vectorize]
Used while tuning: a kernel that spills (scratch > 0) or drops a wave per SIMD shows up here before it shows up in a timing."""
import os, re, subprocess, sys
HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "clip_decontamination_amd", "csrc")
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from clip_decontamination_amd.build import EXTRA_FLAGS       # the per-unit flags of the real build (e.g. -fno-slp-vectorize)
src = sys.argv[1]
cmd = ["hipcc", "-O3", "-fPIC", "-std=c++17", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-unused-result", *EXTRA_FLAGS.get(src, []), *sys.argv[2:], "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"]
log = subprocess.run(cmd, cwd=HERE, capture_output=True, text=True).stderr
rows, cur = [], {}
for l in log.splitlines():
    m = re.search(r"remark:\s+(.*?)\s*\[-Rpass", l)
    if not m:
        if "error" in l: print(l)
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        if cur: rows.append(cur)
        cur = {"name": t.split(": ")[1]}
    else:
        k, _, v = t.partition(":"); cur[k.strip()] = v.strip()
if cur: rows.append(cur)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    n = re.sub(r"\(sg::\w+(, int)*\)$", "", n.replace("sg::", ""))
    print(f"{n:78s} vgpr {r.get('VGPRs'):>4} agpr {r.get('AGPRs'):>4} scratch {r.get('ScratchSize [bytes/lane]'):>4} occ {r.get('Occupancy [waves/SIMD]')}")
