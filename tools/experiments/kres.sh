#!/bin/bash
# Per-kernel register / scratch / occupancy summary of one .hip file (compiler remarks).
f=$1
/opt/rocm/bin/hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -I include -I ratrack_amd/csrc -c "$f" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import sys,re
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r"Function Name: (\S+)", line)
    if m: cur={"name":m.group(1)}; rows.append(cur); continue
    for key in ["VGPRs","AGPRs","ScratchSize [bytes/lane]","Occupancy [waves/SIMD]","SGPRs","LDS Size [bytes/block]"]:
        m=re.search(re.escape(key)+r": (\d+)", line)
        if m and cur is not None and key not in cur: cur[key]=m.group(1)
import subprocess
for r in rows:
    name=subprocess.run(["c++filt",r["name"]],capture_output=True,text=True).stdout.strip()[:90]
    print("%-92s vgpr %4s agpr %3s scratch %4s occ %2s lds %6s" % (name, r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
'
