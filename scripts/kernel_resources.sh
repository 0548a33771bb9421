#!/bin/bash
# Register / scratch / LDS use of every kernel of engine.hip (compile-only, no GPU): bash scripts/kernel_resources.sh [extra hipcc flags]
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -c trajopt_amd/csrc/engine.hip -o /tmp/engine_res.o \
  -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c '
import re, sys, subprocess
rows, cur = [], None
for line in sys.stdin:
    if "error" in line: print(line.rstrip())
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r":\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+) \[-Rpass", line)
    if m and cur is not None: cur[m.group(1).strip()] = m.group(2)
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print("%-48s %5s %5s %5s %7s %4s %6s" % ("kernel", "SGPR", "VGPR", "AGPR", "scratch", "occ", "LDS"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("void tohc::", "")
    print("%-48s %5s %5s %5s %7s %4s %6s" % (n[:48], r.get("TotalSGPRs"), r.get("VGPRs"), r.get("AGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS Size")))
'
