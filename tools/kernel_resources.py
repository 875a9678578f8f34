#!/usr/bin/env python3
"""Print a table of per-kernel register / LDS / occupancy figures from hipcc's kernel-resource-usage remarks."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "plspm-python_amd", "csrc", "plspm_hip.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value",
       "-Rpass-analysis=kernel-resource-usage", src, "-o", "/dev/null"]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}; rows.append(cur); continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = int(m.group(2))
def demangle(n):
    return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
print("%-58s %5s %5s %5s %6s %6s %4s" % ("kernel", "SGPR", "VGPR", "AGPR", "spill", "LDS", "occ"))
for r in rows:
    print("%-58s %5d %5d %5d %6d %6d %4d" % (demangle(r["name"])[:58], r.get("TotalSGPRs", -1), r.get("VGPRs", -1), r.get("AGPRs", -1),
                                           r.get("VGPRs Spill", 0), r.get("LDS Size", 0), r.get("Occupancy", -1)))
