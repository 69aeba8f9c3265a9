#!/usr/bin/env python3
"""Compile one .hip file for gfx950 and print a compact per-kernel resource table
(VGPR / AGPR / SGPR / spills / LDS / occupancy).  Usage: tools/kinfo.py file.hip [filter]"""
import re, subprocess, sys, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-save-temps=obj", "-I" + root + "/include",
       "-c", src, "-o", "/tmp/kinfo.o", "-Rpass-analysis=kernel-resource-usage"]
r = subprocess.run(cmd, capture_output=True, text=True)
if r.returncode:
    print(r.stderr); sys.exit(1)
cur = {}
rows = []
for line in r.stderr.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: +(.*?) \[-Rpass", line) or re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}; rows.append(cur)
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
def demangle(n):
    try: return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip()
    except Exception: return n
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'spill':>5} {'LDS':>7} {'occ':>3}  kernel")
for c in rows:
    n = demangle(c["name"])
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"\(.*", "", n)
    if flt and flt not in n: continue
    print(f"{c.get('VGPRs','?'):>5} {c.get('AGPRs','?'):>5} {c.get('TotalSGPRs','?'):>5} {c.get('VGPRs Spill','?'):>5} {c.get('LDS Size [bytes/block]','?'):>7} {c.get('Occupancy [waves/SIMD]','?'):>3}  {n}")
