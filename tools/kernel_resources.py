#!/usr/bin/env python3
"""Registers / spills / LDS of every kernel of one csrc/*.hip, from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: tools/kernel_resources.py velocyto.py_amd/csrc/coldeltacor.hip [name filter] [-D...]"""
import re, subprocess, sys
src = sys.argv[1]
flt = [a for a in sys.argv[2:] if not a.startswith("-")]
defs = [a for a in sys.argv[2:] if a.startswith("-")]
r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", *defs, "-c", src, "-o", "/dev/null",
                    "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
cur = None
rows = {}
for line in r.stderr.splitlines():
    m = re.search(r"remark: (.*?): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
    elif cur:
        rows[cur][k] = v
for name, d in rows.items():
    if flt and not all(f in name for f in flt):
        continue
    print(f"{name}\n    VGPRs {d.get('VGPRs')}  AGPRs {d.get('AGPRs')}  SGPRs {d.get('SGPRs')}  spill V {d.get('VGPRs Spill')} S {d.get('SGPRs Spill')}  "
          f"scratch {d.get('ScratchSize [bytes/lane]')} B/lane  occupancy {d.get('Occupancy [waves/SIMD]')}  LDS {d.get('LDS Size [bytes/block]')}")
