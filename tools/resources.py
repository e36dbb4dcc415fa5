#!/usr/bin/env python3
"""Per-kernel register / scratch report of the hot kernels (hipcc -Rpass-analysis=kernel-resource-usage); `make resources`.

    python tools/resources.py [extra hipcc flags] [translation units, e.g. gadapt_tu_bwd_target]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
UNITS = ["gadapt_tu_fwd", "gadapt_tu_bwd_target", "gadapt_tu_bwd_source", "gadapt_tu_smallmesh"]   # the hot kernels' translation units
flags = [a for a in sys.argv[1:] if a.startswith("-")]
units = [a for a in sys.argv[1:] if not a.startswith("-")] or UNITS
from concurrent.futures import ThreadPoolExecutor


def remarks(unit):
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude", "-Ig_adaptivity_amd/csrc", *flags, "-c", "-o", "/dev/null",
           f"g_adaptivity_amd/csrc/{unit}.hip", "-Rpass-analysis=kernel-resource-usage"]
    return subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True).stderr


with ThreadPoolExecutor(max_workers=4) as ex:
    out = "\n".join(ex.map(remarks, units))
cur, rec = None, {}
for line in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur, rec = t.split(":", 1)[1].strip(), {}
        try:                                                 # demangled: the template arguments tell the instantiations apart
            cur = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip().split("(")[0] or cur
        except OSError:
            pass
    elif cur and ":" in t:
        k, v = t.rsplit(":", 1)
        rec[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size") and ("grand_" in cur or "wide" in cur or "smallmesh" in cur):
            print("%-78s VGPR %4s  AGPR %3s  scratch %5s  spill %4s  waves/SIMD %s" % (
                cur, rec.get("VGPRs"), rec.get("AGPRs"), rec.get("ScratchSize [bytes/lane]"), rec.get("VGPRs Spill"),
                rec.get("Occupancy [waves/SIMD]")))
