#!/usr/bin/env python3
"""Diagnostic: per-basic-block instruction mix of one kernel in the gfx950 ISA of the hot-path source.

    python tools/isa_blocks.py <mangled-name-substring> [extra hipcc flags]
"""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
out = "/tmp/isa_blocks.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Iinclude", *sys.argv[2:], "-S",
                "--cuda-device-only", "-o", out, "g_adaptivity_amd/csrc/gadapt_kernels.hip"], cwd=ROOT, check=True)
text = open(out).read()
start = re.search(r"^(\S*%s\S*):" % re.escape(name), text, re.M)
body = text[start.end():text.index(".end_amdhsa_kernel", start.end())]
blocks, cur = [], ["entry", collections.Counter(), []]
for line in body.splitlines():
    t = line.strip()
    if not t or t.startswith(";") or t.startswith("."):
        if re.match(r"^\.LBB\d+_\d+:", t):
            blocks.append(cur)
            cur = [t.rstrip(":"), collections.Counter(), []]
        continue
    op = t.split()[0]
    if op.endswith(":"):
        continue
    cls = ("mfma" if "mfma" in op else "scratch" if op.startswith("scratch") else "global_ld" if op.startswith("global_load") else
           "global_st" if op.startswith("global_store") or op.startswith("global_atomic") else "ds_rd" if op.startswith("ds_read") else
           "ds_wr" if op.startswith("ds_write") else "ds_other" if op.startswith("ds_") else "waitcnt" if op == "s_waitcnt" else
           "barrier" if op == "s_barrier" else "branch" if op.startswith("s_cbranch") or op == "s_branch" else
           "dpp" if "dpp" in t else "trans" if re.match(r"v_(exp|log|rcp|rsq|sqrt)", op) else
           "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "other")
    cur[1][cls] += 1
    if cls == "waitcnt":
        cur[2].append(t.replace("s_waitcnt ", ""))
blocks.append(cur)
for label, cnt, waits in blocks:
    n = sum(cnt.values())
    if n >= 12:
        print("%-12s %5d  %s  %s" % (label, n, dict(cnt.most_common()), " | ".join(waits[:10])))
