"""Kernels whose global loads are serialised: a load followed at once by s_waitcnt vmcnt(0) with more loads behind it (a guard per
element makes every load a branch with its own wait).  python tools/dbg/serial_loads.py flamo_amd/csrc/response.hip [regex]"""
import re
import subprocess
import sys
import tempfile
import os

src = sys.argv[1]
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(tempfile.mkdtemp(), "k.s")
extra = sys.argv[3:]
subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "--cuda-device-only", "-S",
                "-I" + os.path.join(root, "flamo_amd", "csrc"), *extra, src, "-o", out], check=True, capture_output=True)
lines = open(out).read().splitlines()
i = 0
rows = []
while i < len(lines):
    m = re.match(r"^(_Z\w+):\s", lines[i])
    if m and i + 1 < len(lines):
        j = i
        while j < len(lines) and "s_endpgm" not in lines[j]:
            j += 1
        body = [l for l in lines[i:j] if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        loads = [k for k, l in enumerate(body) if "global_load" in l or "buffer_load" in l]
        serial = sum(1 for k in loads if any("vmcnt(0)" in body[k + d] for d in (1, 2) if k + d < len(body)))
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        if loads and pat.search(name):
            rows.append((serial, len(loads), len(body), name[:120]))
        i = j
    i += 1
for r in sorted(rows, reverse=True)[:40]:
    print(f"{r[0]:4d} of {r[1]:4d} loads waited for at once   {r[2]:6d} instr   {r[3]}")
