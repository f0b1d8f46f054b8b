"""VGPR / SGPR / LDS / scratch of the library's kernels, read from the code objects inside libflamo_hip.so.
  python tools/dbg/kernel_resources.py [regex]"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
tmp = tempfile.mkdtemp()
so = os.path.join(tmp, "lib.so")
shutil.copy(os.path.join(ROOT, "flamo_amd", "libflamo_hip.so"), so)
subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", so], capture_output=True, cwd=tmp)
for f in sorted(glob.glob(so + ".*gfx950")):
    out = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
    for blk in out.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
        if not pat.search(dem):
            continue
        g = lambda k: re.search(r"\." + k + r":\s+(\d+)", blk).group(1)  # noqa: E731
        print(f"{dem[:110]:110s} vgpr {g('vgpr_count'):>3s} agpr {blk.split()[0]:>3s} sgpr {g('sgpr_count'):>3s} lds {g('group_segment_fixed_size'):>6s} "
              f"scratch {g('private_segment_fixed_size'):>4s}")
shutil.rmtree(tmp)
