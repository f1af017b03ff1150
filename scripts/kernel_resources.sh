#!/bin/bash
# VGPR / SGPR / LDS / spill figures of every kernel of one translation unit:  scripts/kernel_resources.sh dcn_colpath.hip [filter]
SRC=detectron2_amd/csrc/$1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fno-gpu-rdc -w -Xclang -target-feature -Xclang -packed-fp32-ops $EXTRA \
  -S --cuda-device-only -o /tmp/kres.s "$SRC" 2>/dev/null
python3 - "$2" <<'PY'
import re, sys
flt = sys.argv[1] if len(sys.argv) > 1 else ""
txt = open("/tmp/kres.s").read()
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", txt, re.S):
    pass
blocks = re.split(r"\n  - \.agpr_count:", txt)
for b in blocks[1:]:
    name = re.search(r"\.name:\s+(\S+)", b)
    if not name or flt not in name.group(1):
        continue
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, b) or [None, "?"])[1]
    import subprocess
    dem = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
    print("%-110s vgpr %s agpr %s sgpr %s lds %s scratch %s spill v%s s%s" % (dem[:110], g("vgpr_count"), b.split("\n")[0].strip(), g("sgpr_count"),
          g("group_segment_fixed_size"), g("private_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
PY
