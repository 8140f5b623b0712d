#!/bin/bash
# Measurement aid: registers, spills and scratch of every kernel of aclhip.hip (cross-compiled here, no GPU needed). usage: tools/kernel_resources.sh [filter] [extra hipcc flags]
out=${TMPDIR:-/tmp}/aclhip_resources.s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-kernarg-preload-count=16 --offload-device-only -S ${2:-} "$(dirname "$0")/../acl_amd/csrc/aclhip.hip" -o $out 2>/dev/null
python3 - "$out" "${1:-}" <<'PY'
import re, sys
text = open(sys.argv[1]).read()
flt = sys.argv[2]
print("%-72s %5s %5s %7s %7s %7s" % ("kernel", "vgpr", "sgpr", "vspill", "sspill", "scratch"))
for block in text.split("  - .agpr_count:")[1:]:
    get = lambda key: re.search(r"\." + key + r":\s+(\S+)", block)
    name = get("name").group(1)
    name = re.sub(r"^_ZN6aclhip\d+", "", name)
    name = re.sub(r"E[vP].*$", "", name)
    if flt and flt not in name:
        continue
    print("%-72s %5s %5s %7s %7s %7s" % (name, get("vgpr_count").group(1), get("sgpr_count").group(1), get("vgpr_spill_count").group(1), get("sgpr_spill_count").group(1), get("private_segment_fixed_size").group(1)))
PY
