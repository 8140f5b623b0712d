#!/bin/bash
# (history: this set ran on commit 5d59a74 / its working tree -- the knobs and variant libraries it names are not part of the shipped tree; results: profiles/r05_experiments.md 8)
# round 5, set Q: phase stamps (tools/phase_times.py) of the pose kernel before / after the prologue change of set P.
#   *_phase1: entry | seek done | window decoded | stores issued          (-DACLHIP_EXP_PHASE_TIMES=1)
#   *_phase2: entry | inputs arrived | clip record arrived | seek done    (-DACLHIP_EXP_PHASE_TIMES=2; printed under phase1's names)
# base = the commit before (4649d62) with the same stamps patched in.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r5q
for lib in base_phase1 phase1 base_phase2 phase2; do
	export ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_$lib.so
	for spec in "one_clip qvv48" "one_clip qv32" "256_clips qvv48"; do echo "== $lib"; python tools/phase_times.py $spec; done
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r5q/phase_times.txt
