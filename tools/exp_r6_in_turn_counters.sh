#!/bin/bash
# round 6: counters of the rig's kernel with its arguments read from the kernarg segment (7 and 8 waves per SIMD). Output: gpurun_out/r06m_*
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/profile_round6.sh r06m cinematic 2>&1 | tail -3
ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_ab_turn8.so bash tools/profile_round6.sh r06m8 cinematic 2>&1 | tail -3
