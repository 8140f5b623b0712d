#!/bin/bash
# Builds acl_amd/lib/libaclhip_exp.so: the library WITH round 3's slower kernel variants (acl_amd/csrc/kernels_experiments.inl) behind their
# environment knobs. Point ACLHIP_LIBRARY at it (tools/exp_r3*.sh, tools/pmc_variants.sh do).
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -Wall -Wextra -ldl -mllvm -amdgpu-kernarg-preload-count=16 -DACLHIP_EXPERIMENTS "$@" acl_amd/csrc/aclhip.hip -o acl_amd/lib/libaclhip_exp.so
