#!/bin/bash
# round 5, set I (one gpurun call): the whole GPU suite, then the evidence round for profiles/r05_*
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r5i_pytest.txt 2>&1; tail -4 gpurun_out/r5i_pytest.txt | head -3
bash tools/profile_round5.sh r05 2>&1 | tail -45
