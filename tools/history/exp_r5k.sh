#!/bin/bash
# round 5, set K: two instances per wave for one-window poses (ACLHIP_PAIR_KERNELS=1) -- parity of the whole pose test set under it, then A/B
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
ACLHIP_PAIR_KERNELS=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_layouts.py tests/test_gpu_instance_writers.py tests/test_gpu_instance_lists.py tests/test_gpu_full_size.py tests/test_gpu_database.py tests/test_random_sweep.py tests/test_gpu_launch_shape.py tests/test_gpu_lifetime.py -m gpu -x -q > gpurun_out/r5k_pytest.txt 2>&1; grep -n "passed\|failed" gpurun_out/r5k_pytest.txt
P=ACLHIP_PAIR_KERNELS=1
python tools/variant_sweep.py --repeats 300 --workloads one_clip,one_clip_lods,256_clips,database one:ACLHIP_PAIR_KERNELS=0 pair:$P one2:ACLHIP_PAIR_KERNELS=0 pair2:$P | tee gpurun_out/r5k_pairs.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip --layout qv32 one:ACLHIP_PAIR_KERNELS=0 pair:$P one2:ACLHIP_PAIR_KERNELS=0 pair2:$P | tee -a gpurun_out/r5k_pairs.txt
python tools/variant_sweep.py --repeats 300 --workloads one_clip --layout qvv40 one:ACLHIP_PAIR_KERNELS=0 pair:$P one2:ACLHIP_PAIR_KERNELS=0 pair2:$P | tee -a gpurun_out/r5k_pairs.txt
python tools/variant_sweep.py --repeats 300 --workloads 256_clips --order locality one:ACLHIP_PAIR_KERNELS=0 pair:$P | tee -a gpurun_out/r5k_pairs.txt
