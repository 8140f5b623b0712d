cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
python tools/debug_layout.py crowd_rig_1200 2>&1 | grep mismatches
for w in one_clip cinematic; do echo "== $w"; bash tools/pmc_one.sh $w "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD" "--no-extras"; done
