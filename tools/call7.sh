cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for e in NO_DECODE NO_STORE NO_DMA ONLY_STORE ONLY_DECODE; do ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_$e.so bash tools/ab.sh $e cinematic one_clip; done
