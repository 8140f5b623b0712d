cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/ab.sh base cinematic one_clip
for v in prio wpb8 wpb2 wpb1; do ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_$v.so bash tools/ab.sh $v cinematic one_clip; done
