cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/ab.sh w312 cinematic one_clip
for q in 192 240 384 456; do ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_w$q.so bash tools/ab.sh w$q cinematic; done
