cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/ab.sh w318 cinematic
ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_b.so bash tools/ab.sh w320 cinematic
ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_c.so bash tools/ab.sh w312 cinematic
