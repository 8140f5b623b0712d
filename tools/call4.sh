cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/ab.sh prefetch_spills one_clip cinematic
ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_b.so bash tools/ab.sh no_prefetch one_clip cinematic 256_clips database
