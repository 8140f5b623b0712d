cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
# LDS per block 20 KB -> 8 blocks/CU (32 waves). extra 20000 -> 4 blocks (16 waves); 60000 -> 2 blocks (8 waves); 7000 -> 5 blocks(20); 12000 -> 5 blocks
for extra in 0 7000 20000 33000 60000; do
  echo "extra LDS $extra"
  ACLHIP_EXTRA_LDS_BYTES=$extra ACLHIP_LIBRARY=$PWD/acl_amd/lib/libaclhip_ONLY_STORE.so bash tools/ab.sh only_store cinematic one_clip
  ACLHIP_EXTRA_LDS_BYTES=$extra bash tools/ab.sh full cinematic one_clip
done
