#!/bin/bash
# cycle stamps of one workgroup of the halo conv: tools/run_stamps.sh  (needs mad_amd/libmad_hip_stamps.so, selected
# through MAD_LIB_PATH)
cd $GRAFT_REPO_ROOT
export MAD_LIB_PATH=$GRAFT_REPO_ROOT/mad_amd/libmad_hip_stamps.so
for cfg in "256 1 32" "256 1 1" "256 1 2" "256 0 32" "256 0 1"; do
  echo "=== H fuse B = $cfg"
  timeout -k 10 120 python -u tools/halo_stamps.py $cfg 2>&1 | grep -v amdgpu.ids
done
