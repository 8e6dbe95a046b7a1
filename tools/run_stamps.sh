#!/bin/bash
# cycle stamps of one workgroup of the halo conv: tools/run_stamps.sh  (needs mad_amd/libmad_hip_stamps.so)
cd $GRAFT_REPO_ROOT
cp mad_amd/libmad_hip.so /tmp/base.so
cp mad_amd/libmad_hip_stamps.so mad_amd/libmad_hip.so
for cfg in "256 1 32" "256 1 1" "256 1 2" "256 0 32" "256 0 1"; do
  echo "=== H fuse B = $cfg"
  timeout -k 10 120 python -u tools/halo_stamps.py $cfg 2>&1 | grep -v amdgpu.ids
done
cp /tmp/base.so mad_amd/libmad_hip.so
