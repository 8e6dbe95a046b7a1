#!/bin/bash
# A/B of experimental library builds on the GPU box: tools/ab.sh "<command>" variant...
# runs the command with the regular library ("base") and with each mad_amd/libmad_hip_<variant>.so in its place
cd $GRAFT_REPO_ROOT
cmd=$1; shift
cp mad_amd/libmad_hip.so /tmp/base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so mad_amd/libmad_hip.so; else cp mad_amd/libmad_hip_$v.so mad_amd/libmad_hip.so; fi
  echo "== $v"; timeout -k 10 200 bash -c "$cmd" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/base.so mad_amd/libmad_hip.so
