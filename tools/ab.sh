#!/bin/bash
# A/B of experimental library builds on the GPU box: tools/ab.sh "<command>" variant...
# runs the command with the regular library ("base") and with each mad_amd/libmad_hip_<variant>.so, selected through
# MAD_LIB_PATH (mad_amd/_lib.py): the product file mad_amd/libmad_hip.so is never touched
cd $GRAFT_REPO_ROOT
cmd=$1; shift
for v in base "$@"; do
  if [ "$v" = base ]; then lib=$GRAFT_REPO_ROOT/mad_amd/libmad_hip.so; else lib=$GRAFT_REPO_ROOT/mad_amd/libmad_hip_$v.so; fi
  echo "== $v"; MAD_LIB_PATH=$lib timeout -k 10 ${AB_TIMEOUT:-200} bash -c "$cmd" 2>&1 | grep -v amdgpu.ids
done
