#!/bin/bash
# A/B of experimental library builds on one whole denoiser evaluation (eval wall time from tools/layer_times.py)
cd $GRAFT_REPO_ROOT
cp mad_amd/libmad_hip.so /tmp/base.so
for v in "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so mad_amd/libmad_hip.so; else cp mad_amd/libmad_hip_$v.so mad_amd/libmad_hip.so; fi
  echo "== $v: $(timeout -k 10 200 python -u tools/layer_times.py 2>&1 | grep 'eval wall')"
done
cp /tmp/base.so mad_amd/libmad_hip.so
