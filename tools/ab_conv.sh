#!/bin/bash
# A/B of experimental builds of the halo kernel: swaps the library, runs the conv micro-benchmark
cd $GRAFT_REPO_ROOT
cp mad_amd/libmad_hip.so /tmp/base.so
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/base.so mad_amd/libmad_hip.so; else cp mad_amd/libmad_hip_$v.so mad_amd/libmad_hip.so; fi
  echo "== $v"; timeout -k 10 200 python -u tools/bench_conv.py 32 2>&1 | grep "H=" | head -${LINES_N:-4}
done
cp /tmp/base.so mad_amd/libmad_hip.so
