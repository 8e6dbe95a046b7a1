#!/bin/bash
# A/B of experimental builds of the halo kernel on the conv micro-benchmark: tools/ab_conv.sh variant...
# (libraries selected through MAD_LIB_PATH; the product file is never touched)
cd $GRAFT_REPO_ROOT
for v in base "$@"; do
  if [ "$v" = base ]; then lib=$GRAFT_REPO_ROOT/mad_amd/libmad_hip.so; else lib=$GRAFT_REPO_ROOT/mad_amd/libmad_hip_$v.so; fi
  echo "== $v"; MAD_LIB_PATH=$lib timeout -k 10 200 python -u tools/bench_conv.py 32 2>&1 | grep "H=" | head -${LINES_N:-4}
done
