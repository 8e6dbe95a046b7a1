#!/bin/bash
# A/B of experimental library builds on whole denoiser evaluations: tools/ab_model.sh variant...  (see tools/ab_eval.py)
cd $GRAFT_REPO_ROOT
timeout -k 10 ${AB_TIMEOUT:-300} python -u tools/ab_eval.py base "$@" 2>&1 | grep -v amdgpu.ids
