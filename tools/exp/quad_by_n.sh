#!/bin/bash
# four-lanes-per-rollout line search: launch time against the number of agents (fresh C3-shaped problems of N agents)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for n in ${NS:-16 64 128 256 448 832 1408 1600}; do
  echo "N=$n: $(timeout -k 10 100 python tools/profile_config.py c3 $n 2>&1 | grep -v amdgpu.ids | grep -E "forward|backward " | tr '\n' ' ')"
done
