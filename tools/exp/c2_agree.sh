#!/bin/bash
# centralised backward pass: wavefronts per workgroup x (every wavefront arrives | one arrival per workgroup)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in 0 1; do for w in ${WPBS:-1 2 4}; do
  export MAS_HIP_CENTRAL_WPB=$w MAS_HIP_CENTRAL_COMBINE=$c
  echo "wpb $w combine $c: $(timeout -k 10 200 python tools/profile_config.py c2 2>&1 | grep -v amdgpu.ids | head -2 | tr '\n' ' ')"
done; done
