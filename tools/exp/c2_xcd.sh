#!/bin/bash
# centralised backward pass (pipelined kernel): working workgroups spread over the XCDs / all on one XCD
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for x in 0 1; do
  export MAS_HIP_CENTRAL_ONE_XCD=$x
  echo "one_xcd $x: $(timeout -k 10 200 python tools/profile_config.py c2 2>&1 | grep -v amdgpu.ids | head -2 | tr '\n' ' ')"
done
