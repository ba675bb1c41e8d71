#!/bin/bash
# centralised backward pass: wavefronts per workgroup 1 / 4 / 8 / 16 -- C2 timing and parity
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for w in ${WPBS:-1 2 4 8 16}; do
  export MAS_HIP_CENTRAL_WPB=$w
  timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "centralized or config2" > gpurun_out/c2wpb_$w.txt 2>&1
  echo "wpb $w: $(tail -1 gpurun_out/c2wpb_$w.txt)"
  timeout -k 10 200 python tools/profile_config.py c2 2>&1 | grep -v amdgpu.ids | head -4
done
