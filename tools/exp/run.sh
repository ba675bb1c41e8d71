cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for cfg in "4096 4096 0" "64 4096 0" "64 4096 1024"; do
  set -- $cfg
  rm -rf gpurun_out/exp_$1_$3
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/exp_$1_$3 -o t -- python3 tools/exp/forward_by_round.py $1 $2 $3 > gpurun_out/exp_$1_$3.log 2>&1
done
ls gpurun_out/exp_*/
