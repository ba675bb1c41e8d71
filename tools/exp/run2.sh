cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/exp_ident
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/exp_ident -o t -- python3 tools/exp/forward_identical.py > gpurun_out/exp_ident.log 2>&1
tail -3 gpurun_out/exp_ident.log
