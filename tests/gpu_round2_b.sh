# full GPU suite
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1100 python3 -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/b_tests.log 2>&1
rc=$?
tail -25 gpurun_out/b_tests.log
exit $rc
