cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 150 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_slice5 -o p --output-format csv -- python3 timeviper_amd/devtools/bench_ops.py --ops scan --impl 3 > gpurun_out/pmc_slice5.log 2>&1
echo "rc=$?"; tail -1 gpurun_out/pmc_slice5.log
