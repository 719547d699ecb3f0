# Dev tool: rocprofv3 PMC passes (one counter set per pass, no tracing) on the scan op.
# usage (on the GPU box): bash timeviper_amd/devtools/pmc_scan.sh [impl] [first pass] [last pass]
# then: python timeviper_amd/devtools/summarize_pmc_scan.py <round tag> <impl>   (writes profiles/<tag>_ssd_scan_{pmc.md,traffic.json})
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
IMPL=${1:-0}
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "WRITE_SIZE"; do
  i=$((i+1))
  if [ $i -lt ${2:-1} ] || [ $i -gt ${3:-9} ]; then continue; fi
  rm -rf gpurun_out/pmc_slice$i
  timeout 150 rocprofv3 --pmc $set -d gpurun_out/pmc_slice$i -o p --output-format csv -- python3 timeviper_amd/devtools/bench_ops.py --ops scan --model-dt --dt-std 1.3 --impl $IMPL > gpurun_out/pmc_slice$i.log 2>&1
  echo "pass $i rc=$?"; tail -1 gpurun_out/pmc_slice$i.log
done
