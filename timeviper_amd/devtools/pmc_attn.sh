# Dev tool: rocprofv3 PMC passes on the attention / patch-embed ops (MFMA utilisation evidence).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
           "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set -d gpurun_out/pmc_attn$i -o p --output-format csv -- python3 timeviper_amd/devtools/bench_ops.py --ops attn,patch > gpurun_out/pmc_attn$i.log 2>&1
  echo "pass $i rc=$?"
done
