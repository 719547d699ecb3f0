# Dev tool: rocprofv3 PMC passes on the ViT attention kernel (TV_FA_W64=0: the generated tile loop, the default; 5: the compiled streaming kernel).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in 0 5; do
  export TV_FA_W64=$v
  i=0
  for set in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM"; do
    i=$((i+1))
    timeout 150 rocprofv3 --pmc $set -d gpurun_out/pmc_vit${v}_$i -o p --output-format csv -- python3 timeviper_amd/devtools/bench_ops.py --ops attn > gpurun_out/pmc_vit${v}_$i.log 2>&1
    echo "variant $v pass $i rc=$?"
  done
done
