# Dev tool: where the head-per-wave scan kernel's HBM reads come from.  FETCH_SIZE / L2 passes on a -DTV_HEAD_ABLATE build with
# one class of memory operations switched off at a time (TV_HEAD_DBG; results of such runs are wrong, only the counters matter).
# usage (on the GPU box): bash timeviper_amd/devtools/pmc_scan_attrib.sh ; then python timeviper_amd/devtools/summarize_scan_attrib.py <tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp -r timeviper_amd/lib /tmp/lib_plain
TV_EXTRA_HIPCC_FLAGS=-DTV_HEAD_ABLATE python3 -m timeviper_amd.build --jobs 8 > gpurun_out/attrib_build.log 2>&1 || { tail -20 gpurun_out/attrib_build.log; exit 1; }
export TV_EXTRA_HIPCC_FLAGS=-DTV_HEAD_ABLATE
for dbg in 0 32 64 8 128 4 236; do
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    tag=$(echo $set | cut -c1-3)
    rm -rf gpurun_out/attrib_${dbg}_$tag
    TV_HEAD_DBG=$dbg timeout 150 rocprofv3 --pmc $set -d gpurun_out/attrib_${dbg}_$tag -o p --output-format csv -- python3 timeviper_amd/devtools/bench_ops.py --ops scan --model-dt --dt-std 1.3 > gpurun_out/attrib_${dbg}_$tag.log 2>&1
    echo "dbg $dbg $tag rc=$? $(grep ssd_scan gpurun_out/attrib_${dbg}_$tag.log | tail -1)"
  done
done
unset TV_EXTRA_HIPCC_FLAGS
rm -rf timeviper_amd/lib && mv /tmp/lib_plain timeviper_amd/lib
