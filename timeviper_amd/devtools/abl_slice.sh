for d in ${ABL_LIST:-0 30 29 27 23 15 31}; do
  echo -n "dbg=$d  "; TV_MARCH_DBG=$d timeout 120 python timeviper_amd/devtools/bench_ops.py --ops scan --impl 3 2>&1 | grep ssd_scan
done
