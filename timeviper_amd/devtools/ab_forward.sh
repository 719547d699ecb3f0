#!/bin/bash
# Dev tool: A/B of run-time switches INSIDE the 10 240-frame forward (same box, back to back; two bench steps a run: the
# frames/s of two runs of the same setting agree to +- 0.05 %, boxes differ by +- 4 %).
# usage (on the GPU box, repo root):
#   timeviper_amd/devtools/ab_forward.sh <roofline kernel substring> "<ENV=.. ENV=..>" "<ENV=..>" ...
#   e.g.  ab_forward.sh flash_fwd_stream "TV_X=0" "TV_FA_ONES=0 TV_FA_TRIM=0" "TV_X=0"
#         ab_forward.sh ssd_scan "TV_X=0" "TIMEVIPER_HIP_LIB=scratch/libs/variant.so"
# prints per run: the settings, frames/s, and frac / avg launch us of the first roofline whose kernel label holds the substring
pat=$1; shift
for setting in "$@"; do
  env $setting python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = [r for r in [d['roofline']] + d.get('rooflines', []) if '$pat' in r['kernel']]
r = r[0] if r else {}
print('$setting:', d['value'], 'frames/s |', '$pat', 'frac', r.get('frac'), 'avg launch us', r.get('avg_launch_us'))"
done
