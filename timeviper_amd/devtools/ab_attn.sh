#!/bin/bash
# Dev tool: same-box A/B of the ViT attention probe between a saved build and the current tree.
# usage (on the GPU box): bash timeviper_amd/devtools/ab_attn.sh <base.so>
#   <base.so> is a copy of lib/libtimeviper_hip.so of the tree to compare against, saved beforehand as
#   timeviper_amd/lib/variants/<base.so> (the directory is git-ignored but travels with gpurun); it is loaded through
#   TIMEVIPER_HIP_LIB, which skips the build-id check.  Three alternating rounds of devtools/attn_prologue_probe.py.
for i in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export TIMEVIPER_HIP_LIB=$PWD/timeviper_amd/lib/variants/$1; else unset TIMEVIPER_HIP_LIB; fi
    echo -n "$v: "; timeout 100 python timeviper_amd/devtools/attn_prologue_probe.py 2>&1 | grep -v amdgpu.ids | head -1
  done
done
