#!/bin/bash
# usage: ab.sh <base.so> : alternates base/new on the ViT attention probe
for i in 1 2 3; do
  for v in base new; do
    if [ $v = base ]; then export TIMEVIPER_HIP_LIB=$PWD/timeviper_amd/lib/variants/$1; else unset TIMEVIPER_HIP_LIB; fi
    echo -n "$v: "; timeout 100 python timeviper_amd/devtools/attn_prologue_probe.py 2>&1 | grep -v amdgpu.ids | head -1
  done
done
