#!/bin/bash
# A/B of two builds of libjda.so on the same box: tools/ab.sh <libA> <libB> [variants args...]
cd "$(dirname "$0")/.."
A=$1; B=$2; shift 2
for rep in 1 2; do
  for L in "$A" "$B"; do
    echo "== $L"
    JDA_LIB_PATH=$PWD/$L VAR_STEPS=20 python tools/variants.py "JDA_LANES=1 JDA_SIDE_STREAM=0" "" 2>&1 | grep -v amdgpu.ids
  done
done
