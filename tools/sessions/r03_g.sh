#!/bin/bash
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
for v in "JDA_FILTER0=1" "JDA_FILTER0=0"; do env $v VAR_STEPS=20 python tools/variants.py "JDA_LANES=1 JDA_SIDE_STREAM=0" "" 2>&1 | grep -v amdgpu; done
