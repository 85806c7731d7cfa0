#!/bin/bash
# build the library; non-zero exit (and the compiler's errors) when it fails -- use as `tools/gb.sh && gpurun ...`
cd "$(dirname "$0")/.."
out=$(python jda_amd/build.py "$@" 2>&1)
if echo "$out" | grep -q "error"; then echo "$out" | grep -E "error" -A4 | head -30; exit 1; fi
ls -la --time-style=full-iso jda_amd/libjda.so | cut -c30-80
