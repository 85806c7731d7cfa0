#!/bin/bash
cd "$(dirname "$0")/.."
for L in "$@"; do echo "== $L"; JDA_LIB_PATH=$PWD/$L python tools/host_variants.py "" "AHEAD=1" 2>&1 | grep -v amdgpu.ids; done
