#!/bin/bash
# Round 5: k_finish_wide with replay and regression side by side (wide_conc): parity tests + single-frame latency A/B
TAG=${1:-r05_c}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "wide or golden or dialects_agree or c_caller or smoke or extreme or handful or single" > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
for v in 1 0 1 0; do echo "== JDA_WIDE_CONC=$v"; JDA_WIDE_CONC=$v timeout 200 python tools/latency.py 2>/dev/null | head -2; done > $O/latency_ab.txt
cat $O/latency_ab.txt
