#!/bin/bash
# Round 5: hand-off pixel packs (handoff_pack): parity, then step times with and without
TAG=${1:-r05_k}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_scan_persistent.py tests/test_device_post.py tests/test_reentrant.py -m gpu -x -q > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "full_batch or config or golden or two_lanes or bench_plain" >> $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
export VAR_STEPS=10
JDA_LANES=1 JDA_SIDE_STREAM=0 timeout 300 python tools/variants.py "" "JDA_HANDOFF_PACK=0" "" "JDA_HANDOFF_PACK=0" > $O/variants.txt 2>&1; cat $O/variants.txt
timeout 600 python tools/pipe_variants.py "" "JDA_HANDOFF_PACK=0" "" "JDA_HANDOFF_PACK=0" > $O/pipe_variants.txt 2>&1; cat $O/pipe_variants.txt
export VAR_STEPS=4
timeout 600 python tools/config2_variants.py "" "JDA_HANDOFF_PACK=0" "JDA_PACK_MB=2048" > $O/config2_variants.txt 2>&1; cat $O/config2_variants.txt
