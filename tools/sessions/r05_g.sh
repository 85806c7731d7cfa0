#!/bin/bash
TAG=${1:-r05_g}
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
timeout 900 python -m pytest tests -m gpu -x -q -k "method0 or pyramid or multiscale or cv_resize or wide" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
