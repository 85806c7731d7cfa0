#!/bin/bash
# r05_z: k_finish's LDS window tile (JDA_FIN_TILE) on the FDDB-shaped job and on the headline leg: is 84 (which takes in a
# 77-px / 81-px level) a better default where such a level exists?
mkdir -p gpurun_out/r05_z
timeout 600 python tools/fddb_job.py 20 "" "JDA_FIN_TILE=84" "" "JDA_FIN_TILE=84" "JDA_FIN_TILE=80" 2>&1 | grep "ms per job" | tee gpurun_out/r05_z/fddb.txt
timeout 600 python tools/pipe_variants.py "" "JDA_FIN_TILE=84" "" "JDA_FIN_TILE=84" 2>&1 | tail -6 | tee gpurun_out/r05_z/pipe.txt
