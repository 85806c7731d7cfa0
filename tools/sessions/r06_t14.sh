#!/bin/bash
# r06_t14: fp32 k_finish held to five waves per SIMD (96 VGPRs, 24-48 bytes of scratch): 20 windows per CU instead of 16 --
# the headline's ~4.9 k stage-0 survivors fit ONE round of resident windows instead of one and a fifth
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t14; mkdir -p $O; cd $R
for i in 1 2 3; do for lib in jda_amd/libjda.so jda_amd/libjda_fin5.so; do
  echo -n "$lib pipe ahead1: "; JDA_LIB_PATH=$lib PIPE_STEPS=120 PIPE_AHEAD=1 python tools/pipe.py 2>>$O/log.txt | tail -1
  echo -n "$lib pipe ahead2: "; JDA_LIB_PATH=$lib PIPE_STEPS=120 PIPE_AHEAD=2 python tools/pipe.py 2>>$O/log.txt | tail -1
done; done
for lib in jda_amd/libjda.so jda_amd/libjda_fin5.so; do
  echo "$lib:"; JDA_LIB_PATH=$lib VAR_STEPS=30 python tools/variants.py "" "JDA_LANES=1 JDA_SIDE_STREAM=0" 2>>$O/log.txt | cut -c1-110
  echo -n "  C job: "; JDA_LIB_PATH=$lib python tools/fddb_job.py 5 2>>$O/log.txt | tail -1 | cut -c60-130
  echo -n "  shard: "; JDA_LIB_PATH=$lib python tools/shard_job.py 15 2>>$O/log.txt | tail -1 | cut -c1-110
  echo -n "  configs[2]: "; JDA_LIB_PATH=$lib python tools/config2.py 2>>$O/log.txt | tail -1 | cut -c150-260
  echo -n "  latency: "; JDA_LIB_PATH=$lib python tools/latency.py 2>>$O/log.txt | tail -2 | tr '\n' ' ' | cut -c1-200; echo
done
cd /tmp; export TMPDIR=/tmp
for lib in libjda.so libjda_fin5.so; do
  JDA_LIB_PATH=$R/jda_amd/$lib JDA_LANES=1 JDA_SIDE_STREAM=0 VAR_STEPS=10 timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$lib -- python $R/tools/variants.py "" > /dev/null 2>&1
  echo "== $lib, launches back to back"; python $R/tools/rocpd_summary.py $(find $O/kt_$lib -name "*.db" | head -1) k_ | head -9 | cut -c1-150
  rm -rf $O/kt_$lib
done
