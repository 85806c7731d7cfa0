# r06: chunks of a ragged job on one stream (in order) vs one stream per lane (side by side)
run() { echo "== $*"; env "$@" python tools/cpp_job.py 5 2>&1 | grep "CPP ragged" | sed 's/CPP ragged job, 2845 images resident: //'; }
run A=0
run JDA_RAGGED_ONE_STREAM=1
run JDA_RAGGED_ONE_STREAM=1 JDA_RAGGED_LANES=2
run JDA_RAGGED_ONE_STREAM=1 JDA_RAGGED_LANES=4
run JDA_RAGGED_ONE_STREAM=1 JDA_RAGGED_CHUNK_WINDOWS_CPP=16000000
run JDA_RAGGED_ONE_STREAM=1 JDA_RAGGED_CHUNK_WINDOWS_CPP=4000000
run A=0
for os in 0 1; do echo "== C ragged one_stream $os"; JDA_RAGGED_ONE_STREAM=$os python tools/ragged_bench.py --variants device 2>&1 | tail -1 | cut -c1-260; done
for os in 0 1; do echo "== C ragged one_stream $os chunk 8M"; JDA_RAGGED_CHUNK_WINDOWS=8000000 JDA_RAGGED_ONE_STREAM=$os python tools/ragged_bench.py --variants device 2>&1 | tail -1 | cut -c1-260; done
