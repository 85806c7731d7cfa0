# r06: with bounded workspaces a chunk's memory no longer grows with its windows: how large should chunks be?
run() { echo "== $*"; env "$@" python tools/cpp_job.py 5 2>&1 | grep "CPP ragged" | sed 's/CPP ragged job, 2845 images resident: //'; }
run A=0
run JDA_RAGGED_CHUNK_WINDOWS_CPP=16000000
run JDA_RAGGED_CHUNK_WINDOWS_CPP=32000000
run JDA_RAGGED_CHUNK_WINDOWS_CPP=64000000
run JDA_RAGGED_CHUNK_WINDOWS_CPP=32000000 JDA_RAGGED_LANES=2
run JDA_RAGGED_CHUNK_WINDOWS_CPP=120000000 JDA_RAGGED_SPLIT=1
run JDA_RAGGED_CHUNK_WINDOWS_CPP=120000000 JDA_RAGGED_SPLIT=2
for cw in 4000000 8000000 12000000 17000000 34000000; do echo "== C ragged chunk $cw"; JDA_RAGGED_CHUNK_WINDOWS=$cw python tools/ragged_bench.py --variants device 2>&1 | tail -1 | cut -c1-300; done
