R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_s1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt -- python $R/tools/shard_job.py 10 8 0 > $O/run.txt 2>&1
cd $R
python tools/shard_timeline.py $(find $O/kt -name "*.db" | head -1) > $O/timeline.txt 2>&1
timeout 100 python tools/experiments/r06_shard_host_times.py > $O/host_times.txt 2>&1
rm -rf $O/kt
cat $O/timeline.txt; tail -40 $O/host_times.txt; tail -3 $O/run.txt
