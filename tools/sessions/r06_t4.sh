#!/bin/bash
# r06_t4: streams placed on the hardware queues by probing (StreamPool, JDA_HWQ_PLACE = 1 / 2) against the runtime's deal (0),
# each under three stream histories of the host program (JOB_DUMMY = streams it created first)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06_t4; mkdir -p $O; cd $R
run() { # label, env...
  local label="$1"; shift
  echo "== $label" >> $O/log.txt
  echo -n "$label | cpp job: "; env "$@" python tools/cpp_job.py 5 2>>$O/log.txt | grep "CPP ragged job" | cut -c38-80
  echo -n "$label | C job: "; env "$@" python tools/fddb_job.py 5 2>>$O/log.txt | tail -1 | cut -c60-130
  echo -n "$label | shard: "; env "$@" python tools/shard_job.py 9 2>>$O/log.txt | tail -1 | cut -c1-100
  echo -n "$label | pipe: "; env "$@" PIPE_STEPS=80 PIPE_AHEAD=2 python tools/pipe.py 2>>$O/log.txt | tail -1
  echo -n "$label | host frames: "; env "$@" python tools/host_variants.py "" 2>>$O/log.txt | tail -2 | tr '\n' ' ' | cut -c1-200; echo
}
for d in 0 1 2; do for p in 0 1 2; do run "place$p dummy$d" JDA_HWQ_PLACE=$p JOB_DUMMY=$d; done; done
JDA_HWQ_PLACE=1 python - <<'P' 2>&1 | tail -3
import sys; sys.path.insert(0, '.')
import torch, time
from jda_amd import synth, api
import bench
mp = bench.model_path((5, 540, 27, 4), "cascade", 1, synth.make_frames(8, 640, 480, seed=0, first=10_000_000))
t0 = time.perf_counter(); c = api.Cascador(mp)
d = torch.from_numpy(synth.make_frames(8, 640, 480, seed=0)).cuda()
c.detect_batch_device(d); t1 = time.perf_counter()
print("first call %.1f ms; queues %d streams %d probes %d max mains per queue %d" % ((t1 - t0) * 1e3, *[c.get_option(k) for k in ("hwq_queues", "hwq_streams", "hwq_probes", "hwq_max_mains")]))
q = [c.submit_batch_device(d) for _ in range(3)]
for t in q: c.wait_batch(t)
print("after three tickets: queues %d streams %d probes %d max mains per queue %d" % tuple(c.get_option(k) for k in ("hwq_queues", "hwq_streams", "hwq_probes", "hwq_max_mains")))
P
