"""JOB_DUMMY=k: k streams created (and used) before the cascador's own -- shifts the deal of streams to hardware queues
(experiments on how much a job's time depends on it: tools/sessions/r06_t2.sh)."""
import os, torch
_keep = []
def make():
    for _ in range(int(os.environ.get("JOB_DUMMY", "0"))):
        s = torch.cuda.Stream(); _keep.append(s)
        with torch.cuda.stream(s): torch.zeros(16, device="cuda").add_(1)
    if _keep: torch.cuda.synchronize()
