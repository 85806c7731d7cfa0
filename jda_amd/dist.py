"""Multi-GPU sharding of a batch of frames and the gather of detections.

Frames (and windows) are independent, the model is read-only and small, so the
path shards by frame with NO data-path collective.  The only
exchange is at the end: every rank sends its (bbox, score, landmarks) tuples
to rank 0 -- one all_gather of counts and one gather of padded rows.  On GPUs
the process group's backend is "nccl", which is RCCL over xGMI on ROCm; the
same code runs on "gloo" for the CPU tests.

The reference has no distributed layer at all (its only parallelism is an
OpenMP loop over FDDB folds, reference src/test.cpp:100); nothing here is a
translation of reference code.
"""
import numpy as np


def shard_range(n_items, rank, world):
    """Contiguous block of items owned by `rank` (floor(i*G/N) partition)."""
    lo = (n_items * rank) // world
    hi = (n_items * (rank + 1)) // world
    return lo, hi


def shard_range_weighted(weights, rank, world):
    """Contiguous block of items owned by `rank` when item i costs weights[i] (candidate windows of image i): the
    boundaries sit where the running sum crosses rank / world of the total, so every rank gets about the same work and
    the blocks still partition [0, n) in order.  Every rank computes the same boundaries from the same weights.
    (Images of a ragged job differ in size: equal COUNTS gave the eight shards of the FDDB-shaped job 4.09 M to 4.40 M
    windows; the job takes as long as its slowest shard.)"""
    w = np.asarray(weights, np.float64)
    n = len(w)
    if n == 0 or world <= 1:
        return (0, n) if rank == 0 else (n, n)
    cs = np.cumsum(w)
    total = float(cs[-1])
    if total <= 0:
        return shard_range(n, rank, world)

    def bound(r):
        if r <= 0:
            return 0
        if r >= world:
            return n
        # first item whose END lies beyond the target: the item that straddles a boundary goes to the side holding more of it
        t = total * r / world
        i = int(np.searchsorted(cs, t, side="left"))
        if i < n and (cs[i] - t) < (t - (cs[i] - w[i])):
            i += 1
        return min(i, n)
    lo, hi = bound(rank), bound(rank + 1)
    return lo, max(lo, hi)


def pack_detections(results, landmark_n, frame_offset=0):
    """Per-frame result dicts -> one float32 matrix, a row per detection:
    [frame, x, y, size, score, shape(2L)].  Integers up to 2^24 are exact in
    float32; frames/coordinates beyond that would need the int path."""
    dim = 2 * landmark_n
    rows = []
    for i, r in enumerate(results):
        n = len(r["scores"])
        if n == 0:
            continue
        m = np.empty((n, 5 + dim), np.float32)
        m[:, 0] = frame_offset + i
        m[:, 1:4] = r["bboxes"]
        m[:, 4] = r["scores"]
        m[:, 5:] = r["shapes"]
        rows.append(m)
    return np.concatenate(rows) if rows else np.zeros((0, 5 + dim), np.float32)


def unpack_detections(mat, landmark_n):
    """Inverse of pack_detections: {frame: result dict}."""
    out = {}
    if len(mat) == 0:
        return out
    frames = mat[:, 0].astype(np.int64)
    for f in np.unique(frames):
        sel = mat[frames == f]
        out[int(f)] = dict(bboxes=sel[:, 1:4].astype(np.int32), scores=sel[:, 4].copy(), shapes=sel[:, 5:].copy())
    return out


def gather_detections(mat, device=None, group=None):
    """Gather every rank's detection rows on rank 0 (None on the other ranks).

    One all_gather of row counts, then one gather of rows padded to the maximum
    count.  Payloads are KBs, so this is latency-bound; no reduction is needed
    anywhere on this path.
    """
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return mat
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    width = mat.shape[1]
    cnt = torch.tensor([mat.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    pad = torch.zeros((mx, width), dtype=torch.float32, device=dev)
    if mat.shape[0]:
        pad[: mat.shape[0]] = torch.from_numpy(np.ascontiguousarray(mat)).to(dev)
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0, group=group)
    if rank != 0:
        return None
    parts = [bufs[r][: counts[r]].cpu().numpy() for r in range(world)]
    return np.concatenate(parts) if parts else np.zeros((0, width), np.float32)


def gather_detections_fixed(mat, max_rows, device=None, group=None):
    """Same result as gather_detections with ONE collective and no host sync in between: every
    rank contributes a fixed (1 + max_rows)-row block whose first row carries its row count.
    Falls back to the two-step gather when any rank holds more than max_rows rows (decided
    collectively from the gathered counts, so every rank takes the same branch)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return mat
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    width = mat.shape[1]
    n = mat.shape[0]
    block = torch.zeros((1 + max_rows, width), dtype=torch.float32, device=dev)
    block[0, 0] = float(n)
    if 0 < n <= max_rows:
        block[1:1 + n] = torch.from_numpy(np.ascontiguousarray(mat)).to(dev)
    allb = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(allb, block, group=group)
    counts = [int(c) for c in torch.stack([b[0, 0] for b in allb]).tolist()]     # one device->host sync
    if max(counts) > max_rows:
        return gather_detections(mat, device=device, group=group)
    if rank != 0:
        return None
    parts = [allb[r][1:1 + counts[r]].cpu().numpy() for r in range(world)]
    return np.concatenate(parts) if parts else np.zeros((0, width), np.float32)


class PipelinedGather:
    """gather_detections_fixed split into start() and finish(), with two sets of preallocated buffers:
    the collective of step i runs on the communicator's stream while step i+1 detects, and is
    collected (counts read, valid rows copied to the host on rank 0) one step later.  Every gather
    is still finished inside the region that started it -- call drain() before the closing barrier.
    On a GPU group the rows returned on rank 0 are a view of a pinned buffer that is reused two
    collections later; copy them if they must live longer."""

    def __init__(self, max_rows, width, device=None, group=None, force=False):
        """force: run the collective even in a group of one (tests of the device path on a 1-GPU box)."""
        import torch
        import torch.distributed as dist
        self.group, self.max_rows, self.width = group, max_rows, width
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.pending = None
        if not self.active:
            return
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.dev = torch.device(device) if device is not None else torch.device("cpu")
        self.block = [torch.zeros((1 + max_rows, width), dtype=torch.float32, device=self.dev) for _ in range(2)]
        self.allb = [[torch.empty_like(self.block[0]) for _ in range(self.world)] for _ in range(2)]
        self.slot = 0
        # rank 0 copies the valid rows out through pinned memory: one asynchronous copy per rank, one sync
        self.host = None
        if self.rank == 0 and self.dev.type == "cuda":
            self.host = [torch.empty((self.world * max_rows, width), dtype=torch.float32).pin_memory() for _ in range(2)]

    def start(self, mat):
        """Launches the collective for this rank's rows; returns what the PREVIOUS start() gathered
        (rows of every rank on rank 0, None elsewhere; None on the first call)."""
        import torch
        import torch.distributed as dist
        if not self.active:
            prev, self.pending = self.pending, ("local", mat)
            return prev[1] if prev else None
        s = self.slot
        n = mat.shape[0]
        blk = self.block[s]
        blk[0, 0] = float(n)
        if 0 < n <= self.max_rows:
            blk[1:1 + n].copy_(torch.from_numpy(np.ascontiguousarray(mat)), non_blocking=False)
        work = dist.all_gather(self.allb[s], blk, group=self.group, async_op=True)
        prev, self.pending = self.pending, (work, s, mat)
        self.slot ^= 1
        return self._finish(prev) if prev else None

    def drain(self):
        """Finishes the gather still in flight (rows on rank 0, None elsewhere / when nothing is pending)."""
        prev, self.pending = self.pending, None
        if prev is None:
            return None
        if not self.active:
            return prev[1]
        return self._finish(prev)

    def _finish(self, token):
        import torch
        work, s, mat = token
        work.wait()
        counts = [int(c) for c in torch.stack([b[0, 0] for b in self.allb[s]]).tolist()]     # one device->host sync
        if max(counts) > self.max_rows:       # same decision on every rank: the counts were gathered
            return gather_detections(mat, device=self.dev, group=self.group)
        if self.rank != 0:
            return None
        if self.host is not None:
            # valid rows of every rank land back to back in pinned memory: the result is a view of it
            # (no concatenation), valid until the gather after next is collected
            off = 0
            for r in range(self.world):
                if counts[r]:
                    self.host[s][off:off + counts[r]].copy_(self.allb[s][r][1:1 + counts[r]], non_blocking=True)
                off += counts[r]
            torch.cuda.current_stream(self.dev).synchronize()
            return self.host[s][:off].numpy()
        else:
            parts = [self.allb[s][r][1:1 + counts[r]].cpu().numpy() for r in range(self.world)]
        return np.concatenate(parts) if parts else np.zeros((0, self.width), np.float32)


# ---------------------------------------------------------------------------------------------
# the C entry (include/jda_dist.h, jda_amd/libjda_dist.so): the same gather for C callers, over RCCL
# ---------------------------------------------------------------------------------------------

_dist_lib = None


def dist_lib():
    """ctypes handle of libjda_dist.so (raises when it is not built: `python -m jda_amd.build`)."""
    global _dist_lib
    if _dist_lib is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libjda_dist.so")
        if not os.path.exists(path):
            raise ImportError("jda_amd: %s is missing -- build it with `python -m jda_amd.build`" % path)
        L = C.CDLL(path)
        fpp = C.POINTER(C.POINTER(C.c_float))
        L.jdaDistUniqueId.argtypes = [C.c_char_p]
        L.jdaDistCreate.restype = C.c_void_p
        L.jdaDistCreate.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_int, C.c_int]
        L.jdaDistDestroy.argtypes = [C.c_void_p]
        L.jdaDistLastError.restype = C.c_char_p
        L.jdaDistGatherRows.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int, fpp, C.POINTER(C.c_int)]
        L.jdaDistGatherStart.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_int]
        L.jdaDistGatherCollect.argtypes = [C.c_void_p, fpp, C.POINTER(C.c_int)]
        L.jdaDistPending.argtypes = [C.c_void_p]
        L.jdaDistFree.argtypes = [C.POINTER(C.c_float)]
        _dist_lib = L
    return _dist_lib


def unique_id():
    """128 rendezvous bytes made on rank 0 (ncclGetUniqueId); hand them to the other ranks."""
    import ctypes as C
    buf = C.create_string_buffer(128)
    if dist_lib().jdaDistUniqueId(buf) != 0:
        raise RuntimeError("jdaDistUniqueId: %s" % dist_lib().jdaDistLastError().decode())
    return buf.raw


class CGather:
    """PipelinedGather's interface on top of the C entry points: start(mat) launches this rank's gather
    (jdaDistGatherStart) and returns what the previous start gathered; drain() finishes the one in flight."""

    def __init__(self, rank, world, id_bytes, device, width, max_rows):
        self.L = dist_lib()
        self.width = width
        self.h = self.L.jdaDistCreate(rank, world, id_bytes, device, width, max_rows)
        if not self.h:
            raise RuntimeError("jdaDistCreate: %s" % self.L.jdaDistLastError().decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.jdaDistDestroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _take(self, rc, out, n):
        import ctypes as C
        if rc != 0:
            raise RuntimeError("libjda_dist: %s" % self.L.jdaDistLastError().decode())
        if not out:
            return None
        mat = np.ctypeslib.as_array(out, (max(n.value, 0), self.width)).copy() if n.value else np.zeros((0, self.width), np.float32)
        self.L.jdaDistFree(out)
        return mat

    def gather(self, mat):
        """Blocking exact gather (jdaDistGatherRows): all rows on rank 0, None elsewhere."""
        import ctypes as C
        mat = np.ascontiguousarray(mat, np.float32)
        out, n = C.POINTER(C.c_float)(), C.c_int()
        rc = self.L.jdaDistGatherRows(self.h, mat.ctypes.data_as(C.POINTER(C.c_float)), mat.shape[0], C.byref(out), C.byref(n))
        return self._take(rc, out, n)

    def start(self, mat):
        import ctypes as C
        prev = self._collect() if self.L.jdaDistPending(self.h) >= 1 else None
        mat = np.ascontiguousarray(mat, np.float32)
        if self.L.jdaDistGatherStart(self.h, mat.ctypes.data_as(C.POINTER(C.c_float)), mat.shape[0]) != 0:
            raise RuntimeError("libjda_dist: %s" % self.L.jdaDistLastError().decode())
        return prev

    def start_raw(self, mat):
        """jdaDistGatherStart alone (up to two may be in flight); collect() finishes the oldest."""
        import ctypes as C
        mat = np.ascontiguousarray(mat, np.float32)
        if self.L.jdaDistGatherStart(self.h, mat.ctypes.data_as(C.POINTER(C.c_float)), mat.shape[0]) != 0:
            raise RuntimeError("libjda_dist: %s" % self.L.jdaDistLastError().decode())

    def collect(self):
        return self._collect()

    def _collect(self):
        import ctypes as C
        out, n = C.POINTER(C.c_float)(), C.c_int()
        rc = self.L.jdaDistGatherCollect(self.h, C.byref(out), C.byref(n))
        return self._take(rc, out, n)

    def drain(self):
        last = None
        while self.L.jdaDistPending(self.h) > 0:
            last = self._collect()
        return last


# ---------------------------------------------------------------------------------------------
# first contact: a known ragged pattern through a gather object, checked on rank 0
# ---------------------------------------------------------------------------------------------

def selftest_rows(pattern, rank, world, width, block_rows):
    """Rows rank `rank` contributes to self-test gather `pattern` (0: every block fits, rank 1 is empty;
    1: the last rank overflows its block, so every rank takes the exact path).  Values are exact in fp32."""
    if pattern == 0:
        n = 0 if rank == 1 else 3 * rank + 2
    else:
        n = block_rows + 5 if rank == world - 1 else (0 if rank == 1 else rank + 1)
    i = np.arange(n, dtype=np.float32)[:, None]
    j = np.arange(width, dtype=np.float32)[None, :]
    return (np.float32(100000 * pattern + 1000 * (rank % 64)) + i + j * np.float32(0.5)).astype(np.float32).reshape(n, width)


def gather_selftest(gather, rank, world, width, block_rows):
    """Drives `gather` (CGather or PipelinedGather: start() returns what the previous start gathered, drain() the
    last) with two gathers in flight -- one whose blocks all fit with an empty rank in it, one that overflows a
    block and sends every rank down the exact count-then-rows path -- and compares what arrives on rank 0 with the
    concatenation in rank order.  Every rank must call it (the gathers are collectives).  Returns "ok" or raises."""
    order = (1, 0)          # the overflowing gather first: its exact fallback then runs with the next gather queued behind it
    mine = [selftest_rows(p, rank, world, width, block_rows) for p in order]
    if hasattr(gather, "start_raw"):           # CGather: really two in flight (jdaDistGatherStart x 2, then Collect x 2)
        gather.start_raw(mine[0]); gather.start_raw(mine[1])
        got = [gather.collect(), gather.collect()]
    else:
        first = gather.start(mine[0])
        if first is not None and len(first):
            raise RuntimeError("a fresh gather returned rows before anything was started")
        got = [gather.start(mine[1]), gather.drain()]
    if hasattr(gather, "gather"):              # the blocking exact entry too (jdaDistGatherRows)
        got.append(gather.gather(mine[1]))
    if rank != 0:
        for g in got:
            if g is not None and len(g):
                raise RuntimeError("rows arrived on rank %d" % rank)
        return "ok"
    for k, g in enumerate(got):
        p = order[k] if k < 2 else order[1]
        want = np.concatenate([selftest_rows(p, r, world, width, block_rows) for r in range(world)])
        if g is None or g.shape != want.shape or not np.array_equal(np.asarray(g), want):
            raise RuntimeError("gather %d: rows on rank 0 differ from the ranks' rows in rank order (got %s, want %s)"
                               % (k, None if g is None else g.shape, want.shape))
    return "ok"
