"""FDDB harness I/O: the file formats of the reference's `jda fddb` command
(reference src/test.cpp:73-235), on top of the GPU detector.

  <fddb_dir>/FDDB-folds/FDDB-fold-%02d.txt      image ids, one per line   (test.cpp:107,126)
  <fddb_dir>/images/<id>.jpg                    the images                (test.cpp:95,127)
  <fddb_dir>/result/fold-%02d-out.txt           "<id>\\n<n>\\n" then n lines "x y w h score"
                                                ("%d %d %d %d %lf", test.cpp:153,163)
consumed by the external fddb-evaluation tool (reference README.md:132).

The reference decodes with cv::imread + cvtColor(BGR2GRAY).  Pixel-exact equality with
that decode depends on the JPEG library; the gray conversion itself is reproduced
(OpenCV's fixed-point BGR2GRAY) so that a losslessly stored image gives the same bytes.
"""
import os

import numpy as np

# reference model/config.json:41-45 ("fddb" section) -- the defaults of `jda fddb`
FDDB_DEFAULTS = dict(minimum_size=20, step=5, factor=1.2, overlap=0.3, nms=True)


def fold_list_path(fddb_dir, i):
    return os.path.join(fddb_dir, "FDDB-folds", "FDDB-fold-%02d.txt" % i)


def fold_out_path(fddb_dir, i):
    return os.path.join(fddb_dir, "result", "fold-%02d-out.txt" % i)


def read_fold(path):
    """Image ids of one fold: whitespace-separated tokens, like `fscanf(fin, "%s", path)`."""
    with open(path, "r") as f:
        return f.read().split()


def bgr2gray(rgb):
    """OpenCV's 8-bit BGR2GRAY: (R*4899 + G*9617 + B*1868 + 8192) >> 14 (coefficients
    0.299/0.587/0.114 in Q14, round to nearest)."""
    rgb = rgb.astype(np.int32)
    return ((rgb[..., 0] * 4899 + rgb[..., 1] * 9617 + rgb[..., 2] * 1868 + 8192) >> 14).astype(np.uint8)


def load_gray(path):
    """-> uint8 [h,w] or None if the file cannot be read (the reference skips such images, test.cpp:128-131)."""
    try:
        from PIL import Image
        with Image.open(path) as im:
            if im.mode == "L":
                return np.ascontiguousarray(np.asarray(im, np.uint8))
            return bgr2gray(np.asarray(im.convert("RGB"), np.uint8))
    except Exception:
        return None


def format_entry(image_id, rects, scores):
    """One image's block of a fold-XX-out.txt (test.cpp:153,163; %lf prints 6 decimals)."""
    lines = ["%s\n%d\n" % (image_id, len(scores))]
    for r, s in zip(rects, scores):
        lines.append("%d %d %d %d %f\n" % (int(r[0]), int(r[1]), int(r[2]), int(r[3]), float(s)))
    return "".join(lines)


def detect_image(casc, gray, dialect="cpp", params=None, c_call=None):
    """-> (rects[n,4], scores[n], shapes[n,2L], stats). dialect "cpp" = reference fddb() (src/jda Detect,
    method 1); dialect "c" = the C API's canonical call (c/main.cpp:25), rects are (x,y,size,size)."""
    if dialect == "cpp":
        p = dict(FDDB_DEFAULTS)
        p.update(params or {})
        (res,), st = casc.detect_batch_cpp(gray[None], p["minimum_size"], p["step"], p["factor"], p["overlap"],
                                           p["nms"], stats=True)
        return res["rects"], res["scores"], res["shapes"], st
    call = dict(scale=1.25, min_size=40, max_size=-1, th=-0.5)
    call.update(c_call or {})
    (res,), st = casc.detect_batch(gray[None], call["scale"], call["min_size"], call["max_size"], call["th"], stats=True)
    bb = res["bboxes"]
    rects = np.concatenate([bb, bb[:, 2:3]], 1) if len(bb) else np.zeros((0, 4), np.int32)
    return rects, res["scores"].astype(np.float64), res["shapes"].astype(np.float64), st


class FoldStats:
    """DetectionStatisic accumulation of test.cpp:146-149,204-215."""

    def __init__(self):
        self.patch_n = self.face_patch_n = self.nonface_patch_n = self.cart_gothrough_n = 0

    def add(self, st):
        self.patch_n += st["patch_n"]; self.face_patch_n += st["face_patch_n"]
        self.nonface_patch_n += st["nonface_patch_n"]; self.cart_gothrough_n += st["cart_gothrough_n"]

    @property
    def average_cart_n(self):
        return self.cart_gothrough_n / self.nonface_patch_n if self.nonface_patch_n else 0.0

    def summary(self):
        return ("Patch_n = %d, Non-Face Patch_n = %d, Face Patch_n = %d, Average Cart_N to Reject = %.4f"
                % (self.patch_n, self.nonface_patch_n, self.face_patch_n, self.average_cart_n))


def list_job(fddb_dir, folds=range(1, 11)):
    """All (fold, image_id) pairs of the job in file order."""
    job = []
    for i in folds:
        for image_id in read_fold(fold_list_path(fddb_dir, i)):
            job.append((i, image_id))
    return job


def detect_fold_ragged(casc, grays, c_call=None):
    """Dialect C, one fold's (or shard's) images as ONE jdaDetectBatchRagged job instead of one jdaDetect per image
    (test.cpp:100-170 calls Detect image by image; the results are the same, image by image).
    -> ([(rects, scores, shapes)] per image, stats of the job)"""
    call = dict(scale=1.25, min_size=40, max_size=-1, th=-0.5)
    call.update(c_call or {})
    res, st = casc.detect_ragged(grays, call["scale"], call["min_size"], call["max_size"], call["th"], stats=True)
    out = []
    for r in res:
        bb = r["bboxes"]
        rects = np.concatenate([bb, bb[:, 2:3]], 1) if len(bb) else np.zeros((0, 4), np.int32)
        out.append((rects, r["scores"].astype(np.float64), r["shapes"].astype(np.float64)))
    return out, st


def detect_fold_ragged_cpp(casc, grays, params=None):
    """Dialect CPP -- the dialect the reference's fddb() itself runs (joincascador.Detect, method 1, src/test.cpp:142) --
    with one fold's (or shard's) images as ONE jdaDetectBatchCppRagged job.  -> ([(rects, scores, shapes)], stats)"""
    p = dict(FDDB_DEFAULTS)
    p.update(params or {})
    res, st = casc.detect_ragged_cpp(grays, p["minimum_size"], p["step"], p["factor"], p["overlap"], p["nms"], stats=True)
    return [(r["rects"], r["scores"], r["shapes"]) for r in res], st


def run(casc, fddb_dir, folds=range(1, 11), dialect="cpp", params=None, rank=0, world=1, device=None, log=None,
        ragged=None, c_call=None):
    """The whole `jda fddb` run.  With world > 1 the images are split in contiguous blocks over the
    ranks, every rank detects its block, and the (image, rect, score, landmarks) rows
    are gathered on rank 0, which writes the ten fold-XX-out.txt files.  Returns per-fold stats on rank 0.
    ragged (default: on): the images of a fold that fall into this rank's block are decoded first and go through the
    dialect's ragged entry as one job (jdaDetectBatchCppRagged / jdaDetectBatchRagged); off: one call per image, like the
    reference's loop."""
    from . import dist as jdist
    job = list_job(fddb_dir, folds)
    lo, hi = jdist.shard_range(len(job), rank, world)
    L = casc.L
    rows, local_stats, skipped = [], {}, []
    if ragged is None:
        # (the ragged entry is an additive symbol: an older libjda.so loaded through JDA_LIB_PATH may lack it -- then the
        # fold goes image by image, like the reference's loop)
        from . import api as japi
        ragged = hasattr(japi.lib, "jdaDetectBatchCppRagged" if dialect == "cpp" else "jdaDetectBatchRagged")
    pending = []                                              # (idx, gray) of the fold being collected

    def flush(fold):
        if not pending:
            return []
        grays = [g for _, g in pending]
        per_image, st = detect_fold_ragged_cpp(casc, grays, params) if dialect == "cpp" else detect_fold_ragged(casc, grays, c_call)
        local_stats.setdefault(fold, FoldStats()).add(st)
        done = [(i, r) for (i, _), r in zip(pending, per_image)]
        del pending[:]
        return done

    def emit(idx, rects, scores, shapes):
        n = len(scores)
        m = np.empty((n, 6 + 2 * L), np.float64)
        m[:, 0] = idx; m[:, 1:5] = rects; m[:, 5] = scores; m[:, 6:] = shapes
        rows.append(m)
        if n == 0:          # images with no detection still need their "id\n0\n" block: a marker row with rect w = -1
            z = np.zeros((1, 6 + 2 * L)); z[0, 0] = idx; z[0, 3] = -1
            rows.append(z)

    for idx in range(lo, hi):
        fold, image_id = job[idx]
        gray = load_gray(os.path.join(fddb_dir, "images", image_id + ".jpg"))
        if gray is None:
            skipped.append(idx)                             # unreadable image: skipped like the reference
        elif ragged:
            pending.append((idx, gray))
        else:
            rects, scores, shapes, st = detect_image(casc, gray, dialect, params, c_call)
            local_stats.setdefault(fold, FoldStats()).add(st)
            emit(idx, rects, scores, shapes)
        if ragged and (idx + 1 == hi or job[idx + 1][0] != fold):
            for i, (rects, scores, shapes) in flush(fold):
                emit(i, rects, scores, shapes)
    mat = np.concatenate(rows) if rows else np.zeros((0, 6 + 2 * L))
    stat_rows = np.array([[f, s.patch_n, s.face_patch_n, s.nonface_patch_n, s.cart_gothrough_n]
                          for f, s in sorted(local_stats.items())], np.float64).reshape(-1, 5)
    if world > 1:
        mat = _gather64(mat, device)
        stat_rows = _gather64(stat_rows, device)
        if rank != 0:
            return None
    os.makedirs(os.path.join(fddb_dir, "result"), exist_ok=True)
    by_image = {}
    for r in mat:
        by_image.setdefault(int(r[0]), []).append(r)
    outs = {i: open(fold_out_path(fddb_dir, i), "w") for i in folds}
    for idx, (fold, image_id) in enumerate(job):
        if idx not in by_image:
            continue                                        # unreadable image: skipped like the reference
        rs = [r for r in by_image[idx] if r[3] >= 0]
        outs[fold].write(format_entry(image_id, [r[1:5] for r in rs], [r[5] for r in rs]))
    for f in outs.values():
        f.close()
    stats = {}
    for r in stat_rows:
        s = stats.setdefault(int(r[0]), FoldStats())
        s.patch_n += int(r[1]); s.face_patch_n += int(r[2]); s.nonface_patch_n += int(r[3]); s.cart_gothrough_n += int(r[4])
    if log:
        tot = FoldStats()
        for i in sorted(stats):
            log("Summary of Test-%02d" % i); log(stats[i].summary())
            for k in ("patch_n", "face_patch_n", "nonface_patch_n", "cart_gothrough_n"):
                setattr(tot, k, getattr(tot, k) + getattr(stats[i], k))
        log("Summary of ALL"); log(tot.summary())
    return stats


def _gather64(mat, device):
    """float64 rows -> rank 0 (detection rows carry fp64 scores/landmarks of dialect CPP)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device(device) if device is not None else torch.device("cpu")
    width = mat.shape[1]
    cnt = torch.tensor([mat.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    pad = torch.zeros((mx, width), dtype=torch.float64, device=dev)
    if mat.shape[0]:
        pad[: mat.shape[0]] = torch.from_numpy(np.ascontiguousarray(mat)).to(dev)
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0)
    if rank != 0:
        return None
    return np.concatenate([bufs[r][: counts[r]].cpu().numpy() for r in range(world)])


def make_synthetic_fddb(fddb_dir, n_images=2845, seed=0, max_side=450, fmt="JPEG"):
    """FDDB is not in the container (reference data/ is empty): a stand-in with FDDB's layout --
    10 folds, ids like 2002/08/11/big/img_591, sizes <= 450x450 with varied aspect."""
    from PIL import Image
    from . import synth
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(fddb_dir, "FDDB-folds"), exist_ok=True)
    per = [n_images // 10 + (1 if i < n_images % 10 else 0) for i in range(10)]
    k = 0
    for i in range(1, 11):
        ids = []
        for _ in range(per[i - 1]):
            image_id = "%04d/%02d/%02d/big/img_%d" % (2002 + k % 2, 1 + k % 12, 1 + k % 28, k)
            long_side = int(rng.integers(max_side * 2 // 3, max_side + 1))
            short = int(rng.integers(max_side // 2, long_side + 1))
            w, h = (long_side, short) if rng.random() < 0.5 else (short, long_side)
            gray = synth.make_frames(1, w, h, seed=seed + 1, first=k)[0]
            path = os.path.join(fddb_dir, "images", image_id + ".jpg")
            os.makedirs(os.path.dirname(path), exist_ok=True)
            Image.fromarray(np.stack([gray] * 3, -1)).save(path, fmt, **({"quality": 92} if fmt == "JPEG" else {}))
            ids.append(image_id)
            k += 1
        with open(fold_list_path(fddb_dir, i), "w") as f:
            f.write("\n".join(ids) + "\n")
    return k
