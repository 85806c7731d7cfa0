"""FDDB harness I/O (reference src/test.cpp:73-235): file formats, gray conversion,
sharded run with gather (CPU, fake detector), and end-to-end on the GPU vs the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_format_entry_matches_reference_printf():
    from jda_amd import fddb
    # fprintf(fout, "%s\n%d\n", path, n); fprintf(fout, "%d %d %d %d %lf\n", x, y, w, h, score)
    txt = fddb.format_entry("2002/08/11/big/img_591", [[10, 20, 48, 48], [0, 5, 57, 57]], [1.25, -0.3333333333])
    assert txt == "2002/08/11/big/img_591\n2\n10 20 48 48 1.250000\n0 5 57 57 -0.333333\n"
    assert fddb.format_entry("a/b", [], []) == "a/b\n0\n"


def test_bgr2gray_is_opencv_fixed_point():
    from jda_amd import fddb
    rgb = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [12, 200, 77]]], np.uint8)
    want = [(r * 4899 + g * 9617 + b * 1868 + 8192) >> 14 for r, g, b in rgb[0].tolist()]
    assert fddb.bgr2gray(rgb)[0].tolist() == want == [255, 0, 76, 150, 29, 130]


def test_synthetic_layout_and_fold_reader(tmp_path):
    from jda_amd import fddb
    n = fddb.make_synthetic_fddb(str(tmp_path), n_images=23, seed=1, max_side=120, fmt="PNG")
    assert n == 23
    job = fddb.list_job(str(tmp_path))
    assert len(job) == 23 and [f for f, _ in job] == sorted(f for f, _ in job)
    sizes = [len(fddb.read_fold(fddb.fold_list_path(str(tmp_path), i))) for i in range(1, 11)]
    assert sizes == [3, 3, 3, 2, 2, 2, 2, 2, 2, 2]
    g = fddb.load_gray(os.path.join(str(tmp_path), "images", job[0][1] + ".jpg"))
    assert g.ndim == 2 and g.dtype == np.uint8 and max(g.shape) <= 120
    assert fddb.load_gray(os.path.join(str(tmp_path), "images", "missing.jpg")) is None


class FakeCascador:
    """Stands in for the GPU detector: 'detections' are a pure function of the image."""
    L = 3

    def detect_batch_cpp(self, frames, minimum_size, step, factor, overlap, nms, stats=False):
        g = frames[0]
        n = int(g[0, 0]) % 3
        res = dict(rects=np.array([[i, i + 1, 20 + i, 20 + i] for i in range(n)], np.int32).reshape(n, 4),
                   scores=np.array([float(g.mean()) + i for i in range(n)]),
                   shapes=np.full((n, 6), float(g.shape[1])))
        st = dict(patch_n=g.size, face_patch_n=n, nonface_patch_n=g.size - n, cart_gothrough_n=7 * (g.size - n))
        return ([res], st) if stats else [res]

    def detect_ragged_cpp(self, images, minimum_size, step, factor, overlap, nms, stats=False):
        """A fold's images as one job (fddb.run's default for dialect "cpp"): image by image the per-image answer."""
        res, tot = [], dict(patch_n=0, face_patch_n=0, nonface_patch_n=0, cart_gothrough_n=0)
        for g in images:
            (r,), st = self.detect_batch_cpp(g[None], minimum_size, step, factor, overlap, nms, stats=True)
            res.append(r)
            for k in tot:
                tot[k] += st[k]
        return (res, tot) if stats else res


def _expected_files(fddb_dir):
    from jda_amd import fddb
    c = FakeCascador()
    out = {i: "" for i in range(1, 11)}
    for fold, image_id in fddb.list_job(fddb_dir):
        g = fddb.load_gray(os.path.join(fddb_dir, "images", image_id + ".jpg"))
        r = c.detect_batch_cpp(g[None], 20, 5, 1.2, 0.3, True)[0]
        out[fold] += fddb.format_entry(image_id, r["rects"], r["scores"])
    return out


def test_run_single_process(tmp_path):
    from jda_amd import fddb
    d = str(tmp_path)
    fddb.make_synthetic_fddb(d, n_images=17, seed=2, max_side=90, fmt="PNG")
    os.remove(os.path.join(d, "images", fddb.list_job(d)[4][1] + ".jpg"))       # unreadable image -> skipped
    logs = []
    stats = fddb.run(FakeCascador(), d, log=logs.append)
    want = _expected_files(d) if False else None
    for i in range(1, 11):
        txt = open(fddb.fold_out_path(d, i)).read()
        ids = fddb.read_fold(fddb.fold_list_path(d, i))
        assert txt.count("\n") >= 2 * (len(ids) - (1 if i == 3 else 0))
    assert "Summary of ALL" in logs and sum(s.patch_n for s in stats.values()) > 0
    assert fddb.list_job(d)[4][1] not in open(fddb.fold_out_path(d, 3)).read()


def _worker(rank, world, port, d):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from jda_amd import fddb
    from test_fddb import FakeCascador
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fddb.run(FakeCascador(), d, rank=rank, world=world, device="cpu")
    dist.barrier()
    dist.destroy_process_group()


def test_run_sharded_over_gloo_matches_single(tmp_path):
    """2 ranks, contiguous image blocks, gather on rank 0: the ten files equal the 1-rank run."""
    import torch.multiprocessing as mp
    from jda_amd import fddb
    d = str(tmp_path)
    fddb.make_synthetic_fddb(d, n_images=21, seed=3, max_side=80, fmt="PNG")
    want = _expected_files(d)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, d)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    for i in range(1, 11):
        assert open(fddb.fold_out_path(d, i)).read() == want[i], i


@pytest.mark.gpu
@pytest.mark.parametrize("dialect,ragged", [("cpp", None), ("cpp", False), ("c", None), ("c", False)])
def test_fddb_end_to_end_gpu(built, model_file, tmp_path, dialect, ragged):
    """Real detector: every fold file equals what the oracle produces for the same decoded images (a fold as one ragged
    job of its dialect -- the default -- and image by image like the reference's loop)."""
    from jda_amd import api, fddb
    from oracle.pyoracle import Oracle
    d = str(tmp_path / "fddb")
    fddb.make_synthetic_fddb(d, n_images=12, seed=4, max_side=160, fmt="PNG")
    p, _ = model_file((3, 20, 5, 4), 8, seed=9, cart_th=-0.7, norm_every=6)
    c, o = api.Cascador(p), Oracle(p)
    stats = fddb.run(c, d, dialect=dialect, ragged=ragged)
    total = 0
    for i in range(1, 11):
        want = ""
        for image_id in fddb.read_fold(fddb.fold_list_path(d, i)):
            g = fddb.load_gray(os.path.join(d, "images", image_id + ".jpg"))
            if dialect == "cpp":
                r = o.detect_cpp(g, 20, 5, 1.2, 0.3, True)
                rects = r["rects"]
            else:
                r = o.detect(g, 1.25, 40, -1, -0.5)
                rects = np.concatenate([r["bboxes"], r["bboxes"][:, 2:3]], 1) if len(r["scores"]) else []
            want += fddb.format_entry(image_id, rects, r["scores"])
            total += len(r["scores"])
        assert open(fddb.fold_out_path(d, i)).read() == want, i
    assert total > 0 and sum(s.face_patch_n for s in stats.values()) >= total


@pytest.mark.gpu
def test_config3_fddb_sized_job_in_eight_shards(built, tmp_path):
    """BASELINE.json configs[3] at its stated size on the one GPU a test box has: 2,845 images of FDDB-like sizes
    (<= 450x450, every one its own scan plan), shipped model dimensions in the cascade regime, sharded into the 8
    contiguous blocks the 8 ranks would own (floor(i*G/N)), each block detected and packed into the
    rows a rank hands to the gather; the concatenation in rank order is the job's result in image order, and a
    sample of images is compared with the oracle (reference c/jda.c:443-480).  The RCCL exchange itself needs one
    GPU per rank: tests/test_gpu_parity.py::test_c_gather_entry_over_rccl_group_of_one, tests/test_dist_gloo.py."""
    from jda_amd import api, synth, dist as jd
    from oracle.pyoracle import Oracle
    from conftest import S_DIMS, same
    n_images, world = 2845, 8
    rng = np.random.default_rng(3)
    sizes = [(int(rng.integers(120, 451)), int(rng.integers(120, 451))) for _ in range(n_images)]
    m = synth.make_model(*S_DIMS, seed=1)
    synth.calibrate_thresholds(m, synth.make_frames(8, 450, 450, seed=0, first=10_000_000))
    p = str(tmp_path / "fddb.model"); m.save(p, 8)
    c, o = api.Cascador(p), Oracle(p)
    images = [synth.make_frames(1, w, h, seed=5, first=i)[0] for i, (w, h) in enumerate(sizes)]
    blocks, owned = [], 0
    for r in range(world):
        lo, hi = jd.shard_range(n_images, r, world)
        assert 355 <= hi - lo <= 356                                    # ceil(2845 / 8) per GPU
        owned += hi - lo
        res = [c.detect(images[i]) for i in range(lo, hi)]
        blocks.append(jd.pack_detections(res, c.L, frame_offset=lo))
    assert owned == n_images
    rows = np.concatenate(blocks)                                       # what rank 0 holds after the gather
    assert len(rows) > 0 and (np.diff(rows[:, 0]) >= 0).all()           # rank order == image order
    got = jd.unpack_detections(rows, c.L)
    for i in list(range(0, n_images, 97)) + [n_images - 1]:
        want = o.detect(images[i])
        if len(want["scores"]) == 0:
            assert i not in got
            continue
        for k in ("bboxes", "scores", "shapes"):
            assert same(got[i][k], want[k]), (i, k)
    # The entry the bench times for this config -- every shard as ONE ragged job (jdaDetectBatchRagged), the images
    # of a shard back to back in one packed buffer, packed rows out -- must give the same rows as the per-image calls
    # above, shard by shard, on all 2,845 images (and therefore what the oracle gives on the sampled ones).
    for r in range(world):
        lo, hi = jd.shard_range(n_images, r, world)
        offs, tot = [], 0
        for i in range(lo, hi):
            offs.append(tot); tot += images[i].size
        buf = np.concatenate([images[i].reshape(-1) for i in range(lo, hi)])
        rag = c.detect_ragged_packed(buf, offs, [sizes[i][0] for i in range(lo, hi)], [sizes[i][1] for i in range(lo, hi)],
                                     keep_results="packed", frame_offset=lo)
        assert same(np.asarray(rag), blocks[r]), r
