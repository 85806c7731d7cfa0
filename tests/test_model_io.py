"""Model stream I/O of the C ABI (no GPU): reference c/jda.c:486-716, README.md:84-111."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import TINY_DIMS, S_DIMS


def test_stream_sizes_match_survey(built):
    from jda_amd import api
    # shipped dims double 10,476,464 B, float 5,389,448 B; config-5 dims
    assert api.lib.jdaModelStreamBytes(5, 540, 27, 4, 8) == 10476464
    assert api.lib.jdaModelStreamBytes(5, 540, 27, 4, 4) == 5389448
    assert api.lib.jdaModelStreamBytes(7, 2000, 68, 6, 8) == 512177120
    assert api.lib.jdaModelStreamBytes(7, 2000, 68, 6, 4) == 259560576


@pytest.mark.parametrize("dims", TINY_DIMS)
def test_double_to_float_roundtrip(built, model_file, tmp_path, dims):
    """CreateDouble -> SerializeTo -> CreateFloat, the sequence of reference c/main.cpp:11-14."""
    from jda_amd import api
    p8, m = model_file(dims, 8, seed=5, cart_th=-1.0, norm_every=3)
    h = api.jdaCascadorCreateDouble(os.fsencode(p8))
    assert h
    p4 = str(tmp_path / "f32.model")
    api.jdaCascadorSerializeTo(h, os.fsencode(p4))
    api.jdaCascadorRelease(h)
    blob = open(p4, "rb").read()
    assert blob == m.tobytes(4)                      # same bytes our numpy writer produces
    hdr = np.frombuffer(blob[:28], "<i4")
    assert list(hdr) == [0, dims[0], dims[1], dims[2], dims[3], dims[0] + 1, -1]   # c/jda.c:652-665
    h2 = api.jdaCascadorCreateFloat(os.fsencode(p4))
    assert h2
    p4b = str(tmp_path / "f32b.model")
    api.jdaCascadorSerializeTo(h2, os.fsencode(p4b))
    api.jdaCascadorRelease(h2)
    assert open(p4b, "rb").read() == blob             # float -> float is the identity


def test_narrowing_is_a_plain_cast(built, model_file, tmp_path):
    """A model with genuine doubles narrows like (float)f8 (c/jda.c:509-552)."""
    from jda_amd import api
    p8, m = model_file((2, 8, 5, 3), 8, seed=9, f32_exact=False)
    c = api.Cascador(p8, "double")
    p4 = str(tmp_path / "n.model")
    c.serialize(p4)
    assert open(p4, "rb").read() == m.tobytes(4)       # numpy astype(f4) == C cast (round to nearest)


def test_matches_reference_serializer(built, model_file, tmp_path):
    """Byte-identical float file to the reference's own jdaCascadorSerializeTo."""
    from jda_amd import api
    from oracle import pyoracle
    dims = (3, 20, 5, 4)
    if pyoracle.reference_lib_path(*dims) is None:
        pytest.skip("no reference build available")
    p8, m = model_file(dims, 8, seed=2, cart_th=-0.7, f32_exact=False)
    ref = pyoracle.Reference(p8, dims, 8)
    pr = str(tmp_path / "ref.model")
    ref.serialize(pr)
    c = api.Cascador(p8, "double")
    po = str(tmp_path / "ours.model")
    c.serialize(po)
    assert open(pr, "rb").read() == open(po, "rb").read()


def test_create_errors(built, model_file, tmp_path):
    from jda_amd import api
    assert not api.jdaCascadorCreateDouble(b"/nonexistent/file.model")    # NULL like c/jda.c:487-488
    assert not api.jdaCascadorCreateFloat(b"/nonexistent/file.model")
    p8, m = model_file((2, 8, 5, 3), 8)
    # wrong real type for the file, truncated file, garbage header: refused (NULL), never a crash
    assert not api.jdaCascadorCreateFloat(os.fsencode(p8))
    blob = open(p8, "rb").read()
    pt = str(tmp_path / "trunc.model")
    open(pt, "wb").write(blob[:-9])
    assert not api.jdaCascadorCreateDouble(os.fsencode(pt))
    assert "size" in api.last_error()
    pg = str(tmp_path / "garbage.model")
    open(pg, "wb").write(b"\xff" * 4096)
    assert not api.jdaCascadorCreate(os.fsencode(pg))
    api.jdaCascadorRelease(None)                                          # NULL accepted
    api.jdaCascadorSerializeTo(None, b"/tmp/never")


def test_info_and_autodetect(built, model_file):
    from jda_amd import api
    for rb in (8, 4):
        p, m = model_file((3, 70, 9, 5), rb, multi_scale=True)
        c = api.Cascador(p)
        assert (c.T, c.K, c.L, c.D) == (3, 70, 9, 5)
        assert c.source_real_bytes == rb and c.multi_scale


def test_partial_training_header_is_accepted(built, tmp_path):
    """Snapshot files carry (stage_idx, cart_idx) of the training status (cascador.cpp:93-104);
    the C loader ignores them (c/jda.c:499-505)."""
    from jda_amd import api, synth
    m = synth.make_model(2, 8, 5, 3, seed=1)
    p = str(tmp_path / "snap.model")
    m.save(p, 8, header_stage=1, header_cart=4)
    assert api.Cascador(p).T == 2


def test_shipped_dims_plumbing_cpu(built, tmp_path):
    """BASELINE.json configs[0] up to the GPU boundary: S-dims double file (10,476,464 B) ->
    SerializeTo -> float file (5,389,448 B) -> CreateFloat."""
    from jda_amd import api, synth
    m = synth.make_model(*S_DIMS, seed=1)
    p8 = str(tmp_path / "jda.model")
    m.save(p8, 8)
    assert os.path.getsize(p8) == 10476464
    c = api.Cascador(p8, "double")
    p4 = str(tmp_path / "jda_float32.model")
    c.serialize(p4)
    assert os.path.getsize(p4) == 5389448
    c2 = api.Cascador(p4, "float")
    assert (c2.T, c2.K, c2.L, c2.D) == S_DIMS


def test_impossible_training_status_is_refused_by_the_cpp_entries(built, tmp_path):
    """Header ints 5, 6 (current_stage_idx, current_cart_idx; cascador.cpp:93-104): dialect CPP's Validate stops there
    (cascador.cpp:178,199-209).  A training snapshot is run the way Validate runs it (tests/test_cpp_entries.py, GPU); a
    status the reference's own loader asserts against (cascador.cpp:138-141) is refused, not run to the end silently.  Dialect C ignores the header (c/jda.c:499-505).  No GPU needed: the
    refusal comes before any device work."""
    import numpy as np
    from jda_amd import api, synth
    m = synth.make_model(3, 8, 5, 3, seed=2)
    full = str(tmp_path / "full.model"); m.save(full, 8)
    frame = np.zeros((1, 60, 80), np.uint8)
    for hs, hc in ((1, 8), (3, 2), (-1, -1), (1, -2), (5, -1)):
        bad = str(tmp_path / ("bad_%d_%d.model" % (hs, hc))); m.save(bad, 8, header_stage=hs, header_cart=hc)
        c = api.Cascador(bad)
        for call in (lambda: c.detect_batch_cpp(frame), lambda: c.trace_cpp(frame), lambda: c.detect_batch_cpp_pyramid(frame)):
            with pytest.raises(api.JdaError, match="partial model"):
                call()
    assert api.Cascador(full).T == 3


def test_mutated_streams_are_refused_or_loaded_never_a_crash(built):
    """tools/fuzz_model.py: truncated, resized, bit-flipped, padded model files and files given to the other real size's
    creator -- every create returns NULL with a reason or a handle that serialises; (the same script runs on the sanitizer
    build: profiles/r05_host_asan.txt)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_model.py"), "400"], capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "no crash" in out.stdout and " 0 creates accepted" not in out.stdout, out.stdout[-500:]
