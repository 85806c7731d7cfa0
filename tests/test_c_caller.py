"""Boundary proof in C: tests/c/main_like.c replays the reference's demo driver (reference
c/main.cpp:11-53: CreateDouble -> SerializeTo -> Release -> CreateFloat -> jdaDetect x 10 ->
ResultRelease -> Release) as a plain C99 program.  It is compiled twice with gcc -- once against
jda_amd/libjda.so, once against the reference's own c/jda.c build (oracle/_ref) -- and both must
print the same bytes for the same model file and frame; the float model files both write must be
identical too (header quirk of c/jda.c:652-665 included)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import S_DIMS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "main_like.c")


def _compile(tmp_path, name, libpath):
    exe = str(tmp_path / name)
    libdir, lib = os.path.split(libpath)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC, "-o", exe,
                           "-L", libdir, "-l:" + lib, "-Wl,-rpath," + libdir])
    return exe


def test_c_caller_compiles_and_links(built, tmp_path):
    """CPU part: the caller is valid C99 against include/jda.h and links the product library."""
    from jda_amd import api
    _compile(tmp_path, "main_like", api.LIB_PATH)


@pytest.mark.gpu
def test_c_caller_matches_the_reference_build(built, tmp_path):
    from jda_amd import api, synth
    from oracle import pyoracle
    calib = synth.make_frames(8, 640, 480, seed=0, first=10_000_000)
    m = synth.make_model(*S_DIMS, seed=1)
    synth.calibrate_thresholds(m, calib)
    model = str(tmp_path / "jda.model")
    m.save(model, 8)
    assert os.path.getsize(model) == 10476464                              # SURVEY.md header item 4
    frame = synth.make_frames(1, 640, 480, seed=5)[0]
    raw = str(tmp_path / "gray.raw")
    frame.tofile(raw)
    ours = _compile(tmp_path, "main_ours", api.LIB_PATH)
    f32_ours = str(tmp_path / "ours_float32.model")
    a = subprocess.run([ours, model, f32_ours, raw, "640", "480"], capture_output=True, text=True, timeout=600)
    assert a.returncode == 0, a.stderr
    assert a.stdout.startswith("01 n=")
    assert os.path.getsize(f32_ours) == 5389448
    lines = a.stdout.splitlines()
    assert len(lines) >= 10 and int(lines[0].split("n=")[1].split()[0]) > 0, "the frame must have detections"
    ref_lib = pyoracle.reference_lib_path(*S_DIMS)
    if ref_lib is None:
        pytest.skip("no compiled reference for the shipped dimensions travelled to this box")
    theirs = _compile(tmp_path, "main_ref", ref_lib)
    f32_ref = str(tmp_path / "ref_float32.model")
    b = subprocess.run([theirs, model, f32_ref, raw, "640", "480"], capture_output=True, text=True, timeout=600)
    assert b.returncode == 0, b.stderr
    assert a.stdout == b.stdout
    assert open(f32_ours, "rb").read() == open(f32_ref, "rb").read()
