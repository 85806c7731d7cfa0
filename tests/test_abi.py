"""The C-ABI shared library: loads without a GPU, exports every symbol that
include/jda.h declares, the header is valid C, and a plain C program can link it."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "jda.h")

REFERENCE_SYMBOLS = ["jdaCascadorCreateDouble", "jdaCascadorCreateFloat", "jdaCascadorSerializeTo",
                     "jdaCascadorRelease", "jdaDetect", "jdaResultRelease"]    # reference c/jda.h:31-68


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.findall(r"JDA_API\s+[^;(]*?\b(jda\w+)\s*\(", src)


def test_every_declared_symbol_is_exported(built):
    from jda_amd import api
    names = declared_symbols()
    assert set(REFERENCE_SYMBOLS) <= set(names)
    assert len(names) >= 15
    for n in names:
        assert hasattr(api.lib, n), n


def test_header_is_plain_c_and_links(built, tmp_path):
    """What a maintainer of the reference would do: compile a C caller against
    include/jda.h and link libjda.so (no torch, no C++)."""
    from jda_amd import api
    src = tmp_path / "caller.c"
    src.write_text(r'''
#include <stdio.h>
#include "jda.h"
int main(void) {
  void *c = jdaCascadorCreateDouble("/nonexistent.model");
  if (c != NULL) return 1;                      /* NULL on a missing file, c/jda.c:487-488 */
  long long n = 0; int levels = 0;
  if (jdaCountWindows(640, 480, 1.25f, 40, -1, &n, &levels) != 0) return 2;
  printf("%lld %d\n", n, levels);
  jdaResult r = {0, 0, NULL, NULL, NULL};
  jdaResultRelease(r);                          /* free(NULL) is fine */
  jdaCascadorRelease(NULL);
  return 0;
}
''')
    exe = tmp_path / "caller"
    libdir = os.path.dirname(api.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src),
                           "-o", str(exe), "-L", libdir, "-l:libjda.so", "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split() == ["38245", "11"]


def test_detect_without_gpu_fails_loudly(built, model_file):
    """No CPU fallback: on a box without a HIP device the detect entry must raise, not
    quietly compute elsewhere."""
    import numpy as np
    import torch
    from jda_amd import api
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    p, _ = model_file((2, 8, 5, 3), 8)
    c = api.Cascador(p)
    with pytest.raises(api.JdaError, match="no usable HIP device"):
        c.detect(np.zeros((64, 64), np.uint8))
    with pytest.raises(api.JdaError):
        c.detect_batch(np.zeros((2, 64, 64), np.uint8))
    with pytest.raises(api.JdaError):
        c.trace(np.zeros((64, 64), np.uint8))


def test_product_never_touches_the_oracle():
    """oracle/ is test infrastructure: nothing under jda_amd/ may import or load it."""
    pkg = os.path.join(ROOT, "jda_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "pyoracle" not in txt and "libjda_oracle" not in txt and "oracle/_ref" not in txt, f
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
    deps = subprocess.run(["readelf", "-d", os.path.join(pkg, "libjda.so")], capture_output=True, text=True).stdout
    assert "oracle" not in deps


def test_cmake_target_builds_the_same_library(tmp_path):
    """CMakeLists.txt (shape of reference c/CMakeLists.txt:19-22): one shared library exporting c/jda.h
    plus the demo driver, built without Python."""
    import shutil
    if not shutil.which("cmake") or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("cmake or hipcc not available")
    b = str(tmp_path / "b")
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    subprocess.check_call(["cmake", "-S", ROOT, "-B", b] + gen, stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--build", b, "-j", "8"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(b, "libjda.so"))
    for n in declared_symbols():
        assert hasattr(lib, n), n
    assert os.path.exists(os.path.join(b, "jda-test"))


def test_dist_library_exports_its_header(built):
    """libjda_dist.so (RCCL gather of detection rows): loads without a GPU and exports every symbol of include/jda_dist.h;
    libjda.so itself does not depend on RCCL."""
    from jda_amd import build as lib_build
    from jda_amd import dist as jd
    lib_build.build_dist()
    src = open(os.path.join(ROOT, "include", "jda_dist.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"JDA_API\s+[^;(]*?\b(jda\w+)\s*\(", src)
    assert len(names) >= 9
    L = jd.dist_lib()
    for n in names:
        assert hasattr(L, n), n
    deps = subprocess.run(["readelf", "-d", os.path.join(ROOT, "jda_amd", "libjda.so")], capture_output=True, text=True).stdout
    assert "rccl" not in deps and "torch" not in deps


def test_options_round_trip_and_start_from_the_environment(built, model_file, monkeypatch):
    """jdaSetOption / jdaGetOption (include/jda.h): per cascador, seeded ONCE from JDA_<KEY> when the cascador is created,
    unknown keys refused.  No GPU needed: the options live in the host object."""
    from jda_amd import api
    p, _ = model_file((2, 8, 5, 3), 8, seed=1)
    documented = {"handoff": 128, "lanes": 2, "dense": 1, "plan_cache": 64, "predict": 1, "wide_max": 1024,
                  "ragged_chunk_windows": 4000000, "ragged_chunk_windows_cpp": 8000000, "filter0": 1,
                  "kernel_d2h": 1, "h2d_stream": 1, "h2d_min_bytes": 8 << 20,
                  "ragged_chunk_min_windows": 1500000, "ragged_split": 3, "ragged_single_windows": 5000000,
                  "scan_p": 1, "scan_p_slots": 5, "device_post": 1, "device_post_min_frames": 16,
                  "w_pad": 1, "w_stream_mb": 8, "lm_deep": 1, "max_lanes": 16, "hwq_place": 1, "ws_bound": 1, "ws_factor_pct": 400, "ws_min_entries": 65536}
    for k in list(os.environ):
        if k.startswith("JDA_") and k not in ("JDA_LIB_PATH",):
            monkeypatch.delenv(k)
    c = api.Cascador(p)
    for k, v in documented.items():
        assert c.get_option(k) == v, k
    monkeypatch.setenv("JDA_HANDOFF", "96"); monkeypatch.setenv("JDA_H2D_STREAM", "2")
    c2 = api.Cascador(p)                                   # the environment is read when the cascador is created ...
    assert c2.get_option("handoff") == 96 and c2.get_option("h2d_stream") == 2
    assert c.get_option("handoff") == 128                  # ... and only then
    monkeypatch.setenv("JDA_HANDOFF", "64")
    assert c2.get_option("handoff") == 96
    c2.set_option("handoff", 112)
    assert c2.get_option("handoff") == 112 and c.get_option("handoff") == 128
    for k in ("scan_p_tile_kb", "scan_p_grid", "ragged_chunk_min_windows", "workspace_mb"):      # sizes and counts: no negative values
        with pytest.raises(api.JdaError):
            c.set_option(k, -1)
    with pytest.raises(api.JdaError):
        c.set_option("no_such_option", 1)
    assert api.lib.jdaGetOption(c.h, b"no_such_option") == -1
    c.close(); c2.close()


def test_no_exception_crosses_the_c_abi(built, tmp_path):
    """Reference behaviour on an allocation failure: NULL / nothing, never a crash (c/jda.c:487-493).  Every extern "C"
    entry of abi.cpp is a function-try-block; the `test_throw` option makes jdaDetectBatch throw inside it -- the call
    returns -1 (jdaDetect: an empty result), jdaGetLastError says why, and the process lives."""
    import numpy as np
    from jda_amd import api, synth
    p = str(tmp_path / "m.model")
    synth.make_model(2, 8, 5, 3, seed=1).save(p, 8)
    c = api.Cascador(p)
    frames = synth.make_frames(2, 64, 48, seed=1)
    for code, text in ((1, "out of host memory"), (2, "injected by test_throw")):
        c.set_option("test_throw", code)
        with pytest.raises(api.JdaError, match=text):
            c.detect_batch(frames)
        assert "jdaDetectBatch" in api.last_error()
        img = np.ascontiguousarray(frames[0])
        r = api.lib.jdaDetect(c.h, img.ctypes.data_as(C.POINTER(C.c_ubyte)), 64, 48, 1.25, 0.1, 24, -1, -0.5)
        assert r.n == 0 and r.landmark_n == 5 and text in api.last_error()
        api.lib.jdaResultRelease(r)
    c.set_option("test_throw", 0)
    # the cascador is still usable: nothing was left locked or pinned by the unwinding
    assert c.get_option("test_throw") == 0
    c.set_option("handoff", 64)
    c.close()
    # a model file that cannot be read is NULL + a reason, as before
    with pytest.raises(api.JdaError):
        api.Cascador(str(tmp_path / "missing.model"))
