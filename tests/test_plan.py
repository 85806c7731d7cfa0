"""Window enumeration (reference c/jda.c:320-339,459-460; cascador.cpp:310-376)."""
import pytest


def test_counts_from_survey(built):
    from jda_amd import api
    # SURVEY.md 8a-3
    assert api.count_windows(640, 480, 1.25, 40, -1) == (38245, 11)
    assert api.count_windows(1920, 1080, 1.25, 40, -1) == (303222, 15)
    assert api.count_windows(1920, 1080, 1.5, 40, -1) == (125350, 8)


@pytest.mark.parametrize("w,h,scale,mn,mx", [
    (640, 480, 1.25, 40, -1), (200, 150, 1.25, 40, -1), (97, 131, 1.1, 10, -1), (320, 240, 1.4, 24, 100),
    (64, 64, 2.0, 30, 0), (23, 50, 1.25, 40, -1), (24, 24, 1.25, 0, -1), (450, 337, 1.2, 20, 333)])
def test_agrees_with_oracle_and_python(built, w, h, scale, mn, mx):
    from jda_amd import api, synth
    from oracle.pyoracle import Oracle  # noqa: F401  (enumeration is model independent)
    import ctypes as C
    from oracle import build as ob
    lib = C.CDLL(ob.build_oracle())
    lib.orc_count_windows_c.restype = C.c_longlong
    lib.orc_count_windows_c.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_int)]
    nl = C.c_int()
    n = lib.orc_count_windows_c(w, h, scale, mn, mx, C.byref(nl))
    assert api.count_windows(w, h, scale, mn, mx) == (n, nl.value)
    lv, tot = synth.levels_c(w, h, scale, mn, mx)
    assert (tot, len(lv)) == (n, nl.value)
    xs, ys, ws = synth.window_table(w, h, scale, mn, mx)
    assert len(xs) == n
    if n:
        assert (xs + ws).max() <= w and (ys + ws).max() <= h


def test_non_growing_scale_is_refused(built):
    """`win_size *= scale` never grows for scale <= 1+1/24: the reference loops forever (c/jda.c:331)."""
    from jda_amd import api
    for s in (1.0, 0.9, 1.04):
        with pytest.raises(api.JdaError):
            api.count_windows(640, 480, s, 40, -1)


def test_cpp_counts(built):
    from jda_amd import synth
    # model/config.json fddb params on 640x480 -> 140,215 windows / 19 levels
    lv, tot = synth.levels_cpp(640, 480, 20, 5, 1.2)
    assert (tot, len(lv)) == (140215, 19)


@pytest.mark.parametrize("w,h,scale", [(640, 480, 1.25), (1920, 1080, 1.5), (1920, 1080, 1.25), (131, 97, 1.2), (450, 337, 1.25)])
def test_scan_tiles_cover_every_window_and_fit_lds(built, model_file, w, h, scale):
    """k_scan's tile plan (host side, no GPU): every window of a tiled level belongs to exactly one tile of at most
    512 windows, the LDS rows cover the tile plus the 16-byte lead-in of unaligned tile origins, node offsets
    fit their packing (16 bits in mode 1, 21 bits in modes 2/3) and a pixel tile fits the 160 KiB of a CU."""
    from jda_amd import api
    p, _ = model_file((2, 8, 5, 3), 8)
    c = api.Cascador(p)
    levels = c.plan_tiles(w, h, scale, 24, -1)
    assert len(levels) == api.count_windows(w, h, scale, 24, -1)[1]
    for lv in levels:
        assert lv["mode"] in (1, 2, 3)
        assert lv["tiles_x"] * lv["tw"] >= lv["nx"] and (lv["tiles_x"] - 1) * lv["tw"] < lv["nx"]
        assert lv["tiles_y"] * lv["th"] >= lv["ny"] and (lv["tiles_y"] - 1) * lv["th"] < lv["ny"]
        assert lv["tw"] * lv["th"] <= 512
        if lv["mode"] == 2:
            assert lv["pitch"] == w and (lv["win"] - 1) * w + lv["win"] - 1 < 2 ** 21
            continue
        pw = lv["win"] + (lv["tw"] - 1) * lv["step"]
        ph = lv["win"] + (lv["th"] - 1) * lv["step"]
        lead = max((tx * lv["tw"] * lv["step"]) & 15 for tx in range(lv["tiles_x"]))
        assert lv["pitch"] % 16 == 0 and lv["pitch"] >= lead + pw
        assert lv["pitch"] * ph <= 160 * 1024
        top = (lv["win"] - 1) * lv["pitch"] + lv["win"] - 1 + 15
        assert top <= 65535 if lv["mode"] == 1 else top < 2 ** 21
