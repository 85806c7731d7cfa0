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
