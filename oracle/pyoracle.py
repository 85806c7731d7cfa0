"""ctypes bindings of the CPU checkers (TEST INFRASTRUCTURE, not product).

  Oracle     -- oracle/libjda_oracle.so, our C restatement (jda_oracle.c)
  Reference  -- oracle/_ref/libjda_ref_<dims>.so, the reference's own c/jda.c
                compiled by oracle/build.py (present only if it was built in a
                container that has /root/reference; the files travel)

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

u8p = C.POINTER(C.c_ubyte)


def _ptr(a, ty):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


class Oracle:
    def __init__(self, model_path):
        self.lib = C.CDLL(_build.build_oracle())
        L = self.lib
        L.orc_load.restype = C.c_void_p
        L.orc_load.argtypes = [C.c_char_p]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.orc_count_windows_c.restype = C.c_longlong
        L.orc_count_windows_c.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.orc_count_windows_cpp.restype = C.c_longlong
        L.orc_count_windows_cpp.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]
        L.orc_trace_c.restype = C.c_longlong
        L.orc_trace_c.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                  C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_uint), C.POINTER(C.c_float)]
        L.orc_detect_c.restype = C.c_int
        L.orc_detect_c.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.c_float, C.c_int,
                                   C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_trace_cpp.restype = C.c_longlong
        L.orc_trace_cpp.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                    C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_uint), C.POINTER(C.c_double)]
        L.orc_detect_cpp.restype = C.c_int
        L.orc_detect_cpp.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int,
                                     C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_resize.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
        L.orc_resize_cv.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
        L.orc_count_windows_pyramid.restype = C.c_longlong
        L.orc_count_windows_pyramid.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int)]
        L.orc_detect_cpp_pyramid.restype = C.c_int
        L.orc_detect_cpp_pyramid.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                             C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_detect_cpp_pyramid_ms.restype = C.c_int
        L.orc_detect_cpp_pyramid_ms.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                                C.c_double, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_pyramid_dims.argtypes = [C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4
        self.h = L.orc_load(os.fsencode(model_path))
        if not self.h:
            raise RuntimeError("oracle could not load model %s" % model_path)
        d = (C.c_int * 6)()
        L.orc_dims(self.h, d)
        self.T, self.K, self.L, self.D, self.real_bytes, self.dim = list(d)

    def close(self):
        if self.h:
            self.lib.orc_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def count_windows(self, w, h, scale=1.25, min_size=40, max_size=-1):
        nl = C.c_int()
        n = self.lib.orc_count_windows_c(w, h, scale, min_size, max_size, C.byref(nl))
        return n, nl.value

    def trace(self, img, scale=1.25, min_size=40, max_size=-1, want_shapes=True):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        n, _ = self.count_windows(w, h, scale, min_size, max_size)
        if n < 0:
            raise ValueError("scan does not terminate")
        carts = np.zeros(n, np.int32)
        score = np.zeros(n, np.float32)
        hsh = np.zeros(n, np.uint32)
        shapes = np.zeros((n, self.dim), np.float32) if want_shapes else None
        got = self.lib.orc_trace_c(self.h, _ptr(img, C.c_ubyte), w, h, scale, min_size, max_size,
                                   _ptr(carts, C.c_int), _ptr(score, C.c_float), _ptr(hsh, C.c_uint),
                                   _ptr(shapes, C.c_float))
        assert got == n
        return dict(carts_n=carts, score=score, path_hash=hsh, shapes=shapes)

    def detect(self, img, scale=1.25, min_size=40, max_size=-1, th=-0.5, nms=True):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        n, _ = self.count_windows(w, h, scale, min_size, max_size)
        if n < 0:
            raise ValueError("scan does not terminate")
        bb = np.zeros((max(n, 1), 3), np.int32)
        sc = np.zeros(max(n, 1), np.float32)
        sh = np.zeros((max(n, 1), self.dim), np.float32)
        k = self.lib.orc_detect_c(self.h, _ptr(img, C.c_ubyte), w, h, scale, min_size, max_size, th, int(nms),
                                  _ptr(bb, C.c_int), _ptr(sc, C.c_float), _ptr(sh, C.c_float))
        if k < 0:
            raise RuntimeError("oracle detect failed")
        return dict(bboxes=bb[:k].copy(), scores=sc[:k].copy(), shapes=sh[:k].copy())

    def trace_cpp(self, img, minimum_size=20, step=5, factor=1.2):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        nl = C.c_int()
        n = self.lib.orc_count_windows_cpp(w, h, minimum_size, step, factor, C.byref(nl))
        if n < 0:
            raise ValueError("bad scan parameters")
        carts = np.zeros(n, np.int32)
        score = np.zeros(n, np.float64)
        hsh = np.zeros(n, np.uint32)
        shapes = np.zeros((n, self.dim), np.float64)
        got = self.lib.orc_trace_cpp(self.h, _ptr(img, C.c_ubyte), w, h, minimum_size, step, factor,
                                     _ptr(carts, C.c_int), _ptr(score, C.c_double), _ptr(hsh, C.c_uint),
                                     _ptr(shapes, C.c_double))
        if got != n:
            raise RuntimeError("oracle cpp trace failed (multi-scale model?)")
        return dict(carts_n=carts, score=score, path_hash=hsh, shapes=shapes)

    def detect_cpp(self, img, minimum_size=20, step=5, factor=1.2, overlap=0.3, nms=True):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        nl = C.c_int()
        n = self.lib.orc_count_windows_cpp(w, h, minimum_size, step, factor, C.byref(nl))
        if n < 0:
            raise ValueError("bad scan parameters")
        rc = np.zeros((max(n, 1), 4), np.int32)
        sc = np.zeros(max(n, 1), np.float64)
        sh = np.zeros((max(n, 1), self.dim), np.float64)
        k = self.lib.orc_detect_cpp(self.h, _ptr(img, C.c_ubyte), w, h, minimum_size, step, factor, overlap, int(nms),
                                    _ptr(rc, C.c_int), _ptr(sc, C.c_double), _ptr(sh, C.c_double))
        if k < 0:
            raise RuntimeError("oracle cpp detect failed")
        return dict(rects=rc[:k].copy(), scores=sc[:k].copy(), shapes=sh[:k].copy())

    def resize(self, img, dw, dh):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros((dh, dw), np.uint8)
        self.lib.orc_resize(_ptr(img, C.c_ubyte), w, h, _ptr(out, C.c_ubyte), dw, dh)
        return out

    def set_similarity_transform(self, on):
        """Global switch of the oracle's dialect CPP (Config::with_similarity_transform)."""
        self.lib.orc_set_similarity_transform(1 if on else 0)

    def resize_cv(self, img, dw, dh):
        """Restatement of cv::resize(INTER_LINEAR) for 8-bit gray (parity unpinned)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros((dh, dw), np.uint8)
        self.lib.orc_resize_cv(_ptr(img, C.c_ubyte), w, h, _ptr(out, C.c_ubyte), dw, dh)
        return out

    def detect_cpp_pyramid(self, img, origin_size=48, step=5, factor=1.2, overlap=0.3, nms=True, half_size=0, quarter_size=0):
        """half_size / quarter_size > 0: the per-window patches of a multi-scale model (cascador.cpp:243-245)."""
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        nl = C.c_int()
        n = self.lib.orc_count_windows_pyramid(w, h, origin_size, step, factor, C.byref(nl))
        if n < 0:
            raise ValueError("bad pyramid parameters")
        rc = np.zeros((max(n, 1), 4), np.int32)
        sc = np.zeros(max(n, 1), np.float64)
        sh = np.zeros((max(n, 1), self.dim), np.float64)
        k = self.lib.orc_detect_cpp_pyramid_ms(self.h, _ptr(img, C.c_ubyte), w, h, origin_size, half_size, quarter_size, step,
                                               factor, overlap, int(nms), _ptr(rc, C.c_int), _ptr(sc, C.c_double), _ptr(sh, C.c_double))
        if k < 0:
            raise RuntimeError("oracle pyramid detect failed (multi-scale model?)")
        return dict(rects=rc[:k].copy(), scores=sc[:k].copy(), shapes=sh[:k].copy(), windows=n, levels=nl.value)

    def pyramid_dims(self, w, h):
        v = [C.c_int() for _ in range(4)]
        self.lib.orc_pyramid_dims(w, h, *[C.byref(x) for x in v])
        return tuple(x.value for x in v)


class _RefResult(C.Structure):
    _fields_ = [("n", C.c_int), ("landmark_n", C.c_int), ("bboxes", C.POINTER(C.c_int)),
                ("shapes", C.POINTER(C.c_float)), ("scores", C.POINTER(C.c_float))]


def reference_lib_path(T, K, L, D):
    p = _build.ref_path(T, K, L, D)
    if os.path.exists(p):
        return p
    if _build.reference_available():
        return _build.build_ref(T, K, L, D)
    return None


class Reference:
    """The reference's own compiled c/jda.c for one dimension set."""

    def __init__(self, model_path, dims, real_bytes):
        p = reference_lib_path(*dims)
        if p is None:
            raise FileNotFoundError("no reference build for dims %s" % (dims,))
        self.lib = C.CDLL(p)
        L = self.lib
        for f in ("jdaCascadorCreateDouble", "jdaCascadorCreateFloat"):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [C.c_char_p]
        L.jdaCascadorSerializeTo.argtypes = [C.c_void_p, C.c_char_p]
        L.jdaCascadorRelease.argtypes = [C.c_void_p]
        sig = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float]
        L.jdaDetect.restype = _RefResult
        L.jdaDetect.argtypes = sig
        L.ref_detect_raw.restype = _RefResult
        L.ref_detect_raw.argtypes = sig
        L.jdaResultRelease.argtypes = [_RefResult]
        L.ref_resize.argtypes = [u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
        L.ref_dims.argtypes = [C.POINTER(C.c_int)]
        d = (C.c_int * 4)()
        L.ref_dims(d)
        assert tuple(d) == tuple(dims)
        self.dims = tuple(dims)
        self.dim = 2 * dims[2]
        create = L.jdaCascadorCreateDouble if real_bytes == 8 else L.jdaCascadorCreateFloat
        self.h = create(os.fsencode(model_path))
        if not self.h:
            raise RuntimeError("reference could not open %s" % model_path)

    def close(self):
        if self.h:
            self.lib.jdaCascadorRelease(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def serialize(self, path):
        self.lib.jdaCascadorSerializeTo(self.h, os.fsencode(path))

    def _run(self, fn, img, scale, min_size, max_size, th):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        r = fn(self.h, _ptr(img, C.c_ubyte), w, h, scale, 0.1, min_size, max_size, th)
        n = r.n
        out = dict(bboxes=np.ctypeslib.as_array(r.bboxes, (n, 3)).copy() if n else np.zeros((0, 3), np.int32),
                   scores=np.ctypeslib.as_array(r.scores, (n,)).copy() if n else np.zeros(0, np.float32),
                   shapes=np.ctypeslib.as_array(r.shapes, (n, self.dim)).copy() if n else np.zeros((0, self.dim), np.float32))
        self.lib.jdaResultRelease(r)
        return out

    def detect(self, img, scale=1.25, min_size=40, max_size=-1, th=-0.5):
        return self._run(self.lib.jdaDetect, img, scale, min_size, max_size, th)

    def detect_raw(self, img, scale=1.25, min_size=40, max_size=-1, th=-0.5):
        """Pre-NMS, pre-relocation survivors (jdaInternalDetect)."""
        return self._run(self.lib.ref_detect_raw, img, scale, min_size, max_size, th)

    def resize(self, img, dw, dh):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros((dh, dw), np.uint8)
        self.lib.ref_resize(_ptr(img, C.c_ubyte), w, h, _ptr(out, C.c_ubyte), dw, dh)
        return out
