"""ctypes binding of libjda.so -- the host-side mirror of the reference's C API.

The function names, argument order and ownership rules are the reference's
(reference c/jda.h:31-68): jdaCascadorCreateDouble / jdaCascadorCreateFloat /
jdaCascadorSerializeTo / jdaCascadorRelease / jdaDetect / jdaResultRelease.
On top of that sit thin numpy-friendly wrappers for the additive batch and
trace entry points of include/jda.h.

There is no fallback: if libjda.so is missing this module raises at import
time, and if no HIP device is usable every detect call raises JdaError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JDA_LIB_PATH") or os.path.join(_HERE, "libjda.so")

JDA_DIALECT_C = 0
JDA_DIALECT_CPP = 1


class JdaError(RuntimeError):
    pass


class jdaResult(C.Structure):
    _fields_ = [("n", C.c_int), ("landmark_n", C.c_int), ("bboxes", C.POINTER(C.c_int)),
                ("shapes", C.POINTER(C.c_float)), ("scores", C.POINTER(C.c_float))]


class jdaResultD(C.Structure):
    _fields_ = [("n", C.c_int), ("landmark_n", C.c_int), ("rects", C.POINTER(C.c_int)),
                ("shapes", C.POINTER(C.c_double)), ("scores", C.POINTER(C.c_double))]


class jdaModelInfo(C.Structure):
    _fields_ = [("T", C.c_int), ("K", C.c_int), ("landmark_n", C.c_int), ("tree_depth", C.c_int),
                ("multi_scale", C.c_int), ("source_real_bytes", C.c_int)]


class jdaStats(C.Structure):
    _fields_ = [("patch_n", C.c_longlong), ("face_patch_n", C.c_longlong), ("nonface_patch_n", C.c_longlong),
                ("cart_gothrough_n", C.c_longlong), ("stage_done_n", C.c_longlong * 16),
                ("average_cart_n", C.c_double), ("gpu_ms", C.c_double), ("scan_ms", C.c_double),
                ("host_ms", C.c_double), ("scan_cart_n", C.c_longlong), ("scan_patch_n", C.c_longlong),
                ("scan_launches", C.c_int), ("handoff_n", C.c_longlong), ("cart_total_n", C.c_longlong),
                ("call_ms", C.c_double), ("dense_passes", C.c_int), ("scan_lds_ms", C.c_double),
                ("scan_lds_cart_n", C.c_longlong), ("scan_fallbacks", C.c_int),
                ("ws_regrows", C.c_int)]

    def asdict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_ if k != "stage_done_n"}
        d["stage_done_n"] = list(self.stage_done_n)
        return d


class jdaDetectOptions(C.Structure):
    _fields_ = [("dialect", C.c_int), ("nms", C.c_int), ("nms_overlap", C.c_float), ("cpp_step", C.c_int),
                ("hip_stream", C.c_void_p), ("stats", C.POINTER(jdaStats))]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("jda_amd: %s is missing -- build it with `python -m jda_amd.build` "
                          "(there is no pure-Python or CPU fallback)" % LIB_PATH)
    # If torch is going to share device pointers with us it must be the one to
    # load the HIP runtime first (both resolve to the same libamdhip64.so.7).
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(LIB_PATH)
    u8p = C.POINTER(C.c_ubyte)
    lib.jdaCascadorCreateDouble.restype = C.c_void_p
    lib.jdaCascadorCreateDouble.argtypes = [C.c_char_p]
    lib.jdaCascadorCreateFloat.restype = C.c_void_p
    lib.jdaCascadorCreateFloat.argtypes = [C.c_char_p]
    lib.jdaCascadorCreate.restype = C.c_void_p
    lib.jdaCascadorCreate.argtypes = [C.c_char_p]
    lib.jdaDebugPlanTiles.restype = C.c_int
    lib.jdaDebugPlanTiles.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int, C.POINTER(C.c_int), C.c_int]
    lib.jdaCascadorSerializeTo.restype = None
    lib.jdaCascadorSerializeTo.argtypes = [C.c_void_p, C.c_char_p]
    lib.jdaCascadorRelease.restype = None
    lib.jdaCascadorRelease.argtypes = [C.c_void_p]
    lib.jdaDetect.restype = jdaResult
    lib.jdaDetect.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float]
    lib.jdaResultRelease.restype = None
    lib.jdaResultRelease.argtypes = [jdaResult]
    lib.jdaGetLastError.restype = C.c_char_p
    lib.jdaCascadorInfo.argtypes = [C.c_void_p, C.POINTER(jdaModelInfo)]
    lib.jdaSetDevice.argtypes = [C.c_void_p, C.c_int]
    lib.jdaSetOption.argtypes = [C.c_void_p, C.c_char_p, C.c_longlong]
    lib.jdaGetOption.restype = C.c_longlong
    lib.jdaGetOption.argtypes = [C.c_void_p, C.c_char_p]
    lib.jdaCountWindows.argtypes = [C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                    C.POINTER(C.c_longlong), C.POINTER(C.c_int)]
    lib.jdaDetectOptionsInit.restype = None
    lib.jdaDetectOptionsInit.argtypes = [C.POINTER(jdaDetectOptions)]
    lib.jdaDetectBatch.argtypes = [C.c_void_p, C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                   C.c_int, C.c_int, C.c_float, C.POINTER(jdaDetectOptions), C.POINTER(jdaResult)]
    lib.jdaDetectBatchDevice.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_float,
                                         C.c_float, C.c_int, C.c_int, C.c_float, C.POINTER(jdaDetectOptions),
                                         C.POINTER(jdaResult)]
    if hasattr(lib, "jdaDetectBatchRagged"):        # (older builds, loaded through JDA_LIB_PATH for A/B runs, lack it)
      lib.jdaDetectBatchRagged.argtypes = [C.c_void_p, C.POINTER(u8p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                         C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.POINTER(jdaDetectOptions),
                                         C.POINTER(jdaResult)]
      lib.jdaDetectBatchRaggedDevice.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int),
                                               C.POINTER(C.c_int), C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                               C.c_float, C.POINTER(jdaDetectOptions), C.POINTER(jdaResult)]
    if hasattr(lib, "jdaDetectBatchRaggedDeviceRows"):     # (r06: the job's detections as one matrix of rows)
        lib.jdaDetectBatchRaggedRows.argtypes = [C.c_void_p, C.POINTER(u8p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                                 C.c_float, C.c_float, C.c_int, C.c_int, C.c_float, C.POINTER(jdaDetectOptions),
                                                 C.c_int, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int)]
        lib.jdaDetectBatchRaggedDeviceRows.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int),
                                                       C.POINTER(C.c_int), C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                                       C.c_float, C.POINTER(jdaDetectOptions), C.c_int,
                                                       C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int)]
        lib.jdaRowsRelease.restype = None
        lib.jdaRowsRelease.argtypes = [C.POINTER(C.c_float)]
    if hasattr(lib, "jdaDetectBatchCppRaggedDeviceRows"):
        dpp = C.POINTER(C.POINTER(C.c_double))
        lib.jdaDetectBatchCppRaggedRows.argtypes = [C.c_void_p, C.POINTER(u8p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                                    C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(jdaStats),
                                                    C.c_int, dpp, C.POINTER(C.c_int)]
        lib.jdaDetectBatchCppRaggedDeviceRows.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int),
                                                          C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                                          C.c_int, C.POINTER(jdaStats), C.c_int, dpp, C.POINTER(C.c_int)]
        lib.jdaRowsDRelease.restype = None
        lib.jdaRowsDRelease.argtypes = [C.POINTER(C.c_double)]
    lib.jdaDetectBatchSubmit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_float,
                                         C.c_float, C.c_int, C.c_int, C.c_float, C.POINTER(jdaDetectOptions)]
    lib.jdaDetectBatchSubmitHost.argtypes = [C.c_void_p, C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_float,
                                             C.c_float, C.c_int, C.c_int, C.c_float, C.POINTER(jdaDetectOptions)]
    lib.jdaDetectBatchWait.argtypes = [C.c_void_p, C.c_int, C.POINTER(jdaStats), C.POINTER(jdaResult)]
    lib.jdaTraceBatch.argtypes = [C.c_void_p, C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                  C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_uint), C.POINTER(C.c_float)]
    lib.jdaBuildPyramid.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, u8p, C.POINTER(C.c_int), C.POINTER(C.c_int),
                                    u8p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.jdaResultDRelease.restype = None
    lib.jdaResultDRelease.argtypes = [jdaResultD]
    lib.jdaDetectBatchCpp.argtypes = [C.c_void_p, C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_double, C.c_double, C.c_int, C.POINTER(jdaStats), C.POINTER(jdaResultD)]
    if hasattr(lib, "jdaDetectBatchCppDevice"):     # (r06; older builds loaded through JDA_LIB_PATH lack them)
        lib.jdaDetectBatchCppDevice.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                C.c_double, C.c_double, C.c_int, C.POINTER(jdaStats), C.POINTER(jdaResultD)]
        lib.jdaDetectBatchCppRagged.argtypes = [C.c_void_p, C.POINTER(u8p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                                C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, C.POINTER(jdaStats),
                                                C.POINTER(jdaResultD)]
        lib.jdaDetectBatchCppRaggedDevice.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_int),
                                                      C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                                      C.c_int, C.POINTER(jdaStats), C.POINTER(jdaResultD)]
        lib.jdaResultsDRelease.restype = None
        lib.jdaResultsDRelease.argtypes = [C.POINTER(jdaResultD), C.c_int]
        lib.jdaResultsDPack.argtypes = [C.POINTER(jdaResultD), C.c_int, C.c_int, C.POINTER(C.c_double), C.c_int]
    lib.jdaDetectBatchCppPyramid.argtypes = [C.c_void_p, C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_double, C.c_double, C.c_int, C.POINTER(jdaStats), C.POINTER(jdaResultD)]
    if hasattr(lib, "jdaDetectBatchCppPyramidMS"):
        lib.jdaDetectBatchCppPyramidMS.argtypes = [C.c_void_p, C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                                   C.c_double, C.c_double, C.c_int, C.POINTER(jdaStats), C.POINTER(jdaResultD)]
    lib.jdaResizeCv.argtypes = [C.c_void_p, u8p, C.c_int, C.c_int, u8p, C.c_int, C.c_int]
    lib.jdaSetSimilarityTransform.argtypes = [C.c_void_p, C.c_int]
    lib.jdaNmsC.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_float), C.c_int, C.c_float, u8p]
    lib.jdaNmsCpp.argtypes = [C.POINTER(C.c_int), C.POINTER(C.c_double), C.c_int, C.c_double, C.POINTER(C.c_int)]
    lib.jdaResultsRelease.restype = None
    lib.jdaResultsRelease.argtypes = [C.POINTER(jdaResult), C.c_int]
    lib.jdaResultsPack.argtypes = [C.POINTER(jdaResult), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_int]
    lib.jdaModelStreamBytes.restype = C.c_longlong
    lib.jdaModelStreamBytes.argtypes = [C.c_int] * 5
    if hasattr(lib, "jdaTraceBatchCpp"):
        lib.jdaTraceBatchCpp.argtypes = [C.c_void_p, C.POINTER(u8p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                         C.POINTER(C.c_uint), C.POINTER(C.c_double)]
    return lib


lib = _load()

# reference-named entry points, callable exactly like the C functions
jdaCascadorCreateDouble = lib.jdaCascadorCreateDouble
jdaCascadorCreateFloat = lib.jdaCascadorCreateFloat
jdaCascadorCreate = lib.jdaCascadorCreate
jdaCascadorSerializeTo = lib.jdaCascadorSerializeTo
jdaCascadorRelease = lib.jdaCascadorRelease
jdaDetect = lib.jdaDetect
jdaResultRelease = lib.jdaResultRelease


def last_error():
    return (lib.jdaGetLastError() or b"").decode()


def count_windows(width, height, scale=1.25, min_size=40, max_size=-1):
    n, nl = C.c_longlong(), C.c_int()
    if lib.jdaCountWindows(width, height, scale, min_size, max_size, C.byref(n), C.byref(nl)) != 0:
        raise JdaError(last_error())
    return n.value, nl.value


def nms_c(bboxes, scores, overlap=0.3):
    """Host NMS of dialect C (reference c/jda.c:237-316): boolean keep mask in scan order."""
    bboxes = np.ascontiguousarray(bboxes, np.int32).reshape(-1, 3)
    scores = np.ascontiguousarray(scores, np.float32)
    keep = np.zeros(len(scores), np.uint8)
    if lib.jdaNmsC(bboxes.ctypes.data_as(C.POINTER(C.c_int)), scores.ctypes.data_as(C.POINTER(C.c_float)),
                   len(scores), overlap, keep.ctypes.data_as(C.POINTER(C.c_ubyte))) < 0:
        raise JdaError("jdaNmsC failed")
    return keep.astype(bool)


def nms_cpp(rects, scores, overlap=0.3):
    """Host NMS of dialect CPP (reference cascador.cpp:387-429): picked indices, best first."""
    rects = np.ascontiguousarray(rects, np.int32).reshape(-1, 4)
    scores = np.ascontiguousarray(scores, np.float64)
    picked = np.zeros(max(len(scores), 1), np.int32)
    n = lib.jdaNmsCpp(rects.ctypes.data_as(C.POINTER(C.c_int)), scores.ctypes.data_as(C.POINTER(C.c_double)),
                      len(scores), overlap, picked.ctypes.data_as(C.POINTER(C.c_int)))
    if n < 0:
        raise JdaError("jdaNmsCpp failed")
    return picked[:n].copy()


def _frame_ptrs(frames):
    """unsigned char*[n] for the frames of a contiguous [n,h,w] array.  From integer addresses: a ctypes cast per frame
    costs more than a millisecond per 256-frame batch, which is the order of the whole GPU pass."""
    n = frames.shape[0]
    base, stride = frames.ctypes.data, frames.strides[0]
    arr = (C.c_void_p * max(n, 1))(*[base + i * stride for i in range(n)])
    return C.cast(arr, C.POINTER(C.POINTER(C.c_ubyte)))


def _image_ptrs(images):
    arr = (C.c_void_p * max(len(images), 1))(*[im.ctypes.data for im in images])
    return C.cast(arr, C.POINTER(C.POINTER(C.c_ubyte)))


def _u8(a):
    return a.ctypes.data_as(C.POINTER(C.c_ubyte))


def _ivec(v, ctype=C.c_int, dtype=np.int32):
    """A sequence of integers as a C array argument without a Python loop (a job of thousands of images passes three of
    them per call).  Returns (keep-alive array, pointer)."""
    a = np.ascontiguousarray(v, dtype)
    if a.size == 0:
        a = np.zeros(1, dtype)
    return a, a.ctypes.data_as(C.POINTER(ctype))


class _RowsOwner:
    """Keeps the library's malloc'd row matrix until the last numpy view of it is gone, then gives it back
    (jdaRowsRelease / jdaRowsDRelease)."""
    def __init__(self, ptr, release):
        self._ptr, self._release = ptr, release

    def __del__(self):
        if self._ptr:
            self._release(self._ptr)
            self._ptr = None


def _owned_rows(ptr, n_rows, width, release, dtype):
    """The library's rows as a numpy array WITHOUT a copy (15 MB per dialect-CPP FDDB job): the array's base holds a
    _RowsOwner.  (numpy arrays are not garbage-collector tracked: the owner must not point back at the array.)"""
    owner = _RowsOwner(ptr, release)
    if n_rows <= 0:
        return np.empty((0, width), dtype)

    class _Holder:
        pass
    h = _Holder()
    h.__array_interface__ = np.ctypeslib.as_array(ptr, (n_rows, width)).__array_interface__
    h._owner = owner
    return np.asarray(h)


def _take(r):
    """jdaResult -> dict of numpy copies, then release the C arrays."""
    n, dim = r.n, 2 * r.landmark_n
    out = dict(
        bboxes=np.ctypeslib.as_array(r.bboxes, (n, 3)).copy() if n else np.zeros((0, 3), np.int32),
        scores=np.ctypeslib.as_array(r.scores, (n,)).copy() if n else np.zeros(0, np.float32),
        shapes=np.ctypeslib.as_array(r.shapes, (n, dim)).copy() if n else np.zeros((0, dim), np.float32))
    lib.jdaResultRelease(r)
    return out


def _take_d(r):
    n, dim = r.n, 2 * r.landmark_n
    out = dict(
        rects=np.ctypeslib.as_array(r.rects, (n, 4)).copy() if n else np.zeros((0, 4), np.int32),
        scores=np.ctypeslib.as_array(r.scores, (n,)).copy() if n else np.zeros(0, np.float64),
        shapes=np.ctypeslib.as_array(r.shapes, (n, dim)).copy() if n else np.zeros((0, dim), np.float64))
    lib.jdaResultDRelease(r)
    return out


class Cascador:
    """Owning handle around the opaque void* of the C API."""

    def __init__(self, model_path, real="auto", device=None):
        p = os.fsencode(model_path)
        if real == "double":
            self.h = lib.jdaCascadorCreateDouble(p)
        elif real == "float":
            self.h = lib.jdaCascadorCreateFloat(p)
        else:
            self.h = lib.jdaCascadorCreate(p)
        if not self.h:
            raise JdaError("cannot load model %s: %s" % (model_path, last_error()))
        info = jdaModelInfo()
        lib.jdaCascadorInfo(self.h, C.byref(info))
        self.T, self.K, self.L, self.D = info.T, info.K, info.landmark_n, info.tree_depth
        self.multi_scale = bool(info.multi_scale)
        self.source_real_bytes = info.source_real_bytes
        self.dim = 2 * self.L
        if device is not None and lib.jdaSetDevice(self.h, int(device)) != 0:
            raise JdaError(last_error())

    def close(self):
        if getattr(self, "h", None):
            lib.jdaCascadorRelease(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_similarity_transform(self, on):
        """Dialect CPP: Config face.similarity_transform (reference common.cpp:214)."""
        if lib.jdaSetSimilarityTransform(self.h, 1 if on else 0) != 0:
            raise JdaError("jdaSetSimilarityTransform failed")

    def set_option(self, key, value):
        """jdaSetOption: tuning knobs of this cascador (include/jda.h); never changes results."""
        if lib.jdaSetOption(self.h, key.encode(), int(value)) != 0:
            raise JdaError(last_error())

    def get_option(self, key):
        return int(lib.jdaGetOption(self.h, key.encode()))

    def serialize(self, path):
        lib.jdaCascadorSerializeTo(self.h, os.fsencode(path))

    # -- single frame, exactly the reference call -----------------------------
    def detect(self, img, scale=1.25, step=0.1, min_size=40, max_size=-1, th=-0.5):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        r = lib.jdaDetect(self.h, _u8(img), w, h, scale, step, min_size, max_size, th)
        err = last_error()
        if r.n == 0 and err:
            lib.jdaResultRelease(r)
            raise JdaError(err)
        return _take(r)

    def _opts(self, nms, stats):
        o = jdaDetectOptions()
        lib.jdaDetectOptionsInit(C.byref(o))
        o.nms = 1 if nms else 0
        st = jdaStats() if stats else None
        if st is not None:
            o.stats = C.pointer(st)
        return o, st

    # -- batch of host frames ---------------------------------------------------
    def detect_batch(self, frames, scale=1.25, min_size=40, max_size=-1, th=-0.5, nms=True, stats=False):
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w = frames.shape
        ptrs = _frame_ptrs(frames)
        res = (jdaResult * max(n, 1))()
        o, st = self._opts(nms, stats)
        rc = lib.jdaDetectBatch(self.h, ptrs, n, w, h, scale, 0.1, min_size, max_size, th, C.byref(o), res)
        if rc != 0:
            raise JdaError(last_error())
        out = [_take(res[i]) for i in range(n)]
        return (out, st.asdict()) if stats else out

    # -- batch resident in device memory (torch uint8 CUDA tensor [n,h,w]) ------------
    def detect_batch_device(self, d_frames, scale=1.25, min_size=40, max_size=-1, th=-0.5, nms=True,
                            stats=False, keep_results=True, frame_offset=0, hip_stream=None):
        assert d_frames.is_cuda and d_frames.dtype.itemsize == 1 and d_frames.is_contiguous()
        n, h, w = d_frames.shape
        res = (jdaResult * max(n, 1))()
        o, st = self._opts(nms, stats)
        if hip_stream is not None:      # a hipStream_t handle (e.g. torch.cuda.Stream().cuda_stream): work is ordered on it
            o.hip_stream = C.c_void_p(hip_stream)
        rc = lib.jdaDetectBatchDevice(self.h, C.c_void_p(d_frames.data_ptr()), h * w, n, w, h, scale, 0.1,
                                      min_size, max_size, th, C.byref(o), res)
        if rc != 0:
            raise JdaError(last_error())
        if keep_results == "packed":
            # one C call: rows [frame, x, y, size, score, shape...] of every detection of the batch
            rows = lib.jdaResultsPack(res, n, frame_offset, None, 0)
            out = np.empty((max(rows, 0), 5 + self.dim), np.float32)
            if rows > 0:
                lib.jdaResultsPack(res, n, frame_offset, out.ctypes.data_as(C.POINTER(C.c_float)), rows)
            lib.jdaResultsRelease(res, n)
        elif keep_results:
            out = [_take(res[i]) for i in range(n)]
        else:
            out = [res[i].n for i in range(n)]
            lib.jdaResultsRelease(res, n)
        return (out, st.asdict()) if stats else out

    # -- images of different sizes in one job --------------------------------------------
    def _collect(self, res, n, keep_results, frame_offset=0):
        if keep_results == "packed":
            rows = lib.jdaResultsPack(res, n, frame_offset, None, 0)
            out = np.empty((max(rows, 0), 5 + self.dim), np.float32)
            if rows > 0:
                lib.jdaResultsPack(res, n, frame_offset, out.ctypes.data_as(C.POINTER(C.c_float)), rows)
            lib.jdaResultsRelease(res, n)
        elif keep_results:
            out = [_take(res[i]) for i in range(n)]
        else:
            out = [res[i].n for i in range(n)]
            lib.jdaResultsRelease(res, n)
        return out

    def detect_ragged(self, images, scale=1.25, min_size=40, max_size=-1, th=-0.5, nms=True, stats=False,
                      keep_results=True):
        """jdaDetectBatchRagged: a list of uint8 [h, w] arrays of different sizes in host memory."""
        images = [np.ascontiguousarray(im, np.uint8) for im in images]
        n = len(images)
        ptrs = _image_ptrs(images)
        ws = (C.c_int * max(n, 1))(*[im.shape[1] for im in images])
        hs = (C.c_int * max(n, 1))(*[im.shape[0] for im in images])
        res = (jdaResult * max(n, 1))()
        o, st = self._opts(nms, stats)
        rc = lib.jdaDetectBatchRagged(self.h, ptrs, ws, hs, n, scale, 0.1, min_size, max_size, th, C.byref(o), res)
        if rc != 0:
            raise JdaError(last_error())
        out = self._collect(res, n, keep_results)
        return (out, st.asdict()) if stats else out

    def detect_ragged_packed(self, buf, offsets, widths, heights, scale=1.25, min_size=40, max_size=-1, th=-0.5,
                             nms=True, stats=False, keep_results=True, frame_offset=0):
        """The same for images packed in ONE buffer (image i = buf[offsets[i] : offsets[i] + w*h], rows back to
        back): a numpy uint8 array (host entry, jdaDetectBatchRagged) or a torch uint8 CUDA tensor
        (jdaDetectBatchRaggedDevice)."""
        n = len(offsets)
        _kw, ws = _ivec(widths)
        _kh, hs = _ivec(heights)
        o, st = self._opts(nms, stats)
        host = isinstance(buf, np.ndarray)
        if host:
            assert buf.dtype == np.uint8 and buf.flags.c_contiguous
            _ko, ptrs = _ivec(np.asarray(offsets, np.uint64) + np.uint64(buf.ctypes.data), C.POINTER(C.c_ubyte), np.uint64)
        else:
            assert buf.is_cuda and buf.dtype.itemsize == 1 and buf.is_contiguous()
            _ko, offs = _ivec(offsets, C.c_size_t, np.uint64)
        if keep_results == "packed" and hasattr(lib, "jdaDetectBatchRaggedDeviceRows"):
            # rows straight from the library (jdaDetectBatchRagged[Device]Rows): no jdaResult per image in between
            rp, nr = C.POINTER(C.c_float)(), C.c_int(0)
            if host:
                rc = lib.jdaDetectBatchRaggedRows(self.h, ptrs, ws, hs, n, scale, 0.1, min_size, max_size, th, C.byref(o),
                                                  frame_offset, C.byref(rp), C.byref(nr))
            else:
                rc = lib.jdaDetectBatchRaggedDeviceRows(self.h, C.c_void_p(buf.data_ptr()), offs, ws, hs, n, scale, 0.1,
                                                        min_size, max_size, th, C.byref(o), frame_offset, C.byref(rp), C.byref(nr))
            if rc != 0:
                raise JdaError(last_error())
            out = _owned_rows(rp, nr.value, 5 + self.dim, lib.jdaRowsRelease, np.float32)
            return (out, st.asdict()) if stats else out
        res = (jdaResult * max(n, 1))()
        if host:
            rc = lib.jdaDetectBatchRagged(self.h, ptrs, ws, hs, n, scale, 0.1, min_size, max_size, th, C.byref(o), res)
        else:
            rc = lib.jdaDetectBatchRaggedDevice(self.h, C.c_void_p(buf.data_ptr()), offs, ws, hs, n, scale, 0.1,
                                                min_size, max_size, th, C.byref(o), res)
        if rc != 0:
            raise JdaError(last_error())
        out = self._collect(res, n, keep_results, frame_offset)
        return (out, st.asdict()) if stats else out

    # -- two batches in flight from one thread -----------------------------------
    def submit_batch_device(self, d_frames, scale=1.25, min_size=40, max_size=-1, th=-0.5, nms=True, stats=False):
        """Queues the scan of a batch and returns a ticket; collect it with wait_batch (the frames are kept
        alive until then).  Submit batch i+1 before waiting for batch i to overlap host and device work.
        stats=True: the pass is bracketed with timing events, so wait_batch(stats=True) reports gpu_ms / scan_ms
        (the counters are reported either way)."""
        assert d_frames.is_cuda and d_frames.dtype.itemsize == 1 and d_frames.is_contiguous()
        n, h, w = d_frames.shape
        o, _ = self._opts(nms, stats)
        t = lib.jdaDetectBatchSubmit(self.h, C.c_void_p(d_frames.data_ptr()), h * w, n, w, h, scale, 0.1,
                                     min_size, max_size, th, C.byref(o))
        if t < 0:
            raise JdaError(last_error())
        if not hasattr(self, "_pending"):
            self._pending = {}
        self._pending[t] = (d_frames, n)
        return t

    def submit_batch_host(self, frames, scale=1.25, min_size=40, max_size=-1, th=-0.5, nms=True, stats=False):
        """Submit for frames in host memory (numpy uint8 [n,h,w]; kept alive until wait_batch)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        n, h, w = frames.shape
        ptrs = _frame_ptrs(frames)
        o, _ = self._opts(nms, stats)
        t = lib.jdaDetectBatchSubmitHost(self.h, ptrs, n, w, h, scale, 0.1, min_size, max_size, th, C.byref(o))
        if t < 0:
            raise JdaError(last_error())
        if not hasattr(self, "_pending"):
            self._pending = {}
        self._pending[t] = ((frames, ptrs), n)
        return t

    def wait_batch(self, ticket, stats=False, keep_results=True, frame_offset=0):
        d_frames, n = self._pending.pop(ticket)
        res = (jdaResult * max(n, 1))()
        st = jdaStats() if stats else None
        rc = lib.jdaDetectBatchWait(self.h, ticket, C.byref(st) if stats else None, res)
        if rc != 0:
            raise JdaError(last_error())
        if keep_results == "packed":
            rows = lib.jdaResultsPack(res, n, frame_offset, None, 0)
            out = np.empty((max(rows, 0), 5 + self.dim), np.float32)
            if rows > 0:
                lib.jdaResultsPack(res, n, frame_offset, out.ctypes.data_as(C.POINTER(C.c_float)), rows)
            lib.jdaResultsRelease(res, n)
        elif keep_results:
            out = [_take(res[i]) for i in range(n)]
        else:
            out = [res[i].n for i in range(n)]
            lib.jdaResultsRelease(res, n)
        return (out, st.asdict()) if stats else out

    def plan_tiles(self, width, height, scale=1.25, min_size=40, max_size=-1):
        """How k_scan would tile each pyramid level of a dialect-C call (no GPU needed)."""
        out = np.zeros((64, 10), np.int32)
        n = lib.jdaDebugPlanTiles(self.h, width, height, scale, min_size, max_size,
                                  out.ctypes.data_as(C.POINTER(C.c_int)), 64)
        if n < 0:
            raise JdaError(last_error())
        keys = ("win", "step", "nx", "ny", "mode", "tw", "th", "pitch", "tiles_x", "tiles_y")
        return [dict(zip(keys, (int(v) for v in row))) for row in out[:n]]

    # -- parity instrumentation ---------------------------------------------------
    def trace(self, frames, scale=1.25, min_size=40, max_size=-1):
        frames = np.ascontiguousarray(frames, np.uint8)
        if frames.ndim == 2:
            frames = frames[None]
        n, h, w = frames.shape
        wpf, _ = count_windows(w, h, scale, min_size, max_size)
        tot = n * wpf
        carts = np.zeros(tot, np.int32)
        score = np.zeros(tot, np.float32)
        hsh = np.zeros(tot, np.uint32)
        shapes = np.zeros((tot, self.dim), np.float32)
        ptrs = _frame_ptrs(frames)
        rc = lib.jdaTraceBatch(self.h, ptrs, n, w, h, scale, min_size, max_size,
                               carts.ctypes.data_as(C.POINTER(C.c_int)), score.ctypes.data_as(C.POINTER(C.c_float)),
                               hsh.ctypes.data_as(C.POINTER(C.c_uint)), shapes.ctypes.data_as(C.POINTER(C.c_float)))
        if rc != 0:
            raise JdaError(last_error())
        return dict(carts_n=carts, score=score, path_hash=hsh, shapes=shapes)

    def build_pyramid(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        d = [C.c_int() for _ in range(4)]
        if lib.jdaBuildPyramid(self.h, _u8(img), w, h, None, C.byref(d[0]), C.byref(d[1]), None,
                               C.byref(d[2]), C.byref(d[3])) != 0:
            raise JdaError(last_error())
        hw, hh, qw, qh = [x.value for x in d]
        half = np.zeros((max(hh, 0), max(hw, 0)), np.uint8)
        quarter = np.zeros((max(qh, 0), max(qw, 0)), np.uint8)
        if lib.jdaBuildPyramid(self.h, _u8(img), w, h, _u8(half), C.byref(d[0]), C.byref(d[1]), _u8(quarter),
                               C.byref(d[2]), C.byref(d[3])) != 0:
            raise JdaError(last_error())
        return half, quarter

    # -- dialect CPP ----------------------------------------------------------------
    def detect_batch_cpp(self, frames, minimum_size=20, step=5, factor=1.2, overlap=0.3, nms=True, stats=False):
        frames = np.ascontiguousarray(frames, np.uint8)
        if frames.ndim == 2:
            frames = frames[None]
        n, h, w = frames.shape
        ptrs = _frame_ptrs(frames)
        res = (jdaResultD * max(n, 1))()
        st = jdaStats()
        rc = lib.jdaDetectBatchCpp(self.h, ptrs, n, w, h, minimum_size, step, factor, overlap, 1 if nms else 0,
                                   C.byref(st), res)
        if rc != 0:
            raise JdaError(last_error())
        out = [_take_d(res[i]) for i in range(n)]
        return (out, st.asdict()) if stats else out

    def _collect_d(self, res, n, keep_results, frame_offset=0):
        if keep_results == "packed":
            # one C call: rows [frame, x, y, w, h, score, shape...] (float64) of every detection of the batch
            rows = lib.jdaResultsDPack(res, n, frame_offset, None, 0)
            out = np.empty((max(rows, 0), 6 + self.dim), np.float64)
            if rows > 0:
                lib.jdaResultsDPack(res, n, frame_offset, out.ctypes.data_as(C.POINTER(C.c_double)), rows)
            lib.jdaResultsDRelease(res, n)
            return out
        if keep_results:
            return [_take_d(res[i]) for i in range(n)]
        out = [res[i].n for i in range(n)]
        lib.jdaResultsDRelease(res, n)
        return out

    def detect_batch_cpp_device(self, d_frames, minimum_size=20, step=5, factor=1.2, overlap=0.3, nms=True, stats=False,
                                keep_results=True, frame_offset=0):
        """jdaDetectBatchCppDevice: a torch uint8 CUDA tensor [n, h, w] resident in HBM."""
        assert d_frames.is_cuda and d_frames.dtype.itemsize == 1 and d_frames.is_contiguous()
        n, h, w = d_frames.shape
        res = (jdaResultD * max(n, 1))()
        st = jdaStats()
        rc = lib.jdaDetectBatchCppDevice(self.h, C.c_void_p(d_frames.data_ptr()), h * w, n, w, h, minimum_size, step, factor,
                                         overlap, 1 if nms else 0, C.byref(st) if stats else None, res)
        if rc != 0:
            raise JdaError(last_error())
        out = self._collect_d(res, n, keep_results, frame_offset)
        return (out, st.asdict()) if stats else out

    def detect_ragged_cpp(self, images, minimum_size=20, step=5, factor=1.2, overlap=0.3, nms=True, stats=False,
                          keep_results=True):
        """jdaDetectBatchCppRagged: a list of uint8 [h, w] arrays of different sizes in host memory -- the reference's
        `jda fddb` loop (one joincascador.Detect per image, src/test.cpp:142) as one job."""
        images = [np.ascontiguousarray(im, np.uint8) for im in images]
        n = len(images)
        ptrs = _image_ptrs(images)
        ws = (C.c_int * max(n, 1))(*[im.shape[1] for im in images])
        hs = (C.c_int * max(n, 1))(*[im.shape[0] for im in images])
        res = (jdaResultD * max(n, 1))()
        st = jdaStats()
        rc = lib.jdaDetectBatchCppRagged(self.h, ptrs, ws, hs, n, minimum_size, step, factor, overlap, 1 if nms else 0,
                                         C.byref(st) if stats else None, res)
        if rc != 0:
            raise JdaError(last_error())
        out = self._collect_d(res, n, keep_results)
        return (out, st.asdict()) if stats else out

    def detect_ragged_cpp_packed(self, buf, offsets, widths, heights, minimum_size=20, step=5, factor=1.2, overlap=0.3,
                                 nms=True, stats=False, keep_results=True, frame_offset=0):
        """The same for images packed in ONE buffer (image i = buf[offsets[i] : offsets[i] + w*h]): a numpy uint8 array
        (jdaDetectBatchCppRagged) or a torch uint8 CUDA tensor (jdaDetectBatchCppRaggedDevice)."""
        n = len(offsets)
        _kw, ws = _ivec(widths)
        _kh, hs = _ivec(heights)
        st = jdaStats()
        sp = C.byref(st) if stats else None
        host = isinstance(buf, np.ndarray)
        if host:
            assert buf.dtype == np.uint8 and buf.flags.c_contiguous
            _ko, ptrs = _ivec(np.asarray(offsets, np.uint64) + np.uint64(buf.ctypes.data), C.POINTER(C.c_ubyte), np.uint64)
        else:
            assert buf.is_cuda and buf.dtype.itemsize == 1 and buf.is_contiguous()
            _ko, offs = _ivec(offsets, C.c_size_t, np.uint64)
        if keep_results == "packed" and hasattr(lib, "jdaDetectBatchCppRaggedDeviceRows"):
            # rows straight from the library: no jdaResultD per image, no second copy of 15 MB of rows
            rp, nr = C.POINTER(C.c_double)(), C.c_int(0)
            if host:
                rc = lib.jdaDetectBatchCppRaggedRows(self.h, ptrs, ws, hs, n, minimum_size, step, factor, overlap, 1 if nms else 0, sp,
                                                     frame_offset, C.byref(rp), C.byref(nr))
            else:
                rc = lib.jdaDetectBatchCppRaggedDeviceRows(self.h, C.c_void_p(buf.data_ptr()), offs, ws, hs, n, minimum_size, step,
                                                           factor, overlap, 1 if nms else 0, sp, frame_offset, C.byref(rp), C.byref(nr))
            if rc != 0:
                raise JdaError(last_error())
            out = _owned_rows(rp, nr.value, 6 + self.dim, lib.jdaRowsDRelease, np.float64)
            return (out, st.asdict()) if stats else out
        res = (jdaResultD * max(n, 1))()
        if host:
            rc = lib.jdaDetectBatchCppRagged(self.h, ptrs, ws, hs, n, minimum_size, step, factor, overlap, 1 if nms else 0, sp, res)
        else:
            rc = lib.jdaDetectBatchCppRaggedDevice(self.h, C.c_void_p(buf.data_ptr()), offs, ws, hs, n, minimum_size, step,
                                                   factor, overlap, 1 if nms else 0, sp, res)
        if rc != 0:
            raise JdaError(last_error())
        out = self._collect_d(res, n, keep_results, frame_offset)
        return (out, st.asdict()) if stats else out

    def detect_batch_cpp_pyramid(self, frames, origin_size=48, step=5, factor=1.2, overlap=0.3, nms=True, stats=False,
                                 half_size=0, quarter_size=0):
        """Dialect CPP, detect method 0: the true image pyramid (reference cascador.cpp:216-308).  half_size /
        quarter_size > 0: jdaDetectBatchCppPyramidMS, the per-window patches of a multi-scale model."""
        frames = np.ascontiguousarray(frames, np.uint8)
        if frames.ndim == 2:
            frames = frames[None]
        n, h, w = frames.shape
        ptrs = _frame_ptrs(frames)
        res = (jdaResultD * max(n, 1))()
        st = jdaStats()
        if half_size or quarter_size:
            rc = lib.jdaDetectBatchCppPyramidMS(self.h, ptrs, n, w, h, origin_size, half_size, quarter_size, step, factor, overlap,
                                                1 if nms else 0, C.byref(st), res)
        else:
            rc = lib.jdaDetectBatchCppPyramid(self.h, ptrs, n, w, h, origin_size, step, factor, overlap, 1 if nms else 0,
                                              C.byref(st), res)
        if rc != 0:
            raise JdaError(last_error())
        out = [_take_d(res[i]) for i in range(n)]
        return (out, st.asdict()) if stats else out

    def resize_cv(self, img, out_width, out_height):
        img = np.ascontiguousarray(img, np.uint8)
        h, w = img.shape
        out = np.zeros((out_height, out_width), np.uint8)
        if lib.jdaResizeCv(self.h, _u8(img), w, h, _u8(out), out_width, out_height) != 0:
            raise JdaError(last_error())
        return out

    def trace_cpp(self, frames, minimum_size=20, step=5, factor=1.2):
        frames = np.ascontiguousarray(frames, np.uint8)
        if frames.ndim == 2:
            frames = frames[None]
        n, h, w = frames.shape
        from . import synth
        wpf = synth.count_windows_cpp(w, h, minimum_size, step, factor)
        tot = n * wpf
        carts = np.zeros(tot, np.int32)
        score = np.zeros(tot, np.float64)
        hsh = np.zeros(tot, np.uint32)
        shapes = np.zeros((tot, self.dim), np.float64)
        ptrs = _frame_ptrs(frames)
        rc = lib.jdaTraceBatchCpp(self.h, ptrs, n, w, h, minimum_size, step, factor,
                                  carts.ctypes.data_as(C.POINTER(C.c_int)), score.ctypes.data_as(C.POINTER(C.c_double)),
                                  hsh.ctypes.data_as(C.POINTER(C.c_uint)), shapes.ctypes.data_as(C.POINTER(C.c_double)))
        if rc != 0:
            raise JdaError(last_error())
        return dict(carts_n=carts, score=score, path_hash=hsh, shapes=shapes)
