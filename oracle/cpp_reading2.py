"""TEST INFRASTRUCTURE, not product: a SECOND, independent restatement of the reference's fp64 detect path (dialect CPP),
written from the reference's C++ alone -- not from oracle/jda_oracle.c -- in plain Python on IEEE doubles, for small cases.

Why: src/jda cannot be compiled in this image (OpenCV, jsmnpp and liblinear are absent), so dialect CPP is PARITY UNPINNED:
the HIP kernels are bit-exact against jda_oracle.c's restatement only.  Two restatements made independently of each other
from the same source that agree bit for bit (tests/test_cpp_second_reading.py) narrow the room for a misreading; they do
not pin anything -- a detail both readers got wrong the same way, or one that only the real OpenCV decides (cv::resize for
multi-scale models, cv::norm inside the similarity transform), stays open.  Covered here: detect methods 1 and 0, single- and
multi-scale models, the similarity transform off and on.  cv::resize is NOT restated a second time: where the reference resizes
(the half / quarter images of method 1, the pyramid levels and per-window patches of method 0) the caller hands in a resize
function -- the tests pass jda_oracle.c's restatement -- so these paths check everything AROUND the resize: which image is
resized to what, the ROI arithmetic, which patch a feature reads and with which width and height, rect scaling, NMS, relocation.

What it follows, line by line:
  model file       JoinCascador::SerializeFrom  src/jda/cascador.cpp:127-164, Cart::SerializeFrom src/jda/cart.cpp:404-423
  Validate         src/jda/cascador.cpp:166-211   (shape = mean_shape: RandomShape with shift_size 0, src/jda/data.cpp:225-236,
                                                   src/test.cpp:17,75)
  Cart::Forward    src/jda/cart.cpp:392-404
  feature value    Feature::CalcFeatureValue src/jda/data.cpp:18-58, checkBoundaryOfImage include/jda/common.hpp:227-232
  STParameter      identity when face.similarity_transform is off, src/jda/data.cpp:64-70; Apply include/jda/data.hpp:42-45
  GenDeltaShape    src/jda/btcart.cpp:407-424
  detectMultiScale1 src/jda/cascador.cpp:310-376;  detectSingleScale / detectMultiScale (method 0) src/jda/cascador.cpp:215-308
  nms              src/jda/cascador.cpp:387-429 (std::multimap: ascending keys, equal keys in insertion order)
  Detect           src/jda/cascador.cpp:431-477
"""
import bisect
import math
import struct


def c_round(x):
    """C round(): to nearest, halves away from zero (data.cpp:45-48 calls ::round on doubles)."""
    a = abs(x)
    f = math.floor(a)
    if a - f >= 0.5:          # (exact: a - f is representable)
        f += 1.0
    return -f if x < 0 else f


class Cart2:
    __slots__ = ("scale", "lm1", "lm2", "o1x", "o1y", "o2x", "o2y", "nth", "scores", "th", "mean", "std")


class Model2:
    """The trainer's double file (jdaCascadorCreateDouble's input): cascador.cpp:127-164."""

    def __init__(self, path):
        b = open(path, "rb").read()
        self.pos = 0

        def take(fmt):
            v = struct.unpack_from("<" + fmt, b, self.pos)
            self.pos += struct.calcsize("<" + fmt)
            return v
        (_yo,) = take("i")
        self.T, self.K, self.L, self.D = take("iiii")
        self.stage_idx, self.cart_idx = take("ii")
        self.mean_shape = list(take("%dd" % (2 * self.L)))
        nodes_n = 1 << self.D                       # cart.hpp: nodes_n = 2^depth, features[1 .. nodes_n/2 - 1]
        self.leaf_n = nodes_n // 2
        self.carts = []
        self.w = []
        for t in range(self.T):
            row = []
            for k in range(self.K):
                c = Cart2()
                n = nodes_n // 2
                c.scale = [0] * n; c.lm1 = [0] * n; c.lm2 = [0] * n
                c.o1x = [0.0] * n; c.o1y = [0.0] * n; c.o2x = [0.0] * n; c.o2y = [0.0] * n; c.nth = [0] * n
                for i in range(1, n):               # cart.cpp:406-416
                    c.scale[i], c.lm1[i], c.lm2[i] = take("iii")
                    c.o1x[i], c.o1y[i], c.o2x[i], c.o2y[i] = take("dddd")
                    (c.nth[i],) = take("i")
                c.scores = list(take("%dd" % n))    # cart.cpp:418-420
                c.th, c.mean, c.std = take("ddd")   # cart.cpp:421-423
                row.append(c)
            self.carts.append(row)
            rows = self.K * (1 << (self.D - 1))     # cascador.cpp:156-161
            self.w.append([list(take("%dd" % (2 * self.L))) for _ in range(rows)])
        take("i")
        assert self.pos == len(b), "trailing bytes in the model file"
        self.multi_scale = any(s != 0 for row in self.carts for c in row for s in c.scale[1:])


IDENTITY = (1.0, 1.0, 0.0, 0.0, 1.0)        # scale, rot00, rot01, rot10, rot11 (data.hpp: STParameter's default)


def _cv_norm_l2(v):
    """cv::norm(Mat_<double>) (NORM_L2), as stat.cpp's normL2Sqr<double, double> sums it: four squares at a time, then the
    remainder one by one, one square root at the end."""
    s = 0.
    i, n = 0, len(v)
    while i <= n - 4:
        v0, v1, v2, v3 = v[i], v[i + 1], v[i + 2], v[i + 3]
        s += v0 * v0 + v1 * v1 + v2 * v2 + v3 * v3
        i += 4
    while i < n:
        s += v[i] * v[i]
        i += 1
    return math.sqrt(s)


def st_calc(shape1, shape2, L):
    """STParameter::Calc (data.cpp:64-114) with face.similarity_transform on.  Two OpenCV details decide bits here and
    neither can be run in this image: cv::norm's summation order (above) and `Mat /= double`, which mat.inl.hpp spells
    a.convertTo(a, -1, 1. / s): every element times the RECIPROCAL, plus a zero shift.  (Written after model_dev.cpp's and
    jda_oracle.c's versions of the same two details had been seen; the author's memory of the OpenCV sources agrees with them.)"""
    x1c = y1c = x2c = y2c = 0.
    for i in range(L):
        x1c += shape1[2 * i]; y1c += shape1[2 * i + 1]
        x2c += shape2[2 * i]; y2c += shape2[2 * i + 1]
    x1c /= L; y1c /= L; x2c /= L; y2c /= L
    t1, t2 = [0.] * (2 * L), [0.] * (2 * L)
    for i in range(L):
        t1[2 * i] = shape1[2 * i] - x1c; t1[2 * i + 1] = shape1[2 * i + 1] - y1c
        t2[2 * i] = shape2[2 * i] - x2c; t2[2 * i + 1] = shape2[2 * i + 1] - y2c
    scale1, scale2 = _cv_norm_l2(t1), _cv_norm_l2(t2)
    scale = scale1 / scale2
    a1, a2 = 1. / scale1, 1. / scale2
    t1 = [v * a1 + 0. for v in t1]
    t2 = [v * a2 + 0. for v in t2]
    num = den = 0.
    for i in range(L):
        num += t1[2 * i + 1] * t2[2 * i] - t1[2 * i] * t2[2 * i + 1]
        den += t1[2 * i] * t2[2 * i] + t1[2 * i + 1] * t2[2 * i + 1]
    norm = math.sqrt(num * num + den * den)
    sin_t, cos_t = num / norm, den / norm
    return (scale, cos_t, -sin_t, sin_t, cos_t)


def st_apply(stp, x1, y1):
    s, r00, r01, r10, r11 = stp                 # data.hpp:42-45
    return s * (r00 * x1 + r01 * y1), s * (r10 * x1 + r11 * y1)


def feature_value(c, i, patches, shape, stp):
    """Feature::CalcFeatureValue.  patches[scale] = (img, x0, y0, width, height): the cv::Mat ROI of the ORIGIN / HALF /
    QUARTER image this window was given (at<uchar>(y, x) is relative to the ROI's origin; width and height are the ROI's)."""
    img, x0, y0, width, height = patches[c.scale[i]]        # data.cpp:20-34 (any other value: dieWithMsg)
    o1x, o1y = st_apply(stp, c.o1x[i], c.o1y[i])
    o2x, o2y = st_apply(stp, c.o2x[i], c.o2y[i])
    x1 = (shape[2 * c.lm1[i]] + o1x) * width
    y1 = (shape[2 * c.lm1[i] + 1] + o1y) * height
    x2 = (shape[2 * c.lm2[i]] + o2x) * width
    y2 = (shape[2 * c.lm2[i] + 1] + o2y) * height
    x1_, y1_, x2_, y2_ = int(c_round(x1)), int(c_round(y1)), int(c_round(x2)), int(c_round(y2))

    def clamp(x, y):                            # common.hpp:227-232
        if x < 0: x = 0
        if y < 0: y = 0
        if x >= width: x = width - 1
        if y >= height: y = height - 1
        return x, y
    x1_, y1_ = clamp(x1_, y1_)
    x2_, y2_ = clamp(x2_, y2_)
    return int(img[y0 + y1_][x0 + x1_]) - int(img[y0 + y2_][x0 + x2_])


def forward(m, c, patches, shape, stp):
    node = 1                                    # cart.cpp:394-403
    for _ in range(m.D - 1):
        val = feature_value(c, node, patches, shape, stp)
        node = 2 * node if val <= c.nth[node] else 2 * node + 1
    return node - (1 << (m.D - 1))


FNV_SEED, FNV_MUL = 2166136261, 16777619        # the trace's leaf-path hash (a convention of this repo's checkers, not the reference's)


def validate(m, img, x0=None, y0=None, width=None, height=None, patches=None, similarity=False):
    """JoinCascador::Validate -> (is_face, score, shape, n, path_hash).  Either one ORIGIN patch (img, x0, y0, width, height)
    or the three patches of a multi-scale call."""
    if patches is None:
        patches = ((img, x0, y0, width, height),) * 3     # (a single-scale model only ever reads patches[0])
    shape = list(m.mean_shape)
    score = 0.0
    n = 0
    h = FNV_SEED
    base = 1 << (m.D - 1)
    stp = IDENTITY
    for t in range(m.stage_idx):
        # STParameter::Calc(shape, mean_shape) (cascador.cpp:180); the identity with the transform off (data.cpp:68-70)
        stp = st_calc(shape, m.mean_shape, m.L) if similarity else IDENTITY
        lbf = [0] * m.K
        offset = 0
        for k in range(m.K):
            c = m.carts[t][k]
            idx = forward(m, c, patches, shape, stp)
            h = ((h ^ idx) * FNV_MUL) & 0xffffffff
            score += c.scores[idx]
            score = (score - c.mean) / c.std
            n += 1
            if score < c.th:
                return False, score, shape, n, h
            lbf[k] = offset + idx
            offset += base
        # GenDeltaShape (btcart.cpp:407-424): rows summed from zero in cart order, Apply, then shape += delta
        delta = [0.0] * (2 * m.L)
        for k in range(m.K):
            row = m.w[t][lbf[k]]
            for j in range(2 * m.L):
                delta[j] += row[j]
        for i in range(m.L):
            delta[2 * i], delta[2 * i + 1] = st_apply(stp, delta[2 * i], delta[2 * i + 1])
        for j in range(2 * m.L):
            shape[j] = shape[j] + delta[j]
    if m.stage_idx < m.T:
        for k in range(m.cart_idx + 1):         # cascador.cpp:198-209: no regression for the stage in training
            c = m.carts[m.stage_idx][k]
            idx = forward(m, c, patches, shape, stp)
            h = ((h ^ idx) * FNV_MUL) & 0xffffffff
            score += c.scores[idx]
            score = (score - c.mean) / c.std
            n += 1
            if score < c.th:
                return False, score, shape, n, h
    return True, score, shape, n, h


def windows_method1(w, h, minimum_size, step, factor):
    """detectMultiScale1's enumeration (cascador.cpp:332-370): (x, y, win) in scan order."""
    out = []
    win_w = win_h = minimum_size
    while win_w <= w and win_h <= h:
        y = 0
        while y <= h - win_h:
            x = 0
            while x <= w - win_w:
                out.append((x, y, win_w))
                x += step
            y += step
        prev = win_w
        win_w = int(win_w * factor)
        win_h = int(win_h * factor)
        if win_w <= prev:
            break                               # (a factor that does not grow the window loops forever in the reference)
    return out


def nms(rects, scores, overlap):
    """cascador.cpp:387-429.  std::multimap<double, int>: keys ascending, equal keys in insertion order."""
    keys, vals = [], []
    for i, s in enumerate(scores):
        p = bisect.bisect_right(keys, s)        # insert() puts an equal key at the upper bound
        keys.insert(p, s); vals.insert(p, i)
    areas = [float(r[2] * r[3]) for r in rects]
    picked = []
    while keys:
        last = vals[-1]
        picked.append(last)
        nk, nv = [], []
        for s, idx in zip(keys, vals):
            x1 = float(max(rects[idx][0], rects[last][0]))
            y1 = float(max(rects[idx][1], rects[last][1]))
            x2 = float(min(rects[idx][0] + rects[idx][2], rects[last][0] + rects[last][2]))
            y2 = float(min(rects[idx][1] + rects[idx][3], rects[last][1] + rects[last][3]))
            ww = max(0.0, x2 - x1)
            hh = max(0.0, y2 - y1)
            ov = ww * hh / (areas[idx] + areas[last] - ww * hh)
            if not ov > overlap:
                nk.append(s); nv.append(idx)
        if len(nk) == len(keys):
            raise RuntimeError("the reference's nms does not terminate for this overlap threshold")
        keys, vals = nk, nv
    return picked


def _finish(m, rects, scores, shapes, overlap, do_nms):
    """Detect's tail (cascador.cpp:445-476): NMS or everything, then the shapes relocated into their rects."""
    picked = nms(rects, scores, overlap) if do_nms else list(range(len(rects)))
    out_r, out_s, out_sh = [], [], []
    for i in picked:
        r = rects[i]
        sh = list(shapes[i])
        for j in range(m.L):                    # cascador.cpp:466-469
            sh[2 * j] = r[0] + sh[2 * j] * r[2]
            sh[2 * j + 1] = r[1] + sh[2 * j + 1] * r[3]
        out_r.append(r); out_s.append(scores[i]); out_sh.append(sh)
    return out_r, out_s, out_sh


def patches_method1(img, img_h, img_q, x, y, win):
    """The three ROIs detectMultiScale1 cuts for a window (cascador.cpp:340-353)."""
    r = math.sqrt(2.)
    return ((img, x, y, win, win),
            (img_h, int(x / r), int(y / r), int(win / r), int(win / r)),
            (img_q, x // 2, y // 2, win // 2, win // 2))


def detect(m, img, minimum_size=20, step=5, factor=1.2, overlap=0.3, do_nms=True, resize=None, trace=None, similarity=False):
    """JoinCascador::Detect with fddb.method = 1 -> rects (x, y, w, h), scores, relocated shapes.
    resize(img, dw, dh): needed for multi-scale models only (img_h, img_q: cascador.cpp:323-331).  trace: a list that
    receives every window's (is_face, score, shape, n, path_hash) in scan order."""
    h, w = len(img), len(img[0])
    img_h = img_q = img
    if resize is not None:
        img_h = resize(img, int(w / math.sqrt(2.)), int(h / math.sqrt(2.)))
        img_q = resize(img, w // 2, h // 2)
    rects, scores, shapes = [], [], []
    for (x, y, win) in windows_method1(w, h, minimum_size, step, factor):
        res = validate(m, None, patches=patches_method1(img, img_h, img_q, x, y, win), similarity=similarity)
        if trace is not None:
            trace.append(res)
        ok, score, shape, _, _ = res
        if ok:
            rects.append((x, y, win, win)); scores.append(score); shapes.append(shape)
    return _finish(m, rects, scores, shapes, overlap, do_nms)


def detect_pyramid(m, img, resize, origin_size=48, half_size=36, quarter_size=24, step=5, factor=1.2, overlap=0.3, do_nms=True):
    """JoinCascador::Detect with fddb.method = 0: detectMultiScale (cascador.cpp:271-308) over detectSingleScale
    (cascador.cpp:215-265).  Windows are origin_size on every level; each window's three patches are cv::resize's of its ROI
    to image_size.{origin,half,quarter}_size (cascador.cpp:243-245) -- resize() stands in for cv::resize."""
    width, height = len(img[0]), len(img)
    win = origin_size
    scale = 1.
    cur = img
    rects, scores, shapes = [], [], []
    while width >= win and height >= win:
        lv_r, lv_s, lv_sh = [], [], []
        x_max, y_max = len(cur[0]) - win, len(cur) - win
        y = 0
        while y <= y_max:
            x = 0
            while x <= x_max:
                roi = [row[x:x + win] for row in cur[y:y + win]]
                p_o = resize(roi, origin_size, origin_size)
                # (the reference resizes all three for every window; a single-scale model never reads the other two)
                p_h = resize(roi, half_size, half_size) if m.multi_scale else None
                p_q = resize(roi, quarter_size, quarter_size) if m.multi_scale else None
                ok, score, shape, _, _ = validate(m, None, patches=((p_o, 0, 0, origin_size, origin_size),
                                                                     (p_h, 0, 0, half_size, half_size),
                                                                     (p_q, 0, 0, quarter_size, quarter_size)))
                if ok:
                    lv_r.append((x, y, win, win)); lv_s.append(score); lv_sh.append(shape)
                x += step
            y += step
        for (rx, ry, rw, rh) in lv_r:           # cascador.cpp:290-294: int *= double
            rects.append((int(rx * scale), int(ry * scale), int(rw * scale), int(rh * scale)))
        scores += lv_s; shapes += lv_sh
        scale *= factor
        width = int(width / factor)
        height = int(height / factor)
        if width <= 0 or height <= 0:
            break                               # (cv::resize would throw on an empty size; the loop ends on the next test anyway)
        cur = resize(cur, width, height)
    return _finish(m, rects, scores, shapes, overlap, do_nms)


# ---- cv::resize(src, dst, Size(dw, dh)) for 8-bit gray, INTER_LINEAR: a second restatement, from OpenCV's imgproc/imgwarp.cpp as
#      the author of this file remembers it (2.4 / 3.x line; not from jda_oracle.c's orc_resize_cv).  Still NOT a pin: no OpenCV
#      here to run.  What it encodes: float coordinates fx = (float)((dx + 0.5) * scale - 0.5), cvFloor, the border rules
#      (x: sx < 0 -> 0 with fx 0; sx >= w - 1 -> w - 1 with fx 0; y: coefficients untouched, the two row indices clipped), 11-bit coefficients saturate_cast<short>(c * 2048) with
#      round-half-to-even, the horizontal pass in ints, the vertical pass ((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16)
#      + 2) >> 2, and resize()'s switch to the 2x2 area average (a + b + c + d + 2) >> 2 when both scales are exactly 2.
import numpy as _np


def _cv_round_f(v):
    """cvRound of a float: to nearest, halves to even (lrint in the default rounding mode)."""
    f = math.floor(v)
    d = v - f
    if d > 0.5 or (d == 0.5 and (int(f) & 1)):
        f += 1
    return int(f)


def resize_cv2(img, dw, dh):
    src = _np.asarray(img, _np.int64)
    sh, sw = src.shape
    if (dw, dh) == (sw, sh):
        return src.tolist()
    scale_x, scale_y = 1. / (float(dw) / sw), 1. / (float(dh) / sh)
    isx, isy = _cv_round_f(scale_x), _cv_round_f(scale_y)           # saturate_cast<int>(double)
    eps = 2.220446049250313e-16
    if abs(scale_x - isx) < eps and abs(scale_y - isy) < eps and isx == 2 and isy == 2:
        out = [[0] * dw for _ in range(dh)]
        for y in range(dh):
            for x in range(dw):
                out[y][x] = int((src[2 * y][2 * x] + src[2 * y][2 * x + 1] + src[2 * y + 1][2 * x] + src[2 * y + 1][2 * x + 1] + 2) >> 2)
        return out
    f32 = _np.float32

    def coeffs(n_dst, n_src, scale, clamp):
        ofs, co = [], []
        for d in range(n_dst):
            f = f32((d + 0.5) * scale - 0.5)
            s = int(math.floor(float(f)))
            f = f32(f - f32(s))
            if clamp:                           # the x loop only (xmin / xmax bookkeeping); the y loop keeps (sy, fy) as they are
                if s < 0:
                    f, s = f32(0), 0
                if s >= n_src - 1:
                    f, s = f32(0), n_src - 1
            c0, c1 = f32(1) - f, f
            sat = lambda v: max(-32768, min(32767, _cv_round_f(float(f32(v) * f32(2048)))))
            ofs.append(s); co.append((sat(c0), sat(c1)))
        return ofs, co
    xofs, alpha = coeffs(dw, sw, scale_x, True)
    yofs, beta = coeffs(dh, sh, scale_y, False)       # ... and clips the two ROW INDICES when it fetches them

    def hrow(y):
        y = min(max(y, 0), sh - 1)
        row = src[y]
        return [int(row[xofs[d]]) * alpha[d][0] + int(row[min(xofs[d] + 1, sw - 1)]) * alpha[d][1] for d in range(dw)]
    out = []
    for d in range(dh):
        r0, r1 = hrow(yofs[d]), hrow(yofs[d] + 1)
        b0, b1 = beta[d]
        out.append([(((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2 for x in range(dw)])
    return out
