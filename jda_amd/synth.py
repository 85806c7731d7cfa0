"""Deterministic synthetic cascade models and frames.

The reference's trained model blob is not in the tree (reference
.MISSING_LARGE_BLOBS:1) and neither is FDDB, so every benchmark and parity
case runs on a synthetic model of the right dimensions written in the
reference's binary layout (reference README.md:84-111, cascador.cpp:79-124,
cart.cpp:429-450) and on synthetic 8-bit frames.

Everything is a pure function of its seed: the same model/frames are
regenerated on the GPU box from the same arguments.

The threshold calibrator below walks the cascade in numpy for a SAMPLE of
windows only to choose per-cart thresholds for the "cascade" regime; it is
bench tooling, not the detector and not the oracle.
"""
import math
import os

import numpy as np

RADIUS = [0.3, 0.2, 0.15, 0.12, 0.1]  # reference model/config.json:22
NEG_BIG = -3.0e38  # "never reject" cart threshold (finite in f32 and f64)


class Model:
    """Cascade parameters as numpy arrays (f64 storage, like the trainer's file)."""

    def __init__(self, T, K, L, D):
        self.T, self.K, self.L, self.D = T, K, L, D
        self.node_n = (1 << (D - 1)) - 1
        self.leaf_n = 1 << (D - 1)
        self.dim = 2 * L
        self.mean_shape = np.zeros(self.dim, np.float64)
        self.scale = np.zeros((T, K, self.node_n), np.int32)
        self.lm1 = np.zeros((T, K, self.node_n), np.int32)
        self.lm2 = np.zeros((T, K, self.node_n), np.int32)
        self.off = np.zeros((T, K, self.node_n, 4), np.float64)
        self.nth = np.zeros((T, K, self.node_n), np.int32)
        self.leaf = np.zeros((T, K, self.leaf_n), np.float64)
        self.cth = np.zeros((T, K), np.float64)
        self.cmean = np.zeros((T, K), np.float64)
        self.cstd = np.ones((T, K), np.float64)
        self.w = np.zeros((T, K * self.leaf_n, self.dim), np.float64)

    def stream_bytes(self, real_bytes):
        cart = self.node_n * (16 + 4 * real_bytes) + self.leaf_n * real_bytes + 3 * real_bytes
        stage = self.K * cart + self.K * self.leaf_n * self.dim * real_bytes
        return 28 + self.dim * real_bytes + self.T * stage + 4

    def tobytes(self, real_bytes=8, header_stage=None, header_cart=-1):
        """Serialise in the reference layout; real_bytes 8 = trainer file, 4 = C float file."""
        rt = "<f8" if real_bytes == 8 else "<f4"
        node_dt = np.dtype([("scale", "<i4"), ("lm1", "<i4"), ("lm2", "<i4"), ("off", rt, (4,)), ("th", "<i4")])
        cart_dt = np.dtype([("nodes", node_dt, (self.node_n,)), ("leaf", rt, (self.leaf_n,)),
                            ("th", rt), ("mean", rt), ("std", rt)])
        assert node_dt.itemsize == 16 + 4 * real_bytes
        if header_stage is None:
            # complete trainer model: (T, -1) (cascador.cpp:93-98); float files
            # written by the C library carry (T+1, -1) (c/jda.c:662-665)
            header_stage = self.T if real_bytes == 8 else self.T + 1
        parts = [np.array([0, self.T, self.K, self.L, self.D, header_stage, header_cart], "<i4").tobytes(),
                 self.mean_shape.astype(rt).tobytes()]
        for t in range(self.T):
            carts = np.zeros(self.K, cart_dt)
            carts["nodes"]["scale"] = self.scale[t]
            carts["nodes"]["lm1"] = self.lm1[t]
            carts["nodes"]["lm2"] = self.lm2[t]
            carts["nodes"]["off"] = self.off[t]
            carts["nodes"]["th"] = self.nth[t]
            carts["leaf"] = self.leaf[t]
            carts["th"] = self.cth[t]
            carts["mean"] = self.cmean[t]
            carts["std"] = self.cstd[t]
            parts.append(carts.tobytes())
            parts.append(self.w[t].astype(rt).tobytes())
        parts.append(np.array([0], "<i4").tobytes())
        blob = b"".join(parts)
        assert len(blob) == self.stream_bytes(real_bytes)
        return blob

    def save(self, path, real_bytes=8, **kw):
        with open(path, "wb") as f:
            f.write(self.tobytes(real_bytes, **kw))
        return path


def make_mean_shape(L, rng):
    """A plausible face-like point set inside [0.2, 0.8]^2."""
    ang = np.linspace(0.0, 2.0 * math.pi, L, endpoint=False) + rng.uniform(0, 0.3)
    rad = 0.12 + 0.16 * ((np.arange(L) % 3) / 2.0)
    pts = np.stack([0.5 + rad * np.cos(ang), 0.52 + rad * np.sin(ang) * 1.1], 1)
    pts += rng.uniform(-0.02, 0.02, pts.shape)
    return np.clip(pts, 0.2, 0.8).reshape(-1)


def make_model(T=5, K=540, L=27, D=4, seed=1, cart_th=NEG_BIG, multi_scale=False,
               f32_exact=True, w_sigma=2e-3, norm_every=None):
    """Random cascade following SURVEY.md 8d's synthetic-model recipe.

    cart_th      scalar threshold given to every cart (NEG_BIG = all-pass regime);
                 use calibrate_thresholds() afterwards for the cascade regime.
    f32_exact    draw every real in float32 so the f64 and f32 files describe
                 exactly the same model (narrowing is then the identity).
    norm_every   carts whose 1-based index is a multiple get a non-trivial
                 (mean, std); default 10*L like the trainer (btcart.cpp:130,173-181,
                 model/config.json:27).
    """
    rng = np.random.default_rng(seed)
    m = Model(T, K, L, D)
    f = (lambda a: np.asarray(a, np.float32).astype(np.float64)) if f32_exact else (lambda a: np.asarray(a, np.float64))
    m.mean_shape = f(make_mean_shape(L, rng))
    shp = (T, K, m.node_n)
    m.scale = (rng.integers(0, 3, shp) if multi_scale else np.zeros(shp)).astype(np.int32)
    m.lm1 = rng.integers(0, L, shp).astype(np.int32)
    m.lm2 = rng.integers(0, L, shp).astype(np.int32)
    # offsets uniform in a disc of radius RADIUS[t] (generation rule cart.cpp:361-388)
    n = T * K * m.node_n * 2
    pts = np.empty((0, 2))
    while len(pts) < n:
        c = rng.uniform(-1, 1, (2 * n + 16, 2))
        pts = np.concatenate([pts, c[(c ** 2).sum(1) <= 1.0]])
    pts = pts[:n].reshape(T, K, m.node_n, 2, 2)
    rad = np.array([RADIUS[min(t, len(RADIUS) - 1)] for t in range(T)]).reshape(T, 1, 1, 1, 1)
    m.off = f((pts * rad).reshape(T, K, m.node_n, 4))
    m.nth = rng.integers(-40, 41, shp).astype(np.int32)
    m.leaf = f(rng.normal(0.0, 0.5, (T, K, m.leaf_n)))
    m.cth = f(np.full((T, K), cart_th))
    if norm_every is None:
        norm_every = 10 * L
    kk = np.arange(1, K + 1)
    normed = (kk % norm_every) == 0
    m.cmean = f(np.where(normed[None, :], rng.normal(0, 0.1, (T, K)), 0.0))
    m.cstd = f(np.where(normed[None, :], rng.uniform(0.8, 1.25, (T, K)), 1.0))
    m.w = f(rng.normal(0.0, w_sigma, (T, K * m.leaf_n, m.dim)))
    return m


def make_frames(n, width, height, seed=0, first=0):
    """n u8 frames: 4 low-frequency cosines (amplitude 60 in total) + noise (sigma 12) + 128.

    Frame i depends only on (seed, first + i), so any slice of a batch can be
    regenerated on its own (used by the multi-GPU sharding).
    """
    out = np.empty((n, height, width), np.uint8)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    for i in range(n):
        rng = np.random.default_rng([seed, first + i])
        img = np.full((height, width), 128.0, np.float32)
        for _ in range(4):
            fx, fy = rng.uniform(-3.0, 3.0, 2) * 2 * math.pi / max(width, height)
            ph = rng.uniform(0, 2 * math.pi)
            img += np.float32(15.0) * np.cos(xx * np.float32(fx) + yy * np.float32(fy) + np.float32(ph))
        img += rng.normal(0.0, 12.0, img.shape).astype(np.float32)
        out[i] = np.clip(img, 0, 255).astype(np.uint8)
    return out


# ----------------------------------------------------------------------------
# window enumeration in Python (reference c/jda.c:320-339,459-460), used by the
# calibrator, by tests and by bench accounting
# ----------------------------------------------------------------------------

def levels_c(width, height, scale=1.25, min_size=40, max_size=-1):
    f32 = np.float32
    min_size = max(min_size, 24)
    if max_size <= 0:
        max_size = min(width, height)
    max_size = min(max_size, width, height)
    grow = lambda w: int(f32(w) * f32(scale))
    if grow(24) <= 24:
        raise ValueError("scale does not grow the window")
    win, out, base = 24, [], 0
    while win < min_size:
        win = grow(win)
    while win <= max_size:
        step = int(f32(win) * f32(0.1))
        nx, ny = (width - win) // step + 1, (height - win) // step + 1
        out.append(dict(win=win, step=step, nx=nx, ny=ny, base=base))
        base += nx * ny
        win = grow(win)
    return out, base


def levels_cpp(width, height, minimum_size=20, step=5, factor=1.2):
    """detectMultiScale1's window sizes (reference src/jda/cascador.cpp:310-376)."""
    if minimum_size < 1 or step < 1 or int(minimum_size * factor) <= minimum_size:
        raise ValueError("bad scan parameters")
    win, out, base = minimum_size, [], 0
    while win <= width and win <= height:
        nx, ny = (width - win) // step + 1, (height - win) // step + 1
        out.append(dict(win=win, step=step, nx=nx, ny=ny, base=base))
        base += nx * ny
        win = int(win * factor)
    return out, base


def count_windows_cpp(width, height, minimum_size=20, step=5, factor=1.2):
    return levels_cpp(width, height, minimum_size, step, factor)[1]


def window_table(width, height, scale=1.25, min_size=40, max_size=-1):
    """(x, y, win) int32 arrays of every window of a frame in scan order."""
    lv, tot = levels_c(width, height, scale, min_size, max_size)
    xs, ys, ws = [], [], []
    for l in lv:
        iy, ix = np.mgrid[0:l["ny"], 0:l["nx"]]
        xs.append((ix * l["step"]).ravel())
        ys.append((iy * l["step"]).ravel())
        ws.append(np.full(l["nx"] * l["ny"], l["win"]))
    cat = lambda a: (np.concatenate(a) if a else np.zeros(0)).astype(np.int32)
    return cat(xs), cat(ys), cat(ws)


# ----------------------------------------------------------------------------
# threshold calibration ("cascade" regime)
# ----------------------------------------------------------------------------

def calibrate_thresholds(m, frames, tau=27.0, p_final=1e-3, sample=60000, seed=7,
                         scale=1.25, min_size=40, max_size=-1):
    """Set per-cart thresholds so that the fraction of windows still alive after
    global cart c follows max(p_final, exp(-(c+1)/tau)) on a random sample of the
    windows of `frames`: mean reject length ~= tau carts, ~p_final of the windows
    finish (the "Average Cart_N to Reject" regime the reference logs,
    src/test.cpp:154-157).  Dialect-C arithmetic in float32; scale==0 nodes only.
    Returns the realised survival curve on the sample.
    """
    assert not m.scale.any(), "calibration supports scale==0 models"
    f32 = np.float32
    n_frames, H, W = frames.shape
    xs, ys, ws = window_table(W, H, scale, min_size, max_size)
    rng = np.random.default_rng(seed)
    tot = len(xs) * n_frames
    pick = rng.choice(tot, size=min(sample, tot), replace=False)
    fi, wi = pick // len(xs), pick % len(xs)
    x0, y0, win = xs[wi].astype(np.int64), ys[wi].astype(np.int64), ws[wi]
    base = fi.astype(np.int64) * (H * W) + y0 * W + x0
    flat = frames.reshape(-1)
    winf = win.astype(f32)
    n0 = len(pick)
    alive = np.arange(n0)
    score = np.zeros(n0, f32)
    shape = np.tile(m.mean_shape.astype(f32), (n0, 1))
    off = m.off.astype(f32)
    leafv = m.leaf.astype(f32)
    cmean, cstd = m.cmean.astype(f32), m.cstd.astype(f32)
    wf = None
    curve = []
    target_prev = 1.0
    for t in range(m.T):
        lbf = np.zeros((len(alive), m.K), np.int64)
        pos = np.arange(len(alive))   # position of each alive window inside stage_ids
        for k in range(m.K):
            c = t * m.K + k
            sh = shape[alive]
            wn = winf[alive]
            at = np.zeros(len(alive), np.int64)
            for _ in range(m.D - 1):
                l1, l2 = m.lm1[t, k][at], m.lm2[t, k][at]
                o = off[t, k][at]
                r = np.arange(len(alive))
                def coord(v):
                    q = np.trunc(v * wn)
                    q = np.where(np.abs(q) < 2 ** 31, q, -2.0 ** 31).astype(np.int64)
                    return np.clip(q, 0, win[alive] - 1)
                x1 = coord(sh[r, 2 * l1] + o[:, 0]); y1 = coord(sh[r, 2 * l1 + 1] + o[:, 1])
                x2 = coord(sh[r, 2 * l2] + o[:, 2]); y2 = coord(sh[r, 2 * l2 + 1] + o[:, 3])
                b = base[alive]
                feat = flat[b + y1 * W + x1].astype(np.int64) - flat[b + y2 * W + x2].astype(np.int64)
                at = 2 * at + np.where(feat <= m.nth[t, k][at], 1, 2)
            leaf = at - m.node_n
            lbf[pos, k] = k * m.leaf_n + leaf
            s = score[alive] + leafv[t, k][leaf]
            s = ((s - cmean[t, k]) / cstd[t, k]).astype(f32)
            score[alive] = s
            target = max(p_final, math.exp(-(c + 1) / tau))
            n_keep = min(len(alive), int(round(target * n0)))
            n_rej = len(alive) - n_keep
            if n_rej > 0:
                srt = np.sort(s)
                th = srt[n_rej]            # everything strictly below is rejected
                m.cth[t, k] = float(th)
                keep = s >= th
                alive, pos = alive[keep], pos[keep]
            else:
                m.cth[t, k] = NEG_BIG
            curve.append(len(alive) / n0)
            if len(alive) == 0:
                break
        if len(alive) == 0:
            for tt in range(t, m.T):
                for kk in range(m.K):
                    if tt * m.K + kk > c:
                        m.cth[tt, kk] = NEG_BIG
            break
        # stage regression for the survivors, rows added in cart order
        wt = m.w[t].astype(f32)
        acc = shape[alive]
        rows = lbf[pos]
        for k in range(m.K):
            acc = (acc + wt[rows[:, k]]).astype(f32)
        shape[alive] = acc
    return np.array(curve)


def cache_dir():
    d = os.environ.get("JDA_SYNTH_CACHE", "/tmp/jda_synth_cache")
    os.makedirs(d, exist_ok=True)
    return d


def make_dyadic_model(T, K, L, D, win, seed=1, reject=0.0, norm_every=7):
    """A model on which the two numeric dialects MUST walk the same path (tests/test_dialect_differential.py).

    Every real is a small dyadic rational, chosen so that for windows of side `win` (a power of two):
      * (shape + offset) * win is an integer + 0.25 in every stage -- truncation (c/jda.c:373-381) and round()
        (src/jda/data.cpp:37-51) pick the same pixel, and fp32 and fp64 compute it without rounding;
      * every score and every regressed shape coordinate is a sum of multiples of 2^-10 far below 2^14, exact in
        fp32 and in fp64 alike, in any order of accumulation (the sum-from-zero of btcart.cpp:407-424 included);
      * normalising carts divide by a power of two.
    What the dialects still do differently on such a model is nothing: equal leaf paths, equal reject positions,
    scores and shapes equal after fp64 -> fp32 conversion.  reject > 0: cart thresholds on the same grid, so that
    about that fraction of the still-alive windows dies per cart.
    """
    assert win & (win - 1) == 0 and win >= 16
    rng = np.random.default_rng(seed)
    m = Model(T, K, L, D)
    g, bias = 1.0 / win, 0.25 / win
    m.mean_shape = rng.integers(int(0.25 * win), int(0.75 * win) + 1, m.dim) * g + bias
    shp = (T, K, m.node_n)
    m.lm1 = rng.integers(0, L, shp).astype(np.int32)
    m.lm2 = rng.integers(0, L, shp).astype(np.int32)
    r = max(1, int(0.3 * win))
    m.off = rng.integers(-r, r + 1, shp + (4,)) * g
    m.nth = rng.integers(-40, 41, shp).astype(np.int32)
    m.leaf = rng.integers(-256, 257, (T, K, m.leaf_n)) / 256.0
    kk = np.arange(1, K + 1)
    normed = (kk % norm_every) == 0
    m.cmean = np.where(normed[None, :], rng.integers(-64, 65, (T, K)) / 256.0, 0.0)
    m.cstd = np.where(normed[None, :], np.where((kk // norm_every) % 2 == 1, 2.0, 0.5)[None, :], 1.0) * np.ones((T, K))
    m.w = rng.choice(np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, g, -g]), (T, K * m.leaf_n, m.dim))
    if reject > 0:
        # a threshold that climbs with the running score's spread (~0.29 * sqrt(c)): roughly `reject` of the live
        # windows fall below it at every cart; on the 2^-8 grid
        c = np.arange(1, T * K + 1, dtype=np.float64).reshape(T, K)
        m.cth = np.round((-1.6 + 3.2 * reject) * 0.29 * np.sqrt(c) * 256.0) / 256.0
    else:
        m.cth = np.full((T, K), NEG_BIG)
    return m
