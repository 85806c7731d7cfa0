"""Quick GPU shake-out: GPU trace/detect vs the CPU oracle on small cases."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jda_amd import synth, api
from oracle.pyoracle import Oracle

def bits(a):
    return a.view(np.uint32) if a.dtype == np.float32 else (a.view(np.uint64) if a.dtype == np.float64 else a)

def check(dims, th, size=(200, 150), nframes=2, multi=False, seed=3):
    m = synth.make_model(*dims, seed=seed, cart_th=th, norm_every=5, multi_scale=multi)
    p = '/tmp/q_%d_%d_%d_%d.model' % dims
    m.save(p, 8)
    frames = synth.make_frames(nframes, size[0], size[1], seed=11)
    o = Oracle(p); c = api.Cascador(p)
    t0 = time.time(); g = c.trace(frames); t1 = time.time()
    bad = {}
    for i in range(nframes):
        r = o.trace(frames[i])
        n = len(r['carts_n'])
        sl = slice(i * n, (i + 1) * n)
        for k in r:
            eq = bits(r[k]) == bits(g[k][sl])
            if eq.ndim > 1: eq = eq.all(1)
            bad[k] = bad.get(k, 0) + int((~eq).sum())
    dets = c.detect_batch(frames, th=-0.5)
    dbad = 0
    for i in range(nframes):
        r = o.detect(frames[i], th=-0.5)
        for k in r:
            if r[k].shape != dets[i][k].shape or not np.array_equal(bits(r[k]), bits(dets[i][k])): dbad += 1
    print(dims, 'th', th, 'multi', multi, 'windows', n, 'trace mismatches', bad, 'detect mismatches', dbad,
          'ndet', [len(d['scores']) for d in dets], 'gpu trace s %.2f' % (t1 - t0), flush=True)
    return sum(bad.values()) + dbad

if __name__ == '__main__':
    tot = 0
    for dims in [(2, 8, 5, 3), (3, 20, 5, 4), (2, 6, 4, 6), (1, 4, 3, 2), (3, 70, 9, 5), (2, 64, 68, 6)]:
        for th in (synth.NEG_BIG, -1.0, -0.3):
            tot += check(dims, th)
    tot += check((3, 20, 5, 4), -1.0, multi=True)
    tot += check((5, 540, 27, 4), synth.NEG_BIG, size=(160, 120), nframes=1)
    tot += check((5, 540, 27, 4), -2.0, size=(320, 240), nframes=2)
    os.environ['JDA_NO_FAST_SCAN'] = '1'
    tot += check((3, 20, 5, 4), -1.0)
    print('TOTAL MISMATCHES', tot)
    sys.exit(1 if tot else 0)
