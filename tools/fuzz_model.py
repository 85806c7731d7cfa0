"""Mutated model streams into jdaCascadorCreateFloat / Double (+ SerializeTo on the ones that load): a create either returns
NULL with a reason or a handle whose header passed validation -- it never crashes or reads out of bounds.  CPU only; run it on
the sanitizer build:  JDA_LIB_PATH=jda_amd/libjda_asan.so LD_PRELOAD=$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libstdc++.so.6) \
                      ASAN_OPTIONS=detect_leaks=0 python tools/fuzz_model.py [rounds]"""
import os
import struct
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jda_amd import api, synth  # noqa: E402


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    rng = np.random.default_rng(5)
    tmp = tempfile.mkdtemp(prefix="jda_fuzz_")
    base = {}
    for rb in (4, 8):
        for dims in ((2, 8, 5, 3), (1, 4, 3, 2), (3, 20, 5, 4)):
            base[(rb, dims)] = synth.make_model(*dims, seed=1, cart_th=-1.0).tobytes(rb)
    keys = list(base)
    p = os.path.join(tmp, "m.model")
    q = os.path.join(tmp, "out.model")
    loaded = refused = 0
    for it in range(rounds):
        rb, dims = keys[rng.integers(len(keys))]
        b = bytearray(base[(rb, dims)])
        kind = rng.integers(7)
        if kind == 0:                                   # truncate anywhere
            b = b[:rng.integers(0, len(b))]
        elif kind == 1:                                 # a header int replaced (sizes, depth, negative, huge)
            i = int(rng.integers(0, 7))
            v = int(rng.choice([0, -1, 1, 2, 7, 31, 32, 33, 255, 65536, 2**31 - 1, -2**31, int(rng.integers(-50, 5000))]))
            b[4 * i:4 * i + 4] = struct.pack("<i", v)
        elif kind == 2:                                 # random bytes flipped
            for _ in range(int(rng.integers(1, 40))):
                b[int(rng.integers(len(b)))] = int(rng.integers(256))
        elif kind == 3:                                 # extra bytes appended
            b += bytes(rng.integers(0, 256, int(rng.integers(1, 4096)), dtype=np.uint8))
        elif kind == 4:                                 # node fields: landmark ids / scale out of range somewhere in the body
            off = 28 + 2 * dims[2] * rb
            for _ in range(int(rng.integers(1, 10))):
                j = off + 4 * int(rng.integers(0, max(1, (len(b) - off - 4) // 4)))
                b[j:j + 4] = struct.pack("<i", int(rng.choice([-1, dims[2], 1000, 3, -7, 2**30])))
        elif kind == 5:                                 # the other real size's creator on this file
            pass
        else:                                           # empty / tiny
            b = b[:int(rng.integers(0, 40))]
        with open(p, "wb") as f:
            f.write(b)
        creators = (api.lib.jdaCascadorCreateFloat, api.lib.jdaCascadorCreateDouble)
        order = creators if rb == 4 else creators[::-1]
        if kind == 5:
            order = order[::-1]
        for cr in order:
            h = cr(p.encode())
            if h:
                loaded += 1
                api.lib.jdaCascadorSerializeTo(h, q.encode())
                api.lib.jdaCascadorRelease(h)
            else:
                refused += 1
                assert api.lib.jdaGetLastError(), "a refused create left no reason"
    print("fuzz_model: %d streams, %d creates accepted, %d refused with a reason, no crash" % (rounds, loaded, refused))


if __name__ == "__main__":
    main()
