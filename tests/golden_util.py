"""Loading of the committed golden vectors (tests/golden/, made by make_golden.py)."""
import hashlib
import json
import os

import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
INDEX = json.load(open(os.path.join(HERE, "index.json")))
NAMES = sorted(INDEX)


def load(name, tmp_dir):
    """-> (meta, arrays, model_path).  Models too big to inline are regenerated
    from their generator arguments and checked against the stored sha256."""
    from jda_amd import synth
    meta = INDEX[name]
    z = np.load(os.path.join(HERE, name + ".npz"))
    arrays = {k: z[k] for k in z.files}
    if meta["model_inline"]:
        blob = arrays["model_bytes"].tobytes()
    else:
        blob = synth.make_model(*meta["dims"], **meta["model_kwargs"]).tobytes(8)
    assert hashlib.sha256(blob).hexdigest() == meta["model_sha256"], "synthetic model generator drifted"
    path = os.path.join(str(tmp_dir), "golden_%s.model" % name)
    with open(path, "wb") as f:
        f.write(blob)
    return meta, arrays, path
