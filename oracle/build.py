#!/usr/bin/env python3
"""Build recipe for the CPU checkers under oracle/ (TEST INFRASTRUCTURE).

  oracle/libjda_oracle.so        our own C restatement (jda_oracle.c)
  oracle/_ref/libjda_ref_<T>_<K>_<L>_<D>.so
                                 the reference's c/jda.c, compiled from where it
                                 lies under /root/reference (never copied into
                                 the repo) with its four compile-time dimension
                                 #defines (c/jda.c:24-27) rewritten on the fly
                                 and oracle/ref_harness.c appended; one library
                                 per dimension set.

Flags follow SURVEY.md 8c: ISO C99 (=> no FMA contraction), -O2, no
-march=native, no -ffast-math.  oracle/_ref/ is git-ignored but is NOT
gpurun-ignored, so the prebuilt libraries travel to the GPU box, where
/root/reference does not exist and nothing is rebuilt.
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF_ROOT = os.environ.get("JDA_REFERENCE_ROOT", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "c", "jda.c")

# dimension sets used by tests, fixtures and the bench CPU baseline
REF_DIMS = [
    (5, 540, 27, 4),   # shipped model dims (SURVEY "S")
    (2, 8, 5, 3),
    (3, 20, 5, 4),
    (2, 6, 4, 6),
    (1, 4, 3, 2),
    (3, 70, 9, 5),
    (2, 64, 68, 6),    # config-5-like: 68 landmarks, depth 6 (small K)
    (7, 2000, 68, 6),  # BASELINE.json configs[4] (SURVEY "X"): NODE=31, LEAF=32, W = 243.7 MB
]

CFLAGS = ["-std=c99", "-O2", "-fPIC", "-shared", "-w"]


def oracle_path():
    return os.path.join(HERE, "libjda_oracle.so")


def ref_path(T, K, L, D):
    return os.path.join(HERE, "_ref", "libjda_ref_%d_%d_%d_%d.so" % (T, K, L, D))


def build_oracle(force=False):
    out, src = oracle_path(), os.path.join(HERE, "jda_oracle.c")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-ffp-contract=off", "-fPIC", "-shared",
                           "-Wall", "-o", out, src, "-lm"])
    return out


def reference_available():
    return os.path.isfile(REF_SRC)


def build_ref(T, K, L, D, force=False):
    """Compile the reference TU for one dimension set. Needs /root/reference."""
    out = ref_path(T, K, L, D)
    if not force and os.path.exists(out):
        return out
    if not reference_available():
        raise RuntimeError("reference sources not present at %s" % REF_SRC)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(REF_SRC, "r") as f:
        tu = f.read()
    subs = {"JDA_T": T, "JDA_K": K, "JDA_LANDMARK_N": L, "JDA_TREE_DEPTH": D}
    for name, val in subs.items():
        tu, n = re.subn(r"(?m)^#define %s[ \t]+\d+[ \t]*$" % name, "#define %s %d" % (name, val), tu)
        if n != 1:
            raise RuntimeError("could not rewrite #define %s in the reference TU" % name)
    with open(os.path.join(HERE, "ref_harness.c"), "r") as f:
        tu += "\n" + f.read()
    # the rewritten TU only ever exists on gcc's stdin
    subprocess.run(["gcc"] + CFLAGS + ["-I", os.path.dirname(REF_SRC), "-x", "c", "-", "-o", out, "-lm"],
                   input=tu.encode(), check=True)
    return out


def build_all(force=False):
    built = [build_oracle(force)]
    if reference_available():
        for d in REF_DIMS:
            built.append(build_ref(*d, force=force))
    return built


if __name__ == "__main__":
    for p in build_all(force="--force" in sys.argv):
        print(p)
