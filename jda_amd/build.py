"""Builds jda_amd/libjda.so (the C-ABI shared library) with hipcc for gfx950.

The library is kept IN-TREE next to this file so it travels to the GPU box
with the repo snapshot.  No torch, no pybind: plain `hipcc -shared`.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libjda.so")
SOURCES = ["k_misc.hip", "k_scan.hip", "k_scan_d.hip", "k_scan_r.hip", "k_scan_dr.hip", "k_scan_p.hip", "k_finish.hip", "k_wide.hip", "k_stage.hip", "k_post.hip",
           "abi.cpp", "pass.cpp", "detect.cpp", "detect_cpp.cpp", "tickets.cpp", "ragged.cpp", "post_host.cpp", "lanes.cpp", "plans.cpp", "model_dev.cpp",
           "model.cpp", "plan.cpp", "post.cpp"]
HEADERS = ["kernels.h", "kernels_common.h", "finish_common.h", "scan_walk.h", "k_scan_impl.h", "model.h", "plan.h", "post.h", "host.h", "pass.h", "run.h", "detect.h", os.path.join("..", "..", "include", "jda.h")]
OBJDIR = os.path.join(HERE, "build")

# -ffp-contract=off: the cascade must round like the reference's scalar code
# (an FMA changes tree paths); denormals are kept; division is IEEE.
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
          "-fno-gpu-flush-denormals-to-zero", "-fvisibility=hidden", "-DJDA_EXPORTS",
          "-Wall", "-Wno-unused-function"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _deps_mtime():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS + [os.path.abspath(__file__)])


def stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def _build(lib, objdir, extra, force=False, verbose=False):
    """One object per translation unit, compiled in parallel (only the stale ones), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(objdir, exist_ok=True)
    hdr_t = _deps_mtime()
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".", "_") + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([hipcc()] + CFLAGS + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed: %s\n%s" % (" ".join(cmd), r.stderr[-8000:]))
        if r.stderr.strip() and verbose:
            print(r.stderr[-4000:])
    with ThreadPoolExecutor(max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(run, jobs))
    run([hipcc()] + LDFLAGS + ["-o", lib] + objs)
    return lib


DIST_LIB = os.path.join(HERE, "libjda_dist.so")


def build_dist(force=False):
    """libjda_dist.so: the RCCL gather of detection rows (include/jda_dist.h), its own library so that
    libjda.so depends on the HIP runtime only."""
    src = os.path.join(CSRC, "dist.cpp")
    deps = [src, os.path.join(HERE, "..", "include", "jda_dist.h"), os.path.join(HERE, "..", "include", "jda.h")]
    if not force and os.path.exists(DIST_LIB) and all(os.path.getmtime(DIST_LIB) >= os.path.getmtime(d) for d in deps):
        return DIST_LIB
    subprocess.check_call([hipcc(), "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-DJDA_EXPORTS", "-Wall",
                           "-o", DIST_LIB, src, "-L/opt/rocm/lib", "-lrccl", "-Wl,-rpath,/opt/rocm/lib"])
    return DIST_LIB


def build_timing():
    """Investigation build with shader-clock stamps in k_scan (tools/scan_timing.py); never the product."""
    return _build(os.path.join(HERE, "libjda_timing.so"), OBJDIR + "_timing", ["-DJDA_SCAN_TIMING"])


def build_variant(name, defines):
    """Experiment build with extra -D flags (A/B on one box through JDA_LIB_PATH); never the product."""
    return _build(os.path.join(HERE, "libjda_%s.so" % name), OBJDIR + "_" + name, ["-D" + d for d in defines])


WD_LIB = os.path.join(HERE, "libjda_wd.so")


def build_watchdog(force=False):
    """libjda_wd.so: the product's objects with ONE translation unit rebuilt -- k_scan_p.hip with its idle watchdog at
    zero (-DJDA_SCAN_P_IDLE_MAX=0), so that every launch of the persistent scan trips it.  Never the product: the GPU
    suite loads it through JDA_LIB_PATH to see that a tripped launch is noticed and the pass rerun with k_scan
    (tests/test_scan_persistent.py).  Built here, in-tree, so that it travels to the GPU box with libjda.so."""
    build()
    src = os.path.join(CSRC, "k_scan_p.hip")
    objdir = OBJDIR + "_wd"
    os.makedirs(objdir, exist_ok=True)
    obj = os.path.join(objdir, "k_scan_p_hip.o")
    main_objs = [os.path.join(OBJDIR, s.replace(".", "_") + ".o") for s in SOURCES]
    newest = max([os.path.getmtime(o) for o in main_objs] + [os.path.getmtime(src), _deps_mtime()])
    if not force and os.path.exists(WD_LIB) and os.path.getmtime(WD_LIB) >= newest:
        return WD_LIB
    subprocess.check_call([hipcc()] + CFLAGS + ["-DJDA_SCAN_P_IDLE_MAX=0", "-c", src, "-o", obj])
    objs = [obj if o.endswith("k_scan_p_hip.o") else o for o in main_objs]
    subprocess.check_call([hipcc()] + LDFLAGS + ["-o", WD_LIB] + objs)
    return WD_LIB


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    return _build(LIB, OBJDIR, [], force, verbose)


if __name__ == "__main__":
    if "--variant" in sys.argv:      # python -m jda_amd.build --variant wpe5 JDA_SCAN_P_WAVES_PER_EU=5
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:])); sys.exit(0)
    if "--bounds" in sys.argv:      # libjda_bounds.so: every model- / plan-derived device access checked against its extent (kernels_common.h: Bc)
        print(_build(os.path.join(HERE, "libjda_bounds.so"), OBJDIR + "_bounds", ["-DJDA_BOUNDS_CHECK"])); sys.exit(0)
    if "--exp" in sys.argv:         # python -m jda_amd.build --exp NAME -DFLAG ...: an experiment build libjda_NAME.so (never the product)
        i = sys.argv.index("--exp")
        print(_build(os.path.join(HERE, "libjda_%s.so" % sys.argv[i + 1]), OBJDIR + "_" + sys.argv[i + 1], sys.argv[i + 2:])); sys.exit(0)
    if "--timing" in sys.argv:
        print(build_timing()); sys.exit(0)
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_dist(force="--force" in sys.argv))
    print(build_watchdog(force="--force" in sys.argv))
