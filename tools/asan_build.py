"""libjda_asan.so: the host translation units (.cpp) of libjda.so rebuilt with AddressSanitizer + UBSan, linked with the
product's kernel objects.  Never the product: `JDA_LIB_PATH=jda_amd/libjda_asan.so LD_PRELOAD=<libclang_rt.asan> pytest ...`
runs the suites with every host-side allocation, table and staging buffer checked (tools/sessions/r05_w.sh)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jda_amd import build as B  # noqa: E402

# g++ and ITS sanitizer runtime: the one that ships with ROCm's clang intercepts the HSA allocator (it is built for
# GPU-side checking with xnack) and aborts inside the HIP runtime's first allocation on this box.
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g"]      # UBSan reports and goes on (every finding of a run), ASan halts
GXX_FLAGS = ["-std=c++17", "-O1", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-DJDA_EXPORTS", "-D__HIP_PLATFORM_AMD__",
             "-I/opt/rocm/include", "-Wall", "-Wno-unused-function"]


def main():
    global SAN
    name = "asan"
    if "--tsan" in sys.argv:          # libjda_tsan.so: ThreadSanitizer instead (lanes, tickets, the uploader and post-processing threads)
        SAN = ["-fsanitize=thread", "-fno-omit-frame-pointer", "-g"]
        name = "tsan"
    B.build()
    objdir = B.OBJDIR + "_" + name
    os.makedirs(objdir, exist_ok=True)
    flags = GXX_FLAGS + SAN
    objs, jobs = [], []
    for s in B.SOURCES:
        main_obj = os.path.join(B.OBJDIR, s.replace(".", "_") + ".o")
        if s.endswith(".hip"):
            objs.append(main_obj)
            continue
        obj = os.path.join(objdir, s.replace(".", "_") + ".o")
        objs.append(obj)
        jobs.append(["g++"] + flags + ["-c", os.path.join(B.CSRC, s), "-o", obj])
    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(" ".join(cmd) + "\n" + r.stderr[-4000:])
    with ThreadPoolExecutor(8) as ex:
        list(ex.map(run, jobs))
    lib = os.path.join(B.HERE, "libjda_%s.so" % name)
    run(["g++", "-shared", "-fPIC"] + SAN + ["-o", lib] + objs + ["-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
    rt = subprocess.run(["g++", "-print-file-name=lib%s.so" % name], capture_output=True, text=True).stdout.strip()
    print(lib)
    print(rt)


if __name__ == "__main__":
    main()
