"""Device table layouts as pure index arithmetic, checked on the host (jda_amd/csrc/kernels.h: lm_index, lm_deep_index):
every node of every cart lands on its own record, a path's last levels inside one group."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_level_major_and_grouped_node_layouts_are_bijections(tmp_path):
    exe = str(tmp_path / "lm_deep_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-I" + os.path.join(ROOT, "jda_amd", "csrc"), os.path.join(ROOT, "tests", "cpp", "lm_deep_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip() == "0", out.stdout + out.stderr
