#!/bin/bash
# r06_w: the GPU suite with the HOST side of libjda.so under AddressSanitizer + UBSan (tools/asan_build.py: g++ and its runtime).
# -s: pytest's capture would swallow the sanitizers' reports (they are written to fd 2 of a process that then dies)
mkdir -p gpurun_out/r06_w
RT="$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libstdc++.so.6)"
export JDA_LIB_PATH=$PWD/jda_amd/libjda_asan.so
export ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1
export PYTHONUNBUFFERED=1
# (the sanitizer's dlopen interceptor loses torch's RUNPATH: its lazily loaded libraries are found through this)
export LD_LIBRARY_PATH=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),\"lib\"))"):$LD_LIBRARY_PATH
LD_PRELOAD=$RT timeout 2400 python -m pytest ${FILES:-tests/test_abi.py tests/test_ragged.py tests/test_cpp_entries.py tests/test_device_post.py tests/test_reentrant.py tests/test_fddb.py tests/test_fuzz_calls.py tests/test_scan_persistent.py} -s -q -m gpu -p no:cacheprovider > gpurun_out/r06_w/suite_full.txt 2>&1; echo "rc=$?" >> gpurun_out/r06_w/suite_full.txt
grep -v "python3.10\|libffi\|_ctypes" gpurun_out/r06_w/suite_full.txt | grep -n "runtime error\|AddressSanitizer\|passed\|failed\|rc=" -A 6 | head -150 > gpurun_out/r06_w/suite.txt
cat gpurun_out/r06_w/suite.txt
