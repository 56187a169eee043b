#!/bin/bash
# The kernel bodies under AddressSanitizer: builds the test-only simulator with -fsanitize=address and no slack on
# its workspace arrays, and runs the simulator tests (encode, decode, tiles, plane entry points) against it, then
# 300 corrupted files through the decoder (tools/fuzz_sim.py).
# Host code only; catches out-of-bounds indexing in the kernel sources before they reach a GPU.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SO=/tmp/libkhg_sim_asan.so
g++ -std=c++17 -O1 -g -fsanitize=address -fno-omit-frame-pointer -ffp-contract=off -fPIC -shared -DKHG_SIM_SLACK=0 -DKHG_PS_BATCH_BITS=20000 -DKHG_CHAIN_GROUPS=2 \
    -I"$ROOT/image-compression_amd/csrc" -o $SO "$ROOT/tests/sim/khg_sim.cpp"
cat > /tmp/run_asan_suite.py <<PY
import sys
sys.path.insert(0, "$ROOT/tests")
import simlib as S
S.SO = "$SO"
S.build = lambda force=False: None
import pytest
sys.exit(pytest.main(["$ROOT/tests/test_sim_kernels.py", "$ROOT/tests/test_tiled.py", "$ROOT/tests/test_ordered_sum.py", "$ROOT/tests/test_index.py", "$ROOT/tests/test_visual.py", "-x", "-q", "-k", "not gloo and not every_rank and not gpu",
                      "-p", "no:cacheprovider"]))
PY
ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(g++ -print-file-name=libasan.so)${LD_PRELOAD:+:$LD_PRELOAD} python /tmp/run_asan_suite.py
KHG_SIM_SO=$SO ASAN_OPTIONS=detect_leaks=0 LD_PRELOAD=$(g++ -print-file-name=libasan.so)${LD_PRELOAD:+:$LD_PRELOAD} python "$ROOT/tools/fuzz_sim.py" 1 300
