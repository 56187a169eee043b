#!/bin/bash
# Developer loop on the GPU box: short bench of alternative builds (tools/bin/*.so).  Usage: tools/gpu_bench_variants.sh a.so b.so ...
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
for L in "$@"; do
  KHG_LIB=$R/tools/bin/$L timeout -k 10 200 python3 $R/tools/bench_variant.py 2>&1 | tail -1 || exit 1
done
