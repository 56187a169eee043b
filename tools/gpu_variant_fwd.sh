#!/bin/bash
# Developer loop on the GPU box: time the level-1 forward launch of alternative builds (tools/bin/*.so).
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
for L in "$@"; do
  echo -n "$L: "
  KHG_LIB=$R/tools/bin/$L timeout -k 10 120 python3 $R/tools/probe_fwd.py 2>&1 | tail -1 || exit 1
done
