#!/bin/bash
# Developer loop on the GPU box: the bench's main pass for several (streams, hardware queues) pairs.
for cfg in "16 16" "24 16" "32 16" "32 24" "48 24" "32 32"; do set -- $cfg
timeout -k 10 200 python3 bench.py --streams $1 --hw-queues $2 --no-cpu-baseline --no-square --no-host-pipeline --no-isolated-kernel > gpurun_out/sw_$1_$2.log 2>&1 || exit 1
python3 -c "
import json,sys
l=[x for x in open('gpurun_out/sw_$1_$2.log') if x.startswith('{')][-1]; d=json.loads(l); print('streams $1 queues $2', d['value'], d['encode_mp_s'], d['decode_mp_s'])"
done
