#!/bin/bash
# Developer loop on the GPU box: the bench's main pass for several streams:hardware-queues pairs
# (KHG_SWEEP="24:16 20:16 ..." overrides the list).
for cfg in ${KHG_SWEEP:-16:16 24:16 32:16 32:24 48:24 32:32}; do
s=${cfg%%:*}; q=${cfg##*:}
timeout -k 10 200 python3 bench.py --streams $s --hw-queues $q --no-cpu-baseline --no-square --no-host-pipeline --no-isolated-kernel > gpurun_out/sw_${s}_$q.log 2>&1 || exit 1
python3 -c "
import json
l=[x for x in open('gpurun_out/sw_${s}_$q.log') if x.startswith('{')][-1]; d=json.loads(l); print('streams $s queues $q', d['value'], d['encode_mp_s'], d['decode_mp_s'])"
done
