#!/bin/bash
# Dry run of bench.py's multi-rank path on ONE GPU: two ranks, both on device 0, the record exchange over gloo (no
# second device, no RCCL between ranks) -- chunk partition, exchange on the second stream, cross-rank boundary match and
# its parity gate, for both exchange forms.  The first real N > 1 run then only adds RCCL between devices.
# Usage (GPU box): bash tools/scale_dryrun.sh [extra bench args]
cd "$(dirname "$0")/.."
rc=0
for ex in allgather boundary; do
  port=$((29500 + RANDOM % 2000))
  out=$(python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port \
        bench.py --gpus 2 --one-device --dist-backend gloo --exchange $ex --batch 32 --steps 4 --warmup 3 \
        --cpu-seconds 1 --no-stage-timing "$@" 2>gpurun_out/scale_dryrun_$ex.err | tail -1)
  echo "$out" | python -c "
import json, sys
d = json.loads(sys.stdin.read())
ok = d['n_gpus'] == 2 and d['parity']['bit_exact_vs_oracle'] is True and d['config']['exchange'] == '$ex' and d['cpu_baseline']
print('$ex', 'ok' if ok else 'FAILED', d['value'], d['config']['exchange_bytes_in_per_gpu_per_step'])
sys.exit(0 if ok else 1)" || rc=1
done
exit $rc
