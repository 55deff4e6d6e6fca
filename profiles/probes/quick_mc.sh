#!/bin/bash
# profiles/probes/quick_mc.sh [env assignments...] -- MC-only timing of the bench workload (2 timed steps), on the GPU box
for kv in "$@"; do export "$kv"; done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-det --steps 2 --warmup 1 > gpurun_out/quick.json 2> gpurun_out/quick.err || { tail -5 gpurun_out/quick.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/quick.json')); print('mc_ms', d['kernels']['walk_mc_ms_per_step'], 'evals/s', d['kernels']['walk_mc_evals_per_s'])"
