for c in 2000 500 125 32 1; do
  TLM_WALK_CHUNK=$c timeout -k 10 300 python bench.py --no-cpu-baseline --no-det --steps 2 --warmup 1 > gpurun_out/sweep_$c.json 2> gpurun_out/sweep_$c.err
  python -c "
import json; d=json.load(open('gpurun_out/sweep_$c.json')); print($c, d['kernels']['walk_mc_ms_per_step'])"
done
