mkdir -p gpurun_out/r3l
for cfg in "0 0" "1 0" "0 1" "1 1"; do
  set -- $cfg
  for B in 256 1024; do
    EAGCN_NO_GBN_FOLD=$1 EAGCN_NO_EDGE_DEFER=$2 timeout 300 python bench.py --no-cpu-baseline --no-extras --batch $B --repeats 9 --steps 40 > gpurun_out/r3l/b${B}_g$1e$2.json 2>/dev/null
    python - <<PY
import json
d=json.loads(open('gpurun_out/r3l/b${B}_g$1e$2.json').read().strip().splitlines()[-1])
print('nofold=$1 nodefer=$2 B=$B', d['ms_per_step'], d['value_min'], d['value_max'], d['kernel_ms_per_step']['bn'], d['kernel_ms_per_step']['head'], d['kernel_ms_per_step']['pack'])
PY
  done
done
