for v in "" "EAGCN_SIDE_PRIORITY=0" "EAGCN_SIDE_AFTER_FWD=1" "EAGCN_SIDE_PRIORITY=0 EAGCN_SIDE_AFTER_FWD=1"; do
  for B in 256 1024; do
  echo "[$v] B=$B $(env $v python bench.py --batch $B --steps 30 --warmup 8 --repeats 7 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'],d['value_min'],d['value_max'])")"
done; done
echo "compact: $(python bench.py --input compact --steps 30 --warmup 8 --repeats 7 --no-cpu-baseline --no-extras 2>/dev/null | python -c "import json,sys;d=json.load(sys.stdin);print(d['value'],d['ms_per_step'])")"
