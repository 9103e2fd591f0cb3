mkdir -p gpurun_out/r2h
python -m pytest tests -m gpu -q --timeout=900 > gpurun_out/r2h/pytest.log 2>&1; echo "pytest rc=$?"
grep -E "^FAILED|^ERROR|passed|failed" gpurun_out/r2h/pytest.log | tail -15
for w in "tox21_c2 256" "tox21_c2 1024" "hiv_c3 1024"; do set -- $w; python bench.py --workload $1 --batch $2 --steps 30 --warmup 8 --repeats 5 --no-cpu-baseline --no-extras > gpurun_out/r2h/bench_$1_$2.json 2> gpurun_out/r2h/bench_$1_$2.err; echo "$1 $2 rc=$? $(python -c "import json;d=json.load(open('gpurun_out/r2h/bench_$1_$2.json'));print(d['value'],d['ms_per_step'],d['roofline']['achieved'],d['roofline']['step_frac'],d['kernel_ms_per_step'])")"; done
