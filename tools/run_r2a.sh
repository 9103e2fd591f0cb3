mkdir -p gpurun_out/r2a
python -m pytest tests -m gpu -q -x --timeout=900 > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" > gpurun_out/r2a/rc.txt
tail -5 gpurun_out/r2a/pytest.log
for w in "tox21_c2 256" "tox21_c2 1024" "hiv_c3 1024" "lipo_c4 512" "c5_synth 1024"; do set -- $w; python bench.py --workload $1 --batch $2 --steps 30 --warmup 8 --no-cpu-baseline > gpurun_out/r2a/bench_$1_$2.json 2> gpurun_out/r2a/bench_$1_$2.err; echo "$1 $2 rc=$?"; done
