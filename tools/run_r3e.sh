# round 3, call E: kernel traces of the new backward / fused-finalize kernels vs the old ones (B=256)
for cfg in "0 0" "1 1"; do
  set -- $cfg
  EAGCN_MOLBWD=$1 EAGCN_BN_FUSED=$2 bash tools/run_prof.sh r3e_m$1f$2 --steps 20 --warmup 5 > gpurun_out/r3e_m$1f$2.txt 2>&1
done
for t in m0f0 m1f1; do echo "== $t"; grep -E "agg|bn_|mol_|unpack|^# " gpurun_out/prof_r3e_$t/summary.txt | cut -c1-60,112-200 | head -24; done
