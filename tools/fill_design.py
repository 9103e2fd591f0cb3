#!/usr/bin/env python3
"""Fill the R6_* placeholders of a DESIGN.md template from a bench.py JSON line: python tools/fill_design.py template.md bench.json > DESIGN.md"""
import json
import sys

tpl = open(sys.argv[1]).read()
line = [l for l in open(sys.argv[2]).read().splitlines() if l.startswith('{')][-1]
d = json.loads(line)
s = d['extra_summary']


def ms(k):
    return '%.3f' % s[k][0] if s[k][0] < 2 else '%.2f' % s[k][0]


def mols(k):
    v = s[k][1]
    return '%.2f M' % (v / 1e6) if v >= 1e6 else '%.0f k' % (v / 1e3)


rep = {
    'R6_C2_MS': ms('c2'), 'R6_C2_MOLS': mols('c2'), 'R6_C2_FP32': ms('c2_fp32_mfma'),
    'R6_B1024_MS': ms('b1024'), 'R6_B1024_MOLS': mols('b1024'), 'R6_B1024_FP32': ms('b1024_fp32_mfma'),
    'R6_HIV_MS': ms('hiv_c3'), 'R6_HIV_MOLS': mols('hiv_c3'), 'R6_HIV_FP32': ms('hiv_c3_fp32_mfma'),
    'R6_LIPO_MS': ms('lipo_c4'), 'R6_LIPO_MOLS': mols('lipo_c4'),
    'R6_C5_MS': ms('c5_synth'), 'R6_C5_MOLS': mols('c5_synth'), 'R6_C5_FP32': ms('c5_synth_fp32_mfma'),
    'R6_TRAIN_MS': ms('train_step'), 'R6_TRAIN_MOLS': mols('train_step'), 'R6_TRAIN_TORCH': ms('train_step_torch_adam'),
    'R6_C2_BF16_MOLS': mols('c2_bf16'), 'R6_C2_BF16': ms('c2_bf16'),
    'R6_CPU1024': '%.0f' % s['cpu_baseline_b1024'][0], 'R6_CPU0': '%.0f' % s['cpu_baseline_configs0'][0], 'R6_CPU': '%.0f' % s['cpu_baseline'][0],
    'R6_RATIO': '%d' % round(s['b1024_gpu_over_cpu'], -2),
    'R6_DOM_US': '%.1f' % s['dominant_kernel_us'], 'R6_DOM_FRAC': '%.2f' % s['roofline_frac'], 'R6_STEP_FRAC': '%.2f' % s['step_frac'],
}
for k in sorted(rep, key=len, reverse=True):
    tpl = tpl.replace(k, rep[k])
sys.stdout.write(tpl)
