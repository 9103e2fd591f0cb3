"""The ISA the BASELINE configs actually run carries no scratch traffic in its hot loops (spills had come back
into the MFMA loops of agg_edge<9,false>, agg_wave<9,*> and bx3_kernel<3,0> unnoticed).  hipcc cross-compiles gfx950 here, no GPU
needed: every dispatched instantiation of the aggregation kernels must have private_segment_fixed_size == 0, and no basic block of
the plane GEMMs that issues matrix instructions may touch scratch (bx3_kernel<3,0> keeps a few one-time spills of its loader
prologue: 256 registers are shared by two roles)."""
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'eagcn_amd', 'csrc')
HIPCC = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-function', '-munsafe-fp-atomics', '-S', '--cuda-device-only']


def _isa(tmp, name):
    out = os.path.join(tmp, name + '.s')
    r = subprocess.run([HIPCC] + FLAGS + ['-o', out, os.path.join(CSRC, name + '.hip')], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return open(out).read()


@pytest.fixture(scope='module')
def isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip('hipcc not found')
    tmp = str(tmp_path_factory.mktemp('isa'))
    names = ['agg', 'lagg', 'gemm_bx3', 'gemm_bx3w', 'head2']
    with ThreadPoolExecutor(max_workers=5) as ex:
        return dict(zip(names, ex.map(lambda n: _isa(tmp, n), names)))


def _kernels(text):
    """{mangled name: (private_segment_fixed_size, vgpr_count)} from the amdhsa.kernels metadata"""
    out = {}
    for m in re.finditer(r'- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)', text, re.S):
        blk = m.group(0)
        out[re.search(r'\.name:\s+(\S+)', blk).group(1)] = (int(re.search(r'\.private_segment_fixed_size:\s+(\d+)', blk).group(1)),
                                                            int(re.search(r'\.vgpr_count:\s+(\d+)', blk).group(1)))
    return out


def _lds_bytes(text):
    """{mangled name: group_segment_fixed_size}"""
    return {re.search(r'\.name:\s+(\S+)', m.group(0)).group(1): int(re.search(r'\.group_segment_fixed_size:\s+(\d+)', m.group(0)).group(1))
            for m in re.finditer(r'- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)', text, re.S)}


def test_dispatched_aggregation_kernels_have_no_scratch(isa):
    ks = _kernels(isa['agg'])
    seen = 0
    for name, (scratch, vgpr) in ks.items():
        m = re.search(r'(agg_wave_kernel|agg_kernel|agg2_kernel|agg_edge_kernel)ILi(\d+)ELb([01])E', name)
        if not m:
            continue
        kind, ct, flag = m.group(1), int(m.group(2)), m.group(3) == '1'
        # what launch_agg / launch_agg_edge dispatch: at most 9 column tiles per workgroup, and an 8-tile layer runs the 9-tile code
        # (agg.hip agg_pick_ct); the K-split forward kernels (agg_kernel / agg2_kernel<CT, false>) likewise
        if ct > 9 or ct == 8:
            continue
        seen += 1
        assert scratch == 0, '%s spills %d bytes per lane (%d VGPRs)' % (name, scratch, vgpr)
    assert seen >= 40
    for name, (scratch, vgpr) in _kernels(isa['lagg']).items():
        if 'lagg_kernel' in name:
            assert scratch == 0 and vgpr <= 168, (name, scratch, vgpr)       # three waves per SIMD (LDS allows three workgroups per CU)
    lds = {n: b for n, b in _lds_bytes(isa['lagg']).items() if 'lagg_kernel' in n}
    assert len(lds) == 5                      # forward / transposed x one / several chunks, + the Weighted_sum staging (several chunks only)
    for name, b in lds.items():
        assert 3 * b <= 160 * 1024, (name, b)                                # ... and three workgroups' LDS fit the CU's 160 KB


def _mfma_blocks_with_scratch(text, kernel):
    lines = text.split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^[A-Za-z_][\w$.]*:', l) and kernel in l)
    end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
    bad, cur, label = [], [], 'entry'
    for l in lines[start + 1:end + 1] + ['.LBB0_0:']:
        m = re.match(r'^(\.LBB\d+_\d+):', l)
        if m:
            if any('v_mfma' in x for x in cur) and any(x.strip().startswith('scratch_') for x in cur):
                bad.append(label)
            label, cur = m.group(1), []
        else:
            cur.append(l)
    return bad


def test_plane_gemm_loops_do_not_touch_scratch(isa):
    for src, kernel in (('gemm_bx3', 'bx3_kernelILi3ELi0E'), ('gemm_bx3', 'bx3_kernelILi1ELi0E'), ('gemm_bx3w', 'bx3w_kernelILi3ELi4ELi0E'),
                        ('gemm_bx3w', 'bx3w_kernelILi1ELi4ELi0E')):
        assert _mfma_blocks_with_scratch(isa[src], kernel) == [], (kernel, 'scratch access inside a block that issues MFMAs')
    ks = _kernels(isa['gemm_bx3w'])
    for name, (scratch, vgpr) in ks.items():
        if 'bx3w_kernelILi3ELi4ELi0E' in name or 'bx3w_kernelILi1ELi4ELi0E' in name:
            assert scratch == 0 and vgpr <= 256, (name, scratch, vgpr)            # two compute waves per SIMD


def test_head_kernels_have_no_scratch(isa):
    """csrc/head2.hip (round 6: eight waves per workgroup, the first operand batch requested before the BatchNorm table is built -- more
    registers live across the table build): every instantiation of the head's launches, the fused middle launch included, keeps its
    operand batches in registers (six k-steps of the 700-wide first product are 120 of them); 256 registers = two waves per SIMD."""
    ks = _kernels(isa['head2'])
    seen = 0
    for name, (scratch, vgpr) in ks.items():
        if not re.search(r'head_(fwd|bwd|mid|bwd_pair|gbn_bwd)_kernel', name):
            continue
        seen += 1
        assert scratch == 0 and vgpr <= 256, '%s: %d bytes of scratch per lane, %d VGPRs' % (name, scratch, vgpr)
    assert seen >= 20
    # round 6: dense 1's backward at large batches with one k-step in flight: three 4-wave workgroups per CU (170 registers)
    light = {n: v for n, v in ks.items() if 'head_bwd_light_kernel' in n}
    assert len(light) == 2
    for name, (scratch, vgpr) in light.items():
        assert scratch == 0 and vgpr <= 168, (name, scratch, vgpr)
