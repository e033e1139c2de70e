"""Instruction-class census of the kernels in a hipcc -S listing (dev tool).
usage: isa_count.py file.s [substring ...]"""
import re
import sys
from collections import Counter

TRANS = {'v_exp_f32', 'v_sin_f32', 'v_cos_f32', 'v_log_f32', 'v_sqrt_f32', 'v_rcp_f32', 'v_rsq_f32'}


def classify(i):
    if i.startswith('v_pk_'): return 'v_pk'
    if i in TRANS: return 'trans'
    if i.startswith('v_mad_u64'): return 'mad64'
    if i.startswith('v_readlane') or i.startswith('v_writelane') or i.startswith('v_readfirstlane'): return 'lane_rw'
    if i.startswith('v_mov') or i.startswith('v_accvgpr'): return 'v_mov'
    if i.startswith('v_cndmask'): return 'v_cndmask'
    if i.startswith('v_'): return 'v_other'
    if i.startswith('s_waitcnt'): return 's_waitcnt'
    if i.startswith('s_'): return 'salu'
    if i.startswith('ds_'): return 'ds'
    if i.startswith('scratch_'): return 'scratch'
    if i.startswith('global_') or i.startswith('buffer_') or i.startswith('flat_'): return 'vmem'
    return 'other'


def main():
    lines = open(sys.argv[1]).read().split('\n')
    pats = sys.argv[2:]
    cur, body = None, []
    for l in lines:
        m = re.match(r'^(_Z\w+):', l)
        if m:
            cur, body = m.group(1), []
            continue
        if cur and l.startswith('.Lfunc_end'):
            if not pats or any(p in cur for p in pats):
                ins = [x.strip().split()[0] for x in body if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
                c = Counter(classify(i) for i in ins)
                dpp = sum(1 for x in body if 'dpp' in x or 'row_' in x)
                print(cur[:90], len(ins), dict(sorted(c.items())), 'dpp', dpp)
            cur = None
            continue
        if cur:
            body.append(l)


main()
