"""usage: isa_mix.py file.s mangled-substring  -- instruction mix of the hottest (largest) loop of a kernel"""
import re, sys, collections
src = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(src) if l.startswith('_ZN') and sys.argv[2] in l and l.rstrip().split(':')[0].endswith(sys.argv[2].split('*')[-1]) or (l.startswith('_ZN') and sys.argv[2] in l))
end = next(i for i in range(start, len(src)) if src[i].strip().startswith('s_endpgm'))
labels = {}
for i in range(start, end):
    m = re.match(r'^(\.LBB\d+_\d+):', src[i])
    if m: labels[m.group(1)] = i
loops = []
for i in range(start, end):
    m = re.match(r'\s+s_c?branch\S*\s+(\.LBB\d+_\d+)', src[i])
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((i - labels[m.group(1)], labels[m.group(1)], i))
loops.sort(reverse=True)
def cls(op):
    if op.startswith('v_mfma'): return 'mfma'
    if 'dpp' in op: return 'dpp'
    if op.startswith('v_pk_'): return 'pk'
    if op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_sin', 'v_cos')): return 'trans'
    if op.startswith('v_cvt'): return 'cvt'
    if op.startswith(('v_and', 'v_or', 'v_xor', 'v_lshl', 'v_lshr', 'v_ashr', 'v_bfe', 'v_perm', 'v_alignb', 'v_bfi', 'v_not')): return 'bitops'
    if op.startswith(('v_mov', 'v_accvgpr')): return 'mov'
    if op.startswith('v_cndmask'): return 'cndmask'
    if op.startswith('v_cmp'): return 'cmp'
    if op.startswith(('v_mul_lo', 'v_mul_hi', 'v_mad_u', 'v_mad_i')): return 'imul'
    if op.startswith('v_'): return 'valu_other'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('s_waitcnt'): return 'wait'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_'): return 'salu'
    return 'other'
for n, a, b in loops[:int(sys.argv[3]) if len(sys.argv) > 3 else 1]:
    mix = collections.Counter()
    dppin = 0
    for l in src[a:b + 1]:
        t = l.strip().split()
        if not t or t[0].startswith(('.', ';')) or t[0].endswith(':'): continue
        c = cls(t[0])
        if c != 'dpp' and 'dpp' in l.split(';')[0]: c = 'dpp'
        mix[c] += 1
    tot = sum(mix.values())
    valu = sum(v for k, v in mix.items() if k in ('dpp', 'pk', 'trans', 'cvt', 'bitops', 'mov', 'cndmask', 'cmp', 'imul', 'valu_other'))
    print('loop lines %d-%d: %d instructions, %d VALU, %d MFMA' % (a, b, tot, valu, mix['mfma']))
    print('  ' + ', '.join('%s %d' % kv for kv in mix.most_common()))
