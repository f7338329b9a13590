"""Instruction mix of each kernel in a hipcc -S --cuda-device-only assembly file."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
pat = re.compile(r'^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:', re.S | re.M)
for m in pat.finditer(txt):
    name, body = m.group(1), m.group(2)
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    ops = collections.Counter()
    for line in body.split('\n'):
        line = line.strip()
        if not line or line[0] in '.;/' or line.endswith(':'):
            continue
        ops[line.split()[0]] += 1
    g = collections.Counter()
    for op, c in ops.items():
        if op.startswith('v_mfma'):
            g['mfma'] += c
        elif op.startswith('v_') and 'f64' in op:
            g['valu_f64'] += c
        elif op.startswith('v_'):
            g['valu_other'] += c
        elif op.startswith('ds_'):
            g['lds'] += c
        elif op.startswith('s_waitcnt'):
            g['waitcnt'] += c
        elif op.startswith('s_'):
            g['salu'] += c
        elif op.startswith(('global_', 'scratch_', 'buffer_', 'flat_')):
            g['vmem'] += c
        else:
            g['other'] += c
    print(name[:90], 'total', sum(ops.values()))
    print('  ', dict(g))
    print('  ', ops.most_common(18))
