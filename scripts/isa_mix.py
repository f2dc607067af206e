#!/usr/bin/env python3
"""Dev helper: instruction mix of the innermost loop of each kernel in a hipcc -S output.
usage: isa_mix.py file.s [substring-filter]"""
import re, sys, collections
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
parts = re.split(r'\n(_Z[\w]+):[^\n]*\n', txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1].split('.Lfunc_end')[0]
    if flt not in name:
        continue
    lines = body.split('\n')
    labels = {}
    for n, l in enumerate(lines):
        m = re.match(r'(\.LBB\d+_\d+):', l)
        if m:
            labels[m.group(1)] = n
    best = None
    for n, l in enumerate(lines):
        m = re.search(r's_c?branch\w* (\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            span = (labels[m.group(1)], n)
            if best is None or span[1] - span[0] > best[1] - best[0]:
                best = span
    if not best:
        continue
    loop = lines[best[0]:best[1] + 1]
    c = collections.Counter()
    for l in loop:
        m = re.match(r'\s+([vs]_[a-z0-9_]+|global_\w+|ds_\w+|buffer_\w+|scratch_\w+)', l)
        if m:
            c[m.group(1)] += 1
    g = collections.Counter()
    for k, v in c.items():
        if k.startswith('v_') and 'f64' in k:
            g['valu_f64'] += v
        elif k.startswith('v_'):
            g['valu_other'] += v
        elif k.startswith('s_'):
            g['salu'] += v
        else:
            g['mem'] += v
    m = re.search(r'\.vgpr_count:\s+(\d+)', parts[i + 1])
    print(name)
    print('  loop insts', sum(c.values()), dict(g))
    print('  ', sorted(c.items(), key=lambda kv: -kv[1])[:30])
