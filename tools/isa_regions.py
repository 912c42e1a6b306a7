#!/usr/bin/env python3
"""Developer tool: per-kernel summary of a hipcc -S listing — instruction count, MFMA count, and where the scratch (spill)
instructions sit relative to the MFMA stream (so a spill can be attributed to a stage / job).
usage: isa_regions.py file.s [kernel-substring]"""
import sys
lines = open(sys.argv[1]).read().split('\n')
want = sys.argv[2] if len(sys.argv) > 2 else ''
starts = [i for i, l in enumerate(lines) if l.startswith('_Z') and ': ' in l and '@' in l]
for si, i0 in enumerate(starts):
    name = lines[i0].split(':')[0]
    if want not in name:
        continue
    i1 = starts[si + 1] if si + 1 < len(starts) else len(lines)
    seg = lines[i0:i1]
    ins = [l for l in seg if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;')]
    m16 = m4 = 0
    ev = []
    last_label = ''
    for l in seg:
        if l and not l.startswith('\t') and l.endswith(':') or (l.startswith('.LBB') ):
            last_label = l.split(':')[0]
        if 'v_mfma_f32_16x16x4' in l:
            m16 += 1
        elif 'v_mfma_f32_4x4x1' in l:
            m4 += 1
        elif 'scratch_' in l:
            ev.append((last_label, m16, m4, l.strip().split()[0]))
    print(f'{name}: {len(ins)} instructions, mfma16 {m16}, mfma4 {m4}, scratch ops {len(ev)}')
    agg = {}
    for lab, a, b, op in ev:
        key = (a // 64 * 64, b // 16 * 16)
        agg.setdefault(key, [0, 0])
        agg[key][0 if 'store' in op else 1] += 1
    for key in sorted(agg):
        print(f'   after mfma16 >= {key[0]:4d}, mfma4 >= {key[1]:4d}: {agg[key][0]} stores, {agg[key][1]} loads')

# ---- loops: a backward branch to a label closes a loop; report scratch traffic inside loops that hold MFMAs ----
def loops_report(seg, base):
    import re
    lab = {}
    for i, l in enumerate(seg):
        m = re.match(r'^(\.LBB[0-9_]+):', l)
        if m:
            lab[m.group(1)] = i
    loops = []
    for i, l in enumerate(seg):
        m = re.match(r'^\ts_c?branch\S*\s+(\.LBB[0-9_]+)', l)
        if m and m.group(1) in lab and lab[m.group(1)] < i:
            loops.append((lab[m.group(1)], i))
    for a, b in loops:
        body = seg[a:b]
        nm = sum(1 for l in body if 'v_mfma' in l)
        if nm == 0:
            continue
        ns = [l.strip() for l in body if 'scratch_' in l]
        ni = sum(1 for l in body if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'))
        nv = sum(1 for l in body if l.startswith('\tv_') and 'v_mfma' not in l)
        nl = sum(1 for l in body if l.startswith('\tds_'))
        nb = sum(1 for l in body if l.startswith('\tbuffer_'))
        nsalu = sum(1 for l in body if l.startswith('\ts_') and not l.startswith('\ts_waitcnt') and not l.startswith('\ts_nop'))
        nw = sum(1 for l in body if l.startswith('\ts_waitcnt'))
        nn = sum(1 for l in body if l.startswith('\ts_nop'))
        print(f'   loop lines {base + a}-{base + b}: {ni} instr, {nm} mfma, {nv} valu, {nl} ds, {nb} buffer, {nsalu} salu, {nw} waitcnt, {nn} nop, scratch {len(ns)}')
        for s_ in ns:
            print('        ', s_[:70])

for si, i0 in enumerate(starts):
    name = lines[i0].split(':')[0]
    if want not in name:
        continue
    i1 = starts[si + 1] if si + 1 < len(starts) else len(lines)
    loops_report(lines[i0:i1], i0 + 1)
