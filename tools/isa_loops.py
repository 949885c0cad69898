"""Per-loop instruction mix of one kernel in a hipcc -S listing (loops = backward branches to .LBB labels).
usage: python tools/isa_loops.py <file.s> <mangled kernel name prefix> [min instructions]"""
import re, sys
txt = open(sys.argv[1]).read().split('\n')
name = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 150
a = next(i for i, l in enumerate(txt) if l.startswith(name) and l.rstrip().split(';')[0].rstrip().endswith(':'))
b = next(i for i in range(a, len(txt)) if txt[i].startswith('.Lfunc_end'))
lines = txt[a:b]
labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
loops = []
for i, l in enumerate(lines):
    m = re.search(r'\b(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', l)
    if m and m.group(2) in labels and labels[m.group(2)] < i:
        loops.append((labels[m.group(2)], i, m.group(2)))
keys = ['v_pk_fma_f32', 'v_mfma_f32_16x16x4_f32', 'v_readlane_b32', 'v_writelane_b32', 's_load_dwordx16', 'ds_read2_b32', 'ds_read_b128',
        'ds_write_b64', 'ds_write_b128', 'global_load_dwordx4', 'global_store_dwordx4', 'scratch_load_dword', 'scratch_store_dword',
        'scratch_load_dwordx2', 'scratch_load_dwordx4', 's_waitcnt', 'v_mov_b32', 's_nop']
seen = set()
for s, e, lab in sorted(loops):
    if lab in seen: continue
    seen.add(lab)
    e = max(x[1] for x in loops if x[2] == lab)
    c = {}
    for l in lines[s:e + 1]:
        t = l.split()
        if t and re.match(r'^(v_|s_|ds_|global_|scratch_|buffer_)', t[0]):
            op = re.sub(r'_e32$|_e64$', '', t[0]); c[op] = c.get(op, 0) + 1
    n = sum(c.values())
    if n < minn: continue
    valu = sum(v for k, v in c.items() if k.startswith('v_') and 'mfma' not in k)
    print(f"loop {lab} lines {s}-{e} instrs {n} valu {valu} | " + ' '.join(f"{k}={c[k]}" for k in keys if c.get(k)))
