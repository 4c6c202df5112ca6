"""Scan hipcc's device assembly (-S) for compiler-generated uses of the fixed operand-ring registers
v[64:87] outside the inline-asm k-loops.  usage: check_ring_regs.py file.s"""
import re, sys
rx = re.compile(r'\bv(6[4-9]|7[0-9]|8[0-7])\b|v\[(6[4-9]|7[0-9]|8[0-7]):|v\[\d+:(6[4-9]|7[0-9]|8[0-7])\]')
cur, inasm, hits = None, False, {}
for l in open(sys.argv[1]):
    m = re.match(r'^(_Z\S+):', l)
    if m: cur = m.group(1)
    if 'ASMSTART' in l: inasm = True
    elif 'ASMEND' in l: inasm = False
    elif not inasm and cur and rx.search(l.split(';')[0]):
        hits.setdefault(cur, []).append(l.strip())
for k, v in hits.items():
    print(k, len(v)); [print('   ', x) for x in v[:6]]
print("kernels with outside uses:", len(hits))
