#!/usr/bin/env python3
"""Where a kernel's MFMA loops sit in its code object: python tools/loop_places.py <llvm-objdump -d listing> <mangled kernel name>
(address of each backward branch target that encloses MFMAs, modulo 64 = its offset inside an instruction-cache line)."""
import sys, re
t = sys.argv[1]
L = open(t).read().split('\n')
name = sys.argv[2]
start = [i for i, l in enumerate(L) if l.endswith(f'<{name}>:')][0]
end = next((i for i in range(start + 1, len(L)) if re.match(r'^[0-9a-f]+ <_Z', L[i])), len(L))
body = L[start:end]
def addr(l):
    m = re.search(r'//\s*([0-9A-Fa-f]+):', l)
    return int(m.group(1), 16) if m else None
a0 = addr(body[1])
mf = [i for i, l in enumerate(body) if 'v_mfma' in l]
last = max(a for a in map(addr, body) if a is not None)
print(t, 'kernel start %x' % a0, 'bytes', last - a0, 'mfma', len(mf))
for i, l in enumerate(body):
    m = re.search(r's_cbranch_\w+ (\d+)', l) or re.search(r's_branch (\d+)', l)
    if m:
        off = int(m.group(1))
        if off >= 32768:
            tgt = addr(l) + 4 + (off - 65536) * 4
            n = sum(1 for j in mf if addr(body[j]) and tgt <= addr(body[j]) <= addr(l))
            if n: print('   loop target %x (mod 64 = %d) branch at %x, bytes %d, mfma inside %d' % (tgt, tgt % 64, addr(l), addr(l) - tgt, n))
