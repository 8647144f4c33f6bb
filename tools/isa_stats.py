"""ISA summary of a kernel in a hipcc -S listing: python tools/isa_stats.py file.s <mangled-name-substring> ..."""
import re
import sys

s = open(sys.argv[1]).read()
for name in sys.argv[2:]:
    m = re.search(r'^(\w*' + re.escape(name) + r'\w*):', s, re.M)
    if not m:
        print(name, 'not found')
        continue
    k = s[m.start():s.index('s_endpgm', m.start())]
    lines = k.split('\n')
    mf = [n for n, l in enumerate(lines) if 'v_mfma' in l]
    acc = [n for n, l in enumerate(lines) if 'v_accvgpr' in l]
    print(m.group(1)[:60], 'lines', len(lines), 'mfma', len(mf), 'ds_read_b128', k.count('ds_read_b128'), 'scratch',
          k.count('scratch_'))
    print('  vmcnt(0) at', [n for n, l in enumerate(lines) if 'vmcnt(0)' in l])
    if mf:
        print('  mfma range', mf[0], mf[-1], 'accvgpr moves inside', len([n for n in acc if mf[0] <= n <= mf[-1]]),
              'total', len(acc))
