"""Static census of a kernel's ISA between barriers: spill traffic, MFMA, VALU, LDS, waits per barrier interval.
usage: python isa_census.py file.s mangled_kernel_name_substring"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and key in l and l.rstrip().endswith(':') or (l.startswith('_Z') and key in l and '; @' in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
body = lines[start:end]
iv = []
cur = dict(n=0, mfma=0, valu=0, sst=0, sld=0, ds=0, buf=0, vm0=0, wl=0)
tot = dict(cur)
for l in body:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.') or t.endswith(':'):
        continue
    op = t.split()[0]
    cur['n'] += 1
    if op.startswith('v_mfma'): cur['mfma'] += 1
    elif op.startswith('v_readlane') or op.startswith('v_writelane'): cur['wl'] += 1
    elif op.startswith('v_'): cur['valu'] += 1
    elif op.startswith('scratch_store'): cur['sst'] += 1
    elif op.startswith('scratch_load'): cur['sld'] += 1
    elif op.startswith('ds_'): cur['ds'] += 1
    elif op.startswith('buffer_'): cur['buf'] += 1
    elif op == 's_waitcnt' and 'vmcnt(0)' in t: cur['vm0'] += 1
    if op == 's_barrier':
        iv.append(cur); cur = dict.fromkeys(cur, 0)
iv.append(cur)
print('interval  instr  mfma  valu  lane  sc_st  sc_ld    ds   buf  vmcnt0')
for i, c in enumerate(iv):
    print(f"{i:7d} {c['n']:6d} {c['mfma']:5d} {c['valu']:5d} {c['wl']:5d} {c['sst']:6d} {c['sld']:6d} {c['ds']:5d} {c['buf']:5d} {c['vm0']:6d}")
    for k in tot: tot[k] += c[k]
print('total  ', tot)
