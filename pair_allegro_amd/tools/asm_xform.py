"""Edits a device assembly file of fused.hip for fault bisection (see tools/asm_variant.sh).  usage: asm_xform.py in.s MODE out.s
MODE: lgkm0 | vm0                        every s_waitcnt gets lgkmcnt(0) / vmcnt(0)                                  (whole file)
      lgkm_every | vm_every | nop_every  a wait / s_nop 0 in front of every instruction                             (selected kernels)
      wait_before_mfma | nop_after_mfma  full wait before every MFMA / 24 wait states after every MFMA run
      nopb_<mfma|valu|ds|vmem|salu|sync> s_nop 0 BEFORE every instruction of the class
      nopa_<mfma|trans|perm|lane|pk>     s_nop 0 AFTER every instruction of the class
      rg_<k>_<n> | rgp_<k>_<n> | rgs_<k>_<n> | rgx_0_<n>_<a-b,c,...>   nopb_valu only in chunk k / chunks <= k / >= k / the listed chunks of n
Selected kernels = the 4-wave, two-layer, tabulated-two-body instances of the two bf16-split arithmetics (KSEL / KS below).
Round 4 result: nop_every, nopb_valu cure the fault; lgkm0, nopa_mfma, nopb_<anything else> do not; two sites (chunks 22/23 and 25 of 64):
`buffer_store_dwordx4 v[6:9], ..., s23 offen` followed at once by a VALU write of v6."""
import re, sys
INSTR = re.compile(r'^\t(s_|v_|ds_|buffer_|global_|scratch_|flat_)')
def xf(lines, mode):
    out = []
    n = len(lines)
    active = False
    icount = {}
    if mode.startswith('rg'):
        # pre-count instructions per selected kernel
        cur = None
        KS = re.compile(r'^_ZN4ahip7k_fusedILi4ELb0ELi[12]ELb1ELi2EEEvNS_9FusedArgsE:')
        for l in lines:
            if KS.match(l): cur = l; icount[cur] = 0
            if l.startswith('.Lfunc_end'): cur = None
            if cur and INSTR.match(l): icount[cur] += 1
    curk = None; idx = 0
    KSEL = re.compile(r'^_ZN4ahip7k_fusedILi4ELb0ELi[12]ELb1ELi2EEEvNS_9FusedArgsE:')
    for i, l in enumerate(lines):
        if KSEL.match(l): active = True; curk = l; idx = 0
        if l.startswith('.Lfunc_end'): active = False
        is_i = bool(INSTR.match(l)) and (active or mode in ('lgkm0', 'vm0'))
        mn = l.split()[0] if is_i else ''
        if mode == 'lgkm0' and mn == 's_waitcnt':
            l = re.sub(r'lgkmcnt\(\d+\)', 'lgkmcnt(0)', l)
        if mode == 'vm0' and mn == 's_waitcnt':
            l = re.sub(r'vmcnt\(\d+\)', 'vmcnt(0)', l)
        if is_i and mode == 'lgkm_every' and mn not in ('s_waitcnt',):
            out.append('\ts_waitcnt lgkmcnt(0)\n')
        if is_i and mode == 'vm_every' and mn not in ('s_waitcnt',):
            out.append('\ts_waitcnt vmcnt(0)\n')
        if is_i and mode == 'nop_every':
            out.append('\ts_nop 0\n')
        if is_i and mode == 'wait_before_mfma' and mn.startswith('v_mfma'):
            out.append('\ts_waitcnt vmcnt(0) lgkmcnt(0)\n')
        if is_i and mode.startswith('rg'):
            parts = mode.split('_'); k = int(parts[1]); nn = int(parts[2])
            tot = icount[curk]
            ck = (idx * nn) // tot
            sel = (ck == k) if len(mode.split('_')[0]) == 2 else False
            if mode.startswith('rgx_'):
                sel = any(int(r.split('-')[0]) <= ck <= int(r.split('-')[-1]) for r in mode.split('_', 3)[3].split(','))
            if mode.startswith('rgp_'): sel = ck <= k
            if mode.startswith('rgs_'): sel = ck >= k
            if sel and mn.startswith('v_') and not mn.startswith('v_mfma'): out.append('\ts_nop 0\n')
            idx += 1
        if is_i and mode.startswith('nopb_'):
            cls = mode[5:]
            hit = {'mfma': mn.startswith('v_mfma'), 'valu': mn.startswith('v_') and not mn.startswith('v_mfma'),
                   'ds': mn.startswith('ds_'), 'vmem': mn.startswith(('buffer_', 'global_', 'scratch_', 'flat_')),
                   'salu': mn.startswith('s_') and mn not in ('s_waitcnt', 's_nop', 's_barrier'),
                   'sync': mn in ('s_waitcnt', 's_barrier')}[cls]
            if hit: out.append('\ts_nop 0\n')
        out.append(l)
        if is_i and mode.startswith('nopa_'):
            cls = mode[5:]
            hit = {'mfma': mn.startswith('v_mfma'), 'trans': mn.split('_e')[0] in ('v_exp_f32', 'v_rcp_f32', 'v_sqrt_f32', 'v_sin_f32', 'v_cos_f32', 'v_rsq_f32', 'v_log_f32'),
                   'perm': mn.startswith('v_perm'), 'lane': mn.startswith(('v_readlane', 'v_writelane', 'v_readfirstlane')),
                   'pk': mn.startswith('v_pk_')}[cls]
            if hit: out.append('\ts_nop 0\n')
        if mode == 'nop_after_mfma' and mn.startswith('v_mfma'):
            # next instruction
            j = i + 1
            while j < n and not INSTR.match(lines[j]): j += 1
            if j < n and not lines[j].split()[0].startswith('v_mfma'):
                out.append('\ts_nop 15\n\ts_nop 7\n')
    return out
src, mode, dst = sys.argv[1:4]
open(dst, 'w').writelines(xf(open(src).readlines(), mode))
