#!/usr/bin/env python3
"""Instruction mix, register use and DPP hazards of one kernel in a gfx950 assembly listing.
   hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o k.s egopose_amd/csrc/egp_kernels.hip
   python tools/isa_stats.py k.s k_pd_torque_grid58IdE
The DPP check: a VALU write of a VGPR needs two wait states before a DPP read of it; the compiler does not look into inline
asm, so the listing is checked instead (v_*_dpp source 0 against the destinations of the two instructions before it)."""
import collections, re, sys


def regs(tok):
    tok = tok.strip().rstrip(',')
    m = re.fullmatch(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r'v(\d+)', tok)
    return {int(m.group(1))} if m else set()


def main():
    text, pat = open(sys.argv[1]).read(), sys.argv[2]
    m = re.search(r'^(\w*%s\w*):[^\n]*\n(.*?)\n\s*s_endpgm' % re.escape(pat), text, re.S | re.M)
    if not m:
        sys.exit("kernel %s not found" % pat)
    name, body = m.group(1), m.group(2)
    ins = []
    for l in body.split('\n'):
        l = l.split(';')[0].strip()
        if not l or l.startswith('.') or l.endswith(':'):
            continue
        ins.append(l)
    c = collections.Counter(i.split()[0] for i in ins)
    valu = sum(v for k, v in c.items() if k.startswith('v_'))
    print("%s: %d instructions, %d VALU, %d DS, %d global/flat, %d scratch" % (
        name, len(ins), valu, sum(v for k, v in c.items() if k.startswith('ds_')),
        sum(v for k, v in c.items() if k.startswith(('global_', 'flat_', 'buffer_'))), sum(v for k, v in c.items() if k.startswith('scratch_'))))
    print("  " + ", ".join("%s %d" % kv for kv in c.most_common(24)))
    for key in ("num_vgpr", "num_agpr", "numbered_sgpr", "private_seg_size"):
        mm = re.search(r'\.set %s\.%s, (\d+)' % (re.escape(name), key), text)
        print("  %s = %s" % (key, mm.group(1) if mm else "?"))
    mm = re.search(r'\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel' % re.escape(name), text, re.S)
    if mm:
        g = re.search(r'group_segment_fixed_size (\d+)', mm.group(1))
        print("  LDS = %s B" % (g.group(1) if g else "?"))
    bad = 0
    for i, l in enumerate(ins):
        op = l.split()[0]
        if not op.endswith('_dpp'):
            continue
        ops = l[len(op):].split(',')
        src0 = regs(ops[1].split()[0])
        prev, back = [], i - 1
        waits = 0
        while back >= 0 and waits < 2:
            p = ins[back]
            pop = p.split()[0]
            if pop == 's_nop':
                waits += int(p.split()[1], 0) + 1
            else:
                waits += 1
                prev.append(p)
            back -= 1
        for p in prev:
            pop = p.split()[0]
            if pop.startswith('v_') and not pop.startswith(('v_cmp', 'v_readlane', 'v_readfirstlane')):
                if regs(p[len(pop):].split(',')[0]) & src0:
                    bad += 1
                    print("  DPP HAZARD: '%s' right after '%s'" % (l, p))
            if pop.startswith('v_cmpx'):
                bad += 1
                print("  EXEC HAZARD: '%s' after '%s'" % (l, p))
    print("  DPP hazards: %d" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
