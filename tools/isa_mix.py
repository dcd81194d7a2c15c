"""Instruction mix of the MFMA-carrying basic blocks of a kernel in a hipcc -save-temps .s file.
usage: python tools/isa_mix.py file.s <substring of mangled kernel name> [...]"""
import collections
import re
import sys


def main(path, pats):
    s = open(path).read()
    starts = [(m.start(), m.group(1)) for m in re.finditer(r'^(_ZN3ire\S*):', s, flags=re.M)]
    for idx, (pos, name) in enumerate(starts):
        if not any(p in name for p in pats):
            continue
        end = s.find('.Lfunc_end', pos)
        body = s[pos:end]
        blocks = re.split(r'^(\.LBB[0-9_]+):', body, flags=re.M)
        print(name[:100])
        for b in range(1, len(blocks), 2):
            bl = [l.strip() for l in blocks[b + 1].split('\n') if l.strip() and not l.strip().startswith(('.', ';'))]
            if not any(l.startswith('v_mfma') for l in bl):
                continue
            c = collections.Counter()
            for l in bl:
                op = l.split()[0]
                key = ('mfma' if op.startswith('v_mfma') else 'trans' if op.startswith(('v_exp', 'v_rcp', 'v_rsq', 'v_sqrt')) else 'vpk' if op.startswith('v_pk_') else
                       'valu' if op.startswith('v_') else 'ds' if op.startswith('ds_') else 'vmem' if op.startswith(('buffer_', 'global_', 'scratch_')) else
                       'wait' if op.startswith('s_waitcnt') else 'nop' if op.startswith('s_nop') else 'salu' if op.startswith('s_') else 'other')
                c[key] += 1
            print(' block', blocks[b], 'insts', len(bl), dict(c))
            v = collections.Counter(l.split()[0] for l in bl if l.startswith('v_') and not l.startswith('v_mfma'))
            print('   valu ops:', v.most_common(30))
            d = collections.Counter(l.split()[0] for l in bl if l.startswith(('ds_', 'buffer_', 'global_', 'scratch_')))
            print('   mem ops:', d.most_common(12))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2:])
