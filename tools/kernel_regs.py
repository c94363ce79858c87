"""Register / spill / LDS figures of every kernel in a hipcc -save-temps assembly file: python tools/kernel_regs.py <file.s> [name filter]"""
import re, sys
s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in s.split('- .agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    if flt not in name: continue
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk).group(1)
    print(name[:70], 'agpr', blk.split()[0], 'vgpr', g('vgpr_count'), 'spill', g('vgpr_spill_count'), 'sgpr', g('sgpr_count'), 'scratch', g('private_segment_fixed_size'))
