#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a hipcc -S dump.
usage: tools_isa_blocks.py file.s kernel_substring"""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
starts = [i for i, l in enumerate(lines) if re.match(r'^_Z.*:\s*(;.*)?$', l) and key in l]
s = starts[0]
e = next(i for i in range(s, len(lines)) if 's_endpgm' in lines[i])
blk = 'entry'; stats = {blk: [0] * 6}; order = [blk]
for l in lines[s:e]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blk = m.group(1); stats[blk] = [0] * 6; order.append(blk); continue
    if 'v_mfma' in l: stats[blk][0] += 1
    elif 'scratch_' in l: stats[blk][1] += 1
    elif 'ds_read' in l or 'ds_write' in l: stats[blk][2] += 1
    elif 'global_load' in l or 'global_store' in l: stats[blk][3] += 1
    elif re.match(r'\s+v_', l): stats[blk][4] += 1
    elif re.match(r'\s+s_', l): stats[blk][5] += 1
for b in order:
    if any(stats[b][:2]) or stats[b][4] > 30:
        print('%-10s mfma %4d scratch %3d lds %3d vmem %3d valu %4d salu %4d' % ((b,) + tuple(stats[b])))
