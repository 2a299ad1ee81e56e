#!/usr/bin/env python3
"""Post-process the physical-mode passes of tools/profile_round.sh (newton_kt / newton_sq / newton_fetch / newton_write)."""
import collections
import csv
import glob
import sys

out = sys.argv[1]


def counter(dirname):
    f = glob.glob('%s/%s/*/*_counter_collection.csv' % (out, dirname))
    agg = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(f[0])):
        if 'newton' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            meta = {k: r[k] for k in ('Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'VGPR_Count', 'Accum_VGPR_Count',
                                      'SGPR_Count', 'LDS_Block_Size', 'Scratch_Size')}
    return {k: sum(v) / len(v) for k, v in agg.items()}, meta, {k: len(v) for k, v in agg.items()}


print('# command: python3 tools/newton_bench.py --steps 40   (batch 1024, 3 species, 512 points; launches: 5 warm-up steps, 40 timed steps)')
print(open(glob.glob('%s/newton_kt.log' % out)[0]).read().strip().splitlines()[-1])
print('## rocprofv3 --kernel-trace --stats')
for f in glob.glob('%s/newton_kt/*/*_kernel_stats.csv' % out):
    print(open(f).read().strip())
sq, meta, n = counter('newton_sq')
fetch, _, _ = counter('newton_fetch')
write, _, _ = counter('newton_write')
print('## newton kernel dispatch:', meta)
waves = float(meta['Grid_Size']) / 64
print('## SQ counters per wave (mean over %d launches): ' % n.get('SQ_WAVES', 0) + ', '.join('%s=%.1f' % (k, v / waves) for k, v in sorted(sq.items())))
print('## HBM counters per launch: FETCH_SIZE=%.1f KiB (x2 on gfx950 -> %.2f MB), WRITE_SIZE=%.1f KiB (%.2f MB)'
      % (fetch['FETCH_SIZE'], 2 * fetch['FETCH_SIZE'] * 1024 / 1e6, write['WRITE_SIZE'], write['WRITE_SIZE'] * 1024 / 1e6))
