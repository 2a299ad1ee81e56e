#!/usr/bin/env python3
"""Post-process tools/profile_round.sh output: per-kernel stats + HBM bytes per launch of the step kernel.
gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128-B request for wide coalesced
streaming reads -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.  Both are reported in KiB."""
import collections
import csv
import glob
import json
import sys

out, args = sys.argv[1], sys.argv[2]


def counter(dirname):
    f = glob.glob('%s/%s/*/*_counter_collection.csv' % (out, dirname))
    agg = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(f[0])):
        if 'step_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
            meta = {k: r[k] for k in ('Kernel_Name', 'Grid_Size', 'Workgroup_Size', 'VGPR_Count', 'Accum_VGPR_Count',
                                      'SGPR_Count', 'LDS_Block_Size', 'Scratch_Size')}
    # median: the handful of short settling launches (8 steps) and the slow ones after an upload do not belong to the figure
    return {k: sorted(v)[len(v) // 2] for k, v in agg.items()}, meta, {k: len(v) for k, v in agg.items()}


print('# command: python3 bench.py ' + args)
print('## rocprofv3 --kernel-trace --stats')
for f in glob.glob('%s/kt/*/*_kernel_stats.csv' % out):
    print(open(f).read().strip())
fetch, meta, n = counter('fetch')
write, _, _ = counter('write')
sq, _, _ = counter('sq')
print('## step kernel dispatch:', json.dumps(meta))
fs, ws = fetch['FETCH_SIZE'], write['WRITE_SIZE']
hbm = (2.0 * fs + ws) * 1024.0
print('## HBM counters (median over %d launches): FETCH_SIZE=%.1f KiB (x2 on gfx950 -> %.3f MB), WRITE_SIZE=%.1f KiB (%.3f MB)'
      % (n['FETCH_SIZE'], fs, 2 * fs * 1024 / 1e6, ws, ws * 1024 / 1e6))
print('## hbm_bytes_per_launch = %.0f' % hbm)
waves = float(meta['Grid_Size']) / 64
print('## SQ counters per wave: ' + ', '.join('%s=%.1f' % (k, v / waves) for k, v in sorted(sq.items())))
toks = args.split()
def opt(name, default):
    return int(toks[toks.index(name) + 1]) if name in toks else default
rec = {'batch': opt('--batch', 1024), 'nspecies': opt('--nspecies', 3), 'nx': opt('--nx', 512),
       'steps_per_launch': opt('--steps-per-launch', 256), 'method': 'Crank-Nicolson',
       'hbm_bytes_per_launch': hbm, 'fetch_size_kib': fs, 'write_size_kib': ws,
       'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024',
       'kernel': meta.get('Kernel_Name'),
       # per wave and launch; a lane of the batch is advanced by Grid_Size/64/batch waves
       'sq_insts_valu_per_wave': sq.get('SQ_INSTS_VALU', 0.0) / waves, 'sq_insts_lds_per_wave': sq.get('SQ_INSTS_LDS', 0.0) / waves,
       'waves_per_lane': waves / opt('--batch', 1024)}
json.dump(rec, open(out + '/hbm_traffic.json', 'w'), indent=1)
