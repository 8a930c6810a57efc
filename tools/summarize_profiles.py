#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel stats + PMC passes) into the tracked summaries under profiles/.

    python tools/summarize_profiles.py <gpurun_out/dir> <tag>

Expects <dir>/ktrace/*_kernel_stats.csv and the PMC passes <dir>/pmc_fetch, <dir>/pmc_write, <dir>/pmc_sq (each collected in
its own rocprofv3 run with --kernel-trace only, as the MI355X guide prescribes).  Writes profiles/<tag>_rocprofv3_kernel_stats.csv,
profiles/<tag>_rocprofv3_pmc_summary.md and profiles/<tag>_pmc_traffic.json (HBM bytes per launch of each kernel, read by bench.py
for roofline.traffic)."""
import collections
import csv
import glob
import json
import re
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(src, tag, workload='sp1_2p20'):
    out_dir = os.path.join(ROOT, 'profiles')
    os.makedirs(out_dir, exist_ok=True)
    ks = glob.glob(os.path.join(src, 'ktrace', '*_kernel_stats.csv'))
    if ks:
        shutil.copy(ks[0], os.path.join(out_dir, tag + '_rocprofv3_kernel_stats.csv'))
    data = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.defaultdict(int))
    for sub in ('pmc_fetch', 'pmc_write', 'pmc_sq'):
        for f in glob.glob(os.path.join(src, sub, '*_counter_collection.csv')):
            for r in csv.DictReader(open(f)):
                k = r['Kernel_Name']
                if 'zkv::k_' not in k or 'setup' in k:
                    continue
                k = re.sub(r'<.*?>', '', k.split('(')[0].replace('zkv::', '').replace('void ', '')).strip()
                data[k][r['Counter_Name']] += float(r['Counter_Value'])
                calls[k][r['Counter_Name']] += 1
    if not data:
        print('no PMC data under', src)
        return
    # average per dispatch (one counter row per dispatch per counter; dimensions are already summed by rocprofv3 csv)
    per = {k: {c: v / max(1, calls[k][c]) for c, v in d.items()} for k, d in data.items()}
    cols = ['FETCH_SIZE', 'WRITE_SIZE', 'SQ_WAVES', 'SQ_INSTS_VALU', 'SQ_INSTS_SALU', 'SQ_WAVE_CYCLES', 'SQ_WAIT_INST_ANY', 'SQ_BUSY_CYCLES']
    lines = ['# %s - rocprofv3 PMC summary (MI355X, bench.py workload %s)\n' % (tag, workload),
             'Each counter set was collected in its own pass with --kernel-trace only:',
             '    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-wire --no-mulmod',
             '    rocprofv3 --pmc WRITE_SIZE --kernel-trace ...      rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES ... --kernel-trace ...\n',
             'FETCH_SIZE / WRITE_SIZE in KiB per dispatch as rocprofv3 reports them.  The gfx950 x2 FETCH_SIZE correction of',
             'MI355X_MICROARCH.md applies to wide (16 B/lane) streaming reads; these kernels issue 4 B/lane struct-of-arrays and scratch',
             'accesses, which that guide lists as uncalibrated, so raw values are given -- except for k_wire_* (16 B/lane coalesced',
             'streaming of calldata), whose FETCH_SIZE is doubled in the traffic figure as the guide prescribes.\n',
             '| kernel | ' + ' | '.join(cols) + ' |', '|---|' + '---|' * len(cols)]
    for k in sorted(per):
        lines.append('| %s | ' % k + ' | '.join('%.6g' % per[k].get(c, float('nan')) for c in cols) + ' |')
    lines.append('')
    traffic = {}
    for k, d in per.items():
        if 'FETCH_SIZE' in d and 'WRITE_SIZE' in d:
            traffic[k] = ((2.0 if k.startswith('k_wire') else 1.0) * d['FETCH_SIZE'] + d['WRITE_SIZE']) * 1024.0
        if 'SQ_INSTS_VALU' in d and d.get('SQ_WAVES'):
            lines.append('%s: %.3g VALU instructions per wave, %.3g cycles per wave (SQ_WAVE_CYCLES counts quad-cycles), %.2f wave-cycles per VALU instruction'
                         % (k, d['SQ_INSTS_VALU'] / d['SQ_WAVES'], d['SQ_WAVE_CYCLES'] * 4 / d['SQ_WAVES'], d['SQ_WAVE_CYCLES'] * 4 / d['SQ_INSTS_VALU']))
    lines.append('')
    lines.append('HBM traffic per launch = (FETCH_SIZE + WRITE_SIZE) KiB: ' + ', '.join('%s %.3g GB' % (k, v / 1e9) for k, v in sorted(traffic.items())))
    open(os.path.join(out_dir, tag + '_rocprofv3_pmc_summary.md'), 'w').write('\n'.join(lines) + '\n')
    json.dump({'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bytes per launch', 'workload': workload, 'bytes_per_launch': traffic},
              open(os.path.join(out_dir, tag + '_pmc_traffic.json'), 'w'), indent=1)
    print('\n'.join(lines[-6:]))


if __name__ == '__main__':
    main(*sys.argv[1:4])
