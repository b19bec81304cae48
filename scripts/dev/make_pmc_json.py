"""Dev: turn the two rocprofv3 --pmc passes of scripts/dev/run_pmc.sh into profiles/<tag>_pmc_hbm_traffic.json
(bench.py reads the newest one for `roofline.traffic`).  Usage: python scripts/dev/make_pmc_json.py <tag>"""
import collections
import csv
import glob
import json
import sys

tag = sys.argv[1]
vals = {}
for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
    files = glob.glob('gpurun_out/pmc_%s_%s/**/*counter_collection.csv' % (tag, counter), recursive=True)
    agg = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get('Counter_Name') == counter:
                agg[row['Kernel_Name']].append(float(row['Counter_Value']))
    vals[counter] = agg
kernels = {}
for k in vals['FETCH_SIZE']:
    if 'empose' not in k or k not in vals['WRITE_SIZE']:
        continue
    f, w = vals['FETCH_SIZE'][k], vals['WRITE_SIZE'][k]
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    kernels[k] = {'FETCH_SIZE_KiB_mean': fm, 'WRITE_SIZE_KiB_mean': wm, 'dispatches': len(f),
                  'hbm_bytes_per_launch_corrected': (2.0 * fm + wm) * 1024.0}
note = ('rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of `python bench.py '
        '--steps 2 --warmup 1 --no_cpu_baseline --no_profile`; counter unit is KiB. MI355X_MICROARCH.md (HBM section): '
        'on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming read, so '
        'hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.')
out = 'gpurun_out/%s_pmc_hbm_traffic.json' % tag
json.dump({'_note': note, 'kernels': kernels}, open(out, 'w'), indent=1, sort_keys=True)
for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]['hbm_bytes_per_launch_corrected'] * kv[1]['dispatches']):
    print('%-70s n=%3d  %.1f MB/launch' % (k[:70], v['dispatches'], v['hbm_bytes_per_launch_corrected'] / 1e6))
print('wrote', out)
