"""Recompute every field of bench.py's `roofline` object from the committed profiles alone:
    python tools/roofline_from_profiles.py [tag]          (default: the latest profiles/rNN_final*_pmc.json)
kernel time   <- profiles/<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats average of render_kernel_pipe)
counters      <- profiles/<tag>_pmc.json         (separate --pmc passes, tools/prof_forward.sh)
HBM traffic   <- profiles/traffic.json
The accounting itself (algorithmic VALU table, peaks, issue prices) is bench.py's, imported from there."""
import csv, importlib.util, json, os, re, sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location('bench', os.path.join(root, 'bench.py'))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

if len(sys.argv) > 1:
    tag = sys.argv[1]
    pmc = json.load(open(os.path.join(root, 'profiles', f'{tag}_pmc.json')))['counters_mean_per_launch']
    src = f'profiles/{tag}_pmc.json'
else:
    pmc, src = bench.latest_pmc()
    tag = re.sub(r'_pmc\.json$', '', os.path.basename(src))
ms = None
with open(os.path.join(root, 'profiles', f'{tag}_kernel_stats.csv')) as fh:
    for row in csv.DictReader(fh):
        if 'render_kernel_pipe<1, 0' in row['Name'] and int(row['Calls']) > 0:          # the auto-arithmetic kernel of the headline step
            ms = float(row['AverageUs']) / 1e3
traffic = json.load(open(os.path.join(root, 'profiles', 'traffic.json'))).get('render_kernel_hbm_bytes_per_launch')
out = bench.roofline(ms, bench.N_ITEMS * bench.RES * bench.RES, bench.S_COARSE, bench.S_FINE, bench.PLANE, bench.N_ITEMS, pmc, src, traffic)
out['kernel_ms_source'] = f'profiles/{tag}_kernel_stats.csv (under the profiler)'
print(json.dumps(out, indent=1))
