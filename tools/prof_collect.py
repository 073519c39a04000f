"""Turn the rocpd sqlite files written by tools/prof_forward.sh into the small files kept under profiles/:
<tag>_kernel_stats.csv (per-kernel averages), <tag>_pmc.json (mean counters per launch of the render kernel) and
traffic.json (HBM bytes per launch from FETCH_SIZE / WRITE_SIZE).   usage: python tools/prof_collect.py <dir> <tag>"""
import collections, glob, json, os, sqlite3, sys

d, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
prof = os.path.join(root, 'profiles')
KERNEL = 'render_kernel_pipe<1, 0'         # TP = 1, MLP = auto: what the headline step launches (it runs the f16x3 body at config 2)


def db(name):
    f = glob.glob(os.path.join(d, name, '**', '*.db'), recursive=True)
    return sqlite3.connect(f[0]) if f else None


c = db('stats')
if c is not None:
    rows = c.execute('select name, total_calls, total_duration, average, percentage from top_kernels').fetchall()
    with open(os.path.join(prof, f'{tag}_kernel_stats.csv'), 'w') as fh:
        fh.write('Name,Calls,TotalDurationUs,AverageUs,Percentage\n')
        for name, calls, tot, avg, pct in rows:
            short = name if len(name) < 120 else name[:60] + '...' + name[-40:]
            fh.write('"%s",%d,%d,%.1f,%.4f\n' % (short.replace('"', "'"), calls, tot, avg, pct))

counters, repack_fetch = {}, None
for name in ('sq1', 'sq2', 'tcc', 'fetch', 'write'):
    c = db(name)
    if c is None:
        continue
    per = collections.defaultdict(list)
    for disp, ctr, val in c.execute("select dispatch_id, counter_name, sum(value) from counters_collection "
                                    "where kernel_name like ? group by dispatch_id, counter_name", ('%' + KERNEL + '%',)):
        per[ctr].append(val)
    for ctr, vals in per.items():
        counters[ctr] = sum(vals) / len(vals)
    if name == 'fetch':
        vals = [v for (v,) in c.execute("select sum(value) from counters_collection where kernel_name like '%nchw_to_nhwc%' "
                                         "and counter_name = 'FETCH_SIZE' group by dispatch_id")]
        repack_fetch = sum(vals) / len(vals) if vals else None
try:
    head = open(os.path.join(root, 'g-nerf_amd', 'gnerf_hip', 'BUILD_HEAD')).read().strip()
except OSError:
    head = None
json.dump({'kernel': KERNEL, 'workload': 'bench.py config 2 (65536 rays x 96 samples per launch)', 'head': head, 'counters_mean_per_launch': counters},
          open(os.path.join(prof, f'{tag}_pmc.json'), 'w'), indent=1)
if 'FETCH_SIZE' in counters and 'WRITE_SIZE' in counters:
    json.dump({'render_kernel_hbm_bytes_per_launch': int(counters['FETCH_SIZE'] * 1024 * 2 + counters['WRITE_SIZE'] * 1024),
               'fetch_size_raw_KB': counters['FETCH_SIZE'], 'write_size_raw_KB': counters['WRITE_SIZE'],
               'repack_kernel_fetch_size_raw_KB': repack_fetch, 'head': head,
               'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; bytes = FETCH_SIZE*1024*2 (gfx950 reports half '
                       'the bytes of 16-B/lane loads; calibrated on nchw_to_nhwc_kernel in the same run, which reads 100.66 MB) + WRITE_SIZE*1024'},
              open(os.path.join(prof, 'traffic.json'), 'w'), indent=1)
print(json.dumps({k: counters[k] for k in sorted(counters)}, indent=1))
