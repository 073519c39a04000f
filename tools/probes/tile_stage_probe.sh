#!/bin/bash
# Builds and runs tools/probes/tile_stage_probe.hip on the GPU box (through gpurun): timing + checksum first, then one rocprofv3 --pmc pass
# per arm for the L1 -> L2 request count and a --kernel-trace --stats pass -> gpurun_out/tile_stage_probe.json
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
mkdir -p $R/gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-unused-value $R/tools/probes/tile_stage_probe.hip -o /tmp/tile_stage_probe || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 /tmp/tile_stage_probe both > /tmp/tsp_timing.json || { cat /tmp/tsp_timing.json; echo "probe failed"; exit 1; }
for arm in direct staged; do
  rm -rf /tmp/tsp_$arm
  timeout -k 10 180 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum SQ_INSTS_VALU SQ_INSTS_LDS -d /tmp/tsp_$arm -o run -- /tmp/tile_stage_probe $arm > /tmp/tsp_$arm.log 2>&1
done
python3 - <<'PY' > $R/gpurun_out/tile_stage_probe.json
import collections, glob, json, sqlite3
out = json.load(open('/tmp/tsp_timing.json'))
for arm in ('direct', 'staged'):
    f = glob.glob(f'/tmp/tsp_{arm}/**/*.db', recursive=True)
    if not f:
        out[arm + '_counters'] = 'no counters: ' + open(f'/tmp/tsp_{arm}.log').read()[-300:]
        continue
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for kname, disp, ctr, val in sqlite3.connect(f[0]).execute('select kernel_name, dispatch_id, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name'):
        if 'lookup_kernel' in kname:
            per[kname[:60]][ctr].append(val)
    out[arm + '_counters_per_launch'] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in per.items()}
    for k, d in per.items():
        if 'TCP_TCC_READ_REQ_sum' in d:
            out[arm + '_l1_to_l2_read_bytes_per_launch'] = sum(d['TCP_TCC_READ_REQ_sum']) / len(d['TCP_TCC_READ_REQ_sum']) * 128
print(json.dumps(out, indent=1))
PY
cat $R/gpurun_out/tile_stage_probe.json
