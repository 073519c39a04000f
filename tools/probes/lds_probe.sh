#!/bin/bash
# Runs tools/probes/lds_probe.hip under rocprofv3 counters (on the GPU box through gpurun) -> gpurun_out/lds_probe.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/probes/lds_probe.hip -o /tmp/lds_probe || exit 1
: > $R/gpurun_out/lds_probe.txt
for m in ${PROBES:-linear pitch36 same_bank b32_linear}; do
  rm -rf /tmp/lp_$m
  timeout 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE -d /tmp/lp_$m -o run -- /tmp/lds_probe ${m//_/ } > /tmp/lp_$m.log 2>&1
  python3 - "$m" >> $R/gpurun_out/lds_probe.txt <<'PY'
import glob, sqlite3, sys, collections
m = sys.argv[1]
f = glob.glob(f'/tmp/lp_{m}/**/*.db', recursive=True)
if not f:
    print(m, 'no counters:', open(f'/tmp/lp_{m}.log').read()[-300:]); sys.exit(0)
per = collections.defaultdict(list)
for disp, ctr, val in sqlite3.connect(f[0]).execute('select dispatch_id, counter_name, sum(value) from counters_collection group by dispatch_id, counter_name'):
    per[ctr].append(val)
print(m, {k: sum(v) / len(v) for k, v in per.items()})
PY
done
cat $R/gpurun_out/lds_probe.txt
