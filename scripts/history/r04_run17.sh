#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 600 python -m pytest tests/test_train_gpu.py -q -m gpu -x -k "flat_cast" 2>&1 | tail -3
rm -rf /tmp/prof_train; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o p -- python bench.py --train --steps 2 --warmup 1 > /dev/null 2>&1
cp $(find /tmp/prof_train -name "*kernel_stats.csv" | head -1) $O/r04c_train_rocprofv3_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/r04c_train_rocprofv3_kernel_stats.csv")))
n=3
aten=[r for r in rows if 'at::' in r['Name']]
print("total ms/step %.1f, aten ms/step %.1f" % (sum(float(r['TotalDurationNs']) for r in rows)/1e6/n, sum(float(r['TotalDurationNs']) for r in aten)/1e6/n))
for r in sorted(aten,key=lambda r:-float(r['TotalDurationNs']))[:10]:
    print("%8.2f ms/step %6d calls/step  %s" % (float(r['TotalDurationNs'])/1e6/n, int(r['Calls'])/n, r['Name'][:130]))
for r in rows:
    if 'cast_kernel' in r['Name']: print("cast", float(r['TotalDurationNs'])/1e6/n, int(r['Calls'])/n, r['Name'][:80])
PY
