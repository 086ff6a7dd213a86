#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_train_gpu.py tests/test_bwd_gpu.py tests/test_two_ranks_one_gpu.py -q -m gpu -x 2>&1 | tail -6
rm -rf /tmp/prof_train; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train -o p -- python bench.py --train --steps 2 --warmup 1 > $O/r04h_bench_train_prof.json 2>/dev/null
cp $(find /tmp/prof_train -name "*kernel_stats.csv" | head -1) $O/r04h_train_rocprofv3_kernel_stats.csv
python bench.py --train --steps 3 --warmup 1 > $O/r04h_bench_train_bf16.json 2>/dev/null
python - <<PY
import csv, json
rows=list(csv.DictReader(open("$O/r04h_train_rocprofv3_kernel_stats.csv")))
n=3
aten=[r for r in rows if 'at::' in r['Name'] or 'rocclr' in r['Name']]
print("total ms/step %.1f, aten ms/step %.1f" % (sum(float(r['TotalDurationNs']) for r in rows)/1e6/n, sum(float(r['TotalDurationNs']) for r in aten)/1e6/n))
for r in sorted(aten,key=lambda r:-float(r['TotalDurationNs']))[:14]:
    print("%8.2f ms/step %7.1f calls/step  %s" % (float(r['TotalDurationNs'])/1e6/n, int(r['Calls'])/n, r['Name'][:150]))
print("train", json.load(open("$O/r04h_bench_train_bf16.json"))["value"])
PY
