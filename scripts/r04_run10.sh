#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
scripts/bin/attn_v2_p1 8 5 9216 20 | tail -1
C="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
for v in v1 v2; do
  rm -rf /tmp/pmc_$v; rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$v -o p -- scripts/bin/attn_$v 8 5 9216 5 > /dev/null 2>&1
  python - <<PY
import csv,glob,collections
f=glob.glob("/tmp/pmc_$v/**/*counter_collection.csv",recursive=True)[0]
acc=collections.defaultdict(lambda:[0,0.0])
for r in csv.DictReader(open(f)):
    if "attn_fwd" in r["Kernel_Name"]:
        a=acc[r["Counter_Name"]]; a[0]+=1; a[1]+=float(r["Counter_Value"])
print("$v", {k: "%.4g" % (v[1]/v[0]) for k,v in sorted(acc.items())}, "dispatches", max(v[0] for v in acc.values()))
PY
done 2>&1 | tee $O/r04_attn_pmc_v1_v2.txt
