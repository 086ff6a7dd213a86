"""Per-dispatch averages of rocprofv3 --pmc counters for the kernels whose name contains a substring, with the derived busy fractions (MI355X_MICROARCH.md units:
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs, SQ_BUSY_CYCLES is
summed over the 32 shader engines, GRBM_GUI_ACTIVE over the 8 XCDs).
usage: python scripts/pmc_kernel_counters.py <rocprofv3 output dir> <kernel name substring> [label]"""
import collections
import csv
import glob
import os
import sys

d, pat = sys.argv[1], sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else pat
files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
assert files, "no *_counter_collection.csv under %s" % d
acc = collections.defaultdict(lambda: [0, 0.0])
names = set()
for r in csv.DictReader(open(files[0])):
    if pat in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        names.add(r["Kernel_Name"].split("(")[0][:90])
if not acc:
    print("%s: no dispatch matches %r" % (label, pat))
    sys.exit(0)
avg = {k: v[1] / v[0] for k, v in acc.items()}
n = max(v[0] for v in acc.values())
print("%s — %s, %d dispatches, averages per dispatch" % (label, sorted(names), n))
for k in sorted(avg):
    print("  %-28s %.4e" % (k, avg[k]))
dur = avg.get("GRBM_GUI_ACTIVE", 0) / 8 or avg.get("SQ_BUSY_CYCLES", 0) / 32
if dur:
    out = ["kernel duration %.4e cycles" % dur]
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        out.append("matrix pipe busy %.1f %% of them" % (100 * avg["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / dur))
    if "SQ_ACTIVE_INST_VALU" in avg:
        out.append("VALU busy %.1f %%" % (100 * 4 * avg["SQ_ACTIVE_INST_VALU"] / 1024 / dur))
    if "SQ_WAVE_CYCLES" in avg:
        for c, t in (("SQ_WAIT_INST_ANY", "issue-stalled"), ("SQ_WAIT_ANY", "parked (waitcnt / barrier)")):
            if c in avg:
                out.append("%s %.1f %% of the wave cycles" % (t, 100 * avg[c] / avg["SQ_WAVE_CYCLES"]))
    if "SQ_LDS_BANK_CONFLICT" in avg and "SQ_LDS_IDX_ACTIVE" in avg and avg["SQ_LDS_IDX_ACTIVE"]:
        out.append("LDS bank-conflict cycles %.2f %% of the LDS-active cycles" % (100 * avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]))
    print("  => " + "; ".join(out))
