"""Per-dispatch averages of rocprofv3 --pmc counters for the kernels whose name contains a substring, with the derived busy fractions (MI355X_MICROARCH.md units:
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves, SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs, SQ_BUSY_CYCLES is
summed over the 32 shader engines, GRBM_GUI_ACTIVE over the 8 XCDs).

usage: python scripts/pmc_kernel_counters.py <rocprofv3 output dir>[,<dir of another pass>...] <kernel name substring> [label]

Round 6: (i) several passes — gfx950 has EIGHT SQ counter slots per pass; round 5 asked for nine in one pass and got one stale
`SQ_LDS_BANK_CONFLICT` value repeated for three different kernels.  The LDS pair (SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE) is collected in its OWN pass
(scripts/pmc_passes.sh) and merged here by directory; a counter that comes back with the identical value for every dispatch of different kernels is flagged.
(ii) the name filter matches the substring against the raw kernel name AND its demangled form, both with blanks removed — rocprofv3 prints mangled names
(`_ZN5e2eft13igemm5_kernelIDF16_Li1ELb0EEEvNS_11IgemmParamsEi`) in some output modes and demangled ones (`void e2eft::igemm5_kernel<_Float16, 1, false>(...)`) in others."""
import collections
import csv
import glob
import os
import subprocess
import sys

dirs, pat = sys.argv[1].split(","), sys.argv[2]
label = sys.argv[3] if len(sys.argv) > 3 else pat
files = []
for d in dirs:
    got = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert got, "no *_counter_collection.csv under %s" % d
    files += got


def demangle_e2eft(n):
    """`_ZN5e2eft<len><name>I<template args>E...` -> `e2eft::name<args>` for the argument kinds the library's kernels use (the image's c++filt is
    binutils 2.38: it does not know `DF16_` = _Float16 / `DF16b` = __bf16 and leaves such names mangled)"""
    import re
    m = re.match(r"_ZN5e2eft(\d+)", n)
    if not m:
        return n
    ln = int(m.group(1))
    i = m.end()
    name, rest = n[i:i + ln], n[i + ln:]
    if not rest.startswith("I"):
        return "e2eft::" + name
    rest, args = rest[1:], []
    while rest and not rest.startswith("E"):
        for tok, txt in (("DF16_", "_Float16"), ("DF16b", "__bf16"), ("f", "float"), ("d", "double"), ("h", "unsigned char"), ("i", "int"), ("l", "long")):
            if rest.startswith(tok):
                args.append(txt)
                rest = rest[len(tok):]
                break
        else:
            m2 = re.match(r"L([bijlm])(n?\d+)E", rest)
            if not m2:
                return n
            v = m2.group(2).replace("n", "-")
            args.append({"0": "false", "1": "true"}.get(v, v) if m2.group(1) == "b" else v)
            rest = rest[m2.end():]
    return "e2eft::%s<%s>" % (name, ", ".join(args))


def demangle(names):
    mangled = [n for n in names if n.startswith("_Z")]
    out = {n: n for n in names}
    if not mangled:
        return out
    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "llvm-cxxfilt", "c++filt"):
        try:
            r = subprocess.run([tool], input="\n".join(mangled) + "\n", capture_output=True, text=True, timeout=60)
        except (OSError, subprocess.TimeoutExpired):
            continue
        lines = r.stdout.splitlines()
        if r.returncode == 0 and len(lines) == len(mangled):
            out.update(dict(zip(mangled, lines)))
            break
    for n in mangled:
        if out[n] == n:
            out[n] = demangle_e2eft(n)
    return out


rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
dm = demangle({r["Kernel_Name"] for r in rows})
squash = lambda s: s.replace(" ", "")
want = squash(pat)
acc = collections.defaultdict(lambda: [0, 0.0])
names = set()
per_counter_all = collections.defaultdict(set)      # counter -> set of (kernel, value) over ALL kernels of the run: detects a stale / multiplexed counter
for r in rows:
    raw = r["Kernel_Name"]
    per_counter_all[r["Counter_Name"]].add((raw, r["Counter_Value"]))
    if want in squash(raw) or want in squash(dm[raw]):
        a = acc[r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        names.add(dm[raw].split("(")[0][:90])
if not acc:
    print("%s: no dispatch matches %r (raw or demangled); kernels in the run: %s" % (label, pat, sorted({dm[r["Kernel_Name"]].split("(")[0][:70] for r in rows})[:12]))
    sys.exit(0)
avg = {k: v[1] / v[0] for k, v in acc.items()}
n = max(v[0] for v in acc.values())
print("%s — %s, %d dispatches, averages per dispatch" % (label, sorted(names), n))
for k in sorted(avg):
    flag = ""
    kernels = {kn for kn, _ in per_counter_all[k]}
    values = {v for _, v in per_counter_all[k]}
    if len(kernels) > 1 and len(values) == 1:
        flag = "   <-- the SAME value for all %d kernels of the run: stale / not collected in this pass, ignore" % len(kernels)
    print("  %-28s %.4e%s" % (k, avg[k], flag))
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
