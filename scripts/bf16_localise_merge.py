"""side-by-side table of scripts/bf16_localise.py's two sides: python scripts/bf16_localise_merge.py <cpu.tsv> <hip.tsv> [out.tsv]"""
import sys


def load(p):
    rows, notes = {}, []
    for line in open(p):
        if line.startswith("#"):
            notes.append(line.rstrip())
            continue
        f = line.rstrip("\n").split("\t")
        if f[0] == "probe":
            continue
        rows[(f[0], f[1])] = (float(f[2]), float(f[3]), float(f[4]))
    return rows, notes


cpu, ncpu = load(sys.argv[1])
hip, nhip = load(sys.argv[2])
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
out.write("# relative L2 error against the fp32 run of the same implementation: median (min .. max) over the draws; ratio = HIP median / torch-bf16 median\n")
out.write("# torch bf16 (CPU oracle with a bf16 state dict): %s\n# HIP bf16: %s\n" % (ncpu[-1].lstrip("# "), nhip[-1].lstrip("# ")))
out.write("%-58s %-4s %-34s %-34s %s\n" % ("probe", "dir", "torch bf16 (CPU)", "HIP bf16", "ratio"))
for key in cpu:
    if key not in hip:
        continue
    c, h = cpu[key], hip[key]
    out.write("%-58s %-4s %.3e (%.3e .. %.3e)   %.3e (%.3e .. %.3e)   %.2f\n" % (key[0], key[1], c[0], c[1], c[2], h[0], h[1], h[2], h[0] / c[0]))
