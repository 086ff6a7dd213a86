"""HBM traffic per kernel launch from rocprofv3 PMC counters -> profiles/r01_pmc_hbm_traffic.json (read by bench.py's roofline.traffic).

Collect (on the GPU box; counters in their OWN runs, --kernel-trace only, as gpurun requires):
    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
    for c in FETCH_SIZE WRITE_SIZE; do
      rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$c -o p -- \\
          python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-train-leg
    done
    python scripts/pmc_traffic.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/r01_pmc_hbm_traffic.json [steps_profiled]
steps_profiled = the number of inference steps inside one profiled process (default 8 = 1 warm-up + 1 timed + 1 fully instrumented + 3 stage-timing steps
+ ... : bench.py prints it as `steps_in_process` on stderr); every group also gets `launches_per_step`, which bench.py matches against its own launch count
before it prints the figure (a profile of another build — another launch population — is not printed).
Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of a wide
coalesced read -> doubled.  The correction is re-checked on every collection with gn_apply, whose read and write volumes are equal by
construction (raw FETCH / WRITE must come out at 0.50)."""
import csv
import glob
import json
import os
import sys

# "igemm" = every kernel behind e2eft_conv2d_fwd / e2eft_gemm, the family bench.py times (the persistent igemm5 / igemm6 take the big launches,
# igemm2 the rest, conv3x3_narrow the 3-channel conv_out); "igemm2" / "igemm5" / ... separately
GROUPS = {"igemm": ("igemm2_kernel", "igemm5_kernel", "igemm6_kernel", "conv3x3_narrow", "conv_thin_in_kernel"), "igemm6": ("igemm6_kernel",), "conv_thin_in": ("conv_thin_in_kernel",), "igemm5": ("igemm5_kernel",), "igemm2": ("igemm2_kernel",), "gn_apply": ("gn_apply_kernel",),
          "attn_fwd": ("attn_fwd_kernel",), "attn512_fwd": ("attn512_fwd_kernel",), "conv3x3_narrow": ("conv3x3_narrow",)}


def load(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    assert files, "no *_counter_collection.csv under %s" % d
    per = {}
    with open(files[0]) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != counter:
                continue
            # fp32 instantiations are not part of the fp16 step: the one-time build of the cross-attention fold (modules.Attention._fold, 80 small GEMMs in the
            # first step of a process) runs on igemm2_kernel<float, ...> since round 6 and would make launches / step fractional
            if "igemm2_kernelIf" in row["Kernel_Name"] or "igemm2_kernel<float" in row["Kernel_Name"]:
                continue
            for g, pats in GROUPS.items():
                if any(pat in row["Kernel_Name"] for pat in pats):
                    a = per.setdefault(g, [0, 0.0])
                    a[0] += 1
                    a[1] += float(row["Counter_Value"])
    return per


def main():
    fd, wd, out = sys.argv[1:4]
    steps = float(sys.argv[4]) if len(sys.argv) > 4 else 8.0
    fetch, write = load(fd, "FETCH_SIZE"), load(wd, "WRITE_SIZE")
    res = {"steps_profiled": steps, "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over `python bench.py --steps 1 --warmup 1 "
                     "--no-cpu-baseline --no-train-leg` (B=8, 768x768, fp16), aggregated by scripts/pmc_traffic.py; counter unit KiB; FETCH_SIZE doubled "
                     "per MI355X_MICROARCH.md (gfx950 reports half of wide coalesced reads) - re-checked on gn_apply, whose read and write volumes are "
                     "equal by construction (raw FETCH / WRITE ratio recorded below)", "kernels": {}}
    for g in GROUPS:
        if g not in fetch or g not in write:
            continue
        n = fetch[g][0]
        rf, rw = fetch[g][1] / n, write[g][1] / write[g][0]
        res["kernels"][g] = {"launches": n, "launches_per_step": n / steps, "fetch_bytes_per_launch_corrected": 2 * rf * 1024, "write_bytes_per_launch": rw * 1024,
                             "hbm_bytes_per_launch": (2 * rf + rw) * 1024, "raw_fetch_kib_per_launch": rf, "raw_write_kib_per_launch": rw}
    if "gn_apply" in res["kernels"]:
        k = res["kernels"]["gn_apply"]
        res["calibration_gn_apply_raw_fetch_over_write"] = k["raw_fetch_kib_per_launch"] / k["raw_write_kib_per_launch"]
    # identity of the build the counters were collected on: bench.py prints the figure only for a library with this id ("same build" used to be
    # inferred from launch counts alone)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from diffusion_e2e_ft_amd import build as _b
    res["build_id"] = _b.built_id()
    res["source_id"] = _b.source_id()
    assert res["build_id"] == res["source_id"], "the library on this box was not built from the sources on this box: %r" % (res,)
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps({g: round(v["hbm_bytes_per_launch"] / 1e6, 1) for g, v in res["kernels"].items()}), res.get("calibration_gn_apply_raw_fetch_over_write"))


if __name__ == "__main__":
    main()
