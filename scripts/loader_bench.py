"""Training-input throughput ("nothing measures the loader"): how many Hypersim samples per second reach the GPU ready for
training.e2e_ft_loss, against what 8 GPUs consume (8 x ~65 images/s at configs[2]'s bf16 step, 8 x ~9 in the reference's fp32 recipe).

    python scripts/loader_bench.py [--samples 24] [--workers 16] [--batch 8] [--epochs 3] [--height 768 --width 1024]

Writes `--samples` synthetic Hypersim-format samples (full-size PNGs: rgb, 16-bit depth, normals; tests/dataset_fixture.py) into a temporary directory, then times
  reference_cpu   what the reference's `Hypersim.__getitem__` does per sample on ONE core (its DataLoader default num_workers = 0, train.py:143-149): Pillow decode,
                  align_normals (numpy float64), Pillow resizes, torch.quantile, normalisation — restated with the same library calls (oracle/dataprep_ref.py);
  decode_only     data.Hypersim.__getitem__ (decode only) on `--workers` threads;
  device_loader   data.DeviceLoader end to end (decode threads -> pinned staging -> upload -> e2eft_align_normals_u8 / aug / quantile / prepare kernels), when a GPU
                  is present: images/s of batches ready on the device, and the device time of the preparation alone.
One JSON line on stdout."""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def reference_cpu_sample(pr, near=1e-5, far=65.0):
    from PIL import Image
    from oracle import dataprep_ref
    rgb = Image.open(pr["rgb_path"]).convert("RGB")
    depth = np.array(Image.open(pr["depth_path"])) / 1000
    dimg = Image.fromarray(depth)
    nimg = Image.open(pr["normal_path"]).convert("RGB")
    nimg = Image.fromarray(dataprep_ref.align_normals_u8_ref(np.array(nimg), np.array(dimg)))
    rgb, nimg, dimg = rgb.resize((640, 480), Image.BILINEAR), nimg.resize((640, 480), Image.BILINEAR), dimg.resize((640, 480), Image.NEAREST)
    tt = lambda im: torch.from_numpy(np.array(im)).permute(2, 0, 1).float().div(255)
    return dataprep_ref.prepare_sample_ref(tt(rgb), torch.from_numpy(np.array(dimg, np.float32))[None], tt(nimg), near, far)


def rank_worker(args):
    """the HOST side of one data-parallel rank: its shard of every epoch decoded by its own thread pool and stacked into staging buffers, one batch of look-ahead.
    The device part (upload + align / augment / quantile / prepare kernels) is left out on purpose: in training every rank has its OWN GPU and spends 0.5 ms per batch
    there (`device_prepare`), while N processes sharing the one GPU of this box time-slice their contexts and measure the driver, not the loader (first version of
    this bench: 0.7 images/s per rank).  prints {"images", "seconds"}"""
    from concurrent.futures import ThreadPoolExecutor
    from diffusion_e2e_ft_amd import data
    root_dir, split_path = args.tree.split("::")
    torch.set_num_threads(1)      # (as the parent: N processes with a 256-thread intra-op pool each would measure the OpenMP spin-waits)
    ds = data.Hypersim(root_dir, transform=True, split_path=split_path)
    torch.manual_seed(0)
    loader = data.DeviceLoader(ds, batch_size=args.batch, device="cpu", shuffle=True, drop_last=True, workers=args.workers, prefetch=3, rank=args.rank_worker, world=args.ranks)
    pool = ThreadPoolExecutor(args.workers)

    def epoch():
        n, pending = 0, None
        for idx in loader.index_batches():
            futs = [pool.submit(ds.__getitem__, i) for i in idx]
            if pending is not None:
                loader._stage([f.result() for f in pending])
                n += len(pending)
            pending = futs
        if pending is not None:
            loader._stage([f.result() for f in pending])
            n += len(pending)
        return n

    epoch()
    t_go = float(os.environ["LOADER_BENCH_GO"])      # crude start barrier: every rank sleeps until the same wall-clock second
    time.sleep(max(0.0, t_go - time.time()))
    t0 = time.perf_counter()
    n = sum(epoch() for _ in range(args.epochs))
    print(json.dumps({"rank": args.rank_worker, "images": n, "seconds": time.perf_counter() - t0, "late_start_s": max(0.0, time.time() - t_go - (time.perf_counter() - t0))}))
    pool.shutdown()


def run_ranks(args, root_dir, split_path):
    import subprocess
    env = dict(os.environ, LOADER_BENCH_GO=str(time.time() + 60.0), OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")        # imports + the warm-up epoch of every rank fit in 60 s
    cmd = [sys.executable, os.path.abspath(__file__), "--samples", str(args.samples), "--workers", str(args.workers), "--batch", str(args.batch), "--epochs", str(args.epochs),
           "--ranks", str(args.ranks), "--tree", root_dir + "::" + split_path]
    procs = [subprocess.Popen(cmd + ["--rank-worker", str(r)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, env=env) for r in range(args.ranks)]
    res = []
    for pr in procs:
        o, _ = pr.communicate(timeout=600)
        lines = [l for l in o.splitlines() if l.startswith("{")]
        res.append(json.loads(lines[-1]) if lines else {"images": 0, "seconds": 1.0, "failed": True})
    slowest = max(r["seconds"] for r in res)
    return {"ranks": args.ranks, "decode_threads_per_rank": args.workers, "images_per_s": sum(r["images"] for r in res) / slowest,
            "per_rank_images_per_s": [round(r["images"] / r["seconds"], 1) for r in res],
            "late_starts_s": [round(r.get("late_start_s", 0.0), 1) for r in res],
            "what": "HOST side of N ranks (N processes, each DeviceLoader(rank=r, world=N)'s index order, own decode pool, own staging buffers; device part excluded: "
                    "each rank has its own GPU in training, `device_prepare` gives its cost per batch); total images / slowest rank's time"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=24)
    ap.add_argument("--workers", type=int, default=16)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--height", type=int, default=768)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--ranks", type=int, default=1, help="N > 1: N processes, each with its own DeviceLoader(rank=r, world=N) over the same tree (one rank per GPU in "
                    "training; here they share the visible device(s) round-robin) — the aggregate is what N data-parallel ranks are fed with")
    ap.add_argument("--rank-worker", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--tree", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.rank_worker >= 0:
        return rank_worker(args)
    import dataset_fixture as dfx
    from diffusion_e2e_ft_amd import data
    out = {"samples": args.samples, "resolution": [args.height, args.width], "workers": args.workers, "batch": args.batch, "host_cores": os.cpu_count(),
           "consumers_images_per_s": {"8 GPUs x bf16 step (66 images/s each)": 528, "8 GPUs x fp32 recipe (8.8 images/s each)": 70}}
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()
        root_dir, split_path = dfx.make_hypersim_tree(tmp, n=args.samples, H=args.height, W=args.width)
        out["fixture_seconds"] = round(time.perf_counter() - t0, 1)
        ds = data.Hypersim(root_dir, transform=True, split_path=split_path)
        assert len(ds) == args.samples
        # (1) the reference's per-sample CPU work, one core
        torch.set_num_threads(1)
        n_ref = min(8, len(ds))
        reference_cpu_sample(ds.pairs[0])
        t0 = time.perf_counter()
        for i in range(n_ref):
            reference_cpu_sample(ds.pairs[i])
        out["reference_cpu"] = {"images_per_s_per_core": n_ref / (time.perf_counter() - t0), "cores": 1,
                                "what": "Pillow decode + align_normals (numpy) + Pillow resize + torch.quantile + normalisation, as load.py:214-283"}
        # (2) decode only, thread pool
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(args.workers) as ex:
            list(ex.map(ds.__getitem__, range(min(len(ds), args.workers))))
            t0 = time.perf_counter()
            for _ in range(args.epochs):
                list(ex.map(ds.__getitem__, range(len(ds))))
            out["decode_only"] = {"images_per_s": args.epochs * len(ds) / (time.perf_counter() - t0), "threads": args.workers}
        # (3) end to end on the device
        if torch.cuda.is_available():
            dev = torch.device("cuda", 0)
            loader = data.DeviceLoader(ds, batch_size=args.batch, device=dev, shuffle=True, drop_last=True, workers=args.workers, prefetch=3)
            for b in loader:        # warm-up epoch: tables, allocator, kernels
                last = b
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n = 0
            for _ in range(args.epochs):
                for b in loader:
                    n += b["rgb"].shape[0]
                    last = b
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out["device_loader"] = {"images_per_s": n / dt, "batches": n // args.batch, "output": {k: list(v.shape) for k, v in last.items() if isinstance(v, torch.Tensor)}}
            # the device preparation alone (inputs resident)
            samples = [ds[i] for i in range(args.batch)]
            st = loader._stage(samples)
            dv = {k: v.to(dev) for k, v in st.items()}
            flips = [i % 2 == 0 for i in range(args.batch)]
            for _ in range(3):
                data.finish_samples(dv["rgb_u8"], dv["depth"], dv["normal_u8"], "hypersim", flip=flips)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                data.finish_samples(dv["rgb_u8"], dv["depth"], dv["normal_u8"], "hypersim", flip=flips)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            out["device_prepare"] = {"ms_per_batch": ms, "images_per_s": args.batch / ms * 1e3, "what": "align_normals + flip / Pillow-exact resize + quantiles + prepare, batch resident"}
            loader.close()
            if args.ranks > 1:
                out["device_loader_ranks"] = run_ranks(args, root_dir, split_path)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
