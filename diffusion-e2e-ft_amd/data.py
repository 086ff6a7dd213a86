"""Training input pipeline pieces that belong on the GPU + the dataset mixer.

  MixedDataLoader   /root/reference/training/dataloaders/load.py:18-59 — interleaves two loaders 9:1 (same draw order from
                    numpy's global RNG, same length arithmetic).
  prepare_batch     load.py:236-283 (Hypersim) / :342-375 (Virtual KITTI 2): what __getitem__ does to a decoded + augmented sample,
                    batched on the device: validity mask, 2 % / 98 % depth quantiles, clamp + normalise, normal renormalisation.
  hflip_sample      load.py:76-84: synchronised horizontal flip incl. the sign change of the normals' x component.
  augment_hypersim / augment_vkitti   load.py:67-152: the synchronised transforms (h-flip, PIL-exact bilinear / nearest resize to 480 x 640,
                    ToTensor; h-flip, ToTensor, KITTI benchmark crop 352 x 1216) on batches of DECODED images resident on the device.
  Hypersim / VirtualKITTI2   load.py:160-283 / :285-375: the dataset classes — same constructor arguments, same file discovery (`_find_pairs`), same
                    `len()`; `__getitem__` stops after DECODING (host numpy arrays as the files hold them): everything the reference does to a sample after
                    that — the Hypersim normal-orientation fix, the synchronised transform, the validity mask / quantiles / normalisation — is batched on
                    the device by `DeviceLoader`.
  DeviceLoader      the `torch.utils.data.DataLoader(dataset, shuffle=True, batch_size=...)` of train.py:364-365 for those classes: torch's own sampler
                    (the reference's index order), a thread pool that decodes ahead, pinned staging buffers, upload + device preparation on a side stream;
                    yields the batch dict train.py consumes (train.py:470-475).  `MixedDataLoader(DeviceLoader(hypersim), DeviceLoader(vkitti), 9, 1)` is
                    the reference's training input.
File decoding itself is a callable (`decoder=`): the default uses Pillow when it is importable (every file of both datasets is a PNG / JPEG it reads,
16-bit depth included — the reference's cv2.imread of the KITTI depth returns the same integers); the package does not import an image library otherwise."""
import csv
import os
import random
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import ops

NEAR_FAR = {"hypersim": (1e-5, 65.0), "vkitti": (1e-5, 80.0)}   # load.py:161, :286
DOMAIN = {"hypersim": "indoor", "vkitti": "outdoor"}            # load.py:283, :375


class MixedDataLoader:
    def __init__(self, loader1, loader2, split1=9, split2=1):
        self.loader1, self.loader2, self.split1, self.split2 = loader1, loader2, split1, split2
        self.frac1, self.frac2 = self.get_split_fractions()
        self.randchoice1 = None

    def get_split_fractions(self):   # load.py:34-41: use all of the scarcer loader, subsample the other to keep split1:split2
        n1, n2 = len(self.loader1), len(self.loader2)
        return min((n2 / n1) * (self.split1 / self.split2), 1), min((n1 / n2) * (self.split2 / self.split1), 1)

    def create_split(self):          # load.py:43-46
        choice = [True] * int(len(self.loader1) * self.frac1) + [False] * int(len(self.loader2) * self.frac2)
        np.random.shuffle(choice)
        return choice

    def __iter__(self):
        self.loader_iter1, self.loader_iter2 = iter(self.loader1), iter(self.loader2)
        self.randchoice1 = self.create_split()
        self.indx = 0
        return self

    def __next__(self):
        if self.indx == len(self.randchoice1):
            raise StopIteration
        first = self.randchoice1[self.indx]
        self.indx += 1
        return next(self.loader_iter1 if first else self.loader_iter2)

    def __len__(self):
        return int(len(self.loader1) * self.frac1) + int(len(self.loader2) * self.frac2)


@ops.tensor_scoped
def hflip_sample(rgb01, depth, normal01=None):
    """synchronised horizontal flip of [..., H, W] tensors; the normal map's x channel (index 0 of dim -3) becomes 1 - x, the float
    form of `255 - x` on the uint8 image (load.py:79-84)."""
    rgb01, depth = rgb01.flip(-1), depth.flip(-1)
    if normal01 is None:
        return rgb01, depth
    n = normal01.flip(-1).clone()
    n[..., 0, :, :] = 1.0 - n[..., 0, :, :]
    return rgb01, depth, n


@torch.no_grad()
@ops.tensor_scoped
def prepare_batch(rgb01, depth, normal01, dataset="hypersim", near_plane=None, far_plane=None):
    """rgb01, normal01: [B,3,H,W] in [0,1] (ToTensor of the decoded images); depth: [B,1,H,W] metres -> the batch dict train.py consumes
    (train.py:470-475): rgb [-1,1], depth [B,3,H,W] in [-1,1], metric [B,1,H,W], normals unit / zero, val_mask bool, domain."""
    near, far = NEAR_FAR[dataset]
    near = near if near_plane is None else near_plane
    far = far if far_plane is None else far_plane
    f = lambda t: t.float().contiguous()
    rgb01, depth, normal01 = f(rgb01), f(depth), f(normal01)
    q = ops.masked_quantiles(depth, near, far, 0.02, 0.98)
    rgb, depth3, metric, normals, mask = ops.prepare_sample(rgb01, depth, normal01, near, far, q)
    return {"rgb": rgb, "depth": depth3, "metric": metric, "normals": normals, "val_mask": mask, "domain": [DOMAIN[dataset]] * rgb.shape[0]}


# ---- synchronised augmentation on the device (csrc/dataaug.hip) ---------------------------------------------------------------------
_PIL_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc (src/libImaging/Resample.c) for the bilinear (triangle) filter over the full input
    range: bounds int32 [out, 2] = (first tap, tap count), coefficients int32 [out, ksize] in 22-bit fixed point.  float64 arithmetic in
    Pillow's order, so the integer tables are the ones Pillow builds and the device resize (aug_resample_h/v kernels) is bit-exact."""
    import math
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            v = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - v if v < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        kk[xx] = [int(-0.5 + v * (1 << _PIL_PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << _PIL_PRECISION_BITS)) for v in w]
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def pil_nearest_map(in_size, out_size):
    """source index of every output index for Image.resize(..., NEAREST): Pillow's ImagingScaleAffine walks xo = scale / 2, xo += scale in
    double precision and truncates — the ACCUMULATED sum, so that exact half-way positions (768 -> 480: (7 + 0.5) * 1.6 = 12) land where
    Pillow's rounding error puts them (11)"""
    a0 = float(in_size) / out_size
    steps = np.full(out_size, a0, dtype=np.float64)
    steps[0] = a0 * 0.5
    xo = np.add.accumulate(steps)          # sequential double additions, as the C loop
    return np.clip(xo.astype(np.int32), 0, in_size - 1)


_TABLES = {}


def _tables(kind, in_size, out_size, device):
    key = (kind, in_size, out_size, str(device))
    if key not in _TABLES:
        if kind == "bilinear":
            b, k = pil_bilinear_coeffs(in_size, out_size)
            _TABLES[key] = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device))
        elif kind == "nearest":
            _TABLES[key] = torch.from_numpy(pil_nearest_map(in_size, out_size)).to(device)
        else:   # ("crop", offset): in_size is the offset here
            _TABLES[key] = (in_size + torch.arange(out_size, dtype=torch.int32)).to(device)
    return _TABLES[key]


def _flip_flags(flip, batch, device):
    if flip is None:
        return None
    return torch.as_tensor(flip, dtype=torch.uint8).reshape(batch).to(device).contiguous()


@torch.no_grad()
@ops.tensor_scoped
def augment_hypersim(rgb_u8, depth, normal_u8=None, size=(480, 640), flip=None):
    """SynchronizedTransform_Hyper (load.py:67-101) on a batch: rgb_u8 / normal_u8 uint8 [B,H0,W0,3] as decoded, depth fp32 [B,H0,W0]; flip:
    per-image booleans (the reference draws `random.random() > 0.5` per sample).  Returns rgb01 [B,3,h,w], depth [B,1,h,w], normal01 or None
    — the inputs of prepare_batch."""
    dev = rgb_u8.device
    B, H0, W0, _ = rgb_u8.shape
    h, w = size
    fl = _flip_flags(flip, B, dev)
    xt, yt = _tables("bilinear", W0, w, dev), _tables("bilinear", H0, h, dev)
    rgb01 = ops.aug_resample_bilinear_u8(rgb_u8.contiguous(), (h, w), xt, yt, flip=fl)
    d = ops.aug_gather(depth.float().contiguous(), _tables("nearest", H0, h, dev), _tables("nearest", W0, w, dev), flip=fl)[:, None]
    n01 = None if normal_u8 is None else ops.aug_resample_bilinear_u8(normal_u8.contiguous(), (h, w), xt, yt, flip=fl, invert_x_on_flip=True)
    return rgb01, d, n01


KB_CROP = (352, 1216)     # KITTI benchmark crop (load.py:112-131)


@torch.no_grad()
@ops.tensor_scoped
def augment_vkitti(rgb_u8, depth, normal_u8=None, flip=None):
    """SynchronizedTransform_VKITTI (load.py:104-152): h-flip, ToTensor, crop the bottom 352 rows / centred 1216 columns"""
    dev = rgb_u8.device
    B, H0, W0, _ = rgb_u8.shape
    ch, cw = KB_CROP
    top, left = int(H0 - ch), int((W0 - cw) / 2)
    fl = _flip_flags(flip, B, dev)
    ym, xm = _tables("crop", top, ch, dev), _tables("crop", left, cw, dev)
    rgb01 = ops.aug_gather(rgb_u8.contiguous(), ym, xm, flip=fl)
    d = ops.aug_gather(depth.float().contiguous(), ym, xm, flip=fl)[:, None]
    n01 = None if normal_u8 is None else ops.aug_gather(normal_u8.contiguous(), ym, xm, flip=fl, invert_x_on_flip=True)
    return rgb01, d, n01


# ---- dataset classes (decode on the host) + the device-side loader -----------------------------------------------------------------------------------
def pil_decoder(path, kind):
    """default `decoder`: kind "rgb" / "normal" -> uint8 [H,W,3] (`Image.open(p).convert('RGB')`, load.py:217,224,327,333), "depth" -> the integer
    array as stored (uint16 millimetres / centimetres; `np.array(Image.open(p))` load.py:220-221, `cv2.imread(p, ANYCOLOR | ANYDEPTH)` :329)"""
    from PIL import Image
    with Image.open(path) as im:
        if kind == "depth":
            return np.array(im)
        return np.array(im.convert("RGB"))


class _DecodedDataset:
    """base: `pairs` (file triplets), `decoder`, the per-dataset constants DeviceLoader needs"""
    name = None

    def __len__(self):
        return len(self.pairs)

    def _decode(self, rgb_path, depth_path, normal_path):
        rgb = np.ascontiguousarray(self.decoder(rgb_path, "rgb"))
        normal = np.ascontiguousarray(self.decoder(normal_path, "normal"))
        return rgb, self.decoder(depth_path, "depth"), normal


class Hypersim(_DecodedDataset):
    """load.py:160-283.  root_dir / transform / near_plane / far_plane as the reference; `split_path` (the reference hard-codes the relative path below)
    and `decoder` are additions.  `__getitem__(i)` -> {"rgb_u8" uint8 [768,1024,3], "depth" fp32 [768,1024] metres, "normal_u8" uint8 [768,1024,3]}:
    decoded, depth converted as the reference converts it (uint16 / 1000 in float64, stored as a float32 PIL image: load.py:220-222)."""
    name = "hypersim"
    FOCAL = 886.81                   # load.py:230

    def __init__(self, root_dir, transform=True, near_plane=1e-5, far_plane=65.0, split_path=None, decoder=None):
        self.root_dir = root_dir
        self.split_path = split_path or os.path.join("data/hypersim/processed/train/filename_meta_train.csv")
        self.near_plane, self.far_plane = near_plane, far_plane
        self.align_cam_normal = True
        self.decoder = decoder or pil_decoder
        self.pairs = self._find_pairs()
        self.transform = (480, 640) if transform else None          # SynchronizedTransform_Hyper(H=480, W=640)

    def _find_pairs(self):           # load.py:170-183
        pairs = []
        with open(self.split_path, newline="") as f:
            for row in csv.DictReader(f):
                if str(row["included_in_public_release"]).strip().lower() not in ("true", "1") or row["split_partition_name"] != "train":
                    continue
                rgb_path = os.path.join(self.root_dir, "train", row["rgb_path"])
                depth_path = os.path.join(self.root_dir, "train", row["depth_path"])
                head, _ = os.path.split(os.path.join(self.root_dir, "train"))
                normal_path = os.path.join(head, "normals", row["scene_name"], "images", "scene_%s_geometry_preview" % row["camera_name"],
                                           "frame.%s.normal_cam.png" % str(int(row["frame_id"])).zfill(4))
                if os.path.exists(rgb_path) and os.path.exists(depth_path) and os.path.exists(normal_path):
                    pairs.append({"rgb_path": rgb_path, "depth_path": depth_path, "normal_path": normal_path})
        return pairs

    @staticmethod
    def inverse_intrinsics(H, W):    # load.py:194-198 with K = [886.81, 886.81, W / 2, H / 2] (:230)
        K = np.array([[Hypersim.FOCAL, 0, W / 2], [0, Hypersim.FOCAL, H / 2], [0, 0, 1]])
        return np.linalg.inv(K)

    def __getitem__(self, idx):
        pr = self.pairs[idx]
        rgb, depth, normal = self._decode(pr["rgb_path"], pr["depth_path"], pr["normal_path"])
        return {"rgb_u8": rgb, "depth": np.ascontiguousarray((depth / 1000).astype(np.float32)), "normal_u8": normal}


class VirtualKITTI2(_DecodedDataset):
    """load.py:285-375.  `__getitem__(i)` -> {"rgb_u8" uint8 [375,1242,3], "depth" fp32 [375,1242] metres (`.astype(np.float32) / 100.0`, :330), "normal_u8"}"""
    name = "vkitti"

    def __init__(self, root_dir, transform=None, near_plane=1e-5, far_plane=80.0, decoder=None):
        self.root_dir = root_dir
        self.near_plane, self.far_plane = near_plane, far_plane
        self.decoder = decoder or pil_decoder
        self.pairs = self._find_pairs()
        self.transform = "kitti_benchmark_crop" if transform else None        # SynchronizedTransform_VKITTI()

    def _find_pairs(self):           # load.py:294-318
        pairs = []
        roots = [os.path.join(self.root_dir, d) for d in ("vkitti_2.0.3_rgb", "vkitti_2.0.3_depth", "vkitti_DAG_normals")]
        for scene in ("Scene01", "Scene02", "Scene06", "Scene18", "Scene20"):
            for weather in ("morning", "fog", "rain", "sunset", "overcast"):
                for camera in ("Camera_0", "Camera_1"):
                    rgb_dir, depth_dir, normal_dir = (os.path.join(r, scene, weather, "frames", k, camera) for r, k in zip(roots, ("rgb", "depth", "normal")))
                    if os.path.exists(rgb_dir) and os.path.exists(depth_dir):
                        for f in sorted(os.listdir(rgb_dir)):      # (the reference walks os.listdir's order, which differs between file systems; sorted: reproducible index -> file map)
                            if f.endswith(".jpg"):
                                stem = f[3:]
                                pairs.append((os.path.join(rgb_dir, "rgb" + stem), os.path.join(depth_dir, "depth" + stem.replace(".jpg", ".png")),
                                              os.path.join(normal_dir, "normal" + stem.replace(".jpg", ".png"))))
        return pairs

    def __getitem__(self, idx):
        rgb, depth, normal = self._decode(*self.pairs[idx])
        return {"rgb_u8": rgb, "depth": np.ascontiguousarray(depth.astype(np.float32) / 100.0), "normal_u8": normal}


@torch.no_grad()
@ops.tensor_scoped
def finish_samples(rgb_u8, depth, normal_u8, dataset, flip=None, transform=True, near_plane=None, far_plane=None, align=True):
    """everything `__getitem__` does after decoding (load.py:225-283 / :336-375), batched on the device: rgb_u8 / normal_u8 uint8 [B,H0,W0,3], depth fp32
    [B,H0,W0] metres, flip = per-sample booleans -> the batch dict.  Hypersim: normals turned towards the camera on the full-resolution image
    (e2eft_align_normals_u8), then flip + Pillow-exact resize to 480 x 640; Virtual KITTI 2: flip + KITTI benchmark crop; then prepare_batch."""
    if dataset == "hypersim":
        if align:
            H0, W0 = rgb_u8.shape[1:3]
            normal_u8 = ops.align_normals_u8(normal_u8.contiguous(), depth.float().contiguous(), Hypersim.inverse_intrinsics(H0, W0).reshape(-1))
        if transform:
            rgb01, d, n01 = augment_hypersim(rgb_u8, depth, normal_u8, size=(480, 640), flip=flip)
        else:
            rgb01, d, n01 = _to_tensor(rgb_u8), depth.float()[:, None].contiguous(), _to_tensor(normal_u8)
    else:
        if transform:
            rgb01, d, n01 = augment_vkitti(rgb_u8, depth, normal_u8, flip=flip)
        else:
            rgb01, d, n01 = _to_tensor(rgb_u8), depth.float()[:, None].contiguous(), _to_tensor(normal_u8)
    return prepare_batch(rgb01, d, n01, dataset, near_plane, far_plane)


def _to_tensor(u8):
    """transforms.ToTensor() of a uint8 HWC image batch (the reference's `transform=None` branch): identity gather through the same kernel"""
    B, H, W, _ = u8.shape
    dev = u8.device
    return ops.aug_gather(u8.contiguous(), _tables("crop", 0, H, dev), _tables("crop", 0, W, dev))


class DeviceLoader:
    """`torch.utils.data.DataLoader(dataset, shuffle=True, batch_size=B, num_workers=0)` (train.py:364-365) for Hypersim / VirtualKITTI2 above, with the
    work split MI355X-first: indices from torch's own RandomSampler / BatchSampler (the reference's order for the same torch RNG state), the flip coin
    per sample from Python's `random` in sample order (load.py:76,134; for ONE loader: see `__iter__`), files decoded by `workers` threads `prefetch` batches ahead, one pinned staging
    buffer set per in-flight batch, upload + `finish_samples` on a side stream so that both hide under the training step.  Iterating yields batch dicts
    on `device` (the consumer's current stream waits for the batch's event)."""

    def __init__(self, dataset, batch_size=1, device="cuda", shuffle=True, drop_last=False, workers=8, prefetch=2, rank=0, world=1):
        self.dataset, self.batch_size, self.device = dataset, int(batch_size), torch.device(device)
        self.shuffle, self.drop_last, self.workers, self.prefetch = shuffle, drop_last, max(1, int(workers)), max(1, int(prefetch))
        # data-parallel training: one DeviceLoader per rank over the SAME epoch order (same torch RNG state on every rank, as `set_seed(args.seed)` leaves it,
        # train.py:117-118); rank r takes batches r, r + world, ... of it — accelerate's BatchSamplerShard rule for a prepared DataLoader.  Each rank decodes
        # only its own batches with its own thread pool and staging buffers: host throughput scales with the ranks (scripts/loader_bench.py --ranks)
        self.rank, self.world = int(rank), max(1, int(world))
        assert 0 <= self.rank < self.world
        from torch.utils.data import BatchSampler, RandomSampler, SequentialSampler
        self._batches = BatchSampler(RandomSampler(dataset) if shuffle else SequentialSampler(dataset), self.batch_size, drop_last)
        self._pool = None
        self._stream = None
        self._lock = threading.Lock()

    def __len__(self):
        n = len(self._batches)
        return n if self.world == 1 else (n - self.rank + self.world - 1) // self.world

    def _stage(self, samples):
        """stack decoded samples into pinned host buffers (one set per call: the upload is asynchronous)"""
        pin = self.device.type == "cuda"
        out = {}
        for key, dt in (("rgb_u8", torch.uint8), ("depth", torch.float32), ("normal_u8", torch.uint8)):
            first = samples[0][key]
            buf = torch.empty((len(samples),) + tuple(first.shape), dtype=dt, pin_memory=pin)
            for i, smp in enumerate(samples):
                if smp[key].shape != first.shape:
                    raise ValueError("DeviceLoader: samples of one batch differ in size (%s vs %s)" % (smp[key].shape, first.shape))
                buf[i] = torch.from_numpy(smp[key])
            out[key] = buf
        return out

    def _finish(self, staged, flips):
        ds = self.dataset
        if self.device.type != "cuda":
            raise RuntimeError("DeviceLoader prepares batches with libe2eft kernels: it needs a HIP device (got %s)" % self.device)
        if self._stream is None:
            self._stream = torch.cuda.Stream(self.device)
        with torch.cuda.stream(self._stream):
            dev = {k: v.to(self.device, non_blocking=True) for k, v in staged.items()}
            batch = finish_samples(dev["rgb_u8"], dev["depth"], dev["normal_u8"], ds.name, flip=flips if ds.transform else None,
                                   transform=bool(ds.transform), near_plane=ds.near_plane, far_plane=ds.far_plane, align=getattr(ds, "align_cam_normal", False))
            ev = torch.cuda.Event()
            ev.record(self._stream)
        return batch, ev, staged

    def index_batches(self):
        """the index lists of one epoch, consuming torch's global RNG exactly as `iter(DataLoader(...))` does: the iterator first draws its `base_seed`
        (an int64 it hands to worker processes; drawn with num_workers = 0 too), then RandomSampler draws the seed of its permutation"""
        torch.empty((), dtype=torch.int64).random_()
        it = iter(self._batches)
        if self.world == 1:
            return it
        return (b for k, b in enumerate(it) if k % self.world == self.rank)

    def __iter__(self):
        """`iter(loader)` draws the iterator's `base_seed` EAGERLY, as `DataLoader.__iter__` does (torch's `_BaseDataLoaderIter.__init__`); the permutation seed is
        drawn by the sampler at the first `next()`.  So `MixedDataLoader`'s `iter(loader1), iter(loader2)` followed by interleaved `next()` calls consumes torch's
        RNG in the reference's order.  What does NOT carry over to mixed loaders is the flip coin: the reference draws it inside `__getitem__` at
        `next()` time from the global `random`, this loader draws it `prefetch` batches ahead — a single loader reproduces the reference's sequence, two
        interleaved loaders reproduce it only in distribution."""
        it = self.index_batches()
        return self._generate(it)

    def _generate(self, it):
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="e2eft-decode")
        pending = []          # [(futures of one batch, flips)]

        def submit():
            idx = next(it, None)
            if idx is None:
                return False
            # the reference draws the coin inside the transform, i.e. once per sample in sample order, only when a transform exists (load.py:76,134)
            flips = [random.random() > 0.5 for _ in idx] if self.dataset.transform else None
            pending.append(([self._pool.submit(self.dataset.__getitem__, i) for i in idx], flips))
            return True

        for _ in range(self.prefetch):
            if not submit():
                break
        ready = None          # the batch whose upload + preparation is already in flight on the side stream
        while pending or ready is not None:
            nxt = None
            if pending:
                futs, flips = pending.pop(0)
                submit()
                nxt = self._finish(self._stage([f.result() for f in futs]), flips)
            if ready is not None:
                batch, ev, staged = ready
                torch.cuda.current_stream(self.device).wait_event(ev)
                for v in batch.values():
                    if isinstance(v, torch.Tensor):
                        v.record_stream(torch.cuda.current_stream(self.device))
                yield batch
                del staged
            ready = nxt

    def close(self):
        if self._pool is not None:
            self._pool.shutdown(wait=False, cancel_futures=True)
            self._pool = None
