"""CPU tests of the training input pipeline pieces: MixedDataLoader pinned to the reference's own class (load.py imported with its
four unavailable third-party imports stubbed — only names used at import time), synchronised flip, and the oracle's edge cases."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import dataprep_ref

REF = "/root/reference/training/dataloaders/load.py"


def _import_reference_load():
    import importlib.util
    stubs = {}
    for name in ("torchvision", "torchvision.transforms", "PIL", "PIL.Image", "pandas", "cv2"):
        if name not in sys.modules:
            stubs[name] = types.ModuleType(name)
    if "torchvision" in stubs:
        stubs["torchvision"].transforms = stubs["torchvision.transforms"]
    if "PIL" in stubs:
        stubs["PIL"].Image = stubs["PIL.Image"]
    sys.modules.update(stubs)
    try:
        spec = importlib.util.spec_from_file_location("ref_load", REF)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        for k in stubs:
            sys.modules.pop(k, None)
    return m


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present")
@pytest.mark.parametrize("n1,n2,s1,s2", [(90, 10, 9, 1), (50, 50, 9, 1), (300, 7, 9, 1), (13, 200, 3, 2)])
def test_mixed_dataloader_matches_reference_class(n1, n2, s1, s2, capsys):
    from diffusion_e2e_ft_amd.data import MixedDataLoader
    ref = _import_reference_load()
    a, b = [("a", i) for i in range(n1)], [("b", i) for i in range(n2)]
    np.random.seed(123)
    want_loader = ref.MixedDataLoader(a, b, s1, s2)
    want = list(want_loader)
    np.random.seed(123)
    got_loader = MixedDataLoader(a, b, s1, s2)
    got = list(got_loader)
    assert got == want and len(got_loader) == len(want_loader) == len(got)
    assert (got_loader.frac1, got_loader.frac2) == (want_loader.frac1, want_loader.frac2)
    assert list(got_loader) != [] and len(list(got_loader)) == len(want)   # re-iterable, a fresh shuffle per epoch


def test_mixed_dataloader_split_arithmetic():
    from diffusion_e2e_ft_amd.data import MixedDataLoader
    m = MixedDataLoader(list(range(900)), list(range(100)), 9, 1)
    assert (m.frac1, m.frac2, len(m)) == (1, 1, 1000)
    m = MixedDataLoader(list(range(100)), list(range(100)), 9, 1)           # loader2 subsampled to keep 9:1
    assert m.frac1 == 1 and abs(m.frac2 - 1 / 9) < 1e-12 and len(m) == 100 + 11
    np.random.seed(0)
    picks = list(m)
    assert sum(1 for _ in picks) == 111


def test_hflip_sample():
    from diffusion_e2e_ft_amd.data import hflip_sample
    g = torch.Generator().manual_seed(1)
    rgb, d, n = torch.rand(3, 4, 6, generator=g), torch.rand(1, 4, 6, generator=g), torch.rand(3, 4, 6, generator=g)
    r2, d2, n2 = hflip_sample(rgb, d, n)
    assert torch.equal(r2, rgb.flip(-1)) and torch.equal(d2, d.flip(-1))
    assert torch.equal(n2[1:], n.flip(-1)[1:]) and torch.allclose(n2[0], 1 - n.flip(-1)[0])
    u8 = (n * 255).round()                                                    # the uint8 form of the reference: 255 - x
    assert torch.allclose(hflip_sample(rgb, d, u8 / 255)[2][0] * 255, 255 - u8.flip(-1)[0], atol=1e-4)


def test_oracle_edge_cases():
    H, W = 6, 8
    rgb, nrm = torch.rand(3, H, W), torch.rand(3, H, W)
    out = dataprep_ref.prepare_sample_ref(rgb, torch.zeros(1, H, W), nrm, 1e-5, 65.0)           # nothing valid
    assert not out["val_mask"].any() and (out["depth"] == 0).all() and (out["metric"] == 0).all() and (out["normals"] == 0).all()
    out = dataprep_ref.prepare_sample_ref(rgb, torch.full((1, H, W), 3.0), nrm, 1e-5, 65.0)     # constant depth: lo == hi
    assert not out["val_mask"].any() and (out["depth"] == 0).all() and (out["metric"] == 0).all()
    d = torch.linspace(0.5, 70, H * W).view(1, H, W)
    out = dataprep_ref.prepare_sample_ref(rgb, d, nrm, 1e-5, 65.0)
    assert out["val_mask"].sum() == (d < 65).sum() and out["depth"].shape == (3, H, W)
    assert float(out["depth"].min()) == -1.0 and float(out["depth"].max()) == 1.0
    assert (out["metric"][~out["val_mask"]] == out["metric"].max()).all()                       # invalid pixels sit on the far quantile


# ---- device augmentation: the host-built tables are Pillow's (data.pil_bilinear_coeffs / pil_nearest_map) ---------------------------------
def _emulate_resample(img_u8, size):
    """numpy model of csrc/dataaug.hip's two passes (integer arithmetic, 22-bit coefficients, uint8 after each pass)"""
    from diffusion_e2e_ft_amd.data import pil_bilinear_coeffs
    H0, W0, _ = img_u8.shape
    h, w = size
    xb, xk = pil_bilinear_coeffs(W0, w)
    yb, yk = pil_bilinear_coeffs(H0, h)
    half = 1 << 21
    mid = np.zeros((H0, w, 3), dtype=np.uint8)
    for xo in range(w):
        x0, n = xb[xo]
        acc = (img_u8[:, x0:x0 + n, :].astype(np.int64) * xk[xo, :n, None].astype(np.int64)).sum(1) + half
        mid[:, xo, :] = np.clip(acc >> 22, 0, 255)
    out = np.zeros((h, w, 3), dtype=np.uint8)
    for yo in range(h):
        y0, n = yb[yo]
        acc = (mid[y0:y0 + n].astype(np.int64) * yk[yo, :n, None, None].astype(np.int64)).sum(0) + half
        out[yo] = np.clip(acc >> 22, 0, 255)
    return out


@pytest.mark.parametrize("src,dst", [((96, 128), (60, 80)), ((77, 101), (48, 64)), ((40, 56), (60, 84)), ((50, 50), (50, 50)), ((33, 47), (11, 17))])
def test_bilinear_tables_reproduce_pillow_bit_for_bit(src, dst):
    """transforms.Resize((H, W)) on a PIL image is Image.resize(..., BILINEAR) (training/dataloaders/load.py:69,87-90): down- and up-scaling,
    odd sizes, identity"""
    from PIL import Image
    rng = np.random.default_rng(src[0] * 1000 + dst[1])
    img = rng.integers(0, 256, size=(src[0], src[1], 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(img).resize((dst[1], dst[0]), resample=Image.BILINEAR))
    got = _emulate_resample(img, dst)
    assert np.array_equal(got, want), int(np.abs(got.astype(int) - want.astype(int)).max())


@pytest.mark.parametrize("n_in,n_out", [(768, 480), (1024, 640), (101, 64), (64, 101), (375, 352)])
def test_nearest_map_is_pillows(n_in, n_out):
    from PIL import Image
    from diffusion_e2e_ft_amd.data import pil_nearest_map
    ramp = np.arange(n_in, dtype=np.float32)[None, :].repeat(3, 0)
    want = np.asarray(Image.fromarray(ramp, mode="F").resize((n_out, 3), resample=Image.NEAREST))[0].astype(np.int32)
    assert np.array_equal(pil_nearest_map(n_in, n_out), want)
