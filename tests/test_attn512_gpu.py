"""csrc/attn512.hip — the fused attention of the VAE mid block (one 512-wide head; GeoWizard/geowizard/models/unet_2d_blocks.py:589-601) — against
torch CPU fp32 `scaled_dot_product_attention` at sizes the CPU finishes instantly: ragged query / key counts (masked last key tile, partial
query blocks, out-of-range DMA rows), one tile, an odd and an even number of tiles, row-strided q / k / v (slices of one fused projection), and
the inputs that exercise the data-dependent parts of the kernel: a dominant key LATE in the sequence (the deferred rescale of the 256 accumulator
registers must fire in the middle of the run), scores that creep up by less than the threshold per tile (the reference maximum lags behind, the
probabilities exceed 1), and queries whose maximum moves at DIFFERENT tiles inside one wave (per-lane alpha, wave-uniform branch).
The full-size cases (9216 tokens, batch 8) are in tests/test_fullsize_parity_gpu.py and tests/test_benchmarked_configs_gpu.py."""
import pytest
import torch
import torch.nn.functional as TF

from util import rel_err

pytestmark = pytest.mark.gpu
TOL = {torch.float16: 4e-3, torch.bfloat16: 3e-2}


def _ref(q, k, v, scale):
    return TF.scaled_dot_product_attention(q.float()[:, None], k.float()[:, None], v.float()[:, None], scale=scale)[:, 0]


def _run(dev, dtype, B, Nq, Nk, seed, tweak=None, strided=True, scale=512 ** -0.5):
    from diffusion_e2e_ft_amd import ops
    g = torch.Generator().manual_seed(seed)
    q, k, v = (torch.randn(B, n, 512, generator=g).to(dtype) for n in (Nq, Nk, Nk))
    if tweak is not None:
        tweak(q, k, v)
    ref = _ref(q, k, v, scale)
    if strided and Nq == Nk:     # q | k | v as column slices of one [B, N, 1536] buffer (what the module passes)
        buf = torch.cat([q, k, v], dim=2).to(dev)
        qd, kd, vd = buf[..., :512], buf[..., 512:1024], buf[..., 1024:]
    else:
        qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    out = ops.attention512(qd, kd, vd, scale)
    torch.cuda.synchronize()
    assert torch.isfinite(out.float()).all()
    return rel_err(out.float(), ref)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,Nq,Nk", [(1, 32, 32), (2, 128, 128), (1, 100, 100), (3, 200, 77), (2, 333, 1000), (1, 1024, 1024), (1, 160, 2080)])
def test_attn512_against_sdpa(dev, dtype, B, Nq, Nk):
    e = _run(dev, dtype, B, Nq, Nk, seed=Nq * 7 + Nk)
    assert e <= TOL[dtype], e


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_attn512_late_dominant_key_forces_the_rescale(dev, dtype):
    """key 700 (tile 21 of 32) scores ~ +40 in the exponent domain against query 5 and ~ +25 against query 37: the reference maximum of those
    queries must move in the middle of the run and everything accumulated before has to be rescaled, for those lanes only"""
    def tweak(q, k, v):
        k[0, 700] = 0.35 * q[0, 5] + 0.2 * q[0, 37]
        k[0, 901] = 0.3 * q[0, 64]                     # another wave, another tile
        v[0, 700] = 3.0
    e = _run(dev, dtype, 1, 128, 1024, seed=11, tweak=tweak)
    assert e <= TOL[dtype], e


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
def test_attn512_scores_creeping_up_below_the_threshold(dev, dtype):
    """every 32-key tile beats the previous one by ~3 in the exponent domain (threshold 6): the reference maximum is updated only every other
    tile and probabilities reach ~2^6 in between — the result must not care"""
    def tweak(q, k, v):
        q[:] = 0
        q[..., 0] = 4.0
        k[:] *= 0.05
        n = k.shape[1]
        k[0, :, 0] = (torch.arange(n) // 32).float() * (3.0 / (4.0 * 512 ** -0.5 * 1.4427)) + 0.01 * torch.randn(n)
    e = _run(dev, dtype, 1, 64, 512, seed=12, tweak=tweak)
    assert e <= TOL[dtype] * 2, e


def test_attn512_rejects_what_it_cannot_do(dev):
    from diffusion_e2e_ft_amd import _lib
    lib = _lib.load()
    d = _lib.AttnDesc()
    d.dtype, d.batch, d.heads, d.nq, d.nk_seg, d.kv_nseg, d.kv_bmod, d.ldq, d.ldk, d.ldv, d.ldo, d.scale = 1, 1, 2, 32, 32, 1, 1, 512, 512, 512, 512, 0.1
    x = torch.zeros(32 * 512, dtype=torch.float16, device=dev)
    import ctypes as C
    p = C.c_void_p(x.data_ptr())
    assert lib.e2eft_attn512_fwd(C.byref(d), p, p, p, p, None, 0, None) == 1 and b"one head" in lib.e2eft_last_error()
    d.heads, d.dtype = 1, 0
    assert lib.e2eft_attn512_fwd(C.byref(d), p, p, p, p, None, 0, None) == 1 and b"dtype" in lib.e2eft_last_error()


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("B,N", [(1, 1536), (3, 2112), (2, 4000)])
def test_attn512_key_split_tail_equals_the_unsplit_result(dev, dtype, B, N):
    """fewer query blocks than CUs (or a partial last round): the tail blocks are cut along the keys and merged from fp32 partials.  Same numbers
    as the unsplit launch to accumulation-order rounding, and right against SDPA; ragged key chunks and a ragged last query block included."""
    from diffusion_e2e_ft_amd import ops, _lib
    import ctypes as C
    d = _lib.AttnDesc()
    d.dtype, d.batch, d.heads, d.nq, d.nk_seg, d.kv_nseg, d.kv_bmod, d.ldq, d.ldk, d.ldv, d.ldo, d.scale = _lib.dtype_id(dtype), B, 1, N, N, 1, B, 512, 512, 512, 512, 0.05
    assert _lib.load().e2eft_attn512_workspace_bytes(C.byref(d)) > 0, "this shape was meant to exercise the split"
    g = torch.Generator().manual_seed(N)
    q, k, v = (torch.randn(B, N, 512, generator=g).to(dtype) for _ in range(3))
    k[0, N - 40] = 0.4 * q[0, 3]            # a dominant key in the LAST key chunk: the parts of a query end in different reference frames
    ref = _ref(q, k, v, 512 ** -0.5)
    qd, kd, vd = q.to(dev), k.to(dev), v.to(dev)
    a = ops.attention512(qd, kd, vd, 512 ** -0.5)
    ops.ATTN512_SPLIT_TAIL = False
    try:
        b = ops.attention512(qd, kd, vd, 512 ** -0.5)
    finally:
        ops.ATTN512_SPLIT_TAIL = True
    torch.cuda.synchronize()
    assert rel_err(a.float(), ref) <= TOL[dtype] and rel_err(b.float(), ref) <= TOL[dtype]
    assert rel_err(a.float(), b.float()) <= (2e-3 if dtype == torch.float16 else 1.6e-2)
