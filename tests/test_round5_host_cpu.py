"""Round-5 host logic that needs no GPU: the algebra of the folded two-token cross-attention (modules.Attention._fold builds its constants with torch only), the
fp32 split-K plan, the build id plumbing.  (The device halves: tests/test_cross_attn_fold_gpu.py, tests/test_train_gpu.py, tests/test_abi.py.)"""
import os

import pytest
import torch


def test_fold_constants_reproduce_two_token_cross_attention_in_float64(monkeypatch):
    """to_out(softmax(q k^T * scale) v) + residual with TWO shared context tokens == c0 + sigmoid(x G^T) Delta^T + residual (attention.py:338-343 semantics).
    Since round 6 the product builds the constants through the library's own fp32 GEMM (ops.gemm: no vendor BLAS in an inference process); there is no GPU here, so
    THIS test swaps that one call for its torch definition — out = a w^T + bias — and checks the algebra; tests/test_cross_attn_fold_gpu.py runs the real thing."""
    from diffusion_e2e_ft_amd import modules as M, ops
    monkeypatch.setattr(ops, "gemm", lambda a, w, bias=None: a @ w.t() + (0 if bias is None else bias))
    torch.manual_seed(0)
    for heads, C in ((5, 320), (2, 128), (20, 1280)):
        att = M.Attention(C, heads=heads, cross_attention_dim=96).double().eval()
        g = torch.Generator().manual_seed(C)
        x = torch.randn(2, 17, C, generator=g, dtype=torch.float64)
        res = torch.randn(2, 17, C, generator=g, dtype=torch.float64)
        ctx1 = torch.randn(1, 2, 96, generator=g, dtype=torch.float64)
        G, Dt, c0 = att._fold(ctx1, torch.float64)
        hp = G.shape[0]
        assert hp % 2 == 0 and hp >= heads and Dt.shape == (C, hp) and c0.shape == (C,)
        assert (G[heads:] == 0).all() and (Dt[:, heads:] == 0).all()              # padded heads: sigmoid(0) = 0.5 times a zero column
        got = torch.sigmoid(x @ G.t()) @ Dt.t() + c0 + res
        d = C // heads
        q = (x @ att.to_q.weight.t()).view(2, 17, heads, d)
        k = (ctx1[0] @ att.to_k.weight.t()).view(2, heads, d)
        v = (ctx1[0] @ att.to_v.weight.t()).view(2, heads, d)
        p = torch.softmax(torch.einsum("bnhd,lhd->bnhl", q, k) * att.scale, dim=-1)
        a = torch.einsum("bnhl,lhd->bnhd", p, v).reshape(2, 17, C)
        want = a @ att.to_out[0].weight.t() + att.to_out[0].bias + res
        assert (got - want).abs().max().item() < 5e-6          # (the constants are built in fp32: round-off of that, not of the algebra)
        # cached per (context state, weight state): same objects again, rebuilt after an in-place change of either
        assert att._fold(ctx1, torch.float64)[0] is G
        with torch.no_grad():
            ctx1.add_(1.0)
        assert att._fold(ctx1, torch.float64)[0] is not G
        G2 = att._fold(ctx1, torch.float64)[0]
        with torch.no_grad():
            att.to_q.weight.mul_(2.0)
        assert torch.allclose(att._fold(ctx1, torch.float64)[0], 2 * G2)


def test_fp32_split_plan_fills_whole_rounds_of_the_256_row_kernel():
    """ops.splitk_plan(dtype=float32): the weight-gradient GEMMs of the fp32 recipe at configs[2] (profiles/r05a_bench_train_fp32_per_shape.tsv: z9 x 30 tiles ran two
    rounds at 53 %) — chunk counts whose tile totals fill their last round of 256 CUs, chunks of >= 512 columns, operands padded to nsplit * kc"""
    from diffusion_e2e_ft_amd import ops
    for (M, N, K) in [(320, 2880, 82944), (2560, 320, 82944), (640, 5760, 20736), (320, 5760, 82944), (320, 320, 82944), (5120, 640, 20736), (640, 640, 20736), (1280, 1280, 5184)]:
        ns, kc = ops.splitk_plan(M, N, K, torch.float32)
        tiles = ((M + 255) // 256) * ((N + 127) // 128) * ns
        eff = tiles / (-(-tiles // 256) * 256)
        assert kc % 64 == 0 and ns * kc >= K and (ns - 1) * kc < K and kc >= 512 and eff >= 0.93, (M, N, K, ns, kc, eff)
        ns16, kc16 = ops.splitk_plan(M, N, K)                    # the 16-bit rule is untouched
        assert kc16 % 64 == 0 and ns16 * kc16 >= K
    assert ops.splitk_plan(1280, 11520, 5184, torch.float32)[0] == 1          # 450 tiles already: no split


def test_build_id_is_read_from_the_binary_and_follows_the_sources(tmp_path):
    from diffusion_e2e_ft_amd import build as b
    sid = b.source_id()
    assert len(sid) == 16 and all(c in "0123456789abcdef" for c in sid) and b.source_id() == sid
    if os.path.exists(b.LIB):
        assert b.built_id() == sid, "the library in the tree was not built from the sources in the tree"
    fake = tmp_path / "lib.so"
    fake.write_bytes(b"\\x7fELF....E2EFT_BUILD_ID=0123456789abcdef\\0....")
    assert b.built_id(str(fake)) == "0123456789abcdef"
    fake.write_bytes(b"no marker here")
    assert b.built_id(str(fake)) is None and b.built_id(str(tmp_path / "missing.so")) is None
