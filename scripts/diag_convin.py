import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops, _lib
dev = torch.device("cuda")
for dt in (torch.bfloat16, torch.float16):
    for (B, H, W) in ((1, 576, 576), (8, 768, 768), (2, 64, 512)):
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn((B, H, W, 8), generator=g, device=dev).to(dt)
        w = (torch.randn((128, 72), generator=g, device=dev) / 8.5).to(dt)
        b = torch.randn((128,), generator=g, device=dev).to(dt)
        ga, be = torch.ones(128, device=dev).to(dt), torch.zeros(128, device=dev).to(dt)
        _lib.set_option(_lib.OPT_THIN_INPUT_CONV, 1)
        y1 = ops.conv2d(x, w, b, 128, 3, 3, 1, (1, 1, 1, 1), gn_stats=True)
        k1 = _lib.load().e2eft_debug_last_kernel()
        _lib.set_option(_lib.OPT_THIN_INPUT_CONV, 0)
        y0 = ops.conv2d(x, w, b, 128, 3, 3, 1, (1, 1, 1, 1), gn_stats=True)
        _lib.set_option(_lib.OPT_THIN_INPUT_CONV, 1)
        d = (y1.float() - y0.float()).abs()
        n1 = ops.groupnorm(y1, ga, be, 32, 1e-6, True)
        n1o = ops.groupnorm(y1.clone(), ga, be, 32, 1e-6, True)
        n0 = ops.groupnorm(y0, ga, be, 32, 1e-6, True)
        print(dt, (B, H, W), "conv max diff %.3e (mean %.3e, max |y| %.2f)" % (d.max().item(), d.mean().item(), y0.float().abs().max().item()),
              "| GN with thin stats vs own pass: %.3e | GN(thin) vs GN(igemm2): %.3e" % ((n1.float() - n1o.float()).abs().max().item(), (n1.float() - n0.float()).abs().max().item()),
              "nslabs", y1._e2eft_gn.nslabs, y0._e2eft_gn.nslabs, flush=True)
        # where is the conv difference?
        idx = d.flatten().argmax().item()
        print("   argmax at", torch.unravel_index(torch.tensor(idx), d.shape))
