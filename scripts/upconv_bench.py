"""A/B of the 2x-upsample + 3x3 convolution: fused-upsample form (igemm6) against the four-phase form (e2eft_upconv2x_fwd).
usage: python scripts/upconv_bench.py B H W Cin Cout [iters=10] [dtype=fp16] [option=value ...]     (H x W = the LOW-resolution input)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from diffusion_e2e_ft_amd import ops, autograd as F
import _options

a = _options.take(sys.argv[1:])
B, H, W, Ci, Co = (int(v) for v in a[:5])
iters = int(a[5]) if len(a) > 5 else 10
dt = {"fp16": torch.float16, "bf16": torch.bfloat16}[a[6] if len(a) > 6 else "fp16"]
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn((B, H, W, Ci), generator=g, device=dev).to(dt)
conv = torch.nn.Conv2d(Ci, Co, 3, padding=1).to(dev)
w = F.packed_conv_weight(conv, dt)
b = conv.bias.detach().to(dt)
out = torch.empty((B, 2 * H, 2 * W, Co), dtype=dt, device=dev)
fl = 2.0 * B * 4 * H * W * Co * 9 * Ci            # the layer's nominal work (what the fused form multiplies)
for name, wph in (("fused-upsample 3x3", None), ("four 2x2 phases", lambda: F.phase_conv_weight(conv, dt))):
    for _ in range(3):
        ops.conv2d(x, w, b, Co, 3, 3, 1, (1, 1, 1, 1), up_to=(2 * H, 2 * W), out=out, gn_stats=True, w_phase=wph)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        ops.conv2d(x, w, b, Co, 3, 3, 1, (1, 1, 1, 1), up_to=(2 * H, 2 * W), out=out, gn_stats=True, w_phase=wph)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print("upconv B%d %dx%d->%dx%d %d->%d %s: %-20s %.3f ms  nominal %.1f TFLOP/s%s" % (B, H, W, 2 * H, 2 * W, Ci, Co, a[6] if len(a) > 6 else "fp16", name, ms, fl / ms / 1e9,
                                                                                      "" if wph is None else "  (multiplied: %.1f TFLOP/s)" % (fl * 4 / 9 / ms / 1e9)))
