"""Decomposition of the loss-gradient error of the bf16 E2E-FT micro-step (profiles/r04_bf16_gradient_noise.md): given the decoder outputs of the fp32 run and of the
bf16 draws of both sides (dumps of scripts/bf16_localise.py), how much of the error of d loss / d est comes from flipped residual signs, how much from the coefficients of
the smooth part (the chain through the least-squares scale / shift, training/util/loss.py:31-47), and what white noise of the same size would do.  CPU only."""
import os
import sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusion_e2e_ft_amd import training
batch = training.synthetic_batch(1, 576, 576, torch.device("cpu"), seed=3)
y = batch["metric"].double().flatten(); m = batch["val_mask"].bool().flatten()
D = {s: torch.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "r04_est_dump_%s.pt" % s)) for s in ("hip", "cpu")}   # written by scripts/bf16_localise.py with BF16_DUMP=... (both sides, instance "oracle")
def coefs(est):
    pr = est.double().mean(1).flatten()
    p = pr.clamp(-1, 1)
    pm, ym = p[m], y[m]
    a00, a01, a11, b0, b1 = (pm*pm).sum(), pm.sum(), torch.tensor(float(m.sum())).double(), (pm*ym).sum(), ym.sum()
    dd = a00*a11 - a01*a01
    s = (a11*b0 - a01*b1)/dd; h = (-a01*b0 + a00*b1)/dd
    r = s*p + h - y
    sg = torch.sign(r) * m
    Gs, Gh = (sg*p).sum(), sg.sum()
    nv = a11
    c0 = s/nv
    c1 = (Gs*(-b1 + 2*s*a01) + Gh*(-b0 + 2*h*a01))/(dd*nv)
    c2 = (Gs*a11 - Gh*a01)/(dd*nv)
    c3 = (-2*s*a11*Gs + Gh*(2*b1 - 2*h*a11))/(dd*nv)
    return dict(p=p, sg=sg, c=(c0, c1, c2, c3), Gs=Gs, Gh=Gh, s=s, h=h, a01=a01, a00=a00, dd=dd)
def grad(c, sg, p):   # d loss / d pred (before the /3 and the clamp mask)
    return (c[0]*sg + (c[1] + c[2]*y + c[3]*p)) * m
ef = D["cpu"]["ref"][0]
R = coefs(ef)
g0 = grad(R["c"], R["sg"], R["p"])
smooth0 = ((R["c"][1] + R["c"][2]*y + R["c"][3]*R["p"]) * m)
print("reference: c0 %.3e c1 %.3e c2 %.3e c3 %.3e ; Gs %.1f Gh %.0f s %.5f h %.5f mean p %.4f ; energy of smooth part / total %.3f"
      % (*[float(v) for v in R["c"]], R["Gs"], R["Gh"], R["s"], R["h"], R["a01"]/m.sum(), (smooth0.norm()**2/g0.norm()**2)))
for side in ("cpu", "hip"):
    for i, (e, _) in enumerate(D[side]["draws"][:6]):
        X = coefs(e.float())
        g1 = grad(X["c"], X["sg"], X["p"])
        tot = (g1-g0).norm()/g0.norm()
        g_sign = grad(R["c"], X["sg"], R["p"])            # only the signs changed
        g_coef = grad(X["c"], R["sg"], R["p"])            # only the coefficients changed
        g_p = grad(R["c"], R["sg"], X["p"])               # only p inside the smooth term
        print("%s draw %d: total %.4f | signs only %.4f | coefficients only %.4f (dc1 %.1f%% dc2 %.1f%% dc3 %.1f%%) | p-term only %.4f | dGs %.1f dGh %.0f ds %.2e"
              % (side, i, tot, (g_sign-g0).norm()/g0.norm(), (g_coef-g0).norm()/g0.norm(), 100*(X["c"][1]/R["c"][1]-1), 100*(X["c"][2]/R["c"][2]-1), 100*(X["c"][3]/R["c"][3]-1),
                 (g_p-g0).norm()/g0.norm(), X["Gs"]-R["Gs"], X["Gh"]-R["Gh"], X["s"]-R["s"]))
print("--- synthetic: est_fp32 + white noise of 3.4% relative L2, 12 seeds; and the HIP / torch error fields with random global sign / spatial flips")
import statistics
ef32 = ef.float()
tots, coefsonly = [], []
for seed in range(12):
    gN = torch.Generator().manual_seed(100 + seed)
    noise = torch.randn(ef32.shape, generator=gN)
    noise *= 0.034 * ef32.norm() / noise.norm()
    X = coefs((ef32 + noise).to(torch.bfloat16).float())
    g1 = grad(X["c"], X["sg"], X["p"])
    tots.append(float((g1-g0).norm()/g0.norm())); coefsonly.append(float((grad(X["c"], R["sg"], R["p"])-g0).norm()/g0.norm()))
print("white noise: total %s" % ["%.3f" % t for t in tots]); print("             coefficients only %s" % ["%.3f" % t for t in coefsonly])
for side in ("cpu", "hip"):
    outs = []
    for i, (e, _) in enumerate(D[side]["draws"][:6]):
        de = e.float() - ef32
        for tr, name in ((lambda d: -d, "negated"), (lambda d: d.flip(-1), "mirrored x"), (lambda d: d.flip(-2), "mirrored y"), (lambda d: d.roll(97, -1), "shifted 97 px")):
            X = coefs(ef32 + tr(de))
            g1 = grad(X["c"], X["sg"], X["p"])
            outs.append((name, float((g1-g0).norm()/g0.norm())))
    for name in ("negated", "mirrored x", "mirrored y", "shifted 97 px"):
        v = [o[1] for o in outs if o[0] == name]
        print("%s error fields %-14s: gradient error %s (median %.3f)" % (side, name, ["%.3f" % t for t in v], statistics.median(v)))
