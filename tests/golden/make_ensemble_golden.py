"""Generate tests/golden/ensemble_golden.pt by running the REFERENCE'S OWN ensembling code on the seeded stacks of
tests/golden_cases.py (run here, on CPU, from the repo root: `python tests/golden/make_ensemble_golden.py`).

  ensemble_depths   imported from /root/reference/Marigold/marigold/util/ensemble.py (numpy / torch / scipy only).
  ensemble_normals  marigold_pipeline.py cannot be imported (diffusers / torchvision are not installed); its ensemble_normals
                    (:58-71) is a re-post of GeoWizard/geowizard/utils/normal_ensemble.py:6-23, which imports — that twin is run.
Only outputs are stored; the tests regenerate the inputs from the seeds."""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DEPTH = "/root/reference/Marigold/marigold/util/ensemble.py"
REF_NORMAL = "/root/reference/GeoWizard/geowizard/utils/normal_ensemble.py"


def _import(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    import warnings
    warnings.simplefilter("ignore")
    dep = _import(REF_DEPTH, "ref_ensemble")
    nrm = _import(REF_NORMAL, "ref_normal_ensemble")
    out = {"depth": {}, "normals": {}}
    for name, case in gc.ENSEMBLE_DEPTH_CASES.items():
        aligned, unc = dep.ensemble_depths(gc.ensemble_depth_stack(**case["stack"]), **case["kw"])
        out["depth"][name] = {"aligned": aligned.clone(), "uncertainty": unc.clone()}
        print(name, tuple(aligned.shape), float(aligned.min()), float(aligned.max()), float(unc.mean()))
    for name, kw in {"n6": {}, "n3": dict(n=3, H=17, W=29, seed=39)}.items():
        out["normals"][name] = nrm.ensemble_normals(gc.ensemble_normal_stack(**kw)).clone()
        print(name, tuple(out["normals"][name].shape))
    torch.save(out, os.path.join(HERE, "ensemble_golden.pt"))


if __name__ == "__main__":
    main()
