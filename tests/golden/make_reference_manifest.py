"""Writes tests/golden/reference_manifest.json: sha256 of every file of the reference tree (/root/reference) that a test imports or exec()s in
process.  tests/conftest.py re-hashes them at collection time and skips the tests that execute reference code when a file differs from what was
reviewed (running third-party source with the developer's privileges deserves at least a pinned content check).  Re-run after
reviewing a changed reference tree:  python tests/golden/make_reference_manifest.py"""
import hashlib
import json
import os

REF = os.environ.get("E2EFT_REFERENCE", "/root/reference")
FILES = [
    "GeoWizard/geowizard/models/unet_2d_condition.py", "GeoWizard/geowizard/models/unet_2d_blocks.py", "GeoWizard/geowizard/models/attention.py",
    "GeoWizard/geowizard/models/transformer_2d.py", "GeoWizard/geowizard/models/geowizard_pipeline.py", "GeoWizard/geowizard/utils/normal_ensemble.py",
    "GeoWizard/geowizard/utils/depth_ensemble.py", "Marigold/marigold/marigold_pipeline.py", "Marigold/marigold/util/ensemble.py",
    "Marigold/marigold/util/batchsize.py", "Marigold/marigold/util/image_util.py", "Marigold/src/util/alignment.py", "Marigold/src/util/metric.py",
    "training/train.py", "training/util/loss.py", "training/util/lr_scheduler.py", "training/util/noise.py", "training/util/unet_prep.py",
    "training/dataloaders/load.py",
]


def digest(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        h.update(f.read())
    return h.hexdigest()


def current():
    out = {}
    for rel in FILES:
        p = os.path.join(REF, rel)
        if os.path.exists(p):
            out[rel] = digest(p)
    return out


if __name__ == "__main__":
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "reference_manifest.json"), "w") as f:
        json.dump(current(), f, indent=1, sort_keys=True)
    print("wrote reference_manifest.json (%d files)" % len(current()))
