"""Import the reference's OWN source files from where they lie (/root/reference), with tests/stubs standing in for the third-party
packages this image lacks (diffusers, xformers, torchvision, cv2).  TEST INFRASTRUCTURE ONLY — see tests/stubs/README.md.

    with reference_modules() as ref:
        unet = ref.UNet2DConditionModel(**cfg)            # GeoWizard/geowizard/models/unet_2d_condition.py (vendored diffusers UNet)
        pipe = ref.MarigoldPipeline(unet, vae, ...)       # Marigold/marigold/marigold_pipeline.py
        ref.run_source("training/train.py", 470, 566, ns) # execute a line range of a reference file in a prepared namespace

Inside the context the stub directory sits at the END of sys.path (real packages win where they are installed) and the reference's
package roots at the front; on exit both are removed again and every module that was loaded from either place is dropped from
sys.modules, so the rest of the test session never sees a stub.  Nothing is copied: the reference's code is executed in place.
"""
import contextlib
import importlib
import os
import sys
import textwrap
import types

HERE = os.path.dirname(os.path.abspath(__file__))
STUBS = os.path.join(HERE, "stubs")
REF = os.environ.get("E2EFT_REFERENCE", "/root/reference")
REF_ROOTS = [os.path.join(REF, "GeoWizard"), os.path.join(REF, "Marigold")]


def reference_available():
    return os.path.isdir(os.path.join(REF, "GeoWizard", "geowizard", "models"))


class _Ref(types.SimpleNamespace):
    def run_source(self, relpath, first, last, namespace):
        """exec lines first..last (1-based, inclusive) of a reference file, dedented, in `namespace` — for code that lives inside a
        function body the reference never exposes (training/train.py::main)"""
        path = os.path.join(REF, relpath)
        with open(path) as f:
            lines = f.readlines()[first - 1:last]
        src = textwrap.dedent("".join(lines))
        code = compile("\n" * (first - 1) + src, path, "exec")   # line numbers in tracebacks point into the reference file
        exec(code, namespace)
        return namespace


@contextlib.contextmanager
def reference_modules():
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF)
    # resolve the transformers names the reference imports BEFORE the torchvision stub becomes visible (transformers probes torchvision.io)
    from transformers import CLIPImageProcessor, CLIPTextModel, CLIPTokenizer, CLIPVisionModelWithProjection  # noqa: F401
    before = set(sys.modules)
    saved_path = list(sys.path)
    sys.path[:0] = REF_ROOTS
    sys.path.append(STUBS)
    try:
        ref = _Ref()
        ref.diffusers = importlib.import_module("diffusers")
        ref.uses_stub_diffusers = bool(getattr(ref.diffusers, "IS_E2EFT_TEST_STUB", False))
        ref.unet_2d_condition = importlib.import_module("geowizard.models.unet_2d_condition")
        ref.unet_2d_blocks = importlib.import_module("geowizard.models.unet_2d_blocks")
        ref.attention = importlib.import_module("geowizard.models.attention")
        ref.transformer_2d = importlib.import_module("geowizard.models.transformer_2d")
        ref.geowizard_pipeline = importlib.import_module("geowizard.models.geowizard_pipeline")
        ref.marigold_pipeline = importlib.import_module("marigold.marigold_pipeline")
        ref.UNet2DConditionModel = ref.unet_2d_condition.UNet2DConditionModel
        ref.AutoencoderKL = ref.diffusers.AutoencoderKL
        ref.DDIMScheduler = ref.diffusers.DDIMScheduler
        ref.MarigoldPipeline = ref.marigold_pipeline.MarigoldPipeline
        ref.DepthNormalEstimationPipeline = ref.geowizard_pipeline.DepthNormalEstimationPipeline
        yield ref
    finally:
        sys.path[:] = saved_path
        for name in set(sys.modules) - before:
            f = getattr(sys.modules[name], "__file__", None) or ""
            if f.startswith(STUBS) or f.startswith(REF):
                del sys.modules[name]


# ---- the configurations of oracle/config.py as keyword arguments of the reference's classes ----------------------------------------
def ref_unet_kwargs(cfg):
    kw = dict(sample_size=cfg["sample_size"], in_channels=cfg["in_channels"], out_channels=cfg["out_channels"],
              down_block_types=tuple(cfg["down_block_types"]), up_block_types=tuple(cfg["up_block_types"]),
              block_out_channels=tuple(cfg["block_out_channels"]), layers_per_block=cfg["layers_per_block"],
              attention_head_dim=tuple(cfg["attention_head_dim"]), cross_attention_dim=cfg["cross_attention_dim"],
              norm_num_groups=cfg["norm_num_groups"], norm_eps=cfg["norm_eps"], use_linear_projection=cfg["use_linear_projection"],
              flip_sin_to_cos=cfg["flip_sin_to_cos"], freq_shift=cfg["freq_shift"])
    if cfg.get("class_embed_type"):
        kw.update(class_embed_type=cfg["class_embed_type"], projection_class_embeddings_input_dim=cfg["projection_class_embeddings_input_dim"])
    return kw


def ref_vae_kwargs(cfg):
    n = len(cfg["block_out_channels"])
    return dict(in_channels=cfg["in_channels"], out_channels=cfg["out_channels"], latent_channels=cfg["latent_channels"],
                down_block_types=("DownEncoderBlock2D",) * n, up_block_types=("UpDecoderBlock2D",) * n,
                block_out_channels=tuple(cfg["block_out_channels"]), layers_per_block=cfg["layers_per_block"],
                norm_num_groups=cfg["norm_num_groups"], scaling_factor=cfg["scaling_factor"], sample_size=64)
