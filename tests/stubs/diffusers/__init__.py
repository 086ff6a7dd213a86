"""Stub of the `diffusers` top-level namespace (see tests/stubs/README.md).  TEST INFRASTRUCTURE ONLY."""
__version__ = "0.30.2"
IS_E2EFT_TEST_STUB = True

from .configuration_utils import ConfigMixin, register_to_config  # noqa: E402,F401
from .models.modeling_utils import ModelMixin  # noqa: E402,F401
from .pipelines import DiffusionPipeline  # noqa: E402,F401
from .schedulers import DDIMScheduler, DDPMScheduler, LCMScheduler  # noqa: E402,F401
from .models.autoencoders.autoencoder_kl import AutoencoderKL  # noqa: E402,F401


class UNet2DConditionModel:  # the Marigold pipeline only uses this name as a type annotation (marigold_pipeline.py:123)
    def __init__(self, *a, **k):
        raise NotImplementedError("stub: use the reference's vendored geowizard.models.unet_2d_condition.UNet2DConditionModel")
