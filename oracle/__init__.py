"""oracle/ — CPU restatement of the reference's single-step denoising path.  TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import this package; the product
(`diffusion-e2e-ft_amd/`) never does and fails loudly when its HIP library is missing.

What is restated, and from where (paths relative to /root/reference):
  unet_ref.py      SD-v2 `UNet2DConditionModel` forward: wiring from the vendored twin
                   GeoWizard/geowizard/models/unet_2d_condition.py:916-1221, unet_2d_blocks.py:634-777,1027-1273,2201-2481,
                   transformer_2d.py:326-423, attention.py:292-413,425-513,719-777; leaf modules (ResnetBlock2D, Attention,
                   GEGLU, Timesteps, Downsample2D, Upsample2D) from diffusers==0.30.2 (requirements.txt:2 — NOT in the tree,
                   not installable here), restated from their published definitions as plain torch.nn.functional calls.
  vae_ref.py       SD `AutoencoderKL` encoder / decoder / quant convs (diffusers 0.30.2 autoencoder_kl.py, vae.py; block twins
                   unet_2d_blocks.py:509-631,1276-1333,2484-2541).
  pipeline_ref.py  Marigold/marigold/marigold_pipeline.py:372-538 (single_infer, encode_rgb, decode_depth/normal),
                   training/train.py:233-243,470-556 (E2E-FT step forward), DDIM constants (diffusers scheduling_ddim.py),
                   GeoWizard/geowizard/models/geowizard_pipeline.py:252-344.
  losses_ref.py    training/util/loss.py:13-67, lr_scheduler.py:10-36, unet_prep.py:6-20.

Pinning status (round 2)
  * losses_ref / lr schedule / conv_in replacement / pyramid noise / ensembling / MixedDataLoader / alignment + metrics: PINNED — checked against
    the reference's own Python modules imported from /root/reference (tests/test_oracle_pins.py, tests/golden/make_*golden.py).
  * unet_ref / vae_ref / pipeline_ref: PINNED TO THE REFERENCE'S WIRING — tests/test_reference_wiring_cpu.py imports the reference's vendored
    UNet composition code (GeoWizard/geowizard/models/*.py), its two pipelines and its train.py step body, executes them in place and asserts
    equality with this package to fp32 round-off on the same strictly-loaded state dict (SD-v2 and GeoWizard configurations).
    What remains "restated from the published definition" are the third-party LEAF modules of diffusers 0.30.2 (ResnetBlock2D, Attention +
    AttnProcessor2_0, GEGLU, Timesteps, Downsample2D, Upsample2D, DDIMScheduler): not in the tree, not installable here; tests/stubs/ holds
    their plain torch.nn compositions and is replaced by the real package wherever diffusers is installed (tests/refimport.py).
"""
