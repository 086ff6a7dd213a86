"""diffusion-e2e-ft on MI355X: hand-written HIP (gfx950) kernels behind a C ABI (libe2eft.so) for the
single-step Marigold / GeoWizard denoising path, with the reference's module / pipeline surface on top."""
__version__ = "0.1.0"
