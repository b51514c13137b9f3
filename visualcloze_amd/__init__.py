"""visualcloze_amd — MI355X-native denoising path for VisualCloze (gfx950 HIP kernels behind the
reference's `Flux.forward` / `Sampler.sample_ode` call surface).  See DESIGN.md."""
__version__ = "0.1.0"
