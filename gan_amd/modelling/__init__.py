"""Model registry (reference: modelling/__init__.py:15-45): same function names, same keys for the
models in scope (DCGAN, Progressive GAN, StyleGAN, StyleGAN2); unknown names raise KeyError like the reference's dict lookup."""
from .dcgan import DcGanDiscriminator, DcGanGenerator
from .progressive_gan import ProgressiveGanDiscriminator, ProgressiveGanGenerator
from .stylegan import StyleGanGenerator
from .stylegan2 import StyleGan2Generator

_GENERATORS = dict(dcgan=DcGanGenerator, progressive_gan=ProgressiveGanGenerator, stylegan=StyleGanGenerator, stylegan2=StyleGan2Generator)
_DISCRIMINATORS = dict(
    dcgan=DcGanDiscriminator,
    progressive_gan=ProgressiveGanDiscriminator,
    stylegan=ProgressiveGanDiscriminator,
    stylegan2=ProgressiveGanDiscriminator,
)


def build_generator(name: str, *args, **kwargs):
    return _GENERATORS[name](*args, **kwargs)


def build_discriminator(name: str, *args, **kwargs):
    return _DISCRIMINATORS[name](*args, **kwargs)
