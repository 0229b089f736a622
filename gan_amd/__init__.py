"""gan_amd: MI355X-native (gfx950) StyleGAN2 training hot path behind the module API of gau-nernst/gan.

Public surface mirrors the reference's files:
    gan_amd.modelling            build_generator / build_discriminator       (modelling/__init__.py)
    gan_amd.modelling.stylegan2  ModulatedConv2d, StyleGan2Generator(Block)  (modelling/stylegan2.py)
    gan_amd.modelling.nvidia_ops upfirdn2d, UpFIRDn2d, Blur, bias_leaky_relu (modelling/nvidia_ops.py)
    gan_amd.losses               get_loss, get_regularizer, R1, WGAN-GP      (losses.py)
    gan_amd.utils                EMA, make_parser, prepare_model             (utils.py)
    gan_amd.train_celeba         TrainConfig, train_step                     (train_celeba.py:111-149)
    gan_amd.distributed          data-parallel gradient buckets on RCCL      (new: reference is single-GPU)

All tensor math on the hot path runs in hand-written HIP kernels (gan_amd/csrc -> libgan_amd.so,
C ABI in include/gan_amd.h).  There is no CPU or PyTorch fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
