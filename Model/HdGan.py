"""Drop-in for the reference's Model/HdGan.py: same names, MI355X kernels underneath (cta_gan_amd.Model.HdGan)."""
from cta_gan_amd.Model.HdGan import (DataPrefetcher, Discriminator, Discriminator_m, GANLoss, Generator,  # noqa: F401
                                     NLayerDiscriminator, ResidualBlock)
