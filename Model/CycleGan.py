"""Drop-in for the reference's Model/CycleGan.py (cta_gan_amd.Model.CycleGan)."""
from cta_gan_amd.Model.CycleGan import Discriminator, Generator, ResidualBlock  # noqa: F401
