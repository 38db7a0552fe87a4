"""Namespace of the HIP-backed classes with the reference's names (for oracle.golden_cases)."""
from types import SimpleNamespace


def hip_namespace():
    from cta_gan_amd.Model import HdGan as H
    from cta_gan_amd.trainer.reg import Reg
    from cta_gan_amd.trainer.transformer import Transformer_2D
    from cta_gan_amd.trainer.utils import ReplayBuffer, smooothing_loss
    return SimpleNamespace(Generator=H.Generator, ResidualBlock=H.ResidualBlock, Discriminator=H.Discriminator,
                           NLayerDiscriminator=H.NLayerDiscriminator, Discriminator_m=H.Discriminator_m, GANLoss=H.GANLoss, Reg=Reg,
                           Transformer_2D=Transformer_2D, smooothing_loss=smooothing_loss, ReplayBuffer=ReplayBuffer,
                           device="cuda")
