from cta_gan_amd.trainer.transformer import *  # noqa: F401,F403
