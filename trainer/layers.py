from cta_gan_amd.trainer.layers import *  # noqa: F401,F403
