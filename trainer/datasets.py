from cta_gan_amd.trainer.datasets import *  # noqa: F401,F403
