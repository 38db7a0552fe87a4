from cta_gan_amd.trainer.utils import *  # noqa: F401,F403
