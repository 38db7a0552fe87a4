from cta_gan_amd.trainer.reg import *  # noqa: F401,F403
