"""Drop-in for the reference's `trainer` package, hot path only (cta_gan_amd.trainer)."""
from cta_gan_amd.trainer import (Cyc_Trainer, Hd_Trainer_x, Hd_Trainer_x1, Hd_Trainer_x2, P2p_Trainer,  # noqa: F401
                                 Reg_Trainer)
