"""Host-side mirror of the reference's `trainer` package for the hot path only:
layers / reg / transformer / utils (networks and losses) and the step bodies of
Hd_Trainer_x1/x2, Cyc_Trainer, P2p_Trainer and Reg_Trainer.  Data loading, DICOM export, Visdom logging
and validation metrics are out of scope."""
from .CycTrainer import Cyc_Trainer  # noqa: F401
from .HdTrainer import Hd_Trainer_x, Hd_Trainer_x1, Hd_Trainer_x2  # noqa: F401
from .p2pTrainer import P2p_Trainer  # noqa: F401
from .RegTrainer import Reg_Trainer  # noqa: F401
