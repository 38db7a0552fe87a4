"""MI355X-native drop-in for the reference's `Model/CycleGan.py` (:6-103): the same three classes as
`Model/HdGan.py` (the reference files are duplicates of each other), HIP-backed."""
from .HdGan import Discriminator, Generator, ResidualBlock  # noqa: F401
