"""RL-library adapters over the drop-in environment classes (what colosseumrl/envs/wrappers provides in the reference).
gym / ray are optional: see ``spaces`` and ``rllib``."""
from .rllib import RllibWrapper
from .spaces import Box, Dict, Discrete, HAVE_GYM

__all__ = ["RllibWrapper", "Box", "Dict", "Discrete", "HAVE_GYM"]
