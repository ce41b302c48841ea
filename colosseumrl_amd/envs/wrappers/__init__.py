from .rllib import RllibWrapper
