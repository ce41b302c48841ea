"""colosseumrl_amd -- MI355X-native vectorised stepper for the colosseumrl board-game environments.

Drop-in surface (same names as the reference package):
    BaseEnvironment, SimpleConfigParser, config.get_environment / available_environments,
    envs.tron.TronGridEnvironment, envs.tictactoe.TicTacToe{2,3,4}PlayerEnv, envs.blokus.BlokusEnvironment
Batched surface (new): batched.TronBatch / TTTBatch / BlokusBatch, parallel.ShardedRollout.
"""
from .BaseEnvironment import BaseEnvironment, SimpleConfigParser
from .config import get_environment, available_environments, ENVIRONMENT_CLASSES

__version__ = "0.1.0"
