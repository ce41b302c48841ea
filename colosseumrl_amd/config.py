"""Environment registry -- same names and lazy lookup as reference colosseumrl/config.py:37-77."""
from typing import Callable, Dict, List, Type

from .BaseEnvironment import BaseEnvironment


def _blokus() -> Type[BaseEnvironment]:
    from .envs.blokus.BlokusEnvironment import BlokusEnvironment
    return BlokusEnvironment


def _tron() -> Type[BaseEnvironment]:
    from .envs.tron.TronGridEnvironment import TronGridEnvironment
    return TronGridEnvironment


def _tic_tac_toe(n: int) -> Callable[[], Type[BaseEnvironment]]:
    def pick() -> Type[BaseEnvironment]:
        from .envs import tictactoe
        table = {2: "TicTacToe2PlayerEnv", 3: "TicTacToe3PlayerEnv", 4: "TicTacToe4PlayerEnv"}
        if n not in table:
            raise ValueError("No Tic Tac Toe with {} players".format(n))
        return getattr(tictactoe, table[n])
    return pick


# 'test' (the networking smoke-test guessing game, reference config.py:16-18) is outside the hot path
# and not provided; every accelerated game keeps its reference name.
ENVIRONMENT_CLASSES: Dict[str, Callable[[], Type[BaseEnvironment]]] = {
    "blokus": _blokus,
    "tron": _tron,
    "tictactoe": _tic_tac_toe(2),
    "tictactoe_3p": _tic_tac_toe(3),
    "tictactoe_4p": _tic_tac_toe(4),
}


def get_environment(environment: str) -> Type[BaseEnvironment]:
    """Environment class by registry name (reference config.py:47-62)."""
    return ENVIRONMENT_CLASSES[environment]()


def available_environments() -> List[str]:
    """Registry names (reference config.py:65-74)."""
    return list(ENVIRONMENT_CLASSES.keys())
