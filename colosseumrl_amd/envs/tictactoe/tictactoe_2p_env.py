"""2-player 3x3 TicTacToe -- drop-in for ``colosseumrl.envs.tictactoe.tictactoe_2p_env``
(reference colosseumrl/envs/tictactoe/tictactoe_2p_env.py:101-407)."""
from .tictactoe_base import TicTacToeEnvBase, action_to_string, string_to_action  # noqa: F401


class TicTacToe2PlayerEnv(TicTacToeEnvBase):
    SHAPE = (3, 3)
    PLAYERS = 2
    K = 3
    REL_MOD = 2      # reference 2p:27
