"""4-player 3x3x3 TicTacToe -- drop-in for ``colosseumrl.envs.tictactoe.tictactoe_4p_env``
(reference colosseumrl/envs/tictactoe/tictactoe_4p_env.py:131-438)."""
from .tictactoe_base import TicTacToeEnvBase, action_to_string, string_to_action  # noqa: F401


class TicTacToe4PlayerEnv(TicTacToeEnvBase):
    SHAPE = (3, 3, 3)
    PLAYERS = 4
    K = 3
    REL_MOD = 3      # the reference's 4-player observation reduces ids modulo 3 (4p:50); kept as is
