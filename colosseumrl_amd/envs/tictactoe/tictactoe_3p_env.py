"""3-player 3x5 TicTacToe (3 in a row) -- drop-in for ``colosseumrl.envs.tictactoe.tictactoe_3p_env``
(reference colosseumrl/envs/tictactoe/tictactoe_3p_env.py:102-408)."""
from .tictactoe_base import TicTacToeEnvBase, action_to_string, string_to_action  # noqa: F401


class TicTacToe3PlayerEnv(TicTacToeEnvBase):
    SHAPE = (3, 5)
    PLAYERS = 3
    K = 3
    REL_MOD = 3      # reference 3p:27
