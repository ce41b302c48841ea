"""The environment plug-in contract (drop-in for ``colosseumrl.BaseEnvironment``).

Same class name, method names, argument meaning and return shapes as the
reference ABC (reference colosseumrl/BaseEnvironment.py:10-283) so that agents,
``match_server.server_app`` and ``ClientEnvironment`` written against it keep
working when they import this package instead.  ``SimpleConfigParser`` mirrors
reference BaseEnvironment.py:286-341.
"""
from abc import ABC, abstractmethod
from typing import Dict, List, Tuple, Union

import numpy as np


class BaseEnvironment(ABC):
    """A turn-based multi-player game: pure functions from (state, actions) to the next state."""

    def __init__(self, config: str = ""):
        # the reference keeps the raw string; subclasses parse it (BaseEnvironment.py:13-22)
        self._config = config

    # ---- static facts about the game -------------------------------------------------------
    @property
    @abstractmethod
    def min_players(self) -> int:
        """Fewest players a match may start with."""
        raise NotImplementedError

    @property
    @abstractmethod
    def max_players(self) -> int:
        """Most players a match may hold (equal to ``min_players`` for every shipped game)."""
        raise NotImplementedError

    @staticmethod
    @abstractmethod
    def observation_names() -> List[str]:
        """Keys of the dict returned by ``state_to_observation``."""
        raise NotImplementedError

    @property
    @abstractmethod
    def observation_shape(self) -> Dict[str, tuple]:
        """numpy shape of every observation entry, by key."""
        raise NotImplementedError

    # ---- game dynamics ---------------------------------------------------------------------
    @abstractmethod
    def new_state(self, num_players: int = None) -> Tuple[object, List[int]]:
        """Fresh state and the players (numbered 0..n-1) who move first."""
        raise NotImplementedError

    def add_player(self, state: object) -> object:
        """Optional hook for games whose roster can grow; none of ours can (BaseEnvironment.py:96-116)."""
        raise RuntimeError("Cannot add new players to an existing game.")

    def remove_player(self, state: object, player: int) -> object:
        """Optional hook for dropping a player mid-game (BaseEnvironment.py:118-137)."""
        raise RuntimeError("Cannot remove players from an existing game.")

    @abstractmethod
    def next_state(self, state: object, players: List[int], actions: List[str]) \
            -> Tuple[object, List[int], List[float], bool, Union[List[int], None]]:
        """Apply ``actions[i]`` for ``players[i]``.

        Returns ``(new_state, next_players, rewards, terminal, winners)``; ``winners`` is ``None``
        until the game is over (BaseEnvironment.py:140-171).
        """
        raise NotImplementedError

    def compute_ranking(self, state: object, players: List[int], winners: List[int]) -> Dict[int, int]:
        """Final placement: 0 for winners, 1 for everyone else (BaseEnvironment.py:173-195)."""
        top = set(winners)
        return {p: (0 if p in top else 1) for p in players}

    @abstractmethod
    def valid_actions(self, state: object, player: int) -> List[str]:
        """Every action string ``player`` may submit in ``state``."""
        raise NotImplementedError

    @abstractmethod
    def is_valid_action(self, state: object, player: int, action: str) -> bool:
        """Whether ``action`` is legal for ``player`` in ``state``."""
        raise NotImplementedError

    @abstractmethod
    def state_to_observation(self, state: object, player: int) -> Dict[str, np.ndarray]:
        """What ``player`` gets to see of ``state``, as named numpy arrays."""
        raise NotImplementedError

    # ---- optional state transport ----------------------------------------------------------
    @staticmethod
    def serializable() -> bool:
        """True when ``serialize_state``/``deserialize_state`` are implemented (BaseEnvironment.py:243-251)."""
        return False

    @staticmethod
    def serialize_state(state: object) -> bytearray:
        raise NotImplementedError

    @staticmethod
    def deserialize_state(serialized_state: bytearray) -> object:
        raise NotImplementedError


class SimpleConfigParser:
    """``;``-separated typed config strings; each slot is a type or ``(type, default)``.

    Behaviour follows reference BaseEnvironment.py:286-341: the literal ``"None"`` parses to
    ``None``; trailing slots fall back to their defaults; a missing required slot raises
    ``ValueError``.
    """

    def __init__(self, *types):
        self.types = []
        for spec in types:
            if isinstance(spec, (tuple, list)):
                self.types.append((spec[0], True, spec[1]))
            else:
                self.types.append((spec, False, None))

    def parse(self, config):
        fields = [] if config is None else config.split(";")
        options = []
        used = 0
        for text, (conv, _, _) in zip(fields, self.types):
            options.append(None if text == "None" else conv(text))
            used += 1
        if len(fields) == len(self.types):
            return options
        if not self.types[used][1]:
            raise ValueError("Required Argument not provided: Option {}".format(used))
        options.extend(default for (_, _, default) in self.types[used:])
        return options

    def store(self, *args):
        options = [str(default) for _, _, default in self.types]
        last = 0
        for last, value in enumerate(args):
            options[last] = str(value)
        if len(args) < len(self.types) and not self.types[last + 1][1]:
            raise ValueError("Not enough required arguments provided.")
        return ";".join(options)
