"""Multi-agent dict API over a drop-in environment (colosseumrl/envs/wrappers/rllib.py:8-55).

``reset() -> {player: observation}`` and ``step({player: action}) -> observations, rewards, dones, infos`` with players
keyed by their number as a string and ``dones['__all__']`` = terminal -- the contract RLlib's ``MultiAgentEnv`` expects.
ray is optional: when it is importable the wrapper derives from ``MultiAgentEnv`` as the reference's does, otherwise from
``object``; nothing else changes.  Every ``step`` is ONE ``next_state`` call of the wrapped class, i.e. one fused launch
on host-mapped memory whose by-products serve the ``state_to_observation`` calls that follow."""
from typing import Dict

from ...BaseEnvironment import BaseEnvironment

try:                                                      # pragma: no cover - depends on the installation
    from ray.rllib import MultiAgentEnv as _Base
except ImportError:
    _Base = object


class RllibWrapper(_Base):
    # ---- what a concrete wrapper provides (the constructor's arguments are handed to all three factories)
    def create_env(self, *args, **kwargs) -> BaseEnvironment:
        """The drop-in environment instance to wrap."""
        raise NotImplementedError("create_env")

    def create_observation_space(self, *args, **kwargs):
        """Space of ONE player's observation (``self.env`` exists when this is called)."""
        raise NotImplementedError("create_observation_space")

    def create_action_space(self, *args, **kwargs):
        """Space of ONE player's action."""
        raise NotImplementedError("create_action_space")

    # ---- optional hooks
    def action_map(self, action):
        """Library action -> the environment's action string; identity unless overridden."""
        return action

    def create_done_dict(self, state, players, rewards, terminal, action_dict) -> Dict[str, bool]:
        """Per-player done flags of the players that acted (``'__all__'`` is added by ``step``): the game's terminal flag."""
        return dict.fromkeys(action_dict, terminal)

    def create_info_dict(self, state, players, rewards, terminal, action_dict) -> Dict:
        """The info dict of a step; empty unless overridden."""
        return {}

    # ---- the episode loop
    def __init__(self, *args, **kwargs):
        self.env = self.create_env(*args, **kwargs)
        self.action_space = self.create_action_space(*args, **kwargs)
        self.observation_space = self.create_observation_space(*args, **kwargs)
        self.state = None
        self.players = None

    def reset(self):
        self.state, self.players = self.env.new_state()
        return {str(p): self.env.state_to_observation(self.state, p) for p in self.players}

    def step(self, action_dict):
        # players that are to move but sent nothing play the empty action, as in the reference
        actions = [self.action_map(action_dict[key]) if key in action_dict else "" for key in map(str, self.players)]
        self.state, self.players, rewards, terminal, _ = self.env.next_state(self.state, self.players, actions)
        observations = {key: self.env.state_to_observation(self.state, int(key)) for key in action_dict}
        reward_dict = {key: rewards[int(key)] for key in action_dict}
        done_dict = self.create_done_dict(self.state, self.players, rewards, terminal, action_dict)
        done_dict["__all__"] = terminal
        return observations, reward_dict, done_dict, self.create_info_dict(self.state, self.players, rewards, terminal, action_dict)
