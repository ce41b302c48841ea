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
    # ---- what a concrete wrapper provides
    def create_env(self, *args, **kwargs) -> BaseEnvironment:
        raise NotImplementedError

    def create_observation_space(self, *args, **kwargs):
        raise NotImplementedError

    def create_action_space(self, *args, **kwargs):
        raise NotImplementedError

    def create_done_dict(self, state, players, rewards, terminal, action_dict) -> Dict[str, bool]:
        return dict.fromkeys(action_dict, terminal)

    def create_info_dict(self, state, players, rewards, terminal, action_dict) -> Dict:
        return {}

    def action_map(self, action):
        return action

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
