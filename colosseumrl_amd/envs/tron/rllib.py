"""The reference's hand-rolled Tron RL environments (colosseumrl/envs/tron/rllib.py): ``TronRayEnvironment`` (all players
driven through an action dict, :14-66), ``SimpleAvoidAgent`` (the scripted opponent, :68-95) and
``TronRaySinglePlayerEnvironment`` (one learner against scripted opponents, :98-157).  Rendering is the reference's
pygame GUI and out of scope: ``render`` returns None, ``close`` does nothing."""
import random

from ..wrappers.spaces import Discrete
from .TronGridEnvironment import TronGridEnvironment
from .TronRllibEnvironment import tron_observation_space

ACTION_NAMES = {0: "forward", 1: "right", 2: "left"}     # integer action -> the environment's action string; a dict, as in the
                                                         # reference (envs/tron/rllib.py:16): -1 or 3 from a faulty policy is a KeyError


class TronRayEnvironment:
    action_space = Discrete(3)

    def __init__(self, board_size=15, num_players=4):
        self.env = TronGridEnvironment.create(board_size=board_size, num_players=num_players)
        self.state = None
        self.players = None
        self.observation_space = tron_observation_space(board_size, num_players)

    def reset(self):
        self.state, self.players = self.env.new_state()
        return {str(p): self.env.state_to_observation(self.state, p) for p in range(self.env.num_players)}

    def step(self, action_dict):
        actions = [ACTION_NAMES[action_dict.get(str(p), 0)] for p in self.players]       # silent players go forward
        self.state, self.players, rewards, terminal, _ = self.env.next_state(self.state, self.players, actions)
        alive = set(self.players)
        asked = [int(key) for key in action_dict]
        observations = {str(p): self.env.state_to_observation(self.state, p) for p in asked}
        reward_dict = {str(p): rewards[p] for p in asked}
        dones = {str(p): p not in alive for p in asked}
        dones["__all__"] = terminal
        return observations, reward_dict, dones, {}

    def render(self, mode="human"):
        return None

    def close(self):
        pass


class SimpleAvoidAgent:
    """Goes straight while the cell ahead is free, otherwise tries one side picked at random and falls back to the other;
    with probability `noise` it moves at random.  Works on the player-relative observation (own head first)."""

    def __init__(self, noise=0.1):
        self.noise = noise

    def __call__(self, env, observation):
        if random.random() <= self.noise:
            return random.choice(tuple(ACTION_NAMES.values()))
        board = observation["board"]
        n = board.shape[0]
        head, direction = observation["heads"][0], observation["directions"][0]
        x, y = head % n, head // n

        def free(d):
            cx, cy = env.next_cell(x, y, d, n)
            return board[cy, cx] == 0
        if free(direction):
            return "forward"
        turn, first, other = random.choice([(1, "right", "left"), (-1, "left", "right")])
        return first if free((direction + turn) % 4) else other


class TronRaySinglePlayerEnvironment:
    """gym-style single-agent view: the first player is the learner, every other player is `agent`."""
    action_space = Discrete(3)

    def __init__(self, board_size=15, num_players=4, spawn_offset=2, agent=None):
        self.env = TronGridEnvironment.create(board_size=board_size, num_players=num_players)
        self.state = None
        self.players = None
        self.human_player = None
        self.spawn_offset = spawn_offset
        self.agent = agent if agent is not None else SimpleAvoidAgent()
        self.observation_space = tron_observation_space(board_size, num_players)

    def _get_observation(self, player):
        return self.env.state_to_observation(self.state, player)

    def reset(self):
        self.state, self.players = self.env.new_state(spawn_offset=self.spawn_offset)
        self.human_player = self.players[0]
        return self._get_observation(self.human_player)

    def step(self, action: int):
        me = self.human_player
        actions = [ACTION_NAMES[action] if p == me else self.agent(self.env, self._get_observation(p)) for p in self.players]
        self.state, self.players, rewards, _, _ = self.env.next_state(self.state, self.players, actions)
        return self._get_observation(me), rewards[me], me not in self.players, {}

    def render(self, mode="human"):
        return None

    def close(self):
        pass
