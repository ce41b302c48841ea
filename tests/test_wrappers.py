"""The RL-library wrappers over the drop-in classes (colosseumrl/envs/wrappers/rllib.py, envs/tron/rllib.py,
envs/tron/TronRllibEnvironment.py): the dict plumbing on a CPU toy environment, the scripted agent on hand-made
observations, and -- on the GPU -- the Tron wrappers next to the oracle."""
import random

import numpy as np
import pytest

from colosseumrl_amd.BaseEnvironment import BaseEnvironment
from colosseumrl_amd.envs.tron import TronGridEnvironment
from colosseumrl_amd.envs.tron.rllib import ACTION_NAMES, SimpleAvoidAgent
from colosseumrl_amd.envs.wrappers import RllibWrapper
from colosseumrl_amd.envs.wrappers.spaces import Box, Dict, Discrete


class CountDown(BaseEnvironment):
    """Two players alternate; a state is (counter, log); the game ends when the counter reaches 0."""

    @property
    def min_players(self):
        return 2

    @property
    def max_players(self):
        return 2

    @staticmethod
    def observation_names():
        return ["counter"]

    @property
    def observation_shape(self):
        return {"counter": (1,)}

    def new_state(self, num_players=2):
        return (3, []), [0, 1]

    def next_state(self, state, players, actions):
        counter, log = state
        log = log + [tuple(actions)]
        counter -= 1
        nxt = [] if counter == 0 else ([1] if counter == 1 else [0, 1])
        return (counter, log), nxt, [counter, -counter], counter == 0, [0] if counter == 0 else None

    def valid_actions(self, state, player):
        return ["a", "b", ""]

    def is_valid_action(self, state, player, action):
        return action in ("a", "b", "")

    def state_to_observation(self, state, player):
        return {"counter": np.array([state[0] * 10 + player])}

    @staticmethod
    def serialize_state(state):
        return bytearray(repr(state).encode())

    @staticmethod
    def deserialize_state(serialized_state):
        return eval(bytes(serialized_state).decode())


class CountDownWrapper(RllibWrapper):
    def create_env(self, *args, **kwargs):
        return CountDown("")

    def create_observation_space(self, *args, **kwargs):
        return Dict({"counter": Box(0, 40, shape=(1,))})

    def create_action_space(self, *args, **kwargs):
        return Discrete(2)

    def action_map(self, action):
        return "ab"[action]


def test_rllib_wrapper_dict_plumbing():
    w = CountDownWrapper()
    assert w.action_space.n == 2 and w.observation_space["counter"].shape == (1,)
    obs = w.reset()
    assert sorted(obs) == ["0", "1"] and obs["1"]["counter"][0] == 31
    obs, rew, done, info = w.step({"0": 1})                     # player 1 to move too, but silent: it plays ''
    assert w.state[1] == [("b", "")] and sorted(obs) == ["0"] and obs["0"]["counter"][0] == 20
    assert rew == {"0": 2} and done == {"0": False, "__all__": False} and info == {}
    obs, rew, done, _ = w.step({"0": 0, "1": 0})
    assert w.state[1][-1] == ("a", "a") and rew == {"0": 1, "1": -1} and w.players == [1]
    obs, rew, done, _ = w.step({"1": 1, "0": 0})                # only player 1 is asked: player 0's entry is ignored
    assert w.state[1][-1] == ("b",) and done == {"1": True, "0": True, "__all__": True}
    assert obs["0"]["counter"][0] == 0 and obs["1"]["counter"][0] == 1


def test_space_descriptors():
    assert Discrete(3).n == 3
    b = Box(0, 4, shape=(5, 5))
    assert tuple(b.shape) == (5, 5)
    d = Dict({"board": b})
    assert d["board"] is b or tuple(d["board"].shape) == (5, 5)


def test_simple_avoid_agent_rules():
    agent = SimpleAvoidAgent(noise=0.0)
    board = np.zeros((7, 7), np.int64)
    obs = {"board": board, "heads": np.array([3 * 7 + 3]), "directions": np.array([1])}      # at (3, 3) heading east
    assert agent(TronGridEnvironment, obs) == "forward"
    board[3, 4] = 2                                              # ahead blocked: a side picked at random, both free
    random.seed(0)
    picks = {agent(TronGridEnvironment, obs) for _ in range(40)}
    assert picks == {"left", "right"}
    board[4, 3] = 2                                              # south (a right turn when heading east) blocked too
    assert {agent(TronGridEnvironment, obs) for _ in range(40)} == {"left"}
    board[2, 3] = 2                                              # boxed in: whatever it picks is a turn
    assert {agent(TronGridEnvironment, obs) for _ in range(40)} <= {"left", "right"}
    random.seed(1)
    noisy = SimpleAvoidAgent(noise=1.0)
    assert {noisy(TronGridEnvironment, obs) for _ in range(60)} == set(ACTION_NAMES.values())


@pytest.mark.gpu
def test_tron_rllib_wrappers_vs_oracle():
    """TronRllibEnvironment / TronRayEnvironment / TronRaySinglePlayerEnvironment on the HIP drop-in class, every step
    checked against the oracle stepping the same actions."""
    from colosseumrl_amd.envs.tron.TronRllibEnvironment import TronRllibEnvironment
    from colosseumrl_amd.envs.tron.rllib import TronRayEnvironment, TronRaySinglePlayerEnvironment
    from oracle import oracle as O
    code = {0: 0, 1: 1, 2: -1}                                   # wrapper action -> oracle action (forward, right, left)
    rng = np.random.default_rng(3)

    def fresh_oracle(N, P):
        sh, sd = O.tron_start_positions(N, P)
        st = O.TronState(N, P, 1)
        O.tron_reset(st, sh, sd)
        return st

    for cls, kwargs in ((TronRllibEnvironment, dict(board_size=11, num_players=4)), (TronRayEnvironment, dict(board_size=11, num_players=4))):
        w = cls(**kwargs)
        assert w.action_space.n == 3 and tuple(w.observation_space["board"].shape) == (11, 11)
        for episode in range(3):
            obs = w.reset()
            st = fresh_oracle(11, 4)
            assert sorted(obs) == ["0", "1", "2", "3"]
            for t in range(60):
                alive = list(w.players)
                acts = {str(p): int(rng.integers(0, 3)) for p in alive if rng.random() < 0.9}    # some players stay silent
                if cls is TronRllibEnvironment and len(acts) < len(alive):
                    acts = {str(p): int(rng.integers(0, 3)) for p in alive}                      # '' is no Tron action
                a = np.zeros((4, 1), np.int8)
                for p in alive:
                    a[p, 0] = code[acts.get(str(p), 0)]
                obs, rew, done, _ = w.step(acts)
                r2, t2, _ = O.tron_step(st, a)
                assert np.array_equal(w.state[0].reshape(-1), st.board[0]) and np.array_equal(w.state[1], st.heads[:, 0])
                assert done["__all__"] == bool(t2[0])
                for key in acts:
                    p = int(key)
                    assert rew[key] == r2[p, 0]
                    gone = p not in w.players
                    assert done[key] == (gone or bool(t2[0]) if cls is TronRllibEnvironment else gone)
                    want = O.tron_observe(st, np.array([p], np.int8))
                    assert np.array_equal(obs[key]["board"].reshape(-1), want[0][0]) and np.array_equal(obs[key]["heads"], want[1][:, 0])
                if done["__all__"]:
                    break
            assert done["__all__"]
    random.seed(5)
    single = TronRaySinglePlayerEnvironment(board_size=13, num_players=3, agent=SimpleAvoidAgent(noise=0.05))
    lengths = []
    for episode in range(4):
        ob = single.reset()
        assert ob["heads"][0] == single.state[1][single.human_player]
        for t in range(200):
            ob, r, done, info = single.step(int(rng.integers(0, 3)))
            assert info == {}
            if done or len(single.players) == 0:
                break
        lengths.append(t + 1)
        assert done or len(single.players) == 0
    assert max(lengths) >= 3
