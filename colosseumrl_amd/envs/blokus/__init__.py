from .BlokusEnvironment import *  # noqa: F401,F403
from .BlokusEnvironment import BlokusEnvironment, action_to_string, string_to_action
