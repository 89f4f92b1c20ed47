"""/root/reference/src/DynamicMCTS.py:5-34: one expansion per simulation (k_tree_step, kind DYNAMIC)."""
from . import _lib
from .MCTS import MCTS, Node  # noqa: F401


class DynamicMCTS(MCTS):
    _KIND = _lib.MCTS_DYNAMIC

    def __init__(self, **kwargs):
        super().__init__(**kwargs)
