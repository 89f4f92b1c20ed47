"""/root/reference/src/FixedMCTS.py:4-34: chain expansion up to MaxDepth plies per simulation,
base-class evaluator (priors = ones, random rollouts)."""
from . import _lib
from .MCTS import MCTS, Node  # noqa: F401


class FixedMCTS(MCTS):
    _KIND = _lib.MCTS_FIXED

    def __init__(self, **kwargs):
        self.MaxDepth = kwargs.get('maxDepth')
        explorationRate = kwargs.get('explorationRate')
        timeLimit = kwargs.get('timeLimit')
        playLimit = kwargs.get('playLimit')
        if self.MaxDepth <= 0:
            raise ValueError('MaxDepth for MCTS must be > 0.')
        super().__init__(explorationRate, timeLimit, playLimit)

    def _max_depth(self):
        return int(self.MaxDepth)
