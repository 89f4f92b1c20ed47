"""/root/reference/src/RandomMCTS.py:4-25: uniform random mover with the FindMove signature
(evaluation opponent only; host-side, the legality mask still comes from the HIP kernel)."""
import numpy as np

from .MCTS import MCTS


class RandomMCTS(MCTS):
    def __init__(self, *args, **kwargs):
        self.Root = None

    def FindMove(self, state, *args, **kwargs):
        legal = state.LegalActions()
        action = np.random.choice([i for i in range(len(legal)) if legal[i] == 1])
        winRate = np.random.random()
        childProbability = legal.copy()
        s = sum(childProbability)
        if s > 0:
            childProbability /= s
        return self._applyAction(state, action), winRate, childProbability

    def DropRoot(self, *args, **kwargs):
        return

    def ResetRoot(self, *args, **kwargs):
        return

    def MoveRoot(self, *args, **kwargs):
        return
