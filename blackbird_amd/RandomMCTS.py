"""The uniform-random opponent of Blackbird.TestRandom (/root/reference/src/RandomMCTS.py:4-25): a player with the
searcher interface that never searches.  The legality mask comes from the game object (the HIP game kernel); the
batched arena (blackbird_amd/arena.py) recognises this class and draws its moves for all games at once."""
import numpy as np

from .MCTS import MCTS


def _nothing(self, *_args, **_kwargs):
    return None


class RandomMCTS(MCTS):
    def __init__(self, *_args, **_kwargs):
        self.Root = None
        self.TimeLimit = None
        self.PlayLimit = None

    def FindMove(self, state, *_args, **_kwargs):
        """(next state, a uniform 'win rate', the uniform distribution over the legal moves)."""
        mask = np.asarray(state.LegalActions(), dtype=np.float64)
        moves = np.flatnonzero(mask == 1)
        successor = self._applyAction(state, int(np.random.choice(moves)))
        return successor, np.random.random(), (mask / len(moves) if len(moves) else mask)

    # a player without a tree has nothing to drop, reset or re-root
    DropRoot = ResetRoot = MoveRoot = _nothing
