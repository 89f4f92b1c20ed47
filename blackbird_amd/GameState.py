"""The reference's abstract game interface (/root/reference/src/GameState.py:1-28) plus the de-facto members its callers
use (AsInputArray, GameType, LegalMoves, __eq__, __hash__).

Concrete games keep the reference's attribute layout (`Board`, `Player`, `PreviousPlayer`) on the host, but every rule
evaluation -- LegalActions, ApplyAction, Winner, AsInputArray -- is computed by the HIP kernels through the C ABI
(bb_game_*).  There is no host-side rules code to fall back on.

The contract, method by method (a concrete game implements all of them; calling one on the base raises
NotImplementedError exactly as in the reference):

    Copy()                     a new state equal to this one
    LegalActions()             float64 vector over the game's LegalMoves actions, 1.0 where the action is legal
    LegalActionShape()         shape of that vector
    ApplyAction(action)        play `action` in place; ValueError('Tried to make an illegal move.') if it is not legal
    Winner(prevAction=None)    None while the game goes on, 0 for a draw, else the winning player (1 / 2)
    NumericRepresentation()    a numeric form of the position
    SerializeState(s, pi, v)   the position with its search results as bytes
    EvalToString(eval)         how an evaluation is printed (str by default)
"""


def _required(signature):
    """An interface method a concrete game must provide."""
    name = signature.split('(')[0]

    def method(self, *args, **kwargs):
        raise NotImplementedError('%s.%s' % (type(self).__name__, signature))
    method.__name__ = name
    method.__doc__ = 'GameState.%s -- to be provided by the concrete game.' % signature
    return method


class GameState(object):
    GAME_ID = None  # BB_GAME_* of the concrete class

    def __init__(self):
        self.Board = self.Player = self.PreviousPlayer = None

    def EvalToString(self, eval):
        return str(eval)

    # engine plumbing (not part of the reference interface): the position in the engine's packed layout
    # (include/blackbird_hip.h) and back
    _packed = _required('_packed()')
    _from_packed = classmethod(_required('_from_packed(packed)'))


for _sig in ('Copy()', 'LegalActions()', 'LegalActionShape()', 'ApplyAction(action)', 'Winner(prevAction=None)',
             'NumericRepresentation()', 'SerializeState(state, policy, eval)'):
    setattr(GameState, _sig.split('(')[0], _required(_sig))
del _sig
