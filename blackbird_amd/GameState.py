"""Mirror of the reference's abstract game interface (/root/reference/src/GameState.py:1-28) plus the
de-facto members its callers use (AsInputArray, GameType, LegalMoves, __eq__, __hash__).

Concrete games keep the reference's attribute layout (`Board`, `Player`, `PreviousPlayer`) on the
host, but every rule evaluation -- LegalActions, ApplyAction, Winner, AsInputArray -- is computed
by the HIP kernels through the C ABI (bb_game_*).  There is no host-side rules code to fall back on.
"""


class GameState(object):
    GAME_ID = None  # BB_GAME_* of the concrete class

    def __init__(self):
        self.Board = None
        self.Player = None
        self.PreviousPlayer = None

    def Copy(self):
        raise NotImplementedError

    def LegalActions(self):
        raise NotImplementedError

    def LegalActionShape(self):
        raise NotImplementedError

    def ApplyAction(self, action):
        raise NotImplementedError

    def Winner(self, prevAction=None):
        raise NotImplementedError

    def NumericRepresentation(self):
        raise NotImplementedError

    def EvalToString(self, eval):
        return str(eval)

    def SerializeState(self, state, policy, eval):
        raise NotImplementedError

    # ---- engine plumbing (not part of the reference interface) ----------------------------------
    def _packed(self):
        """This position in the engine's packed layout (include/blackbird_hip.h)."""
        raise NotImplementedError

    @classmethod
    def _from_packed(cls, packed):
        raise NotImplementedError
