"""DragonChess (mirror of /root/reference/src/DragonChess.py:10-371).  White = king + three pawns and
moves twice per turn; Black = the full army; a side wins by capturing the king.  Move legality,
ApplyAction, Winner and AsInputArray run in the HIP kernels (csrc/games.hip.h: DragonChess)."""
import numpy as np

from . import _lib
from .GameState import GameState


class BoardState(GameState):
    GAME_ID = _lib.GAME_DRAGONCHESS
    piece_map = {1: 10, -1: 11, 2: 0, -2: 1, 3: 4, -3: 5, 4: 6, -4: 7, 5: 2, -5: 3, 6: 8, -6: 9}
    int_to_letter = {0: ' ', 1: 'K', 2: 'P', 3: 'N', 4: 'B', 5: 'R', 6: 'Q',
                     -1: 'k', -2: 'p', -3: 'n', -4: 'b', -5: 'r', -6: 'q'}
    letter_to_int = {v: k for k, v in int_to_letter.items() if k != 0}
    GameType = 'DragonChess'
    LegalMoves = 4032
    possible_moves = [f'{a} {b}' for a in range(64) for b in range(64) if a != b]
    move_to_int = {m: i for i, m in enumerate(possible_moves)}
    int_to_move = {i: m for i, m in enumerate(possible_moves)}
    fen = 'rnbqkbnr/pppppppp/8/8/8/8/3PPP2/4K3 w kq - 0 1'

    def __init__(self):
        b, p, pv, cs = _lib.unpack_dc(_lib.game_initial(self.GAME_ID))
        self.board = b[0].astype(np.float64)
        self.Player = int(p[0])
        self.PreviousPlayer = None
        (self._white_castle_kingside, self._white_castle_queenside, self._black_castle_kingside,
         self._black_castle_queenside) = [bool(x) for x in cs[0]]

    @property
    def Board(self):
        return self.board

    def _castles(self):
        return [self._white_castle_kingside, self._white_castle_queenside, self._black_castle_kingside,
                self._black_castle_queenside]

    def _packed(self):
        return _lib.pack_dc(self.board[None], [self.Player], [self.PreviousPlayer or 0], [self._castles()])

    def _load(self, packed):
        b, p, pv, cs = _lib.unpack_dc(packed)
        self.board = b[0].astype(np.float64)
        self.Player = int(p[0])
        self.PreviousPlayer = int(pv[0]) or None
        (self._white_castle_kingside, self._white_castle_queenside, self._black_castle_kingside,
         self._black_castle_queenside) = [bool(x) for x in cs[0]]

    @classmethod
    def _from_packed(cls, packed):
        s = cls()
        s._load(packed)
        return s

    def Copy(self):
        copy = BoardState()
        copy.Player = self.Player
        copy.PreviousPlayer = self.PreviousPlayer  # DragonChess keeps it (DragonChess.py:66-76)
        (copy._white_castle_kingside, copy._white_castle_queenside, copy._black_castle_kingside,
         copy._black_castle_queenside) = self._castles()
        copy.board = np.copy(self.board)
        return copy

    def LegalActions(self):
        return _lib.game_legal(self.GAME_ID, self._packed())[0].astype(np.float64)

    def LegalActionShape(self):
        return np.array([0 for _ in range(4212)], dtype=np.int8)

    def AsInputArray(self):
        return _lib.game_encode(self.GAME_ID, self._packed())

    def ApplyAction(self, action):
        nxt, status = _lib.game_apply(self.GAME_ID, self._packed(), [int(action)])
        if status[0] != 0:
            raise ValueError('Tried to make an illegal move.')
        self._load(nxt)

    def Winner(self, prevAction=None):
        w = int(_lib.game_winner(self.GAME_ID, self._packed())[0])
        return None if w < 0 else w

    def EvalToString(self, eval):
        return str(eval)

    def __str__(self):
        fboard = np.flip(self.board, axis=0)
        rep = '-----------------\n'
        for row in range(8):
            rep += '|' + ''.join(f'{self.int_to_letter[int(fboard[row, col])]}|' for col in range(8)) + '\n'
        return rep + '-----------------'

    def __eq__(self, other):
        if other.Player != self.Player:
            return False
        if other._castles() != self._castles():
            return False
        return (other.Board == self.Board).all()

    def __hash__(self):
        return "{0}{1}".format(self.Player, str(self)).__hash__()
