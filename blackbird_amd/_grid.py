"""Shared implementation of the two stone-placement games (Connect4, TicTacToe) on top of the HIP
game kernels.  Follows /root/reference/src/Connect4.py and TicTacToe.py member for member."""
import numpy as np

from . import _lib
from .GameState import GameState


class GridBoardState(GameState):
    Players = {0: ' ', 1: 'X', 2: 'O'}
    Dirs = [(0, 1), (1, 1), (1, 0), (1, -1)]
    _ROWS = _COLS = 0  # set by subclasses
    _FLIP_STR = False  # Connect4 prints the top row first (Connect4.py:116-127)

    def __init__(self):
        self.Board = np.zeros((self._ROWS, self._COLS, 2), dtype=np.int8)
        self.Player = 1
        self.PreviousPlayer = None

    # ---- packed form -------------------------------------------------------------------------------
    def _packed(self):
        return _lib.pack_grid(self.GAME_ID, self.Board[None], [self.Player], [self.PreviousPlayer or 0])

    @classmethod
    def _from_packed(cls, packed):
        b, p, pv = _lib.unpack_grid(cls.GAME_ID, np.asarray(packed).reshape(1, 2))
        s = cls()
        s.Board = b[0]
        s.Player = int(p[0])
        s.PreviousPlayer = int(pv[0]) or None
        return s

    # ---- reference interface ---------------------------------------------------------------------------
    def Copy(self):
        copy = type(self)()
        copy.Player = self.Player  # PreviousPlayer is NOT copied (Connect4.py:24-28, TicTacToe.py:23-27)
        copy.Board = np.copy(self.Board)
        return copy

    def LegalActions(self):
        return _lib.game_legal(self.GAME_ID, self._packed())[0].astype(np.float64)

    def ApplyAction(self, action):
        nxt, status = _lib.game_apply(self.GAME_ID, self._packed(), [int(action)])
        if status[0] != 0:
            raise ValueError('Tried to make an illegal move.')
        b, p, pv = _lib.unpack_grid(self.GAME_ID, nxt)
        self.Board = b[0]
        self.Player = int(p[0])
        self.PreviousPlayer = int(pv[0])

    def AsInputArray(self):
        return _lib.game_encode(self.GAME_ID, self._packed())

    def Winner(self, prevAction=None):
        prev = None if prevAction is None else [int(prevAction)]
        w = int(_lib.game_winner(self.GAME_ID, self._packed(), prev)[0])
        if w < 0:
            return None
        return np.float64(w) if w > 0 else 0

    def _collapsed(self):
        array = np.zeros(self.Board.shape[:2])
        array[self.Board[:, :, 0] == 1] = 1
        array[self.Board[:, :, 1] == 1] = 2
        return array

    def __str__(self):
        array = self._collapsed()
        rows = reversed(range(array.shape[0])) if self._FLIP_STR else range(array.shape[0])
        s = ''
        for i in rows:
            s += '[ '
            for j in range(array.shape[1]):
                s += ' {} '.format(self.Players[array[i, j]])
                if j < array.shape[1] - 1:
                    s += '|'
            s += ']\n'
        return s

    def __eq__(self, other):
        if other.Player != self.Player:
            return False
        return (other.Board == self.Board).all()

    def __hash__(self):
        return "{0}{1}".format(self.Player, str(self)).__hash__()
