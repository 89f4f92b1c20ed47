"""TicTacToe 3x3 (mirror of /root/reference/src/TicTacToe.py:9-138); rules run in the HIP kernels."""
import numpy as np

from . import _lib
from ._grid import GridBoardState


class BoardState(GridBoardState):
    GAME_ID = _lib.GAME_TICTACTOE
    Size = 3
    InARow = 3
    BoardShape = np.array([Size, Size], dtype=np.int8)
    LegalMoves = Size ** 2
    GameType = 'TicTacToe'
    _ROWS, _COLS = Size, Size

    def LegalActionShape(self):
        return self.BoardShape

    def EvalToString(self, eval):
        return str(eval.reshape(3, 3))

    def _coordsToIndex(self, coords):
        return coords[0] * self.Size + coords[1]

    def _indexToCoords(self, index):
        return (index // self.Size, index % self.Size)
