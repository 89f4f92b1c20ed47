"""Connect4 7x6 (mirror of /root/reference/src/Connect4.py:9-135); rules run in the HIP kernels."""
import numpy as np

from . import _lib
from ._grid import GridBoardState


class BoardState(GridBoardState):
    GAME_ID = _lib.GAME_CONNECT4
    Width = 7
    Height = 6
    InARow = 4
    BoardShape = np.array([Width, Height], dtype=np.int8)
    LegalMoves = Width
    GameType = 'Connect4'
    _ROWS, _COLS = Height, Width
    _FLIP_STR = True

    def LegalActionShape(self):
        return np.array([self.Width], dtype=np.int8)
