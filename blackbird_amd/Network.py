"""Mirror of /root/reference/src/Network.py:9-112.  getEvaluation/getPolicy run the fused HIP
tower (bb_net_eval) on the int8 planes AsInputArray returns; weights live in an .npz with the
reference's TensorFlow variable names instead of a TF checkpoint.  train() is the step right after
the hot path: PyTorch-ROCm optimiser over the same parameters, then the new
weights are pushed back into the engine."""
import os

import numpy as np

from . import _lib
from . import weights as W
from .MCTS import _seed_from_numpy

_GAME_BY_SHAPE = {(6, 7, 3): _lib.GAME_CONNECT4, (3, 3, 3): _lib.GAME_TICTACTOE, (8, 8, 17): _lib.GAME_DRAGONCHESS}


class Network:
    def __init__(self, name, networkConstructor=None, tensorflowConfig={}):
        self.batchCount = 0
        self._netName = name
        self._constructor = networkConstructor
        self._weights = None
        self._eval_engine = None
        self._trainer = None
        alpha = getattr(networkConstructor, 'alpha', None)
        epsilon = getattr(networkConstructor, 'epsilon', None)
        self.alpha = 0.2 if alpha is None else alpha          # (an explicit epsilon of 0 switches the noise off)
        self.epsilon = 0.3 if epsilon is None else epsilon
        if not self.loadModel(name):
            if networkConstructor is not None and getattr(networkConstructor, 'inputShape', None):
                self._weights = networkConstructor()
                self.saveModel(name)

    # ---- weights -------------------------------------------------------------------------------------
    def _ensure_weights(self, in_planes):
        if self._weights is None:
            if self._constructor is None:
                raise _lib.BlackbirdHipError('no saved model %r and no NetworkFactory to build one' % self._netName)
            self._weights = self._constructor(in_planes)
            self.saveModel(self._netName)
        return self._weights

    def _engine_for(self, shape):
        game = _GAME_BY_SHAPE.get(tuple(int(x) for x in shape))
        if game is None:
            raise ValueError('no game with input planes of shape %r' % (shape,))
        if self._eval_engine is None or self._eval_engine.game != game:
            self._eval_engine = _lib.Engine(game, n_slots=1, sims_per_move=2, evaluator=_lib.EVAL_NET,
                                            alpha=float(self.alpha), epsilon=float(self.epsilon),
                                            seed=_seed_from_numpy())
            self._eval_engine.load_weights(W.flatten(self._ensure_weights(shape[2])))
        return self._eval_engine

    def _weights_changed(self):
        if self._eval_engine is not None:
            self._eval_engine.load_weights(W.flatten(self._weights))

    # ---- reference API ---------------------------------------------------------------------------------
    def getEvaluation(self, state):
        """Network.py:48-54: value of `state` (int8 [n,H,W,C]) for the side to move; returns element 0."""
        state = np.asarray(state)
        v, _l, _p = self._engine_for(state.shape[1:]).net_eval(planes=state.astype(np.int8))
        return v[0]

    def getPolicy(self, state):
        """Network.py:56-64: softmax policy mixed with Beta(alpha,1-alpha) noise (always on, as in the
        reference graph, NetworkFactory.py:176-182); returns row 0."""
        state = np.asarray(state)
        eng = self._engine_for(state.shape[1:])
        eng._noise_calls = getattr(eng, '_noise_calls', 0) + 1
        _v, _l, p = eng.net_eval(planes=state.astype(np.int8), noise=eng._noise_calls)  # fresh draw per call
        return p[0]

    def train(self, state, eval, policy, learningRate=0.01, teacher=None):
        """Network.py:66-84 (see blackbird_amd/training.py for the loss)."""
        from .training import Trainer
        if teacher is not None:
            # the reference's teacher branch (NetworkFactory.py:205-218) calls the removed tf.log and cannot run; refusing
            # is better than silently training without the term
            raise NotImplementedError('policy distillation from a teacher (hasTeacher) is not supported')
        state = np.asarray(state)
        self._ensure_weights(state.shape[-1])
        if self._trainer is None:
            cfg = getattr(self._constructor, 'NetworkConfig', None) or {}
            self._trainer = Trainer(self._weights, alpha=self.alpha, epsilon=self.epsilon,
                                    optimizer=(cfg.get('training') or {}).get('optimizer', 'adam'),
                                    momentum=(cfg.get('training') or {}).get('momentum', 0.9))
        self._trainer.step(state, eval, policy, learningRate)
        self._weights = self._trainer.export()
        self._weights_changed()
        self.batchCount += 1

    def saveModel(self, name=None):
        """Network.py:86-98"""
        if name is None:
            name = self.Name
        saveDir = os.path.join('blackbird_models', name)
        os.makedirs(saveDir, exist_ok=True)
        W.save_npz(os.path.join(saveDir, 'best.npz'), self._weights)

    def loadModel(self, name):
        """Network.py:100-112"""
        path = os.path.join('blackbird_models', name, 'best.npz')
        if not os.path.isfile(path):
            return False
        self._weights = W.load_npz(path)
        self._trainer = None  # parameters and optimiser state of the old weights do not carry over
        self._weights_changed()
        return True
