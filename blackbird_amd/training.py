"""The step right after the hot path: Network.train's loss and optimiser
(/root/reference/src/NetworkFactory.py:185-245) on PyTorch-ROCm.  PyTorch is used for this
optimiser step only; self-play never touches it.

Reproduced exactly as written in the reference graph, quirks included:
  * value loss  = mean((evaluation - label)^2)                                         (:187-188)
  * policy      = ((1-eps)*softmax + eps*Beta noise) / sum over ALL elements (batch too) (:176-182)
  * policy loss = -mean(log(policy) @ policyLabel^T)  -- the full BxB cross matrix      (:190-194)
  * L2 term     = mean over non-bias trainable variables of sum(v^2)/2, no coefficient  (:196-201)
  * batch norm runs in inference mode (moving statistics are constants), epsilon 1e-3
  * optimiser: Adam / Momentum / plain SGD by config                                     (:234-242)
"""
import numpy as np
import torch
import torch.nn.functional as Fn

from . import weights as W


class Trainer:
    def __init__(self, weights, alpha=0.2, epsilon=0.3, optimizer='adam', momentum=0.9, device=None):
        self.device = torch.device(device or ('cuda' if torch.cuda.is_available() else 'cpu'))
        self.alpha, self.epsilon = float(alpha), float(epsilon)
        self.C, self.F, self.R, self.D, self.A = W.infer_shape(weights)
        self.params, self.consts = {}, {}
        for k, v in weights.items():
            t = torch.tensor(np.asarray(v, dtype=np.float32), device=self.device)
            if k.endswith('moving_mean') or k.endswith('moving_variance'):
                self.consts[k] = t
            else:
                self.params[k] = t.requires_grad_(True)
        self.kind, self.momentum = optimizer, momentum
        self.opt = None

    def _make_opt(self, lr):
        ps = list(self.params.values())
        if self.kind == 'adam':
            return torch.optim.Adam(ps, lr=lr, betas=(0.9, 0.999), eps=1e-8)
        if self.kind == 'momentum':
            return torch.optim.SGD(ps, lr=lr, momentum=self.momentum)
        return torch.optim.SGD(ps, lr=lr)

    def _get(self, k):
        return self.params[k] if k in self.params else self.consts[k]

    def _conv(self, x, name):
        k = self._get(f'{name}/kernel').permute(3, 2, 0, 1)
        return Fn.conv2d(x, k, self._get(f'{name}/bias'), padding=k.shape[-1] // 2)

    def _bn(self, x, name):
        g, b, m, v = (self._get(f'{name}/{f}').view(1, -1, 1, 1) for f in W.BN_FIELDS)
        return g * (x - m) / torch.sqrt(v + 1e-3) + b

    def forward(self, boards):
        x = boards.permute(0, 3, 1, 2)
        x = torch.relu(self._bn(self._conv(x, 'resTower/conv_block/conv'), 'resTower/conv_block/batch_norm'))
        for i in range(self.R):
            h = torch.relu(self._bn(self._conv(x, f'resTower/block_{i}/conv_1'), f'resTower/block_{i}/batch_norm_1'))
            h = self._bn(self._conv(h, f'resTower/block_{i}/conv_2'), f'resTower/block_{i}/batch_norm_2')
            x = torch.relu(h + x)
        v = torch.relu(self._bn(self._conv(x, 'value/convolution'), 'value/batch_norm')).permute(0, 2, 3, 1)
        v = torch.relu((v @ self._get('value/dense_1/kernel') + self._get('value/dense_1/bias')).sum(dim=(1, 2)))
        value = torch.tanh((v @ self._get('value/dense_2/kernel') + self._get('value/dense_2/bias')).sum(dim=1))
        p = torch.relu(self._bn(self._conv(x, 'policy/convolution'), 'policy/batch_norm')).permute(0, 2, 3, 1)
        logits = (p @ self._get('policy/policy/kernel') + self._get('policy/policy/bias')).sum(dim=(1, 2))
        return value, logits

    def loss(self, boards, evalLabel, policyLabel, noise=None):
        value, logits = self.forward(boards)
        base = torch.softmax(logits, dim=1)
        if noise is None:  # A independent Beta(alpha, 1-alpha) draws shared across the batch
            noise = torch.distributions.Beta(self.alpha, 1.0 - self.alpha).sample((self.A,)).to(self.device)
        policy = (1 - self.epsilon) * base + self.epsilon * noise.view(1, -1)
        policy = policy / policy.sum()
        lossEvaluation = torch.mean((value - evalLabel) ** 2)
        lossPolicy = -torch.mean(torch.log(policy) @ policyLabel.t())
        l2 = [0.5 * (p ** 2).sum() for k, p in self.params.items() if 'bias' not in k]
        lossParam = torch.stack(l2).mean()
        return lossEvaluation + lossPolicy + lossParam, (lossEvaluation, lossPolicy, lossParam)

    def step(self, state, eval, policy, learningRate):
        boards = torch.tensor(np.asarray(state, dtype=np.float32), device=self.device)
        ev = torch.tensor(np.asarray(eval, dtype=np.float32).reshape(-1), device=self.device)
        pl = torch.tensor(np.asarray(policy, dtype=np.float32), device=self.device)
        if self.opt is None:
            self.opt = self._make_opt(float(learningRate))
        for g in self.opt.param_groups:
            g['lr'] = float(learningRate)
        self.opt.zero_grad()
        total, parts = self.loss(boards, ev, pl)
        total.backward()
        self.opt.step()
        return float(total.detach()), [float(p.detach()) for p in parts]

    def export(self):
        out = {k: v.detach().cpu().numpy().copy() for k, v in self.params.items()}
        out.update({k: v.cpu().numpy().copy() for k, v in self.consts.items()})
        return out
