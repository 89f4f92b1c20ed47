"""The step right after the hot path: Network.train's loss and optimiser
(/root/reference/src/NetworkFactory.py:185-245) on PyTorch-ROCm.  PyTorch is used for this
optimiser step only; self-play never touches it.

Reproduced exactly as written in the reference graph, quirks included:
  * value loss  = mean((evaluation - label)^2)                                         (:187-188)
  * policy      = ((1-eps)*softmax + eps*Beta noise) / sum over ALL elements (batch too) (:176-182)
  * policy loss = -mean(log(policy) @ policyLabel^T)  -- the full BxB cross matrix      (:190-194)
  * L2 term     = mean over non-bias trainable variables of sum(v^2)/2, no coefficient  (:196-201)
  * batch norm runs in inference mode (moving statistics are constants), epsilon 1e-3
  * optimiser: tf.compat.v1.train Adam / Momentum / GradientDescent by config, their update rules written out
    in `Trainer._apply` (TF1's Adam puts epsilon outside the bias correction; torch.optim.Adam does not)  (:234-242)
The teacher term (:205-218) calls the removed `tf.log` in the reference and cannot run there; Network.train refuses it.
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
        self.kind, self.momentum = optimizer, float(momentum)
        # optimiser slots, one per trainable variable, as the TF1 optimisers keep them (NetworkFactory.py:234-242)
        self.t = 0                                                         # AdamOptimizer's step count (beta powers)
        self.m = {k: torch.zeros_like(v) for k, v in self.params.items()}  # Adam first moment / Momentum accumulator
        self.v = {k: torch.zeros_like(v) for k, v in self.params.items()}  # Adam second moment

    def _apply(self, grads, lr):
        """One optimiser update, written out so that it is the reference's optimiser and not a look-alike:
          tf.compat.v1.train.AdamOptimizer(lr) (beta1 0.9, beta2 0.999, epsilon 1e-8):
              lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2
              var -= lr_t * m / (sqrt(v) + epsilon)        (epsilon OUTSIDE the bias correction, unlike torch.optim.Adam)
          MomentumOptimizer(lr, momentum): accum = momentum * accum + g;  var -= lr * accum
          GradientDescentOptimizer(lr):    var -= lr * g"""
        with torch.no_grad():
            if self.kind == 'adam':
                b1, b2, eps = 0.9, 0.999, 1e-8
                self.t += 1
                lr_t = lr * (1.0 - b2 ** self.t) ** 0.5 / (1.0 - b1 ** self.t)
                for k, p in self.params.items():
                    g = grads[k]
                    self.m[k].mul_(b1).add_(g, alpha=1.0 - b1)
                    self.v[k].mul_(b2).addcmul_(g, g, value=1.0 - b2)
                    p.sub_(lr_t * self.m[k] / (self.v[k].sqrt() + eps))
            elif self.kind == 'momentum':
                for k, p in self.params.items():
                    self.m[k].mul_(self.momentum).add_(grads[k])
                    p.sub_(lr * self.m[k])
            else:
                for k, p in self.params.items():
                    p.sub_(lr * grads[k])

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

    def gradients(self, state, eval, policy, noise=None):
        """Loss, its three terms and d loss / d variable for one batch (what optimizer.minimize(loss) differentiates).
        noise: the A Beta(alpha, 1-alpha) draws of the graph's Dirichlet node, or None to draw them."""
        boards = torch.tensor(np.asarray(state, dtype=np.float32), device=self.device)
        ev = torch.tensor(np.asarray(eval, dtype=np.float32).reshape(-1), device=self.device)
        pl = torch.tensor(np.asarray(policy, dtype=np.float32), device=self.device)
        if noise is not None:
            noise = torch.tensor(np.asarray(noise, dtype=np.float32), device=self.device)
        total, parts = self.loss(boards, ev, pl, noise)
        names = list(self.params)
        gs = torch.autograd.grad(total, [self.params[k] for k in names])
        return float(total.detach()), [float(p.detach()) for p in parts], dict(zip(names, gs))

    def step(self, state, eval, policy, learningRate, noise=None):
        total, parts, grads = self.gradients(state, eval, policy, noise)
        self._apply(grads, float(learningRate))
        return total, parts

    def export(self):
        out = {k: v.detach().cpu().numpy().copy() for k, v in self.params.items()}
        out.update({k: v.cpu().numpy().copy() for k, v in self.consts.items()})
        return out
