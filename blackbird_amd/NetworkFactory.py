"""Network description + initialiser (mirror of /root/reference/src/NetworkFactory.py:4-20).

The reference's factory builds a TensorFlow graph; here it carries the same configuration keys and,
when called, returns freshly initialised weights under the reference's variable names
(blackbird_amd.weights).  The forward pass itself is the fused HIP kernel (csrc/net.hip.h)."""
from . import weights as W


class NetworkFactory(object):
    def __init__(self, networkConfig, policyShape, inputShape=None, seed=0):
        self.NetworkConfig = networkConfig
        self.alpha = networkConfig.get('policy').get('dirichlet').get('alpha')
        self.epsilon = networkConfig.get('policy').get('dirichlet').get('epsilon')
        self.policyShape = policyShape
        self.hasTeacher = networkConfig.get('hasTeacher')
        self.inputShape = inputShape  # (H, W, C); the reference hard-codes C=17 (NetworkFactory.py:24)
        self.seed = seed

    def __call__(self, in_planes=None):
        C = in_planes if in_planes is not None else (self.inputShape[2] if self.inputShape else 17)
        return W.init_weights(C, self.NetworkConfig['filters'], self.NetworkConfig['blocks'],
                              self.NetworkConfig.get('eval').get('dense'), self.policyShape, seed=self.seed)
