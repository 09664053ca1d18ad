"""Neural-Kernel-Network layers (Sun et al. 2018) -- host-side parameter containers.

Mirrors gpflowSlim/neural_kernel_network/neural_kernel_network_wrapper.py: NKNWrapper :28-61,
Linear :90-130 (positive weights and bias, uniform init in [1/(2 in), 3/(2 in)]), Product :134-152,
Activation :155-173.  `forward` is the numpy evaluation (used for Kdiag and by the tests); for K(X, X2)
the layers are compiled into the kernel program and evaluated per matrix entry inside the HIP
kernel-matrix build (csrc/kmat.hip, nkn_tile_kernel).
"""
import math

import numpy as np

from .. import _backend as be
from .._settings import settings
from ..params import Parameter
from ..transforms import positive


class _KernelLayer(object):
    def __init__(self, input_dim, name):
        self.input_dim = input_dim
        self.name = name

    def forward(self, input):
        raise NotImplementedError

    @property
    def parameters(self):
        raise NotImplementedError

    def _nodes(self, layer_index):
        raise NotImplementedError


class Linear(_KernelLayer):
    """y = A x + b with positive weights and bias (neural_kernel_network_wrapper.py:90-130)"""

    def __init__(self, input_dim, output_dim, name='Linear'):
        super(Linear, self).__init__(input_dim, name=name)
        self.output_dim = output_dim
        min_w, max_w = 1. / (2 * input_dim), 3. / (2 * input_dim)
        weights = np.random.uniform(low=min_w, high=max_w, size=[output_dim, input_dim]).astype(settings.float_type)
        self._weights = Parameter(weights, transform=positive, name='weights')
        self._bias = Parameter(0.01 * np.ones([self.output_dim], dtype=settings.float_type), transform=positive,
                               name='bias')

    @property
    def weights(self):
        return self._weights.value

    @property
    def bias(self):
        return self._bias.value

    def forward(self, input):
        return np.matmul(input, np.transpose(self.weights)) + self.bias

    @property
    def parameters(self):
        return [self._weights, self._bias]

    def _nodes(self, layer_index):
        if self.input_dim > 16 or self.output_dim > 16:
            raise NotImplementedError("NKN layers wider than 16 are not supported on the device")
        W, b = np.atleast_2d(self.weights), np.atleast_1d(self.bias)
        nodes = []
        for o in range(self.output_dim):
            nd = be.KernNode()
            nd.op = be.K_NKN_LINROW
            nd.n_dims = self.input_dim
            nd.active_dims[0] = layer_index
            nd.variance = float(b[o])
            for j in range(self.input_dim):
                nd.lengthscales[j] = float(W[o, j])
            nodes.append(nd)
        return nodes


class Product(_KernelLayer):
    """products of `step` consecutive inputs (neural_kernel_network_wrapper.py:134-152)"""

    def __init__(self, input_dim, step, name='Product'):
        super(Product, self).__init__(input_dim, name=name)
        assert isinstance(step, int) and step > 1, 'step must be number greater than 1'
        assert int(math.fmod(input_dim, step)) == 0, 'input dim must be multiples of step'
        self.step = step
        self.output_dim = input_dim // step

    def forward(self, input):
        output = np.reshape(input, [np.shape(input)[0], -1, self.step])
        return np.prod(output, -1)

    @property
    def parameters(self):
        return []

    def _nodes(self, layer_index):
        if self.step > 4:
            raise NotImplementedError("Product step > 4 is not supported on the device")
        nd = be.KernNode()
        nd.op = be.K_NKN_PRODUCT
        nd.n_dims = self.step
        nd.active_dims[0] = layer_index
        return [nd]


class Activation(_KernelLayer):
    """elementwise activation (neural_kernel_network_wrapper.py:155-173).  The device evaluates named
    activations only: activation_fn = 'exp' (or numpy.exp)."""

    def __init__(self, input_dim, activation_fn, activation_fn_params=None, name='Activation'):
        super(Activation, self).__init__(input_dim, name=name)
        if activation_fn in ('exp', np.exp):
            self.activation_fn, self._code = np.exp, 1.0
        else:
            raise NotImplementedError("only the exp activation is available on the device path")
        self.output_dim = input_dim
        self._parameters = list(activation_fn_params or [])

    def forward(self, input):
        return self.activation_fn(input)

    @property
    def parameters(self):
        return self._parameters

    def _nodes(self, layer_index):
        nd = be.KernNode()
        nd.op = be.K_NKN_ACT
        nd.period = self._code
        nd.active_dims[0] = layer_index
        return [nd]


class NKNWrapper(object):
    """neural_kernel_network_wrapper.py:28-61: hparams = [{'name': 'Linear', 'params': {...}}, ...]"""

    def __init__(self, hparams):
        self._LAYERS = dict(Linear=Linear, Product=Product, Activation=Activation)
        self._layers = [self._LAYERS[l['name']](**l['params']) for l in hparams]

    def forward(self, input):
        outputs = input        # [nm, k]
        for l in self._layers:
            outputs = l.forward(outputs)
        return outputs

    @property
    def parameters(self):
        params = []
        for l in self._layers:
            params = params + l.parameters
        return params

    def _nodes(self):
        nodes = []
        for i, l in enumerate(self._layers):
            nodes.extend(l._nodes(i))
        return nodes
