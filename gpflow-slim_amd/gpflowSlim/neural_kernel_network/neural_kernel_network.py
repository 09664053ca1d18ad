"""NeuralKernelNetwork kernel: primitive kernels -> Linear / Product / Activation layers.

Mirrors gpflowSlim/neural_kernel_network/neural_kernel_network.py:24-47.  K(X, X2) is one fused HIP
pass (all primitives and the whole network per matrix entry, no [N*M, k] stack in memory).
"""
import numpy as np

from ..kernels import Kernel, Combination


class NeuralKernelNetwork(Kernel):
    def __init__(self, input_dim, primitive_kernels, nknWrapper):
        super(NeuralKernelNetwork, self).__init__(input_dim)
        self._primitive_kernels = primitive_kernels
        self._nknWrapper = nknWrapper
        self._parameters = self._parameters + self._nknWrapper.parameters
        for kern in self._primitive_kernels:
            if isinstance(kern, Combination):
                raise NotImplementedError("NKN primitives must be primitive kernels (no Sum / Product) on the device path")
            self._parameters = self._parameters + kern.parameters

    def Kdiag(self, X, presliced=False):
        """neural_kernel_network.py:35-39"""
        primitive_values = np.stack([kern.Kdiag(X, presliced) for kern in self._primitive_kernels], 1)
        return np.squeeze(self._nknWrapper.forward(primitive_values), -1)

    def _nodes(self, presliced, d_all):
        nodes = []
        for kern in self._primitive_kernels:
            nodes.extend(kern._nodes(presliced, d_all))
        return nodes + self._nknWrapper._nodes()

    def _grad_layout(self, d_all):
        """Slot order of csrc/grad_general.hip: the primitives' slots, then per Linear layer and output o the row
        W[o, :] and bias[o] (the parameters the reference trains through autodiff,
        neural_kernel_network_wrapper.py:90-120)."""
        out = []
        for kern in self._primitive_kernels:
            out.extend(kern._grad_layout(d_all))
        for layer in self._nknWrapper._layers:
            if hasattr(layer, "_weights"):
                for o in range(layer.output_dim):
                    out.extend((layer._weights, o * layer.input_dim + j) for j in range(layer.input_dim))
                    out.append((layer._bias, o))
        return out
