from .neural_kernel_network import NeuralKernelNetwork
from .neural_kernel_network_wrapper import NKNWrapper, Linear, Product, Activation
