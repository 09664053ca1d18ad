"""Covariance functions of the exact-GP path, evaluated on the GPU.

API mirror of gpflowSlim/kernels.py for the kernels on the hot path: Kernel (active-dims
slicing :217-253, ``+`` / ``*`` :277-281), Static/White/Constant/Bias :308-357, Stationary
:360-429, RBF :432-439, Exponential :557-565, Matern12/32/52 :569-610, Periodic :769-819,
Combination/Sum/Product :1000-1084.

Differences from the reference, on purpose: values are eager numpy fp64 arrays (there is no
TensorFlow graph); ``K`` compiles the kernel *tree* into one reverse-Polish program and
evaluates it in a single fused HIP pass (csrc/kmat.hip) instead of one TF op per arithmetic
step; ``name=None`` is accepted for Static kernels (the reference crashes on it).
"""
from functools import reduce

import numpy as np

from . import transforms
from . import _backend as be
from ._settings import settings
from .params import Parameter


class Kernel(object):
    """kernels.py:33-76"""

    def __init__(self, input_dim, active_dims=None, name=None):
        self._name = name
        self.input_dim = int(input_dim)
        if active_dims is None:
            self.active_dims = slice(input_dim)
        elif isinstance(active_dims, slice):
            self.active_dims = active_dims
            if active_dims.start is not None and active_dims.stop is not None and active_dims.step is not None:
                assert len(range(active_dims.start, active_dims.stop, active_dims.step)) == input_dim
        else:
            self.active_dims = np.array(active_dims, dtype=np.int32)
            assert len(active_dims) == input_dim
        self._parameters = []

    @property
    def name(self):
        return self._name

    @property
    def parameters(self):
        return self._parameters

    # ---- program construction -------------------------------------------------------------
    def _dims(self, presliced, d_all):
        """Column indices this kernel reads (Kernel._slice, kernels.py:238-245)."""
        if presliced:
            return list(range(self.input_dim))
        if isinstance(self.active_dims, slice):
            return list(range(*self.active_dims.indices(d_all)))
        return [int(d) for d in self.active_dims]

    def _nodes(self, presliced, d_all):
        raise NotImplementedError

    def _program(self, d_all, presliced=False):
        return be.make_program(self._nodes(presliced, d_all))

    def _grad_layout(self, d_all):
        """One entry per gradient slot of gps_gpr_lml_grad, in slot order: (Parameter or None, index
        into the parameter's flattened value or None = add to every element)."""
        raise NotImplementedError

    # ---- reference API -------------------------------------------------------------------
    def K(self, X, X2=None, presliced=False):
        X = np.asarray(X, dtype=settings.float_type)
        prog = self._program(X.shape[1], presliced)
        return be.get_handle().kmat(prog, X, None if X2 is None else np.asarray(X2, dtype=settings.float_type))

    def Kdiag(self, X, presliced=False):
        raise NotImplementedError

    # eager helpers of the reference (kernels.py:68-75): there they build the tensor and evaluate it; here K / Kdiag are eager
    def compute_K(self, X, Z):
        return self.K(X, Z)

    def compute_K_symm(self, X):
        return self.K(X)

    def compute_Kdiag(self, X):
        return self.Kdiag(X)

    def _slice(self, X, X2):
        """kernels.py:217-253: the columns this kernel reads (host-side view; the device kernels gather the active dims
        themselves from the un-sliced X)."""
        X = np.asarray(X, dtype=settings.float_type)
        dims = self._dims(False, X.shape[1])
        Xs = X[:, dims]
        X2s = None if X2 is None else np.asarray(X2, dtype=settings.float_type)[:, dims]
        if Xs.shape[1] != self.input_dim:
            raise ValueError("Input 1st dimension does not match kernel dimension.")
        return Xs, X2s

    def Kdim(self, dim, X, X2=None):
        """kernels.py:287-306: the covariance along one input dimension -- X [n, 1] is placed in column `dim` of an otherwise
        zero [n, input_dim] input."""
        def pad(a):
            a = np.asarray(a, dtype=settings.float_type)
            out = np.zeros((a.shape[0], self.input_dim), dtype=settings.float_type)
            out[:, dim] = a[:, 0]
            return out
        return self.K(pad(X), None if X2 is None else pad(X2), presliced=True)

    def __add__(self, other):
        return Sum([self, other])

    def __mul__(self, other):
        return Product([self, other])


class Static(Kernel):
    """kernels.py:308-325"""

    def __init__(self, input_dim, variance=1.0, active_dims=None, name=None):
        super().__init__(input_dim, active_dims, name=name)
        self._variance = Parameter(variance, transform=transforms.positive, name='variance')
        self._parameters = self._parameters + [self._variance]

    @property
    def variance(self):
        return self._variance.value

    def Kdiag(self, X, presliced=False):
        return np.ones(np.shape(X)[0], dtype=settings.float_type) * self.variance


class White(Static):
    """kernels.py:328-338"""

    def _grad_layout(self, d_all):
        return [(self._variance, None)]

    def _nodes(self, presliced, d_all):
        return [be.primitive_node(be.K_WHITE, np.squeeze(self.variance))]


class Constant(Static):
    """kernels.py:341-350"""

    def _grad_layout(self, d_all):
        return [(self._variance, None)]

    def _nodes(self, presliced, d_all):
        return [be.primitive_node(be.K_CONSTANT, np.squeeze(self.variance))]


class Bias(Constant):
    """kernels.py:353-357"""
    pass


class Stationary(Kernel):
    """kernels.py:360-429"""
    _op = None

    def __init__(self, input_dim, variance=1.0, lengthscales=None,
                 active_dims=None, ARD=False, min_ls=1e-6, name='kernel'):
        super().__init__(input_dim, active_dims, name=name)
        self._variance = Parameter(variance, transform=transforms.positive, name='variance',
                                   dtype=settings.float_type)
        if ARD:
            if lengthscales is None:
                lengthscales = np.ones(input_dim, dtype=settings.float_type)
            else:
                lengthscales = lengthscales * np.ones(input_dim, dtype=settings.float_type)
        else:
            lengthscales = 1.0 if lengthscales is None else lengthscales
        self.ARD = ARD
        self._ls = Parameter(lengthscales, transform=transforms.Log1pe(min_ls), name='ls')
        self._parameters = self._parameters + [self._variance, self._ls]

    @property
    def variance(self):
        return self._variance.value

    @property
    def lengthscales(self):
        return self._ls.value

    def _nodes(self, presliced, d_all):
        dims = self._dims(presliced, d_all)
        ls = np.atleast_1d(self.lengthscales)
        if ls.size not in (1, len(dims)):
            raise ValueError("lengthscales do not match the active dims")
        return [be.primitive_node(self._op, np.squeeze(self.variance), dims, ls)]

    def Kdiag(self, X, presliced=False):
        """kernels.py:428-429"""
        return np.ones(np.shape(X)[0], dtype=settings.float_type) * self.variance

    def _grad_layout(self, d_all):
        nd = len(self._dims(False, d_all))
        ard = np.atleast_1d(self.lengthscales).size > 1
        return [(self._variance, None)] + [(self._ls, d if ard else None) for d in range(nd)]

    def _dist(self, op, X, X2):
        # the reference calls these on already-sliced inputs (K slices first, kernels.py:436-438): columns 0 .. input_dim - 1
        X = np.asarray(X, dtype=settings.float_type)
        ls = np.atleast_1d(self.lengthscales)
        node = be.primitive_node(op, 1.0, list(range(X.shape[1])), ls)
        return be.get_handle().kmat(be.make_program([node]), X, None if X2 is None else np.asarray(X2, dtype=settings.float_type))

    def square_dist(self, X, X2):
        """kernels.py:408-421: max(0, |a|^2 + |b|^2 - 2 a.b) with a = X / lengthscales -- evaluated by the same device tile pass
        (program op GPS_K_SQDIST) every stationary K is built on."""
        return self._dist(be.K_SQDIST, X, X2)

    def euclid_dist(self, X, X2):
        """kernels.py:424-426: sqrt(square_dist + 1e-12)"""
        return self._dist(be.K_EUCLID, X, X2)

    def dimwise(self, dim):
        """kernels.py:441-444, 579-582, ...: the one-dimensional factor of this kernel along `dim` (variance^(1 / input_dim))."""
        ls = np.atleast_1d(self.lengthscales)
        return type(self)(input_dim=1, variance=float(np.squeeze(self.variance)) ** (1.0 / self.input_dim),
                          lengthscales=float(ls[dim] if self.ARD else ls[0]), name='%s_dimwise_%d' % (type(self).__name__, dim))


class RBF(Stationary):
    """kernels.py:432-439"""
    _op = be.K_RBF


class Exponential(Stationary):
    """kernels.py:557-565"""
    _op = be.K_EXPONENTIAL


class Matern12(Stationary):
    """kernels.py:569-577"""
    _op = be.K_MATERN12


class Matern32(Stationary):
    """kernels.py:585-594"""
    _op = be.K_MATERN32


class Matern52(Stationary):
    """kernels.py:601-610"""
    _op = be.K_MATERN52


class Periodic(Kernel):
    """kernels.py:769-819 (no ARD for lengthscale or period, as in the reference)"""

    def __init__(self, input_dim, period=1.0, variance=1.0, lengthscales=1.0, active_dims=None, name='kernel'):
        super().__init__(input_dim, active_dims, name=name)
        self._variance = Parameter(variance, transform=transforms.positive, name='variance')
        self._ls = Parameter(lengthscales, transform=transforms.positive, name='ls')
        self._period = Parameter(period, transform=transforms.positive, name='period')
        self._parameters = self._parameters + [self._variance, self._ls, self._period]

    @property
    def variance(self):
        return self._variance.value

    @property
    def lengthscales(self):
        return self._ls.value

    @property
    def period(self):
        return self._period.value

    def _nodes(self, presliced, d_all):
        dims = self._dims(presliced, d_all)
        return [be.primitive_node(be.K_PERIODIC, np.squeeze(self.variance), dims,
                                  float(np.squeeze(self.lengthscales)), period=float(np.squeeze(self.period)))]

    def Kdiag(self, X, presliced=False):
        """kernels.py:803-804"""
        return np.full(np.shape(X)[0], np.squeeze(self.variance), dtype=settings.float_type)

    def _grad_layout(self, d_all):
        return [(self._variance, None), (self._ls, None), (self._period, None)]


_SCALARS = (int, float, np.floating, np.integer)


class Combination(Kernel):
    """kernels.py:1000-1057"""
    _fold_op = None

    def __init__(self, kern_list, name='kernel'):
        extra_dims = np.asarray([], dtype=int)
        active_dims = reduce(np.union1d, (np.r_[x.active_dims] for x in kern_list if isinstance(x, Kernel)),
                             extra_dims)
        input_dim = active_dims.size
        super().__init__(input_dim=input_dim, name=name, active_dims=active_dims)
        # flatten instances of the same class (kernels.py:1019-1029)
        self.kern_list = []
        self.const_list = []
        for k in kern_list:
            if isinstance(k, self.__class__):
                self.kern_list.extend(k.kern_list)
                self.const_list.extend(k.const_list)
            elif isinstance(k, _SCALARS):
                self.const_list.append(float(k))
            elif isinstance(k, Kernel):
                self.kern_list.append(k)
            else:
                raise TypeError("can only combine Kernel instances and scalars")
        for kern in kern_list:
            if isinstance(kern, Kernel):
                self._parameters = self._parameters + kern.parameters

    @property
    def on_separate_dimensions(self):
        """kernels.py:1039-1057"""
        if np.any([isinstance(k.active_dims, slice) for k in self.kern_list]):
            return False
        dimlist = [k.active_dims for k in self.kern_list]
        overlapping = False
        for i, dims_i in enumerate(dimlist):
            for dims_j in dimlist[i + 1:]:
                if np.any(dims_i.reshape(-1, 1) == dims_j.reshape(1, -1)):
                    overlapping = True
        return not overlapping

    def _nodes(self, presliced, d_all):
        # children slice from the *original* X (kernels.py:1073), left fold in list order and then
        # over const_list (reduce(op, [K1, K2, ...] + const_list))
        nodes = []
        first = True
        for k in self.kern_list:
            nodes.extend(k._nodes(False, d_all))
            if not first:
                nodes.append(be.op_node(self._fold_op))
            first = False
        for c in self.const_list:
            nodes.append(be.primitive_node(be.K_CONSTANT, c))
            if not first:
                nodes.append(be.op_node(self._fold_op))
            first = False
        return nodes

    def _fold(self, values):
        raise NotImplementedError

    def _grad_layout(self, d_all):
        out = []
        for k in self.kern_list:
            out.extend(k._grad_layout(d_all))
        out.extend([(None, None)] * len(self.const_list))     # scalar constants: slot exists, no parameter
        return out

    def Kdiag(self, X, presliced=False):
        return self._fold([k.Kdiag(X) for k in self.kern_list] + self.const_list)


class Sum(Combination):
    """kernels.py:1071-1076"""
    _fold_op = be.K_ADD

    def _fold(self, values):
        return reduce(np.add, values)


class Product(Combination):
    """kernels.py:1079-1084"""
    _fold_op = be.K_MUL

    def _fold(self, values):
        return reduce(np.multiply, values)


# the reference also registers these spellings
Add = Sum
Prod = Product
