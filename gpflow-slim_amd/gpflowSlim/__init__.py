"""gpflowSlim -- MI355X-native implementation of GPflow-Slim's exact-GP hot path.

Drop-in for the reference's ``gpflowSlim.kernels`` / ``gpflowSlim.models.GPR`` / ``predict_f`` /
``gpflowSlim.conditionals`` surface; all O(N^2) and O(N^3) work runs in hand-written HIP kernels
reached through a C ABI (include/gpflowslim_hip.h).  No TensorFlow, no CPU fallback.
"""
from ._settings import settings
from . import transforms
from . import params
from . import kernels
from . import mean_functions
from . import densities
from . import likelihoods
from . import conditionals
from . import features
from . import kullback_leiblers
from . import models
from . import neural_kernel_network
from ._backend import NotPositiveDefiniteError, get_handle, set_handle, Handle, load_library

__version__ = "0.1.0"
