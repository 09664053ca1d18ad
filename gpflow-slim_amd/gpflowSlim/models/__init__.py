from .model import Model, GPModel
from .gpr import GPR
from .svgp import SVGP
from .sgpr import SGPR, GPRFITC
