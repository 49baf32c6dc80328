"""Energy-based models of the reference's ``ku.ebm`` package, on MI355X."""
from .engine import MODE_COMPLEX, MODE_VISIBLE_BERNOULLI, MODE_VISIBLE_GAUSSIAN
from .rbm import RBM
from .dbn import DBN

__all__ = ["RBM", "DBN", "MODE_VISIBLE_BERNOULLI", "MODE_VISIBLE_GAUSSIAN", "MODE_COMPLEX"]
