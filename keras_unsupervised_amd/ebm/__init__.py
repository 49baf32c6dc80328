"""Energy-based models of the reference's ``ku.ebm`` package, on MI355X."""
from .engine import MODE_COMPLEX, MODE_VISIBLE_BERNOULLI, MODE_VISIBLE_GAUSSIAN
from .rbm import RBM
from .dbn import DBN
from .checkpoint import load_dbn, load_rbm, save_dbn, save_rbm

__all__ = ["RBM", "DBN", "save_rbm", "load_rbm", "save_dbn", "load_dbn", "MODE_VISIBLE_BERNOULLI", "MODE_VISIBLE_GAUSSIAN", "MODE_COMPLEX"]
