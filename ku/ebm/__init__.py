"""``ku.ebm`` -> ``keras_unsupervised_amd.ebm`` (same names, same call signatures)."""
from keras_unsupervised_amd.ebm import (DBN, MODE_COMPLEX, MODE_VISIBLE_BERNOULLI,  # noqa: F401
                                        MODE_VISIBLE_GAUSSIAN, RBM)
