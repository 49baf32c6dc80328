"""MI355X-native RBM / DBN contrastive-divergence training with the ``ku.ebm`` class surface.

    from keras_unsupervised_amd.ebm import RBM, DBN      # or: from ku.ebm import RBM, DBN

The HIP library (``csrc/libkurbm.so``, C ABI in ``include/kurbm.h``) is loaded on first use and
is mandatory: nothing in this package computes on the CPU.
"""
__version__ = "0.1.0"
