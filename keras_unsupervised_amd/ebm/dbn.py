"""Deep belief network: an ordered stack of RBMs trained greedily, layer by layer.

Host-side mirror of the reference ``ku.ebm.DBN`` (reference ku/ebm/dbn.py:11-95) with the
repairs listed in SURVEY.md 8(a): ``self.rbm_layer`` -> the argument / loop variable
(dbn.py:25, :28, :54-55, :94), ``rbm_layers`` -> ``_rbm_layers`` (dbn.py:27), and the empty
``range(len(layers), -1)`` of ``inv_transform`` -> a reverse walk (dbn.py:92).  All compute is in
the RBM layers; between layers the data stays on the device.
"""
from .engine import DeviceMatrix
from .rbm import _unwrap


def _dims(layer):
    """(n_in, n_out) of a layer; n_in is None until the layer is built."""
    n_in = layer.input_shape[1] if getattr(layer, "input_shape", None) else None
    out_shape = getattr(layer, "output_shape", None)
    n_out = out_shape[1] if out_shape else layer.output_dim
    return n_in, n_out


class DBN(object):
    """Deep belief network."""

    def add_stack(self, rbm_layer):
        """Append an RBM; its input width must equal the previous layer's output width
        (dbn.py:14-32).  An unbuilt layer has no input width yet and is checked when it is built
        by fit / transform."""
        if hasattr(self, "_rbm_layers"):
            n_in, _ = _dims(rbm_layer)
            _, prev_out = _dims(self._rbm_layers[-1])
            if n_in is not None and prev_out != n_in:
                raise ValueError("A previous RBM layer's output dimension must "
                                 "be equal to a next one's input dimension.")      # dbn.py:29-30
            self._rbm_layers.append(rbm_layer)
        else:
            self._rbm_layers = [rbm_layer]

    def _layers(self):
        if not hasattr(self, "_rbm_layers"):
            raise ValueError("Any rbm layer doesn't exist.")                       # dbn.py:47-48
        return self._rbm_layers

    @staticmethod
    def _to_device(layer, X):
        X = _unwrap(X)
        if isinstance(X, DeviceMatrix):
            return X, "device"
        layer._ensure_built(X.shape[1])
        return layer._as_device(X)

    def fit(self, V, verbose=1):
        """Greedy layer-wise training (dbn.py:34-55): fit layer l on V_p, then
        V_p <- layer.transform(V_p) -- the next layer sees SAMPLED hidden states."""
        layers = self._layers()
        V_p, _ = self._to_device(layers[0], V)      # the reference's V.copy(): V itself is never written
        for rbm_layer in layers:
            print("Train {0:s}.".format(str(rbm_layer.name)))                      # dbn.py:53
            rbm_layer.fit(V_p, verbose=verbose)
            V_p = rbm_layer.transform(V_p)

    def transform(self, V):
        """Chain of layer.transform (dbn.py:57-75)."""
        layers = self._layers()
        V_p, kind = self._to_device(layers[0], V)
        for rbm_layer in layers:
            V_p = rbm_layer.transform(V_p)
        return layers[0]._ret(V_p, kind, False)

    def inv_transform(self, H):
        """Reverse chain of layer.inv_transform (dbn.py:77-95, loop repaired)."""
        layers = self._layers()
        H_p, kind = self._to_device(layers[-1], H) if layers[-1].built else (None, None)
        if H_p is None:
            raise ValueError("inv_transform needs built RBM layers")
        for rbm_layer in reversed(layers):
            H_p = rbm_layer.inv_transform(H_p)
        return layers[0]._ret(H_p, kind, False)
