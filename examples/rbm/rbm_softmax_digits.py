#!/usr/bin/env python3
"""RBM features -> softmax digit classifier, on MI355X.

The caller on one side of the hot path: the shape of the reference's
examples/rbm/rbm_softmax_mnist.py -- a `MNISTClassifier(conf)` with `train()` / `test()`, an RBM trained
unsupervised by `rbm.fit(V)` (reference :83), whose STOCHASTIC hidden features `rbm(x)` (reference :58,
ku/ebm/rbm.py:80-86) feed a 10-way softmax layer trained with Adam (reference :61-72, :87-91), and
`test()` writing `solution.csv` (reference :123-127).

Data: Kaggle `train.csv` / `test.csv` in the working directory if present (reference :98, :131);
otherwise scikit-learn's bundled 8x8 digits, scaled to [0,1] and nearest-neighbour upsampled to
28x28 = 784 pixels (MNIST itself is not available offline).  The RBM runs on the HIP kernels; the
softmax head is a torch Linear layer on the same device (it is not part of the hot path).
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from ku.ebm import RBM  # noqa: E402
from keras_unsupervised_amd.ebm import load_rbm, save_rbm  # noqa: E402


def load_digits_784():
    from sklearn.datasets import load_digits
    d = load_digits()
    img = (d.images / 16.0).astype(np.float32)                       # [n, 8, 8] in [0, 1]
    up = np.repeat(np.repeat(img, 4, axis=1), 4, axis=2)[:, 2:30, 2:30]   # 32x32 -> centre 28x28
    return up.reshape(len(up), 784), d.target.astype(np.int64)


class MNISTClassifier(object):
    """Digit classifier: RBM (unsupervised, CD) + softmax (supervised)."""

    MODEL_STEM = "digit_classification_model"
    IMAGE_SIZE = 784

    def __init__(self, conf, workdir="."):
        self.conf, self.workdir = conf, workdir
        self.hps, self.nn_arch = conf["hps"], conf["nn_arch"]
        self.device = torch.device("cuda", torch.cuda.current_device())
        stem = os.path.join(workdir, self.MODEL_STEM)
        if conf.get("model_loading"):
            self.rbm = load_rbm(stem + ".rbm")
            self.head = torch.nn.Linear(self.nn_arch["output_dim"], 10).to(self.device)
            self.head.load_state_dict(torch.load(stem + ".head.pt", map_location=self.device))
        else:
            self.rbm = RBM(conf["rbm_hps"], self.nn_arch["output_dim"], name="rbm",
                           mode=conf.get("rbm_mode", 1))            # reference default mode: Gaussian
            self.head = torch.nn.Linear(self.nn_arch["output_dim"], 10).to(self.device)

    # -- data -----------------------------------------------------------------------------
    def _load_training_data(self):
        path = os.path.join(self.workdir, "train.csv")
        if self.conf.get("data", "auto") != "digits" and os.path.exists(path):
            import pandas as pd
            df = pd.read_csv(path)
            return (df.iloc[:, 1:].values / 255.0).astype(np.float32), df.iloc[:, 0].values.astype(np.int64)
        V, y = load_digits_784()
        n = int(0.8 * len(V))
        return V[:n], y[:n]

    def _load_test_data(self):
        path = os.path.join(self.workdir, "test.csv")
        if self.conf.get("data", "auto") != "digits" and os.path.exists(path):
            import pandas as pd
            return (pd.read_csv(path).values / 255.0).astype(np.float32), None
        V, y = load_digits_784()
        n = int(0.8 * len(V))
        return V[n:], y[n:]

    # -- model ----------------------------------------------------------------------------
    def _features(self, Vt):
        """rbm(x): sampled hidden units, fresh draws on every call (the layer's forward pass)."""
        return self.rbm(Vt)

    def train(self, verbose=1):
        V, y = self._load_training_data()
        print("Train the RBM model.")
        self.rbm.fit(V, verbose=0)
        print("Train the NN model.")
        Vt = torch.from_numpy(V).to(self.device)
        yt = torch.from_numpy(y).to(self.device)
        opt = torch.optim.Adam(self.head.parameters(), lr=self.hps["lr"] * 10,
                               betas=(self.hps["beta_1"], self.hps["beta_2"]))
        bs = self.hps["batch_size"]
        for epoch in range(self.hps["epochs"]):
            tot = 0.0
            for lo in range(0, len(V), bs):
                x = self._features(Vt[lo:lo + bs].contiguous())
                loss = torch.nn.functional.cross_entropy(self.head(x), yt[lo:lo + bs])
                opt.zero_grad()
                loss.backward()
                opt.step()
                tot += float(loss.detach()) * len(x)
            if verbose:
                print("epoch %d/%d  loss %.4f" % (epoch + 1, self.hps["epochs"], tot / len(V)))
        print("Save the model.")
        stem = os.path.join(self.workdir, self.MODEL_STEM)
        save_rbm(self.rbm, stem + ".rbm")
        torch.save(self.head.state_dict(), stem + ".head.pt")

    def predict(self, V, n_draws=8):
        """Class probabilities, averaged over a few stochastic feature draws."""
        Vt = torch.from_numpy(np.ascontiguousarray(V, dtype=np.float32)).to(self.device)
        with torch.no_grad():
            p = sum(torch.softmax(self.head(self._features(Vt)), dim=1) for _ in range(n_draws)) / n_draws
        return p.cpu().numpy()

    def test(self):
        V, y = self._load_test_data()
        res = self.predict(V)
        with open(os.path.join(self.workdir, "solution.csv"), "w") as f:
            f.write("ImageId,Label\n")
            for i, v in enumerate(res):
                f.write(str(i + 1) + "," + str(int(np.argmax(v))) + "\n")
        acc = float((res.argmax(1) == y).mean()) if y is not None else None
        if acc is not None:
            print("held-out accuracy: %.4f" % acc)
        return acc


def main():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "rbm_softmax_conf.json")) as f:
        conf = json.load(f)
    mc = MNISTClassifier(conf)
    ts = time.time()
    if conf["mode"] == "train":
        mc.train()
        mc.test()
    else:
        mc.test()
    print("Elapsed time: {0:f}s".format(time.time() - ts))


if __name__ == "__main__":
    main()
