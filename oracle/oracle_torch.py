"""ORACLE — TEST INFRASTRUCTURE ONLY: the BiLSTM graph as torch-CPU library calls (addmm = library sgemm, vectorised sigmoid / tanh,
intra-op threads).  Only bench.py's `cpu_baseline.gemm` leg imports this (a fairer stand-in for what TensorFlow-CPU / Eigen does with
the graph than the scalar-ish C loop nest: VERDICT r03 item 6); the product (deepmod_amd/) never does, and nothing here touches a GPU.

Restates /root/reference/bin/DeepMod_scripts/myMultiBiRNN.py:38-61 exactly as oracle_np.predict_windows_np does (gate order i, j, f, o;
forget bias +1.0; zero state; 11 live steps per direction; head on the centre output) - checked against the C oracle by
tests/test_oracle_golden.py::test_torch_restatement_equals_c_oracle."""
from __future__ import annotations

from typing import Dict

import numpy as np

from .oracle_np import HEAD_B, HEAD_W, HID, LIVE, WIN, cell_name


class TorchGraph:
    def __init__(self, weights: Dict[str, np.ndarray], threads: int):
        import torch
        self.torch = torch
        torch.set_num_threads(max(1, int(threads)))
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
        self.cells = {(d, l): (t(weights[cell_name(d, l, "kernel")]), t(weights[cell_name(d, l, "bias")])) for d in ("fw", "bw") for l in range(3)}
        self.head_w, self.head_b = t(weights[HEAD_W]), t(weights[HEAD_B])

    def predict(self, x: np.ndarray) -> np.ndarray:
        torch = self.torch
        with torch.inference_mode():
            xt = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32))
            n = xt.shape[0]
            finals = []
            for d in ("fw", "bw"):
                h = [xt.new_zeros(n, HID) for _ in range(3)]
                c = [xt.new_zeros(n, HID) for _ in range(3)]
                for s in range(LIVE):
                    inp = xt[:, s if d == "fw" else WIN - 1 - s, :]
                    for l in range(3):
                        kern, bias = self.cells[(d, l)]
                        g = torch.addmm(bias, torch.cat([inp, h[l]], 1), kern)
                        gi, gj, gf, go = g.split(HID, 1)
                        c[l] = c[l] * torch.sigmoid(gf + 1.0) + torch.sigmoid(gi) * torch.tanh(gj)
                        h[l] = torch.tanh(c[l]) * torch.sigmoid(go)
                        inp = h[l]
                finals.append(h[2])
            return torch.softmax(torch.addmm(self.head_b, torch.cat(finals, 1), self.head_w), 1).numpy()
