"""Fixture generator (dev container only; reads /root/reference): the DATA files of a shipped model directory that are small
enough to commit - the TF bundle `.index` (tensor names, shapes, offsets: 1.7 KB) and the `checkpoint` state file - so that a
test can rebuild the directory `train_deepmod/<model>/` with a synthetic `.data-00000-of-00001` of the real byte layout
(the real weight shards are absent from the reference checkout: .MISSING_LARGE_BLOBS).  No source text is copied.

    python tests/golden/make_golden_model_dir.py
"""
import os
import shutil

REF = "/root/reference/train_deepmod"
HERE = os.path.dirname(os.path.abspath(__file__))
MODELS = {"rnn_conmodA_E1m2wd21_f7ne1u0_4": "mod_train_conmodA_E1m2wd21_f3ne1u0",       # BASELINE.json configs[4]: the 6mA model
          "rnn_conmodC_P100wd21_f7ne1u0_4": "mod_train_conmodC_P100wd21_f3ne1u0"}       # configs[0] / [1]: the 5mC model

for model, prefix in MODELS.items():
    out = os.path.join(HERE, "model_dirs", model)
    os.makedirs(out, exist_ok=True)
    for name in (prefix + ".index", "checkpoint"):
        shutil.copyfile(os.path.join(REF, model, name), os.path.join(out, name))
        print("wrote", os.path.join(out, name), os.path.getsize(os.path.join(out, name)), "bytes")
