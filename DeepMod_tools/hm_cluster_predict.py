#!/usr/bin/env python
"""Same command line as the reference's DeepMod_tools/hm_cluster_predict.py:
    python hm_cluster_predict.py <prefix of the merged per-chromosome BED files> <motif folder> [cluster-model prefix]
The reference hard-wires the model path (:85: train_deepmod/na12878_cluster_train_mod-keep_prob0.7-nb25-chr1/Cg.cov5.nb25
relative to its own checkout).  The checkpoint is not shipped with this repository, so the prefix is the third argument or
the DEEPMOD_CLUSTER_MODEL environment variable; the reference's relative location is only tried last."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from deepmod_amd import cluster  # noqa: E402

if len(sys.argv) < 3:
    print("Usage: python {} pred-prefix motif-folder [cluster-model-prefix]".format(sys.argv[0]))
    sys.exit(1)
model = sys.argv[3] if len(sys.argv) > 3 else os.environ.get('DEEPMOD_CLUSTER_MODEL') or os.path.join(
    ROOT, 'train_deepmod', 'na12878_cluster_train_mod-keep_prob0.7-nb25-chr1', 'Cg.cov5.nb25')
if not os.path.isfile(model + '.index'):
    sys.exit("cluster model checkpoint %r not found: pass its prefix as the third argument or set DEEPMOD_CLUSTER_MODEL "
             "(the reference ships it under train_deepmod/na12878_cluster_train_mod-keep_prob0.7-nb25-chr1/)" % model)
cluster.hm_cluster_predict(sys.argv[1], sys.argv[2], model)
