"""GPU (-m gpu): stream-ordering stress (tests/stress_stream_order.py).  The same loop with asynchronous launches and with every
kernel serialised (AMD_SERIALIZE_KERNEL=3, HIP_LAUNCH_BLOCKING=1) must give the same bits: an ordering bug shows up as a
difference between the two (or as a wrong counter against the oracle inside the loop)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(tmp_path, tag, iters, big_every, env_extra):
    env = dict(os.environ, **env_extra)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'stress_stream_order.py'), str(iters), str(big_every), str(tmp_path / ('rdv_' + tag))],
                         env=env, capture_output=True, text=True, timeout=1500)
    assert res.returncode == 0, res.stdout[-1000:] + res.stderr[-3000:]
    return json.loads([l for l in res.stdout.splitlines() if l.startswith('{')][-1])


def test_create_follow_add_grow_reduce_fetch_destroy_loop(tmp_path, gpu_device):
    a = _run(tmp_path, 'async', 200, 10, {})                                   # 200 iterations, every 10th on 3e8 positions (3.6 GB of counters)
    b = _run(tmp_path, 'serial', 200, 10, {'AMD_SERIALIZE_KERNEL': '3', 'AMD_SERIALIZE_COPY': '3', 'HIP_LAUNCH_BLOCKING': '1'})
    assert a['iterations'] == b['iterations'] == 200
    assert a['digest'] == b['digest']
