"""CPU: the host-only part of the C ABI (alignment walk, row builder of a worker batch, BED formatter: readmap.inc, rowsbatch.inc,
bedtext.inc) rebuilt by gcc with -fsanitize=address,undefined (tests/asan/host_shim.cpp) and driven by tests/asan/fuzz_host.py with
valid batches and with damaged container tables - offsets that decrease or run past their arrays, truncated columns, absurd clips,
contigs, strands, CIGARs.  Container files come from disk: a damaged one must become an error code (and a line of the error ledger),
never an out-of-bounds access.  A sanitizer report aborts the driver."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT


def test_host_abi_under_address_sanitizer(tmp_path):
    gxx = shutil.which('g++')
    asan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip() if shutil.which('gcc') else ''
    if not gxx or not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip('g++ / libasan not available')
    shim = str(tmp_path / 'libdm_host_asan.so')
    build = subprocess.run([gxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-fno-omit-frame-pointer',
                            '-shared', '-fPIC', '-pthread', '-o', shim, os.path.join(ROOT, 'tests', 'asan', 'host_shim.cpp')],
                           capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    scratch = str(tmp_path / 'scratch')
    os.makedirs(scratch)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1:print_stacktrace=1')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'asan', 'fuzz_host.py'), shim, scratch, '250'], env=env, capture_output=True,
                         text=True, timeout=1500)
    assert res.returncode == 0 and 'FUZZ-OK' in res.stdout, res.stdout[-1500:] + res.stderr[-4000:]
    import re
    m = re.search(r"'resident_ok': (\d+)", res.stdout)           # round 6: the resident form (no per-event values) went through the same damaged records
    assert m and int(m.group(1)) > 20, res.stdout[-1500:]
