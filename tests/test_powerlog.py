"""CPU: deepmod_amd/powerlog.py (bench.py's on-box socket power / shader clock sampler) against a fake hwmon directory: device lookup by
PCI bus id, unit conversion (microwatts -> W, Hz -> MHz), the summary over a time window, and the no-device answer.  No reference
counterpart (measurement plumbing of bench.py's roofline.power)."""
import os
import time

from deepmod_amd import powerlog


def _fake_device(tmp_path, power_uw=1350000000, sclk_hz=2200000000, average=False):
    h = tmp_path / 'sys' / 'bus' / 'pci' / 'devices' / '0000:a4:00.0' / 'hwmon' / 'hwmon7'
    os.makedirs(h)
    (h / ('power1_average' if average else 'power1_input')).write_text('%d\n' % power_uw)
    (h / 'freq1_input').write_text('%d\n' % sclk_hz)
    (h / 'power1_cap').write_text('1400000000\n')
    return h


def test_power_and_clock_are_sampled_from_the_devices_hwmon_files(tmp_path, monkeypatch):
    h = _fake_device(tmp_path)
    real_glob = powerlog.glob.glob
    monkeypatch.setattr(powerlog.glob, 'glob', lambda pat: real_glob(str(tmp_path) + pat) if pat.startswith('/sys/') else real_glob(pat))
    monkeypatch.setattr(powerlog.os.path, 'exists', lambda p, _e=os.path.exists: _e(p))
    assert powerlog.hwmon_dir('0000:A4:00.0') == str(h)               # the runtime prints the bus id in either case
    log = powerlog.PowerLog('0000:a4:00.0', period_s=0.001)
    assert log.available and log.cap_w() == 1400.0
    t0 = time.perf_counter()
    with log:
        time.sleep(0.05)
        (h / 'power1_input').write_text('1000000000\n')
        time.sleep(0.05)
    t1 = time.perf_counter()
    s = log.summary(t0, t1)
    assert s['available'] and s['samples'] >= 10 and s['socket_power_cap_w'] == 1400.0
    assert s['socket_power_w']['max'] == 1350.0 and s['socket_power_w']['min'] == 1000.0
    assert s['sclk_mhz']['median'] == 2200.0
    late = log.summary(t0, t1, skip_s=0.07)                            # the second half only
    assert late['samples'] >= 1 and late['socket_power_w']['max'] == 1000.0


def test_older_drivers_power1_average_and_a_box_without_the_device(tmp_path, monkeypatch):
    _fake_device(tmp_path, average=True)
    real_glob = powerlog.glob.glob
    monkeypatch.setattr(powerlog.glob, 'glob', lambda pat: real_glob(str(tmp_path) + pat) if pat.startswith('/sys/') else real_glob(pat))
    log = powerlog.PowerLog('0000:a4:00.0')
    assert log.available and log.power_file.endswith('power1_average')
    none = powerlog.PowerLog('0000:ff:00.0')
    assert not none.available
    with none:
        pass
    assert none.summary() == {'available': False, 'why': "no amdgpu hwmon power file for PCI device '0000:ff:00.0'"}
    assert not powerlog.PowerLog(None).available
