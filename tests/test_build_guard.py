"""Build-time guard of the hand-scheduled kernels (no GPU): what the shipped libdeepmod_hip.so contains, read from its gfx950 code
object (tools/isa_lint.py).  The product kernels run on a register budget without head-room and lean on inline asm the compiler's
hazard recogniser cannot see; a compiler update or a stray -D must fail HERE, not as a wrong probability on the GPU box.
Reference: none (the reference is TensorFlow 1.x graph execution, myMultiBiRNN.py:38-61) - this guards the from-scratch kernels."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)

from deepmod_amd import _lib  # noqa: E402

pytestmark = pytest.mark.skipif(not os.path.exists(_lib.LIB_PATH), reason="libdeepmod_hip.so not built")


@pytest.fixture(scope="module")
def report():
    import isa_lint
    if not os.path.exists(os.path.join(isa_lint.LLVM, "llvm-objdump")):
        pytest.skip("no llvm-objdump in this image")
    return isa_lint.report(_lib.LIB_PATH)


def test_no_experiment_switch_in_the_shipped_library():
    lib = _lib.load()
    assert lib.dm_build_flags().decode() == "experiment=0 switches=0"
    v = lib.dm_version().decode()
    assert "gfx950" in v and "hip " in v and "clang" in v.lower(), v


def test_a_stray_ablation_macro_is_a_compile_error():
    """-DDM16Q_ABL_NOCELL (a timing-only kernel) without -DDM_EXPERIMENT must not preprocess."""
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    src = os.path.join(ROOT, "deepmod_amd", "csrc", "deepmod_hip.hip")
    for macro in ("-DDM16Q_ABL_NOCELL", "-DDM16S_ABL_2PROD", "-DDM_ABL_NOEPI", "-DDM16Q_PRE=0", "-DDM_WITH_F16S"):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "--cuda-host-only", "-E", macro, src, "-o", os.devnull], capture_output=True, text=True)
        assert r.returncode != 0 and "DM_EXPERIMENT" in r.stderr, (macro, r.stderr[-300:])


@pytest.mark.parametrize("kernel", ["f16q", "f16qi8", "f32"])
def test_product_kernels_have_no_scratch_and_no_vgpr_spills(report, kernel):
    r = report[kernel]
    assert not r.get("missing"), "kernel symbol not found in the code object"
    res = r["resources"]
    assert res["private_segment_fixed_size"] == 0, res
    assert res["vgpr_spill_count"] == 0, res
    if kernel == "f16qi8":       # a few scalars (fold-scale addresses) parked in VGPR lanes (v_writelane: no memory traffic)
        assert res["sgpr_spill_count"] <= 8, res
    elif kernel != "f32":        # the fp32 kernel parks kernel arguments in VGPR lanes
        assert res["sgpr_spill_count"] == 0, res
    assert res["vgpr_count"] <= (512 if kernel != "f32" else 256), res      # one wave per SIMD / two


# MFMAs in the unrolled code of each kernel: a different count means the schedule (or the arithmetic) changed - re-run the GPU parity
# suite and the evidence script, then update.  f16q: 1,100 (the three stages of step 0) + 2,450 (one later step);
# f16qi8: f16 hi*hi + the mixed k32-steps, int8 cross terms; f32: 25 N-tiles x (5 + 3 + ...) k-steps
EXPECTED_MFMA = {"f16q": {"v_mfma_f32_16x16x32_f16": 3550}, "f16qi8": {"v_mfma_f32_16x16x32_f16": 1450, "v_mfma_i32_16x16x64_i8": 1050},
                 "f32": {"v_mfma_f32_16x16x4_f32": 500}}


def test_three_classifier_kernels_ship(report):
    """Round 6: the product is three classifier kernels - lstm16q::bilstm_f16q_kernel<0> (default), <1> (opt-in int8 cross terms), lstm32::bilstm_f32_kernel
    (fallback).  The 32x32x16 kernels of rounds 2-3 (lstm16s::) are compiled only into experiment builds (DM_WITH_F16S=1, tools/experiments/f16s)."""
    assert sorted(report) == ["f16q", "f16qi8", "f32"] and not any(r.get("missing") for r in report.values())
    blob = open(_lib.LIB_PATH, "rb").read()            # symbol names of the host stubs and of the embedded code object
    assert b"bilstm_f16q_kernel" in blob and b"bilstm_f32_kernel" in blob and b"bilstm_f16s_kernel" not in blob and b"lstm16s" not in blob
    assert not os.path.exists(os.path.join(ROOT, "deepmod_amd", "csrc", "lstm_f16s.hip.inc"))


@pytest.mark.parametrize("kernel", sorted(EXPECTED_MFMA))
def test_mfma_census_of_the_unrolled_body(report, kernel):
    assert report[kernel]["mfma"] == EXPECTED_MFMA[kernel]


@pytest.mark.parametrize("kernel", ["f16q", "f16qi8", "f32"])
def test_no_vector_instruction_reads_an_mfma_result_inside_the_hazard_window(report, kernel):
    """Every non-MFMA vector instruction that touches an MFMA's destination comes at least the required wait states later (hipcc pads
    its own code to exactly that; an inline-asm reader would show up below it, as round 4's v_min_f32 did)."""
    for opcode, t in report[kernel]["tightest_use"].items():
        assert t["wait_states"] >= t["required"], (opcode, t)


def test_profile_staleness_hash_covers_every_source_of_a_kernel(monkeypatch):
    """bench.py marks a committed rocprof / PMC summary `stale` when the kernel it was taken on has changed - the hash must cover
    every file the kernel's translation unit is compiled from (the unit, everything of csrc/ it includes, transitively) and the compiler flags."""
    import re
    import bench
    import __graft_entry__ as ge
    csrc = os.path.join(ROOT, "deepmod_amd", "csrc")

    def closure(name, seen):
        if name in seen:
            return seen
        seen.add(name)
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(os.path.join(csrc, name)).read(), flags=re.M):
            if os.path.dirname(inc) == "" and os.path.exists(os.path.join(csrc, inc)):
                closure(inc, seen)
        return seen
    for prec, unit in (("f16x3", "kern_f16q0.hip"), ("f16i8", "kern_f16q1.hip"), ("f32", "kern_f32.hip")):
        assert closure(unit, set()) <= set(bench.KERNEL_SOURCES[prec]), (prec, sorted(closure(unit, set()) - set(bench.KERNEL_SOURCES[prec])))
        assert set(ge.UNIT_DEPS[unit]) | {unit} >= closure(unit, set())            # ... and the build's own dependency list rebuilds the unit for each of them
    before = bench.kernel_source_sha("f16x3")
    monkeypatch.setattr(ge, "HIPCC_FLAGS", ge.HIPCC_FLAGS + ["-O2"])
    assert bench.kernel_source_sha("f16x3") != before
