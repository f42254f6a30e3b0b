"""Dev tool: build ablated variants of the BiLSTM kernel (timing only - results are wrong by
construction) and time them on the GPU box.

    python tools/ablate.py build            # here (cross-compile)
    python tools/ablate.py run              # on the GPU box (via gpurun)

DM_ABL_PREC picks the kernel the variants are timed on: 0 fp32 (lstm_f32.hip.inc, DM_ABL_* macros), 1 split-f16 step-major
(lstm_f16s.hip.inc, DM16S_* macros, variants s_*), 2 split-f16 layer-major (lstm_f16.hip.inc, DM16_* macros).
"""
import os, subprocess, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ABL = os.path.join(ROOT, "tools", "_abl")
VARIANTS = {
    "base": [],
    "timing": ["-DDM_TIMING"],
    "noepi": ["-DDM_ABL_NOEPI"],
    "noseq": ["-DDM_ABL_NOSEQ"],
    "nobar": ["-DDM_ABL_NOBAR"],
    "nodma": ["-DDM_ABL_NODMA"],
    "noldsb": ["-DDM16_ABL_NOLDSB"], "nodma16": ["-DDM16_ABL_NODMA"], "noldsb_nodma": ["-DDM16_ABL_NOLDSB", "-DDM16_ABL_NODMA"],
    "bdepth1": ["-DDM16_BDEPTH=1"], "bdepth3": ["-DDM16_BDEPTH=3"], "bdepth4": ["-DDM16_BDEPTH=4"],
    "tiles3": ["-DDM_TRACE2=3"], "tiles0": ["-DDM_TRACE2=0"], "tiles6": ["-DDM_TRACE2=6"],
    "nobar16": ["-DDM16_ABL_NOBAR"], "nobar_nodma16": ["-DDM16_ABL_NOBAR", "-DDM16_ABL_NODMA"], "skeleton16": ["-DDM16_ABL_NOBAR", "-DDM16_ABL_NODMA", "-DDM16_ABL_NOLDSB"],
    "noldsb_nz": ["-DDM16_ABL_NOLDSB", "-DDM16_ABL_NOLDSB_NONZERO"],
    "defer": ["-DDM16_DEFER"], "defer_trace": ["-DDM16_DEFER", "-DDM_TRACE"],
    "piece2": ["-DDM16_PIECE_EVERY=2"], "piece3": ["-DDM16_PIECE_EVERY=3"],
    "prod1": ["-DDM16_ABL_1PROD"], "noseq16": ["-DDM16_ABL_NOSEQ"],
    "floor16": ["-DDM16_ABL_1PROD", "-DDM16_ABL_NOCELL", "-DDM16_ABL_NOSEQ", "-DDM16_ABL_NODMA", "-DDM16_ABL_NOBAR", "-DDM16_ABL_NOLDSB", "-DDM16_ABL_NOLDSB_NONZERO"],
    "prod2": ["-DDM16_ABL_2PROD"], "nocell16": ["-DDM16_ABL_NOCELL"], "prod2_nocell": ["-DDM16_ABL_2PROD", "-DDM16_ABL_NOCELL"],
    # lstm_f16s.hip.inc (DM_ABL_PREC=1)
    "s_nocell": ["-DDM16S_ABL_NOCELL"], "s_nodma": ["-DDM16S_ABL_NODMA"], "s_nobar": ["-DDM16S_ABL_NOBAR"], "s_prod2": ["-DDM16S_ABL_2PROD"],
    "s_noldsa": ["-DDM16S_ABL_NOLDSA"], "s_b64": ["-DDM16S_ABL_B64"], "s_hionly": ["-DDM16S_ABL_LO_ONLY"], "s_nobar_nodma": ["-DDM16S_ABL_NOBAR", "-DDM16S_ABL_NODMA"],
    "s_floor": ["-DDM16S_ABL_NOCELL", "-DDM16S_ABL_NODMA", "-DDM16S_ABL_NOBAR", "-DDM16S_ABL_NOLDSA"],
    "s_ad2": ["-DDM16S_ADIST=2"], "s_ad4": ["-DDM16S_ADIST=4"], "s_ad5": ["-DDM16S_ADIST=5"], "s_ad6": ["-DDM16S_ADIST=6"], "s_pre0": ["-DDM16S_PRE=0"], "s_pre1": ["-DDM16S_PRE=1"], "s_pre3": ["-DDM16S_PRE=3"], "s_pre4": ["-DDM16S_PRE=4"],
    "s_nocell_noldsa": ["-DDM16S_ABL_NOCELL", "-DDM16S_ABL_NOLDSA"],
    # (round 3's timing-only "int8 cross terms" variants s_i8t / s_i8t_nox became the real DM_PREC_F16I8: git show 8a84c8d:deepmod_amd/csrc/lstm_f16s.hip.inc)
    # round 4: the operand-toggle dial (tools/lo_trunc_dial.py): packer knob from the environment + a_lo mask of m bits
    "alo0": ["-DDM_WLO_TRUNC_ENV"], "alo3": ["-DDM_WLO_TRUNC_ENV", "-DDM16S_ALO_TRUNC=3"], "alo5": ["-DDM_WLO_TRUNC_ENV", "-DDM16S_ALO_TRUNC=5"],
    "alo6": ["-DDM_WLO_TRUNC_ENV", "-DDM16S_ALO_TRUNC=6"],
    "q_nop0": ["-DDM16Q_NOPN=0"], "q_nop1": ["-DDM16Q_NOPN=1"], "q_nop3": ["-DDM16Q_NOPN=3"], "q_nop7": ["-DDM16Q_NOPN=7"],
    "q_snake": ["-DDM16Q_SNAKE"], "q_pre0": ["-DDM16Q_PRE=0"], "q_pre6": ["-DDM16Q_PRE=6"], "q_snake_pre6": ["-DDM16Q_SNAKE", "-DDM16Q_PRE=6"],
    "q_novmwait": ["-DDM16Q_ABL_NOVMWAIT"], "q_novmwait_nobar": ["-DDM16Q_ABL_NOVMWAIT", "-DDM16Q_ABL_NOBAR"],      # is the DMA's latency exposed (the wait before the barrier)?
    "q_nobar": ["-DDM16Q_ABL_NOBAR"], "q_nodma": ["-DDM16Q_ABL_NODMA"], "q_nocell": ["-DDM16Q_ABL_NOCELL"], "q_nodma_nobar": ["-DDM16Q_ABL_NODMA", "-DDM16Q_ABL_NOBAR"],      # round 5: where the merged-mixed kernel's time goes (timing only)
    "q_i8t": ["-DDM16Q_ABL_I8T"],        # round 5: timing only - both cross terms of a k32-step as one int8 16x16x64 MFMA (the price of an int8 mode on this shape, before its fold / pack instructions)
    "q_lopack_mix": ["-DDM16Q_LOPACK=1"],   # round 6: the lo halves packed by v_fma_mixlo / mixhi_f16 (2 instructions fewer per super-tile, bit-identical)
    "q_lopack_none": ["-DDM16Q_LOPACK=2"],  # round 6: timing only - no lo pack at all (what those two conversions cost)
    "q_mix1": ["-DDM16Q_ABL_MIX1"],      # round 5: the mixed k32-step issued as one product (timing only)
    "q_base": [], "q_pre1": ["-DDM16Q_PRE=1"], "q_pre3": ["-DDM16Q_PRE=3"], "q_ainit": ["-DDM16Q_AINIT"], "q_trans3": ["-DDM16Q_TRANS_COST=3"], "q_trans1": ["-DDM16Q_TRANS_COST=1"],
    "q_nochunk": ["-DDM16Q_NOCHUNK"], "q_nop": ["-DDM16Q_DEBUG_NOP"], "q_nop_nochunk": ["-DDM16Q_DEBUG_NOP", "-DDM16Q_NOCHUNK"],
    "s_mfma16": ["-DDM16S_ABL_MFMA16"], "s_mfma16_pad": ["-DDM16S_ABL_MFMA16", "-DDM16S_ABL_MFMA16_PAD"],                         # round 4: timing-only, every 32x32x16 MFMA as two 16x16x32 (same MACs, half the accumulator registers per FLOP)
    "roles_dma_m": ["-DDM_WITH_F16X3_ROLES", "-DDM16R_DMA_M"],
    "roles": ["-DDM_WITH_F16X3_ROLES"],                        # round 4: the wave-pair experiment kernel (tools/experiments/f16r, tools/roles_ab.py)
    "trace": ["-DDM_TRACE"],                                   # f16x3 kernel: per-wave timeline of one stage
    "w4": ["-DDM16_WAVES=4", "-DDM16_MT=2"],                   # f16x3 kernel: 4 waves x 2 M-tiles (one wave per SIMD)
    "w4timing": ["-DDM16_WAVES=4", "-DDM16_MT=2", "-DDM_TIMING"],
    "w4trace": ["-DDM16_WAVES=4", "-DDM16_MT=2", "-DDM_TRACE"],
    "trace_noldsb": ["-DDM_TRACE", "-DDM16_ABL_NOLDSB"],
    "trace_nodma": ["-DDM_TRACE", "-DDM16_ABL_NODMA"],
    "mfma_only": ["-DDM_ABL_NOEPI", "-DDM_ABL_NOBAR", "-DDM_ABL_NODMA", "-DDM_ABL_NOSEQ"],
}
EXTRA = sys.argv[3:] if len(sys.argv) > 3 else []


def build(names):
    os.makedirs(ABL, exist_ok=True)
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    for name in names:
        out = os.path.join(ABL, "lib_%s.so" % name)
        # any switch needs the umbrella (csrc/kernels.h: a stray -D is a compile error); one object directory per variant
        flags = VARIANTS[name] + (["-DDM_EXPERIMENT"] if VARIANTS[name] else [])
        # the 32x32x16 kernels (DM16S_* switches, the roles experiment) are a translation unit of experiment builds only since round 6
        if any(f.startswith(("-DDM16S_", "-DDM_WITH_F16X3_ROLES", "-DDM16R_", "-DDM_WLO_TRUNC")) for f in flags) and "-DDM_WITH_F16S" not in flags:
            flags = flags + ["-DDM_WITH_F16S"]
        ge.build_library(out, flags, objdir=os.path.join(ABL, "obj_" + name), quiet=True)
        print("built", out)


PREC = int(os.environ.get('DM_ABL_PREC', '0'))


def run(names, n=65536, reps=int(os.environ.get('DM_ABL_REPS', '6'))):
    sys.path.insert(0, ROOT)
    import numpy as np
    from deepmod_amd import _lib, model, synth
    res = {}
    x = synth.synthetic_windows(n, seed=1)
    w = synth.synthetic_weights(7, 1.0)
    for name in names:
        path = os.path.join(ABL, "lib_%s.so" % name)
        if not os.path.exists(path):
            continue
        _lib._LIB = None
        _lib.LIB_PATH = path
        m = model.BiLSTMModel(w, 0)
        m.set_option(_lib.DM_OPT_PROFILE, 1)
        m.set_option(_lib.DM_OPT_PRECISION, PREC)
        dx = model.DeviceArray.from_host(x, 0)
        dc = model.DeviceArray((n,), np.uint8, 0)
        m.predict_windows(dx, cls=dc, want_prob=False)
        m.profile_reset()
        for _ in range(reps):
            m.predict_windows(dx, cls=dc, want_prob=False)
        ms, launches, _ = m.profile_get()
        res[name] = ms / launches
        print("%-14s %.3f ms/launch  %.3g windows/s  %.1f%% of fp32 MFMA peak" %
              (name, res[name], n / res[name] * 1e3, n * 8.924e6 / (res[name] * 1e-3) / 157.3e12 * 100), flush=True)
        if name.startswith("tiles"):
            import ctypes
            lib = _lib.load()
            lib.dm_debug_timing.restype = ctypes.c_longlong
            lib.dm_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
            cnt = lib.dm_debug_timing(m._h, None, 0)
            buf = np.zeros(cnt, np.uint64)
            lib.dm_debug_timing(m._h, buf.ctypes.data, cnt)
            t = buf[:256].reshape(8, 32).astype(np.int64)[:, :27]
            t0 = t[:, 0].min()
            print("   per-tile start times of one MFMA block (cycles after the first wave entered it); last two rows: block end, barrier exit")
            print("   %-8s" % "tile" + "".join("  wave%d" % w for w in range(8)))
            for i in range(27):
                print("   %-8s" % (i if i < 25 else ("end", "barrier")[i - 25]) + "".join("%7d" % (t[w, i] - t0) for w in range(8)))
        if "trace" in name:
            import ctypes
            lib = _lib.load()
            lib.dm_debug_timing.restype = ctypes.c_longlong
            lib.dm_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
            cnt = lib.dm_debug_timing(m._h, None, 0)
            buf = np.zeros(cnt, np.uint64)
            lib.dm_debug_timing(m._h, buf.ctypes.data, cnt)
            t = buf[:256].reshape(8, 32).astype(np.int64)
            t = t[t[:, 0] > 0]
            t0 = t[:, 0].min()
            labels = ["stage start", "A loaded"] + sum([["k%d mfma end" % i, "k%d dma done" % i, "k%d barrier exit" % i] for i in range(7)], []) + ["epilogue end"]
            print("   %-18s" % "event" + "".join("  wave%d" % w for w in range(len(t))))
            for i, lb in enumerate(labels):
                print("   %-18s" % lb + "".join("%7d" % (t[w, i] - t0) for w in range(len(t))))
        if "timing" in name:
            import ctypes
            lib = _lib.load()
            lib.dm_debug_timing.restype = ctypes.c_longlong
            lib.dm_debug_timing.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong]
            cnt = lib.dm_debug_timing(m._h, None, 0)
            buf = np.zeros(cnt, np.uint64)
            lib.dm_debug_timing(m._h, buf.ctypes.data, cnt)
            t = buf.reshape(-1, 8).astype(np.float64)
            t = t[t[:, 6] > 0]
            tot = t[:, 6].mean()
            names = ["chunk prologue (A operand, DMA issue)", "chunk MFMA block", "dma wait (vmcnt 0)", "barrier",
                     "pass prologue", "step epilogue (cell update, bias re-init, publish)", "kernel total"]
            if PREC == 1:
                names[0] = "stage A-operand load (LDS slab / global / x)"
            for i, nm in enumerate(names):
                print("   %-52s %12.0f cycles  %5.1f%%" % (nm, t[:, i].mean(), 100 * t[:, i].mean() / tot))
            print("   %-52s %12.0f cycles  %5.1f%%  (inside the epilogue)" % ("publish slab / head reduce", t[:, 7].mean(), 100 * t[:, 7].mean() / tot))
            print("   unaccounted %.1f%%" % (100 * (tot - t[:, :6].sum(axis=1).mean()) / tot))
        m.close(); dx.free(); dc.free()
    return res


if __name__ == "__main__":
    names = sys.argv[2].split(",") if len(sys.argv) > 2 and sys.argv[2] != "all" else list(VARIANTS)
    if sys.argv[1] == "build":
        build(names)
    elif sys.argv[1] == "run1":
        run(names)
    else:  # one process per variant: same-named kernel symbols would interpose across dlopen()ed variants
        for nm in names:
            subprocess.call([sys.executable, os.path.abspath(__file__), "run1", nm])
