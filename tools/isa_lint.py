"""Build-time lint of the hand-scheduled kernels in libdeepmod_hip.so (no GPU needed).

The product kernels lean on inline asm (LDS reads with counted s_waitcnt, v_fma_mix_f32, empty-asm pins) that hipcc's hazard
recogniser cannot see into, and on a register budget with no head-room.  This module extracts the gfx950 code object from the
shared library (llvm-objdump --offloading), reads the kernels' resource metadata (llvm-readelf --notes) and their disassembly, and
reports per kernel: scratch bytes, spill counts, register counts, MFMA count by opcode, and the smallest number of wait states
between an MFMA and a later non-MFMA vector instruction that reads or overwrites its destination registers (the hazard the 4-pass
MFMA exposed in round 4: tools/q_debug.py).  Wait states are counted the way LLVM's GCNHazardRecognizer counts them: one per
instruction, N + 1 for `s_nop N`.  tests/test_build_guard.py asserts on the report;  `python tools/isa_lint.py` prints it.
"""
from __future__ import annotations

import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
LIB = os.path.join(ROOT, "deepmod_amd", "csrc", "libdeepmod_hip.so")

PRODUCT_KERNELS = {
    "f16q": "_ZN7lstm16q18bilstm_f16q_kernelILi0EEEvN5lstmc6ParamsE",
    "f16qi8": "_ZN7lstm16q18bilstm_f16q_kernelILi1EEEvN5lstmc6ParamsE",
    "f32": "_ZN6lstm3217bilstm_f32_kernelENS_6ParamsE",
}
# experiment builds only (DM_WITH_F16S=1, tools/experiments/f16s): the 32x32x16 kernels of rounds 2-3 - product kernels until round 5
EXPERIMENT_KERNELS = {
    "f16s": "_ZN7lstm16s18bilstm_f16s_kernelILi0EEEvN5lstmc6ParamsE",
    "f16i8": "_ZN7lstm16s18bilstm_f16s_kernelILi1EEEvN5lstmc6ParamsE",
}
# (passes of 4 cycles, XDL?) of the MFMAs the product issues.  Wait states before a VALU instruction may read or overwrite the result, as LLVM's
# GCNHazardRecognizer enforces them for gfx950 (and as hipcc's own output shows: it pads to exactly these): XDL (the 16-bit / 8-bit
# dot-product forms) passes + 3, + 1 on gfx950; the fp32 forms (not XDL) passes + 2
MFMA_KIND = {"v_mfma_f32_16x16x32_f16": (4, True), "v_mfma_f32_32x32x16_f16": (8, True), "v_mfma_i32_32x32x32_i8": (8, True),
             "v_mfma_i32_16x16x64_i8": (4, True), "v_mfma_f32_16x16x4_f32": (8, False)}


def required_wait_states(opcode: str) -> int:
    passes, xdl = MFMA_KIND.get(opcode, (16, True))
    return passes + (4 if xdl else 2)


def extract_code_objects(lib: str, workdir: str) -> list:
    """-> paths of the gfx950 code objects embedded in `lib`, one per translation unit (llvm-objdump writes the bundles next to its input:
    work on a copy)."""
    local = os.path.join(workdir, "lib.so")
    shutil.copy(lib, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], check=True, capture_output=True)
    out = [os.path.join(workdir, f) for f in sorted(os.listdir(workdir)) if "amdgcn" in f and "gfx950" in f and os.path.getsize(os.path.join(workdir, f)) > 0]
    if not out:
        raise RuntimeError("no gfx950 code object in %s" % lib)
    return out


def kernel_metadata(code_object: str) -> dict:
    """{kernel symbol: {key: int}} from the amdhsa.kernels note (the keys the guard needs are plain 'name: int' lines)."""
    txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", code_object], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    # entries start with "  - .agpr_count:" (the first key, alphabetically) and carry ".name:" somewhere inside: collect then key by name
    block = {}
    for line in txt.splitlines():
        m = re.match(r"\s+(- )?\.(\w+):\s+(\S+)\s*$", line)
        if not m:
            continue
        if m.group(1) and m.group(2) == "agpr_count":
            if block.get("name"):
                out[block["name"]] = block
            block = {}
        key, val = m.group(2), m.group(3)
        if key in ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
                   "group_segment_fixed_size", "max_flat_workgroup_size"):
            block[key] = int(val)
        elif key == "name" and "name" not in block:
            block["name"] = val
    if block.get("name"):
        out[block["name"]] = block
    return out


def disassemble(code_object: str) -> dict:
    """{symbol: [instruction text, ...]}"""
    txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", code_object], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is None or not line.startswith("\t"):
            continue
        ins = line.split("//")[0].strip()
        if ins:
            cur.append(ins)
    return out


_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def _regs(operand_text: str) -> set:
    s = set()
    for m in _REG.finditer(operand_text):
        if m.group(1):
            s.add((m.group(1), int(m.group(2))))
        else:
            s.update((m.group(3), r) for r in range(int(m.group(4)), int(m.group(5)) + 1))
    return s


def _wait_states(ins: str) -> int:
    m = re.match(r"s_nop\s+(\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def lint_kernel(instrs: list) -> dict:
    """MFMA census and the tightest MFMA -> vector-ALU use of its destination."""
    census = {}
    tightest = None          # (wait states, mfma index, mfma text, user text)
    per_opcode_min = {}
    for i, ins in enumerate(instrs):
        op = ins.split()[0]
        if not op.startswith(("v_mfma", "v_smfmac")):
            continue
        base = re.sub(r"_e64$", "", op)
        census[base] = census.get(base, 0) + 1
        dst = _regs(ins[len(op):].split(",")[0])
        passes = MFMA_KIND.get(base, (16, True))[0]
        ws = 0
        for j in range(i + 1, min(i + 1 + 4 * passes + 16, len(instrs))):
            nxt = instrs[j]
            nop = nxt.split()[0]
            if nop.startswith("v_") and not nop.startswith(("v_mfma", "v_smfmac")) and (_regs(nxt[len(nop):]) & dst):
                if base not in per_opcode_min or ws < per_opcode_min[base][0]:
                    per_opcode_min[base] = (ws, i, ins, nxt)
                break
            if nop.startswith(("s_cbranch", "s_branch", "s_endpgm")):
                break
            ws += _wait_states(nxt)
            if ws > passes + 8:
                break
    return {"mfma": census, "mfma_total": sum(census.values()), "instructions": len(instrs),
            "tightest_use": {k: {"wait_states": v[0], "required": required_wait_states(k), "mfma": v[2], "user": v[3]} for k, v in per_opcode_min.items()}}


def report(lib: str = LIB) -> dict:
    tmp = tempfile.mkdtemp(prefix="dm_isa_lint_")
    try:
        meta, dis = {}, {}
        for co in extract_code_objects(lib, tmp):
            meta.update(kernel_metadata(co))
            dis.update(disassemble(co))
        out = {}
        for short, sym in PRODUCT_KERNELS.items():
            if sym not in dis:
                out[short] = {"missing": True}
                continue
            r = lint_kernel(dis[sym])
            r["resources"] = meta.get(sym, {})
            out[short] = r
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    import json
    print(json.dumps(report(sys.argv[1] if len(sys.argv) > 1 else LIB), indent=1))
