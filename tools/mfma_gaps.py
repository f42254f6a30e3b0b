"""Dev tool: a static issue model of the default kernel's instruction stream (python tools/mfma_gaps.py; LIB=<.so> for another build).

Every instruction between two consecutive MFMAs of bilstm_f16q_kernel<0> is priced at its nominal issue cost (plain VALU / DS / VMEM 4 cycles,
transcendental 8, SALU 1, s_nop n + 1); a v_mfma_f32_16x16x32_f16 occupies the matrix pipe for 16 cycles of which 12 can hide other issues of
the same wave (one wave per SIMD).  `lost` = sum over gaps of max(0, cost - 12): what an in-order wave cannot hide if nothing else stalls.
Round 5 used it to ask whether the schedule leaves the matrix pipe idle (profiles/r05/mfma_gaps.txt): it does - the model says 77 % busy for
the generic step, the launch measures 62 % - and then measured that levelling the gaps does not shorten the launch (ablate_la_spread.txt):
the kernel runs at the power limit, a shorter schedule is paid back in clock."""
import sys, collections, re
sys.path.insert(0, '/root/repo/tools')
import isa_lint, tempfile
lib = __import__('os').environ.get('LIB', '/root/repo/deepmod_amd/csrc/libdeepmod_hip.so')
wd = tempfile.mkdtemp()
instrs = None
for co in isa_lint.extract_code_objects(lib, wd):
    d = isa_lint.disassemble(co)
    for k, v in d.items():
        if 'bilstm_f16q_kernelILi0' in k:
            instrs = v
print(len(instrs))
# cost model in cycles of issue: MFMA 16x16x32 = 16 pipe cycles (4 issue); VALU 4; trans 8 (5/3 rule ~7); ds/vmem 4; salu 1 (interleaves) ; s_nop n+1; s_waitcnt/barrier flagged
def cost(ins):
    op = ins.split()[0]
    if op.startswith('v_mfma'): return 'M'
    if op.startswith(('v_exp', 'v_rcp', 'v_log', 'v_sqrt', 'v_rsq')): return 8
    if op.startswith('v_'): return 4
    if op.startswith(('ds_', 'global_', 'buffer_', 'flat_')): return 4
    if op.startswith('s_nop'):
        return 1 + int(ins.split()[1])
    if op.startswith('s_'): return 1
    return 1
# walk: for each gap between consecutive MFMAs sum the non-MFMA cost; cycles lost ~ max(0, gapcost - 12)
gaps = []
cur = 0; special = []
tot_nonm = 0; nm = 0
for ins in instrs:
    c = cost(ins)
    if c == 'M':
        gaps.append((cur, tuple(special))); cur = 0; special = []; nm += 1
    else:
        cur += c; tot_nonm += c
        op = ins.split()[0]
        if op in ('s_barrier', 's_waitcnt') or op.startswith('global_load_lds'):
            special.append(op if op != 's_waitcnt' else ins)
print('mfma', nm, 'nonmfma issue cycles', tot_nonm, 'per mfma', tot_nonm / nm)
hist = collections.Counter()
lost = 0; lost_big = 0
for g, sp in gaps:
    b = min(g // 4, 40)
    hist[b] += 1
    l = max(0, g - 12)
    lost += l
    if g > 60: lost_big += l
for b in sorted(hist): print('%4d..%-4d cycles: %5d gaps' % (b * 4, b * 4 + 3, hist[b]))
print('model: mfma cycles', nm * 16, 'lost', lost, '(in gaps > 60:', lost_big, ') ratio busy', nm * 16 / (nm * 16 + lost))
big = [(g, sp) for g, sp in gaps if g > 60]
print(len(big), 'big gaps; sample:')
for g, sp in big[:12]: print(g, [s.split()[0] for s in sp][:12])
print('---- generic step only (after the first 1100 MFMAs)')
gg = gaps[1101:]
nmg = len(gg)
lost = sum(max(0, g - 12) for g, _ in gg)
tot = sum(g for g, _ in gg)
print('mfma', nmg, 'nonmfma cycles', tot, 'per mfma', tot / nmg, 'lost(model)', lost, 'busy', nmg * 16 / (nmg * 16 + lost))
hist = collections.Counter()
for g, sp in gg: hist[min(g // 8, 30)] += 1
for b in sorted(hist): print('%4d..%-4d: %5d gaps, lost %6d' % (b * 8, b * 8 + 7, hist[b], sum(max(0, g - 12) for g, _ in gg if min(g // 8, 30) == b)))
big = [(i, g, sp) for i, (g, sp) in enumerate(gg) if g > 40]
print(len(big), 'gaps > 40:')
for i, g, sp in big[:45]: print(i, g, [s.split()[0] + (' ' + s.split()[1] if s.startswith('s_waitcnt') else '') for s in sp][:10])
print('==== instruction dump of selected gaps')
# rebuild with instruction lists
glist = []; cur = []
for ins in instrs:
    if cost(ins) == 'M':
        glist.append(cur); cur = []
    else: cur.append(ins)
gl = glist[1101:]
import os
sel = [int(x) for x in os.environ.get('SEL', '105,117').split(',')]
for i in sel:
    print('--- gap before generic MFMA', i, 'cost', sum(cost(x) for x in gl[i]))
    for x in gl[i]: print('   ', x[:110])
print('==== per tile (between barriers), generic step')
tiles = []; cur = []
for g, sp in gg:
    if any(s.startswith('s_barrier') for s in sp) and cur:
        tiles.append(cur); cur = []
    cur.append((g, sp))
tiles.append(cur)
for ti, t in enumerate(tiles):
    n = len(t); fill = sum(g for g, _ in t); lost = sum(max(0, g - 12) for g, _ in t)
    big = sorted([g for g, _ in t], reverse=True)[:6]
    print('tile %2d: %3d mfma, filler %5d (%.1f/mfma), lost %4d (%.0f%% of mfma time), largest gaps %s' % (ti, n, fill, fill / n, lost, 100 * lost / (16 * n), big))
