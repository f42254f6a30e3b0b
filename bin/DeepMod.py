#!/usr/bin/env python
"""DeepMod (MI355X-native hot path) command line.

Keeps the `detect` sub-command of the reference's bin/DeepMod.py (flag names and defaults of
bin/DeepMod.py:304-338) for the path this build implements: per-read BiLSTM modification calling on
the GPU and the per-position BED summary.  `--wrkBase` holds feature containers (*.dmfeat.npz, see
deepmod_amd/predstore.py) because FAST5 reading and alignment are out of scope here.  `train` and
`getfeatures` are training-side and not built.
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def non_negative(value, name):
    if value < 0:
        raise SystemExit('Error: --%s must be non-negative (got %d)' % (name, value))


def build_parser():
    parser = argparse.ArgumentParser(prog='DeepMod.py', description='Detect DNA modifications from nanopore reads '
                                     '(per-read BiLSTM calling + per-position summary) on AMD MI355X.')
    sub = parser.add_subparsers(dest='cmd')
    com = argparse.ArgumentParser(add_help=False)
    com.add_argument('--outLevel', type=int, choices=[0, 1, 2, 3], default=2, help='0 debug, 1 info, 2 warning, 3 error')
    com.add_argument('--wrkBase', help='working base folder with the input files')
    com.add_argument('--FileID', default='mod', help='unique id of this run')
    com.add_argument('--outFolder', default='./mod_output', help='default output folder')
    com.add_argument('--recursive', type=int, choices=[0, 1], default=1, help='search sub-folders (up to 3 levels)')
    com.add_argument('--threads', type=int, default=4, help='number of worker processes')
    com.add_argument('--files_per_thread', type=int, default=1000, help='input files per worker batch')
    com.add_argument('--windowsize', type=int, default=21, help='window size (odd)')
    com.add_argument('--alignStr', choices=['bwa', 'minimap2'], default='minimap2', help='aligner run on raw containers when on PATH; otherwise side-car <container>.sam files are read')
    com.add_argument('--SignalGroup', choices=['simple', 'rundif'], default='simple', help='accepted for compatibility')
    com.add_argument('--move', action='store_true', default=False, help='accepted for compatibility')
    det = sub.add_parser('detect', parents=[com], help='detect modifications')
    det.add_argument('--Ref', help='reference genome FASTA (raw containers: reference bases of the aligned reads)')
    det.add_argument('--predDet', type=int, choices=[0, 1], default=1, help='1: predict + summarise; 0: summarise only')
    det.add_argument('--predpath', default=None, help='prediction folder for --predDet 0')
    det.add_argument('--modfile', default=None, help='checkpoint prefix of the trained model (TF bundle)')
    det.add_argument('--fnum', type=int, default=7, help='features per event')
    det.add_argument('--hidden', type=int, default=100, help='LSTM hidden units')
    det.add_argument('--basecall_1d', default='Basecall_1D_000', help='accepted for compatibility')
    det.add_argument('--basecall_2strand', default='BaseCalled_template', help='accepted for compatibility')
    det.add_argument('--region', default=None, help='regions of interest, e.g. chr1:1:100000;chr2:10000')
    det.add_argument('--ConUnk', default=True, help='also process contigs whose names contain _ - / :')
    det.add_argument('--outputlayer', default='', choices=['', 'sigmoid'], help="only '' is built")
    det.add_argument('--Base', default='C', choices=['A', 'C', 'G', 'T'], help='base of interest')
    det.add_argument('--mod_cluster', default=0, type=int, choices=[0, 1], help='only 0 is built')
    det.add_argument('--gpus', type=int, default=None, help='GPUs to spread the workers over (default: all visible)')
    det.add_argument('--storePred', type=int, choices=[0, 1], default=0,
                     help='0 (default): streaming detect - per-position counters stay on the GPUs, one RCCL reduce per contig x strand, '
                          'no per-read files; 1: also keep the per-read prediction tables and index files of the reference '
                          '(needed for a later --predDet 0 run)')
    det.set_defaults(func=mDetect)
    for name in ('train', 'getfeatures'):
        p = sub.add_parser(name, help='not built: training-side, outside the accelerated path')
        p.set_defaults(func=lambda a, _n=name: sys.exit("'%s' is not part of this build (inference hot path only)" % _n))
    return parser


def kfd_gpu_count(base='/sys/class/kfd/kfd/topology/nodes'):
    """gfx950 devices this process may open, from /sys/class/kfd (no HIP runtime): topology nodes whose properties are readable (a container's
    device cgroup hides the others) and say gfx_target_version 90500, capped by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES.  None if the
    topology cannot be read."""
    try:
        nodes = os.listdir(base)
    except OSError:
        return None
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(base, node, 'properties')) as fh:
                props = dict(line.split()[:2] for line in fh if len(line.split()) >= 2)
        except OSError:
            continue
        if props.get('gfx_target_version') == '90500' and int(props.get('simd_count', '0')) > 0:
            n += 1
    # A *_VISIBLE_DEVICES filter changes what the runtime shows in ways the topology files do not tell (ROCR and HIP filters compose, an
    # invalid or -1 entry truncates the list): with any of them set the runtime is asked instead
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES', 'GPU_DEVICE_ORDINAL'):
        if os.environ.get(var, '').strip() != '':
            return None
    return n


def mDetect(args):
    from deepmod_amd import _lib, detect
    mo = {k: getattr(args, k) for k in ('outLevel', 'wrkBase', 'FileID', 'outFolder', 'recursive', 'threads', 'files_per_thread',
                                         'windowsize', 'predDet', 'predpath', 'modfile', 'fnum', 'hidden', 'outputlayer', 'Base',
                                         'mod_cluster', 'Ref', 'alignStr', 'SignalGroup', 'basecall_1d', 'basecall_2strand', 'storePred')}
    for k in ('threads', 'files_per_thread', 'windowsize', 'fnum', 'hidden'):
        non_negative(mo[k], k)
    if mo['threads'] < 1:
        mo['threads'] = 1                                # bin/DeepMod.py:78: at least one worker
    if mo['files_per_thread'] < 2:
        mo['files_per_thread'] = 2                       # bin/DeepMod.py:75-76
    if mo['windowsize'] % 2 == 0:
        raise SystemExit('Error: --windowsize must be odd')
    if not mo['outFolder'].endswith('/'):
        mo['outFolder'] += '/'
    mo['region'] = []                                    # bin/DeepMod.py:153-160: "chr:start:end;chr2:start"
    if args.region is None or len(args.region) == 0:
        mo['region'].append([None, None, None])
    else:
        for mr in args.region.split(';'):
            mr_sp = mr.split(':')
            mo['region'].append([mr_sp[0], int(mr_sp[1]) if len(mr_sp) > 1 else None, int(mr_sp[2]) if len(mr_sp) > 2 else None])
    mo['ConUnk'] = args.ConUnk not in (False, 'False', 'false', '0', 0)
    errs = []
    if mo['predDet'] == 1:
        if not mo['wrkBase'] or not os.path.isdir(mo['wrkBase']):
            errs.append('--wrkBase: input folder does not exist')
        if not mo['modfile'] or not os.path.isfile(mo['modfile'] + '.index'):
            errs.append('--modfile: no TF checkpoint at %r' % mo['modfile'])
    else:
        if not mo['predpath'] or not os.path.isdir(mo['predpath']):
            errs.append('--predpath: prediction folder does not exist')
        else:
            mo['outFolder'] = mo['predpath'].rstrip('/')
    if errs:
        raise SystemExit('Error:\n\t' + '\n\t'.join(errs))
    # how many GPUs: with --gpus N the answer is needed only to refuse N > what is there, and the kernel driver's topology files give it without
    # initialising a HIP runtime in THIS process, which never touches a GPU itself (0.14 s of a 1.8 s run; the rank processes initialise
    # their own).  Without --gpus, or when the files say less than N, the runtime is asked.
    ngpu = kfd_gpu_count() if args.gpus else None
    if ngpu is None or ngpu < args.gpus:
        ngpu = _lib.load().dm_device_count()
    if ngpu < 1:
        raise SystemExit('Error: no gfx950 GPU visible (this build has no CPU path)')
    mo['gpus'] = min(args.gpus, ngpu) if args.gpus else ngpu
    if os.environ.get('DEEPMOD_ONE_DEVICE') == '1' and args.gpus:      # test hook: --gpus N processes, all on device 0 (a one-GPU box; RCCL
        mo['gpus'], mo['one_device'] = args.gpus, True                  # refuses duplicate devices: this exercises start-up, rendezvous and the abort path)
    detect.mDetect_manager(mo)


if __name__ == '__main__':
    parser = build_parser()
    if len(sys.argv) < 2:
        parser.print_help()
        sys.exit(1)
    args = parser.parse_args()
    if not hasattr(args, 'func'):
        parser.print_help()
        sys.exit(1)
    args.func(args)
