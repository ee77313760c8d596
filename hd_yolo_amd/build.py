"""Build libhdyolo_hip.so (gfx950) in-tree with hipcc; no GPU needed (cross-compiles).

    python -m hd_yolo_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT = os.path.join(CSRC, 'build')
LIB = os.path.join(OUT, 'libhdyolo_hip.so')
SOURCES = ['api.hip', 'conv_igemm.hip', 'conv3x3.hip', 'conv3x3_c128.hip', 'conv3x3s2.hip', 'conv_deep.hip', 'conv_stem.hip', 'conv_wgrad.hip', 'conv_wgrad3x3.hip', 'conv_wgrad_deep.hip', 'conv_dgrad_s2.hip', 'conv1x1_bwd.hip', 'bn_act.hip', 'pool.hip', 'detect.hip', 'loss.hip', 'roi.hip', 'seg.hip', 'optim.hip', 'exec.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-ffp-contract=off', '-Wall', '-Wno-unused-function',
         '-I' + CSRC, '-I' + os.path.join(os.path.dirname(HERE), 'include')]


def _newer(a, bs):
    return os.path.exists(a) and all(os.path.getmtime(a) >= os.path.getmtime(b) for b in bs)


def build(force=False, verbose=True):
    os.makedirs(OUT, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith('.h')]
    headers.append(os.path.join(os.path.dirname(HERE), 'include', 'hdyolo.h'))
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(OUT, src.replace('.hip', '.o'))
        if not force and _newer(o, [s] + headers):
            return o
        cmd = [hipcc] + FLAGS + ['-c', s, '-o', o]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        return o

    with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or not _newer(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
