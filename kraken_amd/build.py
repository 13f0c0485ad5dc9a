"""
In-tree build of the gfx950 shared library ``kraken_amd/libkraken_amd.so``.

hipcc cross-compiles without a GPU, so this runs in the authoring container, on
the GPU box and from ``__graft_entry__.build()``.  Objects are cached under
``kraken_amd/csrc/_build`` keyed on source mtimes.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
BUILD = os.path.join(CSRC, '_build')
LIB = os.path.join(HERE, 'libkraken_amd.so')
SOURCES = ['conv_mfma.hip', 'conv_x3.hip', 'conv_x3p.hip', 'conv_x6.hip', 'conv1_x3.hip', 'conv_taps_x3.hip', 'gemm_x3.hip', 'norm_x3.hip', 'lstm_rec.hip', 'lstm_small.hip', 'lstm_x3.hip', 'lstm_ws.hip', 'misc_kernels.hip', 'c1gn.hip', 'prep_lines.hip', 'dewarp.hip', 'capi.hip']
# sources compiled a second time with -DKRK_BF16_ONE: the plain-bf16 plan's launchers (name_b1), see csrc/common.h
ONE_TERM = ['conv1_x3.hip', 'conv_taps_x3.hip', 'conv_x3.hip', 'conv_x3p.hip', 'gemm_x3.hip', 'lstm_ws.hip']
# further compiles of one source under a define (kernel experiments selected by a probe switch at run time): {source: [(suffix, -D...)]}
VARIANTS = {}
HEADERS = [os.path.join(CSRC, 'common.h'),
           os.path.join(os.path.dirname(HERE), 'include', 'kraken_amd.h')]
ARCH = 'gfx950'
FLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-function',
         '-fno-gpu-rdc', '-DNDEBUG']


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found; the HIP extension cannot be built')
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, ablate: bool = False, asan: bool = False) -> str:
    """
    Compiles every HIP source for gfx950 and links the C-ABI library. Returns its path.

    ``ablate=True`` builds ``libkraken_amd_ablate.so`` instead: the same kernels with their phase-skipping probe
    branches (``KRK_DBGBIT``, common.h) compiled in -- a measuring tool (tools/lstm_probe.py), selected with
    ``KRAKEN_AMD_LIB``; the release library has no probe code in its hot loops.
    """
    if asan:
        # HOST AddressSanitizer of the C++ side (plan compiler, weight packing, length arithmetic, the C ABI's argument handling): the
        # device code is compiled as always (-fno-gpu-sanitize: GPU ASan needs xnack+ code objects, which this pool does not run).
        # Run through tools/asan_run.sh, which preloads clang's ASan runtime into the (uninstrumented) python.
        bdir, lib_path = BUILD + '_asan', LIB.replace('.so', '_asan.so')
        flags = [f for f in FLAGS if f != '-O3'] + ['-O1', '-g', '-fsanitize=address', '-fno-gpu-sanitize', '-fno-omit-frame-pointer']
    else:
        bdir = BUILD + ('_ablate' if ablate else '')
        lib_path = LIB.replace('.so', '_ablate.so') if ablate else LIB
        flags = FLAGS + (['-DKRK_ABLATE'] if ablate else [])
    os.makedirs(bdir, exist_ok=True)
    hipcc = _hipcc()
    objs, jobs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(bdir, src.replace('.hip', '.o'))
        objs.append(obj)
        if force or _stale(obj, [sp] + HEADERS + [os.path.abspath(__file__)]):
            jobs.append([hipcc, *flags, '-c', sp, '-o', obj])
        for suffix, define in VARIANTS.get(src, ()):
            objv = os.path.join(bdir, src.replace('.hip', f'_{suffix}.o'))
            objs.append(objv)
            if force or _stale(objv, [sp] + HEADERS + [os.path.abspath(__file__)]):
                jobs.append([hipcc, *flags, define, '-c', sp, '-o', objv])
        if src in ONE_TERM:
            obj1 = os.path.join(bdir, src.replace('.hip', '_b1.o'))
            objs.append(obj1)
            if force or _stale(obj1, [sp] + HEADERS + [os.path.abspath(__file__)]):
                jobs.append([hipcc, *flags, '-DKRK_BF16_ONE', '-c', sp, '-o', obj1])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f'hipcc failed: {" ".join(cmd)}\n{r.stdout}\n{r.stderr}')
        return r

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(lib_path, objs):
        run([hipcc, '-shared', '-fPIC', f'--offload-arch={ARCH}', *(['-fsanitize=address', '-fno-gpu-sanitize', '-shared-libsan'] if asan else []),
             *objs, '-o', lib_path])
    return lib_path


def build_asan_host(verbose: bool = False) -> str:
    """
    ``kraken_amd/libkraken_amd_asanhost.so``: the library's HOST code under AddressSanitizer, linked against a host stand-in for the
    HIP runtime (tools/asan/fake_hip.cpp: device memory = malloc, launches = no-ops) instead of libamdhip64 -- runs WITHOUT a GPU
    (tests/test_asan_host.py, tools/asan_host_driver.py).  capi.hip (the plan compiler, the packers, the C ABI) is instrumented; the
    other sources' launch stubs come from the regular objects.  Nothing is computed: a checker of host memory safety only.
    """
    build()
    bdir = BUILD + '_asan'
    os.makedirs(bdir, exist_ok=True)
    hipcc = _hipcc()
    clangxx = os.path.join(os.path.dirname(os.path.realpath(hipcc)), '..', 'lib', 'llvm', 'bin', 'clang++')
    if not os.path.exists(clangxx):
        clangxx = '/opt/rocm/lib/llvm/bin/clang++'
    san = ['-O1', '-g', '-fsanitize=address', '-fno-omit-frame-pointer']
    lib_path = LIB.replace('.so', '_asanhost.so')
    capi, fake = os.path.join(bdir, 'capi.o'), os.path.join(bdir, 'fake_hip.o')
    fake_src = os.path.join(os.path.dirname(HERE), 'tools', 'asan', 'fake_hip.cpp')

    def run(cmd):
        if verbose:
            print(' '.join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError(f'failed: {" ".join(cmd)}\n{r.stdout}\n{r.stderr}')
    if _stale(capi, [os.path.join(CSRC, 'capi.hip')] + HEADERS + [os.path.abspath(__file__)]):
        run([hipcc, *[f for f in FLAGS if f != '-O3'], *san, '-fno-gpu-sanitize', '-c', os.path.join(CSRC, 'capi.hip'), '-o', capi])
    if _stale(fake, [fake_src]):
        run([clangxx, '-std=c++17', '-fPIC', *san, '-c', fake_src, '-o', fake])
    objs = [capi, fake] + [os.path.join(BUILD, src.replace('.hip', '.o')) for src in SOURCES if src != 'capi.hip'] + \
           [os.path.join(BUILD, src.replace('.hip', '_b1.o')) for src in ONE_TERM]
    if _stale(lib_path, objs):
        run([clangxx, '-shared', '-fPIC', '-fsanitize=address', '-shared-libsan', *objs, '-ldl', '-o', lib_path])
    return lib_path


def asan_runtime() -> str:
    """clang's shared ASan runtime, to preload into an uninstrumented python."""
    r = subprocess.run(['/opt/rocm/lib/llvm/bin/clang', '-print-file-name=libclang_rt.asan-x86_64.so'], capture_output=True, text=True)
    return r.stdout.strip()


if __name__ == '__main__':
    try:
        if '--asan-host' in sys.argv:
            print(build_asan_host(verbose=True))
        else:
            print(build(force='--force' in sys.argv, verbose=True, ablate='--ablate' in sys.argv, asan='--asan' in sys.argv))
    except Exception as e:                      # a failed build must be the LAST thing on the screen, not a stale .so
        print(str(e)[-3000:], file=sys.stderr)
        print('BUILD FAILED', flush=True)
        sys.exit(1)
