"""Enumeration of the launch-recipe table through oth__debug_recipe (test infrastructure; no GPU: the choice of kernel
build, detrend form, pilot, schedule, chunk sizes and grid is pure host logic in csrc/api.hip resolve_recipe()).

  python tests/recipes.py --write     regenerate tests/golden/recipes_small.txt and recipes_full.sha256 after an
                                      INTENDED routing change (review the diff of the small table)
"""
import ctypes
import hashlib
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
WINDOWS = {0: 'rect', 1: 'confined', 2: 'wide'}
DETREND = {0: 'none', 1: 'constant', 3: 'fast'}


def _lib():
    sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
    from ofdm_tools import _hip
    return _hip.load()


def recipe(lib, nfft, nperseg, noverlap, window=1, detrend=1, two_channel=0, kernel=0, variant=None, sched=2, nseg=600,
           nstreams=1, cu=256, runtime=0):
    buf = ctypes.create_string_buffer(512)
    rc = lib.oth__debug_recipe(nfft, nperseg, noverlap, window, detrend, two_channel, kernel,
                               variant.encode() if variant else None, sched, nseg, nstreams, cu, runtime, buf, 512)
    if rc:
        return 'error %d: %s' % (rc, lib.oth_last_error(None).decode())
    return buf.value.decode()


def grid(full):
    sizes = (64, 256, 512, 1024, 2048, 4096, 8192, 16384)
    nsegs = (1, 7, 8, 31, 600, 20000, 131071) if full else (2, 600, 131071)
    streams = (1, 8, 64, 100) if full else (1, 64)
    for nfft in sizes:
        for frac, ov in itertools.product((1, 2, 4), (0, 2, 4)):      # nperseg = nfft / frac; overlap 0, 1/2, 3/4
            nperseg = nfft // frac
            nov = 0 if ov == 0 else nperseg - nperseg // ov
            if not full and (frac == 2 or ov == 4):
                continue
            for win, det in itertools.product(WINDOWS, DETREND):
                if not full and (win == 2) != (det == 1 and frac == 1 and ov == 2):      # wide windows: where they change the route
                    if win == 2:
                        continue
                for two in ((0, 1) if frac == 1 and nfft in (1024, 4096) else (0,)):
                    for nseg, ns in itertools.product(nsegs, streams):
                        if two and ns != 1:
                            continue
                        yield dict(nfft=nfft, nperseg=nperseg, noverlap=nov, window=win, detrend=det, two_channel=two,
                                   nseg=nseg, nstreams=ns)


def grid_any(full):
    """Round 6: lengths outside the power-of-two kernels (csrc/fft_any.hip / fft_tl.hip): tiny powers of two, 2-3-5-7-smooth
    lengths, Bluestein in one launch and through the four-step route, powers of two above 16384, and what is refused."""
    sizes = (8, 32, 96, 1000, 1021, 1536, 8191, 10007, 15000, 20000, 32768, 65536, 131072, 1048576, 524288 + 2, 2097152)
    for nfft in sizes:
        for frac, ov in itertools.product((1, 4), (0, 2)):
            nperseg = nfft // frac
            nov = 0 if ov == 0 else nperseg - nperseg // ov
            for det, two in itertools.product((0, 1), (0, 1)):
                for nseg, ns in itertools.product((1, 40, 600, 131071) if full else (2, 600), (1, 8) if full else (1,)):
                    if two and ns != 1:
                        continue
                    yield dict(nfft=nfft, nperseg=nperseg, noverlap=nov, window=1, detrend=det, two_channel=two, nseg=nseg,
                               nstreams=ns)
        yield dict(nfft=nfft, nperseg=nfft, noverlap=nfft // 2, window=1, detrend=1, two_channel=0, nseg=600, nstreams=1,
                   kernel=2)      # OTH_KERNEL_TUNED: refused with a reason


def table(lib, full):
    lines = []
    for k in itertools.chain(grid(full), grid_any(full)):
        key = 'nfft=%d nperseg=%d noverlap=%d window=%s detrend=%s%s nseg=%d streams=%d%s' % (
            k['nfft'], k['nperseg'], k['noverlap'], WINDOWS[k['window']], DETREND[k['detrend']],
            ' two-channel' if k['two_channel'] else '', k['nseg'], k['nstreams'], ' tuned-only' if k.get('kernel') == 2 else '')
        lines.append(key + '  ->  ' + recipe(lib, **k))
    return lines


def digest(lines):
    return hashlib.sha256('\n'.join(lines).encode()).hexdigest()


if __name__ == '__main__':
    lib = _lib()
    small, full = table(lib, False), table(lib, True)
    if '--write' in sys.argv:
        open(os.path.join(GOLDEN, 'recipes_small.txt'), 'w').write('\n'.join(small) + '\n')
        open(os.path.join(GOLDEN, 'recipes_full.sha256'), 'w').write('%s  %d recipes\n' % (digest(full), len(full)))
    print('%d / %d recipes, full digest %s' % (len(small), len(full), digest(full)))
