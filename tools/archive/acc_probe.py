#!/usr/bin/env python3
"""Accuracy of single rectangular-window periodogram rows (spectrum_sensor_v2's chain) against the float64 oracle,
over several seeds and sizes: worst amplitude error in ulps of the row's peak amplitude, worst relative power error
on bins within 20 dB of the row's typical level, mean relative power error.  Backs the single-row criterion of
tests/test_hip_parity.py::check_single_rows.  usage: acc_probe.py   (OFDM_TOOLS_HIP_LIB selects another build)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'gr-ofdm_tools_amd'))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from ofdm_tools import _hip  # noqa: E402
from oracle import ref_cpu as R  # noqa: E402

ctx = _hip.Context(0)
for nfft in (256, 512, 1024, 2048, 4096):
    ulps, typ, means = [], [], []
    for seed in range(12):
        x = R.synth_iq(nfft * 64, 100 + seed)
        ref = R.chain_sensor_v2(x, nfft).astype(np.float64)
        ch = ctx.chain(nfft, None, True, _hip.EPI_MAG2_OVER_N2, 1)
        rows, n = ch.push(x)
        ch.close()
        rows = rows.astype(np.float64)
        ulps.append(np.max(np.abs(np.sqrt(rows) - np.sqrt(ref)) / np.sqrt(ref.max(axis=1, keepdims=True))) * 2.0 ** 23)
        m = ref >= 1e-2 * np.median(ref, axis=1, keepdims=True)
        typ.append(np.max(np.abs(rows - ref)[m] / ref[m]))
        means.append(np.mean(np.abs(rows - ref) / ref))
    print('%5d: amplitude error %.2f ulp of the peak (worst of 12 seeds), power within 20 dB of typical %.2e, mean %.2e'
          % (nfft, max(ulps), max(typ), np.mean(means)))
