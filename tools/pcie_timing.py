#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-pointer entry point ddc_rx_process (rows in pageable host memory)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ua3reo-ddc-transceiver_amd"))
import ua3reo_ddc as u  # noqa: E402

n_ch, samples = 256, 1 << 20
rows = np.random.default_rng(0).integers(0, 256, (n_ch, samples // 2 * 3), dtype=np.uint8)
d = u.Ddc(n_ch)
d.tune_many([(977 * c + 605867) % (1 << 22) for c in range(n_ch)])
d.rx_process(rows, samples)
t0 = time.perf_counter()
for _ in range(3):
    d.rx_process(rows, samples)
dt = (time.perf_counter() - t0) / 3
print("ddc_rx_process, host rows: %d channels x %d samples in %.2f ms = %.0f MS/s (%.1f GB/s of packed ADC over PCIe)" % (
    n_ch, samples, dt * 1e3, n_ch * samples / dt / 1e6, rows.nbytes / dt / 1e9))

# the same rows in page-locked host memory (what a caller that owns its buffers should pass)
import torch  # noqa: E402

pinned = torch.from_numpy(rows).pin_memory()
pr = pinned.numpy()
d.rx_process(pr, samples)
t0 = time.perf_counter()
for _ in range(3):
    d.rx_process(pr, samples)
dt = (time.perf_counter() - t0) / 3
print("ddc_rx_process, pinned host rows: %d channels x %d samples in %.2f ms = %.0f MS/s (%.1f GB/s)" % (
    n_ch, samples, dt * 1e3, n_ch * samples / dt / 1e6, rows.nbytes / dt / 1e9))
