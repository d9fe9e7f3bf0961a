#!/usr/bin/env python3
"""Times the 48 kHz kernels on synthetic I/Q already on the GPU: RX audio, TX audio, panadapter."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ua3reo-ddc-transceiver_amd"))
import torch  # noqa: E402
import ua3reo_ddc as u  # noqa: E402

n_ch = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = 192 * blocks * 8 // 3 // 512 * 512 * 3  # multiple of 192 and of 512
n = 1536 * max(1, (192 * blocks) // 1536)
g = torch.Generator(device="cuda").manual_seed(1)
spec = torch.randint(-3000, 3000, (n_ch, n, 2), device="cuda", generator=g, dtype=torch.int32).to(torch.int16)
voice = torch.randint(-3000, 3000, (n_ch, n, 2), device="cuda", generator=g, dtype=torch.int32).to(torch.int16)
audio = torch.zeros((n_ch, n, 2), dtype=torch.int32, device="cuda")
iq = torch.zeros((n_ch, n, 2), dtype=torch.int16, device="cuda")
bins = torch.zeros((n_ch, n // 512, 256), dtype=torch.float32, device="cuda")
rows = torch.zeros((n_ch, n // 512, 256), dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
for mode in ("USB", "AM", "NFM"):
    d = u.Ddc(n_ch)
    d.audio_configure(u.audio_params(mode=mode, dnr=1, filter_width=2700 if mode != "NFM" else 0))
    for what, fn in (("rx", lambda: d.audio_rx_process_dev(spec.data_ptr(), voice.data_ptr(), n, audio.data_ptr())),
                     ("tx", lambda: d.audio_tx_process_dev(spec.data_ptr(), n, iq.data_ptr())),
                     ("fft", lambda: d.fft_process_dev(spec.data_ptr(), n, bins.data_ptr(), rows.data_ptr()))):
        fn(); d.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            fn()
        d.sync()
        dt = (time.perf_counter() - t0) / 3
        print("%-4s %-3s %d channels x %d frames: %.3f ms  (%.1fx real time)" % (mode, what, n_ch, n, dt * 1e3, n / 48000.0 / dt))
    d.close()
