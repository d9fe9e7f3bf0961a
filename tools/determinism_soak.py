"""Soak of the RX call at the bench shape: the same rows and the same state, call after call, must give the
same bytes every time (a missing wait or a hazard in the ingest kernel's software pipeline would show up as
a rare difference).  usage: python3 tools/determinism_soak.py [calls]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ua3reo-ddc-transceiver_amd"))
import torch
import ua3reo_ddc

calls = int(sys.argv[1]) if len(sys.argv) > 1 else 300
nch, S = 1024, 1 << 20
dev = torch.device("cuda:0")
rb = S // 2 * 3
rows = torch.empty((nch, rb), dtype=torch.uint8, device=dev)
ua3reo_ddc.lib()
ua3reo_ddc.synth_adc_dev(0, nch, S, rows.data_ptr(), rb)
torch.cuda.synchronize()
nf = S // 1024
spec = torch.zeros((nch, nf, 2), dtype=torch.int16, device=dev)
voice = torch.zeros_like(spec)
ref_s = ref_v = None
d = ua3reo_ddc.Ddc(nch, device=0)
phis = [ua3reo_ddc.phrase_from_frequency(1000000 + 48000 * c)[0] for c in range(nch)]
d.tune_many(phis)
bad = 0
t0 = time.time()
for k in range(calls):
    d.rx_reset()
    d.tune_many(phis)
    d.rx_process_dev(rows.data_ptr(), rb, S, spec.data_ptr(), voice.data_ptr())
    d.sync()
    if ref_s is None:
        ref_s, ref_v = spec.clone(), voice.clone()
        assert bool(ref_s.any()) and bool(ref_v.any()), "the first call left the outputs at zero"
    elif not (torch.equal(spec, ref_s) and torch.equal(voice, ref_v)):
        bad += 1
        print("call", k, "differs:", int((spec != ref_s).sum()), int((voice != ref_v).sum()), flush=True)
    spec.zero_(); voice.zero_()
    if k % 500 == 499:
        print("call", k + 1, "of", calls, "differences so far", bad, "%.1f s" % (time.time() - t0), flush=True)
d.close()
print("soak:", calls, "calls,", bad, "differing")
sys.exit(1 if bad else 0)
