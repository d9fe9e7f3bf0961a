"""The RX call that allocates nothing keeps all of its streaming state on the
device, so one captured call can be replayed chunk after chunk from a HIP graph:
three replays over fresh input must equal one continuous run of the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_static_call_replays_from_a_hip_graph(orc):
    torch = pytest.importorskip("torch")
    import ua3reo_ddc as ddc
    ddc.lib()
    n_ch, n, chunks = 8, 1024 * 32, 3
    nf, rb = n // 1024, n // 2 * 3
    rng = np.random.default_rng(77)
    phis = rng.integers(0, 1 << 22, n_ch).astype(np.uint32)
    x = rng.integers(-2048, 2048, size=(n_ch, chunks * n)).astype(np.int16)
    d = ddc.Ddc(n_ch)
    d.tune_many(phis)
    side = torch.cuda.Stream()
    d.set_stream(side.cuda_stream)
    adc = torch.zeros((n_ch, rb), dtype=torch.uint8, device="cuda")
    out = torch.zeros((2, n_ch, nf, 2), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        d.rx_process_dev_static(adc.data_ptr(), rb, n, out[0].data_ptr(), out[1].data_ptr())
    got = []
    for k in range(chunks):
        rows = np.stack([ddc.pack_adc12(x[c, k * n:(k + 1) * n]) for c in range(n_ch)])
        adc.copy_(torch.from_numpy(rows))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        got.append(out.cpu().numpy().copy())
    spec = np.concatenate([o[0] for o in got], axis=1)
    voice = np.concatenate([o[1] for o in got], axis=1)
    for c in range(n_ch):
        ws, wv = orc.RxChain(int(phis[c])).run(x[c])
        assert np.array_equal(spec[c], ws), "SPEC differs on channel %d" % c
        assert np.array_equal(voice[c], wv), "VOICE differs on channel %d" % c
    d.set_stream(None)
    d.close()
