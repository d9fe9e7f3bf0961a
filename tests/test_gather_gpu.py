"""ddc_allgather_decimated_dev with a one-rank RCCL communicator built by the library itself
(ddc_comm_unique_id + ddc_comm_init = ncclGetUniqueId + ncclCommInitRank): the gathered buffer
of every double-buffer slot must hold the step that was computed into that slot, while the next
step's compute is already queued (the event ordering shard.StepGather has on the torch side)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_library_allgather_one_rank_double_buffered(orc):
    torch = pytest.importorskip("torch")
    import ua3reo_ddc as ddc
    ddc.lib()
    nch, n, steps = 16, 1 << 17, 5
    nf, rb = n // 1024, n // 2 * 3
    phis = [(7919 * c + 4242) & 0x3FFFFF for c in range(nch)]
    g = torch.Generator(device="cuda").manual_seed(12)
    adc = torch.randint(0, 256, (steps, nch, rb), dtype=torch.uint8, device="cuda", generator=g)
    d = ddc.Ddc(nch)
    d.tune_many(phis)
    with pytest.raises(ddc.DdcError, match="communicator"):
        d.allgather_decimated_dev(0, 16, 1, 16)       # no communicator yet
    d.comm_init(1, 0, ddc.comm_unique_id())
    with pytest.raises(ddc.DdcError):
        d.comm_init(1, 0, ddc.comm_unique_id())        # one per ctx
    with pytest.raises(ddc.DdcError):
        d.allgather_decimated_dev(2, 16, 1, 16)        # slots are 0 and 1
    ref = ddc.Ddc(nch)
    ref.tune_many(phis)
    outs = torch.zeros((2, 2, nch, nf, 2), dtype=torch.int16, device="cuda")
    gathered = torch.full((2, 1, 2, nch, nf, 2), 0x5A5A, dtype=torch.int16, device="cuda")
    want = torch.zeros((steps, 2, nch, nf, 2), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    seen = {}
    for k in range(steps):
        b = k & 1
        if k >= 2:
            # before slot b is overwritten: the gather of step k - 2 must be complete and right
            assert d.allgather_wait(b) > 0.0
            seen[k - 2] = gathered[b, 0].clone()
        d.allgather_reuse(b)
        d.rx_process_dev(adc[k].data_ptr(), rb, n, outs[b, 0].data_ptr(), outs[b, 1].data_ptr())
        d.allgather_decimated_dev(b, outs[b].data_ptr(), nf, gathered[b].data_ptr())
        ref.rx_process_dev(adc[k].data_ptr(), rb, n, want[k, 0].data_ptr(), want[k, 1].data_ptr())
    d.allgather_wait(-1)
    ref.sync()
    seen[steps - 2] = gathered[(steps - 2) & 1, 0].clone()
    seen[steps - 1] = gathered[(steps - 1) & 1, 0].clone()
    torch.cuda.synchronize()
    for k in range(steps):
        assert torch.equal(seen[k], want[k]), "gathered step %d differs" % k
    # and the stream of channel 3 is the oracle's
    x = np.concatenate([ddc.unpack_adc12(adc[k, 3].cpu().numpy(), n) for k in range(steps)])
    ws, wv = orc.RxChain(phis[3]).run(x)
    assert np.array_equal(np.concatenate([seen[k][0, 3].cpu().numpy() for k in range(steps)]), ws)
    assert np.array_equal(np.concatenate([seen[k][1, 3].cpu().numpy() for k in range(steps)]), wv)
    d.comm_destroy()
    d.close()
    ref.close()
