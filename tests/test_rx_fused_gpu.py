"""The two-kernel RX call plan (sample counts whose 512-sample block count is a multiple of
256: rx_cic outputs formed inside the ingest kernel, one tail kernel for the 48 kHz stages
and the state hand-over) against the oracle and against the six-kernel plan of the same
library: every output bit-exact, for single short tiles, several tiles with a partial last
one, calls of both plans alternating on one stream of samples, many channels at once, and
the raw rx_cic tap.  The plan has two ingest kernels: ddc_rx_front_mx (window moments on the
matrix cores, what the library picks: plan 0) and ddc_rx_front_cic (vector ALU only: plan 2
of ddc_debug_rx_plan); both are held to the same results."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

BLK = 512 * 256  # samples of one workgroup's 256 windows


@pytest.fixture(scope="module")
def ddc():
    import ua3reo_ddc
    ua3reo_ddc.lib()
    return ua3reo_ddc


def rows_of(ddc, chans):
    return np.stack([ddc.pack_adc12(x) for x in chans])


@pytest.mark.parametrize("ingest", [0, 2])
@pytest.mark.parametrize("n_wg", [1, 2, 7, 8, 9, 17])
def test_two_kernel_call_equals_oracle_and_split_plan(ddc, orc, n_wg, ingest):
    """n_wg workgroups per channel: 1 / 2 / 7 (a short tile), 8 (one full tile), 9 and 17
    (tiles that reach back into the previous one, the last tile partial)."""
    n = n_wg * BLK
    nch = 3
    rng = np.random.default_rng(100 + n_wg)
    chans = [rng.integers(-2048, 2048, n).astype(np.int16) for _ in range(nch)]
    chans[1][1000:3000] = -2048
    phis = [605867, 3 << 20, 4194303]
    rows = rows_of(ddc, chans)
    d = ddc.Ddc(nch)
    d.tune_many(phis)
    d.debug_rx_plan(ingest)
    spec, voice = d.rx_process(rows, n)
    cic = d.rx_read_cic(n)
    e = ddc.Ddc(nch)
    e.tune_many(phis)
    e.debug_rx_plan(1)
    spec2, voice2 = e.rx_process(rows, n)
    cic2 = e.rx_read_cic(n)
    assert np.array_equal(cic, cic2)
    assert np.array_equal(spec, spec2) and np.array_equal(voice, voice2)
    for c in range(nch):
        ws, wv, wc = orc.RxChain(phis[c]).run(chans[c], want_cic=True)
        assert np.array_equal(cic[c], wc), c
        assert np.array_equal(spec[c], ws) and np.array_equal(voice[c], wv), c
    # a second call of each carries on from the first's state
    more = [rng.integers(-2048, 2048, BLK).astype(np.int16) for _ in range(nch)]
    s3, v3 = d.rx_process(rows_of(ddc, more), BLK)
    s4, v4 = e.rx_process(rows_of(ddc, more), BLK)
    assert np.array_equal(s3, s4) and np.array_equal(v3, v4)
    for c in range(nch):
        assert d.rx_get_phase(c) == e.rx_get_phase(c) == (phis[c] * (n + BLK)) & 0x3FFFFF
    d.close()
    e.close()


@pytest.mark.parametrize("ingest", [0, 2])
def test_rails_and_the_mixer_product_that_wraps(ddc, orc, ingest):
    """-2048 * -2048 = 2^22 is the one mixer product that leaves 23 bits (FPGA/rx_mixer_shift.v:9 keeps
    bits 22..8): rows held at the negative rail with tuning words that put the NCO on -2048 (sine at
    three quarters of a turn, cosine at half a turn) every few samples, the positive rail, and rows
    that flip between the rails, over two calls."""
    n = 2 * BLK
    phis = [1 << 20, 3 << 20, 1 << 21, 1, 4194303, 605867, (1 << 20) + 1, 2796203]
    nch = len(phis)
    rng = np.random.default_rng(77)
    chans = []
    for c in range(nch):
        x = np.full(2 * n, -2048, np.int16)
        if c % 4 == 1:
            x[:] = 2047
        elif c % 4 == 2:
            x[rng.integers(0, 2, 2 * n) == 1] = 2047
        elif c % 4 == 3:
            x[::7] = rng.integers(-2048, 2048, len(x[::7]))
        chans.append(x)
    d = ddc.Ddc(nch)
    d.tune_many(phis)
    d.debug_rx_plan(ingest)
    outs = [d.rx_process(rows_of(ddc, [x[k * n:(k + 1) * n] for x in chans]), n) for k in range(2)]
    d.close()
    spec = np.concatenate([o[0] for o in outs], axis=1)
    voice = np.concatenate([o[1] for o in outs], axis=1)
    for c in range(nch):
        ws, wv = orc.RxChain(phis[c]).run(chans[c])
        assert np.array_equal(spec[c], ws) and np.array_equal(voice[c], wv), c


@pytest.mark.parametrize("sizes", [[BLK, 1024, 2 * BLK, 3072, BLK], [9 * BLK, BLK, 1024 * 5, 8 * BLK],
                                   [1024, BLK, BLK, BLK]])
def test_both_plans_alternate_on_one_stream_of_samples(ddc, orc, sizes):
    total = sum(sizes)
    nch = 2
    rng = np.random.default_rng(len(sizes) + sizes[0])
    chans = [rng.integers(-2048, 2048, total).astype(np.int16) for _ in range(nch)]
    phis = [85333, 1234567]
    d = ddc.Ddc(nch)
    d.tune_many(phis)
    specs, voices, pos = [], [], 0
    for s in sizes:
        sp, vo = d.rx_process(rows_of(ddc, [c[pos:pos + s] for c in chans]), s)
        specs.append(sp)
        voices.append(vo)
        pos += s
    d.close()
    spec, voice = np.concatenate(specs, axis=1), np.concatenate(voices, axis=1)
    for c in range(nch):
        ws, wv = orc.RxChain(phis[c]).run(chans[c])
        assert np.array_equal(spec[c], ws) and np.array_equal(voice[c], wv), c


def test_many_channels_two_tiles_each_three_calls(ddc, orc):
    """700 channels x 9 workgroups (two tiles per channel, so the arrival counter of the tail
    kernel decides which workgroup saves a channel's state), more workgroups than the GPU holds
    at once, three calls into the same buffers (outputs and work buffers of earlier calls in the
    caches).  Against the six-kernel plan on every channel and the oracle on a few."""
    torch = pytest.importorskip("torch")
    nch, n = 700, 9 * BLK
    rb = n // 2 * 3
    nf = n // 1024
    phis = [(977 * c + 605867) % (1 << 22) for c in range(nch)]
    a = ddc.Ddc(nch)
    b = ddc.Ddc(nch)
    a.tune_many(phis)
    b.tune_many(phis)
    b.debug_rx_plan(1)
    g = torch.Generator(device="cuda").manual_seed(2026)
    out_a = torch.zeros((2, nch, nf, 2), dtype=torch.int16, device="cuda")
    out_b = torch.zeros_like(out_a)
    keep = []
    for k in range(3):
        adc = torch.randint(0, 256, (nch, rb), dtype=torch.uint8, device="cuda", generator=g)
        torch.cuda.synchronize()
        a.rx_process_dev(adc.data_ptr(), rb, n, out_a[0].data_ptr(), out_a[1].data_ptr())
        b.rx_process_dev(adc.data_ptr(), rb, n, out_b[0].data_ptr(), out_b[1].data_ptr())
        a.sync()
        b.sync()
        assert torch.equal(out_a, out_b), "call %d: the two call plans differ" % k
        keep.append((adc[[0, 350, 699]].cpu().numpy(), out_a[:, [0, 350, 699]].cpu().numpy()))
    for i, c in enumerate((0, 350, 699)):
        x = np.concatenate([ddc.unpack_adc12(h[i], n) for h, _ in keep])
        ws, wv = orc.RxChain(phis[c]).run(x)
        assert np.array_equal(np.concatenate([o[0, i] for _, o in keep]), ws), c
        assert np.array_equal(np.concatenate([o[1, i] for _, o in keep]), wv), c
    a.close()
    b.close()


def test_shared_stream_status_registers_and_graph_replay_on_the_two_kernel_plan(ddc, orc):
    """The corners of the two-kernel plan: one ADC row fanned out to every channel (row pitch 0),
    the ADC min / max pass riding along, and a call of this shape captured once in a HIP graph on
    the entry point that allocates nothing and replayed over fresh input."""
    torch = pytest.importorskip("torch")
    n_ch, n = 5, 2 * BLK
    rng = np.random.default_rng(404)
    x = rng.integers(-1500, 1500, 3 * n).astype(np.int16)
    x[77] = 2047
    x[n + 5] = -2048
    phis = rng.integers(0, 1 << 22, n_ch).astype(np.uint32)
    d = ddc.Ddc(n_ch)
    d.tune_many(phis)
    d.adc_stats_enable(True)
    outs = [d.rx_process(ddc.pack_adc12(x[k * n:(k + 1) * n])[None, :], n, row_bytes=0) for k in range(2)]
    spec = np.concatenate([o[0] for o in outs], axis=1)
    voice = np.concatenate([o[1] for o in outs], axis=1)
    for c in range(n_ch):
        ws, wv = orc.RxChain(int(phis[c])).run(x[:2 * n])
        assert np.array_equal(spec[c], ws) and np.array_equal(voice[c], wv), c
    otr, _, lo, hi = ddc.parse_status(d.get_status(0))
    assert (lo, hi, otr) == (-2048, 2047 & 0xFFF, 1)
    d.close()
    # graph replay: three chunks through one captured static call
    rb = n // 2 * 3
    e = ddc.Ddc(n_ch)
    e.tune_many(phis)
    side = torch.cuda.Stream()
    e.set_stream(side.cuda_stream)
    adc = torch.zeros((rb,), dtype=torch.uint8, device="cuda")
    out = torch.zeros((2, n_ch, n // 1024, 2), dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        e.rx_process_dev_static(adc.data_ptr(), 0, n, out[0].data_ptr(), out[1].data_ptr())
    got = []
    for k in range(3):
        adc.copy_(torch.from_numpy(ddc.pack_adc12(x[k * n:(k + 1) * n])))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        got.append(out.cpu().numpy().copy())
    for c in (0, n_ch - 1):
        ws, wv = orc.RxChain(int(phis[c])).run(x)
        assert np.array_equal(np.concatenate([o[0, c] for o in got]), ws), c
        assert np.array_equal(np.concatenate([o[1, c] for o in got]), wv), c
    e.set_stream(None)
    e.close()


def test_state_blob_travels_between_the_two_plans(ddc, orc):
    """A state blob saved after a two-kernel call restores into a ctx that goes on with the
    six-kernel plan (and the other way round): the carried state has one format."""
    n_ch = 2
    rng = np.random.default_rng(505)
    x = rng.integers(-2048, 2048, size=(n_ch, 3 * BLK)).astype(np.int16)
    phis = [605867, 31337]
    a = ddc.Ddc(n_ch)
    a.tune_many(phis)
    a.rx_process(np.stack([ddc.pack_adc12(r[:BLK]) for r in x]), BLK)           # two kernels
    blob = a.state_save()
    b = ddc.Ddc(n_ch)
    b.state_restore(blob)
    b.debug_rx_plan(1)
    s1, v1 = b.rx_process(np.stack([ddc.pack_adc12(r[BLK:2 * BLK]) for r in x]), BLK)  # six kernels
    blob2 = b.state_save()
    c = ddc.Ddc(n_ch)
    c.state_restore(blob2)
    s2, v2 = c.rx_process(np.stack([ddc.pack_adc12(r[2 * BLK:]) for r in x]), BLK)      # two kernels again
    for ch in range(n_ch):
        ws, wv = orc.RxChain(phis[ch]).run(x[ch])
        nf = BLK // 1024
        assert np.array_equal(s1[ch], ws[nf:2 * nf]) and np.array_equal(v1[ch], wv[nf:2 * nf])
        assert np.array_equal(s2[ch], ws[2 * nf:]) and np.array_equal(v2[ch], wv[2 * nf:])
    for d_ in (a, b, c):
        d_.close()


def test_more_channels_than_one_grid_dimension_holds(ddc, orc):
    """40000 receivers on one ADC row: ddc_rx_front_mx takes the channel from the grid's y and z
    (y holds at most 32768), so channels on both sides of that seam against the oracle, and every
    channel against the vector-ALU ingest kernel of the same plan."""
    nch, n = 40000, BLK
    rng = np.random.default_rng(808)
    x = rng.integers(-2048, 2048, n).astype(np.int16)
    phis = ((np.arange(nch, dtype=np.uint64) * 104729 + 605867) % (1 << 22)).astype(np.uint32)
    row = ddc.pack_adc12(x)[None, :]
    a = ddc.Ddc(nch)
    a.tune_many(phis)
    spec, voice = a.rx_process(row, n, row_bytes=0)
    a.close()
    b = ddc.Ddc(nch)
    b.tune_many(phis)
    b.debug_rx_plan(2)
    spec2, voice2 = b.rx_process(row, n, row_bytes=0)
    b.close()
    assert np.array_equal(spec, spec2) and np.array_equal(voice, voice2)
    for c in (0, 1, 32767, 32768, 32769, nch - 1):
        ws, wv = orc.RxChain(int(phis[c])).run(x)
        assert np.array_equal(spec[c], ws) and np.array_equal(voice[c], wv), c


@pytest.mark.parametrize("extra", [1024, 5 * 1024, 127 * 1024])
@pytest.mark.parametrize("n_wg", [1, 9])
def test_any_size_runs_as_a_two_kernel_part_and_a_remainder(ddc, orc, n_wg, extra):
    """A call of n_wg * 131072 + extra samples: the library runs the multiple of 131072 as two
    kernels and the rest as six, both writing into the rows of the whole call.  Outputs, the raw
    rx_cic tap and the carried state against the oracle and against the six-kernel plan alone,
    over two calls; the entry point that allocates nothing does the same inside its fixed area."""
    torch = pytest.importorskip("torch")
    n = n_wg * BLK + extra
    nch = 3
    rng = np.random.default_rng(n_wg * 1000 + extra)
    chans = [rng.integers(-2048, 2048, 2 * n).astype(np.int16) for _ in range(nch)]
    phis = [605867, 3 << 20, 4194303]
    d = ddc.Ddc(nch)
    d.tune_many(phis)
    e = ddc.Ddc(nch)
    e.tune_many(phis)
    e.debug_rx_plan(1)
    outs = []
    for k in range(2):
        rows = rows_of(ddc, [x[k * n:(k + 1) * n] for x in chans])
        sp, vo = d.rx_process(rows, n)
        cic = d.rx_read_cic(n)
        sp2, vo2 = e.rx_process(rows, n)
        cic2 = e.rx_read_cic(n)
        assert np.array_equal(cic, cic2), k
        assert np.array_equal(sp, sp2) and np.array_equal(vo, vo2), k
        outs.append((sp, vo))
    spec = np.concatenate([o[0] for o in outs], axis=1)
    voice = np.concatenate([o[1] for o in outs], axis=1)
    for c in range(nch):
        ws, wv = orc.RxChain(phis[c]).run(chans[c])
        assert np.array_equal(spec[c], ws) and np.array_equal(voice[c], wv), c
        assert d.rx_get_phase(c) == e.rx_get_phase(c) == (phis[c] * 2 * n) & 0x3FFFFF
    d.close()
    e.close()
    # the same through the static work area
    s = ddc.Ddc(nch)
    s.tune_many(phis)
    L = ddc.lib()
    assert L.ddc_rx_static_scratch_need(nch, n) <= L.ddc_rx_static_scratch_bytes()
    rb = n // 2 * 3
    out = torch.zeros((2, nch, n // 1024, 2), dtype=torch.int16, device="cuda")
    got = []
    for k in range(2):
        adc = torch.from_numpy(rows_of(ddc, [x[k * n:(k + 1) * n] for x in chans])).cuda()
        torch.cuda.synchronize()
        s.rx_process_dev_static(adc.data_ptr(), rb, n, out[0].data_ptr(), out[1].data_ptr())
        s.sync()
        got.append(out.cpu().numpy().copy())
    assert np.array_equal(np.concatenate([g[0] for g in got], axis=1), spec)
    assert np.array_equal(np.concatenate([g[1] for g in got], axis=1), voice)
    s.close()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_call_sizes_on_one_stream_of_samples(ddc, orc, seed):
    """Seven calls of random sizes (whole two-kernel parts, two-kernel parts with a remainder, and
    small calls that run as six kernels) on one stream: what a call hands to the next — the NCO
    phase, the six mixer outputs in front of the next window, the moments, the 48 kHz histories —
    crosses every pair of plans; rows that end calls on the rails keep the packed hand-over
    words at their extremes."""
    rng = np.random.default_rng(4000 + seed)
    sizes = []
    for _ in range(7):
        kind = rng.integers(0, 3)
        if kind == 0:
            sizes.append(int(rng.integers(1, 4)) * BLK)
        elif kind == 1:
            sizes.append(int(rng.integers(1, 3)) * BLK + int(rng.integers(1, 200)) * 1024)
        else:
            sizes.append(int(rng.integers(1, 100)) * 1024)
    total = sum(sizes)
    nch = 3
    chans = [rng.integers(-2048, 2048, total).astype(np.int16) for _ in range(nch)]
    pos = 0
    for s in sizes:  # the last samples of every call: rails on one channel, alternating rails on another
        pos += s
        chans[1][pos - 8:pos] = -2048
        chans[2][pos - 8:pos:2] = 2047
        chans[2][pos - 7:pos:2] = -2048
    phis = [int(rng.integers(1, 1 << 22)), 3 << 20, int(rng.integers(1, 1 << 22))]
    d = ddc.Ddc(nch)
    d.tune_many(phis)
    specs, voices, pos = [], [], 0
    for s in sizes:
        sp, vo = d.rx_process(rows_of(ddc, [c[pos:pos + s] for c in chans]), s)
        specs.append(sp)
        voices.append(vo)
        pos += s
    d.close()
    spec, voice = np.concatenate(specs, axis=1), np.concatenate(voices, axis=1)
    for c in range(nch):
        ws, wv = orc.RxChain(phis[c]).run(chans[c])
        assert np.array_equal(spec[c], ws) and np.array_equal(voice[c], wv), (c, sizes)
