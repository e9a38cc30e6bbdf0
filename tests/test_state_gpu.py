"""The product library keeps no state that one call hands to another (include/qsae.h):
concurrent host threads on their own streams, several devices in one process, and two batches in flight on one
stream all return the bits of a plain serial call."""
import threading

import numpy as np
import pytest
import torch

import oracle
from quantizedsae_amd import BaselineSparseAutoencoder, BinarySAE, QuantizedMatryoshkaSAE, ops, synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
D, H, N_BITS = 512, 32768, 4


def make_model(device, seed=31):
    sd = S.binary_sae_params(seed, D, H, N_BITS, 30.0, 0.05, 0.1)
    model = BinarySAE(D, H, gamma=4.0, n_bits=N_BITS)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return model.to(device).eval()


def batch(seed, B, device, bad_rows=()):
    x = S.activations(seed, B, D)
    for r in bad_rows:                      # rows the candidate pass cannot serve: they take the exact fallback
        x[r, 3] = np.nan
    return torch.from_numpy(x).to(device)


def call(model, x, info=None):
    lin, dec = model.encoder.linear, model.decoder
    pw = model._prefilter_weights()
    return ops.binary_forward_prefilter(x, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], model.top_k,
                                        dec.packed()["packed"], dec.n_bits, dec.quantization_step, dec.bias.detach(),
                                        info=info)


def same_bits(a, b):
    return all(torch.equal(u.view(torch.int32), v.view(torch.int32)) for u, v in zip(a, b))


def test_two_host_threads_on_two_streams_match_the_serial_bits():
    """Thread A's batches have rows that need the exact fallback, thread B's have none: with a process-wide 'rows the
    last call flagged' (round 1) one thread's count sized the other's fallback.  Every call also forks a zero-fill kernel
    onto a side stream and joins it back: side stream and events belong to the calling thread."""
    model = make_model(DEV)
    model.decoder.packed(), model._prefilter_weights()
    B = 4096
    xa = [batch(40 + i, B, DEV, bad_rows=(5, 77, 1000 + i)) for i in range(3)]
    xb = [batch(50 + i, B, DEV) for i in range(3)]
    infos = [{} for _ in range(6)]
    serial = [call(model, x, info) for x, info in zip(xa + xb, infos)]
    torch.cuda.synchronize()
    assert [i["flagged_rows"] >= 3 for i in infos[:3]] == [True] * 3
    assert all(i["flagged_rows"] < 3 for i in infos[3:])
    results, errors = {}, []
    gate = threading.Barrier(2)

    def worker(name, xs):
        try:
            stream = torch.cuda.Stream(device=DEV)
            with torch.cuda.stream(stream):
                out = []
                gate.wait()
                for rep in range(4):                      # keep both threads' launches interleaving for a while
                    out = [call(model, x) for x in xs]
                stream.synchronize()
            results[name] = out
        except Exception as e:                            # surfaced in the main thread
            errors.append(e)

    ta = threading.Thread(target=worker, args=("a", xa))
    tb = threading.Thread(target=worker, args=("b", xb))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errors, errors
    for got, want in zip(results["a"] + results["b"], serial):
        assert same_bits(got, want)


def test_two_batches_in_flight_on_one_stream():
    """submit / finish (qsae_prefilter_submit / _finish): batch 2 is queued before batch 1's count is read; outputs
    equal the blocking call's, flagged rows included; the model-level handle API does the same."""
    model = make_model(DEV)
    B = 4096
    x1, x2 = batch(60, B, DEV, bad_rows=(9, 4000)), batch(61, B, DEV)
    want1, want2 = call(model, x1), call(model, x2)
    lin, dec = model.encoder.linear, model.decoder
    pw = model._prefilter_weights()
    args = lambda x: (x, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], model.top_k, dec.packed()["packed"],
                      dec.n_bits, dec.quantization_step, dec.bias.detach())
    p1 = ops.binary_forward_prefilter_submit(*args(x1), slot=0)
    p2 = ops.binary_forward_prefilter_submit(*args(x2), slot=1)
    got1 = p1.finish()
    got2 = p2.finish()
    assert p1.flagged_rows >= 2 and p2.flagged_rows < 2
    assert same_bits(got1, want1) and same_bits(got2, want2)
    h1 = model.forward_submit(x1, slot=0)
    h2 = model.forward_submit(x2, slot=1)
    lat1, rec1, _ = h1.result()
    lat2, rec2, _ = h2.result()
    assert same_bits((lat1, rec1), (want1[2], want1[3])) and same_bits((lat2, rec2), (want2[2], want2[3]))
    # a small batch takes another path: the handle is already finished, same results as forward()
    xs = x1[:64]
    hs = model.forward_submit(xs)
    assert same_bits(hs.result()[:2], model(xs)[:2])


def test_submit_finish_is_guarded_against_its_callers():
    """What may happen between submit and finish: (a) a blocking call on the same stream -- it used to
    share slot 0's workspace and reset the flagged-row list that finish reads; (b) finish called with another stream
    current -- the fallback has to run on the submit stream, on the submit stream's workspace; (c) a second submit into
    a slot that is still in flight raises instead of overwriting the first batch's lists; (d) a handle dropped without
    finish() releases its slot."""
    model = make_model(DEV)
    B = 4096
    x1, x2 = batch(63, B, DEV, bad_rows=(9, 4000, 17)), batch(64, B, DEV, bad_rows=(1,))
    want1, want2 = call(model, x1), call(model, x2)
    lin, dec = model.encoder.linear, model.decoder
    pw = model._prefilter_weights()
    args = lambda x: (x, lin.weight.detach(), lin.bias.detach(), pw["Wq"], pw["meta"], model.top_k, dec.packed()["packed"],
                      dec.n_bits, dec.quantization_step, dec.bias.detach())
    # (a)
    p1 = ops.binary_forward_prefilter_submit(*args(x1), slot=0)
    mid = call(model, x2)                                 # blocking call on the same stream, in between
    got1 = p1.finish()
    assert p1.flagged_rows >= 3
    assert same_bits(got1, want1) and same_bits(mid, want2)
    # (b)
    p1 = ops.binary_forward_prefilter_submit(*args(x1), slot=0)
    other = torch.cuda.Stream(device=DEV)
    with torch.cuda.stream(other):
        got1 = p1.finish()                                # enqueued on the submit stream, whatever is current
    torch.cuda.current_stream().synchronize()
    assert same_bits(got1, want1)
    # (c)
    p1 = ops.binary_forward_prefilter_submit(*args(x1), slot=0)
    with pytest.raises(RuntimeError, match="slot 0"):
        ops.binary_forward_prefilter_submit(*args(x2), slot=0)
    p2 = ops.binary_forward_prefilter_submit(*args(x2), slot=1)
    assert same_bits(p1.finish(), want1) and same_bits(p2.finish(), want2)
    # (d)
    p1 = ops.binary_forward_prefilter_submit(*args(x1), slot=0)
    del p1
    p1 = ops.binary_forward_prefilter_submit(*args(x1), slot=0)
    assert same_bits(p1.finish(), want1)
    torch.cuda.synchronize()


def test_workspace_cache_is_bounded():
    """Scratch buffers are cached per (device, stream, kind, slot); short-lived streams must not leave one behind each."""
    model = make_model(DEV)
    x = batch(65, 4096, DEV)
    want = call(model, x)
    for _ in range(ops.WORKSPACE_CACHE_PER_DEVICE + 4):
        st = torch.cuda.Stream(device=DEV)
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            got = call(model, x)
        st.synchronize()
        assert same_bits(got, want)
    assert len([k for k in ops._workspaces if k[0] == 0]) <= ops.WORKSPACE_CACHE_PER_DEVICE


def test_unpolarised_checkpoint_keeps_the_one_call_forward_and_the_pipeline():
    """A BinarySAE whose decoder logits are not saturated decodes with the reference's soft integers (sae/binary.py:
    26-38).  That path now also runs inside the refinement kernel (qsae_table_forward_prefilter) and through submit /
    finish: bit-identical to the separate decode kernel, and equal to the oracle's soft forward."""
    import warnings
    Hs = 8192
    sd = S.binary_sae_params(33, D, Hs, N_BITS, 30.0, 0.05, 0.1)
    rng = np.random.default_rng(5)
    sd["decoder.weight"] = (rng.standard_normal(sd["decoder.weight"].shape) * 2.0).astype(np.float32)
    model = BinarySAE(D, Hs, gamma=4.0, n_bits=N_BITS)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    model = model.to(DEV).eval()
    B = 4096
    x = batch(66, B, DEV, bad_rows=(7, 2222))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        assert model.decoder.resolved_decode_mode() == "soft" and model.resolved_latent_path(B) == "prefilter"
        lat, rec, _ = model(x)
        assert model.last_flagged_rows >= 2
        model.fuse_decode = False
        lat_s, rec_s, _ = model(x)
        model.fuse_decode = True
        assert same_bits((lat, rec), (lat_s, rec_s))
        h1, h2 = model.forward_submit(x, slot=0), model.forward_submit(torch.roll(x, 1, 0), slot=1)
        lat1, rec1, _ = h1.result()
        lat2, rec2, _ = h2.result()
        assert same_bits((lat1, rec1), (lat, rec)) and same_bits((torch.roll(rec2, -1, 0),), (rec,))
    ok = [r for r in range(0, B, 67) if r not in (7, 2222)]
    want = oracle.binary_forward(x[ok].cpu().numpy(), sd["encoder.0.weight"], sd["encoder.0.bias"], sd["decoder.weight"],
                                 sd["decoder.bias"], n_bits=N_BITS, gamma=4.0, k=model.top_k, soft=True)
    assert np.array_equal(lat[ok].cpu().numpy(), want["latent"])
    np.testing.assert_allclose(rec[ok].cpu().numpy(), want["reconstruction"], rtol=1e-5, atol=1e-5)


def test_baseline_and_matryoshka_have_the_two_call_forward():
    """forward_submit / result for the other two users of the candidate sweep: BaselineSparseAutoencoder (fp32 decoder
    rows decoded by the refinement kernel) and QuantizedMatryoshkaSAE (qsae_encode_bits_prefilter_submit / _finish):
    two batches in flight return the bits of the blocking forward, rows that need the exact kernels included."""
    B = 4096
    base = BaselineSparseAutoencoder(D, 16384).to(DEV).eval()
    x1, x2 = batch(67, B, DEV, bad_rows=(3, 99)), batch(68, B, DEV)
    w1, w2 = base(x1), base(x2)
    base.latent_path = "fused"
    assert same_bits(base(x2), w2)                        # (the one-call decode equals the exact path + separate decode)
    base.latent_path = "auto"
    h1, h2 = base.forward_submit(x1, slot=0), base.forward_submit(x2, slot=1)
    assert same_bits(h1.result(), w1) and same_bits(h2.result(), w2)
    assert same_bits(base.forward_submit(x2, want_dense=False).result(), base.forward_compact(x2))
    mat = QuantizedMatryoshkaSAE(D, 16384, top_k=32, abs_range=4, n_bits=4).to(DEV).eval()
    with torch.no_grad():
        mat.encoder[0].bias.fill_(-0.6)                   # ~2.4 sigma of the pre-activations: few units fire
    assert mat.resolved_bits_path(B) == "prefilter"
    g1, l1 = mat(x1)
    g2, l2 = mat(x2)
    h1, h2 = mat.forward_submit(x1, slot=0), mat.forward_submit(x2, slot=1)
    sg1, sl1 = h1.result()
    sg2, sl2 = h2.result()
    assert same_bits(sl1, l1) and same_bits(sl2, l2) and same_bits(sg1, g1) and same_bits(sg2, g2)
    # slots belong to a module: two models with a batch in flight in "slot 0" of the same stream do not collide
    hb, hm = base.forward_submit(x1, slot=0), mat.forward_submit(x1, slot=0)
    assert same_bits(hb.result(), w1) and same_bits(hm.result()[1], l1)
    # dense activations: the band classification has the two-call form too
    with torch.no_grad():
        mat.encoder[0].bias.fill_(0.0)
    mat.bits_path = "band"
    mat.decoder.precision = "fp32"
    g1, l1 = mat(x1)
    h1, h2 = mat.forward_submit(x1, slot=0), mat.forward_submit(x2, slot=1)
    sg1, sl1 = h1.result()
    h2.result()
    assert mat.last_flagged_rows == 0 and same_bits(sl1, l1) and same_bits(sg1, g1)


def test_sweep_profile_events_are_per_call():
    model = make_model(DEV)
    x = batch(62, 4096, DEV)
    call(model, x)
    ops.kernel_timer.reset()
    ops.sweep_timing(True)
    for _ in range(3):
        call(model, x)
    ops.sweep_timing(False)
    call(model, x)                                        # not timed: the pair of the previous call is not reused
    torch.cuda.synchronize()
    ms, n, frac = ops.sweep_timing_collect(H)
    assert n == 3 and 0.01 < ms < 50.0 and frac == 1.0
    ops.kernel_timer.reset()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs in one process")
def test_one_process_walks_over_two_devices():
    """A host that loops over the devices of a node in ONE process (SAEWrapper.to('cuda:1') after a call on cuda:0):
    LDS limits are raised per device, helper streams / events are per (thread, device), launches go to the tensors'
    device whatever the thread's current device is."""
    outs = []
    for d in (0, 1, 0, 1):
        dev = f"cuda:{d}"
        model = make_model(dev)
        x = batch(70, 4096, dev, bad_rows=(11,))
        with torch.cuda.device(0):                        # deliberately not the tensors' device
            lat, rec, _ = model(x)
            small = model(x[:100])[1]                     # in-place path (dense GEMM + top-k kernels) as well
        torch.cuda.synchronize(dev)
        outs.append((lat.cpu(), rec.cpu(), small.cpu()))
    for o in outs[1:]:
        assert same_bits(o, outs[0])


def test_small_batch_forward_is_capturable_in_a_hip_graph():
    """The in-place path of small batches (dense encoder -> in-place top-k -> sparse decode) has no host read-back: the whole
    forward can be captured once and replayed (torch.cuda.CUDAGraph), new inputs copied into the captured buffer; replays
    return the eager bits.  (The candidate-sweep path of large batches reads its flagged-row count on the host.)"""
    model = make_model(DEV)
    model.decoder.packed()
    xs = [batch(80 + i, 64, DEV) for i in range(3)]
    want = [model(x) for x in xs]
    static_x = xs[0].clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        model(static_x)
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = model(static_x)
    for x, w in zip(xs, want):
        static_x.copy_(x)
        g.replay()
        torch.cuda.synchronize()
        assert same_bits(out[:2], w[:2])
