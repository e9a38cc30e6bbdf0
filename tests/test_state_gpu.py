"""The product library keeps no state that one call hands to another (include/qsae.h):
concurrent host threads on their own streams, several devices in one process, and two batches in flight on one
stream all return the bits of a plain serial call."""
import threading

import numpy as np
import pytest
import torch

from quantizedsae_amd import BinarySAE, ops, synthetic as S

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
