#!/usr/bin/env python3
"""Headline benchmark: activations/sec (+ recon-MSE) of BinarySAE 512->32768 n_bits=4 forward.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one full ``BinarySAE.forward`` (reference signature: dense sparse latent [B, H],
reconstruction [B, D], polarize loss) over one 65536-row batch of synthetic 512-d activations
already resident in HBM, plus the on-device squared-error accumulation of the recon-MSE metric.
Rows shard across ranks with no data-path collective (weak scaling: 65536 rows per GPU); the only
exchange is the final 2-element fp64 all-reduce of the MSE pair and the max-over-ranks timing.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

D, H, N_BITS, GAMMA = 512, 32768, 4, 4.0
ROWS_PER_GPU = 65536
K_TOP = int(H * 0.002)
FLOPS_PER_ROW_ENCODER = 2 * D * H                       # the dominant kernel
FLOPS_PER_ROW = 2 * D * H + 2 * K_TOP * D               # 33 620 992 algorithmic FLOP per row
BYTES_PER_ROW_DENSE = 4 * D + 4 * H + 4 * D             # 135 168 B per row, dense-latent return
PEAK_FP32_MFMA_TFLOPS = 157.3                           # MI355X_MICROARCH.md, chip-level parameters
PEAK_HBM_GBPS = 8000.0                                  # HBM3E, MI355X_MICROARCH.md
PEAK_FP16_MFMA_TFLOPS = 2500.0                          # dense (the ~5 PF headline figure includes 2:1 sparsity)
WEIGHT_BYTES_PER_BATCH = 4 * H * D + 4 * H + (H * D * N_BITS) // 8 + 4 * D      # 75 630 592 B


TRAFFIC_FILES = ("r03_traffic.json", "r02_traffic.json", "r01_traffic.json")      # newest first
TRAFFIC_SOURCE = None


def pmc_traffic(key):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes
    (profiles/rNN_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs of this same
    command, FETCH_SIZE doubled per the gfx950 correction).  bench.py cannot profile itself: this is a
    STATIC figure (see `traffic_source` in the output), quoted only for the kernel/shape it was collected
    on, otherwise null."""
    global TRAFFIC_SOURCE
    for name in TRAFFIC_FILES:
        try:
            doc = json.load(open(ROOT / "profiles" / name))
            val = doc["kernels"][key]["hbm_side_bytes_per_launch"]
            TRAFFIC_SOURCE = (f"static: profiles/{name} (rocprofv3 --pmc passes of this command on an MI355X, FETCH_SIZE x2 per the "
                              "gfx950 correction); not measured during this run")
            return val
        except (OSError, KeyError, ValueError, TypeError):
            continue
    return None


def build_model(device, seed=1):
    """BinarySAE(512, 32768, gamma=4, n_bits=4): xavier-uniform encoder, zero biases, saturated
    +-30 decoder logits with fair random bits, torch generator on device."""
    from quantizedsae_amd import BinarySAE
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    model = BinarySAE(D, H, gamma=GAMMA, n_bits=N_BITS).to(device).eval()
    bound = (6.0 / (D + H)) ** 0.5
    with torch.no_grad():
        model.encoder[0].weight.copy_((torch.rand((H, D), device=device, generator=g) * 2 - 1) * bound)
        model.encoder[0].bias.zero_()
        bits = torch.randint(0, 2, (H, D * N_BITS), device=device, generator=g, dtype=torch.int8)
        model.decoder.weight.copy_((bits.float() * 2 - 1) * 30.0)
        model.decoder.bias.zero_()
    return model


def cpu_baseline(model, x_sample, max_threads):
    """Time the torch-CPU restatement of the reference's op sequence (oracle/torch_restatement.py)
    on the host cores, same weights, a bounded sample of the same batch.  The intra-op thread count
    is swept (the box may expose more hardware threads than its CPU share) and the best is kept."""
    from oracle import torch_restatement as T
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    xs = x_sample.cpu()
    args = (xs, sd["encoder.0.weight"], sd["encoder.0.bias"], sd["decoder.weight"], sd["decoder.bias"])
    kw = dict(n_bits=N_BITS, gamma=GAMMA, k=K_TOP)
    best, threads, reps = float("inf"), 1, 0
    t_all = time.perf_counter()
    for nt in sorted({t for t in (8, 16, 32, 64, max_threads) if t <= max_threads}):
        if time.perf_counter() - t_all > 25.0:
            break
        torch.set_num_threads(nt)
        T.binary_forward(*args, **kw)      # warm-up
        for _ in range(2):
            t0 = time.perf_counter()
            T.binary_forward(*args, **kw)
            dt = time.perf_counter() - t0
            reps += 1
            if dt < best:
                best, threads = dt, nt
    cpu_name = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_name = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except AttributeError:
        affinity = os.cpu_count() or 1
    return {"value": xs.shape[0] / best, "unit": "activations/s", "cores": threads, "kind": "port",
            "host_logical_cpus": os.cpu_count(), "host_cpus_in_affinity_mask": affinity,
            "sample": f"{xs.shape[0]} rows of the same batch and weights, torch-CPU op-sequence restatement of "
                      f"BinarySAE.forward (oracle/torch_restatement.py), fp32, best of {reps} runs over intra-op thread counts up to {max_threads} "
                      f"(best at {threads}), host CPU: {cpu_name}"}


def secondary_configs(device, x):
    """forward() wall time of the other configurations of BASELINE.json (3: ternary, 4: matryoshka; plus the
    baseline top-32 SAE, the residual SAE at half batch, and the compact-output BinarySAE path), same batch,
    synthetic parameters of SURVEY.md 8d.  Reported beside the headline, never part of `value`."""
    from quantizedsae_amd import (BaselineSparseAutoencoder, BinarySAE, QuantizedMatryoshkaSAE, ResidualQuantizedSAE,
                                  TernarySparseAutoencoder)
    out = []

    def run(name, model, call, rows, flops_per_row, submit=None):
        """call(model, x): the blocking forward(), timed with one HIP-event pair per repetition (median) and by the wall
        clock (ms_wall_mean: includes whatever the host waits for).  submit(model, x, slot): the two-call form where the
        model has one -- two batches in flight, wall clock over the whole loop (ms_pipelined_wall_mean)."""
        model = model.to(device).eval()
        xb = x[:rows]
        reps = 10
        with torch.no_grad():
            for _ in range(2):
                r = call(model, xb)
            torch.cuda.synchronize()
            evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
            t0 = time.perf_counter()
            for a, b in evs:                                  # one event pair per repetition, on the launch stream
                a.record()
                r = call(model, xb)
                b.record()
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / reps * 1e3
            gpu = sorted(a.elapsed_time(b) for a, b in evs)
            ms = gpu[reps // 2]
            entry = {"config": name, "rows": rows, "ms_per_step": ms, "ms_min": gpu[0], "ms_max": gpu[-1],
                     "ms_wall_mean": wall, "reps": reps, "timing": "blocking forward(): HIP events per repetition, median",
                     "activations_per_s": rows / ms * 1e3, "algorithmic_tflops": flops_per_row * rows / ms / 1e9}
            if submit is not None:
                n = 2 * reps
                pending = None
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for i in range(n):
                    h = submit(model, xb, i % 2)
                    if pending is not None:
                        r = pending.result()
                    pending = h
                r = pending.result()
                torch.cuda.synchronize()
                wall_p = (time.perf_counter() - t0) / n * 1e3
                entry.update({"ms_pipelined_wall_mean": wall_p,
                              "pipelined": "forward_submit / result, two batches in flight, wall clock over "
                                           f"{n} steps (sync on both sides)",
                              "activations_per_s_pipelined": rows / wall_p * 1e3})
        del r, model
        torch.cuda.empty_cache()
        out.append(entry)

    B = x.shape[0]
    with torch.no_grad():
        m = TernarySparseAutoencoder(D, H)
        m.decoder.weight.normal_(0, 0.5)
        run("config 3: TernarySparseAutoencoder(512,32768), dense latent + reconstruction (exact-fp32 MFMA encoder; decoder on the "
            "bf16 matrix pipe: the latent as three exact bf16 terms against the {-1,0,+1} dictionary, fp32 accumulation)", m,
            lambda mm, xx: mm(xx), B, 4.0 * D * H)
        m.decoder.precision = "fp32"
        run("config 3 with the exact-fp32 MFMA decoder (decoder.precision = 'fp32')", m, lambda mm, xx: mm(xx), B, 4.0 * D * H)
        m.decoder.precision = "auto"
        m.encoder.precision = "emulated"
        run("config 3, OPT-IN encoder.precision = 'emulated': the dense ReLU latent at fp32 accuracy (not bit-identical to the "
            "fmaf chain: two fp16 terms per operand, three partial contractions on the fp16 matrix pipe, fp32 accumulation; "
            "measured rms error 0.8e-7 of a row's largest latent against 1.1e-7 for the exact chain), decoder on the bf16 pipe", m,
            lambda mm, xx: mm(xx), B, 4.0 * D * H)
        m = QuantizedMatryoshkaSAE(D, H, top_k=32, abs_range=4, n_bits=4)
        m.encoder[0].bias.fill_(-0.44)                     # -2.5 standard deviations of the latent: ~200 of 32768 units fire per row
        m.decoder.weight.uniform_(-1, 1)
        m.decoder.weight_mirror.uniform_(-1, 1)
        run("config 4: QuantizedMatryoshkaSAE(512,32768,n_bits=4), 4 reconstruction levels, encoder bias -2.5 sigma "
            "(~200 active units per row): z bits from the fp16 candidate sweep + exact re-evaluation at the cutoff, "
            "decoder walks the active units; bit-identical to the dense kernels", m, lambda mm, xx: mm(xx), B, 4.0 * D * H,
            submit=lambda mm, xx, sl: mm.forward_submit(xx, slot=sl))
        m.bits_path = "dense"
        m.decoder.SPARSE_MAX_ACTIVE_FRACTION = 0.0         # dense decoder whatever the activation density
        run("config 4, same model through the dense kernels only (exact-fp32 MFMA encoder for every latent + dense decoder on "
            "the bf16 matrix pipe: z * 2 scale as three exact bf16 terms, fp32 accumulation)", m,
            lambda mm, xx: mm(xx), B, 4.0 * D * H)
        m = QuantizedMatryoshkaSAE(D, H, top_k=32, abs_range=4, n_bits=4)
        m.decoder.weight.uniform_(-1, 1)
        m.decoder.weight_mirror.uniform_(-1, 1)
        run("config 4 at random init (encoder bias 0: half of the units fire): z bits by fp16 classification of every latent + "
            "exact re-evaluation of the ~1 % inside the error band (qsae_encode_bits_band), decoder on the bf16 matrix pipe", m,
            lambda mm, xx: mm(xx), B, 4.0 * D * H)
        run("baseline_sae: BaselineSparseAutoencoder(512,32768) top-32", BaselineSparseAutoencoder(D, H),
            lambda mm, xx: mm(xx), B, 2.0 * D * H + 2.0 * 32 * D, submit=lambda mm, xx, sl: mm.forward_submit(xx, slot=sl))
        m = ResidualQuantizedSAE(D, H, top_k=32, abs_range=1.5, n_bits=4)
        run("rq_sae: ResidualQuantizedSAE(512,32768,n_bits=4), random init (dense activations)", m, lambda mm, xx: mm(xx),
            min(B, 32768), 4.0 * D * H)
        m = BinarySAE(D, H, gamma=GAMMA, n_bits=N_BITS)
        m.decoder.weight.copy_(torch.where(torch.rand_like(m.decoder.weight) > 0.5, 30.0, -30.0))
        run("config 2, compact outputs (idx, val, reconstruction; no dense latent)", m,
            lambda mm, xx: mm.forward_compact(xx), B, 2.0 * D * H + 2.0 * K_TOP * D,
            submit=lambda mm, xx, sl: mm.forward_submit(xx, slot=sl, want_dense=False))
        import warnings
        m = BinarySAE(D, H, gamma=GAMMA, n_bits=N_BITS)
        m.decoder.weight.normal_(0, 2.0)                   # not polarised: forward() = the reference's soft-integer arithmetic
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            run("config 2, unpolarised decoder (logit std 2): decode_mode auto -> the reference's soft sigmoid-bit integers "
                "(sae/binary.py:26-38) from an fp32 [H,512] table, decoded by the refinement kernel (qsae_table_forward_prefilter); "
                "dense latent + reconstruction", m, lambda mm, xx: mm(xx), B, 2.0 * D * H + 2.0 * K_TOP * D,
                submit=lambda mm, xx, sl: mm.forward_submit(xx, slot=sl))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=ROWS_PER_GPU, help="rows per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-rows", type=int, default=4096)
    ap.add_argument("--latent-path", default="auto", choices=["auto", "fused", "inplace", "prefilter"])
    ap.add_argument("--no-fp32-reference", action="store_true", help="skip the extra fp32-only measurement")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary configurations (rank 0, N=1 only)")
    ap.add_argument("--pipeline", type=int, default=2, choices=[1, 2],
                    help="batches in flight: 2 = submit step i+1 before finishing step i (qsae_prefilter_submit / _finish: "
                         "the 4-byte flagged-row read-back of a step no longer idles the GPU); 1 = blocking forward()")
    ap.add_argument("--batches", type=int, default=4, help="distinct input batches the timed loop rotates over")
    ap.add_argument("--sustained-seconds", type=float, default=2.5,
                    help="extra run of at least this many seconds of steps, reported under `sustained` (0 = skip; N=1 only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no ROCm device visible (there is no CPU path)")
    # QSAE_BENCH_BACKEND=gloo is a rehearsal switch for a one-GPU box (ranks share the card, the three scalar
    # reductions go over gloo); the driver's multi-GPU runs use the default, RCCL with one card per rank
    backend = os.environ.get("QSAE_BENCH_BACKEND", "nccl")
    device = torch.device("cuda", local_rank if backend == "nccl" else local_rank % torch.cuda.device_count())
    torch.cuda.set_device(device)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    from quantizedsae_amd import ops, sharding
    model = build_model(device)
    model.latent_path = args.latent_path
    B = args.rows
    g = torch.Generator(device=device)
    g.manual_seed(1000 + rank)                       # every rank owns a different row shard
    # the timed loop rotates over several distinct batches (step i takes batch i mod n): data-dependent work (survivors per
    # row, rows that need the exact fallback) is not one replayed outcome
    xs = [torch.randn((B, D), device=device, generator=g) for _ in range(max(1, args.batches))]
    x = xs[0]
    model.decoder.packed()                           # pack once, outside the timed region
    sq = torch.zeros((), dtype=torch.float64, device=device)

    def step(acc, i=0):
        xi = xs[i % len(xs)]
        latent, recon, _pol = model(xi)
        ops.sq_err_sum(recon, xi, acc)
        return latent, recon

    def run_steps(n_steps, acc, pipeline=None, marks=None):
        """n_steps full forwards + MSE accumulation.  pipeline 2: step i+1's kernels are queued before step i's
        flagged-row count is read (every step still completes inside the caller's timed region); pipeline 1: the
        blocking forward().  marks: optional list of n_steps + 1 timing events, one recorded per step on the launch
        stream."""
        pipeline = args.pipeline if pipeline is None else pipeline
        if marks:
            marks[0].record()
        if pipeline <= 1:
            for i in range(n_steps):
                step(acc, i)
                if marks:
                    marks[i + 1].record()
            return
        pending, xprev = None, None
        for i in range(n_steps):
            xi = xs[i % len(xs)]
            h = model.forward_submit(xi, slot=i % 2)
            if pending is not None:
                _lat, rec, _pol = pending.result()
                ops.sq_err_sum(rec, xprev, acc)
                if marks:
                    marks[i].record()
            pending, xprev = h, xi
        _lat, rec, _pol = pending.result()
        ops.sq_err_sum(rec, xprev, acc)
        if marks:
            marks[n_steps].record()

    def timed_region(n_steps, acc, pipeline=None):
        """barrier + sync, n_steps forwards, sync + barrier; returns elapsed seconds (max over ranks) and
        the live HIP-event timing of the dominant sweep kernel."""
        ops.kernel_timer.reset()
        ops.kernel_timer.enabled = True
        ops.sweep_timing(True)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        run_steps(n_steps, acc, pipeline)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        ops.kernel_timer.enabled = False
        ops.sweep_timing(False)
        sweep = ops.sweep_timing_collect(H)
        return sharding.max_over_ranks(dt, device=device), sweep

    if args.warmup > 0:
        run_steps(args.warmup, torch.zeros((), dtype=torch.float64, device=device))
    elapsed, (sweep_ms, sweep_n, sweep_frac) = timed_region(args.steps, sq)
    mse = sharding.reduce_mse(sq, args.steps * B * D)
    path_used = model.resolved_latent_path(B)

    # what drops in: the blocking call a user of the reference makes -- SAEWrapper.__call__(batch) -> outputs dict
    # (inference/framework.py:316-319) -- same steps, same batches, same MSE accumulation; reported beside `value`
    from quantizedsae_amd.inference.framework import SAE_REGISTRY, SAEWrapper
    wrapper = SAEWrapper(SAE_REGISTRY["b_sae"], model, device)

    def run_blocking(n_steps, acc):
        for i in range(n_steps):
            xi = xs[i % len(xs)]
            outs = wrapper(xi)
            ops.sq_err_sum(outs["reconstruction"], xi, acc)

    run_blocking(1, torch.zeros((), dtype=torch.float64, device=device))
    sq_b = torch.zeros((), dtype=torch.float64, device=device)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    run_blocking(args.steps, sq_b)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el_block = sharding.max_over_ranks(time.perf_counter() - t0, device=device)
    mse_block = sharding.reduce_mse(sq_b, args.steps * B * D)
    flagged_last = int(model.last_flagged_rows)

    # sustained: >= args.sustained_seconds of steps in one region, one event per step
    sustained = None
    if world == 1 and args.sustained_seconds > 0:
        n_sus = max(args.steps, int(args.sustained_seconds / (elapsed / args.steps)) + 1)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_sus + 1)]
        sq_s = torch.zeros((), dtype=torch.float64, device=device)
        clocks = []

        def sclk():
            try:
                import glob
                for path in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
                    for line in open(path):
                        if "*" in line:
                            return line.split(":")[1].replace("*", "").strip()
            except OSError:
                pass
            return None

        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run_steps(n_sus, sq_s, marks=marks)
        clocks.append(sclk())                                  # (GPU still busy: the queue is one batch deep)
        torch.cuda.synchronize()
        el_sus = time.perf_counter() - t0
        gaps = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(1, n_sus - 1))    # (first and last gaps are not whole steps)
        sustained = {"steps": n_sus, "seconds": el_sus, "value": B * n_sus / el_sus, "unit": "activations/s",
                     "ms_per_step_mean": el_sus / n_sus * 1e3, "ms_step_min": gaps[0], "ms_step_median": gaps[len(gaps) // 2],
                     "ms_step_p99": gaps[min(len(gaps) - 1, int(len(gaps) * 0.99))], "ms_step_max": gaps[-1],
                     "step_timing": "distance between per-step HIP events on the launch stream",
                     "sclk_while_busy": clocks[0],
                     "recon_mse": sharding.reduce_mse(sq_s, n_sus * B * D)}

    # the exact-fp32-only path, same model and batch, measured separately (not part of `value`)
    fp32_ref = None
    if path_used == "prefilter" and not args.no_fp32_reference:
        model.latent_path = "fused"
        n_ref = max(3, min(args.steps, 5))
        step(torch.zeros((), dtype=torch.float64, device=device))
        sq_ref = torch.zeros((), dtype=torch.float64, device=device)
        el_ref, (ms_ref, n_sw, fr_ref) = timed_region(n_ref, sq_ref, pipeline=1)
        mse_ref = sharding.reduce_mse(sq_ref, n_ref * B * D)
        model.latent_path = args.latent_path
        ach = fr_ref * FLOPS_PER_ROW_ENCODER * B / (ms_ref * 1e-3) / 1e12 if n_sw else None
        fp32_ref = {"value": world * B * n_ref / el_ref, "unit": "activations/s", "ms_per_step": el_ref / n_ref * 1e3,
                    "steps": n_ref, "recon_mse": mse_ref,
                    "note": "latent_path='fused': every latent computed by the exact-fp32 MFMA sweep (no fp16 pass); "
                            "outputs bit-identical to the default path",
                    "roofline": {"bound": "mfma", "kernel": "gemm_nt_f32_dma_kernel<EpiFilter<256,128,4,2>> (fp32 MFMA sweep)",
                                 "achieved": ach, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                                 "frac": (ach / PEAK_FP32_MFMA_TFLOPS) if ach else None, "avg_kernel_ms": ms_ref,
                                 "traffic": pmc_traffic("sweep_dma_fp32") if B == ROWS_PER_GPU else None}}

    if rank == 0:
        total_rows = world * B * args.steps
        value = total_rows / elapsed
        if path_used == "prefilter" and sweep_n:
            enc_ms, peak = sweep_ms, PEAK_FP16_MFMA_TFLOPS
            achieved = sweep_frac * FLOPS_PER_ROW_ENCODER * B / (enc_ms * 1e-3) / 1e12
            kname = (f"sweep_xstat_f16_kernel<32> (fp16 MFMA candidate sweep {B}x512 @ 512x{int(round(H * sweep_frac))}: "
                     "activation panel stationary in registers, weights streamed once per workgroup, in-kernel pilot pass and "
                     "threshold, threshold filter in the MFMA shadows) together with fill_zero_co_kernel, which writes the zeros of "
                     "the dense [B,32768] latent from the 16 VGPRs per SIMD the sweep leaves free (second stream, joined before "
                     "the refinement); avg_kernel_ms is the HIP-event time of the pair")
            tkey = "sweep_xstat_f16+fill_zero_co"
        elif sweep_n:
            enc_ms, peak = sweep_ms, PEAK_FP32_MFMA_TFLOPS
            achieved = sweep_frac * FLOPS_PER_ROW_ENCODER * B / (enc_ms * 1e-3) / 1e12
            kname = (f"gemm_nt_f32_dma_kernel<EpiFilter<256,128,4,2>> (fp32 MFMA encoder sweep {B}x512 @ "
                     f"512x{int(round(H * sweep_frac))} with threshold-filter epilogue)")
            tkey = "sweep_dma_fp32"
        else:
            enc_ms, peak = ops.kernel_timer.mean_ms("encode_dense"), PEAK_FP32_MFMA_TFLOPS
            achieved = FLOPS_PER_ROW_ENCODER * B / (enc_ms * 1e-3) / 1e12 if enc_ms else None
            kname = f"gemm_nt_f32_kernel<..., EpiDense> (encoder {B}x512 @ 512x{H}, fp32 MFMA)"
            tkey = None
        out = {
            "metric": "activations/sec, BinarySAE 512->32768 n_bits=4 forward (dense latent + reconstruction + MSE)",
            "value": value,
            "unit": "activations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BinarySAE(512, 32768, gamma=4.0, n_bits=4) k=65, {B} rows per GPU per step, "
                                   "reference forward() signature (dense [B,32768] latent, [B,512] reconstruction, "
                                   "polarize loss) + recon-MSE accumulation",
                       "rows_per_gpu": B, "input_dim": D, "hidden_dim": H, "n_bits": N_BITS, "gamma": GAMMA,
                       "top_k": K_TOP, "parallelism": f"row-sharded x{world}, no data-path collective",
                       "latent_path": path_used, "batches_in_flight": args.pipeline,
                       "precision_note": ("every returned value (latent, reconstruction, MSE) is exact fp32 and bit-identical "
                                          "to the fp32-only path; with latent_path='prefilter' an fp16 MFMA pass with a rigorous "
                                          "per-row error bound only selects ~70 candidate hidden units per row, which are then "
                                          "re-evaluated with the exact fp32 fmaf chain and ranked exactly (DESIGN.md 7); at this "
                                          "batch size the re-evaluation runs slice-major over the hidden units (select / chains "
                                          "with a 4-MiB slice of the encoder weights resident in each XCD's L2 / rank + row "
                                          "decode, DESIGN.md 7.6)")},
            "recon_mse": mse,
            "forward_blocking": {"value": world * B * args.steps / el_block, "unit": "activations/s",
                                 "ms_per_step": el_block / args.steps * 1e3, "steps": args.steps, "recon_mse": mse_block,
                                 "what": "the same steps through the call a user of the reference makes: SAEWrapper.__call__(batch) "
                                         "-> outputs dict (inference/framework.py:316-319), i.e. BinarySAE.forward() with its one "
                                         "host round trip per batch (the flagged-row count); one batch in flight",
                                 "flagged_rows_last_batch": flagged_last},
            "sustained": sustained,
            "input_batches": len(xs),
            "whole_path_tflops_per_gpu": value / world * FLOPS_PER_ROW / 1e12,
            "whole_path_vs_fp32_mfma_roofline": value / world * FLOPS_PER_ROW / 1e12 / PEAK_FP32_MFMA_TFLOPS,
            "whole_path_dense_gbps_per_gpu": value / world * BYTES_PER_ROW_DENSE / 1e9,
            "roofline": {"bound": "mfma", "kernel": kname,
                         "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": (achieved / peak) if achieved else None,
                         "avg_kernel_ms": enc_ms,
                         "traffic": pmc_traffic(tkey) if (tkey and B == ROWS_PER_GPU) else None},
            "fp32_only_path": fp32_ref,
        }
        # the whole step against the HBM roofline: SURVEY 8(d) bytes per row (x in, dense latent + reconstruction out) plus
        # the weights once per batch, over the step time the driver also sees
        step_bytes = BYTES_PER_ROW_DENSE * B + WEIGHT_BYTES_PER_BATCH
        step_ms = elapsed / args.steps * 1e3
        out["roofline"]["whole_step"] = {
            "bound": "hbm", "algorithmic_bytes_per_step": step_bytes, "ms_per_step": step_ms,
            "achieved": step_bytes / (step_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
            "frac": step_bytes / (step_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS,
            "frac_of_achievable_6300": step_bytes / (step_ms * 1e-3) / 1e9 / 6300.0}
        out["roofline"]["traffic_source"] = TRAFFIC_SOURCE
        if path_used == "prefilter" and sweep_n and enc_ms:
            # The sweep / fill pair carries two resources: 2.06 PFLOP of fp16 MFMA (0.82 ms at peak) and the 8.6 GB of
            # zeros of the dense latent (1.07 ms at 8 TB/s).  The second one binds, so it is the roofline quoted first;
            # the MFMA view of the same interval follows under "also".
            hswept = int(round(H * sweep_frac))
            algo_bytes = 2 * B * D + 2 * hswept * D + 4 * B * H + 8 * 320 * B      # x~, W~ (fp16), zeros, ~320 records/row
            gbps = algo_bytes / (enc_ms * 1e-3) / 1e9
            mfma_view = {"bound": "mfma", "what": "fp16 MFMA work of the same launch", "achieved": achieved,
                         "peak": peak, "unit": "TFLOP/s", "frac": (achieved / peak) if achieved else None,
                         # static context, not measured by this run: the clock the chip holds under this kernel and the share of
                         # its cycles the matrix pipes are busy (rocprofv3 --pmc GRBM_GUI_ACTIVE / SQ_VALU_MFMA_BUSY_CYCLES)
                         "held_clock_ghz_static": 1.71, "mfma_busy_frac_at_held_clock_static": 0.63,
                         "static_source": "profiles/r03_headline_pmc.txt (tools/r03_headline_pmc.sh)"}
            out["roofline"].update({"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                                    "frac": gbps / PEAK_HBM_GBPS, "algorithmic_bytes_per_launch": algo_bytes,
                                    "also": mfma_view})
        if world == 1 and not args.no_secondary:
            out["secondary_configs"] = secondary_configs(device, x)
        if world == 1 and not args.no_cpu_baseline:
            threads = os.cpu_count() or 1
            try:
                threads = len(os.sched_getaffinity(0))
            except AttributeError:
                pass
            out["cpu_baseline"] = cpu_baseline(model, x[: args.cpu_sample_rows], threads)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
