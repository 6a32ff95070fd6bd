#!/usr/bin/env python3
"""Headline benchmark: 1-second clips per second through the DFD-CLIP hot path on MI355X.

One step = one pass of the hot path over one batch of synthetic clips that is already
resident in HBM, x [16, 30, 3, 224, 224] per GPU.  BASELINE.json's metric is "clips/sec ...
fwd+bwd", so the timed step is the TRAIN step of the reference (src/trainer.py:108-177):
frozen-encoder forward + decoder forward + decoder backward + SGD(momentum 0.95) step
(BASELINE configs[2] at N = 1; the encoder has no backward in the reference).  `--mode infer`
times configs[1] instead (forward-only `Detector.predict`, the inference.py path); the default
run reports it as `forward_only` next to the headline.  With --gpus N the driver launches one
rank per GPU (torch.distributed, backend nccl = RCCL); clips are independent, so ranks shard
them: the only collectives are the gradient all-reduce of the decoder parameters (train) and
the all-gather of per-clip logits (infer); scaling is weak (16 clips per GPU).

Prints ONE JSON line on rank 0 with the whole-job clips/s plus
  roofline     — the dominant kernel (the MLP c_fc GEMM, M x 3072 x 768 with QuickGELU epilogue):
                 algorithmic FLOPs per launch / its average launch duration measured with HIP
                 events on the launch stream during the timed steps, against the 2.5 PFLOP/s
                 dense bf16 MFMA peak;
                 (with HIP graphs on, the encoder's pass is one graph launch; the last three timed steps
                 launch its kernels one by one so that these launches can be bracketed — same kernels,
                 same stream, inside the timed region);
  cpu_baseline — the CPU oracle (this repo's PyTorch-CPU port of the reference path) timed on
                 this host on a bounded sample (one 30-frame clip), rank 0, N=1 only.
Host figures: `host_enqueue_ms_each_step` is the wall time the host spent inside each timed step's calls,
`host_enqueue_ms_median` their median, `host_enqueue_ms_per_step` their mean over the steps in which the host did not
block, `host_queue_stall_ms_total` the time of those in which it did, and `host_cpu_ms_per_step` the CPU seconds of all
threads over the region (the runtime's spinning included).  The stall: one of the kernel-by-kernel steps blocks for
70-160 ms in most runs, wherever they sit in the region and whether there are two or three of them (a burst of a few
hundred launches and event records on a stream the host is many steps ahead on; the runtime waits for the stream to
drain).  At the END of the region that costs nothing — the GPU has the remaining
work queued — which is why the profiled steps are the last three (placed third to fifth they cost 5-18 % of the
measured rate; keeping the host four steps behind with a blocking-sync event per step was worse still: 864-873 ->
721-779 clips/s).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _launcher():
    """dfd-clip_amd/launch.py loaded by path: the parent of a multi-rank run imports neither torch nor the HIP library."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_dfd_launch", os.path.join(ROOT, "dfd-clip_amd", "launch.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__" and "--gpus" in " ".join(sys.argv[1:]):
    # `python bench.py --gpus N` with N > 1 and no rendezvous in the environment: this process becomes the launcher
    # (the reference's `accelerate launch`, scripts/cross-manipulation-train.sh:6) BEFORE anything can touch the GPU
    _ap = argparse.ArgumentParser(add_help=False)
    _ap.add_argument("--gpus", type=int, default=1)
    _n = _ap.parse_known_args()[0].gpus
    _l = _launcher()
    if _n > 1 and not _l.under_launcher():
        sys.exit(_l.spawn_ranks(_n, [os.path.abspath(__file__)] + sys.argv[1:]))

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0  # dense, /opt/skills/guides/MI355X_MICROARCH.md chip table
PEAK_F32_TFLOPS = 157.3
PEAK_FP8_TFLOPS = 5000.0   # dense, block-scaled e4m3 (same table)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arch", default="ViT-B/16")
    ap.add_argument("--clips", type=int, default=16, help="clips per GPU per step")
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp8"],
                    help="fp8 = BASELINE configs[4]: e4m3 operands for the encoder's q|k|v, c_fc and c_proj GEMMs (use with --arch ViT-L/14)")
    ap.add_argument("--frame-chunk", type=int, default=-1, help="frames per encoder pass (-1 = package default)")
    ap.add_argument("--streams", type=int, default=-1, help="HIP streams for independent frame chunks (-1 = package default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kv-export", action="store_true",
                    help="A/B: let the projection's epilogue export K/V + positional embedding (round 1's hand-over) instead of the "
                         "decoder reading them in place")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the informational ViT-L/14 lines (BASELINE configs[3] bf16 / configs[4] fp8) of the default single-GPU run")
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI; the default) or gloo (rehearsal on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--no-graphs", action="store_true",
                    help="launch the decoder's training kernels one by one instead of replaying them as HIP graphs")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="run the frozen encoder on the same stream as the decoder (no overlap of step N+1's encoder pass "
                         "with step N's backward / all-reduce / optimizer)")
    ap.add_argument("--ingest", default="f32", choices=["f32", "u8"],
                    help="f32: frames already transformed (the headline's input); u8: raw uint8 frames through the ingest kernel")
    ap.add_argument("--ingest-size", type=int, nargs=2, default=None, metavar=("H", "W"),
                    help="with --ingest u8: source frame size (default = the model's resolution, i.e. no resize)")
    ap.add_argument("--spare-cus", type=int, default=-1, help="CUs the encoder GEMMs leave to the decoder stream in pipelined training (-1 = package default)")
    ap.add_argument("--spare-layers", type=int, default=-1, help="encoder blocks at the start of a pass whose GEMMs (--spare-gemms) leave the spare CUs free (-1 = package default, 0 = all)")
    ap.add_argument("--collective-layers", type=int, default=6,
                    help="world size > 1: encoder blocks at the start of a pass in which every GEMM leaves the spare CUs to the RCCL all-reduce")
    ap.add_argument("--spare-gemms", default=None, help="lab: comma list of the block's GEMMs (qkv,out,fc,proj) that leave the spare CUs free")
    ap.add_argument("--gemm-stream-out", default=None,
                    help="comma list of encoder GEMM outputs stored non-temporally (qkv,out,fc,proj; 'none'); default = package default")
    ap.add_argument("--no-spare-in-eval", action="store_true", help="A/B: in forward-only mode no GEMM leaves spare CUs to the decoder stream (round 2's behaviour)")
    ap.add_argument("--gemm-dynamic", action="store_true",
                    help="A/B: the persistent GEMM hands out the tiles after a workgroup's first from per-XCD counters instead of dealing them statically")
    ap.add_argument("--adapter", default="none", choices=["none", "nln", "z0", "ln"],
                    help="CompInvAdapter 768-x-768-<struct>, x = 256 (every configs/deepfake/*.yaml enables one); default none = headline")
    return ap.parse_args()


def build_model(args, device):
    from dfd_clip_amd.detector import Detector
    from dfd_clip_amd.weights import ARCHS, random_state_dict
    cfg = Detector.get_default_config()
    cfg.architecture = args.arch
    cfg.out_dim = [2]
    cfg.losses = ["auc_roc"]
    layers = ARCHS[args.arch][3]
    if args.arch == "ViT-B/16":
        cfg.decode_mode, cfg.decode_indices = "index", [6, 7, 8, 9, 10, 11]  # every configs/deepfake/*.yaml
    else:
        cfg.decode_mode, cfg.decode_stride = "stride", 2
    if args.adapter != "none":
        from dfd_clip_amd.config import ConfigNode
        cfg.adapter = ConfigNode({"type": "normal", "frozen": 0, "struct": {"type": f"768-x-768-{args.adapter}", "x": 256}})
    sd = random_state_dict(cfg, args.frames, seed=0)
    det = Detector(cfg, args.frames, None, precision=args.precision)
    det.load_state_dict(sd)
    det = det.to(device)
    if args.frame_chunk >= 0:
        det.encoder.frame_chunk = args.frame_chunk
    if args.streams >= 0:
        det.encoder.streams = args.streams
    if args.kv_export:
        det.kv_in_place = False
    if args.spare_cus >= 0:
        det.pipeline_spare_cus = args.spare_cus
    if args.spare_layers >= 0:
        det.pipeline_spare_layers = args.spare_layers
    det.pipeline_spare_in_eval = not args.no_spare_in_eval
    if args.spare_gemms is not None:
        on = set(args.spare_gemms.split(",")) - {"none", ""}
        det.encoder.spare_gemms = {k: k in on for k in det.encoder.spare_gemms}
    if args.gemm_stream_out is not None:
        on = set(args.gemm_stream_out.split(",")) - {"none", ""}
        det.encoder.stream_out = {k: k in on for k in det.encoder.stream_out}
    return det, cfg, sd, layers


def secondary_configs(args, device):
    """Informational, outside the headline's timed region, single GPU only: the forward-only rate of the other two
    encoder configurations BASELINE.json names — configs[3] ViT-L/14 bf16 at 8 clips x 30 frames, and configs[4] the
    same model with e4m3 operands (static scales calibrated on the synthetic batch) at 8 and at 16 clips x 30.
    Each: 5 warm-up passes, then the better of two windows of 8 passes; pipelined like the headline's forward-only leg."""
    import copy
    out = []
    for prec, clips in (("bf16", 8), ("fp8", 8), ("fp8", 16)):
        a = copy.copy(args)
        a.arch, a.precision, a.clips, a.adapter = "ViT-L/14", prec, clips, "none"
        det, _, _, _ = build_model(a, device)
        det.eval()
        # as the headline's forward-only leg: batches follow each other, so the encoder's pass of one overlaps the decoder
        # of the previous one, and recurring input buffers replay as graphs
        det.static_graphs, det.pipeline_encoder, det.inputs_ready = not args.no_graphs, not args.no_pipeline, True
        g = torch.Generator(device=device).manual_seed(99)
        x = torch.randn(clips, args.frames, 3, 224, 224, device=device, generator=g)
        m = torch.ones(clips, args.frames, dtype=torch.bool, device=device)
        with torch.no_grad():
            if prec == "fp8":
                det.calibrate_fp8(x[:2])
            for _ in range(5):
                det.predict(x, m)
            n, dt = 8, float("inf")
            for _ in range(2):  # the better of two windows: a fresh model's first steps can hit a one-off allocator stall
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(n):
                    det.predict(x, m)
                torch.cuda.synchronize()
                dt = min(dt, (time.perf_counter() - t0) / n)
        out.append({"workload": f"BASELINE configs[{3 if prec == 'bf16' else 4}]: ViT-L/14 forward-only Detector.predict, {clips} clips x "
                                f"{args.frames} frames, {prec}", "value": round(clips / dt, 2), "unit": "clips/s", "ms_per_step": round(dt * 1e3, 2),
                    "dtype": prec})
        del det, x, m
        torch.cuda.empty_cache()
    out.append(adapter_train_line(args, device))
    return out


def adapter_train_line(args, device):
    """Informational: the train step of the configuration every shipped configs/deepfake/*.yaml uses — ViT-B/16 with a
    trainable CompInvAdapter 768-x-768-nln, x = 256 (reference src/models.py:783-940) — same batch, graphs and
    pipelining as the headline: 3 warm-up steps, then the better of two windows of 6 steps."""
    import copy
    a = copy.copy(args)
    a.arch, a.precision, a.adapter = "ViT-B/16", "bf16", "nln"
    det, _, _, _ = build_model(a, device)
    det.train()
    det.static_graphs, det.pipeline_encoder, det.inputs_ready = not args.no_graphs, not args.no_pipeline, True
    B, T = args.clips, args.frames
    g = torch.Generator(device=device).manual_seed(77)
    x = torch.randn(B, T, 3, 224, 224, device=device, generator=g)
    m = torch.ones(B, T, dtype=torch.bool, device=device)
    y = torch.arange(B, device=device) % 2
    opt = det.configure_optimizers(0.01 / 25)

    def step():
        det.zero_grad(set_to_none=True)
        losses, _, other = det(x, [y], m, train=True, single_task=0)
        (losses[0].mean() + sum(other.values())).backward()
        opt.step()

    for _ in range(3):
        step()
    n, best = 6, None
    for _ in range(2):
        torch.cuda.synchronize()
        t0, c0 = time.perf_counter(), time.process_time()
        for _ in range(n):
            step()
        enq, cpu = time.perf_counter() - t0, time.process_time() - c0
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        if best is None or dt < best[0]:
            best = (dt, enq / n, cpu / n)
    failed = det.decoder._graphs_failed or det.adapter._graphs_failed
    line = {"workload": f"ViT-B/16 + CompInvAdapter 768-x-768-nln (x = 256) train step, {B} clips x {T} frames, bf16", "value": round(B / best[0], 2),
            "unit": "clips/s", "ms_per_step": round(best[0] * 1e3, 2), "host_enqueue_ms_per_step": round(best[1] * 1e3, 2),
            "host_cpu_ms_per_step": round(best[2] * 1e3, 2), "dtype": "bf16", "hip_graphs": bool(det.static_graphs) and not failed}
    del det, x, m, opt
    torch.cuda.empty_cache()
    return line


def cpu_baseline(cfg, sd, args):
    """The oracle as the CPU baseline ("port"): one T-frame clip, fp32, no_grad, all host threads."""
    from oracle import ref_cpu
    from dfd_clip_amd.weights import ARCHS, resolve_layer_indices, synthetic_clips
    res, patch, width, layers, heads, _ = ARCHS[args.arch]
    x, m, _ = synthetic_clips(1, args.frames, res, seed=1234, masked_tail=False)
    kw = dict(heads=heads, patch=patch, layer_indices=resolve_layer_indices(cfg, layers), out_dims=[2],
              num_frames=args.frames)
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = min(cores, 16)  # the GPU box gives one GPU's job a 16-core share; more threads only oversubscribe
    torch.set_num_threads(cores)
    times = []
    with torch.no_grad():
        for i in range(3):
            t0 = time.perf_counter()
            ref_cpu.detector_predict(sd, x, m, **kw)
            times.append(time.perf_counter() - t0)
    t = sorted(times[1:])[0] if len(times) > 1 else times[0]
    return {"value": round(1.0 / t, 4), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"1 clip x {args.frames} frames {args.arch} fp32, oracle/ref_cpu.detector_predict, best of 2 after 1 warm-up, "
                      f"torch {torch.__version__} CPU threads={cores}"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the environment describes {world} rank(s) (WORLD_SIZE): refusing to report a line whose n_gpus is wrong")
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the all-reduce shares the chip with persistent GEMMs that leave it 32 CUs: one workgroup per channel
        os.environ.setdefault("NCCL_MAX_NCHANNELS", "32")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(args.dist_backend)
    ranks_seen = 1
    if dist is not None:  # the world size the COLLECTIVE layer sees, not the one the environment claims
        one = torch.ones(1, device=device if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(one)
        ranks_seen = int(one.item())

    from dfd_clip_amd import capi
    from dfd_clip_amd.weights import ARCHS
    if args.gemm_dynamic:
        capi.gemm_set_variant(3)
    det, cfg, sd, layers = build_model(args, device)
    res, patch, width, _, heads, _ = ARCHS[args.arch]
    B, T = args.clips, args.frames
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    if args.ingest == "u8":
        ih, iw = args.ingest_size or (res, res)
        x = torch.randint(0, 256, (B, T, 3, ih, iw), device=device, generator=g, dtype=torch.uint8)
    else:
        x = torch.randn(B, T, 3, res, res, device=device, generator=g)
    m = torch.ones(B, T, dtype=torch.bool, device=device)
    tokens = (res // patch) ** 2 + 1
    M = B * T * tokens

    y = (torch.arange(B, device=device) % 2)
    from dfd_clip_amd import dist as ddist
    ddist.broadcast_parameters(det)
    # the ~450 small decoder launches of a train step replay as two HIP graphs (forward / backward kernels): the host then
    # enqueues a step in a few ms instead of ~14, which keeps the step GPU-bound on a busy host.  Round 1 saw
    # multi-second steps with graphs + pipelining in a two-ranks-on-one-GPU gloo rehearsal and switched graphs off at
    # world > 1; round 2 re-ran that rehearsal five times in three variants without a single slow step and with the
    # device draining every queued step in 22-27 ms (host-side, box-dependent, not a stream/event cycle),
    # so graphs are on everywhere — guarded: the warm-up below times a few steps and every rank falls back to eager
    # launches together if graphs make a step pathologically slow on this machine.
    det.static_graphs = not args.no_graphs
    # the frozen encoder runs on its own stream: step N+1's encoder pass overlaps step N's decoder backward,
    # gradient all-reduce and optimizer step (the inputs are resident before the timed region: inputs_ready)
    det.pipeline_encoder = not args.no_pipeline
    det.inputs_ready = True
    if world > 1:
        # the gradient all-reduce of step N runs beside the first blocks of step N+1's encoder pass (decoder forward +
        # backward take ~3 ms of the ~18 ms pass, the collective 1-3 ms after that): every GEMM of those blocks leaves
        # the spare CUs to it (expected cost: one extra round of q|k|v and c_fc tiles per block, ~0.4 ms per step)
        det.pipeline_collective_layers = args.collective_layers
    opt = det.configure_optimizers(0.01 / 25)
    trainable = [p for p in det.parameters() if p.requires_grad]

    @torch.no_grad()
    def infer_step():
        logits, _ = det.predict(x, m)
        if dist is not None:  # evaluation contract: gather per-clip logits (reference callbacks/metrics.py:98-99)
            ddist.gather_for_metrics(logits[0])
        return logits

    def train_step():
        det.zero_grad(set_to_none=True)
        task_losses, _, other = det(x, [y], m, train=True, single_task=0)
        (task_losses[0].mean() + sum(other.values())).backward()
        ddist.allreduce_gradients(trainable)  # RCCL over xGMI: one flat all-reduce of the decoder gradients
        opt.step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    host_ms = [0.0, 0.0]
    host_steps = []  # host milliseconds of each timed step's enqueue (the last ones launch the encoder kernel by kernel)

    PRIME_STEPS = 4

    def timed(fn, steps, profile):
        # dominant kernel: c_fc GEMM (M x 4D x D, QuickGELU epilogue), HIP events around its launches on the launch stream.
        # When the encoder's pass replays as one HIP graph there are no individual launches to bracket, so the LAST
        # `prof_steps` steps of the timed region launch the encoder's kernels one by one and those launches are the ones
        # timed (same kernels, same stream, inside the timed region; the steps are GPU-bound either way).
        prof_steps = min(3, steps) if profile and det.static_graphs else (steps if profile else 0)
        barrier()
        t0 = time.perf_counter()
        c0 = time.process_time()
        per_step = []
        for i in range(steps):
            if prof_steps and i == steps - prof_steps:
                det.encoder_graph_pause = True
                capi.profile_gemm(epilogue=capi.EPI_BIAS_QUICKGELU)
            ts = time.perf_counter()
            fn()
            per_step.append((time.perf_counter() - ts) * 1e3)
        det.encoder_graph_pause = False
        host_steps[:] = per_step
        enq = time.perf_counter() - t0  # the host has enqueued everything (no sync inside a step)
        cpu = time.process_time() - c0  # CPU seconds of all threads of this process spent doing so
        barrier()
        dt = time.perf_counter() - t0
        host_ms[0] = enq / steps * 1e3
        host_ms[1] = cpu / steps * 1e3
        spans = capi.profile_gemm_collect() if profile else []
        if dist is not None:
            tt = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = tt.item()
        return dt, spans

    if args.precision == "fp8":  # static activation scales from the synthetic batch itself, before anything is timed
        det.calibrate_fp8(x[:2])
    det.train(args.mode == "train")
    step = train_step if args.mode == "train" else infer_step
    # Setup, before the W warm-up steps the caller asked for: HIP graphs are captured the second time a signature is seen and
    # the pipelined path alternates between two K/V sets, so four steps see every capture through (like the lazy
    # initialisations of the first step, it is one-off work that does not belong to a step's time).
    for _ in range(PRIME_STEPS):
        step()
    for _ in range(args.warmup):
        step()
    graph_note = None
    if world > 1 and det.static_graphs and args.mode == "train":
        # guard (see above): two more untimed steps, host-synchronised; > 10x the single-GPU step time on any rank
        # switches every rank to eager decoder launches
        barrier()
        t0 = time.perf_counter()
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        slow = torch.tensor([(time.perf_counter() - t0) / 2], device=device, dtype=torch.float64)
        dist.all_reduce(slow, op=dist.ReduceOp.MAX)
        if slow.item() > 0.25:
            det.static_graphs = False
            graph_note = f"HIP graphs switched off: {slow.item() * 1e3:.0f} ms/step with graphs at world {world}"
            for _ in range(2):
                step()
    dt, spans = timed(step, args.steps, True)
    host_enqueue_ms, host_cpu_ms = host_ms
    host_step_ms = [round(v, 2) for v in host_steps]
    # a step in which the host blocked on the runtime's queue depth (it runs many graph-replay steps ahead of the GPU) is not
    # host work: such steps (> 10 x the median and > 20 ms) are reported as `host_queue_stall_ms_total` and left out of the mean
    med = sorted(host_steps)[len(host_steps) // 2] if host_steps else 0.0
    stalled = [v for v in host_steps if v > max(10 * med, 20.0)]
    worked = [v for v in host_steps if v <= max(10 * med, 20.0)]
    if worked:
        host_enqueue_ms = sum(worked) / len(worked)
    host_stall_ms = sum(stalled)
    fwd_only = None
    if args.mode == "train":  # informational: BASELINE configs[1], forward-only, outside the headline's timed region
        det.eval()
        for _ in range(PRIME_STEPS):
            infer_step()
        dt_i, _ = timed(infer_step, max(2, args.steps // 2), False)
        fwd_only = world * B * max(2, args.steps // 2) / dt_i

    if rank == 0:
        # roofline of the dominant kernel: algorithmic FLOPs of its launches / the time the kernel was running.
        # Frame chunks run on two streams, so launches can overlap each other and other kernels: the
        # denominator is the UNION of the launches' HIP-event intervals (a lower bound on the kernel's rate).
        busy, cur_s, cur_e = 0.0, None, None
        for st_ms, en_ms, _ in sorted(spans):
            if cur_e is None or st_ms > cur_e:
                busy += (cur_e - cur_s) if cur_e is not None else 0.0
                cur_s, cur_e = st_ms, en_ms
            else:
                cur_e = max(cur_e, en_ms)
        busy += (cur_e - cur_s) if cur_e is not None else 0.0
        avg_ms = busy / max(1, len(spans))
        achieved = sum(fl for _, _, fl in spans) / (busy * 1e-3) / 1e12 if spans else None
        launch_m = int(round(spans[0][2] / (2.0 * 4 * width * width))) if spans else M
        peak = {"bf16": PEAK_BF16_TFLOPS, "fp32": PEAK_F32_TFLOPS, "fp8": PEAK_FP8_TFLOPS}[args.precision]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "dominant_kernel_traffic.json")
        if os.path.exists(tpath):  # PMC-measured HBM bytes per launch; only valid for the shape it was measured on
            tj = json.load(open(tpath))
            if tj.get("shape") == {"M": launch_m, "N": 4 * width, "K": width}:
                traffic = tj.get("hbm_bytes_per_launch")
        line = {
            "metric": f"1-sec clips/sec (30x224x224 frames) {args.arch}", "value": round(world * B * args.steps / dt, 3),
            "unit": "clips/s", "n_gpus": world, "ranks_seen": ranks_seen, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "host_enqueue_ms_per_step": round(host_enqueue_ms, 3), "host_cpu_ms_per_step": round(host_cpu_ms, 3), "host_enqueue_ms_median": round(med, 3), "host_queue_stall_ms_total": round(host_stall_ms, 1), "host_enqueue_ms_each_step": host_step_ms,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.precision, "data": "synthetic" if args.ingest == "f32" else "synthetic uint8 frames %dx%d" % tuple(x.shape[-2:]),
            "config": {"workload": (f"BASELINE configs[2] (fwd+bwd): {args.arch} train step = frozen-encoder forward + decoder "
                                    f"forward/backward + SGD step" + (" + RCCL gradient all-reduce" if world > 1 else "")
                                    if args.mode == "train" else
                                    f"BASELINE configs[1]: {args.arch} forward-only Detector.predict (inference.py path)")
                                   + f", {B} clips x {T} frames x 3x{res}x{res} per GPU, decode layers {det.layer_indices}, "
                                     f"random-init weights, inputs resident in HBM",
                       "mode": args.mode, "adapter": args.adapter, "clips_per_gpu": B, "frames_per_clip": T, "hip_graphs": bool(det.static_graphs) and not det.decoder._graphs_failed and not (det.adapter is not None and det.adapter._graphs_failed), "encoder_graph": bool(det._enc_graphs) and not det._enc_graphs_failed, "pipelined_encoder": bool(det.pipeline_encoder), "frame_chunk": det.encoder.frame_chunk,
                       "streams": det.encoder.streams},
            "roofline": {"bound": "mfma", "kernel": "c_fc GEMM + QuickGELU (M=%d, N=%d, K=%d)" % (launch_m, 4 * width, width),
                         "achieved": round(achieved, 2) if achieved else None, "peak": peak, "unit": "TFLOP/s",
                         "frac": round(achieved / peak, 4) if achieved else None, "traffic": traffic,
                         "launches_timed": len(spans), "avg_launch_ms": round(avg_ms, 4)},
        }
        if det.decoder._graphs_failed and not graph_note:
            graph_note = "HIP graph capture refused by the runtime, decoder on eager launches: " + det.decoder._graphs_failed[:160]
        if graph_note:
            line["config"]["note"] = graph_note
        if fwd_only is not None:
            line["forward_only"] = {"value": round(fwd_only, 3), "unit": "clips/s", "workload": "BASELINE configs[1]: Detector.predict"}
        if world == 1 and not args.no_secondary and args.arch == "ViT-B/16" and args.precision == "bf16" and args.mode == "train":
            try:  # informational: a failure here must not cost the headline line
                line["secondary"] = secondary_configs(args, device)
            except Exception as e:  # noqa: BLE001
                line["secondary"] = [{"error": f"{type(e).__name__}: {e}"[:300]}]
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(cfg, sd, args)
            except Exception as e:  # noqa: BLE001
                line["cpu_baseline"] = {"value": None, "unit": "clips/s", "cores": 0, "kind": "port", "sample": f"failed: {type(e).__name__}: {e}"[:300]}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
