"""Lab diagnostic for the multi-process stall of graphs + pipelined encoder + a collective (round 1:
gpurun_out/r2d.log 1,176 ms/step, dbg.log all-reduce 8,004 ms; two gloo ranks sharing ONE GPU).

Splits a train step's host time into forward / backward / device drain / all-reduce / optimizer, and reports
how many KFD user queues each process holds (/sys/class/kfd/kfd/proc/<pid>/queues) — the hypothesis being
hardware-queue oversubscription when two processes that each own normal + high-priority + graph + gloo
copy streams share one GPU (one process per GPU on a real node never does).
Knobs (env): G=0/1 decoder HIP graphs, P=0/1 pipelined encoder, S=0/1 drain the device before the collective
(separates "GPU work is slow" from "the collective is slow"), STAGE=0/1 whether the gradient all-reduce stages
through this repo's persistent pinned buffer (1) or hands the CUDA tensor to ProcessGroupGloo (0), PRIO=0 puts the encoder stream at normal priority;
GPU_MAX_HW_QUEUES is ROCclr's own knob (HSA queues per priority level per process).
Run: torchrun --nproc-per-node 2 tools/lab/dbg_pipe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
from dfd_clip_amd.config import default_detector_config
from dfd_clip_amd.detector import Detector
from dfd_clip_amd.weights import random_state_dict
from dfd_clip_amd import dist as ddist


def kfd_queues():
    d = f"/sys/class/kfd/kfd/proc/{os.getpid()}/queues"
    try:
        return len(os.listdir(d))
    except OSError as e:
        return f"n/a ({e.__class__.__name__})"


torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank = dist.get_rank()
cfg = default_detector_config(); cfg.architecture = "ViT-B/16"; cfg.decode_mode = "index"; cfg.decode_indices = [6,7,8,9,10,11]; cfg.out_dim=[2]; cfg.losses=["auc_roc"]
T, B = 30, 8
det = Detector(cfg, T, None, precision="bf16"); det.load_state_dict(random_state_dict(cfg, T, seed=0)); det = det.cuda().train()
det.static_graphs = os.environ.get("G", "1") == "1"; det.pipeline_encoder = os.environ.get("P", "1") == "1"; det.inputs_ready = True
if os.environ.get("PRIO", "1") == "0":
    det._enc_stream = torch.cuda.Stream()
drain = os.environ.get("S", "0") == "1"
ddist._STAGE_GLOO = os.environ.get("STAGE", "1") == "1"  # 0 = hand the CUDA tensor to ProcessGroupGloo as round 1 did
x = torch.randn(B, T, 3, 224, 224, device="cuda"); m = torch.ones(B, T, dtype=torch.bool, device="cuda"); y = torch.arange(B, device="cuda") % 2
ddist.broadcast_parameters(det)
opt = det.configure_optimizers(0.001)
tr = [p for p in det.parameters() if p.requires_grad]
tag = f"G={int(det.static_graphs)} P={int(det.pipeline_encoder)} S={int(drain)} PRIO={os.environ.get('PRIO', '1')} STAGE={int(ddist._STAGE_GLOO)} MAXQ={os.environ.get('GPU_MAX_HW_QUEUES', 'default')}"
for step in range(8):
    t0 = time.perf_counter(); det.zero_grad(set_to_none=True)
    losses, _, other = det(x, [y], m, train=True, single_task=0); t1 = time.perf_counter()
    (losses[0].mean() + sum(other.values())).backward(); t2 = time.perf_counter()
    if drain:
        torch.cuda.synchronize()
    t2b = time.perf_counter()
    ddist.allreduce_gradients(tr); t3 = time.perf_counter()
    opt.step(); t4 = time.perf_counter()
    print(f"[{tag}] rank {rank} step {step}: fwd {1e3*(t1-t0):.1f} bwd {1e3*(t2-t1):.1f} drain {1e3*(t2b-t2):.1f} "
          f"allreduce {1e3*(t3-t2b):.1f} opt {1e3*(t4-t3):.1f} ms  kfd_queues={kfd_queues()}", flush=True)
torch.cuda.synchronize(); dist.barrier()
