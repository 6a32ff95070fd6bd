"""Lab diagnostic: host-side time of each phase of a train step (forward / backward / all-reduce / optimizer)
with two gloo ranks sharing one GPU; G=0/1 toggles the decoder HIP graphs, P=0/1 the pipelined encoder.
Used to isolate the multi-second all-reduce stalls of graphs + pipelining under GPU oversubscription
(Multi-GPU).  Run: torchrun --nproc-per-node 2 tools/lab/dbg_pipe.py"""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, torch.distributed as dist
from dfd_clip_amd.config import default_detector_config
from dfd_clip_amd.detector import Detector
from dfd_clip_amd.weights import random_state_dict
from dfd_clip_amd import dist as ddist
torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank = dist.get_rank()
cfg = default_detector_config(); cfg.architecture = "ViT-B/16"; cfg.decode_mode = "index"; cfg.decode_indices = [6,7,8,9,10,11]; cfg.out_dim=[2]; cfg.losses=["auc_roc"]
T, B = 30, 8
det = Detector(cfg, T, None, precision="bf16"); det.load_state_dict(random_state_dict(cfg, T, seed=0)); det = det.cuda().train()
det.static_graphs = os.environ.get("G", "1") == "1"; det.pipeline_encoder = os.environ.get("P", "1") == "1"; det.inputs_ready = True
x = torch.randn(B, T, 3, 224, 224, device="cuda"); m = torch.ones(B, T, dtype=torch.bool, device="cuda"); y = torch.arange(B, device="cuda") % 2
ddist.broadcast_parameters(det)
opt = det.configure_optimizers(0.001)
tr = [p for p in det.parameters() if p.requires_grad]
for step in range(6):
    t0 = time.perf_counter(); det.zero_grad(set_to_none=True)
    losses, _, other = det(x, [y], m, train=True, single_task=0); t1 = time.perf_counter()
    (losses[0].mean() + sum(other.values())).backward(); t2 = time.perf_counter()
    ddist.allreduce_gradients(tr); t3 = time.perf_counter()
    opt.step(); t4 = time.perf_counter()
    if rank == 0: print(f"step {step}: fwd {1e3*(t1-t0):.1f} bwd {1e3*(t2-t1):.1f} allreduce {1e3*(t3-t2):.1f} opt {1e3*(t4-t3):.1f} ms", flush=True)
torch.cuda.synchronize(); dist.barrier()
