"""Time the pose gather alone (RCCL all_gather of (B,8) f64 per rank).  Launch like bench.py:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/gather_overhead.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from msf_loam_amd import dist as mdist

rank = int(os.environ.get("RANK", 0)); local = int(os.environ.get("LOCAL_RANK", 0))
dev = torch.device("cuda", local); torch.cuda.set_device(dev)
if "RANK" in os.environ:
    dist.init_process_group("nccl", device_id=dev)
B = 1024
g = mdist.PoseGather(B, dev)
poses = torch.zeros((B, 7), dtype=torch.float64, device=dev); status = torch.zeros(B, dtype=torch.int32, device=dev)
for _ in range(20): g.all_gather(poses, status)
torch.cuda.synchronize()
for label, fn in (("pack+all_gather", lambda: g.all_gather(poses, status)),
                  ("all_gather only", (lambda: dist.all_gather_into_tensor(g.recv.view(-1), g.send.view(-1))) if dist.is_initialized() else (lambda: None))):
    t0 = time.perf_counter()
    for _ in range(200): fn()
    t_cpu = (time.perf_counter() - t0) / 200
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / 200
    if rank == 0: print(f"{label}: cpu issue {1e6 * t_cpu:.1f} us/call, gpu-complete {1e6 * t_all:.1f} us/call")
if dist.is_initialized(): dist.destroy_process_group()
