"""Why the RCCL exchange has never carried more than one rank on the one-GPU boxes of this build: RCCL refuses two ranks on one
device.  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/probes/rccl_two_ranks_one_gpu.py
-> "Duplicate GPU detected : rank 0 and rank 1 both on CUDA device 23000" (NCCL 2.26.6 = RCCL of ROCm 7).  The multi-rank
tests therefore run the same exchange over gloo (MFAR_DIST_BACKEND=gloo, ranks sharing the GPU) and RCCL with one rank."""
import os, torch, torch.distributed as dist
r = int(os.environ["RANK"]); w = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=r, world_size=w)
    t = torch.ones(4, device="cuda:0")
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("rank", r, "allreduce", t.tolist(), flush=True)
except Exception as e:
    print("rank", r, "FAILED", type(e).__name__, str(e)[:300], flush=True)
