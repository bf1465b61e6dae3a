"""Can two RCCL ranks share ONE GPU?  (torchrun --nproc-per-node 2 scripts/probe/rccl_same_gpu.py)"""
import os, torch, torch.distributed as dist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
try:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    x = torch.full((4,), float(rank), device="cuda")
    out = torch.empty(4 * world, device="cuda")
    dist.all_gather_into_tensor(out, x)
    torch.cuda.synchronize()
    print(f"rank {rank}: all_gather on one GPU ok: {out.tolist()}", flush=True)
    dist.destroy_process_group()
except Exception as e:
    print(f"rank {rank}: FAILED: {type(e).__name__}: {str(e)[:300]}", flush=True)
