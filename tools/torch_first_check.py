import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
print("torch", torch.__version__, "hip", torch.version.hip, "cuda avail", torch.cuda.is_available())
x = torch.ones(4, device="cuda") * 2
torch.cuda.synchronize()
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.tensor([1.5], dtype=torch.float64, device="cuda"); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
print("nccl ok", t.item())
import __graft_entry__ as g
g.smoke()
dist.destroy_process_group()
print("torch-first + covahip OK")
