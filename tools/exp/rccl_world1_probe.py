"""Does a world-size-1 RCCL all-reduce launch a device kernel?  (run under rocprofv3 --kernel-trace --stats)"""
import os
import torch
import torch.distributed as dist
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29561")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.ones(16 << 20, device="cuda")
y = torch.empty_like(x)
for _ in range(5):
    dist.all_reduce(x, op=dist.ReduceOp.AVG)
    dist.all_reduce(x)
    dist.all_gather_into_tensor(y, x)
    dist.broadcast(x, 0)
torch.cuda.synchronize()
dist.destroy_process_group()
print("done")
