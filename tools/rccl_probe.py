import os, sys, time
sys.path.insert(0, os.environ.get("KASF_ROOT", "/root/repo"))
import torch, bench
import torch.distributed as dist
mode = sys.argv[1]
if mode != "none":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", sys.argv[2]); os.environ["RANK"]="0"; os.environ["WORLD_SIZE"]="1"
    if mode == "nccl_lazy":
        dist.init_process_group("nccl")
    elif mode == "nccl_eager":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    elif mode == "nccl_used":
        dist.init_process_group("nccl")
        t = torch.ones(4, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
    elif mode == "gloo":
        dist.init_process_group("gloo")
r = bench.kernel_rooflines(256 * 27 * 17)
print(mode, {k: round(v["seconds"] * 1e6, 1) for k, v in r.items()}, "threads", len(os.listdir("/proc/self/task")))
