"""Does this RCCL write its NCCL_DEBUG report, and where?  (round 5: bench.py's dp.rccl.debug stayed empty)"""
import os, sys, glob
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
mode = sys.argv[1] if len(sys.argv) > 1 else "file"
os.environ["NCCL_DEBUG"] = "INFO"
os.environ["NCCL_DEBUG_SUBSYS"] = "INIT,GRAPH,TUNING,COLL"
if mode == "file":
    os.environ["NCCL_DEBUG_FILE"] = "/tmp/rccl_probe_%p.log"
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
t = torch.ones(1 << 20, device="cuda")
dist.all_reduce(t)
dist.broadcast(t, 0)
torch.cuda.synchronize()
dist.destroy_process_group()
print("files:", glob.glob("/tmp/rccl_probe_*"), file=sys.stderr)
for f in glob.glob("/tmp/rccl_probe_*"):
    print(open(f, errors="replace").read()[:3000], file=sys.stderr)
