"""Host-side profile of the training step (where do the ~13 ms of issue time per step go?): cProfile over N eager steps."""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda", 0)
model, opt, step, eager = bench.build_workload("cfg2", 16, 256, dev, 0, graph=False)
for _ in range(5):
    step()
torch.cuda.synchronize()
N = int(os.environ.get("N", "20"))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("tottime", "cumtime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue()[:9000])
