"""Does a captured hipMemsetAsync (memset node) re-zero on every replay?  (round 5: the B=1 CAM graph went wrong on its
second replay when the amax pool was zeroed by the library's memset instead of torch.zeros)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from weaklysuperviseddl_amd import ops
from weaklysuperviseddl_amd._lib import lib, check

dev = torch.device("cuda:0")
st = torch.cuda.Stream()
buf = torch.ones(4096, device=dev)
g = torch.cuda.CUDAGraph()
torch.cuda.synchronize()
with torch.cuda.graph(g, stream=st):
    check(lib().wsdl_memset_async(buf.data_ptr(), 0, buf.numel() * 4, torch.cuda.current_stream().cuda_stream))
    buf2 = buf + 1.0
for i in range(3):
    buf.fill_(5.0)
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, "buf max", buf.max().item(), "buf2 max", buf2.max().item(), "(expected 0 and 1)")
