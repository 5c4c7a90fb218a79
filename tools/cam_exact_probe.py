"""Stage-by-stage comparison of the LayerCAM epilogue with torch CPU (diagnostic for the bit-exactness tests)."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from weaklysuperviseddl_amd import ops
dev = torch.device("cuda:0")
T = torch.from_numpy
g = np.load("tests/golden/layercam.npz")
def cnt(a, b):
    a = a.cpu()
    return int((a != b).sum()), float((a - b).abs().max())
for i in range(2):
    acts = [T(g[f"act_layer3_{i}"]), T(g[f"act_layer4_{i}"])]
    grads = [T(g[f"grad_layer3_{i}"]), T(g[f"grad_layer4_{i}"])]
    for l in range(2):
        # one layer, same-size output: the normalised map itself
        for variant, a in (("modular", 1.0), ("notebook", 1.0), ("notebook", 0.5), ("notebook", 2.0), ("modular", 0.5), ("modular", 2.0)):
            ref = oracle.layercam_epilogue([acts[l]], [grads[l]], (14, 14), a, variant)
            cam = ops.layercam_epilogue([acts[l].to(dev)], [grads[l].to(dev)], (14, 14), a, variant)
            print("img", i, "layer", l, variant, a, "map 14x14:", cnt(cam, ref))
            ref = oracle.layercam_epilogue([acts[l]], [grads[l]], (224, 224), a, variant)
            cam = ops.layercam_epilogue([acts[l].to(dev)], [grads[l].to(dev)], (224, 224), a, variant)
            print("img", i, "layer", l, variant, a, "map 224:", cnt(cam, ref))
    for variant, a in (("modular", 1.0), ("notebook", 1.0), ("modular", 0.5), ("modular", 2.0), ("notebook", 0.5), ("notebook", 2.0)):
        ref = oracle.layercam_epilogue(acts, grads, (224, 224), a, variant)
        cam = ops.layercam_epilogue([x.to(dev) for x in acts], [x.to(dev) for x in grads], (224, 224), a, variant)
        print("img", i, "both layers", variant, a, cnt(cam, ref))
