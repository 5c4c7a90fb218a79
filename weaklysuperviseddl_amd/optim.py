"""FlatAdam - torch.optim.Adam semantics (defaults betas=(0.9,0.999), eps=1e-8, no weight decay;
reference TraditionalModel/SegmentationModel.py:91) as ONE kernel launch over flat buffers.

Parameters are re-homed into one contiguous fp32 buffer (``p.data`` become views) and so are their
gradients (``p.grad`` views of one flat buffer), which is also what the data-parallel gradient
all-reduce buckets (``dp.GradBucketReducer``) operate on: 39.6 M parameters = 158.5 MB per replica.
"""
import torch

from . import ops

_ALIGN = 64   # floats; keeps every parameter 256-byte aligned inside the flat buffer


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, grad_scale=1.0):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("FlatAdam: empty parameter list")
        dev = self.params[0].device
        self.lr, self.betas, self.eps, self.grad_scale = lr, betas, eps, grad_scale
        self.step_count = 0
        self.step_dev = torch.zeros(1, device=dev, dtype=torch.int32) if dev.type == "cuda" else None   # device twin
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = n
        self.flat_param = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(n, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(n, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = self.flat_param[off:off + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
                p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
        self.pre_step_hook = None      # e.g. wait for the gradient all-reduce
        self.post_step_hook = None     # e.g. prefetch next step's weight layouts on the side stream
        # direct gradient delivery: the HIP backward kernels write each parameter's gradient straight into
        # its flat_grad slice (ops._sink_of); `_fresh` = not yet written since zero_grad (first write overwrites)
        self._fresh = {id(p): True for p in self.params}
        self.grad_ready_hooks = []     # callables(param): fired when a parameter's gradient has been written
        for p in self.params:
            p._wsdl_grad_sink = self

    def take_fresh(self, p):
        f = self._fresh.get(id(p), False)
        self._fresh[id(p)] = False
        return f

    def grad_ready(self, p):
        for h in self.grad_ready_hooks:
            h(p)

    def zero_grad(self, set_to_none=False):
        if self.flat_grad.is_cuda:
            ops.join_side_stream(self.flat_grad.device)      # a backward without a step may still be writing
        self.flat_grad.zero_()
        for k in self._fresh:
            self._fresh[k] = True
        for p, off in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * off:
                p.grad = self.flat_grad[off:off + p.numel()].view_as(p)

    def step(self):
        if self.pre_step_hook is not None:
            self.pre_step_hook()
        if self.flat_param.is_cuda:
            ops.join_side_stream(self.flat_param.device)     # weight gradients are produced on the side stream
        self.step_count += 1
        ops.bump_param_epoch()          # parameters change behind torch's version counters
        if self.flat_param.is_cuda:
            # the kernel reads the step number from the device: a captured (hipGraph) step replays correctly
            self.step_dev.add_(1)
            ops.adam_step_flat(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.lr,
                               self.betas[0], self.betas[1], self.eps, self.step_count, self.grad_scale,
                               step_dev=self.step_dev)
        else:
            raise ops.WsdlError("FlatAdam.step: parameters are not on the device; there is no CPU fallback")
        if self.post_step_hook is not None:
            self.post_step_hook()
