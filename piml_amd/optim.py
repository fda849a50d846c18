"""torch.optim.Adam stepped by ONE launch of this package (piml_adam_step, piml_amd/csrc/adam.hip).

The optimiser of both training loops is `torch.optim.Adam(self.model.parameters(), lr, weight_decay=...)`
(src/models/simulators.py:69-71, :104-129).  On the device PyTorch steps it with `_foreach_add_(state_steps, 1)` + its multi-tensor
fused kernel: 5 + 13 us per step at PINNSF's 22 parameter tensors, launch-bound -- a tenth of a pointwise pre-training step.
`Adam` below is the same optimiser (same constructor, same state: `step`, `exp_avg`, `exp_avg_sq` per parameter, so state_dict()s are
interchangeable) whose step() is one kernel launch per parameter group, BITWISE equal to PyTorch's fused kernel
(tests/test_losses_gpu.py); whatever that launch does not cover (amsgrad, maximize, a tensor learning rate, CPU parameters, other
dtypes, differentiable steps, a gradient scaler) is left to torch.optim.Adam.step itself."""
import ctypes

import torch

from . import _lib


class Adam(torch.optim.Adam):
    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self._tickets = {}

    def _eligible(self, group, params, grads, exp_avgs, exp_avg_sqs, steps):
        if group.get('amsgrad') or group.get('maximize') or group.get('differentiable') or group.get('decoupled_weight_decay') \
                or not isinstance(group['lr'], float):
            return False
        if getattr(self, 'grad_scale', None) is not None or getattr(self, 'found_inf', None) is not None:
            return False
        dev = params[0].device
        if dev.type != 'cuda':
            return False
        for ts in (params, grads, exp_avgs, exp_avg_sqs):
            for t in ts:
                if t.device != dev or t.dtype != torch.float32 or not t.is_contiguous() or t.is_sparse:
                    return False
        return all(s.device == dev and s.dtype == torch.float32 and s.numel() == 1 for s in steps)

    @torch.no_grad()
    def step(self, closure=None):
        work = []
        for group in self.param_groups:
            params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps = [], [], [], [], [], []
            has_complex = self._init_group(group, params, grads, exp_avgs, exp_avg_sqs, max_sqs, steps)
            if not params:
                continue
            if has_complex or not self._eligible(group, params, grads, exp_avgs, exp_avg_sqs, steps):
                return super().step(closure)                  # (nothing has been stepped yet: the groups are only gathered above)
            work.append((group, params, grads, exp_avgs, exp_avg_sqs, steps))
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self._cuda_graph_capture_health_check()
        L = _lib.lib()
        for group, params, grads, exp_avgs, exp_avg_sqs, steps in work:
            n = len(params)
            tab = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() for t in ts])
            sizes = (ctypes.c_longlong * n)(*[p.numel() for p in params])
            beta1, beta2 = group['betas']
            dev = params[0].device
            cache = self.__dict__.setdefault('_tickets', {})      # (an unpickled / deep-copied optimiser comes without it)
            tickets = cache.get(dev)
            if tickets is None:        # zeroed once; every launch leaves them zero
                tickets = cache[dev] = torch.zeros(L.piml_adam_tickets(), dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L.piml_adam_step(tab(params), tab(grads), tab(exp_avgs), tab(exp_avg_sqs), tab(steps), sizes, n,
                                            float(group['lr']), float(beta1), float(beta2), float(group['weight_decay']),
                                            float(group['eps']), tickets.data_ptr(), torch.cuda.current_stream().cuda_stream),
                           'piml_adam_step')
        return loss
